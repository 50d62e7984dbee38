import sys, os, json
sys.path.insert(0, os.getcwd())
import spring_amd
n, L = 20000000, 150
for cov in (25, 40, 60, 100, 200):
    G = n * L // cov
    out = {}
    for name, kw in (("auto", {}), ("trim", {"deep_bins": 1}), ("mc", {"deep_bins": -1})):
        best = None
        for it in range(2):
            with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_chains=19531, num_thr=8, **kw)) as s:
                s.load_synth(n, L, G, 11, 10000)
                s.run()
                st = s.stats()
            best = st["ms_chains"] if best is None else min(best, st["ms_chains"])
        out[name] = round(best, 1)
    rk = sum(st["dict_numreads"]) / max(sum(st["numkeys"]), 1)
    print("cov=%d reads/key=%.3f lost=%d %s" % (cov, rk, st["lost"], json.dumps(out)), flush=True)
