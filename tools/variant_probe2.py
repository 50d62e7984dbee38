"""Round kernel by pool size and coverage: four chains per wavefront (mc, plan 4,8,16) against one chain per wavefront
(one, plan 8,16) and its deep-bin variant (trim, plan 8,16); default chain counts; chains stage in ms, best of two warm runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spring_amd
L = 150
VARS = [("mc", dict(deep_bins=-1), "4,8,16"), ("one", dict(fused=2, deep_bins=-1), "8,16"), ("trim", dict(deep_bins=1), "8,16")]
for n in [int(x) for x in sys.argv[1].split(",")]:
    for cov in [int(x) for x in sys.argv[2].split(",")]:
        G = n * L // cov
        out = {}
        for name, kw, plan in VARS:
            best = None
            for it in range(3):
                with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=0, num_chains=0, num_thr=8, plan0=tuple(int(x) for x in plan.split(",")), **kw)) as s:
                    s.load_synth(n, L, G, 11, 10000)
                    s.run()
                    st = s.stats()
                if it:
                    best = st["ms_chains"] if best is None else min(best, st["ms_chains"])
            out[name] = round(best, 1)
        rk = sum(st["dict_numreads"]) / max(sum(st["numkeys"]), 1)
        print("n=%d cov=%d chains=%d reads/key=%.3f %s" % (n, cov, st["chains"], rk, json.dumps(out)), flush=True)
