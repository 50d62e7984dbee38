"""Which round kernel and probe plan wins at which coverage (20 M x 150 bp, chains stage in ms, best of two warm runs).
mc = four chains per wavefront (k_round_mc), one = one chain per wavefront (k_round), trim = its deep-bin variant."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spring_amd
n, L = 20000000, 150
VARS = [("mc 4,8,16", dict(deep_bins=-1), "4,8,16"), ("mc 8,16", dict(deep_bins=-1), "8,16"),
        ("one 8,16", dict(fused=2, deep_bins=-1), "8,16"), ("one 4,8,16", dict(fused=2, deep_bins=-1), "4,8,16"),
        ("trim 8,16", dict(deep_bins=1), "8,16"), ("trim 4,16", dict(deep_bins=1), "4,16"), ("trim 4,8,16", dict(deep_bins=1), "4,8,16"),
        ("trim 2,6,8,16", dict(deep_bins=1), "2,6,8,16")]
for cov in [int(x) for x in (sys.argv[1:] or ["60", "100", "200", "400"])]:
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
    print("cov=%d chains=%d %s" % (cov, st["chains"], json.dumps(out)), flush=True)
