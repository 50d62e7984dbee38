"""Genome-like pools (SPRING_SYNTH_GENOMIC): the stage under several kernel choices, the dictionary's bin-size profile and the
reference-equivalent work.  tools/genomic_probe.py <reads> [coverage]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spring_amd as sa
n = int(sys.argv[1]); cov = int(sys.argv[2]) if len(sys.argv) > 2 else 25
L = 150; G = n * L // cov
VARS = json.loads(os.environ.get("GP_VARS", "null")) or [
    dict(), dict(deep_bins=1), dict(deep_bins=1, long_budget=8), dict(deep_bins=1, long_budget=8, num_chains=131072),
    dict(deep_bins=1, num_chains=131072), dict(deep_bins=1, num_chains=262144), dict(deep_bins=1, long_budget=8, num_chains=262144)]
for kw in VARS:
    kw = dict(kw)
    K = kw.pop("num_chains", 0)
    t0 = time.perf_counter()
    with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=8, **kw)) as s:
        s.load_synth(n, L, G, 11, 10000 | sa.SYNTH_GENOMIC)
        s.run()
        st = s.stats()
    print("%-70s chains=%8.1f ms rounds=%5d K=%d lost=%d long=%d unmatched=%d single=%d cands/read=%.1f probes/read=%.1f keyok/read=%.1f hits=%d iterations=%d lost=%d" % (
        json.dumps(dict(kw, num_chains=K)), st["ms_chains"], st["rounds"], st["chains"], st["lost"], st["long_searches"], st["unmatched"],
        st["n_single"], st["cands"] / n, st["probes"] / n, st["keyok"] / n, st["hits"], st["iterations"], st["lost"]), flush=True)
