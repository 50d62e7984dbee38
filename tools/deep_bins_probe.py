"""PhiX-like stress: n reads over a tiny genome -> every dictionary bin holds hundreds to thousands of reads."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import spring_amd
# extra fields of ReorderOpts for A/B runs, as tools/scale_probe.py: SP_OPTS="long_min=512,long_split=128"
XO = {k: (tuple(int(x) for x in v.split(":")) if ":" in v or k.startswith("plan") else int(v))
      for k, v in (kv.split("=") for kv in __import__("os").environ.get("SP_OPTS", "").split(",") if kv)}
for a in sys.argv[1:]:
    n, L, G, K = [int(x) for x in a.split(",")]
    t0 = time.perf_counter()
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=8, **XO)) as s:
        s.load_synth(n, L, G, 21, 10000)
        s.run()
        st = s.stats()
    print("n=%d L=%d G=%d K=%d wall=%.3fs dict=%.1f chains=%.1f ms rounds=%d unmatched=%d single=%d long=%d numkeys=%s" % (
        n, L, G, K, time.perf_counter() - t0, st["ms_dict"], st["ms_chains"], st["rounds"], st["unmatched"], st["n_single"], st["long_searches"], st["numkeys"]), flush=True)
