import sys, time, ctypes as C
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch, numpy as np
import spring_amd
from spring_amd import _lib
L_ = _lib.lib()
# extra fields of ReorderOpts for A/B runs: SP_OPTS="table_mode=2,fused=3,plan0=4:8:16" (same results whatever they are)
XO = {k: (tuple(int(x) for x in v.split(":")) if ":" in v or k.startswith("plan") else int(v))
      for k, v in (kv.split("=") for kv in __import__("os").environ.get("SP_OPTS", "").split(",") if kv)}
def run(n, L, K, stats=False, timed=False, rps=0, err=10000, repeats=False, cov=25, genomic=False):
    G = max(n * L // cov, 4 * L)
    nb = L_.spring_synth_dna_bytes(n, L)
    buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
    rc = L_.spring_synth_dna_device(C.c_void_p(buf.data_ptr()), n, L, G, 11, err | (0x80000000 if repeats else 0) | (0x20000000 if genomic else 0)); assert rc == 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=8, collect_stats=stats, time_search=timed, rounds_per_sync=rps, **XO))
    s.load_dna_device(buf.data_ptr(), nb, n, L, True)
    s.run()
    t1 = time.perf_counter()
    st = s.stats()
    s.close()
    print("n=%d L=%d K=%d wall=%.3fs  unpack=%.1f dict=%.1f chains=%.1f final=%.1f ms rounds=%d unmatched=%d single=%d Mreads/s=%.2f search_ms=%.1f launches=%d lost=%d long=%d splits=%d dev=%.1fGB" % (
        n, L, K, t1-t0, st["ms_unpack"], st["ms_dict"], st["ms_chains"], st["ms_finalize"], st["rounds"], st["unmatched"], st["n_single"], n/(t1-t0)/1e6, st["ms_search_kernel"], st["search_launches"], st["lost"], st["long_searches"], st["long_splits"], st["device_bytes"]/1e9), flush=True)
    return st
for a in sys.argv[1:]:
    f = a.split(",")
    n, L, K = [int(x) for x in f[:3]]
    run(n, L, K, err=int(f[3]) if len(f) > 3 else 10000, repeats=len(f) > 4 and f[4] == "rep",
        cov=int(f[5]) if len(f) > 5 else 25, genomic=len(f) > 4 and f[4] == "gen")  # gen: genome with repeat families
