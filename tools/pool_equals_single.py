#!/usr/bin/env python3
"""A pool over G virtual ranks against one context on the same pool, library defaults everywhere (chain count, kernel variants,
chain groups, candidates per proposal): every stream and the per-tid offsets must be equal.
usage: pool_equals_single.py reads readlen genome G [err_ppm|flags]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spring_amd
from spring_amd.pool import VirtualPool

n, L, gen, G = (int(x) for x in sys.argv[1:5])
err = int(sys.argv[5], 0) if len(sys.argv) > 5 else 10000
with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_thr=8)) as s:
    s.load_synth(n, L, gen, 21, err)
    want = s.run().streams()
K = int(want["stats"]["chains"])
vp = VirtualPool(G, K, 8)
try:
    got = vp.run(lambda st: st.load_synth(n, L, gen, 21, err))
finally:
    vp.close()
ps = got["per_rank_stats"][0]
for k in ("order", "rc", "flag", "pos", "rlen", "order_s", "tid_off", "tid_off_s"):
    assert np.array_equal(got[k], want[k]), k
print("n=%d genome=%d G=%d: chains %d, groups %d (pool %d), candidates %d (pool %d), rounds %d / %d: pool == single context"
      % (n, gen, G, K, want["stats"]["phases"], ps["phases"], want["stats"]["alternatives"], ps["alternatives"], want["stats"]["rounds"], got["rounds"]))
