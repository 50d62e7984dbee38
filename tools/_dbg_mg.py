import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import spring_amd as sa
from helpers import named_set, KEYS
from spring_amd.pool import VirtualPool
for name, K in (("syn5k_150", 64), ("var2k", 24), ("heavy", 48)):
    dna, n, L = named_set(name)
    T = 3
    single = sa.reorder_dna(dna, n, L, sa.ReorderOpts(num_chains=K, num_thr=T, collect_stats=True))
    vp = VirtualPool(1, K, T, collect_stats=True)
    got = vp.run(lambda s: s.load_dna(dna, n, L))
    vp.close()
    print(name, "rounds", single["stats"]["rounds"], got["rounds"], "unmatched", single["stats"]["unmatched"], got["per_rank_stats"][0]["unmatched"],
          "nsingle", len(single["order_s"]), len(got["order_s"]), "nmatched", len(single["order"]), len(got["order"]))
    for k in KEYS:
        a, b = single[k], got[k]
        if len(a) != len(b) or not np.array_equal(a, b):
            m = min(len(a), len(b))
            d = np.nonzero(a[:m] != b[:m])[0]
            print("  differs", k, len(a), len(b), "first diffs at", d[:10], a[d[:10]], b[d[:10]])
    print("  tid_off_s", single["tid_off_s"], got["tid_off_s"])
