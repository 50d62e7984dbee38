"""Parity on deep-coverage pools (the regime of bin compaction, the balanced bin scan and resumed searches):
GPU reorder in the PRODUCTION build (no work counting: that is the build with those mechanisms) == rounds oracle,
and the counting build == rounds oracle including the reference-equivalent work counters.
usage: parity_deep.py reads,readlen,genome,chains[,gen] ...   (gen: a genome-like pool -- repeat families, tandem repeats,
low-complexity runs; chains = 0 then means the library's own choice, and the oracle runs with that count)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

import spring_amd  # noqa: E402
from helpers import KEYS  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

for a in sys.argv[1:]:
    f = a.split(",")
    n, L, G, K = [int(x) for x in f[:4]]
    gen = len(f) > 4 and f[4] == "gen"
    if not gen:
        K = K or max(1, min(65536, n >> 10))
    dna = spring_amd.synth_dna_host(n, L, G, 21, 10000 | spring_amd.SYNTH_GENOMIC) if gen else spring_amd.synth_dna_host(n, L, G, 21)
    read, ln = po.load_dna(dna, n, L)
    outs = []
    for stats in (False, True):
        with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=8, collect_stats=stats)) as st:
            st.load_dna(dna, n, L)
            st.run()
            outs.append((st.streams(), st.stats()))
    K = outs[0][1]["chains"]
    A = int(outs[0][1]["alternatives"])  # candidates per proposal the library chose (2 on contended pools)
    assert outs[1][1]["chains"] == K and outs[1][1]["alternatives"] == A
    t0 = time.time()
    want = po.reorder_rounds(read, ln, L, K, 8, alternatives=A)
    for (got, gst), what in zip(outs, ("production build", "counting build")):
        for k in KEYS:
            assert np.array_equal(got[k], want[k]), (what, k)
        assert np.array_equal(got["tid_off"], want["tid_off"]), what
    for k in ("probes", "keyok", "cands", "hits", "unmatched"):
        assert outs[1][1][k] == want["stats"][k], (k, outs[1][1][k], want["stats"][k])
    print("n=%d L=%d genome=%d (%.0fx) K=%d, %d candidate(s) per proposal: production and counting builds identical to the rounds oracle (%.0f s of oracle); "
          "rounds %d, lost proposals %d, %.1f candidate comparisons per read, %d searches finished by k_long (%d split), production chains stage %.1f ms"
          % (n, L, G, n * L / G, K, A, time.time() - t0, outs[0][1]["rounds"], outs[0][1]["lost"],
             want["stats"]["cands"] / n, outs[0][1]["long_searches"], outs[0][1]["long_splits"], outs[0][1]["ms_chains"]), flush=True)
