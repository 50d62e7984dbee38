"""One-off parity check at 10 M x 150 bp: GPU reorder (auto chains, 8 output sets) == rounds oracle, GPU encoder ==
encoder oracle on the same streams.  Takes a few minutes of CPU for the oracle side.
  python tools/parity_10M.py [reads] [phases]     phases: chain groups (1 / 2; default: what the library chooses at this
                                                  chain count -- two groups from 16 384 chains on, i.e. from 16.8 M reads)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

import spring_amd  # noqa: E402
from helpers import ENC_KEYS, KEYS  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from spring_amd.encoder import EncoderStage  # noqa: E402

n, L = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000, 150
G = n * L // 25
K = max(1, min(65536, n >> 10))
PH = int(sys.argv[2]) if len(sys.argv) > 2 else (2 if K >= 16384 else 1)
oracle_rounds = (lambda *a: po.reorder_rounds_ph(*a)) if PH == 2 else (lambda *a: po.reorder_rounds(*a))
dna = spring_amd.synth_dna_host(n, L, G, 3)
read, ln = po.load_dna(dna, n, L)
with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=8, collect_stats=True, phases=PH)) as st:
    st.load_dna(dna, n, L)
    st.run()
    got = st.streams()
    gst = st.stats()
    with EncoderStage() as enc:
        enc.encode(st)
        ge = enc.streams()
t0 = time.time()
want = oracle_rounds(read, ln, L, K, 8)
print("rounds oracle: %.1f s" % (time.time() - t0), flush=True)
for k in KEYS:
    assert np.array_equal(got[k], want[k]), k
assert np.array_equal(got["tid_off"], want["tid_off"])
for k in ("probes", "keyok", "cands", "hits", "unmatched"):
    assert gst[k] == want["stats"][k], (k, gst[k], want["stats"][k])
print("reorder: %d reads, K=%d, %d chain group(s): streams and work counters identical" % (n, K, PH), flush=True)
# the production build (no counters): the kernel the library picks at this chain count (four chains per wavefront from
# 49 152 chains on, one below) and the other one
for fused, what in ((0, "the library's choice"), (3, "four chains per wavefront"), (2, "one chain per wavefront")):
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=8, fused=fused, phases=PH)) as st:
        st.load_dna(dna, n, L)
        st.run()
        got2 = st.streams()
    for k in KEYS:
        assert np.array_equal(got2[k], want[k]), ("production", fused, k)
    assert np.array_equal(got2["tid_off"], want["tid_off"]) and got2["stats"]["lost"] == want["stats"]["lost"]
    print("reorder, production round kernel (%s): streams identical" % what, flush=True)
t0 = time.time()
we = po.encode(read, ln, L, want, num_thr=8)
print("encoder oracle: %.1f s" % (time.time() - t0), flush=True)
for k in ENC_KEYS:
    a, b = ge[k], we[k]
    assert (a == b) if not isinstance(a, np.ndarray) else np.array_equal(a, b), k
print("encoder: %d contigs, %d aligned singletons: every stream identical" % (we["num_contigs"], we["matched_s"]))
