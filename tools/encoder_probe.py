"""Encoder stage (row f2) at scale: synthetic reads -> reorder -> encode, phase times from HIP events.
usage: encoder_probe.py [n_reads] [read_len] [coverage] [check]
`check` decodes the streams on the host (vectorised) and compares every read with the input."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import spring_amd  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import decode_fixed_len  # noqa: E402
from spring_amd.encoder import EncoderStage  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 150
cov = float(sys.argv[3]) if len(sys.argv) > 3 else 50.0
check = len(sys.argv) > 4 and sys.argv[4] == "check"
G = int(n * L / cov)
names = ("contigs", "sort", "consensus", "pool+dict", "align", "merge", "noise", "tail")
with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=0, num_thr=8)) as st:
    st.load_synth(n, L, G, 7)
    t0 = time.time()
    st.run()
    t1 = time.time()
    s = st.stats()
    print("reorder: n=%d L=%d matched=%d single=%d  %.1f ms" % (n, L, s["n_matched"], s["n_single"], (t1 - t0) * 1e3))
    with EncoderStage() as enc:
        for rep in range(2):
            t0 = time.time()
            info = enc.encode(st)
            t1 = time.time()
            print("encode pass %d: wall %.1f ms, device %.1f ms (%.1f Mreads/s)  contigs=%d seq_len=%d aligned_s=%d "
                  "noisepos=%d passes=%d max_bin=%d" % (rep, (t1 - t0) * 1e3, info["ms_device"], n / info["ms_device"] / 1e3,
                                                        info["num_contigs"], info["seq_len"], info["matched_s"],
                                                        info["n_noisepos"], info["align_passes"], info["max_bin"]))
            print("   " + "  ".join("%s %.1f" % (a, b) for a, b in zip(names, info["ms_phase"])))
        if check:
            t0 = time.time()
            e = enc.streams()
            dna = st.download_dna() if hasattr(st, "download_dna") else None
            print("download %.1f s" % (time.time() - t0))
            na = len(e["pos"])
            reads = decode_fixed_len(e, L)
            # original reads from the synthetic generator
            want = np.frombuffer(spring_amd.synth_dna_host(n, L, G, 7), np.uint8)
            rec = 2 + (L + 3) // 4
            body = want.reshape(n, rec)[:, 2:]
            j = np.arange(L)
            orig = np.frombuffer(b"AGCT", np.uint8)[(body[:, j >> 2] >> (2 * (j & 3))) & 3]
            assert np.array_equal(reads, orig[e["order"][:na]]), "decoded reads differ from the input"
            print("decode check passed for %d aligned reads (%d unaligned)" % (na, len(e["order"]) - na))
