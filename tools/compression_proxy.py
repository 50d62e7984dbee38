"""How the number of chains K changes what the encoder has to store: reorder + encoder on the GPU for several K,
then a general-purpose compressor (xz, preset 6) over each output stream as a stand-in for BSC (not in this image).
usage: compression_proxy.py [n_reads] [read_len] [coverage]"""
import lzma
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import spring_amd  # noqa: E402
from spring_amd.encoder import EncoderStage  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cov = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
G = int(n * L / cov)
xz = lambda b: len(lzma.compress(b, preset=6))  # noqa: E731
print("n=%d L=%d coverage=%.0f  (sizes in bytes after xz -6; bits/base = total*8/(n*L))" % (n, L, cov))
print("%8s %9s %9s %9s | %9s %9s %9s %9s %9s %9s | %10s %9s" % (
    "K", "contigs", "single", "unalign", "seq", "pos", "noise", "noisepos", "rc", "unalign", "total", "bits/base"))
for K in [int(x) for x in os.environ.get("KS", "1,16,256,4096,0").split(",")]:
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=1)) as st:
        st.load_synth(n, L, G, 5)
        st.run()
        s = st.stats()
        with EncoderStage() as enc:
            info = enc.encode(st)
            e = enc.streams()
            packed, tails = enc.seq_packed()
    pos = e["pos"].astype(np.int64)
    dpos = np.diff(pos, prepend=0)                      # the reference stores position differences downstream
    sizes = [xz(packed), xz(dpos.astype(np.int32).tobytes()), xz(e["noise"]), xz(e["noisepos"].tobytes()),
             xz(e["rc"].tobytes()), xz(e["unaligned"])]
    tot = sum(sizes)
    print("%8s %9d %9d %9d | %9d %9d %9d %9d %9d %9d | %10d %9.4f" % (
        K if K else "auto", info["num_contigs"], s["n_single"], info["n_total"] - info["n_aligned"], *sizes, tot,
        tot * 8.0 / (n * L)), flush=True)
