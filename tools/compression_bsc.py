"""What the number of chains K costs in compressed size, measured with the REAL reference BSC (src/libbsc compiled
in place into oracle/_ref/ref_bsc -- test infrastructure, see oracle/Makefile) instead of round 1's xz stand-in.
reorder + encoder on the GPU for several K, every encoder output stream compressed with spring::bsc::BSC_compress
(block size and parameters of the reference's params.h).  Also the CPU port at 8 free-running threads (the
reference's default -t 8) through the encoder oracle, for the size a `-t 8` run of the reference would give.
usage: compression_bsc.py [n_reads] [read_len] [coverage]        env KS=1,16,256,0  PORT8=1"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import spring_amd  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from spring_amd.encoder import EncoderStage  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cov = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
G = int(n * L / cov)
BSC = po.ref_bsc_bin()
assert BSC, "oracle/_ref/ref_bsc missing (make -C oracle, needs /root/reference)"


def bsc(b: bytes) -> int:
    if not len(b):
        return 0
    with tempfile.TemporaryDirectory() as d:
        a, o = os.path.join(d, "in"), os.path.join(d, "out")
        open(a, "wb").write(b)
        subprocess.run([BSC, a, o], check=True, stdout=subprocess.DEVNULL)
        return os.path.getsize(o)


def sizes_of(e, packed):
    pos = e["pos"].astype(np.int64)
    dpos = np.diff(pos, prepend=0)
    return [bsc(packed), bsc(dpos.astype(np.int32).tobytes()), bsc(bytes(e["noise"])), bsc(e["noisepos"].tobytes()),
            bsc(e["rc"].tobytes()), bsc(bytes(e["unaligned"]))]


print("n=%d L=%d coverage=%.0f  (bytes after the reference's BSC; bits/base = total*8/(n*L))" % (n, L, cov))
print("%10s %9s %9s %9s | %9s %9s %9s %9s %9s %9s | %10s %9s" % (
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
    sz = sizes_of(e, packed)
    tot = sum(sz)
    print("%10s %9d %9d %9d | %9d %9d %9d %9d %9d %9d | %10d %9.4f" % (
        K if K else "auto", info["num_contigs"], s["n_single"], info["n_total"] - info["n_aligned"], *sz, tot,
        tot * 8.0 / (n * L)), flush=True)
if os.environ.get("PORT8", "1") != "0":
    dna = spring_amd.synth_dna_host(n, L, G, 5)
    read, ln = po.load_dna(dna, n, L)
    streams = po.reorder_omp(read, ln, L, 8)
    e = po.encode(read, ln, L, streams, num_thr=8)
    seq = np.frombuffer(bytes(e["seq"]), np.uint8)
    code = np.zeros(256, np.uint8)
    for ch, v in zip(b"ACGT", (0, 1, 2, 3)):
        code[ch] = v
    c = code[seq]
    pad = (-len(c)) % 4
    c = np.concatenate([c, np.zeros(pad, np.uint8)]).reshape(-1, 4)
    packed = (c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)).astype(np.uint8).tobytes()
    sz = sizes_of(e, packed)
    tot = sum(sz)
    print("%10s %9d %9d %9s | %9d %9d %9d %9d %9d %9d | %10d %9.4f" % (
        "port -t 8", e["num_contigs"], len(streams["order_s"]), "-", *sz, tot, tot * 8.0 / (n * L)), flush=True)
