"""BASELINE config 4 (100 M x 150 bp paired-end, pe_encode path preserved) on one MI355X.
  part 1 (parity, default 1 M pairs): paired synthetic pool -> reorder (auto chains, 8 output sets) == rounds oracle;
          encoder stage == encoder oracle; spring_order_pe_encode == the REAL reference pe_encode.cpp
          (oracle/_ref/ref_order) on the encoder's read_order.
  part 2 (at size, default 50 M pairs = 100 M reads): the same pipeline on the device; size-independent property of
          pe_encode (pe_encode.cpp:24-84): file-1 reads keep their reordered rank, every mate sits exactly n/2 behind
          its file-1 read in the decompressed order.
usage: pe_config4.py [pairs_parity] [pairs_full]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import spring_amd  # noqa: E402
from helpers import ENC_KEYS, KEYS  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from spring_amd import order_ops as oo  # noqa: E402
from spring_amd.encoder import EncoderStage  # noqa: E402

L = 150
pairs_parity = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pairs_full = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000


def pipeline(n, K, T, stats=False, want_dna=False):
    G = n * L // 25
    t0 = time.perf_counter()
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=T, collect_stats=stats)) as st:
        st.load_synth(n, L, G, 17, 10000 | spring_amd.SYNTH_PAIRED)
        st.run()
        rs = st.stats()
        streams = st.streams() if stats else None
        dna = st.download_dna() if want_dna else None
        with EncoderStage() as enc:
            info = enc.encode(st)
            es = enc.streams()
    t1 = time.perf_counter()
    new_order, ms = oo.pe_encode(es["order"])
    return dict(rs=rs, streams=streams, dna=dna, info=info, es=es, new_order=new_order, wall=t1 - t0, pe_ms=ms)


if pairs_parity:
    n = 2 * pairs_parity
    K, T = max(1, min(65536, n >> 10)), 8
    r = pipeline(n, K, T, stats=True, want_dna=True)
    read, ln = po.load_dna(r["dna"], n, L)
    t0 = time.time()
    want = po.reorder_rounds(read, ln, L, K, T)
    print("rounds oracle: %.1f s" % (time.time() - t0), flush=True)
    for k in KEYS:
        assert np.array_equal(r["streams"][k], want[k]), k
    assert np.array_equal(r["streams"]["tid_off"], want["tid_off"])
    for k in ("probes", "keyok", "cands", "hits", "unmatched"):
        assert r["rs"][k] == want["stats"][k], (k, r["rs"][k], want["stats"][k])
    print("reorder : %d pairs (%d reads), K=%d: streams + work counters identical to the rounds oracle" % (pairs_parity, n, K), flush=True)
    we = po.encode(read, ln, L, want, num_thr=T)
    for k in ENC_KEYS:
        a, b = r["es"][k], we[k]
        assert (a == b) if not isinstance(a, np.ndarray) else np.array_equal(a, b), k
    print("encoder : %d contigs, every stream identical to the encoder oracle" % we["num_contigs"], flush=True)
    if po.ref_order_bin():
        ref = po.ref_order("pe_encode", r["es"]["order"])
        assert np.array_equal(r["new_order"], ref)
        print("pe_encode: identical to the REAL reference pe_encode.cpp on the encoder's read_order (%d entries)" % len(ref), flush=True)
    else:
        assert np.array_equal(r["new_order"], po.pe_encode(r["es"]["order"]))
        print("pe_encode: identical to the oracle twin (oracle/_ref/ref_order not present)", flush=True)

if pairs_full:
    n = 2 * pairs_full
    r = pipeline(n, 0, 8)
    rs, info = r["rs"], r["info"]
    print("at size : %d pairs = %d reads x %d bp: reorder %.1f ms (unpack %.1f dict %.1f chains %.1f final %.1f; %d rounds, %d singletons), "
          "encoder %.1f ms, pe_encode kernels %.2f ms, wall incl. synth %.2f s -> %.1f Mreads/s through the reorder stage"
          % (pairs_full, n, L, rs["ms_total"], rs["ms_unpack"], rs["ms_dict"], rs["ms_chains"], rs["ms_finalize"], rs["rounds"],
             rs["n_single"], info["ms_device"], r["pe_ms"], r["wall"], n / rs["ms_total"] / 1e3), flush=True)
    order, new = r["es"]["order"], r["new_order"]
    half = n // 2
    assert np.array_equal(np.sort(order), np.arange(n, dtype=np.uint32)), "read_order is not a permutation"
    assert np.array_equal(np.sort(new), np.arange(n, dtype=np.uint32)), "pe_encode output is not a permutation"
    f1 = order < half
    assert np.array_equal(new[f1], np.arange(half, dtype=np.uint32)), "file-1 reads do not keep their reordered rank"
    pos_of = np.empty(n, np.uint32)
    pos_of[order] = np.arange(n, dtype=np.uint32)             # original read -> reordered position
    mates = order[~f1] - half                                  # the file-1 partner of every file-2 read, in reordered order
    assert np.array_equal(new[~f1], new[pos_of[mates]] + half), "a mate is not n/2 behind its file-1 read"
    print("at size : pe_encode properties hold (permutation; file-1 ranks kept; every mate exactly n/2 behind its read)", flush=True)
