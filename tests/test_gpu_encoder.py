"""GPU parity tests of the encoder stage: spring_encoder_* through the C ABI against
oracle/encoder_oracle.c on identical inputs.  Bar: every output stream bit-exact."""
import numpy as np
import pytest

from helpers import (decode_reads, interleave_order_N, make_N_reads, named_set, read_strings, same_encoding,
                     unpack_dnaN)
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _run(name, K, T, nN=0, deep=0, seed=1, split_tables=False):
    import spring_amd
    from spring_amd.encoder import EncoderStage
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    strs = read_strings(read, ln) if (nN or deep) else []
    Nreads = make_N_reads(strs, nN, seed, deep=deep)
    dnaN = po.pack_dnaN(Nreads)
    order_N = interleave_order_N(n, len(Nreads), seed + 7)
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=T)) as st:
        st.load_dna(dna, n, L)
        st.run()
        streams = st.streams()
        with EncoderStage() as enc:
            if split_tables:
                assert enc._L.spring_encoder_set_split_tables(enc._h, 1) == 0
            info = enc.encode(st, dnaN, order_N)
            got = enc.streams()
            packed, tails = enc.seq_packed()
    want = po.encode(read, ln, L, streams, num_thr=T, dnaN=dnaN, order_N=order_N)
    return got, want, info, (packed, tails), (read, ln, Nreads, order_N, n)


@pytest.mark.parametrize("name", ["syn2k_100", "syn5k_150", "syn3k_64", "syn2k_251", "syn1k_511", "syn2k_20", "var2k",
                                  "var_short", "var_long", "heavy", "repeat10k", "dups", "test_1+2", "one", "empty"])
@pytest.mark.parametrize("K,T", [(1, 1), (7, 3)])
def test_encoder_bit_exact_vs_oracle(name, K, T):
    got, want, info, _, _ = _run(name, K, T)
    same_encoding(got, want, name)


@pytest.mark.parametrize("name", ["syn5k_150", "var2k", "syn2k_251", "repeat10k", "syn3k_64"])
def test_encoder_with_N_reads(name):
    got, want, info, _, (read, ln, Nreads, order_N, n) = _run(name, 16, 2, nN=400, seed=3)
    same_encoding(got, want, name)
    assert info["matched_N"] > 0
    # decode -> the original reads (decompress.cpp:236-266)
    dec = decode_reads(got)
    cum = np.zeros(n + len(Nreads), bool)
    cum[order_N] = True
    clean_pos = np.flatnonzero(~cum)
    strs = read_strings(read, ln)
    orig = {int(clean_pos[i]): strs[i] for i in range(n)}
    orig.update({int(order_N[i]): Nreads[i] for i in range(len(Nreads))})
    for o, s in dec.items():
        assert orig[o] == s
    na = len(got["pos"])
    for k, s in enumerate(unpack_dnaN(got["unaligned"])):
        assert orig[int(got["order"][na + k])] == s
    assert sorted(got["order"].tolist()) == list(range(n + len(Nreads)))


@pytest.mark.parametrize("name", ["syn5k_150", "syn2k_100"])
def test_encoder_deep_bins_need_the_fixed_point(name):
    """> MAX_SEARCH_ENCODER reads in one bin: the 1000-read window moves as earlier probes empty the bin."""
    got, want, info, _, _ = _run(name, 8, 2, nN=50, deep=3500, seed=5)
    assert info["max_bin"] > 1000 and info["align_passes"] >= 2
    same_encoding(got, want, name)


def test_seq_packing_matches_pack_compress_seq():
    import ctypes as C
    got, want, info, (packed, tails), _ = _run("syn5k_150", 9, 4)
    L = po.lib()
    L.orc_pack_seq.restype = C.c_uint64
    L.orc_pack_seq.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p]
    off, exp, exp_tails = 0, b"", []
    for t, sl in enumerate(want["seq_len_tid"]):
        sl = int(sl)
        seg = want["seq"][off:off + sl]
        buf = np.zeros(max(sl // 4, 1), np.uint8)
        tail = np.zeros(4, np.uint8)
        nb = L.orc_pack_seq(seg, sl, buf.ctypes.data, tail.ctypes.data)
        exp += buf[:nb].tobytes()
        exp_tails.append(tail[:sl % 4].tobytes().decode())
        off += sl
    assert packed == exp and tails == exp_tails


def test_encoder_100k_many_chains():
    import readsets as rs
    import spring_amd
    from spring_amd.encoder import EncoderStage
    n, L = 100000, 100
    a = rs.np_reads(321, n * L // 30, n, L, 0.01)
    dna = rs.pack_fixed(a)
    read, ln = po.load_dna(dna, n, L)
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=64, num_thr=4)) as st:
        st.load_dna(dna, n, L)
        st.run()
        streams = st.streams()
        with EncoderStage() as enc:
            enc.encode(st)
            got = enc.streams()
    want = po.encode(read, ln, L, streams, num_thr=4)
    same_encoding(got, want, "100k")


def test_reorder_encode_run_file_contract(tmp_path):
    """spring_reorder_encode_run: input_clean_1.dna + input_N.dna + read_order_N.bin in, the encoder's files out."""
    import ctypes as C
    import os

    import spring_amd
    from spring_amd import _lib
    name, T, K = "syn5k_150", 3, 16
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    Nreads = make_N_reads(read_strings(read, ln), 300, 4)
    dnaN = po.pack_dnaN(Nreads)
    order_N = interleave_order_N(n, len(Nreads), 12)
    d = str(tmp_path)
    open(os.path.join(d, "input_clean_1.dna"), "wb").write(dna)
    open(os.path.join(d, "input_N.dna"), "wb").write(dnaN)
    open(os.path.join(d, "read_order_N.bin"), "wb").write(order_N.tobytes())
    o = spring_amd.ReorderOpts(num_chains=K, num_thr=T).to_c()
    info = _lib.EncoderInfo()
    L_ = _lib.lib()
    rc = L_.spring_reorder_encode_run(d.encode(), L, T, 0, n, 0, n + len(Nreads), C.byref(o), C.byref(info))
    assert rc == 0, L_.spring_reorder_last_error()
    want = po.encode(read, ln, L, po.reorder_rounds(read, ln, L, K, T), num_thr=T, dnaN=dnaN, order_N=order_N)
    rd = lambda f: open(os.path.join(d, f), "rb").read()  # noqa: E731
    assert rd("read_pos.bin") == want["pos"].tobytes()
    assert rd("read_noise.txt") == want["noise"]
    assert rd("read_noisepos.bin") == want["noisepos"].tobytes()
    assert rd("read_order.bin") == want["order"].tobytes()
    assert rd("read_rev.txt") == want["rc"].tobytes()
    assert rd("read_lengths.bin") == want["rlen"].tobytes()
    assert rd("read_unaligned.txt") == want["unaligned"]
    assert int.from_bytes(rd("read_unaligned.txt.count"), "little") == want["len_unaligned"]
    Lo = po.lib()
    Lo.orc_pack_seq.restype = C.c_uint64
    Lo.orc_pack_seq.argtypes = [C.c_char_p, C.c_uint64, C.c_void_p, C.c_void_p]
    off = 0
    for t, sl in enumerate(int(x) for x in want["seq_len_tid"]):
        buf, tail = np.zeros(max(sl // 4, 1), np.uint8), np.zeros(4, np.uint8)
        nb = Lo.orc_pack_seq(want["seq"][off:off + sl], sl, buf.ctypes.data, tail.ctypes.data)
        assert rd("read_seq.bin.%d.tmp" % t) == buf[:nb].tobytes()
        assert rd("read_seq.bin.%d.tail" % t) == tail[:sl % 4].tobytes()
        off += sl
    for gone in ("input_clean_1.dna", "input_N.dna", "read_order_N.bin"):
        assert not os.path.exists(os.path.join(d, gone))


def test_encoder_2M_decode_round_trip():
    """Size-independent property at a size the oracle does not reach: decoding the GPU streams the way the
    decompressor does returns every input read (2 M x 100 bp, auto chains, 8 output files)."""
    import spring_amd
    from helpers import decode_fixed_len
    from spring_amd.encoder import EncoderStage
    n, L = 2_000_000, 100
    G = n * L // 40
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=0, num_thr=8)) as st:
        st.load_synth(n, L, G, 21)
        st.run()
        with EncoderStage() as enc:
            info = enc.encode(st)
            e = enc.streams()
    na = len(e["pos"])
    assert info["n_total"] == n and na == info["n_aligned"] and info["matched_s"] > 0
    reads = decode_fixed_len(e, L)
    body = np.frombuffer(spring_amd.synth_dna_host(n, L, G, 21), np.uint8).reshape(n, 2 + (L + 3) // 4)[:, 2:]
    j = np.arange(L)
    orig = np.frombuffer(b"AGCT", np.uint8)[(body[:, j >> 2] >> (2 * (j & 3))) & 3]
    assert np.array_equal(reads, orig[e["order"][:na]])
    un = unpack_dnaN(e["unaligned"], count=50)   # spot-check the head of the unaligned stream
    for k, s in enumerate(un):
        assert s.encode() == orig[e["order"][na + k]].tobytes()
    assert np.array_equal(np.sort(e["order"]), np.arange(n, dtype=np.uint32))


@pytest.mark.parametrize("paired", [False, True])
def test_fastq_to_encoder_streams_end_to_end(paired):
    """Rows f1 -> hot path -> f2 chained on the GPU: FASTQ text in, encoder streams out; decoding them
    (decompress.cpp:236-266) must return the reads of the FASTQ at their original positions, N reads included.
    Overlapping reads of a small genome so that contigs, aligned singletons and aligned N reads all occur."""
    import spring_amd
    from spring_amd.encoder import EncoderStage
    rng = np.random.default_rng(77 + paired)
    G, L, n = 30000, 120, 12000
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, G)]
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b

    def make(nreads, seed):
        r = np.random.default_rng(seed)
        reads = []
        for i in range(nreads):
            ln = int(r.integers(60, L + 1))
            p = int(r.integers(0, G - ln))
            s = genome[p:p + ln].copy()
            flip = r.random(ln) < 0.01
            s[flip] = np.frombuffer(b"ACGT", np.uint8)[r.integers(0, 4, int(flip.sum()))]
            if r.random() < 0.08:
                s[r.integers(0, ln, 2)] = ord("N")
            if r.random() < 0.5:
                s = comp[s[::-1]]
            reads.append(s.tobytes())
        return reads

    files = [make(n, 1)] + ([make(n, 2)] if paired else [])
    texts = [b"".join(b"@r%d\n%s\n+\n%s\n" % (i, s, b"I" * len(s)) for i, s in enumerate(f)) for f in files]
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=32, num_thr=3)) as st:
        info = st.load_fastq(texts[0], texts[1] if paired else None)
        st.run()
        dnaN, order_N = b"", np.zeros(0, np.uint32)
        for w in range(len(files)):  # merge of input_N.dna(.2) / read_order_N.bin(.2), preprocess.cpp:362-383
            b_, o_ = st.fastq_N(w)
            dnaN += b_
            order_N = np.concatenate([order_N, o_ + np.uint32(w * info["num_reads"][0])])
        with EncoderStage() as enc:
            ei = enc.encode(st, dnaN, order_N)
            e = enc.streams()
            # without N images the encoder takes the N reads the front end left on the device: same streams
            ei2 = enc.encode(st)
            e2 = enc.streams()
    assert ei2["matched_N"] == ei["matched_N"] and ei2["n_total"] == ei["n_total"]
    same_encoding(e2, e, "device-resident N reads")
    allreads = [s.decode() for f in files for s in f]
    assert ei["n_total"] == len(allreads) and ei["matched_s"] > 0 and ei["matched_N"] > 0
    dec = decode_reads(e)
    for o, s in dec.items():
        assert allreads[o] == s
    na = len(e["pos"])
    un = unpack_dnaN(e["unaligned"])
    assert len(un) == len(allreads) - na
    for k, s in enumerate(un):
        assert allreads[int(e["order"][na + k])] == s
    assert sorted(e["order"].tolist()) == list(range(len(allreads)))


@pytest.mark.parametrize("recompress", [False, True])
def test_encoder_run_file_contract_after_reorder_run(tmp_path, recompress):
    """spring_encoder_run as a drop-in for call_encoder: it consumes the per-tid files a reorder stage left in
    temp_dir (here spring_reorder_run's; with recompress=True the gzip members are rewritten with real deflate
    blocks, as boost::iostreams::gzip_compressor produces them) and leaves encoder_main's files."""
    import ctypes as C
    import gzip
    import os

    import spring_amd
    from spring_amd import _lib
    name, T, K = "var2k", 3, 12
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    Nreads = make_N_reads(read_strings(read, ln), 250, 8)
    dnaN = po.pack_dnaN(Nreads)
    order_N = interleave_order_N(n, len(Nreads), 18)
    d = str(tmp_path)
    open(os.path.join(d, "input_clean_1.dna"), "wb").write(dna)
    open(os.path.join(d, "input_N.dna"), "wb").write(dnaN)
    open(os.path.join(d, "read_order_N.bin"), "wb").write(order_N.tobytes())
    L_ = _lib.lib()
    o = spring_amd.ReorderOpts(num_chains=K, num_thr=T).to_c()
    assert L_.spring_reorder_run(d.encode(), L, T, 0, n, 0, C.byref(o)) == 0, L_.spring_reorder_last_error()
    if recompress:
        for t in range(T):
            for f in ("read_rev.txt", "tempflag.txt", "temppos.txt", "read_lengths.bin"):
                p = os.path.join(d, "%s.%d" % (f, t))
                raw = gzip.decompress(open(p, "rb").read())
                open(p, "wb").write(gzip.compress(raw, 6))
    info = _lib.EncoderInfo()
    rc = L_.spring_encoder_run(d.encode(), L, T, n + len(Nreads), n, -1, C.byref(info))
    assert rc == 0, L_.spring_reorder_last_error()
    want = po.encode(read, ln, L, po.reorder_rounds(read, ln, L, K, T), num_thr=T, dnaN=dnaN, order_N=order_N)
    rd = lambda f: open(os.path.join(d, f), "rb").read()  # noqa: E731
    assert rd("read_pos.bin") == want["pos"].tobytes()
    assert rd("read_noise.txt") == want["noise"]
    assert rd("read_noisepos.bin") == want["noisepos"].tobytes()
    assert rd("read_order.bin") == want["order"].tobytes()
    assert rd("read_rev.txt") == want["rc"].tobytes()
    assert rd("read_lengths.bin") == want["rlen"].tobytes()
    assert rd("read_unaligned.txt") == want["unaligned"]
    assert int.from_bytes(rd("read_unaligned.txt.count"), "little") == want["len_unaligned"]
    assert info.matched_s == want["matched_s"] and info.matched_N == want["matched_N"]
    left = sorted(os.listdir(d))
    assert not [f for f in left if f.startswith(("temp", "input_", "read_order_N"))], left
    for t in range(T):
        assert not os.path.exists(os.path.join(d, "read_order.bin.%d" % t))


@pytest.mark.parametrize("name", ["syn5k_150", "var2k"])
def test_split_table_alignment_kernel_agrees(name):
    """spring_encoder_set_split_tables forces the one-thread-per-window kernel (the one used when the two dictionary
    windows differ in length, max_readlen <= 50) on data that normally takes the merged-table kernel."""
    got, want, info, _, _ = _run(name, 16, 2, nN=300, deep=1200, seed=6, split_tables=True)
    same_encoding(got, want, name)


def test_python_mirrors_of_call_encoder(tmp_path):
    """spring_amd.encoder.call_encoder / call_reorder_encoder: same files either way, error behaviour of the
    reference's switch ("Wrong bitset size.")."""
    import os

    import spring_amd
    from spring_amd.encoder import call_encoder, call_reorder_encoder
    from spring_amd.reorder import CompressionParams, ReorderError
    dna, n, L = named_set("syn2k_100")
    outs = []
    for mode in (0, 1):
        d = os.path.join(str(tmp_path), "m%d" % mode)
        os.makedirs(d)
        open(os.path.join(d, "input_clean_1.dna"), "wb").write(dna)
        cp = CompressionParams(L, [n, 0], num_thr=2)
        opts = spring_amd.ReorderOpts(num_chains=8, num_thr=2)
        if mode == 0:
            spring_amd.call_reorder(d, cp, opts)
            info = call_encoder(d, cp, n)
        else:
            info = call_reorder_encoder(d, cp, n, opts)
        outs.append({f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))})
        assert info["n_total"] == n
    assert outs[0] == outs[1] and "read_pos.bin" in outs[0] and "read_seq.bin.1.tmp" in outs[0]
    with pytest.raises(ReorderError):
        call_encoder(str(tmp_path), CompressionParams(600, [1, 0]), 1)
