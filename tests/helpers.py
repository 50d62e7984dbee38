"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np

import readsets as rs
from oracle import pyoracle as po

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KEYS = ("order", "rc", "flag", "pos", "rlen", "order_s")


def fastq_clean_reads(path):
    """Reads of a FASTQ without 'N' (what preprocess.cpp:295-304 hands to reorder)."""
    out = []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    for i in range(1, len(lines), 4):
        r = lines[i].strip()
        if b"N" not in r:
            out.append(r)
    return out


def named_set(name):
    """-> (dna bytes, n, max_readlen).  Small sets the oracle finishes in seconds."""
    if name == "test_1":
        reads = fastq_clean_reads(os.path.join(GOLDEN, "test_1.fastq"))
        return rs.pack_var(reads), len(reads), max(len(r) for r in reads)
    if name == "test_1+2":  # paired-end pool: file-1 reads then file-2 reads (reorder.h:233-242)
        r1 = fastq_clean_reads(os.path.join(GOLDEN, "test_1.fastq"))
        r2 = fastq_clean_reads(os.path.join(GOLDEN, "test_2.fastq"))
        reads = r1 + r2
        return rs.pack_var(reads), len(reads), max(len(r) for r in reads)
    if name == "syn2k_100":
        a = rs.np_reads(101, 2000 * 100 // 30, 2000, 100, 0.01)
        return rs.pack_fixed(a), 2000, 100
    if name == "syn5k_150":
        a = rs.np_reads(102, 5000 * 150 // 25, 5000, 150, 0.01)
        return rs.pack_fixed(a), 5000, 150
    if name == "syn20k_100":
        a = rs.np_reads(103, 20000 * 100 // 25, 20000, 100, 0.01)
        return rs.pack_fixed(a), 20000, 100
    if name == "syn3k_64":  # L < 100: dictionary windows shorter than 32 bases (reorder.h:752-759)
        a = rs.np_reads(104, 3000 * 64 // 25, 3000, 64, 0.01)
        return rs.pack_fixed(a), 3000, 64
    if name == "syn2k_251":  # W = 8 limbs
        a = rs.np_reads(105, 2000 * 251 // 25, 2000, 251, 0.005)
        return rs.pack_fixed(a), 2000, 251
    if name == "syn1k_511":  # MAX_READ_LEN: W = 16 limbs, 8 positions per lane in the consensus update
        a = rs.np_reads(113, 1000 * 511 // 25, 1000, 511, 0.004)
        return rs.pack_fixed(a), 1000, 511
    if name == "syn2k_20":  # very short reads: 6-base dictionary windows, maxshift 10
        a = rs.np_reads(114, 2000 * 20 // 40, 2000, 20, 0.0)
        return rs.pack_fixed(a), 2000, 20
    if name == "var_long":  # variable length up to 400: reverse sub-cases with W = 13
        reads = rs.var_length_reads(115, 30000, 1500, 120, 400, 0.005)
        return rs.pack_var(reads), len(reads), max(len(r) for r in reads)
    if name == "var2k":  # variable length 50..150: the three reverse sub-cases + len<=dict.end exclusion
        reads = rs.var_length_reads(106, 12000, 2000, 50, 150, 0.01)
        return rs.pack_var(reads), len(reads), max(len(r) for r in reads)
    if name == "var_short":  # many reads too short for dict 1 / both dicts
        reads = rs.var_length_reads(107, 6000, 1500, 20, 100, 0.01)
        return rs.pack_var(reads), len(reads), max(len(r) for r in reads)
    if name == "heavy":  # one bin with >1000 reads: MAX_SEARCH_REORDER cap + removal encodings
        a = rs.heavy_bin_reads(108, 1500, 1500, 100, 0.02)
        return rs.pack_fixed(a), a.shape[0], 100
    if name == "repeat10k":
        a = rs.np_reads_repeat(109, 40000, 10000, 100, 0.02)
        return rs.pack_fixed(a), 10000, 100
    if name == "dups":  # exact duplicates of few reads
        base = rs.np_reads(110, 3000, 40, 100, 0.0)
        a = np.repeat(base, 30, axis=0)
        np.random.default_rng(5).shuffle(a, axis=0)
        return rs.pack_fixed(a), a.shape[0], 100
    if name == "tandem":  # a 40-base unit repeated 400 times with 4 % point mutations per copy: thousands of distinct keys
        # share the unit's few 16-mers as minimizers -> over-subscribed table lines (TabView::minz, TAG_MARK + redirect)
        rng = np.random.default_rng(117)
        unit = rng.integers(0, 4, 40, dtype=np.uint8)
        g = np.tile(unit, 400)
        m = rng.random(len(g)) < 0.04
        g[m] = (g[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) % 4
        n, L = 6000, 120
        pos = rng.integers(0, len(g) - L + 1, n)
        r = g[pos[:, None] + np.arange(L)[None, :]]
        e = rng.random((n, L)) < 0.01
        r = np.where(e, (r + rng.integers(1, 4, (n, L), dtype=np.uint8)) % 4, r).astype(np.uint8)
        rc = rng.random(n) < 0.5
        r[rc] = (3 - r[rc])[:, ::-1]
        return rs.pack_fixed(np.frombuffer(b"ACGT", dtype=np.uint8)[r]), n, L
    if name == "one":
        a = rs.np_reads(111, 1000, 1, 100, 0.0)
        return rs.pack_fixed(a), 1, 100
    if name == "two_same":
        a = np.repeat(rs.np_reads(112, 1000, 1, 100, 0.0), 2, axis=0)
        return rs.pack_fixed(a), 2, 100
    if name == "empty":
        return b"", 0, 100
    raise KeyError(name)


SMALL_SETS = ["test_1", "test_1+2", "syn2k_100", "syn5k_150", "syn3k_64", "syn2k_251", "syn1k_511", "syn2k_20",
              "var_long", "var2k", "var_short",
              "heavy", "repeat10k", "dups", "tandem", "one", "two_same", "empty"]


def check_invariants(res, read, ln, L, n):
    """Properties every legal reorder output has (any chain count)."""
    order, order_s = res["order"], res["order_s"]
    allr = np.concatenate([order, order_s]).astype(np.int64)
    assert len(allr) == n
    assert np.array_equal(np.sort(allr), np.arange(n)), "output is not a permutation of the clean reads"
    assert set(np.unique(res["rc"]).tolist()) <= {ord("d"), ord("r")}
    assert set(np.unique(res["flag"]).tolist()) <= {ord("0"), ord("1")}
    assert np.array_equal(res["rlen"], ln[order])
    f0 = res["flag"] == ord("0")
    assert np.all(res["pos"][f0] == 0) and np.all(res["rc"][f0] == ord("d"))
    if len(order):
        assert res["flag"][0] == ord("0")
        # every contig has >= 2 reads: a '0' is never followed by another '0' / end of a tid stream
        toff = [int(x) for x in res["tid_off"]]
        for a, b in zip(toff[:-1], toff[1:]):
            if b > a:
                fl = res["flag"][a:b]
                assert fl[0] == ord("0") and fl[-1] == ord("1")
                assert not np.any((fl[:-1] == ord("0")) & (fl[1:] == ord("0")))


# ------------------------------------------------------------------ encoder stage (row f2)

DEC_NOISE = {"A": "CGTN", "C": "AGTN", "G": "TACN", "T": "GCAN", "N": "AGCT"}  # decompress.cpp:664-684
_RC = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def decode_reads(enc):
    """What the decompressor does with the encoder's streams (decompress.cpp:236-266): -> {order: read string}
    for the aligned reads, in their original orientation."""
    seq = enc["seq"].decode()
    noise_lines = enc["noise"].decode().split("\n")
    npos = enc["noisepos"]
    out, k = {}, 0
    for i in range(len(enc["pos"])):
        p, ln = int(enc["pos"][i]), int(enc["rlen"][i])
        r = list(seq[p:p + ln])
        prev = 0
        for ch in noise_lines[i]:
            prev += int(npos[k])
            k += 1
            r[prev] = DEC_NOISE[r[prev]][int(ch)]
        s = "".join(r)
        if chr(enc["rc"][i]) == "r":
            s = "".join(_RC[c] for c in reversed(s))
        out[int(enc["order"][i])] = s
    assert k == len(npos)
    return out


def unpack_dnaN(buf, count=None):
    """read_dnaN_from_bits (util.cpp:350-374) over a whole stream -> list of strings."""
    out, p = [], 0
    while p < len(buf) and (count is None or len(out) < count):
        n = int.from_bytes(buf[p:p + 2], "little")
        p += 2
        body = buf[p:p + (n + 1) // 2]
        p += (n + 1) // 2
        out.append("".join("AGCTN"[(body[j // 2] >> (4 * (j & 1))) & 15] for j in range(n)))
    return out


def decode_dna_fixed(dna: bytes, n: int, L: int):
    """fixed-length .dna record stream (u16 len + 2-bit bases, util.cpp:269-294) -> list of bytes strings."""
    rec = 2 + (L + 3) // 4
    a = np.frombuffer(dna, dtype=np.uint8).reshape(n, rec)[:, 2:]
    codes = np.stack([(a >> s) & 3 for s in (0, 2, 4, 6)], axis=2).reshape(n, -1)[:, :L]
    letters = np.frombuffer(b"AGCT", dtype=np.uint8)[codes]
    return [letters[i].tobytes() for i in range(n)]


def read_strings(read, ln):
    """2-bit limbs -> list of strings (bitsettostring, reorder.h:76-92)."""
    out = []
    lut = np.frombuffer(b"AGCT", dtype=np.uint8)
    for i in range(len(ln)):
        n = int(ln[i])
        j = np.arange(n)
        codes = (read[i][j >> 5] >> (2 * (j & 31)).astype(np.uint64)) & np.uint64(3)
        out.append(lut[codes.astype(np.int64)].tobytes().decode())
    return out


def make_N_reads(strings, count, seed, max_N=3, deep=0):
    """Reads with N for the encoder tests: `count` clean reads re-sampled with 1..max_N bases replaced
    by N (half of them reverse complemented), plus `deep` near-copies of one read whose N sits outside
    both dictionary windows (bins deeper than MAX_SEARCH_ENCODER).  -> list of strings."""
    rng = np.random.default_rng(seed)
    out = []
    strings = [s for s in strings if len(s) > 0]
    if not strings:
        return out
    for _ in range(count):
        s = list(strings[int(rng.integers(len(strings)))])
        for _ in range(int(rng.integers(1, max_N + 1))):
            s[int(rng.integers(len(s)))] = "N"
        s = "".join(s)
        if rng.integers(2):
            s = "".join(_RC[c] for c in reversed(s))
        out.append(s)
    if deep:
        base = strings[int(rng.integers(len(strings)))]
        for k in range(deep):
            s = list(base)
            if len(s) > 50:
                s[45 + int(rng.integers(len(s) - 45))] = "N"
            s = "".join(s)
            if k % 3 == 2:
                s = "".join(_RC[c] for c in reversed(s))
            out.append(s)
    return out


def interleave_order_N(n_clean, nN, seed):
    """Positions of the N reads in the original file (read_order_N.bin): a sorted random subset."""
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(n_clean + nN, size=nN, replace=False)).astype(np.uint32)


ENC_KEYS = ("seq", "seq_len_tid", "pos", "noise", "noisepos", "order", "rlen", "rc", "unaligned", "len_unaligned",
            "matched_s", "matched_N", "num_contigs")


def same_encoding(a, b, what=""):
    for k in ENC_KEYS:
        x, y = a[k], b[k]
        if isinstance(x, np.ndarray):
            assert np.array_equal(x, y), (what, k, len(x), len(y))
        else:
            assert x == y, (what, k)


def decode_fixed_len(e, L):
    """Vectorised decompress.cpp:236-266 for equal-length reads: -> uint8 [n_aligned, L] letters of the aligned
    reads in their original orientation, row i belonging to read e["order"][i]."""
    na = len(e["pos"])
    seq = np.frombuffer(e["seq"], np.uint8)
    reads = seq[e["pos"][:, None].astype(np.int64) + np.arange(L)[None, :]]
    noise = np.frombuffer(e["noise"], np.uint8)
    nl = np.flatnonzero(noise == 10)
    cnt = np.diff(np.concatenate([[-1], nl])) - 1     # substitutions per read
    assert len(cnt) == na and cnt.sum() == len(e["noisepos"])
    rid = np.repeat(np.arange(na), cnt)
    first = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    cs = np.cumsum(e["noisepos"].astype(np.int64))
    col = cs - np.repeat(np.concatenate([[0], cs])[first], cnt)   # position deltas -> positions inside the read
    codes = noise[noise != 10] - ord("0")
    lut = np.zeros((256, 4), np.uint8)
    for r, row in DEC_NOISE.items():
        lut[ord(r)] = np.frombuffer(row.encode(), np.uint8)
    reads[rid, col] = lut[reads[rid, col], codes]
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    rc = e["rc"] == ord("r")
    reads[rc] = comp[reads[rc][:, ::-1]]
    return reads
