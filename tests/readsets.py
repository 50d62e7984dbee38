"""Deterministic synthetic read sets for the parity tests.

np_reads() reproduces the recipe the survey used when it recorded the
reference's own statistics (numpy default_rng(seed),
uniform genome, uniform start positions, i.i.d. substitutions, 50 % reverse
complement), so the oracle can be checked against those known answers.
"""
import numpy as np

_LETTER2CODE = np.zeros(256, dtype=np.uint8)  # SPRING 2-bit code: A0 G1 C2 T3 (util.cpp:270-274)
_LETTER2CODE[ord("A")] = 0
_LETTER2CODE[ord("G")] = 1
_LETTER2CODE[ord("C")] = 2
_LETTER2CODE[ord("T")] = 3


def np_reads(seed, G, n, L, err):
    """-> uint8 [n, L] of ASCII letters."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, G, dtype=np.uint8)
    pos = rng.integers(0, G - L + 1, n)
    r = g[pos[:, None] + np.arange(L)[None, :]]
    e = rng.random((n, L)) < err
    r = np.where(e, (r + rng.integers(1, 4, (n, L), dtype=np.uint8)) % 4, r).astype(np.uint8)
    rc = rng.random(n) < 0.5
    r[rc] = (3 - r[rc])[:, ::-1]  # A0 C1 G2 T3 -> complement = 3-x
    return np.frombuffer(b"ACGT", dtype=np.uint8)[r]


def np_reads_repeat(seed, G, n, L, err):
    """Genome with 4 exact copies of one unit (repeat-rich, many multi-read bins)."""
    rng = np.random.default_rng(seed)
    unit = rng.integers(0, 4, G // 8, dtype=np.uint8)
    g = np.concatenate([unit if k % 2 == 0 else rng.integers(0, 4, G // 8, dtype=np.uint8) for k in range(8)])
    G = len(g)
    pos = rng.integers(0, G - L + 1, n)
    r = g[pos[:, None] + np.arange(L)[None, :]]
    e = rng.random((n, L)) < err
    r = np.where(e, (r + rng.integers(1, 4, (n, L), dtype=np.uint8)) % 4, r).astype(np.uint8)
    rc = rng.random(n) < 0.5
    r[rc] = (3 - r[rc])[:, ::-1]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[r]


def pack_fixed(letters):
    """uint8 [n, L] letters -> bytes of an input_clean_1.dna stream (util.cpp:269-294)."""
    n, L = letters.shape
    codes = _LETTER2CODE[letters]
    L4 = (L + 3) // 4
    pad = np.zeros((n, L4 * 4), dtype=np.uint8)
    pad[:, :L] = codes
    q = pad.reshape(n, L4, 4)
    by = (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).astype(np.uint8)
    rec = np.zeros((n, 2 + L4), dtype=np.uint8)
    rec[:, 0] = L & 0xFF
    rec[:, 1] = L >> 8
    rec[:, 2:] = by
    return rec.tobytes()


def pack_var(reads):
    """list of bytes/str reads (ACGT) of varying length -> .dna stream."""
    out = bytearray()
    for s in reads:
        if isinstance(s, str):
            s = s.encode()
        a = _LETTER2CODE[np.frombuffer(s, dtype=np.uint8)]
        L = len(a)
        L4 = (L + 3) // 4
        pad = np.zeros(L4 * 4, dtype=np.uint8)
        pad[:L] = a
        q = pad.reshape(L4, 4)
        by = (q[:, 0] | (q[:, 1] << 2) | (q[:, 2] << 4) | (q[:, 3] << 6)).astype(np.uint8)
        out += bytes([L & 0xFF, L >> 8]) + by.tobytes()
    return bytes(out)


def var_length_reads(seed, G, n, Lmin, Lmax, err):
    """Variable-length reads (exercises the three reverse sub-cases of updaterefcount)."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, G, dtype=np.uint8)
    comp = np.array([3, 2, 1, 0], dtype=np.uint8)
    letters = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = []
    for _ in range(n):
        L = int(rng.integers(Lmin, Lmax + 1))
        p = int(rng.integers(0, G - L + 1))
        r = g[p:p + L].copy()
        e = rng.random(L) < err
        r[e] = (r[e] + rng.integers(1, 4, int(e.sum()), dtype=np.uint8)) % 4
        if rng.random() < 0.5:
            r = comp[r][::-1]
        out.append(letters[r].tobytes())
    return out


def heavy_bin_reads(seed, n_heavy, n_other, L, err):
    """One genome position covered by n_heavy (>1000) reads + background: exercises the
    MAX_SEARCH_REORDER cap and the bin-removal encodings (SURVEY.md 8(c) item iv)."""
    rng = np.random.default_rng(seed)
    G = 20000
    g = rng.integers(0, 4, G, dtype=np.uint8)
    pos = np.concatenate([np.full(n_heavy, 5000), rng.integers(0, G - L + 1, n_other)])
    rng.shuffle(pos)
    n = len(pos)
    r = g[pos[:, None] + np.arange(L)[None, :]]
    e = rng.random((n, L)) < err
    r = np.where(e, (r + rng.integers(1, 4, (n, L), dtype=np.uint8)) % 4, r).astype(np.uint8)
    rc = rng.random(n) < 0.3
    r[rc] = (3 - r[rc])[:, ::-1]
    return np.frombuffer(b"ACGT", dtype=np.uint8)[r]
