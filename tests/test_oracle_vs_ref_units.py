"""Pins the oracle's main-loop primitives against the REAL reference functions that compile here without Boost
(oracle/_ref/libref_units.so: reorder.h:33-318, util.cpp:31-54 / :269-394, encoder.h:34-122 / :496-570,
encoder.cpp:32-109 / :177-222, taken by line range where they lie -- recipe in oracle/Makefile, driver
oracle/ref_units_driver.cpp).  Three kinds of checks:
  * one call of one function on random state (updaterefcount incl. the three reverse sub-cases and the in-place
    aliasing case, chartobitset / bitsettostring, readDnaFile, the pack / unpack helpers, read_fastq_block,
    buildcontig + writecontig, correct_order, readsingletons);
  * search_match on caller-driven state: real constructdictionary bins with random removals, random
    remainingreads[] (so "passes Hamming but is taken" happens), bins deeper than MAX_SEARCH_REORDER, both
    orientations, every shift -- serial flavour and the rounds schedule's whole-shift-loop search;
  * the SHADOW run: orc_reorder_serial advances, and at every step the reference's own function is run on a state
    that only reference code has touched; every search_match result, every consensus, every count column, every
    bin removal and every seed pick of a full run must agree.
Skipped when the prebuilt library is absent."""
import ctypes as C
import os
import tempfile

import numpy as np
import pytest

import readsets as rs
from helpers import KEYS, named_set, read_strings
from oracle import pyoracle as po

U = po.ref_units()
pytestmark = pytest.mark.skipif(U is None, reason="oracle/_ref/libref_units.so not built (needs /root/reference)")

LETTERS = np.frombuffer(b"ACGT", np.uint8)


def _rand_read(rng, n):
    return LETTERS[rng.integers(0, 4, n)].tobytes()


# ------------------------------------------------------------------ a3 / a13: encodings

@pytest.mark.parametrize("L", [20, 32, 33, 64, 100, 150, 251, 511])
def test_chartobitset_bitsettostring_revcomp_match_reference(L):
    Lb = po.lib()
    W = po.limbs(L)
    rng = np.random.default_rng(L)
    for _ in range(40):
        n = int(rng.integers(0, L + 1))
        s = _rand_read(rng, n)
        a, b = np.zeros(W, np.uint64), np.zeros(W, np.uint64)
        assert U.ref_u_chartobitset(s, n, L, a.ctypes.data) == 0
        Lb.orc_string_to_bits(s, n, L, b.ctypes.data)
        assert np.array_equal(a, b)
        sa, sb = C.create_string_buffer(L + 2), C.create_string_buffer(L + 2)
        assert U.ref_u_bitsettostring(a.ctypes.data, n, L, sa) == 0
        Lb.orc_bits_to_string(b.ctypes.data, n, L, sb)
        assert sa.raw[:n] == s and sb.raw[:n] == s
        U.ref_u_reverse_complement(s, sa, n)
        Lb.orc_reverse_complement(s, sb, n)
        assert sa.raw[:n + 1] == sb.raw[:n + 1]


def test_pack_helpers_match_reference(tmp_path):
    """write_dna_in_bits / read_dna_from_bits / write_dnaN_in_bits / read_dnaN_from_bits (util.cpp:269-374)."""
    rng = np.random.default_rng(2)
    reads = [_rand_read(rng, int(rng.integers(0, 512))) for _ in range(300)] + [b"", b"A", b"ACG", b"ACGT", b"ACGTA"]
    blob = b"".join(r + b"\0" for r in reads)
    p = str(tmp_path / "x.dna").encode()
    assert U.ref_u_write_dna(blob, len(reads), p, 0) == 0
    want = open(p, "rb").read()
    assert rs.pack_var(reads) == want  # the packer every parity test builds its inputs with
    Lb = po.lib()
    Lb.orc_pack_read.restype = C.c_size_t
    got = bytearray()
    for r in reads:
        buf = C.create_string_buffer(2 + 128)
        k = Lb.orc_pack_read(r, len(r), buf)
        got += buf.raw[:k]
    assert bytes(got) == want
    out = C.create_string_buffer(len(blob) + 16)
    assert U.ref_u_read_dna(p, len(reads), 0, out, len(blob) + 16) == len(blob)
    assert out.raw[:len(blob)] == blob
    # 4 bits per base with N (input_N.dna).  Reads of exactly MAX_READ_LEN = 511 bases are left out: the reference's
    # `uint8_t pos_in_bitarray` (util.cpp:330) wraps to 0 at 256 bytes and it writes the length with NO payload -- a
    # stream its own read_dnaN_from_bits cannot read back; the twin and the GPU front end write the 256 bytes.
    U.ref_u_write_dna(b"N" * 511 + b"\0", 1, p, 1)
    assert os.path.getsize(p) == 2 and len(po.pack_dnaN(["N" * 511])) == 258
    readsN = []
    for r in [r for r in reads if len(r) < 511]:
        a = bytearray(r)
        for i in range(len(a)):
            if rng.random() < 0.1:
                a[i] = ord("N")
        readsN.append(bytes(a))
    blobN = b"".join(r + b"\0" for r in readsN)
    assert U.ref_u_write_dna(blobN, len(readsN), p, 1) == 0
    wantN = open(p, "rb").read()
    assert po.pack_dnaN([r.decode() for r in readsN]) == wantN
    assert U.ref_u_read_dna(p, len(readsN), 1, out, len(blob) + 16) == len(blobN)
    assert out.raw[:len(blobN)] == blobN


@pytest.mark.parametrize("name", ["test_1+2", "syn2k_100", "syn5k_150", "var2k", "var_short", "syn1k_511", "syn2k_20"])
def test_readDnaFile_matches_reference(name, tmp_path):
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    W = po.limbs(L)
    # single file, and split in two files as a paired-end pool (reorder.h:233-242)
    rec_end = np.cumsum(2 + (ln.astype(np.int64) + 3) // 4)
    for n0 in (n, n // 3):
        f1, f2 = str(tmp_path / "input_clean_1.dna"), str(tmp_path / "input_clean_2.dna")
        cut = int(rec_end[n0 - 1]) if n0 else 0
        open(f1, "wb").write(dna[:cut])
        open(f2, "wb").write(dna[cut:])
        a = np.zeros((n, W), np.uint64)
        la = np.zeros(n, np.uint16)
        assert U.ref_u_readDnaFile(f1.encode(), f2.encode() if n0 < n else b"", n0, n - n0, L, a.ctypes.data,
                                   la.ctypes.data) == 0
        assert np.array_equal(a, read) and np.array_equal(la, ln)
        assert not os.path.exists(f1)  # the reference deletes its inputs (reorder.h:232, :241)
        if n0 < n:
            assert not os.path.exists(f2)


def test_fastq_front_end_twin_matches_reference_units(tmp_path):
    """orc_preprocess_fastq (the f1 oracle twin) against the REAL read_fastq_block (util.cpp:31-54) + the N test of
    preprocess.cpp:200-203 + the REAL write_dna_in_bits / write_dnaN_in_bits on the reads it returns."""
    rng = np.random.default_rng(4)
    recs = []
    for i in range(300):
        r = bytearray(_rand_read(rng, int(rng.integers(0, 160))))
        if i % 5 == 0:
            for _ in range(int(rng.integers(1, 4))):
                if len(r):
                    r[int(rng.integers(0, len(r)))] = ord("N")
        recs.append(b"@id%d\n" % i + bytes(r) + (b"\r\n" if i % 7 == 0 else b"\n") + b"+\n" + b"I" * len(r) + b"\n")
    p = str(tmp_path / "x.dna").encode()
    for text in (b"".join(recs), b"".join(recs)[:-1], b"", b"@a\nACGT\n+\nIIII"):
        out = C.create_string_buffer(len(text) + 1024)
        used = C.c_long(0)
        got = U.ref_u_read_fastq(text, len(text), 1000, out, len(text) + 1024, C.byref(used))
        assert got >= 0
        ref_reads = out.raw[:used.value].split(b"\0")[:-1] if used.value else []
        assert len(ref_reads) == got
        res = po.preprocess_fastq(text)
        assert res["num_reads"] == got
        clean = [r for r in ref_reads if b"N" not in r]
        withN = [(i, r) for i, r in enumerate(ref_reads) if b"N" in r]
        assert res["num_clean"] == len(clean) and res["num_N"] == len(withN)
        assert res["order_N"].tolist() == [i for i, _ in withN]
        U.ref_u_write_dna(b"".join(r + b"\0" for r in clean), len(clean), p, 0)
        assert open(p, "rb").read() == res["clean"]
        U.ref_u_write_dna(b"".join(r + b"\0" for _, r in withN), len(withN), p, 1)
        assert open(p, "rb").read() == res["ndna"]
    # a record cut after its second line: the reference throws, so does the twin
    bad = b"@a\nACGT\n"
    out = C.create_string_buffer(64)
    assert U.ref_u_read_fastq(bad, len(bad), 10, out, 64, C.byref(C.c_long())) == -1
    with pytest.raises(ValueError):
        po.preprocess_fastq(bad)


# ------------------------------------------------------------------ a11: updaterefcount

def _both_updates(L, cur, cnt, ref, revref, ref_len, reset, rev, shift, n):
    Lb = po.lib()
    W = po.limbs(L)
    ca, cb = cnt.copy(), cnt.copy()
    ra, rb = ref.copy(), ref.copy()
    rra, rrb = revref.copy(), revref.copy()
    la, lb = C.c_int(ref_len), C.c_int(ref_len)
    assert U.ref_u_updaterefcount(L, cur.ctypes.data, ca.ctypes.data, 512, ra.ctypes.data, rra.ctypes.data, C.byref(la),
                                  reset, rev, shift, n) == 0
    Lb.orc_updaterefcount(cur.ctypes.data, cb.ctypes.data, rb.ctypes.data, rrb.ctypes.data, C.byref(lb), reset, rev,
                          shift, n, L)
    assert la.value == lb.value, ("ref_len", L, reset, rev, shift, n, ref_len)
    assert np.array_equal(ra[:W], rb[:W]) and np.array_equal(rra[:W], rrb[:W]), ("consensus", L, reset, rev, shift, n, ref_len)
    assert np.array_equal(ca[:, :L], cb[:, :L]), ("counts", L, reset, rev, shift, n, ref_len)
    return ca, ra, rra, la.value


@pytest.mark.parametrize("L", [20, 64, 100, 150, 251, 511])
def test_updaterefcount_matches_reference_random_states(L):
    """Random count columns (ties, zeros, large counts), every branch of reorder.h:133-200: reset, forward, and the
    three reverse sub-cases -- (1) n - shift >= ref_len incl. d > 0 (the in-place aliasing copy), (2) ref_len + shift
    <= max_readlen, (3) the clipped case."""
    Lb = po.lib()
    W = po.limbs(L)
    rng = np.random.default_rng(1000 + L)
    seen = set()
    for it in range(400):
        n = int(rng.integers(1, L + 1)) if it % 3 else L
        ref_len = int(rng.integers(1, L + 1)) if it % 4 else L
        cur = np.zeros(16, np.uint64)
        Lb.orc_string_to_bits(_rand_read(rng, n), n, L, cur.ctypes.data)
        cnt = np.zeros((4, 512), np.int32)
        kind = it % 5
        if kind == 0:
            cnt[:, :L] = rng.integers(0, 3, (4, L))          # many ties and all-zero columns
        elif kind == 1:
            cnt[:, :L] = rng.integers(0, 400, (4, L))        # counts past one byte
        else:
            cnt[rng.integers(0, 4, L), np.arange(L)] = rng.integers(1, 60, L)
        ref, revref = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
        reset = int(it % 11 == 0)
        rev = int(rng.integers(0, 2))
        if reset:
            shift = 0
            seen.add(("reset", rev))
        elif rev:
            # every index the reference touches stays inside [0, L) as long as shift <= n (true in the run: a reverse
            # probe needs dict.start > shift and the read reaches past dict.end)
            if it % 3 == 0:
                ref_len = int(rng.integers(1, n + 1))  # makes sub-case (1) and its d > 0 aliasing copy likely
            shift = int(rng.integers(0, min(n, L // 2) + 1))
            if it % 7 == 3 and n - shift >= 1:
                ref_len = n - shift  # d = 0: what fixed-length data hits at shift 0
            branch = 1 if n - shift >= ref_len else 2 if ref_len + shift <= L else 3
            seen.add(("rev", branch, n - shift - ref_len > 0))
        else:
            shift = int(rng.integers(0, min(ref_len, L // 2) + 1))
            seen.add(("fwd", ref_len - shift < n))
        _both_updates(L, cur, cnt, ref, revref, ref_len, reset, rev, shift, n)
    assert {("rev", 1, True), ("rev", 1, False), ("rev", 2, False), ("rev", 3, False), ("fwd", True), ("fwd", False),
            ("reset", 0), ("reset", 1)} <= seen, seen


def test_updaterefcount_chain_of_updates_matches_reference():
    """A consensus carried through 300 updates on both sides (state fed forward, not re-randomised)."""
    Lb = po.lib()
    L = 150
    rng = np.random.default_rng(77)
    g = LETTERS[rng.integers(0, 4, 4000)]
    pos = 100
    n = L
    cur = np.zeros(16, np.uint64)
    Lb.orc_string_to_bits(g[pos:pos + n].tobytes(), n, L, cur.ctypes.data)
    cnt = np.zeros((4, 512), np.int32)
    ref, revref = np.zeros(16, np.uint64), np.zeros(16, np.uint64)
    cnt, ref, revref, ref_len = _both_updates(L, cur, cnt, ref, revref, 0, 1, 0, 0, n)
    for it in range(300):
        shift = int(rng.integers(0, 8))
        n = int(rng.integers(60, L + 1))
        pos += shift
        r = g[pos:pos + n].copy()
        e = rng.random(n) < 0.02
        r[e] = LETTERS[rng.integers(0, 4, int(e.sum()))]
        Lb.orc_string_to_bits(r.tobytes(), n, L, cur.ctypes.data)
        cnt, ref, revref, ref_len = _both_updates(L, cur, cnt, ref, revref, ref_len, 0, 0, shift, n)


# ------------------------------------------------------------------ a9: search_match on caller-driven state

def _unit(read, ln, L):
    Lb = po.lib()
    Lb.orc_unit_create.restype = C.c_void_p
    Lb.orc_unit_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    Lb.orc_unit_free.argtypes = [C.c_void_p]
    Lb.orc_unit_remove.argtypes = [C.c_void_p, C.c_uint32]
    Lb.orc_unit_set_remaining.argtypes = [C.c_void_p, C.c_void_p]
    Lb.orc_unit_get_remaining.argtypes = [C.c_void_p, C.c_void_p]
    Lb.orc_unit_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    Lb.orc_unit_rounds_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint32),
                                          C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return Lb.orc_unit_create(read.ctypes.data, ln.ctypes.data, len(ln), L)


def _shl2(a, k):  # bitset << 2k on W limbs
    W = len(a)
    v = int.from_bytes(a.tobytes(), "little")
    v = (v << (2 * k)) & ((1 << (64 * W)) - 1)
    return np.frombuffer(v.to_bytes(8 * W, "little"), np.uint64).copy()


def _shr2(a, k):
    W = len(a)
    v = int.from_bytes(a.tobytes(), "little") >> (2 * k)
    return np.frombuffer(v.to_bytes(8 * W, "little"), np.uint64).copy()


@pytest.mark.parametrize("name", ["syn2k_100", "syn5k_150", "var2k", "var_short", "heavy", "dups", "syn3k_64", "syn2k_251"])
def test_search_match_matches_reference_on_driven_state(name):
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    read = np.ascontiguousarray(read)
    ln = np.ascontiguousarray(ln)
    W = po.limbs(L)
    Lb = po.lib()
    rng = np.random.default_rng(len(name) * 31 + L)
    strs = read_strings(read, ln)
    with tempfile.TemporaryDirectory() as td:
        sh = U.ref_shadow_create(read.ctypes.data, ln.ctypes.data, n, L, td.encode(), 2)
    un = _unit(read, ln, L)
    try:
        # state: a random third of the reads removed from their bins (as the reference does once they are current), and
        # an INDEPENDENT random remainingreads[] so that bins hold taken reads and dead reads are marked remaining
        gone = rng.permutation(n)[: n // 3]
        for r in gone:
            assert U.ref_shadow_remove(sh, int(r)) == 0
            Lb.orc_unit_remove(un, int(r))
        rem = (rng.random(n) < 0.6).astype(np.uint8)
        U.ref_shadow_set_remaining(sh, rem.ctypes.data)
        Lb.orc_unit_set_remaining(un, rem.ctypes.data)
        hits = 0
        calls = 0
        for q in range(1500):
            src = int(rng.integers(0, n))
            s = bytearray(strs[src].encode())
            for _ in range(int(rng.integers(0, 4))):  # 0-3 substitutions: around the threshold of 4 bit differences
                if len(s):
                    s[int(rng.integers(0, len(s)))] = int(LETTERS[rng.integers(0, 4)])
            ref_len = len(s)
            full = np.zeros(W, np.uint64)
            Lb.orc_string_to_bits(bytes(s), ref_len, L, full.ctypes.data)
            rev = int(rng.integers(0, 2))
            shift = int(rng.integers(0, max(1, L // 2))) if q % 3 else 0
            work = _shl2(full, shift) if rev else _shr2(full, shift)
            ka, kb = C.c_uint32(0), C.c_uint32(0)
            fa = U.ref_shadow_search_raw(sh, work.ctypes.data, rev, shift, ref_len, C.byref(ka))
            fb = Lb.orc_unit_search(un, work.ctypes.data, rev, shift, ref_len, C.byref(kb))
            calls += 1
            assert fa == fb, (name, q, rev, shift, ref_len)
            if fa:
                hits += 1
                assert ka.value == kb.value, (name, q, rev, shift)
        ra, rb = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        U.ref_shadow_get_remaining(sh, ra.ctypes.data)
        Lb.orc_unit_get_remaining(un, rb.ctypes.data)
        assert np.array_equal(ra, rb)
        assert hits > 20, (name, hits, calls)
    finally:
        U.ref_shadow_destroy(sh)
        Lb.orc_unit_free(un)


@pytest.mark.parametrize("name", ["syn2k_100", "syn5k_150", "var2k", "heavy", "dups"])
def test_rounds_search_equals_reference_shift_loop(name):
    """The rounds schedule's search primitive (immutable bins + taken[]) against the reference's search_match driven
    through the shift loop of reorder.h:479-558 on the equivalent state (every taken read removed from its bins)."""
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    read = np.ascontiguousarray(read)
    ln = np.ascontiguousarray(ln)
    W = po.limbs(L)
    Lb = po.lib()
    rng = np.random.default_rng(len(name) * 17 + L)
    strs = read_strings(read, ln)
    with tempfile.TemporaryDirectory() as td:
        sh = U.ref_shadow_create(read.ctypes.data, ln.ctypes.data, n, L, td.encode(), 1)
    un = _unit(read, ln, L)
    try:
        taken = (rng.random(n) < 0.5).astype(np.uint8)
        for r in np.nonzero(taken)[0]:
            assert U.ref_shadow_remove(sh, int(r)) == 0
        rem = (1 - taken).astype(np.uint8)
        U.ref_shadow_set_remaining(sh, rem.ctypes.data)
        found = 0
        for q in range(300):
            src = int(rng.integers(0, n))
            s = bytearray(strs[src].encode())
            for _ in range(int(rng.integers(0, 3))):
                if len(s):
                    s[int(rng.integers(0, len(s)))] = int(LETTERS[rng.integers(0, 4)])
            if q % 4 == 0:  # consensus ahead of the read: the match sits at a shift
                cut = int(rng.integers(0, min(20, max(1, len(s) // 3))))
                s = bytearray(_rand_read(rng, cut)) + s[: len(s) - cut]
            ref_len = len(s)
            ref, revref = np.zeros(W, np.uint64), np.zeros(W, np.uint64)
            Lb.orc_string_to_bits(bytes(s), ref_len, L, ref.ctypes.data)
            rcs = C.create_string_buffer(ref_len + 1)
            Lb.orc_reverse_complement(bytes(s), rcs, ref_len)
            Lb.orc_string_to_bits(rcs.raw[:ref_len], ref_len, L, revref.ctypes.data)
            ka, kb = C.c_uint32(0), C.c_uint32(0)
            sa, sb, va, vb = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
            fa = U.ref_shadow_search_loop(sh, ref.ctypes.data, revref.ctypes.data, ref_len, C.byref(ka), C.byref(sa),
                                          C.byref(va))
            fb = Lb.orc_unit_rounds_search(un, ref.ctypes.data, revref.ctypes.data, ref_len, taken.ctypes.data,
                                           C.byref(kb), C.byref(sb), C.byref(vb))
            assert fa == fb, (name, q)
            if fa:
                found += 1
                assert (ka.value, sa.value, va.value) == (kb.value, sb.value, vb.value), (name, q)
        assert found > 30
    finally:
        U.ref_shadow_destroy(sh)
        Lb.orc_unit_free(un)


# ------------------------------------------------------------------ the shadow run

SHADOW_SETS = ["test_1", "test_1+2", "syn2k_100", "syn5k_150", "syn3k_64", "syn2k_251", "syn1k_511", "syn2k_20", "var_long",
               "var2k", "var_short", "heavy", "repeat10k", "dups", "one", "two_same", "empty"]


@pytest.mark.parametrize("name", SHADOW_SETS)
def test_serial_run_shadowed_by_reference_functions(name):
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_serial(read, ln, L)
    with tempfile.TemporaryDirectory() as td:
        got, st, mm = po.reorder_serial_shadow(read, ln, L, td)
    assert mm[:6].tolist() == [0, 0, 0, 0, 0, 0], (name, mm.tolist())
    # the hooks really ran: one search hook per search_match call, one update hook per updaterefcount call
    assert mm[6] == st["search_calls"] and mm[7] == st["updates"]
    assert mm[9] == st["unmatched"] - (1 if n else 0) + (1 if n else 0) or n == 0
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), (name, k)
    assert got["stats"] == want["stats"]


@pytest.mark.slow
def test_serial_run_shadowed_100k_reads():
    """100 000 x 100 bp at 30x: 2.9 M search_match calls, 0.11 M updaterefcount calls, every one cross-checked."""
    n, L = 100_000, 100
    a = rs.np_reads(21, n * L // 30, n, L, 0.01)
    read, ln = po.load_dna(rs.pack_fixed(a), n, L)
    with tempfile.TemporaryDirectory() as td:
        got, st, mm = po.reorder_serial_shadow(read, ln, L, td)
    assert mm[:6].tolist() == [0] * 6, mm.tolist()
    assert mm[6] == st["search_calls"] > 2_000_000 and mm[7] == st["updates"] > 100_000


# ------------------------------------------------------------------ f2 / f3 units: encoder.cpp, encoder.h

def _contig_both(reads, pos, rc, order, abs0, td):
    Lb = po.lib()
    Lb.orc_enc_contig.restype = C.c_long
    Lb.orc_enc_contig.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64),
                                  C.c_void_p, C.c_long, C.c_void_p]
    blob = b"".join(r + b"\0" for r in reads)
    pos = np.ascontiguousarray(pos, np.int64)
    order = np.ascontiguousarray(order, np.uint32)
    cap = sum(len(r) for r in reads) * 8 + 64 * len(reads) + 4096
    res = []
    for which in (0, 1):
        out = np.zeros(cap, np.uint8)
        sizes = np.zeros(7, np.uint64)
        ap = C.c_uint64(abs0)
        if which == 0:
            w = U.ref_u_contig(blob, pos.ctypes.data, bytes(rc), order.ctypes.data, len(reads), td.encode(), C.byref(ap),
                               out.ctypes.data, cap, sizes.ctypes.data)
        else:
            w = Lb.orc_enc_contig(blob, pos.ctypes.data, bytes(rc), order.ctypes.data, len(reads), C.byref(ap),
                                  out.ctypes.data, cap, sizes.ctypes.data)
        assert w >= 0
        res.append((out[:w].tobytes(), sizes.tolist(), ap.value))
    return res


def test_buildcontig_writecontig_match_reference(tmp_path):
    """encoder.cpp:32-109 on random contigs: reads sorted by position, 0-4 % substitutions, ties in the majority vote,
    uncovered stretches never occur (a contig is a chain of overlapping reads), single-read contigs (list_size == 1)."""
    rng = np.random.default_rng(9)
    td = str(tmp_path)
    for trial in range(60):
        count = 1 if trial % 10 == 0 else int(rng.integers(2, 40))
        G = LETTERS[rng.integers(0, 4, 4000)]
        pos = [0]
        lens = [int(rng.integers(30, 152))]
        for _ in range(count - 1):
            # next read starts inside the stretch covered so far (pos sorted ascending like list::sort leaves them)
            pos.append(pos[-1] + int(rng.integers(0, min(lens[-1], 40))))
            lens.append(int(rng.integers(30, 152)))
        reads = []
        for p, n in zip(pos, lens):
            r = G[p:p + n].copy()
            e = rng.random(n) < (0.04 if trial % 3 else 0.3)  # 30 %: plenty of ties / outvoted bases
            r[e] = LETTERS[rng.integers(0, 4, int(e.sum()))]
            reads.append(r.tobytes())
        rc = bytes(rng.choice(np.frombuffer(b"dr", np.uint8), count).tolist())
        order = rng.integers(0, 2**32 - 1, count, dtype=np.uint64).astype(np.uint32)
        a, b = _contig_both(reads, pos, rc, order, int(rng.integers(0, 2**40)), td)
        assert a[1] == b[1], (trial, a[1], b[1])
        assert a[0] == b[0], trial
        assert a[2] == b[2]


def test_correct_order_matches_reference(tmp_path):
    """encoder.cpp:177-222: the twin works on arrays; the reference rewrites read_order.bin.<tid> in place."""
    rng = np.random.default_rng(12)
    for trial in range(12):
        n_clean = int(rng.integers(1, 3000))
        nN = int(rng.integers(0, 400)) if trial else 0
        total = n_clean + nN
        posN = np.sort(rng.choice(total, nN, replace=False)).astype(np.uint32)
        perm = rng.permutation(n_clean).astype(np.uint32)
        ns = int(rng.integers(0, n_clean + 1))
        sing, matched = perm[:ns], perm[ns:]
        T = int(rng.integers(1, 5))
        cuts = np.sort(rng.integers(0, len(matched) + 1, T - 1))
        parts = np.split(matched, cuts)
        for t in range(T):
            parts[t].tofile(str(tmp_path / ("read_order.bin.%d" % t)))
        posN.tofile(str(tmp_path / "read_order_N.bin"))
        order_s = np.concatenate([sing, posN]).astype(np.uint32)
        if len(order_s) == 0:
            order_s = np.zeros(1, np.uint32)
        assert U.ref_u_correct_order(order_s.ctypes.data, len(matched), ns, nN, T, str(tmp_path).encode()) == 0
        want_s = po.correct_order(sing, posN, n_clean)
        assert np.array_equal(order_s[:ns], want_s)
        for t in range(T):
            got = np.fromfile(str(tmp_path / ("read_order.bin.%d" % t)), np.uint32)
            assert np.array_equal(got, po.correct_order(parts[t], posN, n_clean)), (trial, t)
        assert not os.path.exists(str(tmp_path / "read_order_N.bin"))


@pytest.mark.parametrize("L", [40, 100, 150, 302, 511])
def test_encoder_bits3_and_readsingletons_match_reference(L, tmp_path):
    """stringtobitset / bitsettostring with the encoder's 3-bit basemask (encoder.h:496-517, :105-122) and
    readsingletons (encoder.h:541-570) fed by the REAL write_dna_in_bits / write_dnaN_in_bits files."""
    Lb = po.lib()
    Lb.orc_enc_bits3.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    W3 = (3 * L - 1) // 64 + 1
    rng = np.random.default_rng(L)
    sing, withN = [], []
    for i in range(60):
        n = int(rng.integers(1, min(L, 510) + 1))
        s = bytearray(_rand_read(rng, n))
        sing.append(bytes(s))
        for _ in range(int(rng.integers(1, 4))):
            s[int(rng.integers(0, n))] = ord("N")
        withN.append(bytes(s))
    for s in sing + withN:
        a = np.zeros(24, np.uint64)
        back = C.create_string_buffer(len(s) + 1)
        assert U.ref_u_enc_bits3_roundtrip(s, len(s), L, a.ctypes.data, back) == 0
        b = np.zeros(24, np.uint64)
        Lb.orc_enc_bits3(s, len(s), b.ctypes.data, W3)
        assert np.array_equal(a, b)
        assert back.raw[:len(s)] == s
    d = str(tmp_path)
    U.ref_u_write_dna(b"".join(r + b"\0" for r in sing), len(sing), (d + "/temp.dna.singleton").encode(), 0)
    U.ref_u_write_dna(b"".join(r + b"\0" for r in withN), len(withN), (d + "/input_N.dna").encode(), 1)
    os_s = rng.integers(0, 10**6, len(sing)).astype(np.uint32)
    os_N = rng.integers(0, 10**6, len(withN)).astype(np.uint32)
    os_s.tofile(d + "/read_order.bin.singleton")
    os_N.tofile(d + "/read_order_N.bin")
    m = len(sing) + len(withN)
    limbs = np.zeros((m, 24), np.uint64)
    order = np.zeros(m, np.uint32)
    lens = np.zeros(m, np.uint16)
    assert U.ref_u_readsingletons(d.encode(), len(sing), len(withN), L, limbs.ctypes.data, order.ctypes.data,
                                  lens.ctypes.data) == 0
    assert np.array_equal(order, np.concatenate([os_s, os_N]))
    for i, s in enumerate(sing + withN):
        b = np.zeros(24, np.uint64)
        Lb.orc_enc_bits3(s, len(s), b.ctypes.data, W3)
        assert np.array_equal(limbs[i], b) and lens[i] == len(s)
