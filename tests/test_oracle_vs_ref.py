"""Pins the oracle against the REAL reference code that is buildable here
(oracle/_ref/libref_bitset.so = /root/reference/src/bitset_util.{h,cpp} + BooPHF.h compiled
where they lie, recipe oracle/Makefile).  Covers constructdictionary (keys, bins, in-bin order,
dict_numreads), generateindexmasks (through the keys), bbhashdict::findpos/remove and generatemasks.
Skipped when the prebuilt _ref library is absent."""
import ctypes as C
import tempfile

import numpy as np
import pytest

from helpers import named_set
from oracle import pyoracle as po

ref = po.ref_lib()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref not built (needs /root/reference)")


@pytest.mark.parametrize("name", ["test_1+2", "syn2k_100", "syn5k_150", "syn3k_64", "syn2k_251", "syn1k_511", "syn2k_20",
                                  "var_long", "var2k", "var_short", "heavy", "dups"])
@pytest.mark.parametrize("num_thr", [1, 3])
def test_constructdictionary_matches_reference(name, num_thr):
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    W = po.limbs(L)
    s, e = po.dict_windows(L)
    for which in (0, 1):
        keys, sp, ids = po.build_dict(read, ln, L, which)
        probe = np.concatenate([keys, keys ^ np.uint64(0x5555)]) if len(keys) else keys
        bin_size = np.zeros(max(len(probe), 1), np.uint32)
        bin_ids = np.zeros(max(2 * len(ids), 1), np.uint32)
        nk, dn = C.c_uint32(), C.c_uint32()
        with tempfile.TemporaryDirectory() as td:
            rc = ref.ref_build_dict(np.ascontiguousarray(read).ctypes.data, ln.ctypes.data, n, W, s[0], e[0],
                                    s[1], e[1], td.encode(), num_thr, which, probe.ctypes.data, len(probe),
                                    bin_size.ctypes.data, bin_ids.ctypes.data, C.byref(nk), C.byref(dn))
        assert rc == 0
        assert nk.value == len(keys) and dn.value == len(ids)
        # every oracle key is a reference key with the identical bin (same ids, same order)
        o = 0
        for i in range(len(keys)):
            sz = int(bin_size[i])
            assert sz == sp[i + 1] - sp[i], (name, which, i)
            assert np.array_equal(bin_ids[o:o + sz], ids[sp[i]:sp[i + 1]])
            o += sz
        assert o == len(ids)


def test_bin_remove_encoding_matches_reference():
    rng = np.random.default_rng(3)
    for cap in [1, 2, 3, 4, 5, 17, 200, 1500]:
        for trial in range(4):
            ids = np.sort(rng.choice(10_000_000, cap, replace=False)).astype(np.uint32)
            a, b = ids.copy(), ids.copy()
            ea, eb = np.zeros(1, np.uint8), np.zeros(1, np.uint8)
            for cur in rng.permutation(ids):
                la = po.lib().orc_bin_remove(a.ctypes.data, cap, ea.ctypes.data, int(cur))
                lb = ref.ref_bin_remove(b.ctypes.data, cap, eb.ctypes.data, int(cur))
                assert la == lb and ea[0] == eb[0]
                assert np.array_equal(a, b)  # identical memory image incl. tail sentinels
                assert po.lib().orc_bin_live(a.ctypes.data, cap) == ref.ref_bin_live(b.ctypes.data, cap)
            assert ea[0] == 1 and la == 1  # last entry is kept, bin flagged empty


@pytest.mark.parametrize("L", [37, 100, 150])
def test_hamming_masks_match_reference(L):
    W = po.limbs(L)
    rng = np.random.default_rng(L)
    for _ in range(300):
        a = rng.integers(0, 2**63, W, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, W, dtype=np.uint64)
        b = rng.integers(0, 2**63, W, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, W, dtype=np.uint64)
        i = int(rng.integers(0, L))
        j = int(rng.integers(0, L))
        want = ref.ref_mask_hamming(a.ctypes.data, b.ctypes.data, W, L, i, j)  # mask[i][j] = bases [i, L-j)
        got = po.lib().orc_hamming_range(a.ctypes.data, b.ctypes.data, W, i, L - j)
        assert got == want


# ------------------------------------------------------------------ encoder stage: 3 bits per base

def _bits3(strings, L):
    W3 = (3 * L - 1) // 64 + 1
    Lb = po.lib()
    Lb.orc_enc_bits3.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    out = np.zeros((max(len(strings), 1), W3), np.uint64)
    for i, s in enumerate(strings):
        Lb.orc_enc_bits3(s.encode(), len(s), out[i].ctypes.data, W3)
    return out[:len(strings)], W3


@pytest.mark.parametrize("name", ["syn2k_100", "syn5k_150", "syn3k_64", "var2k", "var_short", "dups"])
def test_encoder_dictionary_bpb3_matches_reference(name):
    """constructdictionary<N>(..., bpb = 3, ...) as encoder_main calls it (encoder.h:617-619), reads with N included."""
    from helpers import make_N_reads, read_strings
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    strs = read_strings(read, ln)
    pool = strs[: n // 2] + make_N_reads(strs, 300, 11)
    lens = np.array([len(s) for s in pool], np.uint16)
    b3, W3 = _bits3(pool, L)
    assert W3 <= 16
    Lb = po.lib()
    s, e = (C.c_int * 2)(), (C.c_int * 2)()
    Lb.orc_enc_dict_windows(L, s, e)
    Lb.orc_enc_build_dict.restype = C.c_uint32
    Lb.orc_enc_build_dict.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(C.c_uint32)]
    ref.ref_build_dict_bpb.restype = C.c_int
    ref.ref_build_dict_bpb.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_char_p, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int]
    m = len(pool)
    for which in (0, 1):
        keys = np.zeros(m, np.uint64)
        sp = np.zeros(m + 1, np.uint32)
        ids = np.zeros(m, np.uint32)
        dn = C.c_uint32()
        nk = Lb.orc_enc_build_dict(b3.ctypes.data, lens.ctypes.data, m, L, which, keys.ctypes.data, sp.ctypes.data,
                                   ids.ctypes.data, C.byref(dn))
        keys, sp, ids = keys[:nk], sp[:nk + 1], ids[:dn.value]
        bin_size = np.zeros(max(nk, 1), np.uint32)
        bin_ids = np.zeros(max(2 * len(ids), 1), np.uint32)
        rnk, rdn = C.c_uint32(), C.c_uint32()
        with tempfile.TemporaryDirectory() as td:
            rc = ref.ref_build_dict_bpb(np.ascontiguousarray(b3).ctypes.data, lens.ctypes.data, m, W3, s[0], e[0], s[1],
                                        e[1], td.encode(), 2, which, keys.ctypes.data, nk, bin_size.ctypes.data,
                                        bin_ids.ctypes.data, C.byref(rnk), C.byref(rdn), 3)
        assert rc == 0
        assert rnk.value == nk and rdn.value == dn.value
        o = 0
        for i in range(nk):
            sz = int(bin_size[i])
            assert sz == sp[i + 1] - sp[i]
            assert np.array_equal(bin_ids[o:o + sz], ids[sp[i]:sp[i + 1]])
            o += sz
        assert o == len(ids)


@pytest.mark.parametrize("L", [40, 100, 151])
def test_encoder_hamming_mask_bpb3_matches_reference(L):
    """((a ^ b) & mask[0][L - len]).count() with generatemasks(mask, L, 3) (encoder.h:139-140, :290-297)."""
    W3 = (3 * L - 1) // 64 + 1
    rng = np.random.default_rng(L)
    Lb = po.lib()
    Lb.orc_enc_hamming3.restype = C.c_int
    Lb.orc_enc_hamming3.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    ref.ref_mask_hamming_bpb.restype = C.c_int
    ref.ref_mask_hamming_bpb.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    for _ in range(300):
        a = rng.integers(0, 2**63, W3, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, W3, dtype=np.uint64)
        b = rng.integers(0, 2**63, W3, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, W3, dtype=np.uint64)
        ln = int(rng.integers(1, L + 1))
        want = ref.ref_mask_hamming_bpb(a.ctypes.data, b.ctypes.data, W3, L, 0, L - ln, 3)
        got = Lb.orc_enc_hamming3(a.ctypes.data, b.ctypes.data, W3, ln)
        assert got == want
