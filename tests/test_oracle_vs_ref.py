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
