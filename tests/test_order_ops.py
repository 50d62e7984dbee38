"""order inversion / N-read correction on the GPU vs the literal restatements of
generate_order_se, generate_order_pe (reorder_compress_quality_id.cpp:101-125) and correct_order
(encoder.cpp:177-222).  Bit-exact (index work)."""
import numpy as np
import pytest

from oracle import pyoracle as po


def _data(n, nN, seed):
    rng = np.random.default_rng(seed)
    order = rng.permutation(n).astype(np.uint32)
    total = n + nN
    order_N = np.sort(rng.choice(total, nN, replace=False)).astype(np.uint32) if nN else np.zeros(0, np.uint32)
    return order, order_N


def test_oracle_order_ops_known_answers():
    # hand-checked: 5 clean reads, N reads originally at positions 1 and 4 of 7
    order = np.array([3, 0, 4, 1, 2], np.uint32)
    assert po.generate_order_se(order).tolist() == [1, 3, 4, 0, 2]
    assert po.generate_order_pe(np.array([3, 0, 2, 1], np.uint32)).tolist() == [0, 1]
    # clean index -> original index: 0->0, 1->2, 2->3, 3->5, 4->6
    assert po.correct_order(order, np.array([1, 4], np.uint32), 5).tolist() == [5, 0, 6, 2, 3]
    # pe_encode, 3 pairs (file-1 reads 0..2, mates 3..5): reordered file = [4, 1, 5, 0, 3, 2]
    # file-1 ranks: read 1 -> 0, read 0 -> 1, read 2 -> 2; mates: 4 -> rank(1)+3 = 3, 5 -> rank(2)+3 = 5, 3 -> 4
    assert po.pe_encode(np.array([4, 1, 5, 0, 3, 2], np.uint32)).tolist() == [3, 0, 5, 1, 4, 2]


needs_ref = pytest.mark.skipif(po.ref_order_bin() is None, reason="oracle/_ref/ref_order not built (needs /root/reference)")


@needs_ref
@pytest.mark.parametrize("n", [2, 6, 1000, 100_002, 2_000_000])
def test_oracle_twins_equal_the_real_reference(n):
    """The oracle's literal twins against the reference's own pe_encode.cpp and generate_order_se/pe, compiled
    in place (oracle/_ref/ref_order): this pins row f3's oracle."""
    order, _ = _data(n, 0, 7 * n + 1)
    assert np.array_equal(po.ref_order("se", order), po.generate_order_se(order))
    assert np.array_equal(po.ref_order("pe", order), po.generate_order_pe(order))
    assert np.array_equal(po.ref_order("pe_encode", order), po.pe_encode(order))


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 1000, 3_000_000])
def test_gpu_order_ops_equal_the_real_reference(n):
    from spring_amd import order_ops as oo
    order, _ = _data(n, 0, 3 * n + 5)
    assert np.array_equal(oo.generate_order_se(order)[0], po.ref_order("se", order))
    assert np.array_equal(oo.generate_order_pe(order)[0], po.ref_order("pe", order))
    assert np.array_equal(oo.pe_encode(order)[0], po.ref_order("pe_encode", order))


@pytest.mark.gpu
@pytest.mark.parametrize("n,nN", [(1, 0), (2, 1), (1000, 37), (100_001, 5000), (100_002, 1), (3_000_000, 250_000)])
def test_order_ops_bit_exact(n, nN):
    from spring_amd import order_ops as oo
    order, order_N = _data(n, nN, n + nN)
    got, _ = oo.generate_order_se(order)
    assert np.array_equal(got, po.generate_order_se(order))
    got, _ = oo.generate_order_pe(order)
    assert np.array_equal(got, po.generate_order_pe(order))
    got, _ = oo.correct_order(order, order_N, n)
    assert np.array_equal(got, po.correct_order(order, order_N, n))
    if n % 2 == 0:  # paired-end pools hold 2 x pairs reads
        got, _ = oo.pe_encode(order)
        assert np.array_equal(got, po.pe_encode(order))
    else:
        with pytest.raises(Exception):
            oo.pe_encode(order)


@pytest.mark.gpu
def test_order_ops_full_size_properties():
    """100 M entries: inversion is an involution-like property (order_array[order[i]] == i) and the
    corrected order is strictly the clean->original monotone map."""
    from spring_amd import order_ops as oo
    n, nN = 100_000_000, 3_000_000
    rng = np.random.default_rng(1)
    order = rng.permutation(n).astype(np.uint32)
    inv, ms = oo.generate_order_se(order)
    assert np.array_equal(inv[order], np.arange(n, dtype=np.uint32))
    order_N = np.sort(rng.choice(n + nN, nN, replace=False)).astype(np.uint32)
    ident = np.arange(n, dtype=np.uint32)
    corr, ms2 = oo.correct_order(ident, order_N, n)
    assert np.all(np.diff(corr.astype(np.int64)) >= 1) and not np.isin(corr[:: 997], order_N).any()
    assert int(corr[-1]) <= n + nN - 1
    # pe_encode: a permutation in which mates end up exactly n/2 apart
    new, ms3 = oo.pe_encode(order)
    inv_new = np.empty(n, np.uint32)
    inv_new[new] = np.arange(n, dtype=np.uint32)          # decompressed position -> reordered position
    assert np.array_equal(np.sort(new), np.arange(n, dtype=np.uint32))
    half = n // 2
    first = order[inv_new[:half]]                          # original ids at decompressed positions 0 .. n/2-1
    assert np.all(first < half) and np.array_equal(order[inv_new[half:2 * half]], first + half)
