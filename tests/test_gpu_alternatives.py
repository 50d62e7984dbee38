"""The alternatives schedule on the GPU (`opts.alternatives = 2`): a match proposal carries the next passing read of the
winning bin as a second candidate -- what the reference's thread tries after losing the read_lock race
(reorder.h:303-311) -- and a second resolution pass hands it to a chain that lost its first one.  Executable
specification: oracle/reorder_oracle.c::orc_reorder_rounds_alt (K = 1 is the serial `-t 1` order for every A; checked on
CPU in tests/test_oracle.py).  Here: the HIP path through the C ABI == that oracle with A = 2, bit for bit -- every way a
winner is found (ordered batches with the balanced scan, the tail's serial walk, single-read bins, k_long's three
kernels), the counting build, two virtual ranks (the second candidate travels inside the 64-bit proposal word), the
library's own choice."""
import numpy as np
import pytest

from helpers import KEYS, named_set
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _sa():
    import spring_amd
    return spring_amd


def _same(a, b, what):
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), (what, k, len(a[k]), len(b[k]))
    assert np.array_equal(a["tid_off"], b["tid_off"]) and np.array_equal(a["tid_off_s"], b["tid_off_s"]), what


@pytest.mark.parametrize("name", ["syn2k_100", "syn5k_150", "syn3k_64", "syn2k_251", "syn1k_511", "var_long", "var2k",
                                  "var_short", "heavy", "repeat10k", "dups", "tandem", "test_1+2"])
@pytest.mark.parametrize("K,T", [(1, 1), (2, 1), (16, 2), (64, 8), (300, 3)])
def test_two_candidates_vs_oracle(name, K, T):
    sa = _sa()
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_serial(read, ln, L) if K == 1 else po.reorder_rounds(read, ln, L, K, T, alternatives=2)
    for stats in (False, True):  # production build (balanced scan, resumed searches) and counting build (serial walks)
        got = sa.reorder_dna(dna, n, L, sa.ReorderOpts(num_chains=K, num_thr=T, alternatives=2, collect_stats=stats))
        assert got["stats"]["alternatives"] == 2
        if K == 1:
            for k in KEYS:
                assert np.array_equal(got[k], want[k]), (name, k)
            continue
        _same(got, want, (name, K, T, stats))
        if stats:
            for k in ("unmatched", "probes", "keyok", "cands", "hits", "iterations", "lost"):
                assert got["stats"][k] == want["stats"][k], (name, K, k, got["stats"][k], want["stats"][k])


@pytest.mark.parametrize("budget", [-1, 0, 1, 3])
@pytest.mark.parametrize("n,L,G,K", [(60_000, 100, 300, 256), (40_000, 150, 2_000, 500), (30_000, 150, 400, 37),
                                     (50_000, 120, 5_000, 1024)])
def test_contended_pools_two_candidates(n, L, G, K, budget):
    """Hundreds of chains on a genome of a few hundred bases: most first candidates are lost, bins of hundreds to
    thousands of reads (some beyond MAX_SEARCH_REORDER: the second candidate obeys the same live-entry window).
    budget: searches finished by k_long_list / k_long_scan / k_long_fin (1: nearly every search over a multi-read bin)."""
    sa = _sa()
    outs = {}
    for A in (2, 1):
        with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=2, deep_bins=1, long_budget=budget, long_split=1 if budget == 1 else 0,
                                            alternatives=A)) as st:
            st.load_synth(n, L, G, 23, 10000)
            outs[A] = st.run().streams()
            dna = st.download_dna()
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_rounds(read, ln, L, K, 2, alternatives=2)
    _same(outs[2], want, ("contended", budget))
    assert outs[2]["stats"]["lost"] == want["stats"]["lost"]
    _same(outs[1], po.reorder_rounds(read, ln, L, K, 2), ("contended, one candidate", budget))
    # what the schedule is for: fewer lost proposals, not more rounds
    assert outs[2]["stats"]["lost"] < outs[1]["stats"]["lost"], (outs[2]["stats"]["lost"], outs[1]["stats"]["lost"])
    if budget > 0:
        assert outs[2]["stats"]["long_searches"] > 0


@pytest.mark.parametrize("ranks", [2, 3])
def test_two_candidates_over_virtual_ranks(ranks):
    """The multi-GPU data path: the second candidate travels inside the 64-bit proposal word, k_alt_resolve runs over ALL
    chains after the exchange on every rank: merged streams == one context == oracle."""
    from spring_amd.pool import VirtualPool
    sa = _sa()
    n, L, G, K = 40_000, 150, 2_000, 504
    with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=2, deep_bins=1, alternatives=2)) as st:
        st.load_synth(n, L, G, 23, 10000)
        one = st.run().streams()
        dna = st.download_dna()
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_rounds(read, ln, L, K, 2, alternatives=2)
    _same(one, want, "one context")
    vp = VirtualPool(ranks, K, 2, deep_bins=1, alternatives=2)
    try:
        got = vp.run(lambda s: s.load_dna(dna, n, L))
    finally:
        vp.close()
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), ("virtual ranks", ranks, k)


def test_library_choice():
    """alternatives < 0 (like 0 outside this test suite, see conftest.py) = the library's choice: two candidates on a
    contended pool (a quarter of the dictionary's reads in bins of >= 64 entries), one elsewhere; stats.alternatives says
    which."""
    sa = _sa()
    for G, expect in ((200, 2), (100_000, 1), (6_000_000, 1)):
        n, L, K = 40_000, 150, 200
        with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=2, alternatives=-1)) as st:
            st.load_synth(n, L, G, 23, 10000)
            got = st.run().streams()
            dna = st.download_dna()
        assert got["stats"]["alternatives"] == expect, (G, got["stats"]["alternatives"])
        read, ln = po.load_dna(dna, n, L)
        _same(got, po.reorder_rounds(read, ln, L, K, 2, alternatives=expect), ("library's choice", G))


def test_refused_where_it_cannot_run():
    sa = _sa()
    dna, n, L = named_set("syn2k_100")
    with pytest.raises(sa.ReorderError):
        sa.reorder_dna(dna, n, L, sa.ReorderOpts(num_chains=8, alternatives=2, fused=-1))  # the two-kernel round has no second pass
