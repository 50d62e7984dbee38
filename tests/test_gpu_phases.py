"""The schedule with two chain groups (`opts.phases = 2`): the chains run as two groups whose rounds
alternate, one group's round kernel beside the other's, each group searching on the pool as it was after its own last
round.  Executable specification: oracle/reorder_oracle.c::orc_reorder_rounds_ph (its invariants are checked on CPU in
tests/test_oracle.py).  Here: the HIP path through the C ABI == that oracle, bit for bit -- which also says that two
kernels running side by side on two streams never read what the other writes (any such race would show as a difference
from the sequential oracle, or between repeated runs)."""
import numpy as np
import pytest

from helpers import KEYS, check_invariants
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _sa():
    import spring_amd
    return spring_amd


def _same(a, b, what):
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), (what, k, len(a[k]), len(b[k]))
    assert np.array_equal(a["tid_off"], b["tid_off"]) and np.array_equal(a["tid_off_s"], b["tid_off_s"]), what


def _run(n, L, G, K, T, seed=23, err=10000, fused=3, **kw):
    sa = _sa()
    with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=T, fused=fused, deep_bins=-1, **kw)) as st:
        st.load_synth(n, L, G, seed, err)
        got = st.run().streams()
        dna = st.download_dna()
    return got, dna


@pytest.mark.parametrize("n,L,cov,K,T", [(300_000, 100, 30, 4096, 8), (200_000, 150, 25, 6144, 3), (160_000, 150, 60, 8192, 2),
                                         (100_000, 64, 20, 5000, 1), (60_000, 251, 25, 4096, 2), (40_000, 100, 25, 12288, 5)])
def test_two_groups_vs_oracle(n, L, cov, K, T):
    """Group sizes that differ (6144 -> 4096 + 2048; 5000 -> 2048 + 2952), chains that outnumber the reads of a seed range
    (12288 chains on 40 000 reads: most finish at once), read lengths of both instantiations of the round kernel."""
    got, dna = _run(n, L, n * L // cov, K, T, phases=2)
    assert got["stats"]["phases"] == 2 and got["stats"]["chains"] == K
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_rounds_ph(read, ln, L, K, T)
    _same(got, want, (n, L, K))
    assert got["stats"]["unmatched"] == want["stats"]["unmatched"] and got["stats"]["lost"] == want["stats"]["lost"]
    check_invariants(got, read, ln, L, n)
    # the other mapping of a round to the hardware (one chain per wavefront), and its counting build: the same streams and
    # the oracle's reference-equivalent work counters
    one, _ = _run(n, L, n * L // cov, K, T, phases=2, fused=2)
    _same(one, want, ("one chain per wavefront", n, L, K))
    cnt, _ = _run(n, L, n * L // cov, K, T, phases=2, fused=0, collect_stats=True)
    _same(cnt, want, ("counting build", n, L, K))
    assert cnt["stats"]["phases"] == 2
    for k in ("unmatched", "probes", "keyok", "cands", "hits", "iterations", "lost"):
        assert cnt["stats"][k] == want["stats"][k], (k, cnt["stats"][k], want["stats"][k])


def test_repeated_runs_and_batch_sizes_agree():
    """The same streams whatever the host does around the launches: rounds per look at the running chains, HIP events around
    every round kernel (opts.time_search), repeated runs."""
    n, L, K = 250_000, 150, 4096
    ref, dna = _run(n, L, n * L // 25, K, 4, phases=2)
    for kw in (dict(), dict(rounds_per_sync=1), dict(rounds_per_sync=5), dict(time_search=True), dict()):
        got, _ = _run(n, L, n * L // 25, K, 4, phases=2, **kw)
        _same(got, ref, kw)
        if kw.get("time_search"):
            assert got["stats"]["search_launches"] >= 2 and got["stats"]["ms_search_kernel"] > 0
    read, ln = po.load_dna(dna, n, L)
    _same(ref, po.reorder_rounds_ph(read, ln, L, K, 4), "oracle")


def test_contended_pool_two_groups():
    """A few thousand chains on a genome of a few kilobases: most proposals are lost, many of them to the OTHER group (a read
    it took between this group's search and its mark step)."""
    n, L, G, K = 120_000, 100, 3_000, 4096
    got, dna = _run(n, L, G, K, 2, phases=2)
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_rounds_ph(read, ln, L, K, 2)
    _same(got, want, "contended")
    assert got["stats"]["lost"] == want["stats"]["lost"] and want["stats"]["lost"] > 0


@pytest.mark.parametrize("A", [1, 2])
@pytest.mark.parametrize("budget,flags", [(-1, 0), (0, 0), (1, 0), (3, -1), (-1, -1)])
@pytest.mark.parametrize("n,L,G,K", [(120_000, 100, 3_000, 4096), (90_000, 150, 900, 4300), (150_000, 120, 40_000, 6144)])
def test_two_groups_deep_bin_kernels(n, L, G, K, A, budget, flags):
    """The deep-bin machinery under two chain groups: bin entries that carry their read's taken bit (set once BOTH groups'
    views have the read; a clear flag asks the group's bitmap), dead tails trimmed only where they are dead for both
    groups, bin compaction where the groups meet, the long-search kernels with a queue per group (budget 1: nearly every
    search over a multi-read bin), two candidates per proposal (k_ph_alt_resolve).  Bins of hundreds to thousands of reads,
    some beyond MAX_SEARCH_REORDER."""
    sa = _sa()
    with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=2, deep_bins=1, long_budget=budget, long_split=1 if budget == 1 else 0,
                                        entry_flags=flags, alternatives=A, phases=2)) as st:
        st.load_synth(n, L, G, 23, 10000)
        got = st.run().streams()
        dna = st.download_dna()
    assert got["stats"]["phases"] == 2 and got["stats"]["alternatives"] == A
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_rounds_ph(read, ln, L, K, 2, alternatives=A)
    _same(got, want, ("deep", n, K, A, budget, flags))
    assert got["stats"]["lost"] == want["stats"]["lost"] and got["stats"]["unmatched"] == want["stats"]["unmatched"]
    if budget > 0:
        assert got["stats"]["long_searches"] > 0
    if budget == 0 and flags == 0:  # the counting build (serial walks instead of the balanced scan) and its work counters
        with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=2, deep_bins=1, alternatives=A, phases=2, collect_stats=True)) as st:
            st.load_synth(n, L, G, 23, 10000)
            cnt = st.run().streams()
        _same(cnt, want, ("deep, counting build", n, K, A))
        for k in ("unmatched", "probes", "keyok", "cands", "hits", "iterations", "lost"):
            assert cnt["stats"][k] == want["stats"][k], (k, cnt["stats"][k], want["stats"][k])


def test_one_group_schedule_untouched_and_library_choice():
    """phases = 1 is the rounds schedule of every other test; the library's own choice (negative, like 0 outside this test
    suite: conftest.py) is one group below 16 384 chains and two from there on."""
    n, L, K = 120_000, 100, 4096
    one, dna = _run(n, L, n * L // 25, K, 2, phases=1)
    auto, _ = _run(n, L, n * L // 25, K, 2, phases=-1)
    assert one["stats"]["phases"] == 1 and auto["stats"]["phases"] == 1
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_rounds(read, ln, L, K, 2)
    _same(one, want, "phases = 1")
    _same(auto, want, "library's choice")
    sa = _sa()
    n, K = 200_000, 16384
    with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=2, phases=-1)) as st:  # (kernel mapping and schedule: the library's)
        st.load_synth(n, L, n * L // 25, 23, 10000)
        auto = st.run().streams()
        dna = st.download_dna()
    assert auto["stats"]["phases"] == 2
    read, ln = po.load_dna(dna, n, L)
    _same(auto, po.reorder_rounds_ph(read, ln, L, K, 2), "library's choice, 16 384 chains")
    # a pool of very deep bins, whose long searches go to the k_long kernels: one group, whatever the chain count
    with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=2, phases=-1, alternatives=1)) as st:
        st.load_synth(n, L, 300, 23, 10000)
        auto = st.run().streams()
        dna = st.download_dna()
    assert auto["stats"]["phases"] == 1
    read, ln = po.load_dna(dna, n, L)
    _same(auto, po.reorder_rounds(read, ln, L, K, 2), "library's choice, very deep bins")


def test_refused_where_it_cannot_run():
    sa = _sa()
    for kw in (dict(num_chains=1024, fused=3), dict(num_chains=4096, fused=-1), dict(num_chains=4096, force_literal_update=True),
               dict(num_chains=65536)):  # (the last: fewer reads than chains -- only chain 0 would run, the second group none)
        with sa.ReorderStage(sa.ReorderOpts(num_thr=1, phases=2, **kw)) as st:
            st.load_synth(50_000, 100, 200_000, 5, 10000)
            st.build_dict()
            with pytest.raises(sa.ReorderError):
                st.run_chains()


# ---- two chain groups in the multi-GPU pool path: a rank owns a slice of EACH group, so the groups -- and the output -- are the
# same whatever the number of ranks ("same K => same output for G = 1, 2, 4, 8").  G virtual ranks on one device.

@pytest.mark.parametrize("G", [1, 2, 4])
@pytest.mark.parametrize("K,fused,kw", [(8192, 3, dict(phases=2)), (6144, 0, dict(phases=2)), (16384, 3, dict(phases=-1)),
                                        (16384, 0, dict(phases=-1, collect_stats=True)),
                                        (8192, 0, dict(phases=2, alternatives=2, deep_bins=1))])
def test_pool_with_two_groups_is_independent_of_gpu_count(G, K, fused, kw):
    """phases = 2 asked for, and phases = -1 (the library's choice: two groups from 16 384 chains on -- NOT pinned to one group
    as the older pool tests are): G ranks == one context == the two-group oracle, every stream and the per-tid offsets; the
    counting build's work counters add up over the ranks."""
    from spring_amd.pool import VirtualPool
    sa = _sa()
    n, L, T = 60_000, 100, 3
    A = kw.get("alternatives", 1)
    gen = n * L // (400 if A == 2 else 25)
    okw = dict(deep_bins=-1, fused=fused)
    okw.update(kw)
    with sa.ReorderStage(sa.ReorderOpts(num_chains=K, num_thr=T, **okw)) as st:
        st.load_synth(n, L, gen, 23, 10000)
        single = st.run().streams()
        dna = st.download_dna()
    assert single["stats"]["phases"] == 2 and single["stats"]["chains"] == K
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_rounds_ph(read, ln, L, K, T, A)
    _same(single, want, ("single", K, kw))
    vp = VirtualPool(G, K, T, **okw)
    try:
        got = vp.run(lambda s: s.load_synth(n, L, gen, 23, 10000))
    finally:
        vp.close()
    assert all(int(ps["phases"]) == 2 for ps in got["per_rank_stats"])
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), (G, K, k)
    assert np.array_equal(got["tid_off"], want["tid_off"]) and np.array_equal(got["tid_off_s"], want["tid_off_s"])
    if kw.get("collect_stats"):
        for k in ("probes", "keyok", "cands", "hits", "unmatched", "lost"):
            assert sum(int(ps[k]) for ps in got["per_rank_stats"]) == want["stats"][k], (k, G)


@pytest.mark.parametrize("devices", [(), (0, 0), (0, 0, 0, 0)])
def test_call_reorder_device_lists_at_the_default_schedule(tmp_path, devices):
    """The drop-in with the library's own choice of the schedule (phases = -1; two groups at this chain count) on one device
    and on device lists: ONE file set, the two-group oracle's -- the default output no longer depends on the number of GPUs."""
    from test_gpu_parity import _check_file_set, _read_file_set
    from spring_amd.reorder import CompressionParams
    sa = _sa()
    n, L, K, T = 200_000, 100, 16384, 5
    dna = sa.synth_dna_host(n, L, n * L // 25, 31, 10000)
    (tmp_path / "input_clean_1.dna").write_bytes(dna)
    sa.call_reorder(str(tmp_path), CompressionParams(L, [n, 0], num_thr=T), sa.ReorderOpts(num_chains=K, num_thr=T, phases=-1, devices=devices))
    read, ln = po.load_dna(dna, n, L)
    want = po.reorder_rounds_ph(read, ln, L, K, T)
    _check_file_set(_read_file_set(tmp_path, T), want, read, ln, L, T)


def test_default_chain_count_is_a_multiple_of_2048_where_two_groups_run():
    """From 16 384 chains on the default chain count is a multiple of 2048, so a pool over 2, 4 or 8 GPUs cuts it into the same
    two groups as one GPU does (spring_reorder_auto_chains)."""
    sa = _sa()
    with sa.ReorderStage(sa.ReorderOpts(phases=-1)) as st:
        st.load_synth(20_000_000, 100, 20_000_000 * 100 // 25, 5, 10000)
        st.build_dict()
        k, _ = st.auto_chains()
    assert k == (20_000_000 >> 10) // 2048 * 2048
