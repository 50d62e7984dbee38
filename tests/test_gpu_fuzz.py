"""Randomised differential test: the HIP path vs the oracle on many small random read sets with random
geometry (read length, fixed/variable, error rate, duplicates, coverage) and random schedule parameters
(chains K, output sets T).  Bit-exact or fail; the seed of a failing case is in the assertion message."""
import os

import numpy as np
import pytest

import readsets as rs
from helpers import KEYS, check_invariants
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _random_case(seed, nlo=1, nhi=2500, Ks=(1, 2, 3, 7, 16, 63, 256, 1000, 4096)):
    rng = np.random.default_rng(seed)
    kind = rng.integers(0, 5)
    L = int(rng.choice([20, 33, 50, 64, 75, 100, 101, 127, 128, 150, 151, 200, 250, 300, 400, 511]))
    n = int(rng.integers(nlo, nhi))
    err = float(rng.choice([0.0, 0.002, 0.01, 0.03, 0.08]))
    cov = int(rng.choice([2, 8, 25, 60, 400]))
    G = max(n * L // cov, L + 5)
    if kind == 0:  # fixed length
        dna = rs.pack_fixed(rs.np_reads(seed, G, n, L, err)); maxlen = L
    elif kind == 1:  # variable length
        lmin = int(rng.integers(1, L + 1))
        reads = rs.var_length_reads(seed, max(G, L + 5), n, lmin, L, err)
        dna = rs.pack_var(reads); maxlen = max(len(r) for r in reads)
    elif kind == 2:  # heavy duplicates
        base = rs.np_reads(seed, max(G // 20, L + 5), max(n // 20, 1), L, 0.0)
        a = np.repeat(base, 20, axis=0)[:n]
        flip = rng.random(a.shape) < err
        a = np.where(flip, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, a.shape)], a).astype(np.uint8)
        rng.shuffle(a, axis=0)
        n = a.shape[0]; dna = rs.pack_fixed(a); maxlen = L
    elif kind == 3:  # repeat-rich genome
        dna = rs.pack_fixed(rs.np_reads_repeat(seed, max(G, 16 * L), n, L, err)); maxlen = L
    else:  # mostly short reads + a few at max length (many reads outside one or both dictionaries)
        reads = rs.var_length_reads(seed, max(G, L + 5), n, 1, max(L // 2, 1), err)
        reads += rs.var_length_reads(seed + 1, max(G, L + 5), max(n // 10, 1), L, L, err)
        n = len(reads); dna = rs.pack_var(reads); maxlen = L
    K = int(rng.choice(list(Ks)))
    T = int(rng.choice([1, 2, 3, 8]))
    return dna, n, maxlen, K, T


_BLOCKS = int(os.environ.get("SPRING_FUZZ_BLOCKS", "8"))  # 25 cases per block; raise for a long one-off sweep


@pytest.mark.parametrize("block", range(_BLOCKS))
def test_fuzz_gpu_equals_oracle(block):
    import spring_amd
    for seed in range(1000 + 25 * block, 1000 + 25 * (block + 1)):
        dna, n, L, K, T = _random_case(seed)
        read, ln = po.load_dna(dna, n, L)
        want = po.reorder_rounds(read, ln, L, K, T)
        got = spring_amd.reorder_dna(dna, n, L, spring_amd.ReorderOpts(num_chains=K, num_thr=T, collect_stats=True))
        for k in KEYS:
            assert np.array_equal(got[k], want[k]), ("seed", seed, "n", n, "L", L, "K", K, "T", T, k)
        assert np.array_equal(got["tid_off"], want["tid_off"]), ("seed", seed)
        for k in ("probes", "keyok", "cands", "hits", "iterations", "lost", "unmatched"):
            assert got["stats"][k] == want["stats"][k], ("seed", seed, k, got["stats"][k], want["stats"][k])
        check_invariants(got, read, ln, L, n)
        assert po.check_contigs(read, ln, L, got)["bad"] == 0, ("seed", seed, "replay check")
        if K == 1:
            ser = po.reorder_serial(read, ln, L)
            for k in KEYS:
                assert np.array_equal(got[k], ser[k]), ("seed", seed, "serial", k)
        # the production build (no counters): four chains per wavefront on shallow dictionaries (fused = 3: the library
        # itself takes that kernel from 49 152 chains on); with the deep-bin variant forced on, the balanced scan and the
        # resumed searches; and the one-chain-per-wavefront round (fused = 0: the automatic choice at these sizes, or 2)
        rng = np.random.default_rng(seed + 77)
        kw = dict(deep_bins=int(rng.choice([0, 1, -1])), fused=int(rng.choice([3, 3, 0, 2])))
        if kw["deep_bins"] == 1:  # searches handed to k_long after 1 / 2 / the default number of compare passes, or never
            kw["long_budget"] = int(rng.choice([1, 1, 2, 0, -1]))
        got2 = spring_amd.reorder_dna(dna, n, L, spring_amd.ReorderOpts(num_chains=K, num_thr=T, **kw))
        for k in KEYS:
            assert np.array_equal(got2[k], want[k]), ("seed", seed, "no-stats", kw, "n", n, "L", L, "K", K, "T", T, k)
        assert np.array_equal(got2["tid_off"], want["tid_off"]), ("seed", seed, "no-stats", kw)
        for k in ("lost", "unmatched"):
            assert got2["stats"][k] == want["stats"][k], ("seed", seed, "no-stats", kw, k)
        # two candidates per proposal (the alternatives schedule: deep-bin kernel variants, k_alt_resolve; K = 1 stays serial),
        # production or counting build, long searches handed over after 1 / 2 / the default number of passes or never, cut
        # into parts of 1 / 3 chunks or the default
        if seed % 2 == 0:
            kw3 = dict(alternatives=2, collect_stats=bool(rng.integers(0, 2)), long_budget=int(rng.choice([1, 2, 0, -1])),
                       long_split=int(rng.choice([0, 1, 3])))
            want3 = po.reorder_rounds(read, ln, L, K, T, alternatives=2)
            got3 = spring_amd.reorder_dna(dna, n, L, spring_amd.ReorderOpts(num_chains=K, num_thr=T, **kw3))
            for k in KEYS:
                assert np.array_equal(got3[k], want3[k]), ("seed", seed, "two candidates", kw3, "n", n, "L", L, "K", K, "T", T, k)
            assert np.array_equal(got3["tid_off"], want3["tid_off"]), ("seed", seed, "two candidates", kw3)
            for k in ("lost", "unmatched") + (("probes", "keyok", "cands", "hits") if kw3["collect_stats"] else ()):
                assert got3["stats"][k] == want3["stats"][k], ("seed", seed, "two candidates", kw3, k)


@pytest.mark.parametrize("block", range(max(_BLOCKS // 4, 1)))
def test_fuzz_two_chain_groups(block):
    """The schedule with two chain groups (opts.phases = 2, specification orc_reorder_rounds_ph[_alt]) on random read sets of
    8 192 .. 30 000 reads with 4 096 .. 12 288 chains (a few reads per chain: seed ranges run dry, groups of unequal size, chains
    that never get a seed), a random kernel variant per case: four chains / one chain per wavefront, the deep-bin machinery with
    or without entry flags, long searches handed to the long-search kernels after 1 / 2 passes or never, one or two candidates
    per proposal, production or counting build."""
    import spring_amd
    for seed in range(7000 + 10 * block, 7000 + 10 * (block + 1)):
        dna, n, L, K, T = _random_case(seed, 8192, 30000, (4096, 4100, 5000, 6144, 8192, 12288))
        if n < 8192:  # (the duplicate-heavy generator rounds n down)
            continue
        if n < K:  # fewer reads than chains: only chain 0 would run -- the schedule is refused (one group is what runs then)
            with pytest.raises(spring_amd.ReorderError):
                spring_amd.reorder_dna(dna, n, L, spring_amd.ReorderOpts(num_chains=K, num_thr=T, phases=2))
            auto = spring_amd.reorder_dna(dna, n, L, spring_amd.ReorderOpts(num_chains=K, num_thr=T, phases=-1))
            assert auto["stats"]["phases"] == 1
            read, ln = po.load_dna(dna, n, L)
            want = po.reorder_rounds(read, ln, L, K, T)
            for k in KEYS:
                assert np.array_equal(auto[k], want[k]), ("seed", seed, "fewer reads than chains", k)
            continue
        rng = np.random.default_rng(seed + 5)
        A = int(rng.choice([1, 1, 2]))
        deep = 1 if A == 2 else int(rng.choice([1, -1, -1]))
        kw = dict(deep_bins=deep, alternatives=A, collect_stats=bool(rng.integers(0, 2)))
        if deep == 1:
            kw.update(long_budget=int(rng.choice([1, 2, 0, -1])), long_split=int(rng.choice([0, 1, 3])), entry_flags=int(rng.choice([0, -1])))
        else:
            kw.update(fused=int(rng.choice([3, 3, 0, 2])))
        read, ln = po.load_dna(dna, n, L)
        want = po.reorder_rounds_ph(read, ln, L, K, T, alternatives=A)
        got = spring_amd.reorder_dna(dna, n, L, spring_amd.ReorderOpts(num_chains=K, num_thr=T, phases=2, **kw))
        assert got["stats"]["phases"] == 2
        for k in KEYS:
            assert np.array_equal(got[k], want[k]), ("seed", seed, "two groups", kw, "n", n, "L", L, "K", K, "T", T, k)
        assert np.array_equal(got["tid_off"], want["tid_off"]), ("seed", seed, "two groups", kw)
        for k in ("lost", "unmatched") + (("probes", "keyok", "cands", "hits", "iterations") if kw["collect_stats"] else ()):
            assert got["stats"][k] == want["stats"][k], ("seed", seed, "two groups", kw, k, got["stats"][k], want["stats"][k])
        check_invariants(got, read, ln, L, n)


@pytest.mark.parametrize("block", range(2))
def test_fuzz_single_pool_virtual_ranks(block):
    from spring_amd.pool import VirtualPool
    for seed in range(5000 + 6 * block, 5000 + 6 * (block + 1)):
        dna, n, L, K, T = _random_case(seed)
        G = int(np.random.default_rng(seed).choice([2, 3, 4]))
        K = max(G, (K // G) * G)
        read, ln = po.load_dna(dna, n, L)
        want = po.reorder_rounds(read, ln, L, K, T)
        vp = VirtualPool(G, K, T, fused=3 if seed % 2 else 0)  # four chains per wavefront / the automatic choice (one)
        try:
            got = vp.run(lambda s: s.load_dna(dna, n, L))
        finally:
            vp.close()
        for k in KEYS:
            assert np.array_equal(got[k], want[k]), ("seed", seed, "G", G, "K", K, k)


@pytest.mark.parametrize("block", range(2))
def test_fuzz_two_groups_over_virtual_ranks(block):
    """Two chain groups in the pool path (a rank owns a slice of each group): random read sets of 8 192 .. 30 000 reads, 4 096 ..
    12 288 chains cut into groups of unequal size, 2 or 4 virtual ranks, a random kernel variant (four chains / one chain per
    wavefront, deep-bin machinery, one or two candidates per proposal) == the two-group oracle == whatever the rank count."""
    from spring_amd.pool import VirtualPool
    for seed in range(7600 + 5 * block, 7600 + 5 * (block + 1)):
        dna, n, L, K, T = _random_case(seed, 8192, 30000, (4096, 6144, 8192, 12288))
        if n < max(8192, K):
            continue
        rng = np.random.default_rng(seed + 9)
        G = int(rng.choice([2, 4]))
        A = int(rng.choice([1, 1, 2]))
        deep = 1 if A == 2 else int(rng.choice([1, -1, -1]))
        kw = dict(deep_bins=deep, alternatives=A, phases=2)
        if deep != 1:
            kw.update(fused=int(rng.choice([3, 3, 0, 2])))
        read, ln = po.load_dna(dna, n, L)
        want = po.reorder_rounds_ph(read, ln, L, K, T, alternatives=A)
        vp = VirtualPool(G, K, T, **kw)
        try:
            got = vp.run(lambda s: s.load_dna(dna, n, L))
        finally:
            vp.close()
        for k in KEYS:
            assert np.array_equal(got[k], want[k]), ("seed", seed, "G", G, "K", K, kw, k)
        assert np.array_equal(got["tid_off"], want["tid_off"]) and np.array_equal(got["tid_off_s"], want["tid_off_s"]), ("seed", seed, G)


@pytest.mark.parametrize("block", range(max(_BLOCKS // 2, 1)))
def test_fuzz_encoder_equals_oracle(block):
    """Row f2: reorder on the GPU, then the encoder stage on the GPU vs the encoder oracle on the same streams,
    with random N reads (sometimes thousands of near-copies -> bins deeper than MAX_SEARCH_ENCODER)."""
    import spring_amd
    from helpers import interleave_order_N, make_N_reads, read_strings, same_encoding
    from spring_amd.encoder import EncoderStage
    for seed in range(9000 + 12 * block, 9000 + 12 * (block + 1)):
        dna, n, L, K, T = _random_case(seed)
        rng = np.random.default_rng(seed + 1)
        read, ln = po.load_dna(dna, n, L)
        nN = int(rng.choice([0, 0, 5, 60, 400]))
        deep = int(rng.choice([0, 0, 0, 1500, 2600])) if L > 50 else 0
        Nreads = make_N_reads(read_strings(read, ln), nN, seed, deep=deep) if (nN or deep) else []
        dnaN = po.pack_dnaN(Nreads)
        order_N = interleave_order_N(n, len(Nreads), seed + 2)
        with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=K, num_thr=T)) as st:
            st.load_dna(dna, n, L)
            st.run()
            streams = st.streams()
            with EncoderStage() as enc:
                info = enc.encode(st, dnaN, order_N)
                got = enc.streams()
        want = po.encode(read, ln, L, streams, num_thr=T, dnaN=dnaN, order_N=order_N)
        same_encoding(got, want, ("seed", seed, "n", n, "L", L, "K", K, "T", T, "nN", len(Nreads), "passes",
                                  info["align_passes"]))
