"""CPU tests of the oracle itself (no GPU): known answers recorded from the reference,
K=1 rounds schedule == literal serial restatement, invariants for K>1, golden fixtures."""
import os

import numpy as np
import pytest

import readsets as rs
from helpers import GOLDEN, KEYS, SMALL_SETS, check_invariants, named_set
from oracle import pyoracle as po


def _load(name):
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    return read, ln, n, L


def test_dict_windows_match_survey_table():
    # SURVEY.md section 8 header: L=100 -> [18..49],[50..81]; L=150 -> [43..74],[75..106]
    assert po.dict_windows(100) == ([18, 50], [49, 81])
    assert po.dict_windows(150) == ([43, 75], [74, 106])
    assert po.limbs(100) == 4 and po.limbs(150) == 5 and po.limbs(511) == 16 and po.limbs(32) == 1


def test_reference_fixture_test_1_all_singletons():
    """util/test_1.fastq: 38 clean reads, all singletons, order 0,37,36,...,1."""
    read, ln, n, L = _load("test_1")
    assert n == 38
    r = po.reorder_serial(read, ln, L)
    assert len(r["order"]) == 0
    assert r["order_s"].tolist() == [0] + list(range(37, 0, -1))
    assert r["stats"]["unmatched"] == 38


@pytest.mark.slow
def test_known_answer_1M_100bp_reference_counters():
    """counters the surveyor recorded from the real reference
    (instrumented reorder.h, -t 1) on numpy default_rng(7), G=4 Mb, 1 M x 100 bp, 1 % subs."""
    n, L = 1_000_000, 100
    a = rs.np_reads(7, 4_000_000, n, L, 0.01)
    read, ln = po.load_dna(rs.pack_fixed(a), n, L)
    r = po.reorder_serial(read, ln, L)
    st = r["stats"]
    assert st["unmatched"] == 96_389
    assert len(r["order_s"]) == 93_011
    assert st["search_calls"] == 28_782_325
    assert st["probes"] == 44_478_214
    assert st["keyok"] == 927_453
    assert st["cands"] == 928_070
    assert st["hits"] == 903_611
    assert st["updates"] == 1_096_389
    check_invariants(r, read, ln, L, n)


@pytest.mark.slow
def test_known_answer_1M_150bp_reference_counters():
    """1 M x 150 bp (default_rng(11), G = 6 Mb)."""
    n, L = 1_000_000, 150
    a = rs.np_reads(11, 6_000_000, n, L, 0.01)
    read, ln = po.load_dna(rs.pack_fixed(a), n, L)
    r = po.reorder_serial(read, ln, L)
    st = r["stats"]
    assert st["unmatched"] == 137_664
    assert len(r["order_s"]) == 127_731
    assert st["probes"] == 92_923_476
    assert st["cands"] == 996_385


@pytest.mark.parametrize("name", SMALL_SETS)
def test_rounds_k1_equals_serial(name):
    read, ln, n, L = _load(name)
    a = po.reorder_serial(read, ln, L)
    b = po.reorder_rounds(read, ln, L, 1, 1)
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), (name, k)
    assert a["stats"]["unmatched"] == b["stats"]["unmatched"]
    for k in ("search_calls", "probes", "keyok", "cands", "hits", "updates", "iterations"):
        assert a["stats"][k] == b["stats"][k], (name, k)
    check_invariants(a, read, ln, L, n)


@pytest.mark.parametrize("name", ["syn2k_100", "syn5k_150", "var2k", "heavy", "repeat10k", "dups", "test_1+2"])
@pytest.mark.parametrize("K,T", [(2, 1), (5, 2), (64, 8), (1000, 3)])
def test_rounds_invariants(name, K, T):
    read, ln, n, L = _load(name)
    r = po.reorder_rounds(read, ln, L, K, T)
    check_invariants(r, read, ln, L, n)
    assert len(r["tid_off"]) == T + 1 and int(r["tid_off"][-1]) == len(r["order"])
    # deterministic
    r2 = po.reorder_rounds(read, ln, L, K, T)
    for k in KEYS:
        assert np.array_equal(r[k], r2[k])


def test_more_chains_than_reads():
    read, ln, n, L = _load("two_same")
    r = po.reorder_rounds(read, ln, L, 16, 4)  # floor(N/K)=0: only chain 0 starts (reorder.h:411-420)
    assert r["order"].tolist() == [0, 1] and len(r["order_s"]) == 0
    assert bytes(r["flag"]) == b"01"


def test_write_dna_stream_roundtrip():
    read, ln, n, L = _load("var2k")
    r = po.reorder_serial(read, ln, L)
    s = po.write_dna_stream(read, ln, L, r["order"], r["rc"])
    back, bl = po.load_dna(s, len(r["order"]), L)
    d = r["rc"] == ord("d")
    assert np.array_equal(back[d], read[r["order"][d]])
    assert np.array_equal(bl, ln[r["order"]])
    # 'r' records are reverse complements: applying the transform twice gives the read back
    rr = np.flatnonzero(~d)
    if len(rr):
        s2 = po.write_dna_stream(back, bl, L, rr.astype(np.uint32), np.full(len(rr), ord("r"), np.uint8))
        b2, _ = po.load_dna(s2, len(rr), L)
        assert np.array_equal(b2, read[r["order"][rr]])


@pytest.mark.parametrize("name", ["syn2k_100", "var2k", "heavy"])
def test_golden_fixtures(name):
    """tests/golden/*.npz were written by tests/golden/make_golden.py from the oracle after it
    reproduced the reference's recorded counters; they freeze the expected byte streams."""
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    read, ln, n, L = _load(name)
    a = po.reorder_serial(read, ln, L)
    for k in KEYS:
        assert np.array_equal(a[k], g[k]), (name, k)
    for K in (4, 64):
        b = po.reorder_rounds(read, ln, L, K, 2)
        for k in KEYS:
            assert np.array_equal(b[k], g["K%d_%s" % (K, k)]), (name, K, k)


@pytest.mark.parametrize("name", ["syn5k_150", "var2k", "heavy", "repeat10k"])
@pytest.mark.parametrize("T", [1, 4])
def test_openmp_port_invariants(name, T):
    """CPU-baseline port (free-running threads, non-deterministic for T > 1 like the reference):
    T = 1 equals the serial restatement; any T yields a valid reordering."""
    read, ln, n, L = _load(name)
    r = po.reorder_omp(read, ln, L, T)
    check_invariants(r, read, ln, L, n)
    if T == 1:
        a = po.reorder_serial(read, ln, L)
        for k in KEYS:
            assert np.array_equal(a[k], r[k]), (name, k)


def test_alternatives_schedule_specification():
    """orc_reorder_rounds_alt: A candidates per match proposal resolved in A passes (the costed remedy for contended
    pools).  A = 1 is the schedule the GPU runs; for any A, K = 1 equals the serial order, the
    output keeps every invariant, and lost proposals and rounds go down as A goes up."""
    n, L = 20_000, 100
    dna = rs.pack_fixed(rs.np_reads(3, n * L // 400, n, L, 0.01))
    read, ln = po.load_dna(dna, n, L)
    ser = po.reorder_serial(read, ln, L)
    base = po.reorder_rounds(read, ln, L, 256, 3)
    prev = None
    for A in (1, 2, 4):
        one = po.reorder_rounds(read, ln, L, 1, 1, A)
        for k in KEYS:
            assert np.array_equal(one[k], ser[k]), (A, k)
        r = po.reorder_rounds(read, ln, L, 256, 3, A)
        check_invariants(r, read, ln, L, n)
        if A == 1:
            for k in KEYS:
                assert np.array_equal(r[k], base[k]), k
        else:
            assert r["stats"]["lost"] < prev["stats"]["lost"] and r["stats"]["rounds"] <= prev["stats"]["rounds"]
        prev = r


def test_two_group_schedule_specification():
    """orc_reorder_rounds_ph (opts.phases = 2): every read claimed exactly once and every emitted record consistent with
    its contig (check_invariants), whatever the split of the chains; the group split is what the header says."""
    assert [po.lib().orc_phase_split(k) for k in (4096, 4100, 6144, 8191, 65536, 100000)] == [2048, 2048, 4096, 4096, 32768, 49152]
    for seed, n, L, G, K, T in ((5, 20_000, 100, 60_000, 4096, 3), (6, 30_000, 150, 4_000, 6144, 2), (7, 9_000, 64, 30_000, 4100, 1)):
        a = rs.np_reads(seed, G, n, L, 0.01)
        read, ln = po.load_dna(rs.pack_fixed(a), n, L)
        r = po.reorder_rounds_ph(read, ln, L, K, T)
        check_invariants(r, read, ln, L, n)
        assert len(r["order"]) + len(r["order_s"]) == n
        r2 = po.reorder_rounds_ph(read, ln, L, K, T, alternatives=2)  # two candidates per proposal: fewer lost proposals
        check_invariants(r2, read, ln, L, n)
        assert len(r2["order"]) + len(r2["order_s"]) == n and r2["stats"]["lost"] <= r["stats"]["lost"]
        one = po.reorder_rounds(read, ln, L, K, T)
        # same pool, same chain count: the two schedules find contigs of the same kind (within a few per cent)
        assert abs(int(r["stats"]["unmatched"]) - int(one["stats"]["unmatched"])) <= 0.05 * n


@pytest.mark.parametrize("name", ["syn2k_100", "syn5k_150", "var2k", "var_long", "var_short", "heavy", "repeat10k", "dups",
                                  "tandem", "syn2k_251", "syn1k_511", "syn2k_20", "syn3k_64", "test_1+2"])
def test_replay_check_accepts_every_schedule(name):
    """orc_check_contigs (the schedule-independent replay of a reorder output through the reference's state machine) finds
    nothing to object to in the serial order, the rounds schedule, two candidates per proposal and free-running threads."""
    read, ln, n, L = _load(name)
    outs = [("serial", po.reorder_serial(read, ln, L)), ("rounds", po.reorder_rounds(read, ln, L, 16, 3)),
            ("two candidates", po.reorder_rounds(read, ln, L, 64, 2, alternatives=2)), ("omp", po.reorder_omp(read, ln, L, 4))]
    for what, r in outs:
        c = po.check_contigs(read, ln, L, r)
        assert c["bad"] == 0, (name, what, c)
        assert c["contigs"] == int((r["flag"] == ord("0")).sum()) and c["matches"] == len(r["order"]) - c["contigs"]
        if po.ref_units() is not None:  # the same replay with the consensus kept by the reference's own updaterefcount<N>
            assert po.check_contigs(read, ln, L, r, reference_update=True) == c, (name, what)


def test_replay_check_rejects_corrupted_outputs():
    """... and objects to an output that is a permutation of the reads with consistent flags (check_invariants passes) but
    whose records are not matches: a read moved into another contig, a position off by one, an orientation flipped."""
    read, ln, n, L = _load("syn5k_150")
    good = po.reorder_rounds(read, ln, L, 16, 3)
    assert po.check_contigs(read, ln, L, good)["bad"] == 0
    m = np.flatnonzero(good["flag"] == ord("1"))
    rng = np.random.default_rng(3)
    for trial in range(20):
        r = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in good.items()}
        kind = trial % 4
        i, j = (int(x) for x in rng.choice(m, 2, replace=False))
        if kind == 0:
            r["order"][i], r["order"][j] = r["order"][j], r["order"][i]     # two matched reads change places
            r["rlen"][i], r["rlen"][j] = r["rlen"][j], r["rlen"][i]
        elif kind == 1:
            r["pos"][i] += 1
        elif kind == 2:
            r["rc"][i] = ord("r") if r["rc"][i] == ord("d") else ord("d")
        else:
            r["pos"][i] -= 3
        check_invariants(r, read, ln, L, n)  # (still a permutation with consistent flags)
        c = po.check_contigs(read, ln, L, r)
        assert c["bad"] >= 1, (trial, kind, i, j, c)
        if po.ref_units() is not None:
            assert po.check_contigs(read, ln, L, r, reference_update=True)["bad"] >= 1, (trial, kind)


def test_replay_check_random_geometries():
    """The replay check on the fuzz generator's read sets (read lengths 20 .. 511, fixed / variable, duplicates, repeats, reads
    outside one or both dictionaries, 1 .. 4 096 chains): nothing to object to in the rounds oracle's outputs, with the
    restated and with the reference's own updaterefcount."""
    from test_gpu_fuzz import _random_case
    for seed in range(1000, 1060):
        dna, n, L, K, T = _random_case(seed)
        read, ln = po.load_dna(dna, n, L)
        r = po.reorder_rounds(read, ln, L, K, T)
        c = po.check_contigs(read, ln, L, r)
        assert c["bad"] == 0, (seed, n, L, K, c)
        if po.ref_units() is not None:
            assert po.check_contigs(read, ln, L, r, reference_update=True) == c, (seed, n, L, K)
