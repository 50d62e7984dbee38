"""CPU tests of the encoder oracle (oracle/encoder_oracle.c): the decompressor's inverse
(decompress.cpp:236-266) recovers every read, structure of the streams, golden fixtures."""
import os

import numpy as np
import pytest

from helpers import (ENC_KEYS, GOLDEN, decode_reads, interleave_order_N, make_N_reads, named_set, read_strings,
                     same_encoding, unpack_dnaN)
from oracle import pyoracle as po


def _case(name, K, T, nN, deep=0, seed=3):
    dna, n, L = named_set(name)
    read, ln = po.load_dna(dna, n, L)
    strs = read_strings(read, ln)
    Nreads = make_N_reads(strs, nN, seed, deep=deep)
    dnaN = po.pack_dnaN(Nreads)
    order_N = interleave_order_N(n, len(Nreads), seed + 7)
    streams = po.reorder_rounds(read, ln, L, K, T)
    enc = po.encode(read, ln, L, streams, num_thr=T, dnaN=dnaN, order_N=order_N)
    return enc, strs, Nreads, order_N, n, L


@pytest.mark.parametrize("name", ["syn2k_100", "syn5k_150", "syn3k_64", "syn2k_251", "syn2k_20", "var2k", "var_short",
                                  "heavy", "dups", "test_1+2", "one", "empty"])
@pytest.mark.parametrize("K,T,nN", [(1, 1, 0), (6, 3, 200)])
def test_decoding_the_streams_recovers_every_read(name, K, T, nN):
    enc, strs, Nreads, order_N, n, L = _case(name, K, T, nN)
    isN = np.zeros(n + len(Nreads), bool)
    isN[order_N] = True
    clean_pos = np.flatnonzero(~isN)
    orig = {int(clean_pos[i]): strs[i] for i in range(n)}
    orig.update({int(order_N[i]): Nreads[i] for i in range(len(Nreads))})
    dec = decode_reads(enc)
    for o, s in dec.items():
        assert orig[o] == s
    na = len(enc["pos"])
    un = unpack_dnaN(enc["unaligned"])
    assert len(un) == len(enc["order"]) - na
    for k, s in enumerate(un):
        assert orig[int(enc["order"][na + k])] == s
    assert sorted(enc["order"].tolist()) == list(range(n + len(Nreads)))
    assert enc["len_unaligned"] == sum(len(s) for s in un)
    assert len(enc["seq"]) == int(enc["seq_len_tid"].sum())
    assert enc["noise"].count(b"\n") == na and len(enc["noise"]) == na + len(enc["noisepos"])
    # positions inside a contig are non-decreasing in the output; every read lies inside the consensus
    if na:
        assert int((enc["pos"] + enc["rlen"][:na]).max()) <= len(enc["seq"])


def test_deep_bin_window_is_exercised():
    enc, strs, Nreads, order_N, n, L = _case("syn5k_150", 8, 2, 50, deep=3500, seed=5)
    # 3500 near-identical N reads share both dictionary bins: the first forward probes take 1000 each
    assert 2000 <= enc["matched_N"] < len(Nreads)


@pytest.mark.parametrize("name", ["syn2k_100", "var2k"])
def test_encoder_golden_fixture(name):
    """Fixtures written by tests/golden/make_golden.py from this oracle: guard against drift."""
    path = os.path.join(GOLDEN, "enc_%s.npz" % name)
    z = np.load(path)
    enc, *_ = _case(name, 6, 3, 120, seed=9)
    for k in ENC_KEYS:
        want = z[k]
        got = enc[k]
        if isinstance(got, bytes):
            got = np.frombuffer(got, np.uint8)
        assert np.array_equal(np.asarray(got).ravel(), want.ravel()), k
