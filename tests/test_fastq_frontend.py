"""FASTQ -> N split -> packed reads on the GPU vs the restatement of the sequence side of
preprocess() (read_fastq_block util.cpp:31-54, preprocess.cpp:186-214,:293-304, write_dna[N]_in_bits).
Uses the reference's own fixtures util/test_{1,2}.fastq (62 of 100 reads of test_1 contain N)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN
from oracle import pyoracle as po


def _fixture(name):
    return open(os.path.join(GOLDEN, name), "rb").read()


def _synth_fastq(seed, n, lmin, lmax, pn=0.1, crlf=False, final_newline=True):
    rng = np.random.default_rng(seed)
    eol = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        L = int(rng.integers(lmin, lmax + 1))
        r = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
        if rng.random() < pn and L:
            r[rng.integers(0, L, max(1, L // 20))] = ord("N")
        out += [b"@r%d" % i, r.tobytes(), b"+", b"I" * L]
    t = eol.join(out)
    return t + eol if final_newline else t


def test_oracle_preprocess_on_reference_fixture():
    t = _fixture("test_1.fastq")
    r = po.preprocess_fastq(t)
    assert (r["num_reads"], r["num_clean"], r["num_N"], r["max_readlen"]) == (100, 38, 62, 100)
    # the clean stream is what the reorder tests load (helpers.named_set packs the same reads independently)
    from helpers import named_set
    dna, n, L = named_set("test_1")
    assert r["clean"] == dna and n == 38
    # N reads: 4 bits per base, decodes back to the text
    lines = t.split(b"\n")
    p = 0
    for k, idx in enumerate(r["order_N"].tolist()):
        read = lines[4 * idx + 1]
        ln = r["ndna"][p] | (r["ndna"][p + 1] << 8)
        assert ln == len(read)
        body = r["ndna"][p + 2:p + 2 + (ln + 1) // 2]
        dec = bytes(b"AGCTN"[(body[i // 2] >> (4 * (i % 2))) & 15] for i in range(ln))
        assert dec == read
        p += 2 + (ln + 1) // 2
    assert p == len(r["ndna"])


def test_oracle_preprocess_errors_and_line_endings():
    with pytest.raises(ValueError, match="multiple of 4"):
        po.preprocess_fastq(b"@a\nACGT\n+\n")
    with pytest.raises(ValueError, match="Too long"):
        po.preprocess_fastq(b"@a\n" + b"A" * 600 + b"\n+\n" + b"I" * 600 + b"\n")
    a = po.preprocess_fastq(_synth_fastq(1, 50, 30, 80))
    b = po.preprocess_fastq(_synth_fastq(1, 50, 30, 80, crlf=True))
    c = po.preprocess_fastq(_synth_fastq(1, 50, 30, 80, final_newline=False))
    assert a["clean"] == b["clean"] == c["clean"] and a["ndna"] == b["ndna"] == c["ndna"]
    assert po.preprocess_fastq(b"")["num_reads"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["test_1", "test_2", "pe_fixture", "syn_var", "syn_crlf", "syn_nonl", "syn_big", "empty"])
def test_fastq_frontend_bit_exact(case):
    import spring_amd
    f2 = None
    if case == "test_1": f1 = _fixture("test_1.fastq")
    elif case == "test_2": f1 = _fixture("test_2.fastq")
    elif case == "pe_fixture": f1, f2 = _fixture("test_1.fastq"), _fixture("test_2.fastq")
    elif case == "syn_var": f1 = _synth_fastq(2, 3000, 0, 300, 0.2)
    elif case == "syn_crlf": f1 = _synth_fastq(3, 2000, 20, 150, 0.1, crlf=True)
    elif case == "syn_nonl": f1 = _synth_fastq(4, 2000, 20, 150, 0.1, final_newline=False)
    elif case == "syn_big": f1 = _synth_fastq(5, 200_000, 100, 100, 0.05)
    else: f1 = b""
    want = [po.preprocess_fastq(f1)] + ([po.preprocess_fastq(f2)] if f2 is not None else [])
    with spring_amd.ReorderStage() as s:
        info = s.load_fastq(f1, f2)
        for j, w in enumerate(want):
            assert info["num_reads"][j] == w["num_reads"] and info["num_reads_clean"][j] == w["num_clean"]
            assert info["num_reads_N"][j] == w["num_N"]
            nd, on = s.fastq_N(j)
            assert nd == w["ndna"] and np.array_equal(on, w["order_N"])
        assert info["max_readlen"] == max(w["max_readlen"] for w in want)
        assert s.download_dna() == b"".join(w["clean"] for w in want)
        # and the stage runs on it exactly as on the .dna stream the reference would have written
        n, L = s.n, s.max_readlen
        if n:
            got = s.run().streams()
            read, ln = po.load_dna(b"".join(w["clean"] for w in want), n, L)
            ref = po.reorder_rounds(read, ln, L, max(1, min(65536, n >> 10)), 1)
            for k in ("order", "rc", "flag", "pos", "rlen", "order_s"):
                assert np.array_equal(got[k], ref[k]), (case, k)


@pytest.mark.gpu
@pytest.mark.parametrize("members", [1, 3])
def test_fastq_frontend_gzip_input(members):
    """.gz input (preprocess.cpp:154-183): a gzip image -- one member, or several concatenated members cut in the
    middle of a record -- gives exactly what its text gives; paired input with one file compressed and one not."""
    import gzip
    import spring_amd
    f1, f2 = _synth_fastq(11, 5000, 30, 150, 0.1), _synth_fastq(12, 5000, 30, 150, 0.1)

    def gz(t):
        cuts = [len(t) * i // members for i in range(members + 1)]  # arbitrary byte positions
        return b"".join(gzip.compress(t[a:b], 6) for a, b in zip(cuts[:-1], cuts[1:]))

    with spring_amd.ReorderStage() as s:
        want = s.load_fastq(f1, f2)
        want_dna, want_n = s.download_dna(), [s.fastq_N(0), s.fastq_N(1)]
    with spring_amd.ReorderStage() as s:
        got = s.load_fastq(gz(f1), f2)
        for k in ("num_reads", "num_reads_clean", "num_reads_N", "max_readlen"):
            assert got[k] == want[k], k
        assert s.download_dna() == want_dna
        for j in range(2):
            nd, on = s.fastq_N(j)
            assert nd == want_n[j][0] and np.array_equal(on, want_n[j][1])
    with spring_amd.ReorderStage() as s:  # zero padding after the last member (tape / block padding): ignored, as gzip(1) does
        s.load_fastq(gz(f1) + b"\0" * 700, f2)
        assert s.download_dna() == want_dna
    with spring_amd.ReorderStage() as s:  # anything else after the last member is an error, not silently dropped
        with pytest.raises(spring_amd.ReorderError, match="gzip error"):
            s.load_fastq(gz(f1) + b"\0" * 8 + b"garbage", f2)
    with spring_amd.ReorderStage() as s:
        with pytest.raises(spring_amd.ReorderError, match="gzip error"):
            s.load_fastq(gz(f1)[:-20])  # truncated
    with spring_amd.ReorderStage() as s:
        bad = bytearray(gz(f1)); bad[len(bad) // 2] ^= 0xFF
        with pytest.raises(spring_amd.ReorderError, match="gzip error|multiple of 4|Invalid character|Too long"):
            s.load_fastq(bytes(bad))


@pytest.mark.gpu
def test_fastq_frontend_errors():
    import spring_amd
    with spring_amd.ReorderStage() as s:
        with pytest.raises(spring_amd.ReorderError, match="multiple of 4"):
            s.load_fastq(b"@a\nACGT\n+\n")
    with spring_amd.ReorderStage() as s:
        with pytest.raises(spring_amd.ReorderError, match="Too long read length"):
            s.load_fastq(b"@a\n" + b"A" * 600 + b"\n+\n" + b"I" * 600 + b"\n")
    with spring_amd.ReorderStage() as s:
        with pytest.raises(spring_amd.ReorderError, match="paired files do not match"):
            s.load_fastq(_synth_fastq(6, 10, 50, 50), _synth_fastq(7, 11, 50, 50))
    # characters outside A C G T N (lower case, IUPAC codes) are undefined in the reference's tables: an error here
    for bad in (b"ACGTacgt", b"ACGRT", b"AC.GT"):
        with spring_amd.ReorderStage() as s:
            with pytest.raises(spring_amd.ReorderError, match="Invalid character"):
                s.load_fastq(b"@a\nACGT\n+\nIIII\n@b\n" + bad + b"\n+\n" + b"I" * len(bad) + b"\n")


# ------------------------------------------------------------------ row f4: reorder-only output

def _py_reorder(text, order):
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines = lines[:-1]
    recs = [b"\n".join(lines[4 * i:4 * i + 4]) + b"\n" for i in range(len(lines) // 4)]
    return b"".join(recs[int(k)] for k in order)


@pytest.mark.gpu
@pytest.mark.parametrize("n,crlf,final_nl", [(1, False, True), (7, False, False), (500, True, True), (20000, False, True)])
def test_fastq_reorder_matches_python(n, crlf, final_nl):
    from spring_amd import order_ops as oo
    t = _synth_fastq(n + 5, n, 1, 200, crlf=crlf, final_newline=final_nl)
    rng = np.random.default_rng(n)
    for order in (rng.permutation(n), rng.integers(0, n, max(n // 3, 1)), np.arange(n)[::-1]):
        got, _ = oo.fastq_reorder(t, order.astype(np.uint32))
        assert got == _py_reorder(t, order)
    with pytest.raises(Exception):
        oo.fastq_reorder(t, np.array([n], np.uint32))


@pytest.mark.gpu
def test_reorder_only_pipeline_external_validity():
    """FASTQ -> front end -> reorder -> encoder -> read_order.bin -> reordered FASTQ: same multiset of records,
    and record k of the output is the read the encoder streams decode at position k."""
    import spring_amd
    from helpers import decode_reads
    from spring_amd import order_ops as oo
    from spring_amd.encoder import EncoderStage
    rng = np.random.default_rng(5)
    G, L, n = 20000, 100, 6000
    genome = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, G)]
    recs = []
    for i in range(n):
        p = int(rng.integers(0, G - L))
        s = genome[p:p + L].copy()
        if rng.random() < 0.05:
            s[int(rng.integers(0, L))] = ord("N")
        recs.append(b"@id%d/x\n%s\n+\n%s\n" % (i, s.tobytes(), bytes(33 + (i + np.arange(L)) % 40)))
    text = b"".join(recs)
    with spring_amd.ReorderStage(spring_amd.ReorderOpts(num_chains=16, num_thr=2)) as st:
        st.load_fastq(text)
        st.run()
        dnaN, order_N = st.fastq_N(0)
        with EncoderStage() as enc:
            enc.encode(st, dnaN, order_N)
            e = enc.streams()
    out, _ = oo.fastq_reorder(text, e["order"])
    got = [b"\n".join(x) + b"\n" for x in zip(*[iter(out.split(b"\n")[:-1])] * 4)]
    assert sorted(got) == sorted(recs) and len(got) == n
    assert got == [recs[int(k)] for k in e["order"]]
    dec = decode_reads(e)   # aligned reads by original position
    for k in range(len(e["pos"])):
        assert got[k].split(b"\n")[1].decode() == dec[int(e["order"][k])]
