"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol
include/spring_reorder.h declares, the synthetic generator is deterministic, the Python
mirror rejects bad arguments like the reference does.  No compute calls (no GPU here)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from spring_amd import _lib
    hdr = (open(os.path.join(ROOT, "include", "spring_reorder.h")).read()
           + open(os.path.join(ROOT, "include", "spring_encoder.h")).read())
    declared = set(re.findall(r"\b(spring_(?:reorder|synth|order|encoder|fastq|mg)_\w+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libspring_reorder_hip.so does not export " + name
    assert declared == set(_lib.EXPORTS)


def test_no_cpu_fallback_in_product():
    """The product must not reach the oracle (or any CPU restatement)."""
    for dp, _, fs in os.walk(os.path.join(ROOT, "spring_amd")):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in src.lower(), (f, "mentions oracle")


def test_synth_host_deterministic_and_plausible():
    import spring_amd
    a = spring_amd.synth_dna_host(2000, 100, 8000, 3, 10000)
    b = spring_amd.synth_dna_host(2000, 100, 8000, 3, 10000)
    c = spring_amd.synth_dna_host(2000, 100, 8000, 4, 10000)
    assert a == b and a != c and len(a) == 2000 * 27
    from oracle import pyoracle as po
    read, ln = po.load_dna(a, 2000, 100)
    assert np.all(ln == 100)
    r = po.reorder_serial(read, ln, 100)  # 25x coverage, 1 % errors: most reads must cluster
    assert len(r["order"]) > 1500


def test_synth_paired_pool_is_mate_pairs():
    """SPRING_SYNTH_PAIRED (BASELINE config 4): read j and read npairs + j are the two ends of one fragment of
    ~N(400, 50) bases, on opposite strands -- checked against the generator's own genome, error-free."""
    import spring_amd
    from helpers import decode_dna_fixed
    npairs, L, G, seed = 3000, 100, 60000, 9
    dna = spring_amd.synth_dna_host(2 * npairs, L, G, seed, 0 | spring_amd.SYNTH_PAIRED)
    genome = spring_amd.synth_genome_host(G, seed)
    reads = decode_dna_fixed(dna, 2 * npairs, L)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    frag = []
    for j in range(npairs):
        a, b = reads[j], reads[npairs + j]
        pa, pb = genome.find(a), genome.find(b.translate(comp)[::-1])
        if pa < 0:  # the fragment lies on the reverse strand: the first read is the reverse complement of its END
            pa, pb = genome.find(a.translate(comp)[::-1]), genome.find(b)
            assert pa >= 0 and pb >= 0 and pb <= pa, j
            frag.append(pa + L - pb)
        else:
            assert pb >= pa, j
            frag.append(pb + L - pa)
    frag = np.array(frag)
    assert frag.min() >= L and 390 < frag.mean() < 410 and 40 < frag.std() < 60
    strands = sum(genome.find(reads[j]) >= 0 for j in range(npairs))
    assert 0.4 * npairs < strands < 0.6 * npairs
    with pytest.raises(spring_amd.ReorderError, match="even number"):
        spring_amd.synth_dna_host(7, L, G, seed, spring_amd.SYNTH_PAIRED)


def test_wrong_bitset_size_raises_like_reference():
    import spring_amd
    from spring_amd.reorder import CompressionParams
    with pytest.raises(spring_amd.ReorderError, match="Wrong bitset size"):
        spring_amd.call_reorder("/tmp", CompressionParams(600, [0, 0]))


def test_bench_line_builders_keep_the_contract():
    """bench.py builds its JSON line with headline_line() / roofline_block(): every field of the bench contract is
    there and the derived figures follow from the inputs."""
    import importlib.util
    import types
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = types.SimpleNamespace(reads=100_000_000, readlen=150, steps=3, warmup=1, coverage=25, err_ppm=10000, chains=0, num_thr=8)
    st = dict(ms_unpack=5.0, ms_dict=26.0, ms_chains=400.0, ms_finalize=4.0, unmatched=13958771, n_single=12719866,
              rounds=1968, chains=65536)
    line = bench.headline_line(a, 1, 1.35, st, 600_000_000)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"]
    assert abs(line["value"] - 3 * 100.0 / 1.35) < 0.01 and abs(line["ms_per_step"] - 450.0) < 1e-6
    assert bench.headline_line(a, 4, 1.35, st, 600_000_000)["value"] == pytest.approx(4 * line["value"], rel=1e-4)
    r = bench.roofline_block(164.25e9, 428.0, 1968, 427e6)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["achieved"] - 164.25e9 / 0.428 / 1e9) < 0.01
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) / r["achieved"] < 0.01
    assert 0 < r["frac"] < 1 and r["traffic"] > r["algorithmic_bytes_per_launch"]


def test_ctypes_mirrors_match_the_c_header(tmp_path):
    """spring_amd/_lib.py mirrors spring_reorder_opts / spring_reorder_stats by hand: sizes and the offsets of the
    fields added last must equal what a C compiler makes of include/spring_reorder.h."""
    import ctypes as C
    import subprocess
    from spring_amd import _lib
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "spring_reorder.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(spring_reorder_opts), '
                   'offsetof(spring_reorder_opts, long_budget), offsetof(spring_reorder_opts, alternatives), '
                   'sizeof(spring_reorder_stats), offsetof(spring_reorder_stats, alternatives), '
                   'offsetof(spring_reorder_stats, deep_pool)); return 0; }\n')
    exe = tmp_path / "abi"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [C.sizeof(_lib.Opts), _lib.Opts.long_budget.offset, _lib.Opts.alternatives.offset,
            C.sizeof(_lib.Stats), _lib.Stats.alternatives.offset, _lib.Stats.deep_pool.offset]
    assert got == want, (got, want)
