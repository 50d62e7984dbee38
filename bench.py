#!/usr/bin/env python3
"""bench.py -- Mreads/s through the reorder stage on synthetic reads (BASELINE.json metric).

One "step" = one pass of the whole stage (unpack -> dictionaries -> chains -> streams) over one
batch of synthetic reads that is already resident in HBM as a .dna record stream when the timed
region starts.  N=1 workload = BASELINE configs[2]: 100 M x 150 bp single-end.  For every N>1 the job is the SAME
shared read pool -- BASELINE configs[4], 400 M x 150 bp, which also fits one GPU (108 GB) -- with the chains sharded
over the GPUs and one RCCL all-gather of the proposal words per round, issued by the library on its own stream
(spring_reorder_mg_run): total work is fixed, `"scaling": "strong"`, `value` = pool reads / wall-clock, and rank 0
also runs that pool alone so the line carries the single-GPU rate it is to be compared with; see DESIGN.md
"Multi-GPU".  --lanes switches the N>1 run to independent lanes (one read set per GPU, no data-path collective).

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` for the
round kernel, `cpu_baseline` (the C oracle timed on a bounded sample of the same workload) and `compression_cost`
(what the default chain count costs in compressed size, real BSC).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
RANDOM_REQ_PEAK_G = 49.0  # measured: independent random <=32-byte reads, tables >= 4 GiB (tools/random_gather_bench.hip)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per GPU")
    ap.add_argument("--readlen", type=int, default=150)
    ap.add_argument("--coverage", type=int, default=25)
    ap.add_argument("--err-ppm", type=int, default=10000)
    ap.add_argument("--chains", type=int, default=0, help="0 = library default")
    ap.add_argument("--num-thr", type=int, default=8, help="per-tid output sets (reference default -t 8)")
    ap.add_argument("--cpu-sample", type=int, default=100_000_000,
                    help="reads in the CPU-baseline sample (0 = skip); the default is the whole workload (~105 s at 32 threads)")
    ap.add_argument("--cpu-files-sample", type=int, default=16_000_000, help="reads in the CPU twin of stage_incl_files")
    ap.add_argument("--cpu-threads", type=int, default=0, help="OpenMP threads of the CPU baseline (0 = min(32, cpus))")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--lanes", action="store_true",
                    help="N>1: independent lanes (every rank its own read set) instead of one shared pool")
    ap.add_argument("--pool-reads", type=int, default=400_000_000,
                    help="N>1: reads in the shared pool (BASELINE configs[4]: 400 M), the same for every N")
    ap.add_argument("--pool-chains", type=int, default=524_288, help="N>1: chains of the pool, the same for every N")
    ap.add_argument("--no-single", action="store_true", help="N>1: skip rank 0's single-GPU pass over the same pool")
    ap.add_argument("--cost-sample", type=int, default=4_000_000,
                    help="reads in the compression_cost leg (same coverage and error rate; 0 = skip)")
    ap.add_argument("--force-pool", action="store_true",
                    help="run the shared-pool path even at N=1 (a 1-rank RCCL communicator): exercises the in-library "
                         "ncclAllGather on a single-GPU box")
    ap.add_argument("--sweep-sample", type=int, default=20_000_000,
                    help="reads per pool of the coverage_sweep leg (deep-coverage / contended pools, library defaults; 0 = skip)")
    ap.add_argument("--choice-sample", type=int, default=20_000_000,
                    help="reads in the pools of compression_cost.output_changing_choices (0 = skip)")
    ap.add_argument("--files-sample", type=int, default=100_000_000,
                    help="reads in the file-contract leg (stage_incl_files): the whole workload by default; 0 = skip")
    return ap.parse_args()


def algorithmic_bytes_search(st, W):
    """bytes the search needs, from counted work (reference-equivalent counters):
    P*(8+8) [table slot + bin offsets] + Kv*(4+B) [first id + first read] + C*(4+B+2+1) [id, read, len, flag]."""
    B = 8 * W
    return st["probes"] * 16 + st["keyok"] * (4 + B) + st["cands"] * (7 + B)


def algorithmic_bytes_apply(st):
    """SURVEY.md 8(d), the two remaining chain-phase terms: R*(8+4) [two bin removals per claimed read] +
    E*16 [order, RC, flag, pos, len of every emitted read]."""
    n = st["n_reads"]
    return 2 * n * 12 + n * 16


def compression_cost(spring_amd, a, dev, reads_per_chain, phases=1):
    """bits per base of the reorder + encoder output after BSC (the reference's, oracle/_ref/ref_bsc) for K = default
    and K = num_thr on a sample with the workload's coverage and error rate (encoder.cpp:111-156 is where SPRING
    hands these streams to BSC); one chain group against two on the headline's pool; and `output_changing_choices`: the
    four cells {one group, two groups} x {one candidate per proposal, two} on the pools where the library makes those
    choices from the dictionary (6 400x, PhiX-like, genome-like), at the library's chain count.  The BSC processes of
    all cells run side by side on the host's cores while the GPU runs the next cell."""
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from oracle import pyoracle as po
    from spring_amd.encoder import EncoderStage
    bsc_bin = po.ref_bsc_bin()
    if not bsc_bin:
        return {"error": "oracle/_ref/ref_bsc is not built (make -C oracle ref, needs the reference sources)"}
    n, L = a.cost_sample, a.readlen
    pool = ThreadPoolExecutor(max_workers=max(4, min(48, (os.cpu_count() or 8) // 4)))

    def bsc(b):
        if not len(b):
            return 0
        with tempfile.TemporaryDirectory() as d:
            fi, fo = os.path.join(d, "in"), os.path.join(d, "out")
            open(fi, "wb").write(b)
            subprocess.run([bsc_bin, fi, fo], check=True, stdout=subprocess.DEVNULL)
            return os.path.getsize(fo)

    def sized(n, K, G=None, flags=0, **kw):
        """-> a cell whose "bytes" is still a list of futures (finish() turns it into the sum)"""
        G = G or max(n * L // a.coverage, 2 * L)
        with spring_amd.ReorderStage(spring_amd.ReorderOpts(device=dev, num_chains=K, num_thr=1, **kw)) as st:
            st.load_synth(n, L, G, 5, a.err_ppm | flags)
            st.run()
            sst = st.stats()
            with EncoderStage(dev) as enc:
                info = enc.encode(st)
                e = enc.streams()
                packed, _ = enc.seq_packed()
        dpos = np.diff(e["pos"].astype(np.int64), prepend=0).astype(np.int32)
        futs = [pool.submit(bsc, b) for b in (packed, dpos.tobytes(), bytes(e["noise"]), e["noisepos"].tobytes(), e["rc"].tobytes(), bytes(e["unaligned"]))]
        return {"chains": int(sst["chains"]), "chain_groups": int(sst.get("phases", 1)), "candidates_per_proposal": int(sst.get("alternatives", 1)),
                "contigs": int(info["num_contigs"]), "bytes": futs, "reads": n, "chains_stage_ms": round(sst["ms_chains"], 1)}

    def finish(c):
        if isinstance(c, dict) and isinstance(c.get("bytes"), list):
            tot = sum(f.result() for f in c["bytes"])
            c["bytes"] = int(tot)
            c["bits_per_base"] = round(tot * 8.0 / (c.pop("reads") * L), 4)
        return c

    # the sample runs at the headline run's reads per chain (the default is capped at 65 536 chains: 1 526 reads per
    # chain at 100 M reads; a 4 M-read sample left to the default rule would run at 1 024 and overstate the cost)
    k_same = max(1, int(round(n / max(reads_per_chain, 1.0))))
    d, r = sized(n, k_same, phases=1), sized(n, a.num_thr)
    g1 = g2 = None
    if phases == 2:
        # the headline run's schedule (two chain groups) needs 4 096 chains: a larger sample at the same reads per chain,
        # one group against two (the K = num_thr run above would take a minute on it)
        n2 = max(n, int(4200 * reads_per_chain))
        k2 = max(4096, int(round(n2 / max(reads_per_chain, 1.0))))
        g1, g2 = sized(n2, k2, phases=1), sized(n2, k2, phases=2)
    # where the library itself chooses the schedule (two groups at the cap of 131 072 chains on deep pools) and the number
    # of candidates per proposal (two on contended pools): all four cells on each of those pools, library's chain count
    cells = []
    if a.choice_sample > 0:
        ns = a.choice_sample
        for name, pn, flags, pG in (("6400x", ns, 0, max(ns * L // 6400, 4 * L)), ("PhiX-like", ns // 2, 0, 5400),
                                    ("genome-like 25x", ns, 0x20000000, max(ns * L // 25, 4 * L))):
            row = {"pool": name, "reads": pn}
            try:
                row["library_choice"] = sized(pn, 0, G=pG, flags=flags)
                kk = row["library_choice"]["chains"]
                for ph in (1, 2):
                    for al in (1, 2):
                        try:
                            row["groups_%d_candidates_%d" % (ph, al)] = sized(pn, kk, G=pG, flags=flags, phases=ph, alternatives=al)
                        except Exception as e:  # noqa: BLE001  (a combination the library refuses on this pool)
                            row["groups_%d_candidates_%d" % (ph, al)] = {"refused": str(e)[:160]}
            except Exception as e:  # noqa: BLE001
                row["error"] = repr(e)
            cells.append(row)
    d, r = finish(d), finish(r)
    out = {"sample_reads": n, "read_len": L, "coverage": a.coverage, "default_chains": d, "reference_granularity": r,
           "size_ratio_default_vs_num_thr": round(d["bytes"] / r["bytes"], 4),
           "what": "read streams (consensus, positions, noise, noise positions, orientation, unaligned) after the reference's "
                   "BSC; reads per chain %d (as in the headline run) vs %d (K = num_thr = %d)" % (n // max(d["chains"], 1), n // max(r["chains"], 1), a.num_thr)}
    if g1 is not None:
        g1, g2 = finish(g1), finish(g2)
        out["two_chain_groups"] = {"sample_reads": n2, "one_group": g1, "two_groups": g2,
                                   "size_ratio_two_groups_vs_one": round(g2["bytes"] / g1["bytes"], 4)}
    for row in cells:
        for k in list(row):
            finish(row[k])
        base = row.get("groups_1_candidates_1")
        if isinstance(base, dict) and base.get("bytes"):
            row["size_ratio_vs_one_group_one_candidate"] = {k[7:]: round(v["bytes"] / base["bytes"], 4) for k, v in row.items()
                                                            if k.startswith("groups_") and isinstance(v, dict) and v.get("bytes")}
            lc = row.get("library_choice")
            if isinstance(lc, dict) and lc.get("bytes"):
                row["size_ratio_library_choice"] = round(lc["bytes"] / base["bytes"], 4)
    if cells:
        out["output_changing_choices"] = {"pools": cells, "what": "the same read streams after the reference's BSC for {one chain group, two} x {one candidate "
                                          "per proposal, two} at the chain count the library picks for the pool; library_choice = what "
                                          "opts.phases = 0, opts.alternatives = 0 run (chain_groups / candidates_per_proposal say which cell that is)"}
    pool.shutdown()
    return out


def headline_line(a, world, el, st, G):
    """The contract's JSON line for the lanes / single-GPU run: `el` = max-over-ranks seconds of the a.steps timed steps,
    `st` = the statistics of the last step, every rank processed a.reads reads per step."""
    n, L = a.reads, a.readlen
    return {
        "metric": "Mreads/s through reorder stage", "value": round(n * a.steps * world / el / 1e6, 3), "unit": "Mreads/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "%d x %d bp single-end synthetic reads per GPU (uniform genome %d bp, %dx coverage, "
                        "%.1f%% substitutions, 50%% reverse-complemented), inputs resident in HBM as .dna records"
                        % (n, L, G, a.coverage, a.err_ppm / 1e4),
            "reads_per_gpu": n, "read_len": L, "chains": int(st.get("chains", 0)) or (a.chains or "auto"), "num_thr": a.num_thr,
            "parallelism": "1 process per GPU, independent lanes" if world > 1 else "single GPU",
            "stage_ms": {k: round(st[k], 2) for k in ("ms_unpack", "ms_dict", "ms_chains", "ms_finalize")},
            "unmatched": st["unmatched"], "singletons": st["n_single"], "rounds": st["rounds"],
            "chain_groups": int(st.get("phases", 1)),  # 2: the chains run as two groups whose rounds alternate (opts.phases)
        },
    }


def roofline_block(alg_bytes, kernel_ms, launches, traffic, busy_ms=None):
    """`roofline` of the bench line from the algorithmic bytes of all launches, their summed duration (HIP events) and
    the PMC traffic per launch (or None).  busy_ms: the time during which at least one launch was running (the union of
    their intervals, from the same events).  With the two-group schedule (stats.phases = 2) two launches of the kernel run
    side by side nearly all the time, each of them slowed by the other: `achieved` = bytes of all launches / busy time =
    bytes per launch / average launch duration x the average number of launches running at a time (`concurrent_launches`).
    With one launch at a time (busy_ms = kernel_ms) that is bytes per launch / average launch duration."""
    launches = max(int(launches), 1)
    busy = busy_ms if busy_ms and busy_ms > 0 else kernel_ms
    ach = alg_bytes / (busy * 1e-3) / 1e9 if busy > 0 else 0.0
    return {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic,
            "launches": launches, "avg_launch_us": round(kernel_ms * 1e3 / launches, 2),
            "algorithmic_bytes_per_launch": round(alg_bytes / launches, 1),
            "concurrent_launches": round(kernel_ms / busy, 3) if busy > 0 else None, "busy_ms": round(busy, 2),
            "achieved_per_launch_alone": round(alg_bytes / (kernel_ms * 1e-3) / 1e9, 2) if kernel_ms > 0 else None,
            "achieved_basis": ("algorithmic bytes of all launches / busy_ms = algorithmic_bytes_per_launch / avg_launch_us x "
                               "concurrent_launches: launches of this kernel overlap (two chain groups on two streams), busy_ms is "
                               "the union of the launches' intervals from the same HIP events"
                               if busy > 0 and kernel_ms / busy > 1.01 else "algorithmic_bytes_per_launch / avg_launch_us")}


def physical_cores():
    """distinct (socket, core) pairs of /proc/cpuinfo (None when the file does not say)"""
    try:
        seen, phys = set(), None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                seen.add((phys, ln.split(":")[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def kernels_sha():
    """Identity of the kernels a PMC summary belongs to (profiles/pmc_latest.json carries the same stamp)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("reorder_kernels.hip", "reorder_device.h", "reorder_round_mc.h"):
        h.update(open(os.path.join(ROOT, "spring_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pool_main(a, lanes, torch, spring_amd, L_):
    """N>1: one shared read pool, chains sharded over the GPUs, the exchange inside the library (RCCL)."""
    # RCCL prints a version banner with printf when its first communicator comes up, and stdout must carry exactly
    # one line: fd 1 points at stderr until the JSON line is due (the C stdio buffer is flushed before it comes back)
    sys.stdout.flush()
    saved_fd1 = os.dup(1)
    os.dup2(2, 1)
    try:
        out, rank = _pool_run(a, lanes, torch, spring_amd, L_)
    finally:
        C.CDLL(None).fflush(None)
        os.dup2(saved_fd1, 1)
        os.close(saved_fd1)
    if rank == 0 and out is not None:
        print(json.dumps(out), flush=True)


def _pool_run(a, lanes, torch, spring_amd, L_):
    from spring_amd.pool import DistPool, PoolComm
    world, rank = lanes.world, lanes.rank
    dev = torch.cuda.current_device()
    L = a.readlen
    n = a.pool_reads
    G = max(n * L // a.coverage, 2 * L)
    Ktot = (a.pool_chains + world - 1) // world * world
    nb = L_.spring_synth_dna_bytes(n, L)
    buf = torch.empty(nb, dtype=torch.uint8, device="cuda")     # every rank holds the whole pool (replicated)
    assert L_.spring_synth_dna_device(C.c_void_p(buf.data_ptr()), n, L, G, 13, a.err_ppm) == 0, L_.spring_reorder_last_error()
    torch.cuda.synchronize()
    dist = lanes.dist
    if dist is None:  # --force-pool at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", dev))
    # the exchange: ncclAllGather issued by the library on its own stream.  Should the RCCL communicator not come
    # up on some rank, every rank falls back to the host-staged all-gather over a gloo group (slower per round,
    # same result) and the line says so in config.exchange.
    comm, ok = None, 1
    try:
        comm = PoolComm(dist, torch.device("cuda", dev), transport="rccl")
    except Exception as e:  # noqa: BLE001
        ok = 0
        print("# rank %d: RCCL communicator failed (%s)" % (rank, e), file=sys.stderr, flush=True)
    flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    exchange = "rccl"
    if int(flag.item()) == 0:
        from spring_amd.pool import GroupView
        if comm is not None:
            comm.close()
        gl = dist.new_group(backend="gloo")
        comm = PoolComm(GroupView(dist, gl, "gloo"), torch.device("cuda", dev), transport="host")
        exchange = "host-staged all-gather over gloo (RCCL communicator unavailable)"

    def one_pass(**kw):
        dp = DistPool(comm, Ktot, num_thr=a.num_thr, **kw)
        st = dp.run(lambda s: s.load_dna_device(buf.data_ptr(), nb, n, L, True))
        st["local_matched"], st["local_single"] = st["n_matched"], st["n_single"]
        dp.close()
        return st

    el, st = lanes.timed_steps(one_pass, a.steps, a.warmup)
    tot_matched = lanes.sum_over_ranks(st["local_matched"])
    tot_single = lanes.sum_over_ranks(st["local_single"])
    # one more pass with HIP events around the three pieces of a round (outside the timed region): where a round goes
    stt = one_pass(time_search=True)
    rounds = max(stt["search_launches"], 1)
    groups = max(int(stt.get("phases", 1)), 1)
    budget = {"rounds": stt["rounds"], "chain_groups": groups,
              "round_kernel_us": round(stt["ms_search_kernel"] * 1e3 / rounds, 1),
              "exchange_us": round(stt["ms_exchange"] * 1e3 / rounds, 1),
              "resolve_and_mark_us": round(stt["ms_resolve_mark"] * 1e3 / rounds, 1),
              "replicated_ms": {"unpack": round(stt["ms_unpack"], 1), "dictionaries": round(stt["ms_dict"], 1),
                                "finalize": round(stt["ms_finalize"], 1)},
              "chains_ms": round(stt["ms_chains"], 1),
              "what": "rank 0, per round: k_round over the rank's own chains | all-gather of the proposal words | "
                      "k_mg_resolve + k_mg_mark over ALL chains (replicated work, grows with N); replicated_ms is work every "
                      "rank repeats for the whole pool"}
    if groups == 2:
        # two chain groups: a rank owns a slice of each; a group's all-gather (on the exchange stream), resolve and mark run
        # beside the OTHER group's round kernel.  round_kernels_busy_ms = the time during which at least one round kernel of
        # the rank ran (union of the launches' intervals); chains_ms beyond it is what the exchange + resolve + mark still
        # cost the critical path.  resolve_and_mark_us includes the mark step's wait for the other group's mark step.
        busy = stt.get("ms_search_busy") or 0.0
        budget.update({
            "per": "group-round (a round of one chain group: half of the rank's chains)",
            "round_kernels_busy_ms": round(busy, 1),
            "off_the_round_kernels_ms": round(max(stt["ms_chains"] - busy, 0.0), 1),
            "off_the_round_kernels_frac": round(max(stt["ms_chains"] - busy, 0.0) / max(stt["ms_chains"], 1e-9), 4),
            "what": "rank 0, per group-round: k_round over the rank's slice of the group | in-place all-gather of the group's proposal "
                    "words on the exchange stream | k_mg_resolve + the group's mark step over the whole group (replicated), which "
                    "also waits for the other group's mark step; these run beside the other group's round kernel: chains_ms - "
                    "round_kernels_busy_ms is what they leave on the critical path (timed pass: a batch of rounds at a time)"})
    # the single-GPU rate on the SAME pool and chain count (rank 0 alone, the others wait): what `value` is to be divided by
    single = None
    if world > 1 and not a.no_single:
        lanes.barrier()
        if rank == 0:
            try:
                t0 = time.perf_counter()
                s1 = spring_amd.ReorderStage(spring_amd.ReorderOpts(device=dev, num_chains=Ktot, num_thr=a.num_thr))
                s1.load_dna_device(buf.data_ptr(), nb, n, L, True)
                s1.run()
                s1s = s1.stats()
                s1.close()
                t1 = time.perf_counter() - t0
                single = {"value": round(n / t1 / 1e6, 3), "unit": "Mreads/s", "seconds": round(t1, 3), "rounds": s1s["rounds"],
                          "chain_groups": int(s1s.get("phases", 1)),
                          "what": "the same pool and chain count on rank 0's GPU alone (one pass incl. allocations), with the "
                                  "library's choice of the schedule (two chain groups whose rounds alternate, in the pool as well: "
                                  "the same output)"}
            except Exception as e:  # noqa: BLE001
                single = {"error": repr(e)}
        lanes.barrier()
    out = {
        "metric": "Mreads/s through reorder stage", "value": round(n * a.steps / el / 1e6, 3), "unit": "Mreads/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(el / a.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {
            "workload": "ONE shared pool of %d x %d bp single-end synthetic reads, the same for every N (uniform genome "
                        "%d bp, %dx coverage, %.1f%% substitutions, 50%% reverse-complemented), resident in HBM on every GPU "
                        "as .dna records; %d chains sharded over %d GPUs (%d chain group%s: a rank owns a slice of each), RCCL all-gathers of "
                        "%d proposal bytes per round issued by the library on its exchange stream"
                        % (n, L, G, a.coverage, a.err_ppm / 1e4, Ktot, world, int(st.get("phases", 1)), "s" if int(st.get("phases", 1)) > 1 else "", Ktot * 8),
            "exchange": exchange, "pool_reads": n, "read_len": L, "chains": Ktot, "num_thr": a.num_thr,
            "parallelism": "1 process per GPU, single shared pool: reads + dictionaries replicated, chains sharded, "
                           "all-gather per round (DESIGN.md section 7)",
            "stage_ms_rank0": {k: round(st[k], 2) for k in ("ms_unpack", "ms_dict", "ms_chains", "ms_finalize")},
            "rounds": st["rounds"], "reads_emitted_all_ranks": int(tot_matched + tot_single),
            "note": "strong scaling: pool and chain count are the same for every N > 1; at N=1 the bench runs BASELINE "
                    "configs[2] (100 M reads) instead, so compare with single_gpu_same_pool, not with the N=1 line",
        },
        "round_budget": budget,
        "single_gpu_same_pool": single,
    }
    if rank == 0:
        assert int(tot_matched + tot_single) == n, "the ranks' streams do not add up to the pool"
    comm.close()
    if lanes.dist is None:
        dist.destroy_process_group()
    lanes.close()
    return out, rank


def main():
    a = parse()
    import torch
    import spring_amd
    from spring_amd import _lib
    L_ = _lib.lib()

    from spring_amd.lanes import Lanes
    lanes = Lanes()  # one process per GPU; nccl (= RCCL) when WORLD_SIZE > 1
    world, rank = lanes.world, lanes.rank
    if world == 1:
        torch.cuda.set_device(0)
    dev = torch.cuda.current_device()
    if (world > 1 and not a.lanes) or a.force_pool:
        return pool_main(a, lanes, torch, spring_amd, L_)

    n, L = a.reads, a.readlen
    G = max(n * L // a.coverage, 2 * L)
    seed = lanes.lane_seed(11)  # every rank (lane) has its own genome and reads
    nb = L_.spring_synth_dna_bytes(n, L)
    buf = torch.empty(nb, dtype=torch.uint8, device="cuda")
    rc = L_.spring_synth_dna_device(C.c_void_p(buf.data_ptr()), n, L, G, seed, a.err_ppm)
    assert rc == 0, L_.spring_reorder_last_error()
    torch.cuda.synchronize()

    def one_pass(**kw):
        s = spring_amd.ReorderStage(spring_amd.ReorderOpts(device=dev, num_chains=a.chains, num_thr=a.num_thr, **kw))
        s.load_dna_device(buf.data_ptr(), nb, n, L, True)
        s.run()
        st = s.stats()
        s.close()
        return st

    el, st = lanes.timed_steps(one_pass, a.steps, a.warmup)
    out = headline_line(a, world, el, st, G)

    if rank == 0 and world == 1 and not a.no_roofline:
        # three extra passes: (1) the production round kernel (k_round = apply of the last proposals + search, 85 % of
        # the GPU time) with a HIP-event pair around every launch, on the library's stream; (2) the counting variant
        # for the reference-equivalent work counters (slower, so it is not the one that is timed); (3) the two-kernel
        # round (opts.fused = -1), where the Hamming search is a kernel of its own, for the search-only figure
        st_t = one_pass(time_search=True)
        sr = one_pass(collect_stats=True)
        st_2k = one_pass(time_search=True, fused=-1)
        W = (2 * L - 1) // 64 + 1
        alg_search = algorithmic_bytes_search(sr, W)
        alg = alg_search + algorithmic_bytes_apply(sr)
        ms = st_t["ms_search_kernel"]
        busy = st_t.get("ms_search_busy") or ms
        launches = max(st_t["search_launches"], 1)
        phases = max(int(st_t.get("phases", 1)), 1)
        # HBM bytes per launch from the PMC counters: rocprofv3 cannot run inside this process, so the
        # figure comes from the committed summary of separate --pmc passes over this same command
        # (profiles/README.md; FETCH_SIZE*1024 + WRITE_SIZE*1024, calibrated on tools/random_gather_bench:
        # one 64-byte request per random access in this access pattern).  null if the workload differs.
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            # only a summary taken from THESE kernels counts (the stamp is the hash of the kernel sources)
            if pmc.get("reads") == n and pmc.get("read_len") == L and pmc.get("kernels_sha") == kernels_sha():
                ks = pmc["kernels"].get("sr::k_round_mc") or pmc["kernels"]["sr::k_round"]
                traffic = round(ks["fetch_bytes_per_launch"] + ks["write_bytes_per_launch"], 1)
        except Exception:
            traffic = None
        # the same PMC figure read as a request rate: this kernel's loads are independent random 64-byte
        # requests, for which the measured ceiling on this GPU is ~49 G requests/s
        # (tools/random_gather_bench.hip, profiles/r01_pmc_calibration_random_gather.csv), not 8 TB/s / 64 B
        req = None
        try:
            fetch = ks["fetch_bytes_per_launch"] if traffic is not None else None
            if fetch:
                rate = fetch * launches / 64.0 / (busy * 1e-3) / 1e9  # (all launches' requests over the time any of them ran)
                req = {"achieved": round(rate, 2), "peak": RANDOM_REQ_PEAK_G, "unit": "G random 64-byte requests/s",
                       "frac": round(rate / RANDOM_REQ_PEAK_G, 3)}
        except Exception:
            req = None
        out["roofline"] = roofline_block(alg, ms, launches, traffic, busy)
        pmc_fields = ["traffic", "random_request_ceiling", "valu_issue_frac", "l1_requests_per_chain_round", "l1_request_latency_clocks",
                      "l1_requests_in_flight_per_cu", "insts_per_wavefront", "hbm_write_bytes_per_launch"]
        out["roofline"].update({
            # where each field comes from: the PMC-derived ones are REPLAYED from the committed summary of builder-side
            # rocprofv3 --pmc passes over this same command (rocprofv3 cannot run inside this process), accepted only when
            # the summary's kernel-source hash equals the hash of the kernels this run executes; everything else is measured here
            "traffic_source": ("profiles/pmc_latest.json -- builder-side rocprofv3 --pmc passes (tools/pmc_probe.sh, separate runs per "
                               "counter group, kernels serialised by the collection), kernels_sha %s; replayed, not measured in this run"
                               % kernels_sha()) if traffic is not None else "none (no PMC summary for these kernels / this workload)",
            "pmc_derived_fields": pmc_fields,
            "measured_in_this_run": ["achieved", "frac", "avg_launch_us", "busy_ms", "launches", "concurrent_launches",
                                     "algorithmic_bytes_per_launch", "work", "search_kernel_alone"],
            "traffic_kernels_sha": kernels_sha(),
            "kernel": "sr::k_round_mc (four chains per wavefront: apply of the last proposal + Hamming search)"
                      + ("; two chain groups: a launch covers half of the chains and runs beside the other group's launch" if phases == 2 else ""),
            "phases": phases, "chains_per_launch": int(st["chains"]) // phases,
            "algorithmic_bytes_per_read": round(alg / n, 1),
            "algorithmic_bytes_model": "SURVEY 8(d): P*16 + Kv*(4+B) + C*(7+B) [search] + R*12 + E*16 [claims, emission]; "
                                       "the chain state the kernel also moves (counts, consensus) is not counted",
            "work": {"probes": sr["probes"], "keyok": sr["keyok"], "cands": sr["cands"], "hits": sr["hits"]},
            "random_request_ceiling": req,
            # from the same PMC summary: the share of the launch during which a SIMD's vector ALU is issuing
            # (SQ_ACTIVE_INST_VALU quad-cycles x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE)), and the instruction mix per wavefront
            "valu_issue_frac": ks.get("valu_issue_frac") if traffic is not None else None,
            # the L1 miss queue (profiles/r03_tcp_queue.txt, r04_minimizer_table.txt): L1 -> L2 read requests per chain and round,
            # their mean latency in L1 clocks, and how many are in flight per CU on average (the queue is 64 deep)
            "l1_requests_per_chain_round": (round(ks["l1_read_requests_per_launch"] / (st["chains"] / phases), 1)
                                            if traffic is not None and ks.get("l1_read_requests_per_launch") else None),
            "l1_request_latency_clocks": (round(ks["l1_read_request_latency_clocks"], 0)
                                          if traffic is not None and ks.get("l1_read_request_latency_clocks") else None),
            "l1_requests_in_flight_per_cu": (round(ks["l1_requests_in_flight_per_cu"], 1)
                                             if traffic is not None and ks.get("l1_requests_in_flight_per_cu") else None),
            "insts_per_wavefront": ks.get("insts_per_wave") if traffic is not None else None,
            "hbm_write_bytes_per_launch": round(ks["write_bytes_per_launch"], 1) if traffic is not None else None,
        })
        ms2, l2 = st_2k["ms_search_kernel"], max(st_2k["search_launches"], 1)
        if ms2 > 0:
            a2 = alg_search / (ms2 * 1e-3) / 1e9
            out["roofline"]["search_kernel_alone"] = {
                "kernel": "sr::k_search (two-kernel round, opts.fused = -1: the same search as its own launch)",
                "avg_launch_us": round(ms2 * 1e3 / l2, 2), "achieved": round(a2, 2), "frac": round(a2 / HBM_PEAK_GBS, 5),
                "algorithmic_bytes_per_launch": round(alg_search / l2, 1),
                "stage_ms_chains_two_kernel_round": round(st_2k["ms_chains"], 2)}
    if rank == 0 and world == 1 and not a.no_roofline:
        # row f2: the encoder stage chained on the same workload, streams never leave HBM.
        # Reported beside the headline, never part of `value`.
        try:
            from spring_amd.encoder import EncoderStage
            s = spring_amd.ReorderStage(spring_amd.ReorderOpts(device=dev, num_chains=a.chains, num_thr=a.num_thr))
            s.load_dna_device(buf.data_ptr(), nb, n, L, True)
            s.run()
            with EncoderStage(dev) as enc:
                enc.encode(s)          # first call pays the allocations
                info = enc.encode(s)
            s.close()
            names = ("contigs", "sort", "consensus", "pool_dict", "align", "merge", "noise", "tail")
            out["encoder_stage"] = {
                "ms_device": round(info["ms_device"], 2), "Mreads_per_s": round(n / info["ms_device"] / 1e3, 1),
                "phases_ms": {k: round(v, 2) for k, v in zip(names, info["ms_phase"])},
                "contigs": info["num_contigs"], "consensus_bases": info["seq_len"],
                "singletons_aligned": info["matched_s"], "substitutions": info["n_noisepos"],
                "align_passes": info["align_passes"],
            }
        except Exception as e:  # the widened row must never break the headline line
            out["encoder_stage"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_roofline:
        # SURVEY 8(d) legs (ii) and (iii), reported beside `value` (never part of it):
        # (ii) stage incl. PCIe: the .dna record stream starts in HOST memory and every output stream ends there
        try:
            host = buf.cpu().numpy()
            t0 = time.perf_counter()
            s = spring_amd.ReorderStage(spring_amd.ReorderOpts(device=dev, num_chains=a.chains, num_thr=a.num_thr))
            s.load_dna(host, n, L)
            s.run()
            res = s.streams()
            s.close()
            t_pcie = time.perf_counter() - t0
            out["stage_incl_pcie"] = {"value": round(n / t_pcie / 1e6, 2), "unit": "Mreads/s", "seconds": round(t_pcie, 3),
                                      "what": "host .dna buffer (%.1f GB, pageable; through pinned chunks) -> HBM -> stage -> "
                                              "every output stream back in host arrays (%.1f GB)" % (nb / 1e9, sum(v.nbytes for v in res.values() if hasattr(v, "nbytes")) / 1e9)}
            del res, host
        except Exception as e:
            out["stage_incl_pcie"] = {"error": repr(e)}
        # (iii) stage incl. file I/O = what the reference's "Time for this step" covers (spring.cpp:152-160): the drop-in
        # spring_reorder_run on a temp dir (reads + deletes input_clean_1.dna, writes the per-tid and singleton files)
        if a.files_sample > 0:
            try:
                import shutil
                import tempfile
                nf = min(a.files_sample, n)
                td = tempfile.mkdtemp(prefix="spring_bench_")
                recb = L_.spring_synth_dna_bytes(nf, L)
                with open(os.path.join(td, "input_clean_1.dna"), "wb") as f:
                    f.write(buf[:recb].cpu().numpy().tobytes())   # the first nf reads of the workload
                os.sync()
                # the drop-in prints the reference's "Reordering done, N were unmatched" on stdout (reorder.h:638);
                # this process must print exactly one line there: route fd 1 to stderr for the duration of the call
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    t0 = time.perf_counter()
                    spring_amd.call_reorder(td, spring_amd.CompressionParams(L, [nf, 0], num_thr=a.num_thr),
                                            spring_amd.ReorderOpts(device=dev, num_chains=a.chains, num_thr=a.num_thr))
                    t_files = time.perf_counter() - t0
                finally:
                    C.CDLL(None).fflush(None)  # the library's printf sits in the C stdio buffer
                    os.dup2(saved, 1)
                    os.close(saved)
                out_bytes = sum(os.path.getsize(os.path.join(td, f)) for f in os.listdir(td))
                shutil.rmtree(td, ignore_errors=True)
                out["stage_incl_files"] = {"value": round(nf / t_files / 1e6, 2), "unit": "Mreads/s", "seconds": round(t_files, 3),
                                           "sample": "%d reads (the first of the workload), temp dir %s" % (nf, os.path.dirname(td)),
                                           "what": "spring_reorder_run: read + delete input_clean_1.dna (%.2f GB), stage, write "
                                                   "the %d per-tid file sets + singleton files (%.2f GB)" % (recb / 1e9, a.num_thr, out_bytes / 1e9)}
            except Exception as e:
                out["stage_incl_files"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not a.no_roofline and a.sweep_sample > 0:
        # Beside the headline, never part of `value`: the stage on pools of other depths with the library's defaults
        # (chain count, kernel variant, probe plan, k_long all chosen from the dictionary) -- every BASELINE config is
        # uniform 25x, real read sets are not.  Second, warm run of each pool.
        try:
            del buf
            torch.cuda.empty_cache()
            sweep = []
            ns = a.sweep_sample
            pools = [(ns, 0, max(ns * L // c, 4 * L), "%dx" % c) for c in (100, 400, 1600, 6400, 25600)]
            pools.append((ns // 2, 0, 5400, "PhiX-like (%d reads over 5.4 kb)" % (ns // 2)))
            # a genome with the repeat structure of a real one (SPRING_SYNTH_GENOMIC: Zipf-sized repeat families, tandem
            # repeats, low-complexity runs; 25x): bins of thousands of reads beside single-read bins
            pools.append((ns, 0x20000000, max(ns * L // 25, 4 * L), "genome-like 25x (repeat families, tandem repeats)"))
            if n >= 5 * ns:  # ... and at the headline's size (its repeat families grow with the genome: bins of tens of thousands of reads)
                pools.append((n, 0x20000000, max(n * L // 25, 4 * L), "genome-like 25x at the headline size"))
            for pn, pflag, pG, name in pools:
                pb = L_.spring_synth_dna_bytes(pn, L)
                tb = torch.empty(pb, dtype=torch.uint8, device="cuda")
                assert L_.spring_synth_dna_device(C.c_void_p(tb.data_ptr()), pn, L, pG, 11, a.err_ppm | pflag) == 0
                torch.cuda.synchronize()
                for it in range(2):
                    t0 = time.perf_counter()
                    s = spring_amd.ReorderStage(spring_amd.ReorderOpts(device=dev, num_chains=0, num_thr=a.num_thr))
                    s.load_dna_device(tb.data_ptr(), pb, pn, L, True)
                    s.run()
                    t1 = time.perf_counter() - t0
                    ps = s.stats()
                    s.close()
                sweep.append({"pool": name, "reads": pn, "Mreads_per_s": round(pn / t1 / 1e6, 1), "chains": ps["chains"],
                              "rounds": ps["rounds"], "lost_proposals": ps["lost"], "searches_by_k_long": ps["long_searches"],
                              "split_searches": ps["long_splits"], "candidates_per_proposal": ps["alternatives"],
                              "chains_stage_ms": round(ps["ms_chains"], 1), "dictionary": {"deep": bool(ps["deep_pool"] & 1), "heavy_tail": bool(ps["deep_pool"] & 2)}})
                del tb
            out["coverage_sweep"] = {"read_len": L, "err_ppm": a.err_ppm, "pools": sweep,
                                     "what": "the same stage, library defaults, inputs resident in HBM, whole-stage wall clock"}
        except Exception as e:  # noqa: BLE001
            out["coverage_sweep"] = {"error": repr(e)}
    if rank == 0 and world == 1 and a.cpu_sample > 0:
        # CPU baseline on the GPU box's host cores: the C port of the reference algorithm
        # (oracle/reorder_oracle.c).  Multi-thread leg = free-running OpenMP chains like the
        # reference's `-t T` (orc_reorder_omp); single-thread leg = the `-t 1` restatement.
        # Bounded samples of the same distribution (same generator, coverage, error rate, read length).
        from oracle import pyoracle as po
        T = a.cpu_threads or min(32, os.cpu_count() or 1)  # the port stops scaling at ~32 threads on the GPU box

        def sample(ns):
            Gs = max(ns * L // a.coverage, 2 * L)
            nbs = L_.spring_synth_dna_bytes(ns, L)
            b = torch.empty(nbs, dtype=torch.uint8, device="cuda")
            assert L_.spring_synth_dna_device(C.c_void_p(b.data_ptr()), ns, L, Gs, seed + 7, a.err_ppm) == 0
            return b.cpu().numpy().tobytes()

        ns = min(a.cpu_sample, n)
        dna = sample(ns)
        t0 = time.perf_counter()
        read, ln = po.load_dna(dna, ns, L)
        po.reorder_omp(read, ln, L, T)
        tm = time.perf_counter() - t0
        ph_dict, ph_chains = po.last_omp_phases()
        # the reference's default thread count (-t 8, main.cpp:70) on a quarter of the sample
        ns8 = min(max(min(ns, 32_000_000) // 8, 200_000), ns)
        dna8 = sample(ns8)
        t0 = time.perf_counter()
        read8, ln8 = po.load_dna(dna8, ns8, L)
        po.reorder_omp(read8, ln8, L, 8)
        t8 = time.perf_counter() - t0
        del read8, ln8, dna8
        ns1 = min(max(min(ns, 32_000_000) // 32, 200_000), ns)
        dna1 = sample(ns1)
        t0 = time.perf_counter()
        read1, ln1 = po.load_dna(dna1, ns1, L)
        po.reorder_serial(read1, ln1, L)
        t1 = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": round(ns / tm / 1e6, 4), "unit": "Mreads/s", "cores": T, "kind": "port",
            "sample": ("the whole workload: " if ns == n else "") + "%d x %d bp reads, same generator/coverage/error rate; C port of the reference with %d "
                      "free-running OpenMP threads (load + dictionaries + reorder), %.1f s" % (ns, L, T, tm),
            "phases_s": {"dictionaries": round(ph_dict, 2), "chains": round(ph_chains, 2)},
            "chains_only_value": round(ns / ph_chains / 1e6, 4) if ph_chains > 0 else None,
            "threads_8": {"value": round(ns8 / t8 / 1e6, 4), "sample_reads": ns8, "seconds": round(t8, 1)},
            "single_thread": {"value": round(ns1 / t1 / 1e6, 4), "sample_reads": ns1, "seconds": round(t1, 1)},
            "host_cpus": os.cpu_count(), "host_physical_cores": physical_cores(),
        }
        # the same port on a genome-like pool (SPRING_SYNTH_GENOMIC: repeat families, tandem repeats; coverage_sweep has the
        # GPU's figure): what realistic repeat structure costs the CPU algorithm
        try:
            nsg = min(max(min(ns, 32_000_000) // 4, 200_000), ns)  # (large enough for the repeat families to make deep bins: the cost grows with the pool)
            Gg = max(nsg * L // a.coverage, 2 * L)
            bg = torch.empty(L_.spring_synth_dna_bytes(nsg, L), dtype=torch.uint8, device="cuda")
            assert L_.spring_synth_dna_device(C.c_void_p(bg.data_ptr()), nsg, L, Gg, 11, a.err_ppm | 0x20000000) == 0
            dnag = bg.cpu().numpy().tobytes()
            del bg
            t0 = time.perf_counter()
            readg, lng = po.load_dna(dnag, nsg, L)
            po.reorder_omp(readg, lng, L, T)
            tg = time.perf_counter() - t0
            out["cpu_baseline"]["genome_like"] = {"value": round(nsg / tg / 1e6, 4), "sample_reads": nsg, "threads": T, "seconds": round(tg, 1)}
            del readg, lng, dnag
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"]["genome_like"] = {"error": repr(e)}
    if rank == 0 and world == 1 and a.cpu_sample > 0 and "cpu_baseline" in out:
        # the reference's own span ("Time for this step", spring.cpp:152-160): input_clean_1.dna from disk, load,
        # dictionaries, reorder, and the per-tid output files written -- the CPU twin of stage_incl_files
        try:
            import shutil
            import tempfile
            td = tempfile.mkdtemp(prefix="spring_bench_cpu_")
            fn = os.path.join(td, "input_clean_1.dna")
            nsf = min(a.cpu_files_sample, ns)
            with open(fn, "wb") as f:
                f.write(dna[:L_.spring_synth_dna_bytes(nsf, L)])  # (fixed-length records: the first nsf reads of the sample)
            os.sync()
            t0 = time.perf_counter()
            raw = np.fromfile(fn, np.uint8).tobytes()
            os.remove(fn)
            readf, lnf = po.load_dna(raw, nsf, L)
            res = po.reorder_omp(readf, lnf, L, T)
            toff = res["tid_off"]
            for t in range(len(toff) - 1):
                a0, a1 = int(toff[t]), int(toff[t + 1])
                res["order"][a0:a1].tofile(os.path.join(td, "read_order.bin.%d" % t))
                for k, nm in (("rc", "read_rev.txt"), ("flag", "tempflag.txt"), ("pos", "temppos.txt"), ("rlen", "read_lengths.bin")):
                    res[k][a0:a1].tofile(os.path.join(td, "%s.%d" % (nm, t)))
                with open(os.path.join(td, "temp.dna.%d" % t), "wb") as f:
                    f.write(po.write_dna_stream(readf, lnf, L, res["order"][a0:a1], res["rc"][a0:a1]))
            with open(os.path.join(td, "temp.dna.singleton"), "wb") as f:
                f.write(po.write_dna_stream(readf, lnf, L, res["order_s"], None))
            res["order_s"].tofile(os.path.join(td, "read_order.bin.singleton"))
            tf = time.perf_counter() - t0
            shutil.rmtree(td, ignore_errors=True)
            out["cpu_baseline"]["stage_incl_files_value"] = round(nsf / tf / 1e6, 4)
            out["cpu_baseline"]["stage_incl_files_what"] = (
                "the same port on the first %d reads of the sample through the reference's span: read + delete input_clean_1.dna, load, "
                "dictionaries, reorder at %d threads, write the per-tid and singleton files (uncompressed), %.1f s" % (nsf, T, tf))
            del res, readf, lnf, raw
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"]["stage_incl_files_value"] = None
            out["cpu_baseline"]["stage_incl_files_error"] = repr(e)
    if rank == 0 and world == 1 and a.cost_sample > 0 and not a.no_roofline:
        # what the speed costs: compressed size of the encoder's output streams after the REFERENCE'S OWN BSC (src/libbsc
        # compiled in place, test infrastructure oracle/_ref/ref_bsc, outside any timed region) for the default chain
        # count against the reference's own granularity K = num_thr, same reads
        try:
            out["compression_cost"] = compression_cost(spring_amd, a, dev, n / max(float(st.get("chains", 0)), 1.0), int(st.get("phases", 1)))
        except Exception as e:  # noqa: BLE001
            out["compression_cost"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    lanes.close()


if __name__ == "__main__":
    main()
