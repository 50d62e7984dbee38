"""tools/pmc_aggregate.py <pmc_dir> <reads> <out.json>: per-kernel totals of the rocprofv3 --pmc passes written by
tools/pmc_probe.sh (one sub-directory per pass: fetch, write, sq, tcc, grbm), in the shape bench.py reads
(profiles/pmc_latest.json).  FETCH_SIZE / WRITE_SIZE are in KiB-units of the guide: bytes = value * 1024
(calibrated on tools/random_gather_bench.hip: one 64-byte request per random access, no x2 correction)."""
import collections
import csv
import json
import os
import sys

pmc_dir, reads, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
csv.field_size_limit(1 << 30)


def short(name):
    n = name.split("(")[0].replace("void ", "")
    if n.startswith("sr::k_round_mc<"):
        return "sr::k_round_mc"   # the production instantiation is the only one scale_probe launches
    if n.startswith("sr::k_round<"):
        return "sr::k_round"
    return n


def kernels_sha():
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("reorder_kernels.hip", "reorder_device.h", "reorder_round_mc.h"):
        h.update(open(os.path.join(root, "spring_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


kern = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
dur = collections.defaultdict(lambda: collections.defaultdict(float))
for p in ("fetch", "write", "sq", "insts", "tcc", "grbm", "tcp"):
    d = os.path.join(pmc_dir, p)
    if not os.path.isdir(d):
        continue
    for r in csv.DictReader(open(os.path.join(d, "pmc_counter_collection.csv"))):
        k = short(r["Kernel_Name"])
        kern[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if p == "fetch":
            launches[k].add(r["Dispatch_Id"])
    for r in csv.DictReader(open(os.path.join(d, "pmc_kernel_trace.csv"))):
        dur[short(r["Kernel_Name"])][p] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
res = {"reads": reads, "read_len": 150, "chains": 65536, "kernels_sha": kernels_sha(),
       "source": "tools/pmc_probe.sh (5 separate rocprofv3 --pmc passes) aggregated by tools/pmc_aggregate.py", "kernels": {}}
for k, c in kern.items():
    if not ("k_round" in k or "k_mg_mark" in k or "k_search" in k or "k_apply" in k):
        continue
    n = max(len(launches[k]), 1)
    e = dict(launches=n, total_us_by_pass={p: round(v, 1) for p, v in dur[k].items()})
    e.update(c)
    e["fetch_bytes_per_launch"] = c.get("FETCH_SIZE", 0) * 1024 / n
    e["write_bytes_per_launch"] = c.get("WRITE_SIZE", 0) * 1024 / n
    e["rdreq_per_launch"] = c.get("TCC_EA0_RDREQ_sum", 0) / n
    hm = c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)
    e["l2_hit_rate"] = c.get("TCC_HIT_sum", 0) / hm if hm else None
    # the L1's side of the story (round 3: the launch = L1 -> L2 read requests x their latency / (64 in flight x 256 CUs) while
    # the miss queue is full): requests per launch, mean latency in L1 clocks, requests in flight per CU
    if c.get("TCP_TCC_READ_REQ_sum"):
        e["l1_read_requests_per_launch"] = c["TCP_TCC_READ_REQ_sum"] / n
        e["l1_read_request_latency_clocks"] = c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / c["TCP_TCC_READ_REQ_sum"]
        if c.get("TCP_GATE_EN2_sum"):
            e["l1_requests_in_flight_per_cu"] = c.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / c["TCP_GATE_EN2_sum"]
    e["wait_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None
    # share of the launch during which a SIMD's vector ALU is issuing: SQ_ACTIVE_INST_VALU counts quad-cycles summed over
    # the waves; 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE = busy clocks summed over the 8 XCDs (4.3 M per 217 us launch in
    # round 2's data = 8 x 217 us x 2.5 GHz)
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("GRBM_GUI_ACTIVE"):
        clocks = c["GRBM_GUI_ACTIVE"] / 8.0
        e["valu_issue_frac"] = round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * clocks), 4)
        e["salu_issue_frac"] = round(c.get("SQ_ACTIVE_INST_SCA", 0) * 4.0 / (1024.0 * clocks), 4)
    if c.get("SQ_WAVES") and c.get("SQ_INSTS_VALU"):
        w = c["SQ_WAVES"]
        e["insts_per_wave"] = {k.replace("SQ_INSTS_", "").lower(): round(c[k] / w, 1) for k in sorted(c) if k.startswith("SQ_INSTS_")}
    res["kernels"][k] = e
json.dump(res, open(out, "w"), indent=1)
for k, e in res["kernels"].items():
    print(k, "launches", e["launches"], "fetch MB/launch %.1f" % (e["fetch_bytes_per_launch"] / 1e6),
          "write MB/launch %.1f" % (e["write_bytes_per_launch"] / 1e6), "rdreq M/launch %.2f" % (e["rdreq_per_launch"] / 1e6))
