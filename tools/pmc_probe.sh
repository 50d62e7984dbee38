#!/bin/bash
# tools/pmc_probe.sh <outdir> <reads> -- PMC passes (each in its own rocprofv3 run, kernel-trace only) over the
# headline configuration (K = 65536 chains); aggregate with tools/pmc_aggregate.py <outdir> <reads> <out.json>
set -u
out=$1; n=${2:-20000000}
export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o pmc -- \
     python tools/scale_probe.py "$n",150,65536 > "$out/$name.log" 2>&1
  ls "$out/$name" | head -3
}
mkdir -p "$out"
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN2_sum
