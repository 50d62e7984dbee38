#!/bin/bash
# tools/pmc_probe.sh <outdir> <reads> -- PMC passes (each in its own rocprofv3 run, kernel-trace only)
out=$1; n=${2:-20000000}
export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o pmc -- \
     python tools/scale_probe.py $n,150,0 > $out/$name.log 2>&1
  ls $out/$name | head -5
}
mkdir -p $out
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
