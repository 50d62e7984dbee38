#!/bin/bash
# tools/r2_sweeps.sh -- side measurements of the round (run through gpurun): coverage sweep at 20 M reads, chain-count
# sweep at 100 M, instruction mix of the chain kernels (two rocprofv3 --pmc passes), phase clocks of a debug build.
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_sweeps; mkdir -p $O
python tools/scale_probe.py 20000000,150,0,10000,0,25 20000000,150,0,10000,0,400 20000000,150,0,10000,0,1600 20000000,150,0,10000,0,6400 20000000,150,0,10000,0,25600 > $O/coverage_sweep.txt 2>&1
python tools/scale_probe.py 20000000,150,65536,10000,0,1600 20000000,150,65536,10000,0,400 >> $O/coverage_sweep.txt 2>&1
timeout 300 python tools/deep_bins_probe.py 10000000,150,5400,0 10000000,150,5400,65536 > $O/phix_like.txt 2>&1
timeout 1500 python tools/parity_deep.py 1000000,150,5400,4096 10000000,150,5400,0 10000000,150,5400,65536 20000000,150,1875000,0 > $O/parity_deep.txt 2>&1
python tools/scale_probe.py 100000000,150,65536 100000000,150,49152 100000000,150,32768 100000000,150,131072 > $O/chain_sweep.txt 2>&1
bash tools/pmc_insts.sh $O/insts 100000000 > $O/insts.log 2>&1
# phase clocks: the round kernel compiled with -DSR_PHASE_TIMING (search time by outcome)
L=spring_amd/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Iinclude -Ispring_amd/csrc -DSR_PHASE_TIMING -c spring_amd/csrc/reorder_kernels.hip -o /tmp/rk_tm.o > $O/phase_build.log 2>&1 &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libspring_tm.so /tmp/rk_tm.o $L/reorder_pipeline.o $L/reorder_files.o $L/order_ops.o $L/fastq_kernels.o $L/encoder.o $L/fastq_reorder.o -lz >> $O/phase_build.log 2>&1 &&
SPRING_AMD_LIB=/tmp/libspring_tm.so python tools/phase_clocks.py 100000000 150 > $O/phase_clocks.txt 2>&1
