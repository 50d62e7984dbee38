#!/bin/bash
# Experiment build of the chain kernels: tools/xbuild.sh NAME [extra hipcc flags...]
# Compiles reorder_kernels.hip (+ reorder_pipeline.cpp when XPIPE=1) with the given flags and links them with the
# objects of the last regular build (python -m spring_amd.build) into spring_amd/lib/x_NAME.so; run a tool against it
# with SPRING_AMD_LIB=spring_amd/lib/x_NAME.so.  -DSR_DEV_PROD_ONLY compiles the production k_round variants only
# (seconds instead of a minute).  The .so travels to the GPU box with gpurun; it is git-ignored.
set -eu
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/spring_amd/lib
csrc=$root/spring_amd/csrc
mkdir -p "$lib/x"
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -I$root/include -I$csrc"
/opt/rocm/bin/hipcc $common "$@" -c "$csrc/reorder_kernels.hip" -o "$lib/x/${name}_kernels.o" &
pipe=$lib/reorder_pipeline.o
if [ "${XPIPE:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc $common "$@" -c "$csrc/reorder_pipeline.cpp" -o "$lib/x/${name}_pipeline.o" &
  pipe=$lib/x/${name}_pipeline.o
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$lib/x_$name.so" "$lib/x/${name}_kernels.o" "$pipe" \
  "$lib/reorder_files.o" "$lib/order_ops.o" "$lib/fastq_kernels.o" "$lib/encoder.o" "$lib/fastq_reorder.o" -lz
echo "$lib/x_$name.so"
