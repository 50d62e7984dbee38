cd /tmp && export TMPDIR=/tmp
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT unset)}"
: "${1:?usage: see the header comment}"
cd $GRAFT_REPO_ROOT
./tools/whb 8 4 65536 > gpurun_out/r2_whb.log 2>&1
./tools/whb 8 4 131072 >> gpurun_out/r2_whb.log 2>&1
out=gpurun_out/r2_pmc
mkdir -p $out
run() { name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -o pmc -- python tools/scale_probe.py 100000000,150,65536 > $out/$name.log 2>&1
}
run sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
# keep only compact outputs
for p in sq tcc grbm; do
  python - <<PY
import csv, collections, sys
csv.field_size_limit(1<<30)
d="$out/$p"
import glob
f=glob.glob(d+"/**/pmc_counter_collection.csv", recursive=True)
k=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for fn in f:
  for r in csv.DictReader(open(fn)):
    nm=r["Kernel_Name"].split("(")[0]
    k[nm][r["Counter_Name"]]+=float(r["Counter_Value"]); n[nm].add(r["Dispatch_Id"])
dur=collections.defaultdict(float)
for fn in glob.glob(d+"/**/pmc_kernel_trace.csv", recursive=True):
  for r in csv.DictReader(open(fn)):
    dur[r["Kernel_Name"].split("(")[0]]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
with open("$out/$p.summary.txt","w") as o:
  for nm,c in k.items():
    if "k_search" in nm or "k_apply" in nm:
      o.write("%s launches=%d total_us=%.1f %s\n"%(nm,len(n[nm]),dur[nm],dict(c)))
PY
  rm -rf $out/$p
done
