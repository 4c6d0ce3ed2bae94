#!/bin/bash
# PMC passes for one GEMV variant (run on the GPU box through gpurun).
# usage: prof_gemv.sh <tag> <gemv_one args...>
set -u
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"
P3="SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d $out/p$i -o p$i -- python $R/tools/gemv_one.py "$@" > $out/p$i.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $R/tools/gemv_one.py "$@" > $out/kt.log 2>&1
python - <<PY
import csv, glob, collections, os
out="$out"
for p in sorted(glob.glob(out+"/p*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(p)):
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k,v in agg.items():
        if "gemv" in k or "probe" in k:
            print(os.path.basename(os.path.dirname(p)), k)
            for c,val in v.items(): print("    %-28s %.4g"%(c,val))
for p in glob.glob(out+"/kt/**/*kernel_stats.csv", recursive=True):
    print(open(p).read()[:1500])
PY
