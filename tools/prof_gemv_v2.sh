#!/bin/bash
# rocprofv3 passes for one GEMV launch configuration (run on the GPU box through gpurun): three --pmc passes (own
# runs, --kernel-trace only, as the pool's rules require) + one --kernel-trace --stats pass; summary printed by
# tools/rocpd_summary.py.   usage: prof_gemv_v2.sh <tag> <gemv_v2_one.py args...>
set -u
cd /tmp && export TMPDIR=/tmp
tag=$1; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
P4="FETCH_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d $out/p$i -o p$i -- python $R/tools/gemv_v2_one.py "$@" > $out/p$i.log 2>&1 || tail -3 $out/p$i.log
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $R/tools/gemv_v2_one.py "$@" > $out/kt.log 2>&1
echo "## prof_gemv_v2.sh $tag: gemv_v2_one.py $*"
python $R/tools/rocpd_summary.py $out gemv
find $out -name "*.db" -delete   # keep the text only
