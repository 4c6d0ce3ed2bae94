#!/bin/bash
# rocprofv3 passes for one command (run on the GPU box through gpurun): four --pmc passes (own runs, --kernel-trace
# only, as the pool's rules require) + one --kernel-trace --stats pass; per-kernel summary by tools/rocpd_summary.py.
# usage: prof_kernel.sh <tag> <kernel-name-substring> <command...>
set -u
cd /tmp && export TMPDIR=/tmp
tag=$1; flt=$2; shift; shift
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_MFMA"
P3="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
P4="FETCH_SIZE"
P5="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace -d $out/p$i -o p$i -- "$@" > $out/p$i.log 2>&1 || tail -3 $out/p$i.log
done
timeout 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- "$@" > $out/kt.log 2>&1
echo "## prof_kernel.sh $tag: $*"
python $R/tools/rocpd_summary.py $out "$flt"
find $out -name "*.db" -delete   # keep the text only
