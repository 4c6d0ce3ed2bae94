#!/bin/bash
# rocprofv3 kernel trace + FETCH_SIZE of the 70B launch in its measurement mode (products only): profiles/r05_gqa_stream_*.txt
# usage (on the GPU box, from the repo root): tools/prof_gqa_stream.sh [layers]
set -e
L=${1:-80}
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_gqa_stream
rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python $OLDPWD/tools/gqa_stream.py $L 6 > $out/trace_stdout.txt 2>&1) || true
(cd /tmp && rocprofv3 --pmc FETCH_SIZE -d $out/pmc -o p -- python $OLDPWD/tools/gqa_stream.py $L 6 > $out/pmc_stdout.txt 2>&1) || true
python tools/rocpd_summary.py $out decode_block_gqa
cat $out/trace_stdout.txt | tail -2
