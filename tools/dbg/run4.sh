cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_ws2.so timeout 600 python tools/gqa_waitstat.py 80 4 > gpurun_out/r4/waitstat_ws2.txt 2>&1; tail -6 gpurun_out/r4/waitstat_ws2.txt
bash tools/prof_kernel.sh r4_stream decode_block_gqa python $PWD/tools/gqa_stream.py 80 6 > gpurun_out/r4/pmc_stream.txt 2>&1
cat gpurun_out/r4/pmc_stream.txt | grep -v "^W2026"
