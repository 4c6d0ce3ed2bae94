cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r10
timeout 600 ./tools/ubench/lds_lookup_rate > gpurun_out/r10/lds.txt 2>&1
sed -n '/part 2/,$p' gpurun_out/r10/lds.txt
bash tools/prof_kernel.sh r10_stream decode_block_gqa python $PWD/tools/gqa_stream.py 80 6 > gpurun_out/r10/pmc_stream.txt 2>&1
grep -v "^W2026" gpurun_out/r10/pmc_stream.txt | grep -v "^==" 
