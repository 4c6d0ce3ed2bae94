cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r18
bash tools/prof_bench.sh r06 --no-extras --no-prefill > gpurun_out/r18/prof_bench_7b.log 2>&1; tail -3 gpurun_out/r18/prof_bench_7b.log | cut -c1-300
bash tools/prof_bench.sh r06_70b --model 70b --no-extras --no-prefill > gpurun_out/r18/prof_bench_70b.log 2>&1; tail -3 gpurun_out/r18/prof_bench_70b.log | cut -c1-300
bash tools/prof_kernel.sh r06_engine decode_block_kernel python $PWD/tools/block_stamps.py 32 16 100 > gpurun_out/r06_engine_pmc.txt 2>&1
bash tools/prof_kernel.sh r06_gqa_engine decode_block_gqa python $PWD/tools/gqa_stamps.py 80 40 40 > gpurun_out/r06_gqa_engine_pmc.txt 2>&1
grep -v "^W2026" gpurun_out/r06_gqa_engine_pmc.txt | grep "SQ_WAIT_ANY\|SQ_WAVE_CYCLES\|LDS_BANK\|LDS_IDX\|FETCH\|avg" | head -12
python tools/gqa_stamps.py 16 8 40 > gpurun_out/r06_gqa_block_stamps.txt 2>&1
ls gpurun_out | grep r06
