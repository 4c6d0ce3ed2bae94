cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r25
bash tools/prof_bench.sh r06 --no-extras --no-prefill > gpurun_out/r25/prof_bench_7b.log 2>&1
bash tools/prof_bench.sh r06_70b --model 70b --no-extras --no-prefill > gpurun_out/r25/prof_bench_70b.log 2>&1
bash tools/prof_kernel.sh r06_engine decode_block_kernel python $R/tools/block_stamps.py 32 16 100 > gpurun_out/r25/r06_engine_pmc.txt 2>&1
bash tools/prof_kernel.sh r06_gqa_engine decode_block_gqa python $R/tools/gqa_stamps.py 80 40 40 > gpurun_out/r25/r06_gqa_engine_pmc.txt 2>&1
bash tools/prof_kernel.sh r06_gqa_stream decode_block_gqa python $R/tools/gqa_stream.py 80 6 > gpurun_out/r25/r06_gqa_stream_pmc.txt 2>&1
cd $R
timeout 600 python tools/block_stamps.py 32 16 100 > gpurun_out/r25/r06_block_stamps.txt 2>&1
timeout 600 python tools/gqa_stamps.py 16 8 40 > gpurun_out/r25/r06_gqa_block_stamps.txt 2>&1
timeout 600 python tools/gqa_stream.py 80 12 > gpurun_out/r25/gqa_stream.txt 2>&1
timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r25/tok70b.txt 2>&1
( time timeout 1500 python bench.py ) > gpurun_out/r25/bench_default.log 2>&1
grep '^{"metric"' gpurun_out/r25/bench_default.log | tail -1 > gpurun_out/r25/bench_line.json
tail -3 gpurun_out/r25/bench_default.log | cut -c1-300
ls gpurun_out/
