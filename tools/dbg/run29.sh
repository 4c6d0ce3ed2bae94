cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r29
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/r29/bench_driver.log 2>&1
grep '^{"metric"' gpurun_out/r29/bench_driver.log | tail -1 > gpurun_out/r29/bench_line.json
tail -4 gpurun_out/r29/bench_driver.log | cut -c1-200
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r29/gpu_tests.txt
