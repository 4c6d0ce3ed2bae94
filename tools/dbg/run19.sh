cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r19
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
( time timeout 1500 python bench.py ) > gpurun_out/r19/bench_default.log 2>&1
tail -4 gpurun_out/r19/bench_default.log | cut -c1-600
grep '^{"metric"' gpurun_out/r19/bench_default.log | tail -1 > gpurun_out/r19/bench_line.json
