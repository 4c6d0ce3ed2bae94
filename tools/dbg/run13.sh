cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r13
timeout 600 python -m pytest tests/test_gpu_exhaustive_codes.py -x -q -m gpu -k "decode_core or block_engine" > gpurun_out/r13/probe.txt 2>&1; tail -5 gpurun_out/r13/probe.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r13/pytest_gpu.txt 2>&1; tail -8 gpurun_out/r13/pytest_gpu.txt
