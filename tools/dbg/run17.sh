cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r17
timeout 1500 python -m pytest tests/test_gpu_block_engine_gqa.py -x -q -m gpu --durations=5 > gpurun_out/r17/gqa.txt 2>&1; grep -v amdgpu.ids gpurun_out/r17/gqa.txt | tail -15
timeout 600 python tools/gqa_stream.py 80 8 2>&1 | tail -1
