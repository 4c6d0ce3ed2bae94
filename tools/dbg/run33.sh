cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/dbg/tok_shape.py 5120 13824 40 40 40 2>&1 | grep -v amdgpu | tail -1
timeout 600 python tools/dbg/tok_shape.py 5120 13824 40 40 40 2>&1 | grep -v amdgpu | tail -1
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_qlinear.py tests/test_gpu_decode.py tests/test_gpu_hf_generate.py -m gpu -x -q 2>&1 | tail -3
