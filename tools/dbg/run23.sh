cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r23
timeout 1200 python -m pytest tests/test_gpu_gemv_v2.py tests/test_gpu_ops.py tests/test_gpu_qlinear.py tests/test_gpu_exhaustive_codes.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r23/tests.txt
V="0,0,0,0,0,0;4,0,0,0,0,0;32,0,0,0,0,0;0,0,0,0,0,0;4,0,0,0,0,0;32,0,0,0,0,0"
for i in 1 2; do
timeout 600 python tools/gemv_v2_bench.py --shapes 70b,7b --variants "$V" 2>&1 | grep -v "amdgpu.ids"
done | tee gpurun_out/r23/ab.txt
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --variants "32,0,0,0,0,0" --phases 2>&1 | grep -v "amdgpu.ids\|WGs" | tee gpurun_out/r23/phases.txt
timeout 600 python tools/gemv_v2_bench.py --shapes 70b,7b --groups --variants "0,0,0,0,0,0;32,0,0,0,0,0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r23/groups.txt
