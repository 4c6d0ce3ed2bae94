cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r16
timeout 1200 python -m pytest tests/test_gpu_gemv_v2.py tests/test_gpu_ops.py tests/test_gpu_qlinear.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --groups --variants "32,0,0,0,0,0;4,2,0,0,0,0;4,3,0,0,0,0;0,0,0,0,0,0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r16/groups.txt
for i in 1 2; do
QUIP_GEMV_NIB=0 timeout 600 python tools/gemv_v2_bench.py --shapes 70b --variants "0,0,0,0,0,0" 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --variants "0,0,0,0,0,0" 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r16/auto.txt
