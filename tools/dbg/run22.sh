cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r22
timeout 900 python -m pytest tests/test_gpu_gemv_v2.py -m gpu -x -q 2>&1 | tail -3
V="32,0,0,0,0,0;4,0,0,0,0,0;32,0,0,0,0,0;4,0,0,0,0,0;4,3,0,0,0,0"
for i in 1 2; do
timeout 600 python tools/gemv_v2_bench.py --shapes 70b,7b --variants "$V" 2>&1 | grep -v "amdgpu.ids"
done | tee gpurun_out/r22/ab.txt
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --variants "4,0,0,0,0,0" --phases 2>&1 | grep -v "amdgpu.ids\|WGs" | tee gpurun_out/r22/phases.txt
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --groups --variants "32,0,0,0,0,0;4,0,0,0,0,0;32,0,0,0,0,0;4,0,0,0,0,0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r22/groups.txt
