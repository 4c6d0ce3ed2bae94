cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r17
V="32,0,0,0,0,0;16,0,0,0,0,0;4,0,0,0,0,0;32,0,0,0,0,0;16,0,0,0,0,0;4,0,0,0,0,0;4,2,0,0,0,8;4,2,0,0,12,0"
for i in 1 2 3; do
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --variants "$V" 2>&1 | grep -v "amdgpu.ids" 
done | tee gpurun_out/r17/ab.txt | grep -v "N=  1024" | grep -A8 "N= 28672\|N=  8192 K= 28672"
