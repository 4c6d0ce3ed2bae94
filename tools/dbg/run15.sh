cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r15
V="0,0,0,0,0,0;0,2,0,0,0,1;0,2,0,0,0,2;0,2,0,0,0,8;0,3,0,0,0,2;0,3,0,0,0,4;0,4,0,0,0,2;0,4,0,0,0,4;4,2,0,0,0,2;4,2,0,0,0,4;4,2,0,0,0,8;4,3,0,0,0,2;4,4,0,0,0,2;4,4,0,0,0,8;4,3,0,0,0,8;4,2,0,0,12,0;0,2,0,0,12,0"
for i in 1 2; do
timeout 900 python tools/gemv_v2_bench.py --shapes 70b --variants "$V" 2>&1 | grep -v "^N=  1024\|amdgpu.ids" | tee gpurun_out/r15/time70b_$i.txt | tail -60
done
