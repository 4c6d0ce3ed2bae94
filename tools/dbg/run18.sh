cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r18
timeout 600 python tools/gemv_v2_bench.py --shapes odd,7b,70b --check-only --variants "4,0,0,0,0,0;4,2,0,2,0,0;4,4,100,3,0,1;4,3,0,0,12,8;516,0,0,0,0,0" 2>&1 | grep -c "bit-identical"
timeout 600 python tools/gemv_v2_bench.py --shapes odd,7b,70b --check-only --variants "4,0,0,0,0,0;4,2,0,2,0,0;4,4,100,3,0,1;4,3,0,0,12,8;516,0,0,0,0,0" 2>&1 | grep -c "MISMATCH\|rc \|NOT-ZERO"
V="1796,0,0,0,0,0;4,0,0,0,0,0;260,0,0,0,0,0;516,0,0,0,0,0;1028,0,0,0,0,0;1796,0,0,0,0,0;4,0,0,0,0,0;260,0,0,0,0,0;516,0,0,0,0,0;1028,0,0,0,0,0;32,0,0,0,0,0"
for i in 1 2; do
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --variants "$V" 2>&1 | grep -v "amdgpu.ids" 
done | tee gpurun_out/r18/ab.txt | grep -v "N=  1024" 
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --variants "1796,0,0,0,0,0;4,0,0,0,0,0" --phases 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r18/phases.txt
