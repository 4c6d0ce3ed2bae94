cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r14
V="0,0,0,0,0,0;4,2,0,0,0,0;4,3,0,0,0,0;4,4,0,0,0,0;4,6,0,0,0,0;4,8,0,0,0,0"
timeout 600 python tools/gemv_v2_bench.py --shapes odd,7b,70b --check-only --variants "4,0,0,0,0,0;4,2,0,2,0,0;4,4,100,3,0,1" 2>&1 | tee gpurun_out/r14/check.txt | tail -40
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --check-only --groups --variants "4,0,0,0,0,0" 2>&1 | tail -8
timeout 900 python tools/gemv_v2_bench.py --shapes 70b --variants "$V" --phases 2>&1 | tee gpurun_out/r14/time70b.txt | tail -60
timeout 900 python tools/gemv_v2_bench.py --shapes 7b --variants "0,0,0,0,0,0;4,3,0,0,0,0;4,4,0,0,0,0" 2>&1 | tee gpurun_out/r14/time7b.txt | tail -20
