cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V="4,0,0,0,0,0;4,0,0,0,0,1;4,0,0,0,0,2;4,0,0,0,0,4;4,0,0,0,0,8;32,0,0,0,0,0"
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --check-only --variants "4,0,0,0,0,1;4,3,0,2,0,2;4,0,0,0,12,0" 2>&1 | grep -c "bit-identical"
for i in 1 2; do timeout 600 python tools/gemv_v2_bench.py --shapes 70b --warm-ms 40 --variants "$V" 2>&1 | grep -v "amdgpu\|N=  1024" ; done
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --warm-ms 40 --variants "4,0,0,0,0,2" --phases 2>&1 | grep -v "amdgpu\|WGs"
