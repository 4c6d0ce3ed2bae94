cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
V="32,0,0,0,0,0;32,2,0,0,8,0;32,3,0,0,8,0;32,4,0,0,8,0;32,4,0,0,12,0;32,3,0,0,12,0;4,0,0,0,0,0;4,3,0,0,8,0;4,4,0,0,8,0;4,4,0,0,12,0"
for i in 1 2; do timeout 600 python tools/gemv_v2_bench.py --shapes 70b --warm-ms 40 --variants "$V" 2>&1 | grep -v "amdgpu\|N=  1024" ; done
