cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for w in 0 40 150; do echo "== warm-ms $w"; timeout 600 python tools/gemv_v2_bench.py --shapes 70b,7b --warm-ms $w --variants "0,0,0,0,0,0" 2>&1 | grep -v amdgpu; done
timeout 600 python tools/gemv_v2_bench.py --shapes 70b --groups --warm-ms 40 --variants "0,0,0,0,0,0" 2>&1 | grep -v amdgpu
