cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
QUIP_LIB_PATH=tools/dbg/libquip_argstamp.so timeout 600 python tools/gemv_v2_bench.py --shapes 70b --variants "4,0,0,0,0,0" --phases 2>&1 | grep -v "amdgpu.ids\|WGs"
