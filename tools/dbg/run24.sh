cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r24
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new.pt 3 4 70b && QUIP_LIB_PATH=tools/dbg/libquip_norowsearly.so timeout 300 python tools/dbg/gqa_ab.py /tmp/var.pt 3 4 70b ; python tools/dbg/gqa_ab.py --cmp /tmp/new.pt /tmp/var.pt ) 2>&1 | grep -v amdgpu | tail -3
for i in 1 2; do
timeout 600 python tools/dbg/tok70b.py 2>&1 | tail -1
QUIP_LIB_PATH=tools/dbg/libquip_norowsearly.so timeout 600 python tools/dbg/tok70b.py 2>&1 | tail -1
done
timeout 600 python tools/gqa_stamps.py 16 8 40 2>&1 | grep "14->15\|12->14\|block span"
QUIP_LIB_PATH=tools/dbg/libquip_norowsearly.so timeout 600 python tools/gqa_stamps.py 16 8 40 2>&1 | grep "14->15\|12->14\|block span"
timeout 900 python -m pytest tests/test_gpu_block_engine_gqa.py -m gpu -x -q 2>&1 | tail -3
