cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r20
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new.pt 2 4 && QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 && python tools/dbg/gqa_ab.py --cmp /tmp/new.pt /tmp/old.pt ) 2>&1 | tail -1
for i in 1 2 3; do
echo -n "ohead: "; timeout 600 python tools/dbg/tok70b.py 48 2>&1 | tail -1
echo -n "noohead: "; QUIP_LIB_PATH=$PWD/tools/dbg/libquip_noohead.so timeout 600 python tools/dbg/tok70b.py 48 2>&1 | tail -1
done
python tools/gqa_stamps.py 16 8 40 > gpurun_out/r20/stamps.txt 2>&1; grep "head:\|wait for a\|a gathered\|in(o)\|block span" gpurun_out/r20/stamps.txt
timeout 1500 python -m pytest tests/test_gpu_block_engine_gqa.py -x -q -m gpu -s 2>&1 | grep -v amdgpu | grep "ulps\|passed\|failed\|block(s)\|Error\|assert" | tail -16
