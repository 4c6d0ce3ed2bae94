cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r19
for i in 1 2 3; do
echo -n "pack: "; timeout 600 python tools/dbg/tok70b.py 48 2>&1 | tail -1
echo -n "nopack: "; QUIP_LIB_PATH=$PWD/tools/dbg/libquip_nopack.so timeout 600 python tools/dbg/tok70b.py 48 2>&1 | tail -1
done
python tools/gqa_stamps.py 16 8 40 > gpurun_out/r19/stamps.txt 2>&1; grep "owners\|12->14\|14->15\|block span" gpurun_out/r19/stamps.txt
timeout 1500 python -m pytest tests/test_gpu_block_engine_gqa.py -x -q -m gpu -s 2>&1 | grep -v amdgpu | grep "ulps\|passed\|failed\|block(s)" | tail -12
