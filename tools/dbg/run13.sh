cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r13
for sh in 7b g8; do
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new_$sh.pt 3 4 $sh && QUIP_LIB_PATH=tools/dbg/libquip_noohead.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old_$sh.pt 3 4 $sh ; python tools/dbg/gqa_ab.py --cmp /tmp/new_$sh.pt /tmp/old_$sh.pt ) > gpurun_out/r13/ab_$sh.txt 2>&1
echo "== $sh"; tail -12 gpurun_out/r13/ab_$sh.txt
done
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
QUIP_LIB_PATH=tools/dbg/libquip_noohead.so timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
done
timeout 600 python tools/block_stamps.py 32 16 100 > gpurun_out/r13/stamps_ohead.txt 2>&1; tail -32 gpurun_out/r13/stamps_ohead.txt
QUIP_LIB_PATH=tools/dbg/libquip_noohead.so timeout 600 python tools/block_stamps.py 32 16 100 > gpurun_out/r13/stamps_noohead.txt 2>&1; tail -32 gpurun_out/r13/stamps_noohead.txt
timeout 1500 python -m pytest tests/test_gpu_block_engine.py tests/test_gpu_decode.py tests/test_gpu_exhaustive_codes.py -m gpu -x -q 2>&1 | tail -8
