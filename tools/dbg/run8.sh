cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r8
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new.pt 2 4 && QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 && python tools/dbg/gqa_ab.py --cmp /tmp/new.pt /tmp/old.pt ) > gpurun_out/r8/ab.txt 2>&1
tail -3 gpurun_out/r8/ab.txt
for i in 1 2; do
timeout 600 python tools/gqa_stream.py 80 8 > gpurun_out/r8/stream_new$i.txt 2>&1; tail -1 gpurun_out/r8/stream_new$i.txt
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_nopipe.so timeout 600 python tools/gqa_stream.py 80 8 > gpurun_out/r8/stream_nopipe$i.txt 2>&1; tail -1 gpurun_out/r8/stream_nopipe$i.txt
done
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_ws.so timeout 600 python tools/gqa_waitstat.py 80 4 > gpurun_out/r8/waitstat_ws.txt 2>&1; tail -3 gpurun_out/r8/waitstat_ws.txt
timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r8/tok_new.txt 2>&1; tail -1 gpurun_out/r8/tok_new.txt
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_nopipe.so timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r8/tok_nopipe.txt 2>&1; tail -1 gpurun_out/r8/tok_nopipe.txt
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r8/tok_old.txt 2>&1; tail -1 gpurun_out/r8/tok_old.txt
timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r8/tok_new2.txt 2>&1; tail -1 gpurun_out/r8/tok_new2.txt
timeout 1500 python -m pytest tests/test_gpu_block_engine_gqa.py -x -q -m gpu > gpurun_out/r8/pytest_gqa.txt 2>&1; tail -3 gpurun_out/r8/pytest_gqa.txt
