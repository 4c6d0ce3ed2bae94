cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new.pt 2 4 && QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 && python tools/dbg/gqa_ab.py --cmp /tmp/new.pt /tmp/old.pt ) > gpurun_out/r5/ab.txt 2>&1
tail -3 gpurun_out/r5/ab.txt
timeout 600 python tools/gqa_stream.py 80 8 > gpurun_out/r5/stream_new.txt 2>&1; tail -1 gpurun_out/r5/stream_new.txt
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_ws.so timeout 600 python tools/gqa_waitstat.py 80 4 > gpurun_out/r5/waitstat_ws.txt 2>&1; tail -5 gpurun_out/r5/waitstat_ws.txt
timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r5/tok_new.txt 2>&1; tail -1 gpurun_out/r5/tok_new.txt
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r5/tok_old.txt 2>&1; tail -1 gpurun_out/r5/tok_old.txt
timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r5/tok_new2.txt 2>&1; tail -1 gpurun_out/r5/tok_new2.txt
