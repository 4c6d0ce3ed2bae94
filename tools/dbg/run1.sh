set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r1
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new.pt 2 4 && QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 && python tools/dbg/gqa_ab.py --cmp /tmp/new.pt /tmp/old.pt ) > gpurun_out/r1/ab.txt 2>&1
tail -12 gpurun_out/r1/ab.txt
timeout 600 ./tools/ubench/lds_lookup_rate > gpurun_out/r1/lds.txt 2>&1
cat gpurun_out/r1/lds.txt
timeout 600 python tools/gqa_stream.py 80 8 > gpurun_out/r1/stream_new.txt 2>&1; tail -2 gpurun_out/r1/stream_new.txt
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 600 python tools/gqa_stream.py 80 8 > gpurun_out/r1/stream_old.txt 2>&1; tail -2 gpurun_out/r1/stream_old.txt
timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r1/tok_new.txt 2>&1; tail -1 gpurun_out/r1/tok_new.txt
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 600 python tools/dbg/tok70b.py 32 > gpurun_out/r1/tok_old.txt 2>&1; tail -1 gpurun_out/r1/tok_old.txt
timeout 1500 python -m pytest tests/test_gpu_block_engine_gqa.py -x -q -m gpu > gpurun_out/r1/pytest_gqa.txt 2>&1; tail -5 gpurun_out/r1/pytest_gqa.txt
