cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r15
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new.pt 2 4 && QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 && python tools/dbg/gqa_ab.py --cmp /tmp/new.pt /tmp/old.pt ) > gpurun_out/r15/ab.txt 2>&1
tail -1 gpurun_out/r15/ab.txt
for i in 1 2 3; do
timeout 600 python tools/dbg/tok70b.py 48 2>&1 | tail -1
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_nopre.so timeout 600 python tools/dbg/tok70b.py 48 2>&1 | tail -1
done
python tools/gqa_stamps.py 16 8 40 > gpurun_out/r15/stamps.txt 2>&1; grep -v amdgpu gpurun_out/r15/stamps.txt | tail -34
