cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r7
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 > gpurun_out/r7/ab.txt 2>&1
for v in nopipe noasm; do
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_$v.so timeout 300 python tools/dbg/gqa_ab.py /tmp/$v.pt 2 4 >> gpurun_out/r7/ab.txt 2>&1
echo "== $v" >> gpurun_out/r7/ab.txt
python tools/dbg/gqa_ab.py --cmp /tmp/$v.pt /tmp/old.pt >> gpurun_out/r7/ab.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/r7/ab.txt
