cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r16
for v in default pre222 pre322; do
if [ $v = default ]; then unset QUIP_LIB_PATH; else export QUIP_LIB_PATH=$PWD/tools/dbg/libquip_$v.so; fi
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new_$v.pt 2 4 ) > gpurun_out/r16/ab_$v.txt 2>&1
done
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 > /dev/null 2>&1
for v in default pre222 pre322; do python tools/dbg/gqa_ab.py --cmp /tmp/new_$v.pt /tmp/old.pt | tail -1; done
unset QUIP_LIB_PATH
for i in 1 2; do
for v in default pre222 pre322 nopre; do
if [ $v = default ]; then unset QUIP_LIB_PATH; else export QUIP_LIB_PATH=$PWD/tools/dbg/libquip_$v.so; fi
echo -n "$v: "; timeout 600 python tools/dbg/tok70b.py 48 2>&1 | tail -1
done
done
unset QUIP_LIB_PATH
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_pre322.so python tools/gqa_stamps.py 16 8 40 > gpurun_out/r16/stamps322.txt 2>&1; grep "gathered\|products\|block span" gpurun_out/r16/stamps322.txt
