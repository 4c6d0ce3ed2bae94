cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r9
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new.pt 2 4 && QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 && python tools/dbg/gqa_ab.py --cmp /tmp/new.pt /tmp/old.pt ) > gpurun_out/r9/ab.txt 2>&1
tail -1 gpurun_out/r9/ab.txt
for i in 1 2; do
for v in ws wsnz wsnp ws16; do
echo "== $v $i"
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_$v.so timeout 600 python tools/gqa_waitstat.py 80 8 > gpurun_out/r9/waitstat_${v}_$i.txt 2>&1; tail -4 gpurun_out/r9/waitstat_${v}_$i.txt | cut -c1-330
done
done
for v in ws wsnz ws16 ws; do
echo "== tok $v"
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_$v.so timeout 600 python tools/dbg/tok70b.py 32 2>&1 | tail -1
done
