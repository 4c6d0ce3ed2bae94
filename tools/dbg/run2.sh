set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new.pt 2 4 && QUIP_LIB_PATH=$PWD/tools/dbg/libquip_rep16.so timeout 300 python tools/dbg/gqa_ab.py /tmp/old.pt 2 4 && python tools/dbg/gqa_ab.py --cmp /tmp/new.pt /tmp/old.pt ) > gpurun_out/r2/ab.txt 2>&1
tail -8 gpurun_out/r2/ab.txt
for v in ws ws16 wsnd; do
QUIP_LIB_PATH=$PWD/tools/dbg/libquip_$v.so timeout 600 python tools/gqa_waitstat.py 80 6 > gpurun_out/r2/waitstat_$v.txt 2>&1; tail -8 gpurun_out/r2/waitstat_$v.txt
done
timeout 600 ./tools/ubench/lds_lookup_rate > gpurun_out/r2/lds.txt 2>&1
grep b128 gpurun_out/r2/lds.txt
