cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r12
for sh in 7b g8; do
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new_$sh.pt 3 4 $sh && QUIP_ENG_REP=24 timeout 300 python tools/dbg/gqa_ab.py /tmp/old_$sh.pt 3 4 $sh && python tools/dbg/gqa_ab.py --cmp /tmp/new_$sh.pt /tmp/old_$sh.pt ) > gpurun_out/r12/ab_$sh.txt 2>&1
echo "== $sh"; tail -1 gpurun_out/r12/ab_$sh.txt
done
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
QUIP_ENG_REP=24 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
done
python tools/block_stamps.py 32 16 100 > gpurun_out/r12/stamps_nib.txt 2>&1; tail -32 gpurun_out/r12/stamps_nib.txt
QUIP_ENG_REP=24 python tools/block_stamps.py 32 16 100 > gpurun_out/r12/stamps_24.txt 2>&1; tail -32 gpurun_out/r12/stamps_24.txt
