cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r11
for sh in 7b g8; do
( timeout 300 python tools/dbg/gqa_ab.py /tmp/new_$sh.pt 3 4 $sh && QUIP_ENG_REP=24 timeout 300 python tools/dbg/gqa_ab.py /tmp/old_$sh.pt 3 4 $sh && python tools/dbg/gqa_ab.py --cmp /tmp/new_$sh.pt /tmp/old_$sh.pt ) > gpurun_out/r11/ab_$sh.txt 2>&1
echo "== $sh"; tail -8 gpurun_out/r11/ab_$sh.txt | grep -v amdgpu.ids
done
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
QUIP_ENG_REP=24 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
done
