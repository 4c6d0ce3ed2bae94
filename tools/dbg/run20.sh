cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r20
for kv in 0 1; do
echo "== HIP_FORCE_DEV_KERNARG=$kv"
HIP_FORCE_DEV_KERNARG=$kv timeout 600 python tools/gemv_v2_bench.py --shapes 70b,7b --variants "0,0,0,0,0,0" --phases 2>&1 | grep -v "amdgpu.ids\|WGs"
done | tee gpurun_out/r20/kernarg.txt
