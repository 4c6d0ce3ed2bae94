cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r30
timeout 2400 python tools/engine_soak.py 1500 3 2>&1 | grep -v amdgpu | tee gpurun_out/r30/soak.txt
