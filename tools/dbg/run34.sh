cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python - <<'PY'
import sys, json, torch
sys.path.insert(0, '.')
import bench
out = bench.gemv_per_shape([(28672, 8192), (8192, 28672), ((28672, 28672), 8192), ((8192, 1024, 1024), 8192)], "cuda:0")
print(json.dumps(out, indent=1))
PY
