cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python tools/dbg/tok_shape.py 5120 13824 40 40 40 2>&1 | grep -v amdgpu | tail -3
QUIP_BLOCK_ENGINE=0 timeout 600 python tools/dbg/tok_shape.py 4096 11008 32 32 32 2>&1 | grep -v amdgpu | tail -2
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt13 -o kt -- python $GRAFT_REPO_ROOT/tools/dbg/tok_shape.py 5120 13824 8 40 40 16 > /tmp/kt13.log 2>&1
python - <<PY
import glob, sqlite3
for p in glob.glob("/tmp/kt13/**/*.db", recursive=True):
    c = sqlite3.connect(p)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    for name, n, s, avg in rows[:16]:
        print("%-110s %6d %9.1f us %5.1f%% avg %7.2f" % (name[:110], n, s/1e3, 100*s/tot, avg/1e3))
PY
