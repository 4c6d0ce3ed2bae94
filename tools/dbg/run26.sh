cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r26; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $R/tools/gemv_v2_bench.py --shapes 70b --variants "0,0,0,0,0,0" > $out/kt.log 2>&1
grep -v amdgpu $out/kt.log | tail -12
python - <<PY
import glob, sqlite3
for p in glob.glob("$out/kt/**/*.db", recursive=True):
    c = sqlite3.connect(p)
    rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 2 desc").fetchall()
    for name, n, avg, mn, mx in rows[:12]:
        print("%-90s %7d avg %8.2f us min %8.2f max %8.2f" % (name[:90], n, avg/1e3, mn/1e3, mx/1e3))
    # per-kernel by grid size: distinguish shapes
    try:
        rows = c.execute("select name, grid_size_x, grid_size_y, count(*), avg(end-start) from kernels where name like '%gemv%' group by name, grid_size_x, grid_size_y order by 4 desc").fetchall()
        for r in rows[:20]: print(r[0][:70], r[1], r[2], r[3], "%.2f us" % (r[4]/1e3))
    except Exception as e:
        print("cols:", [x[1] for x in c.execute("pragma table_info(kernels)")])
PY
find $out -name "*.db" -delete
