#!/bin/bash
# kernel trace of the 70B-shaped decode step (run through gpurun): per-kernel calls / avg duration
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/trace70b
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $R/tools/dbg/tok70b.py 16 > $out/kt.log 2>&1
tail -2 $out/kt.log
python - <<PY
import glob, sqlite3
for p in glob.glob("$out/kt/**/*.db", recursive=True):
    c = sqlite3.connect(p)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("total kernel time %.3f ms over %d launches" % (tot / 1e6, sum(r[1] for r in rows)))
    for name, n, s, avg, mn in rows[:16]:
        print("%-110s %6d %10.1f %6.2f%% avg %8.2f min %8.2f" % (name[:110], n, s / 1e3, 100.0 * s / tot, avg / 1e3, mn / 1e3))
PY
find $out -name "*.db" -delete
