#!/bin/bash
# Kernel-trace profile of the headline bench (run on the GPU box through gpurun).
# usage: prof_bench.sh <tag> [bench args...]   -> gpurun_out/bench_<tag>_kernels.txt
set -u
R=$GRAFT_REPO_ROOT
tag=$1; shift
out=$R/gpurun_out/prof_bench_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline "$@" > $out/kt.log 2>&1
tail -1 $out/kt.log | cut -c1-600
python - <<PY > $R/gpurun_out/bench_${tag}_kernels.txt
import glob, sqlite3
dbs = glob.glob("$out/kt/**/*.db", recursive=True)
print("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 16 --warmup 4 --no-cpu-baseline $*")
for p in dbs:
    c = sqlite3.connect(p)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# total kernel time %.3f ms over %d launches" % (tot / 1e6, sum(r[1] for r in rows)))
    print("%-90s %8s %12s %7s %10s %10s %10s" % ("kernel", "calls", "total_us", "pct", "avg_us", "min_us", "max_us"))
    for name, n, s, avg, mn, mx in rows[:60]:
        print("%-90s %8d %12.1f %6.2f%% %10.2f %10.2f %10.2f" % (name[:90], n, s / 1e3, 100.0 * s / tot, avg / 1e3, mn / 1e3, mx / 1e3))
PY
head -40 $R/gpurun_out/bench_${tag}_kernels.txt
