#!/bin/bash
# Profiles of the headline bench (run on the GPU box through gpurun):
#   1. rocprofv3 --kernel-trace --stats       -> gpurun_out/<tag>_bench_kernel_trace.txt
#   2. rocprofv3 --pmc FETCH_SIZE (own pass)   -> gpurun_out/<tag>_bench_fetch_size.txt, gemv_hbm_traffic.json
# usage: prof_bench.sh <tag> [bench args...]
set -u
R=$GRAFT_REPO_ROOT
tag=$1; shift
out=$R/gpurun_out/prof_bench_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline $*"
timeout 900 rocprofv3 --kernel-trace --stats -d $out/kt -o kt -- $CMD > $out/kt.log 2>&1
grep "^{\"metric\"" $out/kt.log | tail -1 | cut -c1-1400 > $R/gpurun_out/${tag}_bench_line.json
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc -o pmc -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $out/pmc.log 2>&1
python - <<PY
import glob, sqlite3, json
out, R, tag = "$out", "$R", "$tag"
with open(f"{R}/gpurun_out/{tag}_bench_kernel_trace.txt", "w") as f:
    print("# rocprofv3 --kernel-trace --stats -- $CMD", file=f)
    print("# bench line:", open(f"{R}/gpurun_out/{tag}_bench_line.json").read().strip(), file=f)
    for p in glob.glob(out + "/kt/**/*.db", recursive=True):
        c = sqlite3.connect(p)
        rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        print("# total kernel time %.3f ms over %d launches" % (tot / 1e6, sum(r[1] for r in rows)), file=f)
        print("%-100s %8s %12s %7s %10s %10s %10s" % ("kernel", "calls", "total_us", "pct", "avg_us", "min_us", "max_us"), file=f)
        for name, n, s, avg, mn, mx in rows[:40]:
            print("%-100s %8d %12.1f %6.2f%% %10.2f %10.2f %10.2f" % (name[:100], n, s / 1e3, 100.0 * s / tot, avg / 1e3, mn / 1e3, mx / 1e3), file=f)
with open(f"{R}/gpurun_out/{tag}_bench_fetch_size.txt", "w") as f:
    print("# rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline (own pass)", file=f)
    for p in glob.glob(out + "/pmc/**/*.db", recursive=True):
        c = sqlite3.connect(p)
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        namecol = "kernel_name" if "kernel_name" in ccols else "name"
        rows = c.execute(f"select {namecol}, count(*), sum(value), avg(value) from counters_collection where counter_name='FETCH_SIZE' group by {namecol} order by 3 desc").fetchall()
        print("%-100s %8s %16s %16s" % ("kernel", "dispatches", "sum FETCH_SIZE", "mean/dispatch"), file=f)
        gsum = gcnt = 0
        esum = ecnt = 0
        for name, n, s, avg in rows[:30]:
            print("%-100s %8d %16.1f %16.2f" % (name[:100], n, s, avg), file=f)
            if "e8p_gemv_mfma_kernel" in name or "e8p_gemv_v2_kernel" in name or "e8p_gemv_v2n_kernel" in name:
                gsum += s; gcnt += n
            if "decode_block_kernel" in name or "decode_block_gqa_kernel" in name:
                esum += s; ecnt += n
        if ecnt:
            raw = esum / ecnt
            j = {"counter": "FETCH_SIZE", "raw_mean_per_launch": raw, "unit_assumed": "KB", "gfx950_correction": 2.0,
                 "hbm_bytes_per_launch": raw * 1024 * 2.0, "dispatches": ecnt,
                 "kernels": "decode_block[_gqa]_kernel dispatches of the run (one per token: all blocks)",
                 "measured_at": "round 6 (%s), tools/prof_bench.sh %s" % (tag, "$*"),
                 "source": "rocprofv3 --pmc FETCH_SIZE pass of bench.py, tools/prof_bench.sh"}
            print("# ENGINE:", json.dumps(j), file=f)
            json.dump(j, open(f"{R}/gpurun_out/{tag}_engine_hbm_traffic.json", "w"))
        if gcnt:
            raw = gsum / gcnt
            # FETCH_SIZE is reported in KiB-ish units of 64-B requests; gfx950 tallies the 128-B requests of a
            # 16 B/lane streaming read at 64 B: double it (MI355X_MICROARCH.md, HBM section)
            j = {"counter": "FETCH_SIZE", "raw_mean_per_launch": raw, "unit_assumed": "KB", "gfx950_correction": 2.0,
                 "hbm_bytes_per_launch": raw * 1024 * 2.0, "gemv_dispatches": gcnt,
                 "kernels": "e8p_gemv_mfma_kernel + e8p_gemv_v2_kernel dispatches of the run",
                 "measured_at": "round 6 (%s), tools/prof_bench.sh %s" % (tag, "$*"),
                 "source": "rocprofv3 --pmc FETCH_SIZE pass of bench.py, tools/prof_bench.sh"}
            print("# GEMV:", json.dumps(j), file=f)
            json.dump(j, open(f"{R}/gpurun_out/{tag}_gemv_hbm_traffic.json", "w"))
PY
head -30 $R/gpurun_out/${tag}_bench_kernel_trace.txt | cut -c1-200
cat $R/gpurun_out/${tag}_bench_fetch_size.txt | cut -c1-200 | head -20
# the raw rocprofv3 databases are tens of MB: keep the summaries only (gpurun copies back <= 64 MiB)
rm -rf $out/kt $out/pmc
