#!/bin/bash
# SURVEY 8d layer micro-bench: every distinct (out, in) of Llama-2-7B / 70B, bs=1 E8P12 GEMV, under rocprofv3:
# run 1 plain (HIP events), run 2 --kernel-trace (kernel-only duration), run 3 --pmc FETCH_SIZE (own pass) -> gpurun_out/<tag>_layer_microbench.txt
set -u
R=$GRAFT_REPO_ROOT
tag=${1:-r02}
out=$R/gpurun_out/layer_mb
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
res=$R/gpurun_out/${tag}_layer_microbench.txt
echo "# tools/layer_microbench.sh: bs=1 E8P12 GEMV per Llama shape; weights cycled through a pool > 256 MB; 200 launches per graph replay" > $res
echo "# kernel per shape: the dispatcher's choice (e8p_gemv_v2_kernel for k >= 8192, e8p_gemv_mfma_kernel below)" >> $res
echo "# columns: out in | algorithmic MB | HIP-event us/launch (incl. launch boundary) GB/s frac-of-8TB/s | rocprofv3 kernel avg us GB/s frac | FETCH_SIZE x2 MB / algorithmic" >> $res
for shape in "4096 4096" "11008 4096" "4096 11008" "8192 8192" "1024 8192" "28672 8192" "8192 28672"; do
  rm -rf $out/kt $out/pmc
  timeout 300 python $R/tools/layer_microbench.py $shape > $out/plain.log 2>&1      # HIP-event column: no profiler attached
  timeout 300 rocprofv3 --kernel-trace -d $out/kt -o kt -- python $R/tools/layer_microbench.py $shape > $out/kt.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc -o pmc -- python $R/tools/layer_microbench.py $shape > $out/pmc.log 2>&1
  python - "$shape" >> $res <<PY
import glob, json, sqlite3, sys
out = "$out"
line = [l for l in open(out + "/plain.log") if l.startswith("{")][-1]
j = json.loads(line)
dur = None
for p in glob.glob(out + "/kt/**/*.db", recursive=True):
    c = sqlite3.connect(p)
    r = c.execute("select count(*), avg(end-start) from kernels where name like '%e8p_gemv_%kernel%'").fetchone()
    if r and r[0]: dur = r[1] / 1e3
fetch = None
for p in glob.glob(out + "/pmc/**/*.db", recursive=True):
    c = sqlite3.connect(p)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    namecol = "kernel_name" if "kernel_name" in cols else "name"
    r = c.execute(f"select count(*), avg(value) from counters_collection where counter_name='FETCH_SIZE' and {namecol} like '%e8p_gemv_%kernel%'").fetchone()
    if r and r[0]: fetch = r[1] * 1024 * 2.0
a = j["algorithmic_bytes"]
k = "%6d %6d | %7.2f | %6.2f %7.1f %.3f" % (j["n"], j["k"], a / 1e6, j["us_per_launch_graph"], j["GBps"], j["frac_8TBps"])
if dur: k += " | %6.2f %7.1f %.3f" % (dur, a / dur / 1e3, a / dur / 1e3 / 8000)
if fetch: k += " | %7.2f %.3f" % (fetch / 1e6, fetch / a)
print(k)
PY
done
rm -rf $out
cat $res
