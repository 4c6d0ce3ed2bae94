cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r14
timeout 1500 python -m pytest tests/test_gpu_decode.py tests/test_gpu_block_engine_gqa.py tests/test_gpu_block_engine.py -x -q -m gpu -s --durations=8 -k "thirty_two or twelve or spiky or measurement or tile" > gpurun_out/r14/deep.txt 2>&1; grep -v amdgpu.ids gpurun_out/r14/deep.txt | tail -25
timeout 1500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r14/bench.txt 2>&1; tail -1 gpurun_out/r14/bench.txt > gpurun_out/r14/bench_line.json; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r14/bench_line.json').read())
print(d["value"], d["roofline"], d["parity"]["max_ulps"], d["parity"]["bound_ulps"])
e=d["extras"]
g=e["llama2_70b_e8p12"]
print("70b", g["tokens_per_s"], g.get("token_roofline_frac"), g.get("gemv_stream_in_launch"))
print("per_shape", g.get("per_shape"))
for k in e:
    if k.startswith("llama2_7b") or k.startswith("llama3"):
        print(k, e[k].get("tokens_per_s"))
h=e["hf_generate_static_cache"]
print({k:v for k,v in h.items() if "teacher" in k or "error" in k})
print({k:v for k,v in h.items() if "tokens_per_s" in k})
PY
