#!/usr/bin/env python3
"""SURVEY 8d layer micro-bench, one shape per process (run under rocprofv3 by tools/layer_microbench.sh):
the default bs=1 E8P12 GEMV launch (quip_e8p_gemv_planes_ws: the product's entry point, which picks the kernel by
shape) on ONE (out, in) shape, weights cycled through a pool
larger than the 256 MB Infinity Cache, 200 launches per hipGraph replay, HIP-event timed.
usage: layer_microbench.py N K"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
from quip_for_all_amd import capi
n, k = int(sys.argv[1]), int(sys.argv[2])
dev = "cuda:0"
L = capi.lib()
wbytes = n * k // 4
npool = max(4, (640 << 20) // wbytes + 1)
g = torch.Generator().manual_seed(0)
pool = [torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev) for _ in range(npool)]
x = torch.randn(1, k, generator=g).half().to(dev)
grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream   # noqa: E731
capi.check(L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, st()), "planes")
y = torch.empty(1, n, dtype=torch.float16, device=dev)
ws = torch.zeros(L.quip_e8p_gemv_workspace_bytes(n) // 4, dtype=torch.int32, device=dev)   # K-split scratch (stays zero)
call = lambda i: capi.check(L.quip_e8p_gemv_planes_ws(planes.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(), y.data_ptr(), n, k, ws.data_ptr(), ws.numel() * 4, st()), "gemv")  # noqa: E731
for i in range(3):
    call(i)
torch.cuda.synchronize()
iters = 200
gr = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.cuda.graph(gr, stream=side):
    for i in range(iters):
        call(i)
torch.cuda.synchronize()
ts = []
for _ in range(4):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3 / iters)
algo = n * k // 4 + 2 * k + 2 * n
us = sorted(ts)[1]
print(json.dumps({"n": n, "k": k, "pool": npool, "algorithmic_bytes": algo, "us_per_launch_graph": round(us, 3),
                  "GBps": round(algo / us / 1e3, 1), "frac_8TBps": round(algo / us / 1e3 / 8000, 4)}))
