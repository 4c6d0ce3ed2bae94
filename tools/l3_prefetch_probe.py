#!/usr/bin/env python3
"""Does a weight matrix that was just read (and therefore sits in the 256 MB Infinity Cache) stream faster through the
bs = 1 GEMV than one that comes from HBM?  Per 70B shape: graph A = (touch W_i, GEMV W_i) x iters, graph B = the touches
alone, graph C = the GEMVs alone (cold: the pool is larger than the cache); GEMV-after-touch = (A - B) / iters.
usage: python tools/l3_prefetch_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa: E402
from quip_for_all_amd import capi  # noqa: E402

dev = "cuda:0"
L = capi.lib()
grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
iters = 60
for n, k in [(28672, 8192), (8192, 28672), (8192, 8192), (11008, 4096), (4096, 11008)]:
    wbytes = n * k // 4
    npool = max(4, (640 << 20) // wbytes + 1)
    g = torch.Generator(device=dev).manual_seed(0)
    pool = [torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32, device=dev).to(torch.int16) for _ in range(npool)]
    x = torch.randn(1, k, device=dev).half()
    planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
    capi.check(L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, st()), "planes")
    y = torch.empty(1, n, dtype=torch.float16, device=dev)
    ws = torch.zeros(max(L.quip_e8p_gemv_workspace_bytes(n) // 4, 1), dtype=torch.int32, device=dev)
    acc = torch.zeros((), dtype=torch.int64, device=dev)

    def gemv(i):
        capi.check(L.quip_e8p_gemv_planes_ws(planes.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(), y.data_ptr(), n, k,
                                             ws.data_ptr(), ws.numel() * 4, st()), "gemv")

    def touch(i):
        torch.sum(pool[i % npool].view(torch.int32).view(-1), (0,), dtype=torch.int64, out=acc)

    def timed(fn):
        for i in range(3):
            fn(i)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.cuda.graph(gr, stream=side):
            for i in range(iters):
                fn(i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            gr.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / iters)
        return sorted(ts)[1]

    tA = timed(lambda i: (touch(i), gemv(i)))
    tB = timed(touch)
    tC = timed(gemv)
    algo = wbytes + 2 * k + 2 * n
    print(f"{n}x{k}: GEMV from HBM {tC:.2f} us ({algo / tC / 1e6:.2f} TB/s) | touch {tB:.2f} us ({wbytes / tB / 1e6:.2f} TB/s) | "
          f"GEMV right after the touch {tA - tB:.2f} us ({algo / (tA - tB) / 1e6:.2f} TB/s)", flush=True)
    del pool
    torch.cuda.empty_cache()
