#!/usr/bin/env python3
"""HBM streaming rate vs the shape of one load instruction (R rows x 1024/R bytes), same workgroup ->
row-range mapping and queue depth as the GEMV.  Graph-timed over a pool of matrices larger than L2+MALL."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
from quip_for_all_amd import capi  # noqa
dev = "cuda:0"
L = capi.lib()
grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
for (n, k) in [(28672, 8192), (8192, 8192), (8192, 28672)]:
    nmat = max(4, (700 << 20) // (n * k // 4))
    pool = [torch.randint(-32768, 32767, (n, k // 8), dtype=torch.int32, device=dev).to(torch.int16) for _ in range(nmat)]
    y = torch.zeros(16, dtype=torch.float16, device=dev)
    for waves in (8,):
        for R in (16, 8, 4, 2, 1):
            def run():
                st = torch.cuda.current_stream().cuda_stream
                for q in pool:
                    capi.check(L.quip_e8p_gemv_tuned(q.data_ptr(), q.data_ptr(), grid.data_ptr(), y.data_ptr(), n, k, 6, 0, R,
                                                     0, 0, waves, 3, None, st), "probe")
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
                run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) * 1e3 / len(pool))
            mb = n * k / 4 / 1e6
            print(f"N={n} K={k} waves={waves} R={R:2d}: {best:6.2f} us  {mb / best:.2f} TB/s", flush=True)
    del pool; torch.cuda.empty_cache()
