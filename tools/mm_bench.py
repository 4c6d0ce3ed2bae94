#!/usr/bin/env python3
"""Graph-timed skinny products per shape and M: the exact rows-mode matrix-core GEMV on digit planes
(e8p_gemv_planes_rows, planes precomputed: passes of up to 5 rows) next to the single-pass fp16 skinny kernel
(e8p_mm_skinny, x already fp16)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
dev = "cuda:0"
cb = Q.codebook.codebook_id["E8P12"](inference=True).to(dev)
for (n, k) in [(4096, 4096), (11008, 4096), (4096, 11008), (8192, 8192), (28672, 8192), (8192, 28672)]:
    nm = max(4, (400 << 20) // (n * k // 4))
    pool = [torch.randint(-32768, 32767, (n, k // 8), dtype=torch.int32, device=dev).to(torch.int16) for _ in range(nm)]
    from quip_for_all_amd.quant import get_hadK
    had, K, _ = get_hadK(k, True)
    had = None if had is None else had.to(dev).half().contiguous()
    for M in (1, 2, 4, 5, 8, 16, 31):
      for mode in ("rows", "skinny"):
        x = torch.randn(M, k, device=dev).half()
        planes = torch.ops.quip_lib.had_transform_planes_rows(x, k, K, had, True, None, 1.0, None, 1e-5, None)
        def run():
            for q in pool:
                if mode == "skinny":
                    torch.ops.quip_lib.e8p_mm_skinny(x, q, cb.grid_packed_abs)
                else:
                    torch.ops.quip_lib.e8p_gemv_planes_rows(planes, q, cb.grid_packed_abs)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / len(pool))
        print(f"N={n} K={k} M={M:2d} {mode:6s}: {best:7.2f} us  ({n * k / 4 / 1e6 / best:.2f} TB/s of codes)", flush=True)
    del pool; torch.cuda.empty_cache()
