#!/usr/bin/env python3
"""Row counts between the skinny regime and prefill: the single-pass skinny kernel on chunks of 32 rows (one launch,
grid.y = chunks) against the 256 x 256-tile fused GEMM, per 7B shape, graph-timed."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
dev = "cuda:0"
cb = Q.codebook.codebook_id["E8P12"](inference=True).to(dev)
for (n, k) in [(4096, 4096), (11008, 4096), (4096, 11008), (8192, 8192)]:
    pool = [torch.randint(-32768, 32767, (n, k // 8), dtype=torch.int32, device=dev).to(torch.int16) for _ in range(8)]
    for M in (32, 48, 64, 128, 256, 512, 1024, 2048, 4096):
        x = torch.randn(M, k, device=dev).half()
        res = {}
        for mode in ("skinny", "batched"):
            def run():
                for q in pool:
                    if mode == "skinny":
                        torch.ops.quip_lib.e8p_mm_skinny(x, q, cb.grid_packed_abs)
                    else:
                        torch.ops.quip_lib.e8p_mm_batched(x, q, cb.grid_packed_abs)
            run(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
                run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record(); torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) * 1e3 / len(pool))
            res[mode] = best
        ys = torch.ops.quip_lib.e8p_mm_skinny(x, pool[0], cb.grid_packed_abs).float()
        yb = torch.ops.quip_lib.e8p_mm_batched(x, pool[0], cb.grid_packed_abs).float()
        print(f"N={n} K={k} M={M:5d}: skinny {res['skinny']:8.1f} us | batched {res['batched']:8.1f} us | "
              f"max |diff| {float((ys - yb).abs().max()):.4f} (|y| max {float(yb.abs().max()):.1f})", flush=True)
