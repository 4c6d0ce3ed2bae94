#!/usr/bin/env python3
"""Phase timeline of the integer-domain GEMV from in-kernel s_memtime stamps
(micro-benchmark aid).  Prints per-phase medians over workgroups in shader-clock
cycles converted with the 100 MHz wall clock ratio measured around the launch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa: E402
from quip_for_all_amd import capi  # noqa: E402

dev = "cuda:0"
L = capi.lib()
grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
names = ["issue loads", "tables+zero", "planes->LDS+bar", "lane consts", "main loop", "barrier", "epilogue"]
for (n, k) in [(28672, 8192), (8192, 28672), (8192, 8192), (4096, 4096)]:
    g = torch.Generator().manual_seed(0)
    pool = [torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev)
            for _ in range(max(2, (600 << 20) // (n * k // 4)))]
    x = torch.randn(1, k, generator=g).half().to(dev)
    y = torch.empty(1, n, dtype=torch.float16, device=dev)
    planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
    L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, torch.cuda.current_stream().cuda_stream)
    for (rep, rows, maxw) in [(0, 2, 16), (0, 1, 16), (0, 2, 8), (0, 1, 8)]:
        dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for i in range(4):
            L.quip_e8p_gemv_tuned(planes.data_ptr(), pool[i % len(pool)].data_ptr(), grid.data_ptr(), y.data_ptr(),
                                  n, k, 4, rep, rows, 0, 0, maxw, 3, dbg.data_ptr(), st)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy().reshape(-1, 8)
        d = d[d[:, 0] != 0].astype(np.int64)
        t0 = d[:, 0].min()
        rel = d - t0
        ph = np.diff(d, axis=1)
        print(f"N={n} K={k} rep={rep} rows={rows} maxw={maxw}: {len(d)} WGs; ticks (s_memtime units)")
        print("   WG start skew: median %d  max %d ; WG end: median %d max %d" %
              (np.median(rel[:, 0]), rel[:, 0].max(), np.median(rel[:, 7]), rel[:, 7].max()))
        for i, nm in enumerate(names):
            print("   %-14s median %7d  p90 %7d" % (nm, np.median(ph[:, i]), np.percentile(ph[:, i], 90)))
    del pool
    torch.cuda.empty_cache()
