#!/usr/bin/env python3
"""Phase timeline (in-kernel s_memtime stamps) and graph timing of grouped / single decode GEMVs at the
Llama-2-7B / 70B shapes.  usage: gemv_group_phases.py [rep rows max_waves]"""
import ctypes
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
rep, rows, maxw = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (0, 0, 0)))
NSETS = int(os.environ.get("NSETS", "24"))
HOT = os.environ.get("HOT", "0") == "1"    # every launch on the same weights (L2 / MALL resident)


def run_case(ns, k):
    g = torch.Generator().manual_seed(0)
    sets = []
    for _ in range(NSETS):
        qs = [torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev) for n in ns]
        sets.append(qs)
    planes = []
    for _ in ns:
        x = torch.randn(1, k, generator=g).half().to(dev)
        pl = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
        L.quip_e8p_x_to_planes(x.data_ptr(), pl.data_ptr(), k, torch.cuda.current_stream().cuda_stream)
        planes.append(pl)
    ys = [torch.empty(1, n, dtype=torch.float16, device=dev) for n in ns]
    cnt = len(ns)
    vp = ctypes.c_void_p * cnt
    nsa = (ctypes.c_int32 * cnt)(*ns)

    def launch(i, dbg):
        st = torch.cuda.current_stream().cuda_stream
        capi.check(L.quip_e8p_gemv_group_tuned(vp(*[p.data_ptr() for p in planes]), vp(*[q.data_ptr() for q in sets[0 if HOT else i % NSETS]]),
                                               grid.data_ptr(), vp(*[y.data_ptr() for y in ys]), nsa, cnt, k, rep, rows, 0,
                                               maxw, dbg, st), "group")
    dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    for i in range(3):
        launch(i, dbg.data_ptr())
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8)
    d = d[d[:, 0] != 0].astype(np.int64)
    rel = d - d[:, 0].min()
    ph = np.diff(d, axis=1)
    # graph timing without stamps
    launch(0, None)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(gr, stream=s):
        for i in range(NSETS):
            launch(i, None)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / NSETS)
    mb = sum(n * k // 4 for n in ns) / 1e6
    print(f"ns={ns} k={k}: {mb:.1f} MB  {best:.2f} us/launch = {mb / best:.2f} TB/s ; {len(d)} WGs")
    print("   WG start skew: median %d max %d ; WG end: median %d max %d (ticks ~ 100 MHz? see s_memtime)" %
          (np.median(rel[:, 0]), rel[:, 0].max(), np.median(rel[:, 7]), rel[:, 7].max()))
    print("   " + "  ".join("%s %d/%d" % (nm, np.median(ph[:, i]), np.percentile(ph[:, i], 90)) for i, nm in enumerate(names)))


for ns, k in [((4096,), 4096), ((4096, 4096, 4096), 4096), ((11008, 11008), 4096), ((4096,), 11008),
              ((8192,), 8192), ((8192, 1024, 1024), 8192), ((28672, 28672), 8192), ((8192,), 28672)]:
    run_case(list(ns), k)
    torch.cuda.empty_cache()
