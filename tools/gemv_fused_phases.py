#!/usr/bin/env python3
"""Phase timeline (s_memtime stamps) of the fused-prologue GEMV at the 7B shapes."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
from quip_for_all_amd import capi  # noqa
dev = "cuda:0"
L = capi.lib()
grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
names = ["issue loads", "tables+zero", "prologue", "lane consts", "main loop", "barrier", "epilogue"]
for ns, k, zmode, rms in [((4096,), 4096, False, False), ((4096, 4096, 4096), 4096, False, True), ((4096, 4096, 4096), 4096, True, True),
                          ((11008, 11008), 4096, True, True)]:
    g = torch.Generator().manual_seed(0)
    qs = [torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev) for n in ns]
    v = lambda: torch.randn(1, k, generator=g).half().to(dev)
    x, z, post, res, w = v(), v(), v(), v(), v()
    pres = [v() for _ in ns]
    ys = [torch.empty(1, n, dtype=torch.float16, device=dev) for n in ns]
    hout = torch.empty(1, k, dtype=torch.float16, device=dev)
    fin = capi.GemvFusedIn()
    fin.x = x.data_ptr(); fin.z = z.data_ptr() if zmode else None; fin.post_scale = post.data_ptr(); fin.residual = res.data_ptr()
    fin.h_out = hout.data_ptr(); fin.rms_weight = w.data_ptr() if rms else None
    for i, p in enumerate(pres):
        fin.pre_scale[i] = p.data_ptr(); fin.scale[i] = 0.01
    fin.z_scale = 0.015; fin.rms_eps = 1e-5
    cnt = len(ns)
    vp = ctypes.c_void_p * cnt
    nsa = (ctypes.c_int32 * cnt)(*ns)
    dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
    for i in range(3):
        capi.check(L.quip_e8p_gemv_fused_tuned(ctypes.byref(fin), vp(*[q.data_ptr() for q in qs]), grid.data_ptr(),
                                               vp(*[y.data_ptr() for y in ys]), nsa, cnt, k, dbg.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream), "fused")
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().reshape(-1, 8)
    d2 = d[2048:2048 + 256].astype(np.int64)
    d2 = d2[d2[:, 0] != 0]
    print("   prologue detail: wait %d | Z-stage %d | rms+mul %d | fht(g0) %d | absmax-reduce %d | planes->LDS %d" % tuple(
        [np.median(d2[:, 0] - d[:len(d2), 2])] + [np.median(d2[:, i + 1] - d2[:, i]) for i in range(5)]))
    d = d[:2048]
    d = d[d[:, 0] != 0].astype(np.int64)
    ph = np.diff(d, axis=1)
    print(f"ns={ns} k={k} z={zmode} rms={rms}: {len(d)} WGs  total median {np.median(d[:,7]-d[:,0]):.0f} ticks")
    print("   " + "  ".join("%s %d" % (nm, np.median(ph[:, i])) for i, nm in enumerate(names)))
