#!/usr/bin/env python3
"""A/B of the two matrix-core E8P12 GEMVs (e8p_gemv_mfma.hip vs e8p_gemv_v2.hip) on one MI355X:
bit identity first, then time per launch (hipGraph of `iters` launches over a weight pool larger than the
Infinity Cache, HIP events on the replay stream; includes the launch boundary).

variants of v2: "rep,slots,blocks,ksplit,max_waves,runlen;..." (0 = auto; rep 32 / 24 / 16 = (32,32) / (32,16) / (16,16)
table copies).  --groups times the grouped launches of a decoder block (q/k/v, gate/up) against the first kernel's.  --phases prints the in-kernel s_memtime
stamps (median / p90 over workgroups, shader-clock ticks)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa: E402
from quip_for_all_amd import capi  # noqa: E402

SHAPES = {
    "7b": [(4096, 4096), (11008, 4096), (4096, 11008)],
    "70b": [(8192, 8192), (1024, 8192), (28672, 8192), (8192, 28672)],
    "odd": [(100, 128), (4100, 1152), (4096, 2048), (777, 11008), (16, 28672)],
}


def algo_bytes(n, k):
    return n * k // 4 + 2 * k + 2 * n


def time_graph(call, iters):
    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for i in range(iters):
                call(i)
    torch.cuda.synchronize()
    import time as _t
    t0 = _t.perf_counter()
    while (_t.perf_counter() - t0) * 1e3 < WARM_MS:      # (an idle device starts at ~1.5 GHz: --warm-ms of load before the timed replays)
        graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        graph.replay()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / iters)
    del graph
    return float(np.median(ts[1:]))


WARM_MS = 0.0
PHASES = ["issue loads", "tables", "digits->LDS+bar", "stream (wave 0)", "barrier", "epilogue"]


def run(n, k, variants, iters, pool_bytes, check_only=False, phases=False):
    dev = "cuda:0"
    L = capi.lib()
    wbytes = n * k // 4
    npool = max(2, min(64, pool_bytes // max(wbytes, 1) + 1))
    if check_only:
        npool = 2
    g = torch.Generator().manual_seed(n * 131 + k)
    pool = [torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev)
            for _ in range(npool)]
    x = torch.randn(1, k, generator=g).half().to(dev)
    grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
    planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    capi.check(L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, st()), "x_to_planes")
    ws = torch.zeros(L.quip_e8p_gemv_v2_workspace_bytes(n) // 4, dtype=torch.int32, device=dev)
    y1 = torch.empty(1, n, dtype=torch.float16, device=dev)
    y2 = torch.empty(1, n, dtype=torch.float16, device=dev)

    def call_v1(i):
        return L.quip_e8p_gemv_tuned(planes.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(), y1.data_ptr(),
                                     n, k, 4, 0, 0, 0, 0, 0, 0, None, st())
    rc1 = call_v1(0)
    ok_ref = rc1 == 0
    line = f"N={n:6d} K={k:6d} {algo_bytes(n, k) / 1e6:7.2f} MB |"
    if ok_ref and not check_only:
        us = time_graph(call_v1, iters)
        line += f" v1 {us:7.2f} us {algo_bytes(n, k) / us / 1e6:5.2f} TB/s {algo_bytes(n, k) / us / 8e6:5.3f} |"
    print(line, flush=True)
    for (rep2, slots, blocks, ksplit, maxw, flags) in variants:
        def call_v2(i):
            return L.quip_e8p_gemv_v2_tuned(planes.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(),
                                            y2.data_ptr(), ws.data_ptr(), n, k, rep2, slots, blocks, ksplit, maxw,
                                            flags, None, st())
        y2.fill_(float("nan"))
        rc = call_v2(0)
        tag = f"   v2 rep={rep2:2d} slots={slots} blocks={blocks:3d} ksplit={ksplit} waves={maxw:2d} runlen={flags}:"
        if rc != 0:
            print(tag, "rc", rc, flush=True)
            continue
        torch.cuda.synchronize()
        msg = ""
        if ok_ref:
            call_v1(0)
            torch.cuda.synchronize()
            same = torch.equal(y1.view(torch.int16), y2.view(torch.int16))
            msg = "bit-identical" if same else f"MISMATCH ({int((y1.view(torch.int16) != y2.view(torch.int16)).sum())} of {n})"
        zero = int(ws.abs().sum().item()) == 0
        if not zero:
            msg += " WORKSPACE-NOT-ZERO"
            ws.zero_()
        if check_only:
            print(tag, msg, flush=True)
            continue
        if phases:
            dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
            for i in range(3):
                dbg.zero_()
                L.quip_e8p_gemv_v2_tuned(planes.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(),
                                         y2.data_ptr(), ws.data_ptr(), n, k, rep2, slots, blocks, ksplit, maxw,
                                         flags, dbg.data_ptr(), st())
                torch.cuda.synchronize()
            d = dbg.cpu().numpy().reshape(-1, 8)
            d2 = d[2048:2048 + 4096 // 2][d[:2048, 0] != 0].astype(np.int64)
            d = d[:2048][d[:2048, 0] != 0].astype(np.int64)
            if d2[:, 0].any():
                print("      kernel start -> arguments in SGPRs %d/%d, -> digit requests issued %d/%d" % (
                    np.median(d2[:, 0] - d[:, 0]), np.percentile(d2[:, 0] - d[:, 0], 90),
                    np.median(d2[:, 1] - d[:, 0]), np.percentile(d2[:, 1] - d[:, 0], 90)))
            ph = np.diff(d[:, :7], axis=1)
            t0 = d[:, 0].min()
            print("      " + "  ".join("%s %d/%d" % (nm, np.median(ph[:, i]), np.percentile(ph[:, i], 90))
                                       for i, nm in enumerate(PHASES)))
            print("      WGs %d  start skew med/max %d/%d  last wave leaves stream (rel. to wave 0) med/max %d/%d  "
                  "kernel span %d ticks" % (len(d), np.median(d[:, 0] - t0), (d[:, 0] - t0).max(),
                                            np.median(d[:, 7] - d[:, 4]), (d[:, 7] - d[:, 4]).max(),
                                            d[:, 6].max() - t0))
        us = time_graph(call_v2, iters)
        print(tag, f"{us:7.2f} us {algo_bytes(n, k) / us / 1e6:5.2f} TB/s {algo_bytes(n, k) / us / 8e6:5.3f}  {msg}",
              flush=True)
    del pool
    torch.cuda.empty_cache()


GROUPS = {"7b": [((4096, 4096, 4096), 4096), ((11008, 11008), 4096)],
          "70b": [((8192, 1024, 1024), 8192), ((28672, 28672), 8192)]}


def run_group(ns, k, variants, iters, pool_bytes):
    import ctypes
    dev = "cuda:0"
    L = capi.lib()
    cnt = len(ns)
    wbytes = sum(n * k // 4 for n in ns)
    npool = max(2, min(32, pool_bytes // wbytes + 1))
    g = torch.Generator().manual_seed(7)
    pool = [[torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev)
             for n in ns] for _ in range(npool)]
    grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
    st = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731
    planes = []
    for _ in ns:
        x = torch.randn(1, k, generator=g).half().to(dev)
        pl = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
        capi.check(L.quip_e8p_x_to_planes(x.data_ptr(), pl.data_ptr(), k, st()), "x_to_planes")
        planes.append(pl)
    y1 = [torch.empty(1, n, dtype=torch.float16, device=dev) for n in ns]
    y2 = [torch.empty(1, n, dtype=torch.float16, device=dev) for n in ns]
    ws = torch.zeros(sum(L.quip_e8p_gemv_v2_workspace_bytes(n) for n in ns) // 4, dtype=torch.int32, device=dev)
    vp = ctypes.c_void_p * cnt
    nsa = (ctypes.c_int32 * cnt)(*ns)
    mb = sum(algo_bytes(n, k) for n in ns)

    def call_v1(i):
        return L.quip_e8p_gemv_group_tuned(vp(*[p.data_ptr() for p in planes]), vp(*[q.data_ptr() for q in pool[i % npool]]),
                                           grid.data_ptr(), vp(*[y.data_ptr() for y in y1]), nsa, cnt, k, 0, 0, 0, 0, None, st())
    assert call_v1(0) == 0
    us = time_graph(call_v1, iters)
    print(f"group N={ns} K={k} {mb / 1e6:7.2f} MB | v1 {us:7.2f} us {mb / us / 1e6:5.2f} TB/s {mb / us / 8e6:5.3f} |", flush=True)
    for (rep, slots, blocks, ksplit, maxw, runlen) in variants:
        def call_v2(i):
            return L.quip_e8p_gemv_v2_group_tuned(vp(*[p.data_ptr() for p in planes]), vp(*[q.data_ptr() for q in pool[i % npool]]),
                                                  grid.data_ptr(), vp(*[y.data_ptr() for y in y2]), ws.data_ptr(), nsa, cnt, k,
                                                  rep, slots, blocks, ksplit, maxw, runlen, None, st())
        for y in y2:
            y.fill_(float("nan"))
        rc = call_v2(0)
        tag = f"   v2 rep={rep:2d} slots={slots} blocks={blocks:3d} ksplit={ksplit} waves={maxw:2d} runlen={runlen}:"
        if rc != 0:
            print(tag, "rc", rc, flush=True)
            continue
        call_v1(0)
        torch.cuda.synchronize()
        same = all(torch.equal(a.view(torch.int16), b.view(torch.int16)) for a, b in zip(y1, y2))
        msg = "bit-identical" if same else "MISMATCH"
        if int(ws.abs().sum().item()) != 0:
            msg += " WORKSPACE-NOT-ZERO"
            ws.zero_()
        us = time_graph(call_v2, iters)
        print(tag, f"{us:7.2f} us {mb / us / 1e6:5.2f} TB/s {mb / us / 8e6:5.3f}  {msg}", flush=True)
    del pool
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="70b")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--pool-mb", type=int, default=768)
    ap.add_argument("--variants", default="0,0,0,0,0,0")
    ap.add_argument("--phases", action="store_true")
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--groups", action="store_true")
    ap.add_argument("--warm-ms", type=float, default=0.0)
    a = ap.parse_args()
    global WARM_MS
    WARM_MS = a.warm_ms
    shapes = []
    for s in a.shapes.split(","):
        shapes += SHAPES.get(s, []) if a.groups else SHAPES[s]
    variants = [tuple(int(v) for v in s.split(",")) for s in a.variants.split(";")]
    if a.groups:
        for s in a.shapes.split(","):
            for ns, k in GROUPS.get(s, []):
                run_group(list(ns), k, variants, a.iters, a.pool_mb << 20)
        return
    for (n, k) in shapes:
        run(n, k, variants, a.iters, a.pool_mb << 20, a.check_only, a.phases)


if __name__ == "__main__":
    main()
