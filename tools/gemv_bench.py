#!/usr/bin/env python3
"""Micro-benchmark of the E8P12 decode GEMV variants on one MI355X.

Weights rotate through a pool larger than the 256 MiB Infinity Cache so every
launch streams from HBM.  Timing: `iters` launches are captured in
one hipGraph (removes the ~5 us/launch host cost of eager ctypes launches) and
the replay is bracketed by HIP events on the replay stream; reported time is
per launch INCLUDING the inter-kernel boundary (~1.2-1.5 us on this chip), so
the GB/s printed here is a lower bound of the in-kernel rate (rocprofv3
kernel-trace gives the kernel-only duration).

variants: "kernel,rep,rows,blocks,waves_g,max_waves,digits;..."
  kernel 0 = integer GEMV on digit planes, 3 = integer GEMV converting fp16 x in-kernel,
  1 = fp16-domain GEMV, 2 = streaming-read probe (no decode)."""
import argparse
import json
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
}


def algo_bytes(n, k):
    return n * k // 4 + 2 * k + 2 * n     # Qidxs + x + y (SU/SV belong to the Hadamard kernels)


def bench(n, k, variants, iters, pool_bytes):
    dev = "cuda:0"
    wbytes = n * k // 4
    npool = max(2, min(64, pool_bytes // wbytes + 1))
    g = torch.Generator().manual_seed(0)
    pool = [torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev)
            for _ in range(npool)]
    x = torch.randn(1, k, generator=g).half().to(dev)
    y = torch.empty(1, n, dtype=torch.float16, device=dev)
    grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
    L = capi.lib()
    planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
    capi.check(L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, torch.cuda.current_stream().cuda_stream),
               "x_to_planes")
    planes_lo = torch.empty(3 * k + 16, dtype=torch.uint8, device=dev)
    capi.check(L.quip_e8p_x_to_planes_laneorder(x.data_ptr(), planes_lo.data_ptr(), k,
                                                 torch.cuda.current_stream().cuda_stream), "x_to_planes_lo")
    torch.cuda.synchronize()
    out = []
    for (kern, rep, rows, blocks, gg, maxw, dig) in variants:
        xin = planes if kern == 4 else (planes_lo if kern == 0 else x)

        def call(i):
            return L.quip_e8p_gemv_tuned(xin.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(),
                                         y.data_ptr(), n, k, kern, rep, rows, blocks, gg, maxw, dig, None,
                                         torch.cuda.current_stream().cuda_stream)
        rc = call(0)
        rec = dict(n=n, k=k, kernel=kern, rep=rep, rows=rows, blocks=blocks, g=gg, maxw=maxw, digits=dig)
        if rc != 0:
            rec["error"] = rc
            out.append(rec)
            print(json.dumps(rec), flush=True)
            continue
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
        ts = []
        for rep_i in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            graph.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / iters)
        us = float(np.median(ts[1:]))
        gbs = algo_bytes(n, k) / (us * 1e-6) / 1e9
        rec.update(us_graph=us, us_graph_min=float(min(ts)), GBps=float(gbs), frac_8TBps=float(gbs / 8000.0),
                   pool=npool)
        out.append(rec)
        print(json.dumps(rec), flush=True)
        del graph
    del pool
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--pool-mb", type=int, default=768)
    ap.add_argument("--json", default="")
    ap.add_argument("--variants", default="")
    a = ap.parse_args()
    shapes = SHAPES["7b"] + SHAPES["70b"] if a.shapes == "all" else SHAPES[a.shapes]
    if a.variants:
        variants = [tuple(int(v) for v in s.split(",")) for s in a.variants.split(";")]
    else:
        # (kernel, rep, rows, blocks, waves_g, max_waves, digits)
        variants = [(2, 0, 2, 0, 0, 16, 0),
                    (4, 0, 1, 0, 0, 16, 0), (4, 0, 2, 0, 0, 16, 0), (4, 0, 3, 0, 0, 16, 0), (4, 0, 4, 0, 0, 16, 0),
                    (4, 0, 2, 0, 0, 12, 0), (4, 0, 3, 0, 0, 12, 0), (4, 0, 2, 0, 0, 8, 0), (4, 0, 4, 0, 0, 8, 0),
                    (4, 0, 2, 512, 0, 8, 0), (4, 0, 2, 128, 0, 16, 0),
                    (0, 32, 4, 0, 0, 16, 3), (0, 1, 4, 512, 0, 8, 3), (3, 32, 2, 0, 0, 16, 3)]
    res = []
    for (n, k) in shapes:
        res += bench(n, k, variants, a.iters, a.pool_mb << 20)
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
