#!/usr/bin/env python3
"""Micro-benchmark of the E8P12 decode GEMV variants on one MI355X.
Weights rotate through a pool larger than the 256 MiB Infinity Cache so every
launch streams from HBM.  Timing: HIP events on the launch stream.
Usage: python tools/gemv_bench.py [--shapes 70b|7b|all] [--json out.json]"""
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
    st = torch.cuda.current_stream().cuda_stream
    out = []
    for (rep, rows, blocks, gg) in variants:
        def call(i):
            return L.quip_e8p_gemv_tuned(x.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(),
                                         y.data_ptr(), n, k, rep, rows, blocks, gg, st)
        rc = call(0)
        if rc != 0:
            out.append(dict(n=n, k=k, rep=rep, rows=rows, blocks=blocks, g=gg, error=rc))
            continue
        for i in range(5):
            call(i)
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i, (a, b) in enumerate(evs):
            a.record()
            call(i)
            b.record()
        torch.cuda.synchronize()
        ts = np.array([a.elapsed_time(b) * 1e3 for a, b in evs])   # us
        # back-to-back launches: total time / iters (hides per-launch event overhead)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(iters):
            call(i)
        b.record()
        torch.cuda.synchronize()
        b2b = a.elapsed_time(b) * 1e3 / iters
        gbs = algo_bytes(n, k) / (b2b * 1e-6) / 1e9
        out.append(dict(n=n, k=k, rep=rep, rows=rows, blocks=blocks, g=gg, us_median=float(np.median(ts)),
                        us_min=float(ts.min()), us_b2b=float(b2b), GBps_b2b=float(gbs),
                        frac_8TBps=float(gbs / 8000.0), pool=npool))
        print(json.dumps(out[-1]), flush=True)
    del pool
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="all")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--pool-mb", type=int, default=768)
    ap.add_argument("--json", default="")
    ap.add_argument("--variants", default="")
    a = ap.parse_args()
    shapes = SHAPES["7b"] + SHAPES["70b"] if a.shapes == "all" else SHAPES[a.shapes]
    if a.variants:
        variants = [tuple(int(v) for v in s.split(",")) for s in a.variants.split(";")]
    else:
        variants = [(0, 0, 0, 0), (16, 4, 0, 0), (16, 2, 0, 0), (16, 1, 0, 0), (1, 4, 0, 0), (1, 2, 0, 0),
                    (16, 4, 0, 4), (16, 4, 0, 8), (16, 2, 512, 0), (1, 4, 512, 0), (1, 4, 1024, 4),
                    (1, 2, 1024, 4), (1, 4, 2048, 4)]
    res = []
    for (n, k) in shapes:
        res += bench(n, k, variants, a.iters, a.pool_mb << 20)
    if a.json:
        os.makedirs(os.path.dirname(a.json) or ".", exist_ok=True)
        with open(a.json, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
