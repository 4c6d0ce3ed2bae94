#!/usr/bin/env python3
"""Measured error of QuantLinear.forward (HIP path) against the float64 oracle, in fp16 ulps of the output scale,
for every BASELINE config shape and codebook: prints max and 99.9th percentile of |y - y64| / ulp(max(|y64|, rms)).
The module-level parity bound of the tests (oracle.parity_bound) is derived from these numbers."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import quip_oracle as O  # noqa
from quip_for_all_amd.qlinear import QuantLinear  # noqa

CASES = [("E8P12", 4096, 4096), ("E8P12", 4096, 11008), ("E8P12", 11008, 4096), ("E8P12", 8192, 8192),
         ("E8P12", 8192, 1024), ("E8P12", 8192, 28672), ("E8P12", 28672, 8192),
         ("E8P12RVQ4B", 4096, 4096), ("E8P12RVQ3B", 4096, 4096), ("D4", 4096, 4096), ("HI", 4096, 4096)]
for cbid, fin, fout in CASES:
    P = O.make_layer(cbid, fin, fout, seed=fin + fout)
    layer = QuantLinear.from_params(P).to("cuda:0").eval()
    for M in (1, 5, 40):
        rng = np.random.default_rng(M)
        x = rng.standard_normal((M, fin)).astype(np.float16)
        with torch.no_grad():
            y = layer(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
        rows = slice(0, min(M, 3))
        y64 = O.qlinear_forward(P, x[rows], mode="exact")
        rms = np.sqrt((y64 ** 2).mean(axis=-1, keepdims=True))
        ref = np.maximum(np.abs(y64), rms)
        ulp = 2.0 ** (np.floor(np.log2(ref)) - 10)
        e = np.abs(y[rows] - y64) / ulp
        print(f"{cbid:11s} {fin:5d}->{fout:5d} M={M:2d}: max {e.max():6.2f} ulp  p99.9 {np.percentile(e, 99.9):5.2f}  "
              f"mean {e.mean():5.3f}   (ulp of max(|y|, rms(y)); rms(y) = {float(rms.mean()):.3f})", flush=True)
