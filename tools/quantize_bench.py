#!/usr/bin/env python3
"""Quantise-time codebook search: the structured HIP kernel (quip_e8p_quantize_f32) next to the reference's
formulation on the same GPU (dense (N, 8) x (8, 65536) fp32 GEMM + arg max, e8p12.py:125-128), per LDLQ-step
batch size N (= out_features of the layer being rounded)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
dev = "cuda:0"
cb = Q.codebook.codebook_id["E8P12"](inference=False).to(dev)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
for N in (4096, 11008, 28672, 262144):
    X = torch.randn(N, 8, device=dev) * 1.03
    k = t(lambda: torch.ops.quip_lib.e8p_quantize(X, cb.grid_packed_abs))
    d = t(lambda: (2 * X @ cb.grid.T - cb.grid_norm).argmax(-1), n=5)
    ops = N * 512 * 27
    print(f"N={N:7d}: structured kernel {k:9.1f} us ({ops / k / 1e6:6.2f} Tops/s VALU-ish) | dense GEMM + argmax {d:10.1f} us "
          f"({2 * N * 8 * 65536 / d / 1e6:6.1f} TFLOP/s) | x{d / k:.0f}")
