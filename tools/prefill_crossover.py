#!/usr/bin/env python3
"""Fused dequant GEMM (e8p_mm_batched) against decompress + dense GEMM (the reference's shape, hipBLASLt) over M, for the
three Llama-2-7B shapes: where the default of codebooks.E8P12_codebook.forward switches.
usage: prefill_crossover.py [codebook ...]   (default E8P12; the others run the tile kernel's MODE 1..4 through mm_batched)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa
dev = "cuda:0"
g = torch.Generator().manual_seed(0)


def t(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for cbid in (sys.argv[1:] or ["E8P12"]):
    for fin, fout in ((4096, 4096), (4096, 11008), (11008, 4096)):
        layer = D.random_quant_linear(fin, fout, cbid, g, dev)
        cb = layer.codebook
        for M in ((256, 512, 1024, 2048, 4096, 8192, 16384, 32768) if cbid == "E8P12" else (512, 2048, 8192, 32768)):
            x = torch.randn(M, layer.q_in_features, device=dev, dtype=torch.float16)
            with torch.no_grad():
                if cbid == "E8P12":
                    fused = t(lambda: torch.ops.quip_lib.e8p_mm_batched(x, layer.Qidxs, cb.grid_packed_abs))
                else:
                    fused = t(lambda: cb.mm_batched(x, layer.Qidxs))
                lib = t(lambda: x @ cb.decompress_weight(layer.Qidxs).T)
            tf = 2.0 * M * layer.q_in_features * layer.q_out_features / fused * 1e-6
            print(f"{cbid:11s} {fin:6d} -> {fout:6d}  M = {M:6d}: fused {fused:9.1f} us ({tf:6.0f} TFLOP/s) | decompress + GEMM {lib:9.1f} us | ratio {fused / lib:.2f}")
