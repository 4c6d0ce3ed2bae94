#!/usr/bin/env python3
"""Config 5 (prefill, M = 16 x 2048 rows): QuantLinear.forward per 7B shape through the M >= 32 path
(Hadamard -> fused dequant MFMA GEMM -> Hadamard; the reference-shaped decompress + dense GEMM timed beside it),
achieved TFLOP/s of the linear layers vs the
2.5 PFLOP/s dense fp16 MFMA peak, and the split between the stages."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
MODEL = sys.argv[2] if len(sys.argv) > 2 else "7b"      # prefill_bench.py 8192 70b: Llama-2-70B shapes
g = torch.Generator().manual_seed(0)
tot_t = tot_f = 0.0
SHAPES = {"7b": {"attn 4096->4096": (4096, 4096, 4), "gate/up 4096->11008": (4096, 11008, 2), "down 11008->4096": (11008, 4096, 1)},
          "70b": {"q/o 8192->8192": (8192, 8192, 2), "k/v 8192->1024": (8192, 1024, 2), "gate/up 8192->28672": (8192, 28672, 2),
                  "down 28672->8192": (28672, 8192, 1)}}
for name, (fin, fout, mult) in SHAPES[MODEL].items():
    layer = D.random_quant_linear(fin, fout, "E8P12", g, dev)
    x = torch.randn(M, fin, device=dev, dtype=torch.float16)
    def t(fn, n=5):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n
    with torch.no_grad():
        full = t(lambda: layer(x))
        W = layer.codebook.decompress_weight(layer.Qidxs)
        dec = t(lambda: layer.codebook.decompress_weight(layer.Qidxs))
        xh = torch.randn(M, layer.q_in_features, device=dev, dtype=torch.float16)
        gemm = t(lambda: xh @ W.T)
        fused = t(lambda: torch.ops.quip_lib.e8p_mm_batched(xh, layer.Qidxs, layer.codebook.grid_packed_abs))
    fl = 2.0 * M * fin * fout
    print(f"{name:22s} M={M}: forward {full:8.3f} ms = {fl / full / 1e9:7.1f} TFLOP/s ({fl / full / 1e9 / 2500:.2%} of 2.5 PF) | "
          f"fused dequant GEMM {fused:.3f} ms ({fl / fused / 1e9:.0f} TFLOP/s) | decompress {dec:.3f} ms + dense GEMM {gemm:.3f} ms "
          f"({fl / gemm / 1e9:.0f} TFLOP/s) | transforms+rest {full - fused:.3f} ms")
    tot_t += mult * full; tot_f += mult * fl
nb = 32 if MODEL == "7b" else 80
print(f"per block: {tot_t:.2f} ms, {tot_f / tot_t / 1e9:.1f} TFLOP/s; {nb} blocks: {nb * tot_t:.1f} ms time-to-first-token (linear layers only)")
