#!/usr/bin/env python3
"""One QuantLinear.forward at M rows, a few times -- rocprofv3 target: which kernels a batched forward launches.
usage: layer_forward_one.py M in_features out_features [repeats] [codebook ...]   (default E8P12; QUIP_BATCHED_MM=fused selects
the fused tile kernel beyond the skinny regime)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa
m, fin, fout = (int(v) for v in sys.argv[1:4])
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator().manual_seed(0)
x = torch.randn(m, fin, device="cuda:0", dtype=torch.float16)
for cbid in (sys.argv[5:] or ["E8P12"]):
    layer = D.random_quant_linear(fin, fout, cbid, g, "cuda:0")
    with torch.no_grad():
        for _ in range(rep):
            y = layer(x)
    torch.cuda.synchronize()
    print("done", cbid, layer.codebook.batched_regime(m, layer.q_out_features, layer.q_in_features), tuple(y.shape))
