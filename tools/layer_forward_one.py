#!/usr/bin/env python3
"""One QuantLinear.forward at M rows, a few times -- rocprofv3 target: which kernels a batched forward launches.
usage: layer_forward_one.py M in_features out_features [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa
m, fin, fout = (int(v) for v in sys.argv[1:4])
rep = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator().manual_seed(0)
layer = D.random_quant_linear(fin, fout, "E8P12", g, "cuda:0")
x = torch.randn(m, fin, device="cuda:0", dtype=torch.float16)
with torch.no_grad():
    for _ in range(rep):
        y = layer(x)
torch.cuda.synchronize()
print("done", tuple(y.shape))
