#!/usr/bin/env python3
"""Run the fused batched product (quip_lib::e8p_mm_batched) a few times -- rocprofv3 target (tools/prof_kernel.sh).
usage: prefill_one.py M N K [launches]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
m, n, k = (int(v) for v in sys.argv[1:4])
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
Qd = torch.randint(-32768, 32768, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev)
x = torch.randn(m, k, device=dev, dtype=torch.float16)
grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
for _ in range(launches):
    y = torch.ops.quip_lib.e8p_mm_batched(x, Qd, grid)
torch.cuda.synchronize()
print("done", float(y.float().abs().mean()))
