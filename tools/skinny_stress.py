#!/usr/bin/env python3
"""Stress: 200 launches of e8p_mm_skinny per shape and row count, fresh activations each, every one compared with
the fp32 product (an intermittent stale-register bug of a first version showed up here, not in single launches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
dev = "cuda:0"
cb = Q.codebook.codebook_id["E8P12"](inference=True).to(dev)
torch.manual_seed(0)
tot = 0
for (n, k) in [(4096, 4096), (64, 2048), (11008, 4096), (4096, 11008)]:
    q = torch.randint(-32768, 32767, (n, k // 8), dtype=torch.int32, device=dev).to(torch.int16)
    W = cb.decompress_weight(q).float()
    for M in (2, 3, 6, 16, 31):
        nbad = 0
        for it in range(200):
            x = torch.randn(M, k, device=dev).half()
            y = torch.ops.quip_lib.e8p_mm_skinny(x, q, cb.grid_packed_abs).float()
            ref = x.float() @ W.T
            if not bool(((y - ref).abs() <= 2e-3 * ref.abs().max()).all()):
                nbad += 1
        tot += nbad
        print(f"N={n} K={k} M={M}: {nbad} / 200 bad", flush=True)
print("total bad", tot)
