#!/usr/bin/env python3
"""Who gets to each stamp of the 7B block launch LAST: the stamps taken with s_memrealtime (one 100 MHz counter for the whole
device; dbg_layer bit 16) so that workgroups on different XCDs compare.  usage: python tools/block_laggards.py [layers] [dbg_layer] [pos]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dl = int(sys.argv[2]) if len(sys.argv) > 2 else layers // 2
pos0 = int(sys.argv[3]) if len(sys.argv) > 3 else 100
shape = D.LlamaShape(hidden=4096, ffn=11008, layers=layers, heads=32, kv_heads=32, vocab=32000)
dec = D.LlamaDecoder(shape, "E8P12", max_len=max(256, pos0 + 16), device="cuda:0", seed=0, device_init=True)
assert dec.block_eng
dec.reset(7)
dec.pos.fill_(pos0)
h = dec.embed[dec.tok].reshape(-1)
dbg = torch.zeros(256 * 32, dtype=torch.int64, device="cuda:0")
args = (dec.eng_layers, h, dec.pos, dec.cos, dec.sin, dec.eng_grid, dec.eng_ws, layers, dec.max_len, shape.rms_eps, 1.0 / math.sqrt(128))
acc = []
for it in range(8):
    dbg.zero_()
    torch.ops.quip_lib.block_engine(*args, dbg, dl | 0x10000)
    torch.cuda.synchronize()
    if it >= 2:
        acc.append(dbg.cpu().numpy().reshape(256, 32).astype(np.float64))
D_ = np.stack(acc)                        # (runs, 256, 32), units of 10 ns
names = {3: "q k v published (next block)", 6: "a published (heads)", 9: "z_o published", 12: "gate/up products done", 13: "columns published",
         14: "rows published (owners) / decode done", 17: "z_d published", 2: "q k v edge done", 11: "gate/up edge done", 8: "in(o) done", 16: "down planes done"}
print("stamp: mean over workgroups (us after the block's stamp 0 mean) | the four latest workgroups (+us behind the mean)")
t0 = D_[:, :, 0].mean(axis=1, keepdims=True)
for i in [6, 8, 9, 11, 12, 13, 14, 16, 17, 2, 3]:
    col = (D_[:, :, i] - t0).mean(axis=0) / 100.0        # us
    ok = D_[:, :, i].min(axis=0) > 0
    if not ok.any():
        continue
    mean = col[ok].mean()
    top = np.argsort(-np.where(ok, col, -np.inf))[:4]
    print(f"  {i:2d} {names.get(i, ''):34s} {mean:8.2f} | " + ", ".join(f"wg {int(t)} +{col[t] - mean:.2f}" for t in top)
          + f" | p50 +{np.median(col[ok]) - mean:.2f} p95 +{np.percentile(col[ok], 95) - mean:.2f}")
