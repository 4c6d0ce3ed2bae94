#!/usr/bin/env python3
"""Where the waves of the 70B launch's measurement mode spend their clocks: a library built with -DQUIP_GQA_WAITSTAT=1
(tools/dbg/build_variant.sh NAME decode_block_gqa.hip "-DQUIP_GQA_WAITSTAT=1 ...", QUIP_LIB_PATH) adds up, per wave, the
s_memtime ticks inside the ring's `s_waitcnt vmcnt` and the launch's total ticks and s_memrealtime span (100 MHz).
usage: python tools/gqa_waitstat.py [layers] [launches]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 80
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
shape = D.LlamaShape(hidden=8192, ffn=28672, layers=layers, heads=64, kv_heads=8, vocab=32000)
dec = D.LlamaDecoder(shape, "E8P12", max_len=256, device="cuda:0", seed=0, device_init=True)
assert dec.block_eng and dec.eng_shape == 1
s = dec.s
h = dec.embed[:1].reshape(-1).clone()
pos = torch.full((1,), 40, dtype=torch.long, device=dec.dev)
dbg = torch.zeros(256 * 8 * 4, dtype=torch.int64, device=dec.dev)
args = (dec.eng_layers, h, pos, dec.cos, dec.sin, dec.eng_grid, dec.eng_ws, layers, dec.max_len, s.rms_eps,
        1.0 / math.sqrt(s.head_dim), dbg, -2, 0, 0.0, 1)
code_bytes = sum(L_[k].Qidxs.numel() * L_[k].Qidxs.element_size() for L_ in dec.layers for k in ("q", "k", "v", "o", "gate", "up", "down"))
ts = []
for it in range(launches + 2):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    torch.ops.quip_lib.block_engine(*args)
    b.record()
    torch.cuda.synchronize()
    if it >= 2:
        us = a.elapsed_time(b) * 1e3
        d = dbg.cpu().numpy().reshape(256, 8, 4).astype(np.float64)
        wait, tot, real = d[..., 0], d[..., 1], d[..., 2]
        line = f"launch {it - 2}: {us:8.1f} us  {code_bytes / us / 1e3 / 8000:.4f} of 8 TB/s"
        if tot.max() > 0:
            clk = tot / (real * 10.0)        # ticks per ns
            line += (f" | wait / total ticks: mean {np.mean(wait / tot):.3f} min {np.min(wait / tot):.3f} max {np.max(wait / tot):.3f}"
                     f" | ticks per ns: mean {clk.mean():.3f} min {clk.min():.3f} max {clk.max():.3f} | kernel span {real.max() / 100:.1f} us"
                     f" | ticks per item and wave: total {tot.mean() / (54 * layers):.0f}, not waiting {(tot - wait).mean() / (54 * layers):.0f}")
        ph = dbg.cpu().numpy().reshape(256, 8, 4)[..., 3]
        if ph.max() > 0:
            perm, iss, mf = [((ph >> sh) & ((1 << 21) - 1)).astype(np.float64) * 256 for sh in (0, 21, 42)]
            n = 54 * layers
            line += (f" | per item: codes -> addresses {perm.mean() / n:.0f}, request {iss.mean() / n:.0f}, look-ups + MFMAs {mf.mean() / n:.0f},"
                     f" rest {(tot - wait - perm - iss - mf).mean() / n:.0f}")
        print(line)
        ts.append(us)
print("median us", float(np.median(ts)), "frac", code_bytes / float(np.median(ts)) / 1e3 / 8000, "status", dec.engine_status())
dec.engine_reset()
