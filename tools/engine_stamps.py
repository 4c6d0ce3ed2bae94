#!/usr/bin/env python3
"""Phase clocks of the persistent MLP launch (csrc/decode_engine.hip, ENG_STAMP) next to the four stage-wise launches
it replaces, on Llama-2-7B's block shape (or `hidden ffn` from the command line); weights rotate through a pool
larger than the caches.
usage: python tools/engine_stamps.py [hidden ffn] [--pool N]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd  # noqa: E402,F401
from quip_for_all_amd.decode import random_quant_linear  # noqa: E402
from quip_for_all_amd.qlinear import (_gemv_planes_grouped, ffn_engine, gemv_unfused, out_transform_group)  # noqa: E402
from quip_for_all_amd.register_lib import ffn_engine_status, ffn_engine_workspace  # noqa: E402

DEV = "cuda:0"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
hidden, ffn = (int(args[0]), int(args[1])) if len(args) >= 2 else (4096, 11008)
pool = 12
g = torch.Generator(device=DEV).manual_seed(0)
blocks = []
for i in range(pool):
    blocks.append((random_quant_linear(hidden, ffn, "E8P12", g, DEV), random_quant_linear(hidden, ffn, "E8P12", g, DEV),
                   random_quant_linear(ffn, hidden, "E8P12", g, DEV)))
x = (torch.randn(1, hidden, device=DEV) * 1.5).to(torch.float16)


def planes_of(b):
    l0 = b[0]
    return list(torch.ops.quip_lib.had_transform_planes_group(
        x, l0.q_in_features, 1, [None, None], True, [b[0]._vec(b[0].SU), b[1]._vec(b[1].SU)],
        [b[0].wscale_float / math.sqrt(hidden), b[1].wscale_float / math.sqrt(hidden)], None, 1e-5, None, 0.0))


planes = [planes_of(b) for b in blocks]
ws = ffn_engine_workspace(ffn, blocks[0][0].K_right, DEV)
L = ffn // blocks[0][0].K_right
dbg = torch.zeros(L * 16, dtype=torch.int64, device=DEV)


def stage(b, p):
    zgu = _gemv_planes_grouped([b[0], b[1]], p)
    gg, uu = out_transform_group([b[0], b[1]], zgu)
    return gemv_unfused(b[2], uu, gate=gg)


def timed(fn, reps=5):
    # capture one pass over the pool in a graph, replay
    for b, p in zip(blocks, planes):
        fn(b, p)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr):
            for b, p in zip(blocks, planes):
                fn(b, p)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / pool)
    return float(np.median(ts))


t_stage = timed(stage)
t_eng = timed(lambda b, p: ffn_engine(b[0], b[1], b[2], p, ws))
print(f"shape ({hidden}, {ffn}): stage-wise 4 launches {t_stage:.2f} us, engine launch {t_eng:.2f} us  (graph replay, pool of {pool} blocks)")
assert ffn_engine_status(ws) == 0

# phase clocks: one launch per block with the stamp buffer, cold weights
names = ["entry", "requests issued", "tables+planes", "gemv gate/up done (wave0)", "barrier", "kmix+publish", "row owner",
         "rows gathered", "kmix_in+max", "planes", "gemv down", "end"]
acc, extra = [], []
for b, p in zip(blocks, planes):
    dbg.zero_()
    ffn_engine(b[0], b[1], b[2], p, ws, dbg)
    torch.cuda.synchronize()
    d16 = dbg.cpu().numpy().reshape(L, 16).astype(np.float64)
    d = d16[:, :12]
    acc.append(np.diff(d, axis=1))      # every XCD has its own clock: only differences inside a workgroup mean something
    extra.append(np.stack([d16[:, 12] - d16[:, 5], d16[:, 13] - d16[:, 6], d16[:, 14] - d16[:, 13], d16[:, 7] - d16[:, 14]], 1))
a = np.stack(acc[2:])                   # (launches, L, 11)
K = blocks[0][0].K_right
print("clocks between stamps (mean over workgroups | mean of the slowest workgroup | row owners' mean), s_memtime ticks:")
for i in range(11):
    print(f"  {i:2d} -> {i + 1:2d} {names[i + 1]:28s} {a[:, :, i].mean():9.0f} {a[:, :, i].max(axis=1).mean():9.0f} {a[:, :K, i].mean():9.0f}")
print(f"  total {a.sum(axis=2).mean():9.0f}")
ex = np.stack(extra[2:])
print(f"  row owners: publish -> inbox complete {ex[:, :K, 0].mean():.0f}; everyone: publish/row work -> poll done {ex[:, :, 1].mean():.0f} "
      f"(row owners {ex[:, :K, 1].mean():.0f}), full sweep {ex[:, :, 2].mean():.0f}, staging + barrier {ex[:, :, 3].mean():.0f}")
