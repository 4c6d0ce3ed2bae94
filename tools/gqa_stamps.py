#!/usr/bin/env python3
"""Phase clocks of one block inside the persistent token launch of the grouped-query 8192-wide shape
(csrc/decode_block_gqa.hip, BSTAMP).  usage: python tools/gqa_stamps.py [layers] [dbg_layer] [pos]"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dl = int(sys.argv[2]) if len(sys.argv) > 2 else layers // 2
pos0 = int(sys.argv[3]) if len(sys.argv) > 3 else 40
shape = D.LlamaShape(hidden=8192, ffn=28672, layers=layers, heads=64, kv_heads=8, vocab=32000)
dec = D.LlamaDecoder(shape, "E8P12", max_len=max(256, pos0 + 16), device="cuda:0", seed=0, device_init=True)
assert dec.block_eng and dec.eng_shape == 1
dec.reset(7)
dec.pos.fill_(pos0)
h = dec.embed[dec.tok].reshape(-1)
dbg = torch.zeros(256 * 32, dtype=torch.int64, device="cuda:0")
acc = []
args = (dec.eng_layers, h, dec.pos, dec.cos, dec.sin, dec.eng_grid, dec.eng_ws, layers, dec.max_len,
        shape.rms_eps, 1.0 / math.sqrt(128))
for it in range(6):
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.ops.quip_lib.block_engine(*args, dbg, dl, 0, 0.0, 1)
    e1.record()
    torch.cuda.synchronize()
    if it >= 2:
        acc.append((dbg.cpu().numpy().reshape(256, 32).astype(np.float64), e0.elapsed_time(e1) * 1e3))
med = np.median([t for _, t in acc])
print(f"status {dec.engine_status()}; launch of {layers} blocks at position {pos0}: {med:.1f} us = {med / layers:.2f} us per block")
D_ = np.stack([d for d, _ in acc])
head = (np.arange(256) % 4 == 0) if pos0 < 128 else np.ones(256, bool)
own = np.arange(256) < 7
kv = np.arange(256) % 2 == 1
seq = [(0, 18, "top -> z_d gathered (descriptor, bases, hand-off)"), (18, 19, "out(down): fwd 8192"), (19, 20, "h update, sumsq"),
       (20, 21, "in(q [, k|v]): rev 8192"), (21, 22, "max, planes"), (22, 3, "products q, k|v + publish"),
       (3, 4, "head: z_q, z_k, z_v gathered"), (4, 5, "head: out(q) fwd 8192, out(k), out(v) in a wave"), (5, 6, "head: attention + publish a"),
       (3, 7, "others: wait for a"), (6, 7, "heads: a gathered"), (7, 8, "in(o): H_64 over the heads, planes    "), (8, 9, "products o + publish"),
       (9, 23, "z_o gathered"), (23, 24, "out(o): fwd 8192"), (24, 25, "h update, sumsq"), (25, 26, "in(gate | up): rev 8192    "), (26, 27, "planes (one matrix)"),
       (27, 11, "products gate, up (28 items per wave)"), (11, 12, "7 x 7 mix, hop 1 sent"), (12, 28, "owners: inbox complete"),
       (28, 29, "owners: fht 4096 x 2"), (29, 30, "owners: SV, SiLU product, SU"), (30, 31, "owners: rev 4096"), (31, 13, "owners: publish, maximum"), (12, 14, "everybody: owners' maxima known"), (14, 15, "rows swept, mixed, planes of down"),
       (15, 16, "products down (14 items per wave) + publish"), (16, 17, "drain")]
print(f"{'ticks between stamps inside block ' + str(dl):62s} {'all':>8s} {'heads':>8s} {'owners':>8s} {'k|v wgs':>8s}")
for a_, b_, name in seq:
    seg = D_[:, :, b_] - D_[:, :, a_]
    ok = (D_[:, :, b_] > 0) & (D_[:, :, a_] > 0)

    def m(mask):
        sel = ok & mask[None, :]
        return f"{seg[sel].mean():8.0f}" if sel.any() else f"{'-':>8s}"
    print(f"  {a_:2d}->{b_:2d} {name:55s} {m(np.ones(256, bool))} {m(head)} {m(own)} {m(kv)}")
span = D_[:, :, 17] - D_[:, :, 0]
print(f"  block span (stamp 0 -> 17): {span.mean():.0f} ticks; ticks per us (launch time / blocks): {span.mean() / (med / layers):.1f}")
