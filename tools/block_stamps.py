#!/usr/bin/env python3
"""Phase clocks of one block inside the persistent token launch (csrc/decode_block.hip, BSTAMP), Llama-2-7B shape.
usage: python tools/block_stamps.py [layers] [dbg_layer] [pos] [g8]      (g8: the Llama-3-8B shape, decode_block_g8.hip)"""
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
g8 = len(sys.argv) > 4 and sys.argv[4] == "g8"
shape = (D.LlamaShape(hidden=4096, ffn=14336, layers=layers, heads=32, kv_heads=8, vocab=32000) if g8 else
         D.LlamaShape(hidden=4096, ffn=11008, layers=layers, heads=32, kv_heads=32, vocab=32000))
dec = D.LlamaDecoder(shape, "E8P12", max_len=max(256, pos0 + 16), device="cuda:0", seed=0, device_init=True)
assert dec.block_eng
dec.reset(7)
dec.pos.fill_(pos0)
h = dec.embed[dec.tok].reshape(-1)
dbg = torch.zeros(256 * 32, dtype=torch.int64, device="cuda:0")
names = ["(top)", "z_d gathered", "edge: out(down) + next block's in(q,k,v)", "next block's gemv q,k,v + publish", "head: z_qkv gathered", "head: out(q,k,v)",
         "attention + publish a", "a gathered", "in(o)", "gemv o + publish", "z_o gathered", "edge: out(o) + in(gate,up)",
         "gemv gate,up", "kmix + publish (mlp hop 1)", "row owner", "rows gathered", "kmix_in + planes", "gemv down + publish"]
acc = []
args = (dec.eng_layers, h, dec.pos, dec.cos, dec.sin, dec.eng_grid, dec.eng_ws, layers, dec.max_len,
        shape.rms_eps, 1.0 / math.sqrt(128))
for it in range(6):
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.ops.quip_lib.block_engine(*args, dbg, dl, 0, 0.0, dec.eng_shape)
    e1.record()
    torch.cuda.synchronize()
    if it >= 2:
        d = dbg.cpu().numpy().reshape(256, 32).astype(np.float64)
        acc.append((d, e0.elapsed_time(e1) * 1e3))
print(f"status {dec.engine_status()}; launch of {layers} blocks: {np.median([t for _, t in acc]):.1f} us = {np.median([t for _, t in acc]) / layers:.2f} us per block")
head = np.arange(256) % 8 == 0
D_ = np.stack([d for d, _ in acc])          # (runs, 256, 18)
print("clocks between stamps inside block %d (mean over workgroups | head workgroups | others), s_memtime ticks:" % dl)
# iteration dl of the rotated loop: stamp 0 right behind the publication of ITS z_q, z_k, z_v, 4 .. 17, then 1 (z_d gathered),
# 2 and 3 (the in-edge, products and publication of q, k, v of block dl + 1)
order = [4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 1, 2, 3]
prev_of = {4: 0, 7: 0, 1: 17}
last = 0
for i in order:
    prev = prev_of.get(i, last)
    last = i
    if i in (4, 5, 6):                      # head-only stamps
        seg = D_[:, head, i] - D_[:, head, prev]
        print(f"  {i:2d} {names[i]:34s} {'':>9s} {seg.mean():9.0f}")
        continue
    if i == 7:                              # from stamp 0 (others) / 6 (heads)
        so = D_[:, ~head, 7] - D_[:, ~head, 0]
        sh = D_[:, head, 7] - D_[:, head, 6]
        print(f"  {i:2d} {names[i]:34s} {'':>9s} {sh.mean():9.0f} {so.mean():9.0f}")
        continue
    seg = D_[:, :, i] - D_[:, :, prev]
    print(f"  {i:2d} {names[i]:34s} {seg.mean():9.0f} {seg[:, head].mean():9.0f} {seg[:, ~head].mean():9.0f}")
span = D_[:, :, 3] - D_[:, :, 0]
print(f"  block span (stamp 0 -> 3 of the next block): {span.mean():.0f} ticks")
# round 5: the edges' stages (fwd / rev): gate-up edge stamps 18..22, q-k-v edge stamps 23, 24, 28, 29, 30
en = ["gathered", "fwd<1>", "h update + sums + ln, su", "rev<NT>", "planes + barrier"]
for title, st, before, after in (("gate / up edge", [18, 19, 20, 21, 22], 10, 11), ("q / k / v edge", [23, 24, 28, 29, 30], 1, 2)):
    print(f"inside the {title} (stamps {st}):")
    for i in range(1, 5):
        seg = D_[:, :, st[i]] - D_[:, :, st[i - 1]]
        print(f"  {en[i]:28s} {seg.mean():9.0f}")
    print(f"  (stamp {before} -> {st[0]}: {(D_[:, :, st[0]] - D_[:, :, before]).mean():.0f}, {st[4]} -> {after}: {(D_[:, :, after] - D_[:, :, st[4]]).mean():.0f})")
ro = np.arange(256) < (28 if g8 else 22)
print("inside the MLP edge: row owners: publish(13) -> inbox complete %.0f, -> rows published(14) %.0f; everyone: 14 -> poll done(26) %.0f (row owners %.0f), sweep(27) %.0f, staging + barrier(15) %.0f" % (
    (D_[:, ro, 25] - D_[:, ro, 13]).mean(), (D_[:, ro, 14] - D_[:, ro, 25]).mean(), (D_[:, :, 26] - D_[:, :, 14]).mean(),
    (D_[:, ro, 26] - D_[:, ro, 14]).mean(), (D_[:, :, 27] - D_[:, :, 26]).mean(), (D_[:, :, 15] - D_[:, :, 27]).mean()))

if os.environ.get("QUIP_ATT_STAMPS"):      # a library built with -DQUIP_ATT_STAMPS=1: stamps 18..22 sit inside the attention (head workgroups)
    an = ["5 -> q, k roped, new row appended (18)", "key rounds (19)", "states to LDS + barrier (20)", "merge of the 16 groups (21)", "barrier (22)", "publication (6)"]
    pts = [5, 18, 19, 20, 21, 22, 6]
    print("inside the attention (head workgroups):")
    for i in range(6):
        print(f"  {an[i]:44s} {(D_[:, head, pts[i + 1]] - D_[:, head, pts[i]]).mean():9.0f}")
