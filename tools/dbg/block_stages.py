"""debug: one block through the persistent launch vs the stage-wise launches, vector by vector (the engine's hand-off
buffers keep z_q z_k z_v a z_o z_d of the last block)"""
import os, sys, math
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quip_for_all_amd import decode as D
from quip_for_all_amd.qlinear import gemv_fused, gemv_group_unfused, chain_planes, ffn_engine, out_transform_group
shape = D.LlamaShape(hidden=4096, ffn=11008, layers=1, heads=32, kv_heads=32, vocab=2048)
dec = D.LlamaDecoder(shape, "E8P12", max_len=32, device="cuda:0", seed=3, device_init=True)
assert dec.block_eng
dec.reset(7)
L = dec.layers[0]
h = dec.embed[dec.tok]
s = dec.s
with torch.no_grad():
    # stage-wise
    qkv = [L["q"], L["k"], L["v"]]
    zs = gemv_group_unfused(qkv, h, rms_weight=L["ln1"], rms_eps=s.rms_eps)
    kc, vc = dec.kcache[0].clone(), dec.vcache[0].clone()
    a = torch.ops.quip_lib.rope_attn_decode_z(list(zs), [l._vec(l.SV) for l in qkv], [1.0 / 64] * 3, dec.cos, dec.sin, dec.pos,
                                              kc, vc, None)
    _, (zo,) = gemv_fused([L["o"]], x=a.reshape(1, s.hidden))
    h2, planes = chain_planes([L["gate"], L["up"]], L["o"], zo, residual=h, rms_weight=L["ln2"], rms_eps=s.rms_eps)
    zd = ffn_engine(L["gate"], L["up"], L["down"], planes, dec.ffn_ws)
    (hf,) = out_transform_group([L["down"]], [zd], residual=[h2])
    # engine
    out = torch.ops.quip_lib.block_engine(dec.eng_layers, h.reshape(-1), dec.pos, dec.cos, dec.sin,
                                          L["q"].codebook.grid_packed_abs, dec.eng_ws, 1, dec.max_len, s.rms_eps, 1.0 / math.sqrt(128))
    torch.cuda.synchronize()
print("status", dec.engine_status())
ws = dec.eng_ws.cpu().numpy()
g = ws[64:64 + 6 * 2048 * 8].view(np.uint32).reshape(6, 2048, 2)
vals = g[:, :, 0].copy().view(np.float16).reshape(6, 4096).astype(np.float32)
tags = g[:, :, 1]
names = ["z_q", "z_k", "z_v", "a", "z_o", "z_d"]
refs = [zs[0], zs[1], zs[2], a, zo, zd]
for i, (nm, r) in enumerate(zip(names, refs)):
    r = r.float().cpu().numpy().reshape(-1)
    d = np.abs(vals[i] - r)
    bad = np.nonzero(d > 0)[0]
    print(f"{nm}: max diff {d.max():.5f} (|ref| max {np.abs(r).max():.3f}), differing {len(bad)} / 4096, tags {np.unique(tags[i])[:4]}",
          (f"first bad idx {bad[:8]}" if len(bad) else ""))
d = (out.float() - hf.reshape(-1).float()).abs()
print("h_out max diff", d.max().item(), "kcache diff", (dec.kcache[0] - kc).abs().max().item(), (dec.vcache[0] - vc).abs().max().item())
