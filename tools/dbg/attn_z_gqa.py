"""debug: grouped-query attention with the transforms in its prologue against transform launches + attention, split mode"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quip_for_all_amd  # noqa: F401,E402
from quip_for_all_amd.register_lib import rope_attn_workspace  # noqa: E402

DEV = "cuda:0"
heads, kvh, hd, pos, max_len = 64, 8, 128, int(sys.argv[1]) if len(sys.argv) > 1 else 300, 512
ns = [heads * hd, kvh * hd, kvh * hd]
g = torch.Generator().manual_seed(heads * 1000 + pos)
zs = [(torch.randn(1, n, generator=g) * 3).half().to(DEV) for n in ns]
svs = [(torch.randint(0, 2, (n,), generator=g).float() * 2 - 1).mul(torch.rand(n, generator=g) + 0.5).half().to(DEV) for n in ns]
scales = [1.0 / np.sqrt(n) for n in ns]
ang = torch.arange(max_len, dtype=torch.float32)[:, None] * (1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd)))[None]
cos = torch.cat([ang.cos(), ang.cos()], -1).to(DEV).contiguous()
sin = torch.cat([ang.sin(), ang.sin()], -1).to(DEV).contiguous()
kc0 = torch.randn(kvh, max_len, hd, generator=g).half().to(DEV)
vc0 = torch.randn(kvh, max_len, hd, generator=g).half().to(DEV)
p = torch.tensor([pos], dtype=torch.long, device=DEV)
outs = [torch.ops.quip_lib.had_transform_group([z], [n], n, 1, [None], False, [None], [sv], [None], [sc], [None], [None], None, 1e-5, None)[0]
        for z, n, sv, sc in zip(zs, ns, svs, scales)]
q, k, v = outs[0].view(heads, hd), outs[1].view(kvh, hd), outs[2].view(kvh, hd)
res = {}
for name in ("ref", "ref2", "got", "got2"):
    kc, vc = kc0.clone(), vc0.clone()
    ws = rope_attn_workspace(heads, hd, DEV)
    if name.startswith("ref"):
        o = torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc, vc, ws)
    else:
        o = torch.ops.quip_lib.rope_attn_decode_z(zs, svs, scales, cos, sin, p, kc, vc, ws)
    res[name] = (o.clone(), kc, vc)
for a, b in (("ref", "ref2"), ("got", "got2"), ("ref", "got")):
    d = (res[a][0].float() - res[b][0].float()).abs()
    print(a, b, "out differs at", int((d > 0).sum()), "max", float(d.max()), "heads", sorted(set((d > 0).nonzero()[:, 0].tolist()))[:16],
          "| caches equal", torch.equal(res[a][1], res[b][1]), torch.equal(res[a][2], res[b][2]))
