import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quip_for_all_amd as Q
from quip_for_all_amd import qlinear as QL
from oracle import quip_oracle as O
DEV = "cuda:0"
def _layer(P): return Q.QuantLinear.from_params(P).to(DEV).eval()
k, fouts = 4096, (4096, 4096, 4096)
layers = [_layer(O.make_layer("E8P12", k, fo, seed=k + fo + i)) for i, fo in enumerate(fouts)]
rng = np.random.default_rng(k + len(fouts))
t = lambda a: torch.from_numpy(a.astype(np.float16)).to(DEV)
w = t(1 + 0.1 * rng.standard_normal(k))
prev = _layer(O.make_layer("E8P12", 1024, k, seed=k + 7))
z = t(rng.standard_normal((1, k)) * 8)
res = t(rng.standard_normal((1, k)))
with torch.no_grad():
    (h_ref,) = QL.out_transform_group([prev], [z], residual=[res])
    for it in range(4):
        h, zs = QL.gemv_fused(layers, prev=prev, z=z, residual=res, rms_weight=w)
        torch.cuda.synchronize()
        bad = (h != h_ref).nonzero()
        print("iter", it, "mismatches", bad.shape[0], bad[:12, 1].tolist(), (h - h_ref).abs().max().item())
        if bad.shape[0]:
            i = bad[0, 1].item()
            print("  h", h[0, i:i+8].tolist(), "ref", h_ref[0, i:i+8].tolist())
