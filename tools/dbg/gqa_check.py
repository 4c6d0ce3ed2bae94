"""the grouped-query persistent block launch (csrc/decode_block_gqa.hip) against the stage-wise step on a 70B-shaped model
of a few blocks: logits, caches, engine status.  usage: python tools/dbg/gqa_check.py [layers] [tokens] [pos0]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quip_for_all_amd import decode as D  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tokens = int(sys.argv[2]) if len(sys.argv) > 2 else 3
pos0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
shape = D.LlamaShape(hidden=8192, ffn=28672, layers=layers, heads=64, kv_heads=8, vocab=2048)


def make(engine):
    os.environ["QUIP_BLOCK_ENGINE"] = "1" if engine else "0"
    return D.LlamaDecoder(shape, "E8P12", max_len=max(64, pos0 + tokens + 8), device="cuda:0", seed=3, device_init=True)


a, b = make(True), make(False)
with torch.no_grad():
    for La, Lb in zip(a.layers, b.layers):
        for k in ("gate", "up", "down"):
            for name in ("had_left", "had_right"):
                if getattr(La[k], name) is not None:
                    getattr(Lb[k], name).copy_(getattr(La[k], name))
a._init_block_engine()
print("block_eng", a.block_eng, getattr(a, "eng_shape", None), "| stage-wise:", b.block_eng, b.chain, b.fused_prologue)
assert a.block_eng and not b.block_eng
for dec in (a, b):
    dec.reset(first_token=7)
if pos0:
    g = torch.Generator(device="cuda:0").manual_seed(pos0)
    kc = (torch.randn(a.kcache[..., :pos0, :].shape, generator=g, device="cuda:0") * 0.5).half()
    vc = (torch.randn(a.vcache[..., :pos0, :].shape, generator=g, device="cuda:0") * 0.5).half()
    for dec in (a, b):
        dec.kcache[..., :pos0, :].copy_(kc)
        dec.vcache[..., :pos0, :].copy_(vc)
        dec.pos.fill_(pos0)
with torch.no_grad():
    for t in range(tokens):
        la = a.step().float().clone()
        torch.cuda.synchronize()
        st = a.engine_status()
        lb = b.step().float().clone()
        rms = lb.pow(2).mean().sqrt().item()
        ulp = 2.0 ** (np.floor(np.log2(rms)) - 10)
        d = (la - lb).abs().max().item()
        p = pos0 + t
        dk = (a.kcache[:, :, p].float() - b.kcache[:, :, p].float()).abs().max().item()
        dv = (a.vcache[:, :, p].float() - b.vcache[:, :, p].float()).abs().max().item()
        print(f"token {t}: status {st:#x} max |dlogit| {d:.5f} = {d / ulp:.2f} ulp(rms {rms:.3f}); finite {bool(torch.isfinite(la).all())}; "
              f"tok {a.tok.item()} / {b.tok.item()}; cache row diff k {dk:.5f} v {dv:.5f}")
        a.tok.copy_(b.tok)
