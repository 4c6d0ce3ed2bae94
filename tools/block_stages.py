#!/usr/bin/env python3
"""Graph-timed launches of every stage of one decoder block, in the order LlamaDecoder._step_fused issues
them (the stage's own inputs are static buffers, so each stage is timed alone, back to back with itself).
usage: block_stages.py [7b|70b] [codebook]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd  # noqa
from quip_for_all_amd import decode as D
from quip_for_all_amd.qlinear import gemv_chain, gemv_fused, out_transform_group, gemv_unfused, gemv_group_unfused
import dataclasses

def graph_time(fn, reps=100):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
        for _ in range(reps): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best

model = sys.argv[1] if len(sys.argv) > 1 else "70b"
cb = sys.argv[2] if len(sys.argv) > 2 else "E8P12"
shape = dataclasses.replace(D.LLAMA2_70B if model == "70b" else D.LLAMA2_7B, layers=2, vocab=1024)
torch.manual_seed(0)
dec = D.LlamaDecoder(shape, codebook=cb, max_len=64, device="cuda:0", device_init=True)
s = dec.s
print("chain", dec.chain, "o_fused", dec.o_fused, "qkv_fused", dec.qkv_fused, "fused_prologue", dec.fused_prologue)
L = dec.layers[1]; P = dec.layers[0]
h = torch.randn(1, s.hidden, device="cuda").half()
qkv = [L["q"], L["k"], L["v"]]; gu = [L["gate"], L["up"]]
zd = torch.randn(1, P["down"].q_out_features, device="cuda").half() if not hasattr(P["down"], "z_dtype") else None
dec.pos.fill_(17)
with torch.no_grad():
    # produce realistic intermediate buffers once
    zd = gemv_unfused(P["down"], torch.randn(1, s.ffn, device="cuda").half() * 0.1, gate=torch.randn(1, s.ffn, device="cuda").half())
    h1, zs = dec._zx(qkv, P["down"], zd, h, L["ln1"])
    q, k, v = out_transform_group(qkv, zs)
    a = dec._attention(1, q, k, v, None, None, None)
    if dec.o_fused:
        _, (zo,) = gemv_fused([L["o"]], x=a.reshape(1, s.hidden))
    else:
        zo = gemv_unfused(L["o"], a.reshape(1, s.hidden))
    h2, zgu = dec._zx(gu, L["o"], zo, h1, L["ln2"])
    g, u = out_transform_group(gu, zgu)
    stages = [
        ("zx(down->qkv) [chain+gemv]", lambda: dec._zx(qkv, P["down"], zd, h, L["ln1"])),
        ("out_transform qkv", lambda: out_transform_group(qkv, zs)),
        ("attention", lambda: dec._attention(1, q, k, v, None, None, None)),
        ("o gemv (fused)" if dec.o_fused else "o gemv (unfused)", (lambda: gemv_fused([L["o"]], x=a.reshape(1, s.hidden))) if dec.o_fused else (lambda: gemv_unfused(L["o"], a.reshape(1, s.hidden)))),
        ("zx(o->gate,up) [chain+gemv]", lambda: dec._zx(gu, L["o"], zo, h1, L["ln2"])),
        ("out_transform gate,up", lambda: out_transform_group(gu, zgu)),
        ("down [planes+gemv]", lambda: gemv_unfused(L["down"], u, gate=g)),
    ]
    tot = 0
    for name, fn in stages:
        t = graph_time(fn); tot += t
        print("%-34s %7.2f us" % (name, t), flush=True)
    print("%-34s %7.2f us  -> %.1f tok/s at %d layers" % ("block", tot, 1e6 / (tot * (80 if model == '70b' else 32)), 80 if model == '70b' else 32))
