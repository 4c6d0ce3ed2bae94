#!/usr/bin/env python3
"""rope_attn_decode latency vs context length (graph of 32 launches = one token's worth for 7B / 70B heads)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd  # noqa
dev = "cuda:0"
for heads, kvh in ((32, 32), (64, 8)):
    hd, max_len = 128, 8192
    q = torch.randn(heads, hd, device=dev).half(); k = torch.randn(kvh, hd, device=dev).half(); v = torch.randn(kvh, hd, device=dev).half()
    kc = [torch.randn(kvh, max_len, hd, device=dev).half() for _ in range(8)]
    vc = [torch.randn(kvh, max_len, hd, device=dev).half() for _ in range(8)]
    cos = torch.randn(max_len, hd, device=dev); sin = torch.randn(max_len, hd, device=dev)
    from quip_for_all_amd.register_lib import rope_attn_workspace
    ws = rope_attn_workspace(heads, hd, dev)
    for pos in (16, 128, 512, 1024, 2048, 4096, 8000):
        p = torch.tensor([pos], device=dev)
        def run():
            for i in range(32):
                torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc[i % 8], vc[i % 8], ws)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
            run()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); g.replay(); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / 32)
        print(f"heads {heads}/{kvh} pos {pos:5d}: {best:7.2f} us per launch", flush=True)
