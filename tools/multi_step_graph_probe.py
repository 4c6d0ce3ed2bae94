#!/usr/bin/env python3
"""What does the boundary between two graph replays cost?  The 7B decode step captured once per graph against n steps per graph.
usage: python tools/multi_step_graph_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa: E402

dec = D.LlamaDecoder(D.LLAMA2_7B, "E8P12", max_len=512, device="cuda:0", seed=0, device_init=True)
dec.capture()
toks = {}
for n in (1, 2, 4, 8, 16):
    dec.reset(1)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        for _ in range(n):
            dec.step()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        dec.reset(1)
        for _ in range(16 // n):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(128 // n):
            g.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 128)
    toks[n] = int(dec.tok.item())
    print(f"{n:2d} steps per graph: {best * 1e6:.1f} us per token = {1 / best:.1f} tok/s; last token {toks[n]}, status {dec.engine_status()}", flush=True)
assert len(set(toks.values())) == 1
