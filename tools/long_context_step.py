#!/usr/bin/env python3
"""Decode step time against context length: the persistent block launch and the stage-wise step (QUIP_BLOCK_ENGINE=0 in a second
process), Llama-2-7B E8P12.  usage: long_context_step.py [positions...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa
poss = [int(a) for a in sys.argv[1:]] or [64, 512, 2048, 4000]
dec = D.LlamaDecoder(D.LLAMA2_7B, "E8P12", max_len=max(poss) + 64, device="cuda:0", seed=0, device_init=True)
dec.capture()
for p in poss:
    dec.reset(first_token=1)
    dec.pos.fill_(p)
    for _ in range(3):
        dec.graph.replay()
    dec.pos.fill_(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 16
    for _ in range(n):
        dec.graph.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{'block engine' if dec.block_eng else 'stage-wise'}: position {p:5d}..{p + n:5d}: {dt * 1e3:.3f} ms per token ({1 / dt:.0f} tok/s), status {dec.engine_status()}", flush=True)
