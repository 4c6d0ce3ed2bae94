"""tokens per second of the 80-layer Llama-2-70B-shaped E8P12 decoder (stage-wise step).  usage: python tools/dbg/tok70b.py [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quip_for_all_amd import decode as D  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dec = D.LlamaDecoder(D.LLAMA2_70B, "E8P12", max_len=256, device="cuda:0", seed=0, device_init=True)
print("attn_z", dec.attn_z, "chain", dec.chain, "fused_prologue", dec.fused_prologue, "ffn_eng", dec.ffn_eng, "block_eng", dec.block_eng)
dec.capture()
for _ in range(8):
    dec.graph.replay()
best = 1e9
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        dec.graph.replay()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / steps)
print(f"70B E8P12: {1 / best:.2f} tok/s, {best * 1e3:.3f} ms per token, {best * 1e6 / 80:.1f} us per block (incl. head)")
