"""tokens per second of a random-init decoder of a named shape (captured step) -- rocprofv3 target.
usage: python tools/dbg/tok_shape.py LLAMA3_8B [codebook] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quip_for_all_amd import decode as D  # noqa: E402

shape = getattr(D, sys.argv[1])
cb = sys.argv[2] if len(sys.argv) > 2 else "E8P12"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 32
dec = D.LlamaDecoder(shape, cb, max_len=256, device="cuda:0", seed=0, device_init=True)
print({k: getattr(dec, k) for k in ("attn_z", "chain", "fused_prologue", "ffn_eng", "block_eng")})
dec.capture()
for _ in range(8):
    dec.graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    dec.graph.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"{sys.argv[1]} {cb}: {1 / dt:.2f} tok/s, {dt * 1e3:.3f} ms per token, {dt * 1e6 / shape.layers:.1f} us per block (incl. head)")
