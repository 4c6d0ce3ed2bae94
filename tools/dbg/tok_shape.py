"""tokens per second of a Llama-shaped E8P12 decoder of any shape (the stage-wise step where no persistent launch is compiled for it).
usage: python tools/dbg/tok_shape.py hidden ffn layers heads kv_heads [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quip_for_all_amd import decode as D  # noqa: E402

hid, ffn, layers, heads, kvh = (int(x) for x in sys.argv[1:6])
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 32
shape = D.LlamaShape(hidden=hid, ffn=ffn, layers=layers, heads=heads, kv_heads=kvh)
dec = D.LlamaDecoder(shape, "E8P12", max_len=256, device="cuda:0", seed=0, device_init=True)
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
codes = layers * (2 * hid * hid + 2 * hid * (hid // heads) * kvh + 3 * hid * ffn) // 4
print(f"hidden {hid} ffn {ffn} x {layers} E8P12: {1 / best:.2f} tok/s, {best * 1e3:.3f} ms per token, {best * 1e6 / layers:.1f} us per block (incl. head); "
      f"codes {codes / 1e9:.2f} GB -> {codes / best / 8e12:.3f} of 8 TB/s")
