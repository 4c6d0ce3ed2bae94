"""tokens per second of the 32-layer 7B decoder on the HI codebook (persistent block launch).  usage: python tools/dbg/hi_tok.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quip_for_all_amd import decode as D
for cb in ("HI",):
    dec = D.LlamaDecoder(D.LLAMA2_7B, cb, max_len=256, device="cuda:0", seed=0, device_init=True)
    print(cb, "block_eng", dec.block_eng, getattr(dec, "eng_codebook", None))
    dec.capture()
    for _ in range(8): dec.graph.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(64): dec.graph.replay()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(cb, "tok/s", 64 / dt, "status", dec.engine_status())
