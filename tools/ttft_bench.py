#!/usr/bin/env python3
"""Time to first token of the 32-layer 7B decoder (LlamaDecoder.prefill: one batched pass over the prompt) per prompt
length.  QUIP_SKINNY_MAX_MN=0 sends every M >= 32 product to the 256 x 256-tile GEMM (the state before the chunked
skinny dispatch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa
dec = D.LlamaDecoder(D.LLAMA2_7B, "E8P12", max_len=2304, device="cuda:0", device_init=True)
graph = "--graph" in sys.argv      # prefill_graph(): the pass replayed from a hipGraph captured per prompt length
run = dec.prefill_graph if graph else dec.prefill
g = torch.Generator().manual_seed(0)
for P in (33, 64, 128, 256, 512, 1024, 2048):
    toks = torch.randint(0, dec.s.vocab, (P,), generator=g).to("cuda:0")
    with torch.no_grad():
        for _ in range(2):
            dec.reset(); run(toks)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            dec.reset()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); run(toks); b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b))
    print(f"prompt {P:5d} tokens: TTFT {best:8.2f} ms" + (" (hipGraph replay)" if graph else ""), flush=True)
