#!/usr/bin/env python3
"""Determinism soak of the persistent block launch: the same greedy generation several times in one process (and across
the long-context threshold) must give the same tokens every time -- a register copied while its load was in flight, or a
hand-off read too early, shows up as a run that differs.  usage: engine_soak.py [tokens] [repeats]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
bad = 0
for cb, shape in (("E8P12", D.LLAMA2_7B), ("D4", D.LLAMA2_7B), ("E8P12RVQ4B", D.LLAMA2_7B), ("HI", D.LLAMA2_7B),
                  ("E8P12RVQ3B", D.LLAMA2_7B), ("E8P12", D.LLAMA3_8B), ("E8P12", D.LLAMA2_70B)):
    dec = D.LlamaDecoder(shape, cb, max_len=n + 8, device="cuda:0", seed=0, device_init=True)
    assert dec.block_eng
    runs = [dec.generate(n, first_token=11, use_graph=True).cpu() for _ in range(reps)]
    same = all(torch.equal(runs[0], r) for r in runs[1:])
    first = next((i for i in range(n) if any(int(r[i]) != int(runs[0][i]) for r in runs[1:])), None)
    print(f"{cb} (hidden {shape.hidden}, ffn {shape.ffn}, launch shape {getattr(dec, 'eng_shape', 0)}): {reps} x {n} greedy tokens through the block launch: {'identical' if same else 'DIFFERENT from token %d' % first}; "
          f"engine status {dec.engine_status()}; distinct tokens {len(set(runs[0].tolist()))}", flush=True)
    bad += 0 if same else 1
    del dec
    torch.cuda.empty_cache()
sys.exit(bad)
