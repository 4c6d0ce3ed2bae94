"""How often do the two largest fp16 logits of a greedy step of the SMALL random-init decoder TIE?  With a tie,
`temperature, top_k=1` (example_generate.py:9-26: logits < pivot are cut, the pivot's ties stay) is a fair race between
the tied tokens, not the arg-max: tests/test_gpu_zz_sampling.py's first assertion then fails by chance."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from quip_for_all_amd.decode import LlamaDecoder, SMALL
runs, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60, 16
ties = runs_with_tie = mismatches = 0
for r in range(runs):
    np.random.seed(1000 + r)              # the K x K factors come from scipy's global generator
    dec = LlamaDecoder(SMALL, max_len=64, device="cuda:0", seed=3)
    dec.reset(7)
    had = False
    with torch.no_grad():
        for _ in range(steps):
            lg = dec.step().float()[0]
            top = torch.topk(lg, 2).values
            if float(top[0]) == float(top[1]):
                ties += 1
                had = True
    runs_with_tie += had
    greedy = dec.generate(steps, first_token=7)
    k1 = dec.generate(steps, first_token=7, temperature=0.6, top_k=1)
    mismatches += not torch.equal(greedy, k1)
    del dec
print(f"{runs} decoders x {steps} greedy steps: {ties} steps with tied top-2 fp16 logits, {runs_with_tie} runs with a tie, "
      f"{mismatches} runs where temperature / top_k=1 differs from greedy")
