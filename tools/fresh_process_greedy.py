#!/usr/bin/env python3
"""Cold-start check of the decode step: N fresh processes each build the same random-init model (SMALL, seeded
generators: torch for the weights, numpy's global one for the K x K Hadamard factors) and decode 16 greedy tokens with
the captured step as their FIRST GPU work; a warm process does the same after unrelated GPU work.  All token lists have
to be equal.  usage: fresh_process_greedy.py [N=20]      (child mode: fresh_process_greedy.py --child [warm])"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def child(warm):
    import numpy as np
    import torch
    from quip_for_all_amd.decode import LlamaDecoder, SMALL
    if warm:       # unrelated GPU work first: allocator, kernels, graphs all warm
        x = torch.randn(2048, 2048, device="cuda:0")
        for _ in range(20):
            x = (x @ x).tanh()
        torch.cuda.synchronize()
        np.random.seed(1)
        d0 = LlamaDecoder(SMALL, max_len=64, device="cuda:0", seed=9)
        d0.generate(8, first_token=3)
        del d0
    np.random.seed(20260929)
    dec = LlamaDecoder(SMALL, max_len=64, device="cuda:0", seed=3)
    toks = dec.generate(16, first_token=7).cpu().tolist()
    print("TOKENS " + json.dumps({"tokens": toks, "engine_status": dec.engine_status(), "ffn_engine": bool(dec.ffn_eng)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(len(sys.argv) > 2)
        sys.exit(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    outs = []
    for i in range(n + 1):
        cmd = [sys.executable, os.path.abspath(__file__), "--child"] + (["warm"] if i == n else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("TOKENS ")]
        if r.returncode != 0 or not line:
            print(f"run {i}: FAILED rc={r.returncode}\n{r.stdout[-1500:]}\n{r.stderr[-1500:]}")
            sys.exit(1)
        d = json.loads(line[0][7:])
        outs.append(d)
        print(f"run {i} ({'warm' if i == n else 'fresh'}): status {d['engine_status']} tokens {d['tokens']}", flush=True)
    same = all(o["tokens"] == outs[0]["tokens"] and o["engine_status"] == 0 for o in outs)
    print(f"{n} fresh processes + 1 warm process, all equal: {same}")
    sys.exit(0 if same else 1)
