#!/usr/bin/env python3
"""1 < M < 32: time per product of the exact rows-mode path (passes of up to 5 rows, planes included) and of the
single-pass fp16 skinny kernel (input transform to fp16 included), hipGraph of 50 launches, weights rotated."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for fin, fout in ((4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192), (8192, 28672), (28672, 8192)):
    layers = [D.random_quant_linear(fin, fout, "E8P12", g, dev) for _ in range(max(2, (300 << 20) // (fin * fout // 4)))]
    for M in (1, 2, 5, 8, 16, 31):
        x = torch.randn(M, fin, device=dev, dtype=torch.float16)
        res = {}
        for name, exact in (("exact rows mode", True), ("default", False)):
            for l in layers:
                l.skinny_exact = exact
            def run():
                for l in layers:
                    l(x)
            with torch.no_grad():
                run(); torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s), torch.cuda.graph(gr, stream=s):
                    run()
                torch.cuda.synchronize()
                ts = []
                for _ in range(4):
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
                    ts.append(a.elapsed_time(b) * 1e3 / len(layers))
                del gr
            res[name] = sorted(ts)[1]
        print(f"{fin:5d}->{fout:5d} M={M:2d}: QuantLinear.forward exact rows mode {res['exact rows mode']:7.2f} us | default "
              f"{res['default']:7.2f} us", flush=True)
    del layers
    torch.cuda.empty_cache()
