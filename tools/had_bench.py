#!/usr/bin/env python3
"""Graph-timed Hadamard launches (input side -> planes, output side -> fp16) per Llama dimension,
and the floor for a dependent chain of trivial kernels in a hipGraph."""
import math, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd  # noqa
from quip_for_all_amd.quant import get_hadK

dev = "cuda"
def graph_time(fn, reps=200):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
        for _ in range(reps): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / reps)
    return best

SIZES = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 11008, 8192, 28672, 1024]
for n in SIZES:
    for rand in (True, False):
        try:
            had, K, qn = get_hadK(n, rand)
        except Exception as e:
            print("skip", n, rand, e); continue
        if qn != n: continue
        hd = None if had is None else had.to(dev).half().contiguous()
        x = torch.randn(1, n, device=dev).half(); su = torch.ones(n, device=dev).half(); w = torch.ones(n, device=dev).half()
        g = torch.randn(1, n, device=dev).half(); res = torch.randn(1, n, device=dev).half()
        op = torch.ops.quip_lib
        t_p = graph_time(lambda: op.had_transform_planes_fused(x, n, K, hd, True, su, 1.0 / math.sqrt(n // K), None, 1e-5, None))
        t_pr = graph_time(lambda: op.had_transform_planes_fused(x, n, K, hd, True, su, 1.0 / math.sqrt(n // K), w, 1e-5, None))
        t_pg = graph_time(lambda: op.had_transform_planes_fused(x, n, K, hd, True, su, 1.0 / math.sqrt(n // K), None, 1e-5, g))
        t_o = graph_time(lambda: op.had_transform_fused(x, n, n, K, hd, False, None, None, su, None, 1.0, None, None, 1e-5, None))
        t_or = graph_time(lambda: op.had_transform_fused(x, n, n, K, hd, False, None, None, su, None, 1.0, res, None, 1e-5, None))
        print("n %6d K %4d rand %d : planes %.2f us  +rms %.2f  +gate %.2f | out %.2f  +res %.2f" % (n, K, rand, t_p, t_pr, t_pg, t_o, t_or), flush=True)
y = torch.zeros(64, device=dev)
print("chain of y.add_(1) [64 floats]: %.2f us/kernel" % graph_time(lambda: y.add_(1.0), 500))
