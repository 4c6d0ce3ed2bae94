#!/usr/bin/env python3
"""Prefill-sized Hadamard launches (rows = 16 x 2048): time and achieved HBM rate (read + write) per transform."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd  # noqa
from quip_for_all_amd.quant import get_hadK
dev = "cuda"
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
op = torch.ops.quip_lib
RAND = not (len(sys.argv) > 3 and sys.argv[3] == 'tables')   # 'tables': get_hadK(use_rand=False) factors
sizes = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [4096, 11008, 1024, 8192]
for n in sizes:
    had, K, qn = get_hadK(n, RAND)
    if qn != n: continue
    hd = None if had is None else had.to(dev).half().contiguous()
    x = torch.randn(rows, n, device=dev).half(); su = torch.ones(n, device=dev).half()
    ti = t(lambda: op.had_transform_fused(x, n, n, K, hd, True, su, None, None, None, 1.0 / math.sqrt(n // K), None, None, 1e-5, None))
    to = t(lambda: op.had_transform_fused(x, n, n, K, hd, False, None, None, su, None, 1.0, None, None, 1e-5, None))
    gb = 2 * rows * n * 2 / 1e9
    print(f"n={n:6d} K={K:3d} rows={rows}: input side {ti:.3f} ms ({gb / ti:.2f} TB/s r+w)  output side {to:.3f} ms ({gb / to:.2f} TB/s r+w)", flush=True)
    del x
y = torch.empty(rows, 4096, device=dev).half(); z = torch.randn(rows, 4096, device=dev).half()
tc = t(lambda: y.copy_(z))
print(f"torch copy_ of {rows} x 4096 fp16: {tc:.3f} ms ({2 * rows * 4096 * 2 / 1e9 / tc:.2f} TB/s r+w)")
