import ctypes, math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quip_for_all_amd  # noqa
from quip_for_all_amd import capi
from quip_for_all_amd.quant import get_hadK
L = capi.lib()
dev = "cuda"
n, rows = 11008, 32768
had, K, _ = get_hadK(n, True)
hd = had.to(dev).half().contiguous()
x = torch.randn(rows, n, device=dev).half(); sv = torch.ones(n, device=dev).half()
op = torch.ops.quip_lib
for _ in range(3):
    op.had_transform_fused(x, n, n, K, hd, False, None, None, sv, None, 1.0, None, None, 1e-5, None)
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 16)()
assert L.quip_had_read_stamps(out) == 0
s = np.array(list(out), dtype=np.int64)
print("row iteration 2 of WG 0 (ticks): stage", s[1]-s[0], "K-mix", s[2]-s[1], "fht+epilogue", s[3]-s[2], "barrier", s[4]-s[3], "total", s[4]-s[0])
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): op.had_transform_fused(x, n, n, K, hd, False, None, None, sv, None, 1.0, None, None, 1e-5, None)
b.record(); torch.cuda.synchronize()
print("ms per transform (with stamps build):", a.elapsed_time(b)/5)
