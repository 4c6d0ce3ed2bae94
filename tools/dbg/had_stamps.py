import ctypes, math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quip_for_all_amd  # noqa
from quip_for_all_amd import capi
from quip_for_all_amd.quant import get_hadK
L = capi.lib()
dev = "cuda"
names = ["hs staged", "row staged", "K-mix", "transpose(tall)/load", "fht", "epilogue"]
def stamps():
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    assert L.quip_had_read_stamps(out) == 0
    return np.array(list(out), dtype=np.int64)
for n in (4096, 11008):
    had, K, _ = get_hadK(n, True)
    hd = None if had is None else had.to(dev).half().contiguous()
    x = torch.randn(1, n, device=dev).half(); su = torch.ones(n, device=dev).half(); g = torch.randn(1, n, device=dev).half()
    op = torch.ops.quip_lib
    for what in ("planes+gate", "out"):
        for _ in range(3):
            if what == "out":
                op.had_transform_fused(x, n, n, K, hd, False, None, None, su, None, 1.0, None, None, 1e-5, None)
            else:
                op.had_transform_planes_fused(x, n, K, hd, True, su, 1.0 / math.sqrt(n // K), None, 1e-5, g)
        s = stamps()
        d = np.diff(s[:7])
        if K == 1:
            print(n, what, "total", s[6] - s[0], " load+prep", s[4] - s[0], "fht", d[4], "epilogue", d[5])
        else:
            print(n, what, "total", s[6] - s[0], dict(zip(names, d.tolist())))
