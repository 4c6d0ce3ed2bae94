"""s_memtime stamps inside the Hadamard chain launch (needs tools/dbg/libquip_stamps.so built with
-DQUIP_HAD_STAMPS; run with QUIP_LIB_PATH pointing at it)."""
import ctypes, math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quip_for_all_amd  # noqa
from quip_for_all_amd import capi
L = capi.lib()
dev = "cuda"
def stamps():
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    assert L.quip_had_read_stamps(out) == 0
    return np.array(list(out), dtype=np.int64)
def graph_time(fn, reps=100):
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
op = torch.ops.quip_lib
for n in (4096, 8192):
    z = torch.randn(1, n, device=dev).half(); post = torch.ones(n, device=dev).half(); res = torch.randn(1, n, device=dev).half()
    su = [torch.ones(n, device=dev).half() for _ in range(3)]; w = torch.ones(n, device=dev).half()
    for cons in (1, 3):
        fn = lambda: op.had_chain_planes_group(z, post, res, 1.0 / math.sqrt(n), n, su[:cons], [1.0 / math.sqrt(n)] * cons, w, 1e-5)
        for _ in range(3): fn()
        s = stamps()
        t = graph_time(fn)
        print(f"chain n={n} consumers={cons}: {t:.2f} us/launch; ticks: start->z loaded+issued {s[7]-s[0]}, fht(z) {s[8]-s[7]}, "
              f"out_elem+store+elementwise+bar {s[9]-s[8]}, rms reduce {s[4]-s[9]}, fht(x) {s[5]-s[4]}, planes epilogue {s[6]-s[5]}, total {s[6]-s[0]}")
    x = torch.randn(1, n, device=dev).half()
    fn = lambda: op.had_transform_planes_fused(x, n, 1, None, True, su[0], 1.0 / math.sqrt(n), w, 1e-5, None)
    for _ in range(3): fn()
    s = stamps(); t = graph_time(fn)
    print(f"planes+rms n={n}: {t:.2f} us; ticks: load+prep+reduce {s[4]-s[0]}, fht {s[5]-s[4]}, epilogue {s[6]-s[5]}, total {s[6]-s[0]}")
    fn = lambda: op.had_transform_fused(x, n, n, 1, None, False, None, None, su[0], None, 1.0, res, None, 1e-5, None)
    for _ in range(3): fn()
    s = stamps(); t = graph_time(fn)
    print(f"out+res n={n}: {t:.2f} us; ticks: load+prep {s[4]-s[0]}, fht {s[5]-s[4]}, epilogue {s[6]-s[5]}, total {s[6]-s[0]}")
