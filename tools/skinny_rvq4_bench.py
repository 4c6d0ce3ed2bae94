#!/usr/bin/env python3
"""E8P12RVQ4B at 1 <= M <= 128 rows, product op alone (graph replay, 50 launches per replay, distinct layers cycled):
bs=1 matrix-core GEMV, exact rows mode (5 rows per pass), the fp16 skinny kernel's RVQ4 mode; E8P12 beside it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quip_for_all_amd import decode as D  # noqa
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
op = torch.ops.quip_lib


def timed(fns, reps=5):
    for f in fns:
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st), torch.cuda.graph(gr, stream=st):
        for f in fns:
            f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); gr.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / len(fns))
    return sorted(ts)[len(ts) // 2]


for fin, fout in ((4096, 4096), (4096, 11008), (11008, 4096)):
    for cbid in ("E8P12", "E8P12RVQ4B"):
        layers = [D.random_quant_linear(fin, fout, cbid, g, dev) for _ in range(24)]     # > 256 MB of codes only for the big ones
        cb = layers[0].codebook
        k = layers[0].q_in_features
        out = []
        for M in (1, 5, 16, 31, 64, 128):
            x = torch.randn(M, k, device=dev, dtype=torch.float16)
            if cbid == "E8P12":
                sk = [(lambda L=L: op.e8p_mm_skinny(x, L.Qidxs, cb.grid_packed_abs)) for L in layers]
            else:
                sk = [(lambda L=L: op.e8prvq4_mm_skinny(x, L.Qidxs, cb.grid_packed_abs, cb.opt_resid_scale)) for L in layers]
            t_sk = timed(sk)
            t_rows = None
            if M <= 31 and (k & (k - 1)) == 0:       # (the exact path's planes: power-of-two widths need no K x K factor here)
                rs = getattr(cb, "planes_resid_scale", 0.0)
                planes = op.had_transform_planes_rows(x, k, 1, None, True, None, 1.0, None, 1e-5, None, rs) if M > 1 else \
                    op.had_transform_planes_fused(x, k, 1, None, True, None, 1.0, None, 1e-5, None, rs)
                rows = [(lambda L=L: (cb.mm_planes_rows(planes, L.Qidxs) if M > 1 else cb.mm_planes(planes, L.Qidxs))) for L in layers]
                t_rows = timed(rows)
            out.append("M=%d skinny %.1f%s" % (M, t_sk, "" if t_rows is None else " exact %.1f" % t_rows))
        print(f"{cbid:11s} {fin:5d}->{fout:5d} (us/launch): " + " | ".join(out), flush=True)
