#!/usr/bin/env python3
"""Run one launch configuration of the second-generation GEMV (or, with kernel=1, the first) a few times over a
rotating weight pool -- the target process of tools/prof_gemv_v2.sh (rocprofv3).
usage: gemv_v2_one.py N K kernel rep slots blocks ksplit max_waves runlen [launches]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa: E402
from quip_for_all_amd import capi  # noqa: E402

n, k, kern, rep, slots, blocks, ksplit, maxw, runlen = [int(v) for v in sys.argv[1:10]]
launches = int(sys.argv[10]) if len(sys.argv) > 10 else 40
dev = "cuda:0"
L = capi.lib()
g = torch.Generator().manual_seed(0)
npool = max(2, (600 << 20) // (n * k // 4))
pool = [torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(dev)
        for _ in range(npool)]
x = torch.randn(1, k, generator=g).half().to(dev)
y = torch.empty(1, n, dtype=torch.float16, device=dev)
grid = Q.codebook.codebook_id["E8P12"](inference=True).to(dev).grid_packed_abs
planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=dev)
ws = torch.zeros(L.quip_e8p_gemv_v2_workspace_bytes(n) // 4, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream
L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, st)
for i in range(launches):
    if kern == 1:
        rc = L.quip_e8p_gemv_tuned(planes.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(), y.data_ptr(), n, k,
                                   4, 0, 0, 0, 0, 0, 0, None, st)
    else:
        rc = L.quip_e8p_gemv_v2_tuned(planes.data_ptr(), pool[i % npool].data_ptr(), grid.data_ptr(), y.data_ptr(),
                                      ws.data_ptr(), n, k, rep, slots, blocks, ksplit, maxw, runlen, None, st)
    assert rc == 0, rc
torch.cuda.synchronize()
print("done", y.float().abs().mean().item())
