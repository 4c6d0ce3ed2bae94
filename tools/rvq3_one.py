#!/usr/bin/env python3
"""bs=1 E8P12RVQ3B GEMV launches on distinct weight matrices (pool > Infinity Cache) -- rocprofv3 target
(tools/prof_kernel.sh): the kernel streams the checkpoint's 3-byte codes, so FETCH_SIZE x 2 per launch should be
n * k * 3 / 8 bytes (+ planes).   usage: rvq3_one.py N K [launches]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import quip_for_all_amd as Q  # noqa
n, k = int(sys.argv[1]), int(sys.argv[2])
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = "cuda:0"
cb = Q.codebook.codebook_id["E8P12RVQ3B"](inference=True).to(dev)
g = torch.Generator().manual_seed(0)
pool = [torch.randint(-2 ** 31, 2 ** 31 - 1, (n, k * 3 // 32), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
        for _ in range(max(2, (600 << 20) // (n * k * 3 // 8)))]
x = torch.randn(1, k, device=dev, dtype=torch.float16)
planes = torch.ops.quip_lib.had_transform_planes_fused(x, k, 1, None, True, None, 1.0 / k ** 0.5, None, 1e-5, None,
                                                       cb.planes_resid_scale)
for i in range(launches):
    y = cb.mm_planes(planes, pool[i % len(pool)])
torch.cuda.synchronize()
print(f"done: {n} x {k}, {n * k * 3 // 8 / 1e6:.2f} MB of codes per launch, |y| {float(y.float().abs().mean()):.3f}")
