"""Seeded sweep over (codebook, in, out, rows, bias, per-channel scale): QuantLinear.forward on the GPU against the
oracle's parity bound, and -- for 1 < rows < 32, the rows-mode range -- every row bit identical to its bs=1
result.  Shapes mix power-of-two widths, the 43 x 2^e family of Llama's ffn (tall / small Hadamard kernels) and
3 x 2^e (wide K > 1); the draw is fixed, so a failure names a reproducible case."""
import numpy as np
import pytest
import torch

from oracle import quip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
DIMS = [256, 512, 688, 1024, 1408, 1536, 2048, 2752, 4096]
ROWS = [1, 2, 3, 5, 8, 17, 31, 32, 40]


def _cases(n=48):
    rng = np.random.Generator(np.random.PCG64(20260929))
    out = []
    while len(out) < n:
        cb = ["E8P12", "E8P12RVQ4B", "E8P12RVQ3B", "D4", "HI"][rng.integers(0, 5)]
        fin, fout = int(DIMS[rng.integers(0, len(DIMS))]), int(DIMS[rng.integers(0, len(DIMS))])
        if cb == "E8P12RVQ3B" and fin % 32:
            continue
        out.append((cb, fin, fout, int(ROWS[rng.integers(0, len(ROWS))]), bool(rng.integers(0, 2)),
                    bool(rng.integers(0, 4) == 0), len(out)))
    return out


@pytest.mark.parametrize("cbid,fin,fout,M,bias,per_channel,seed", _cases())
def test_forward_sweep(cbid, fin, fout, M, bias, per_channel, seed):
    import quip_for_all_amd as Q
    P = O.make_layer(cbid, fin, fout, seed=1000 + seed, bias=bias, per_channel=per_channel)
    layer = Q.QuantLinear.from_params(P).to(DEV).eval()
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((M, fin)) * rng.uniform(0.2, 3.0)).astype(np.float16)
    xd = torch.from_numpy(x).to(DEV)
    with torch.no_grad():
        y = layer(xd)
        if 1 < M < 32:
            layer.skinny_exact = True       # the exact integer path: every row bit identical to its bs=1 result
            ye = layer(xd)
            for r in sorted({0, M // 2, M - 1}):
                assert torch.equal(ye[r:r + 1], layer(xd[r:r + 1])), (r, "row differs from its bs=1 result")
            layer.skinny_exact = False
            if layer.regime(M) == "rows_exact":     # (beyond one exact pass: the fp16 skinny kernel, inside the bound below)
                assert torch.equal(y, ye)
    What = O.qlinear_dense_weight(P)
    x64 = x.astype(np.float64)
    ref = O.qlinear_forward(P, x64, "exact", What)
    err = np.abs(y.cpu().numpy().astype(np.float64) - ref)
    assert np.all(err <= O.ulp_bound(P, x64, What)), float((err / O.ulp_bound(P, x64, What, ulps=1.0)).max())
