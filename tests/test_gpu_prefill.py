"""Fused E8P12 dequant + MFMA GEMM for batches (csrc/e8p_prefill_gemm.hip; replaces the reference's M >= 32
decompress + dense GEMM, codebook/e8p12.py:152-155, origin_order.cu:837-885).  Bound: the float64 product of the
SAME fp16 operands, one fp16 rounding of the result + fp32 accumulation in any order (the MFMA's and cuBLAS's orders
differ too)."""
import numpy as np
import pytest
import torch

from oracle import quip_oracle as O
from tests.test_gpu_ops import DEV, _cb

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Q():
    assert torch.cuda.is_available()
    import quip_for_all_amd as Q
    return Q


def _tol(x64, W64, y64):
    absdot = np.abs(x64) @ np.abs(W64).T
    return 2.0 ** -10 * np.abs(y64) + 2.0 ** -21 * absdot + 1e-7


@pytest.mark.parametrize("m", [1, 32, 40, 255, 256, 257, 700])
@pytest.mark.parametrize("n,k", [(64, 64), (256, 256), (30, 128), (290, 704), (512, 4096), (96, 11008)])
def test_batched_product_against_oracle(Q, m, n, k):
    P = O.make_layer("E8P12", k, n, seed=m + n + k)
    rng = np.random.default_rng(m * 7 + n)
    x = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float16)).to(DEV)
    Qd = torch.from_numpy(P.Qidxs).to(DEV)
    cb = _cb(Q, "E8P12")
    y = torch.ops.quip_lib.e8p_mm_batched(x, Qd, cb.grid_packed_abs)
    assert y.shape == (m, n) and y.dtype == torch.float16
    W64 = O.decompress_e8p(P.Qidxs).astype(np.float64)
    x64 = x.cpu().numpy().astype(np.float64)
    y64 = x64 @ W64.T
    err = np.abs(y.cpu().numpy().astype(np.float64) - y64)
    assert np.all(err <= _tol(x64, W64, y64)), (err.max(), np.unravel_index(err.argmax(), err.shape))


def test_codebook_forward_paths_agree(Q):
    """E8P12_codebook.forward for M >= 32 beyond the skinny regime: decompress + dense GEMM by default (the faster one on
    MI355X), the fused kernel under QUIP_BATCHED_MM=fused; the two agree within fp32 accumulation order, and either is
    exactly linear in x's rows (a row's result does not depend on which other rows are in the batch)"""
    cb = _cb(Q, "E8P12")
    g = torch.Generator().manual_seed(5)
    n, k, m = 1024, 2048, 1400
    Qd = torch.randint(-32768, 32768, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(DEV)
    x = torch.randn(m, k, generator=g).half().to(DEV)
    W = cb.decompress_weight(Qd)
    ref = (x.float() @ W.float().T)
    bound = 2.0 ** -10 * ref.abs() + 2.0 ** -19 * (x.float().abs() @ W.float().abs().T)
    saved = type(cb).batched_mode
    out = {}
    try:
        for mode in ("auto", "fused", "reference"):
            type(cb).batched_mode = mode
            assert cb.batched_regime(m, n, k) == {"auto": "decompress_gemm", "fused": "fused_gemm", "reference": "decompress_gemm"}[mode]
            y = cb(x, Qd)
            assert torch.all((y.float() - ref).abs() <= bound), mode
            y2 = cb(x[37:37 + 1250].contiguous(), Qd)
            if mode == "fused":
                assert torch.equal(y[37:37 + 1250], y2), "rows are independent of their position in the batch"
            else:   # the vendor GEMM may pick another kernel for another M: same bound, not the same bits
                assert torch.all((y2.float() - ref[37:37 + 1250]).abs() <= bound[37:37 + 1250])
            out[mode] = y
    finally:
        type(cb).batched_mode = saved
    assert torch.all((out["fused"].float() - out["auto"].float()).abs() <= 2.0 ** -9 * ref.abs() + 2.0 ** -18 * (x.float().abs() @ W.float().abs().T))


@pytest.mark.parametrize("mode", ["auto", "fused"])
@pytest.mark.parametrize("fin,fout", [(4096, 11008), (11008, 4096)])
def test_config5_full_size(Q, fin, fout, mode):
    """BASELINE configs[4] at its real size: M = 16 x 2048 rows through QuantLinear.forward (batch Hadamard kernels +
    decompress + dense GEMM by default / the fused dequant GEMM).  Sampled rows against the float64 oracle of the whole
    module; with the fused kernel also row-against-sub-batch bit identity of the batch path (the vendor GEMM may choose
    another kernel for another M: those rows are checked against the oracle bound instead)."""
    from quip_for_all_amd.qlinear import QuantLinear
    P = O.make_layer("E8P12", fin, fout, seed=17)
    layer = QuantLinear.from_params(P).to(DEV).eval()
    cbt = type(layer.codebook)
    saved = cbt.batched_mode
    cbt.batched_mode = mode
    try:
        M = 16 * 2048
        assert layer.codebook.batched_regime(M, layer.q_out_features, layer.q_in_features) == ("fused_gemm" if mode == "fused" else "decompress_gemm")
        g = torch.Generator(device=DEV).manual_seed(3)
        x = torch.randn(M, fin, generator=g, device=DEV, dtype=torch.float16)
        with torch.no_grad():
            y = layer(x)
        assert y.shape == (M, fout) and bool(torch.isfinite(y).all())
        rows = [0, 1, 255, 256, 4097, 20000, M - 1]
        xs = x[rows].cpu().numpy()
        yo = O.qlinear_forward(P, xs, mode="exact")
        bound = O.ulp_bound(P, xs)
        err = np.abs(y[rows].cpu().numpy().astype(np.float64) - yo)
        assert np.all(err <= bound), float((err / bound).max())
        # (a sub-batch large enough to stay on the same kernel: up to a few hundred rows QuantLinear takes the
        #  single-pass skinny kernel on chunks of 32 rows, whose fp32 sums run in another order)
        with torch.no_grad():
            ysub = layer(x[4096:4096 + 2048].contiguous())
        if mode == "fused":
            assert torch.equal(ysub, y[4096:4096 + 2048]), "a row's result does not depend on the batch it is in"
        else:
            xr = x[4096:4096 + 8].cpu().numpy()
            errs = np.abs(ysub[:8].cpu().numpy().astype(np.float64) - O.qlinear_forward(P, xr, mode="exact"))
            assert np.all(errs <= O.ulp_bound(P, xr))
        with torch.no_grad():
            y40 = layer(x[4096:4096 + 40].contiguous())      # chunked skinny kernel: same rows inside the same bound
        xs40 = x[4096:4096 + 40].cpu().numpy()
        err40 = np.abs(y40.cpu().numpy().astype(np.float64) - O.qlinear_forward(P, xs40, mode="exact"))
        assert np.all(err40 <= O.ulp_bound(P, xs40))
    finally:
        cbt.batched_mode = saved


@pytest.mark.parametrize("cbid", ["E8P12RVQ4B", "E8P12RVQ3B", "D4", "HI"])
@pytest.mark.parametrize("m,n,k", [(1, 64, 64), (40, 256, 256), (257, 30, 128), (300, 290, 704), (700, 512, 4096),
                                   (255, 96, 11008), (2048, 1024, 1024)])
def test_batched_product_other_codebooks(Q, cbid, m, n, k):
    """the fused tile kernel in the other codebooks' modes (csrc/e8p_prefill_gemm.hip MODE 1..4; the M >= 32 role of
    decompress_* + `x @ W.T`, e8p12_rvq4.py:50-67, e8p12_rvq3.py:109-129, d4.py:128-139, hi.py:52-63): the weights are the
    dense W of the reference's decompress op bit for bit (identity rows: every output is one weight, exactly), the
    product agrees with float64 of the same fp16 operands, a row's result does not depend on its batch"""
    from quip_for_all_amd.qlinear import QuantLinear
    P = O.make_layer(cbid, k, n, seed=m + n + k)
    layer = QuantLinear.from_params(P).to(DEV).eval()
    cb, Qd = layer.codebook, layer.Qidxs
    k, n = layer.q_in_features, layer.q_out_features
    if k % 64 or n % 2:
        pytest.skip("k % 64 != 0: decompress + dense GEMM")
    rng = np.random.default_rng(m * 7 + n)
    x = torch.from_numpy(rng.standard_normal((m, k)).astype(np.float16)).to(DEV)
    y = cb.mm_batched(x, Qd)
    assert y.shape == (m, n) and y.dtype == torch.float16
    Wd = cb.decompress_weight(Qd)
    W64 = Wd.cpu().numpy().astype(np.float64)
    assert np.array_equal(W64, O.decompress(cbid, P.Qidxs, getattr(cb, "opt_resid_scale", 0.0)).astype(np.float64))
    x64 = x.cpu().numpy().astype(np.float64)
    y64 = x64 @ W64.T
    err = np.abs(y.cpu().numpy().astype(np.float64) - y64)
    assert np.all(err <= _tol(x64, W64, y64)), (err.max(), np.unravel_index(err.argmax(), err.shape))
    cols = sorted({0, 1, 3, 4, 7, 8, 31, 32, 33, k // 2 + 5, k - 8, k - 1})
    e = torch.zeros(len(cols), k, dtype=torch.float16, device=DEV)
    for i, c in enumerate(cols):
        e[i, c] = 1.0
    assert torch.equal(cb.mm_batched(e, Qd), Wd[:, cols].T.contiguous()), "every weight exactly as decompress writes it"
    if m > 40:
        assert torch.equal(y[m - 33:], cb.mm_batched(x[m - 33:].contiguous(), Qd)), "rows do not depend on their batch"


@pytest.mark.parametrize("cbid", ["E8P12RVQ4B", "E8P12RVQ3B", "D4", "HI"])
def test_module_forward_fused_mode_other_codebooks(Q, cbid):
    """QuantLinear.forward of the other codebooks beyond the skinny regime: decompress + dense GEMM by default, the fused
    kernel under QUIP_BATCHED_MM=fused; both inside the module's stated bound against the float64 oracle"""
    from quip_for_all_amd.qlinear import QuantLinear
    from quip_for_all_amd.codebook.codebooks import E8P12_codebook
    fin, fout, M = 1024, 2048, 1500
    P = O.make_layer(cbid, fin, fout, seed=23)
    layer = QuantLinear.from_params(P).to(DEV).eval()
    g = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(M, fin, generator=g, device=DEV, dtype=torch.float16)
    rows = [0, 1, 31, 32, 255, 256, 1000, M - 1]
    xs = x[rows].cpu().numpy()
    yo = O.qlinear_forward(P, xs, mode="exact")
    bound = O.ulp_bound(P, xs)
    saved = E8P12_codebook.batched_mode
    try:
        for mode in ("auto", "fused"):
            E8P12_codebook.batched_mode = mode
            assert layer.codebook.batched_regime(M, layer.q_out_features, layer.q_in_features) == \
                ("fused_gemm" if mode == "fused" else "decompress_gemm")
            with torch.no_grad():
                y = layer(x)
            err = np.abs(y[rows].cpu().numpy().astype(np.float64) - yo)
            assert np.all(err <= bound), (cbid, mode, float((err / bound).max()))
    finally:
        E8P12_codebook.batched_mode = saved
