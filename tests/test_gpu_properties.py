"""Size-independent properties of the hot path at BASELINE's full layer sizes (Llama-2-7B and 70B),
where a float64 oracle product would take too long: exact homogeneity under power-of-two scaling,
row-permutation equivariance, sub-matrix consistency, additivity within rounding, the transform's
involution, and decode -> re-quantise idempotence of the packed format."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SHAPES = [(4096, 4096), (11008, 4096), (4096, 11008), (8192, 8192), (1024, 8192), (28672, 8192), (8192, 28672)]


def _setup(n, k, seed):
    import quip_for_all_amd as Q
    cb = Q.codebook.codebook_id["E8P12"](inference=True).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(seed)
    q = torch.randint(-32768, 32768, (n, k // 8), generator=g, dtype=torch.int32, device=DEV).to(torch.int16)
    x = torch.randn(1, k, generator=g, device=DEV).half()
    return Q, cb, q, x


def _gemv(Q, cb, q, x):
    L = Q.capi.lib()
    k = x.shape[1]
    planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=DEV)
    Q.capi.check(L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, torch.cuda.current_stream().cuda_stream), "planes")
    return cb.mm_planes(planes, q)


@pytest.mark.parametrize("n,k", SHAPES)
def test_gemv_exact_structure(n, k):
    Q, cb, q, x = _setup(n, k, n + k)
    y = _gemv(Q, cb, q, x)
    assert torch.isfinite(y).all()
    # (1) power-of-two homogeneity is EXACT (block exponent shifts, integers unchanged)
    y4 = _gemv(Q, cb, q, (x * 4).half())
    assert torch.equal(y4.float(), y.float() * 4) or torch.equal(y4, (y.float() * 4).half())
    # (2) permuting the weight rows permutes the outputs exactly
    perm = torch.randperm(n, device=DEV)
    assert torch.equal(_gemv(Q, cb, q[perm].contiguous(), x), y[:, perm])
    # (3) a row block computed alone equals the same rows of the full product (no cross-row coupling)
    lo, hi = n // 3, n // 3 + 257
    assert torch.equal(_gemv(Q, cb, q[lo:hi].contiguous(), x), y[:, lo:hi])
    # (4) zero input -> exact zeros; sign flip -> exact negation
    assert torch.count_nonzero(_gemv(Q, cb, q, torch.zeros_like(x))) == 0
    assert torch.equal(_gemv(Q, cb, q, -x), -y)


@pytest.mark.parametrize("n,k", SHAPES[:5])
def test_gemv_additivity_and_dense_agreement(n, k):
    """y(x1 + x2) = y(x1) + y(x2) within the fp16 rounding of the three outputs, and the GEMV agrees with
    decompress + dense fp32 matmul on the GPU (two independent kernels)"""
    Q, cb, q, x1 = _setup(n, k, 7 * n + k)
    x2 = torch.randn(1, k, device=DEV).half()
    xs = (x1.float() + x2.float()).half()
    y1, y2, ys = (_gemv(Q, cb, q, t).float() for t in (x1, x2, xs))
    W = cb.decompress_weight(q).float()
    ref = xs.float() @ W.T
    tol = 2.0 ** -9 * ref.abs() + 2.0 ** -10 * (xs.float().abs() @ W.abs().T) / k ** 0.5 + 1e-3
    assert ((ys - ref).abs() <= tol).all()
    xs_exact = x1.float() + x2.float()     # ys used the fp16-rounded sum: allow that rounding too
    slack = ((xs.float() - xs_exact).abs() @ W.abs().T)
    assert ((ys - (y1 + y2)).abs() <= 3 * 2.0 ** -10 * (y1.abs() + y2.abs() + ys.abs()) + slack + 1e-3).all()


@pytest.mark.parametrize("n", [4096, 8192, 11008, 28672])
def test_hadamard_involution_full_width(n):
    """(H_K (x) H_L)/sqrt(L) applied forward and transposed-back is the identity for orthogonal H_K"""
    import quip_for_all_amd  # noqa: F401
    from quip_for_all_amd.quant import get_hadK, matmul_hadU_cuda, matmul_hadUt_cuda
    torch.manual_seed(n)
    had, K, qn = get_hadK(n, True)
    x = torch.randn(3, n, device=DEV).half()
    y = matmul_hadU_cuda(x, had, K, qn)
    back = matmul_hadUt_cuda(y, had, K, qn)
    err = (back.float() - x.float()).abs().max().item()
    assert err <= 6e-3 * (1 + (0 if had is None else 3)), err       # two fp16 materialisations (+ fp16 hadK)
    # energy is preserved by an orthogonal transform
    assert abs(y.float().norm().item() / x.float().norm().item() - 1) < 2e-3


@pytest.mark.parametrize("cbid", ["E8P12", "D4", "HI", "E8P12RVQ4B", "E8P12RVQ3B"])
def test_decompress_is_a_pure_lookup(cbid):
    """decode depends on each code alone: decoding a matrix equals decoding its rows / shuffled rows"""
    import quip_for_all_amd as Q
    cb = Q.codebook.codebook_id[cbid](inference=True).to(DEV)
    layer = Q.QuantLinear(4096, 512, cb, bias=False).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    qi = layer.Qidxs
    if qi.dtype == torch.uint8:
        codes = torch.randint(0, 256, qi.shape, generator=g, device=DEV, dtype=torch.int32).to(torch.uint8)
    elif qi.dtype == torch.int16:
        codes = torch.randint(-32768, 32768, qi.shape, generator=g, device=DEV, dtype=torch.int32).to(torch.int16)
    else:
        codes = torch.randint(-2 ** 31, 2 ** 31 - 1, qi.shape, generator=g, device=DEV, dtype=torch.int64).to(torch.int32)
    W = cb.decompress_weight(codes)
    perm = torch.randperm(codes.shape[0], device=DEV)
    assert torch.equal(cb.decompress_weight(codes[perm].contiguous()), W[perm])
    assert torch.equal(cb.decompress_weight(codes[100:101].contiguous()), W[100:101])
