"""GPU parity tests of the quip_lib ops against the CPU oracle (run with -m gpu on
an MI355X).  Every call goes through torch.ops.quip_lib -> ctypes -> C ABI -> HIP."""
import numpy as np
import pytest
import torch

from oracle import quip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def Q():
    assert torch.cuda.is_available()
    import quip_for_all_amd as Q
    return Q


def _cb(Q, cbid, scale=None):
    kw = {}
    if scale is not None:
        kw["opt_resid_scale"] = scale
    return Q.codebook.codebook_id[cbid](inference=True, **kw).to(DEV)


def _mm_tol(x64, W64, y64):
    """|y - y_fp64| <= 2^-10 |y| (fp16 RN of the result, 1 ulp) + 2^-21 sum|w x|
    (fp32 accumulation in any order) -- SURVEY section 7 parity bound."""
    absdot = np.abs(x64) @ np.abs(W64).T
    return 2.0 ** -10 * np.abs(y64) + 2.0 ** -21 * absdot + 1e-7


CODEBOOKS = [("E8P12", None), ("E8P12RVQ4B", 1 / 3.45), ("E8P12RVQ4B", -1.0), ("E8P12RVQ3B", 1 / 2.04),
             ("D4", None), ("HI", None)]


@pytest.mark.parametrize("cbid,scale", CODEBOOKS)
@pytest.mark.parametrize("n,k", [(256, 256), (40, 1024), (688, 256), (256, 2752)])
def test_decompress_bit_exact(Q, cbid, scale, n, k):
    if cbid == "E8P12RVQ3B" and k % 32:
        pytest.skip("RVQ3 needs k % 32 == 0")
    P = O.make_layer(cbid, k, n, seed=n + k, resid_scale=scale)
    cb = _cb(Q, cbid, scale)
    W = cb.decompress_weight(torch.from_numpy(P.Qidxs).to(DEV))
    ref = O.decompress(cbid, P.Qidxs, P.resid_scale)
    assert W.shape == ref.shape and W.dtype == torch.float16
    np.testing.assert_array_equal(W.cpu().numpy().view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("cbid,scale", CODEBOOKS)
@pytest.mark.parametrize("m", [1, 2, 15, 16, 17, 31])
@pytest.mark.parametrize("n,k", [(256, 256), (264, 704), (16, 4096)])
def test_mm_small(Q, cbid, scale, m, n, k):
    P = O.make_layer(cbid, k, n, seed=11 * m + n, resid_scale=scale)
    cb = _cb(Q, cbid, scale)
    rng = np.random.default_rng(m * 1000 + n)
    x = rng.standard_normal((m, k)).astype(np.float16)
    y = cb.mm(torch.from_numpy(x).to(DEV), torch.from_numpy(P.Qidxs).to(DEV))
    assert y.shape == (m, n) and y.dtype == torch.float16
    W64 = O.decompress(cbid, P.Qidxs, P.resid_scale).astype(np.float64)
    y64 = x.astype(np.float64) @ W64.T
    err = np.abs(y.cpu().numpy().astype(np.float64) - y64)
    assert np.all(err <= _mm_tol(x.astype(np.float64), W64, y64)), err.max()


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (4096, 11008), (1024, 8192), (300, 8192),
                                 (4100, 4096), (64, 64), (8, 28672)])
def test_e8p_gemv_m1_fast_path(Q, n, k):
    """bs=1 decode GEMV (the headline kernel), Llama-7B/70B-shaped rows and ragged N."""
    P = O.make_layer("E8P12", k, n, seed=n ^ k)
    cb = _cb(Q, "E8P12")
    rng = np.random.default_rng(n + k)
    x = rng.standard_normal((1, k)).astype(np.float16)
    y = cb.mm(torch.from_numpy(x).to(DEV), torch.from_numpy(P.Qidxs).to(DEV))
    W64 = O.decompress_e8p(P.Qidxs).astype(np.float64)
    y64 = x.astype(np.float64) @ W64.T
    err = np.abs(y.cpu().numpy().astype(np.float64) - y64)
    assert np.all(err <= _mm_tol(x.astype(np.float64), W64, y64)), err.max()


@pytest.mark.parametrize("rep,rows,blocks,g", [(1, 1, 0, 0), (1, 2, 0, 0), (1, 4, 0, 0), (16, 1, 0, 0),
                                               (16, 2, 0, 0), (16, 4, 0, 0), (16, 4, 64, 2), (1, 2, 512, 1),
                                               (16, 2, 100, 3)])
@pytest.mark.parametrize("n,k", [(1000, 4096), (512, 11008), (96, 8192)])
def test_e8p_gemv_variants(Q, rep, rows, blocks, g, n, k):
    """every tuning variant of the GEMV computes the same thing"""
    from quip_for_all_amd import capi
    P = O.make_layer("E8P12", k, n, seed=5)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((1, k)).astype(np.float16)).to(DEV)
    Qd = torch.from_numpy(P.Qidxs).to(DEV)
    grid = _cb(Q, "E8P12").grid_packed_abs
    y = torch.full((1, n), float("nan"), dtype=torch.float16, device=DEV)
    rc = capi.lib().quip_e8p_gemv_tuned(x.data_ptr(), Qd.data_ptr(), grid.data_ptr(), y.data_ptr(), n, k,
                                        rep, rows, blocks, g, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    W64 = O.decompress_e8p(P.Qidxs).astype(np.float64)
    x64 = x.cpu().numpy().astype(np.float64)
    y64 = x64 @ W64.T
    err = np.abs(y.cpu().numpy().astype(np.float64) - y64)
    assert np.all(err <= _mm_tol(x64, W64, y64)), err.max()


def test_e8p_gemv_special_inputs(Q):
    """zero x, one-hot x (reads back a decoded column exactly), constant codes"""
    n, k = 512, 4096
    cb = _cb(Q, "E8P12")
    P = O.make_layer("E8P12", k, n, seed=9)
    Qd = torch.from_numpy(P.Qidxs).to(DEV)
    W = O.decompress_e8p(P.Qidxs)
    y = cb.mm(torch.zeros(1, k, dtype=torch.float16, device=DEV), Qd)
    assert torch.count_nonzero(y) == 0
    for col in (0, 7, 63, 64, 4095, 2049):
        x = torch.zeros(1, k, dtype=torch.float16, device=DEV)
        x[0, col] = 1.0
        y = cb.mm(x, Qd)
        np.testing.assert_array_equal(y.cpu().numpy()[0], W[:, col])
    for code in (0, 1, 0x0101, 0xFFFF, 0x8000):
        Qc = torch.full((n, k // 8), code, dtype=torch.int32).to(torch.int16).to(DEV) if code < 0x8000 else \
            torch.full((n, k // 8), code - 65536, dtype=torch.int16, device=DEV)
        x = torch.ones(1, k, dtype=torch.float16, device=DEV)
        y = cb.mm(x, Qc)
        w = O.e8p_decode_i8(np.array([code], dtype=np.uint16))[0].astype(np.float64) / 4
        assert abs(float(y[0, 0]) - w.sum() * (k // 8)) <= 2 ** -10 * abs(w.sum() * (k // 8)) + 1e-3


@pytest.mark.parametrize("n", [1, 2, 64, 256, 1024, 4096, 8192, 32768])
@pytest.mark.parametrize("rows", [1, 3])
def test_hadamard_op(Q, n, rows):
    rng = np.random.default_rng(n)
    x = rng.standard_normal((rows, n)).astype(np.float16)
    s = 1.0 / np.sqrt(n)
    y = torch.ops.quip_lib.hadamard(torch.from_numpy(x).to(DEV), s)
    ref = O.fwht(x.astype(np.float64)) * s
    err = np.abs(y.cpu().numpy().astype(np.float64) - ref)
    assert np.all(err <= 2.0 ** -10 * np.abs(ref) + 2.0 ** -20 * np.linalg.norm(ref, axis=1, keepdims=True) + 1e-7)


def test_hadamard_op_3d_noncontiguous(Q):
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((64, 6)).astype(np.float16)).to(DEV).T  # (6, 64) strided
    y = torch.ops.quip_lib.hadamard(x, 0.125)
    ref = O.fwht(x.cpu().numpy().astype(np.float64)) * 0.125
    np.testing.assert_allclose(y.cpu().numpy(), ref, atol=2e-2, rtol=2e-3)
    x3 = torch.from_numpy(rng.standard_normal((2, 43, 256)).astype(np.float16)).to(DEV)
    y3 = torch.ops.quip_lib.hadamard(x3, 1.0 / 16)
    ref3 = O.fwht(x3.cpu().numpy().astype(np.float64)) / 16
    np.testing.assert_allclose(y3.cpu().numpy(), ref3, atol=2e-2, rtol=2e-3)


@pytest.mark.parametrize("n,K", [(688, 43), (11008, 43), (1792, 7), (28672, 7), (4096, 1)])
@pytest.mark.parametrize("transpose", [False, True])
def test_matmul_hadU_cuda(Q, golden, n, K, transpose):
    """quant.matmul_hadU[t]_cuda vs oracle (and vs the reference butterfly goldens)"""
    from quip_for_all_amd import quant
    rng = np.random.default_rng(n + K)
    key = f"fht_{n}_x"
    x = golden[key].astype(np.float16) if key in golden else rng.standard_normal((3, n)).astype(np.float16)
    hadK = None
    if K > 1:
        hadK = (golden[f"fht_{n}_hadK"] if f"fht_{n}_hadK" in golden else O.random_orthogonal(K, rng)).astype(np.float16)
    y = quant.matmul_hadU_cuda(torch.from_numpy(x).to(DEV), None if hadK is None else torch.from_numpy(hadK).to(DEV),
                               K, n, transpose=transpose)
    ref = O.matmul_hadU(x.astype(np.float64), None if hadK is None else hadK.astype(np.float64), K, n,
                        transpose=transpose)
    err = np.abs(y.cpu().numpy().astype(np.float64) - ref)
    assert np.all(err <= 2.0 ** -10 * np.abs(ref) + 2.0 ** -18 * np.linalg.norm(ref, axis=1, keepdims=True) + 1e-6)


def test_ops_reject_bad_arguments(Q):
    cb = _cb(Q, "E8P12")
    x = torch.zeros(1, 256, dtype=torch.float16, device=DEV)
    q = torch.zeros(8, 32, dtype=torch.int16, device=DEV)
    with pytest.raises(ValueError):
        torch.ops.quip_lib.e8p_mm_origorder(x.float(), q, cb.grid_packed_abs)       # dtype
    with pytest.raises(ValueError):
        torch.ops.quip_lib.e8p_mm_origorder(x[:, :128], q, cb.grid_packed_abs)      # k mismatch
    with pytest.raises(ValueError):
        torch.ops.quip_lib.e8p_mm_origorder(x, q.int(), cb.grid_packed_abs)          # index dtype
    with pytest.raises(ValueError):
        torch.ops.quip_lib.hadamard(torch.zeros(2, 24, dtype=torch.float16, device=DEV), 1.0)
    y = torch.ops.quip_lib.e8p_mm_origorder(torch.zeros(0, 256, dtype=torch.float16, device=DEV), q,
                                            cb.grid_packed_abs)
    assert y.shape == (0, 8)
