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


def _mm_tol(x64, W64, y64, xbits=22):
    """|y - y_fp64| <= 2^-10 |y|          fp16 RN of the result (1 ulp)
                     + 2^-21 sum|w x|     fp32 accumulation in any order (fp16-domain kernels,
                                          the reference's tensor-core accumulation)
                     + 2^-xbits max|x| sum|w|   block fixed-point x of the integer-domain GEMV
                                          (|X| < 2^22: elements within 2^-11 of max|x| are exact)
    -- SURVEY section 7 parity bound + DESIGN.md."""
    absdot = np.abs(x64) @ np.abs(W64).T
    fx = 2.0 ** -xbits * np.abs(x64).max(axis=1, keepdims=True) * np.abs(W64).sum(axis=1)[None, :]
    return 2.0 ** -10 * np.abs(y64) + 2.0 ** -21 * absdot + fx + 1e-7


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


@pytest.mark.parametrize("kernel,rep,rows,blocks,g,maxw,digits", [
    (4, 0, 1, 0, 0, 16, 0), (4, 0, 2, 0, 0, 16, 0), (4, 0, 3, 0, 0, 12, 0), (4, 16, 2, 0, 0, 16, 0),
    (4, 16, 1, 0, 0, 8, 0), (4, 16, 3, 100, 0, 16, 0), (4, 32, 2, 0, 0, 8, 0),
    (4, 0, 2, 64, 0, 16, 0), (4, 0, 2, 1000, 0, 8, 0), (4, 0, 2, 3, 0, 16, 0),
    (0, 1, 2, 0, 0, 16, 3), (0, 1, 8, 0, 0, 8, 3), (0, 32, 2, 0, 0, 16, 3), (0, 32, 4, 0, 0, 8, 3),
    (0, 32, 4, 64, 2, 0, 3), (0, 32, 2, 100, 3, 0, 3), (0, 32, 2, 0, 0, 16, 2),
    (3, 32, 2, 0, 0, 16, 3), (3, 1, 2, 0, 0, 8, 3),
    ])
@pytest.mark.parametrize("n,k", [(1000, 4096), (512, 11008), (96, 8192), (37, 28672)])
def test_e8p_gemv_variants(Q, kernel, rep, rows, blocks, g, maxw, digits, n, k):
    """every tuning variant of every GEMV kernel computes the same thing (kernel 4: matrix-core GEMV on
    linear digit planes, 0: VALU integer GEMV on lane-ordered planes, 3: the same converting x itself)"""
    from quip_for_all_amd import capi
    L = capi.lib()
    P = O.make_layer("E8P12", k, n, seed=5)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((1, k)).astype(np.float16)).to(DEV)
    Qd = torch.from_numpy(P.Qidxs).to(DEV)
    grid = _cb(Q, "E8P12").grid_packed_abs
    y = torch.full((1, n), float("nan"), dtype=torch.float16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    xin = x
    if kernel == 4:
        xin = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=DEV)
        assert L.quip_e8p_x_to_planes(x.data_ptr(), xin.data_ptr(), k, st) == 0
    if kernel == 0:
        xin = torch.empty(3 * k + 16, dtype=torch.uint8, device=DEV)
        assert L.quip_e8p_x_to_planes_laneorder(x.data_ptr(), xin.data_ptr(), k, st) == 0
    rc = L.quip_e8p_gemv_tuned(xin.data_ptr(), Qd.data_ptr(), grid.data_ptr(), y.data_ptr(), n, k,
                               kernel, rep, rows, blocks, g, maxw, digits, None, st)
    assert rc == 0
    W64 = O.decompress_e8p(P.Qidxs).astype(np.float64)
    x64 = x.cpu().numpy().astype(np.float64)
    y64 = x64 @ W64.T
    err = np.abs(y.cpu().numpy().astype(np.float64) - y64)
    assert np.all(err <= _mm_tol(x64, W64, y64, xbits=13 if digits == 2 else 22)), err.max()


def test_e8p_mm_without_workspace_matches(Q):
    """the workspace-free ABI entry (VALU integer GEMV converting x itself) agrees with the
    workspace path (matrix-core GEMV) to the last fp16 bit or one ulp: the integer parts are
    identical, only the fp32 combination of the three digit sums is ordered differently"""
    from quip_for_all_amd import capi
    L = capi.lib()
    n, k = 640, 8192
    P = O.make_layer("E8P12", k, n, seed=6)
    x = torch.from_numpy(np.random.default_rng(2).standard_normal((1, k)).astype(np.float16)).to(DEV)
    Qd = torch.from_numpy(P.Qidxs).to(DEV)
    cb = _cb(Q, "E8P12")
    y1 = cb.mm(x, Qd)
    y2 = torch.empty_like(y1)
    assert L.quip_e8p_mm_origorder(x.data_ptr(), Qd.data_ptr(), cb.grid_packed_abs.data_ptr(), y2.data_ptr(),
                                   1, n, k, torch.cuda.current_stream().cuda_stream) == 0
    d = (y1.float() - y2.float()).abs()
    assert torch.all(d <= 2.0 ** -10 * y1.float().abs() + 1e-6)
    assert (d > 0).float().mean() < 0.05


def test_e8p_gemv_i8_dynamic_range(Q):
    """block fixed point: a huge outlier in x must not destroy the small elements beyond the
    stated bound, tiny / subnormal-only inputs must work, and scaling x by 2^s scales y exactly"""
    n, k = 256, 4096
    cb = _cb(Q, "E8P12")
    P = O.make_layer("E8P12", k, n, seed=21)
    Qd = torch.from_numpy(P.Qidxs).to(DEV)
    W64 = O.decompress_e8p(P.Qidxs).astype(np.float64)
    rng = np.random.default_rng(8)
    base = rng.standard_normal((1, k)).astype(np.float16)
    for name, x in (("outlier", np.where(np.arange(k) == 100, np.float16(3000.0), base)),
                    ("tiny", (base.astype(np.float32) * 2.0 ** -16).astype(np.float16)),
                    ("subnormal", np.full((1, k), 2.0 ** -24, np.float16)),
                    ("large", (base.astype(np.float32) * 32).astype(np.float16))):
        x = np.asarray(x, dtype=np.float16).reshape(1, k)
        y = cb.mm(torch.from_numpy(x).to(DEV), Qd).cpu().numpy().astype(np.float64)
        x64 = x.astype(np.float64)
        y64 = x64 @ W64.T
        tol = _mm_tol(x64, W64, y64) + 2.0 ** -24   # fp16 subnormal output spacing
        assert np.all(np.abs(y - y64) <= tol), (name, np.abs(y - y64).max())
    y1 = cb.mm(torch.from_numpy(base).to(DEV), Qd)
    y2 = cb.mm(torch.from_numpy((base.astype(np.float32) * 4).astype(np.float16)).to(DEV), Qd)
    assert torch.equal(y1 * 4, y2)    # power-of-two scaling only moves the block exponent


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


def _decode_planes(planes_u8, n):
    kp = (n + 511) // 512 * 512
    p = planes_u8.cpu().numpy()
    d = p[:3 * kp].reshape(3, kp).view(np.int8).astype(np.int64)
    sh = int(p[3 * kp:3 * kp + 4].view(np.int32)[0])
    X = d[0] * 65536 + d[1] * 256 + d[2]
    return X, sh, kp


@pytest.mark.parametrize("n,K", [(4096, 1), (8192, 1), (11008, 43), (28672, 7), (1024, 1), (1408, 11)])
def test_had_transform_planes(Q, n, K):
    """input-side transform written as int8 digit planes == oracle transform, to the fixed-point step"""
    rng = np.random.default_rng(n + K)
    x = rng.standard_normal((1, n)).astype(np.float16)
    su = (rng.integers(0, 2, n) * 2 - 1).astype(np.float16)
    had = None if K == 1 else O.random_orthogonal(K, rng).astype(np.float16)
    L_ = n // K
    scale = 0.37 / np.sqrt(L_)
    planes = torch.ops.quip_lib.had_transform_planes(
        torch.from_numpy(x).to(DEV), n, K, None if had is None else torch.from_numpy(had).to(DEV), True,
        torch.from_numpy(su).to(DEV), float(scale))
    X, sh, kp = _decode_planes(planes, n)
    assert np.all(np.abs(X) < 2 ** 22)
    assert np.all(X[n:] == 0)                                     # k padding
    ref = O.matmul_hadU(x.astype(np.float64) * su.astype(np.float64),
                        None if had is None else had.astype(np.float64), K, n, scale=0.37, transpose=True)[0]
    got = X[:n].astype(np.float64) * 2.0 ** -sh
    # half a fixed-point step + fp32 butterfly error
    tol = 2.0 ** (-sh - 1) + 2.0 ** -20 * np.linalg.norm(ref) / np.sqrt(n) * np.log2(n) + 1e-9
    assert np.max(np.abs(got - ref)) <= tol * 1.5, (np.max(np.abs(got - ref)), tol)
    # the block exponent is not wasteful: the largest |X| uses >= 22 - log2(sqrt(n)) - 1 bits
    assert np.abs(X).max() >= 2 ** (21 - np.log2(np.sqrt(n)) - 1.5)


def test_gemv_planes_end_to_end(Q):
    """had_transform_planes -> e8p_gemv_planes == oracle (transform then exact product)"""
    n_out, k = 2048, 4096
    P = O.make_layer("E8P12", k, n_out, seed=31)
    cb = _cb(Q, "E8P12")
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, k)).astype(np.float16)
    planes = torch.ops.quip_lib.had_transform_planes(torch.from_numpy(x).to(DEV), k, 1, None, True, None,
                                                     1.0 / np.sqrt(k))
    z = cb.mm_planes(planes, torch.from_numpy(P.Qidxs).to(DEV)).cpu().numpy().astype(np.float64)
    xh = O.fwht(x.astype(np.float64)) / np.sqrt(k)
    W64 = O.decompress_e8p(P.Qidxs).astype(np.float64)
    ref = xh @ W64.T
    assert np.all(np.abs(z - ref) <= 2.0 ** -10 * np.abs(ref) + 2.0 ** -13 * np.abs(xh).max() * np.abs(W64).sum(1)[None] / np.sqrt(k) + 1e-3)


def test_hadamard_more_rows_than_one_grid_dimension(Q):
    """rows > 65535 (prefill batches) are served in slices; every row equals the single-row result"""
    import quip_for_all_amd  # noqa: F401
    rows, n = 70000, 256
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, n, generator=g).half().cuda()
    y = torch.ops.quip_lib.hadamard(x, 1.0 / 16.0)
    pick = torch.tensor([0, 1, 65534, 65535, 65536, 69999], device="cuda")
    ref = torch.ops.quip_lib.hadamard(x[pick].contiguous(), 1.0 / 16.0)
    assert torch.equal(y[pick], ref)


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (4096, 11008), (512, 1408), (1000, 256), (28672, 8192)])
def test_d4_gemv_planes_exact_integer_path(Q, n, k):
    """D4 on the matrix-core GEMV: x -> digit planes -> integer product with the 2w int8 table;
    equals the float64 product with the oracle's D4 weights within one fp16 rounding of y plus the
    22-bit block fixed point of x (the same bound as E8P12's GEMV)"""
    cb = _cb(Q, "D4")
    rng = np.random.default_rng(n + k)
    q = rng.integers(0, 256, size=(n, k // 4), dtype=np.uint8)
    x = rng.standard_normal((1, k)).astype(np.float16)
    xd = torch.from_numpy(x).to(DEV)
    L = Q.capi.lib()
    planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=DEV)
    Q.capi.check(L.quip_e8p_x_to_planes(xd.data_ptr(), planes.data_ptr(), k, torch.cuda.current_stream().cuda_stream), "planes")
    qd = torch.from_numpy(q).to(DEV)
    y = cb.mm_planes(planes, qd).cpu().numpy().astype(np.float64)
    W64 = O.decompress_d4(q).astype(np.float64)
    x64 = x.astype(np.float64)
    y64 = x64 @ W64.T
    assert np.all(np.abs(y - y64) <= _mm_tol(x64, W64, y64)), np.abs(y - y64).max()
    # grouped launch == single launches, and == the generic D4 kernel within its own tolerance
    y2 = cb.mm_planes_group([planes, planes], [qd, qd])
    assert torch.equal(y2[0], y2[1]) and np.array_equal(y2[0].cpu().numpy().astype(np.float64), y)
    yg = cb.mm(xd, qd).cpu().numpy().astype(np.float64)
    assert np.all(np.abs(yg - y64) <= _mm_tol(x64, W64, y64) + 2.0 ** -9 * np.abs(y64))


@pytest.mark.parametrize("n,k,scale", [(4096, 4096, None), (11008, 4096, None), (4096, 2048, 0.25), (512, 256, -1.0),
                                       (8192, 8192, None)])
def test_rvq4_on_the_e8p_gemv_through_the_virtual_vector(Q, n, k, scale):
    """E8P12RVQ4B at bs=1: Hadamard launch writes the planes of x' = [s x_g | x_g], the E8P12 GEMV runs on the
    int16 view of the codes.  Result = W x with W = E8P(main) + s E8P(resid) summed EXACTLY: tight against
    the unrounded float64 weights, and within the reference's own per-weight fp16 rounding (2^-11 |w| per
    weight, origin_order.cu:337-385) of the oracle's rounded weights."""
    cb = _cb(Q, "E8P12RVQ4B", scale)
    s16 = cb.planes_resid_scale
    rng = np.random.default_rng(n + k)
    q = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(n, k // 8), dtype=np.int64).astype(np.int32)
    x = rng.standard_normal((1, k)).astype(np.float16)
    xd, qd = torch.from_numpy(x).to(DEV), torch.from_numpy(q).to(DEV)
    planes = torch.ops.quip_lib.had_transform_planes_fused(xd, k, 1, None, True, None, 1.0 / np.sqrt(k), None, 1e-5,
                                                           None, s16)
    y = cb.mm_planes(planes, qd).cpu().numpy().astype(np.float64)
    xh = O.fwht(x.astype(np.float64)) / np.sqrt(k)
    qu = q.view(np.uint32)
    main = O.e8p_decode_i8((qu >> 16).astype(np.uint16)).astype(np.float64) / 4.0
    res = O.e8p_decode_i8((qu & 0xFFFF).astype(np.uint16)).astype(np.float64) / 4.0
    Wexact = (main + s16 * res).reshape(n, -1)
    ref = xh @ Wexact.T
    wx = np.abs(xh) @ np.abs(Wexact).T
    assert np.all(np.abs(y - ref) <= 2.0 ** -10 * np.abs(ref) + 2.0 ** -20 * wx + 1e-6), np.abs(y - ref).max()
    Wref = O.decompress_e8prvq4(q, cb.opt_resid_scale).astype(np.float64)
    ref2 = xh @ Wref.T
    assert np.all(np.abs(y - ref2) <= 2.0 ** -10 * np.abs(ref2) + 2.0 ** -11 * wx + 1e-6)
    # and the generic (per-weight rounding) kernel agrees within the same bound
    xh16 = torch.ops.quip_lib.had_transform(xd, k, k, 1, None, True, None, None, None, None, 1.0 / np.sqrt(k))
    yg = cb.mm(xh16, qd).cpu().numpy().astype(np.float64)
    assert np.all(np.abs(y - yg) <= 2.0 ** -9 * np.abs(ref2) + 2.0 ** -10 * wx + 1e-6)


@pytest.mark.parametrize("n,k", [(4096, 4096), (11008, 4096), (512, 256), (8192, 8192)])
def test_hi_on_the_d4_mode_through_the_virtual_vector(Q, n, k):
    """HI at bs=1: planes of x' = [x0 x2 0 0 | x4 x6 0 0 | x1 x3 0 0 | x5 x7 0 0], D4 mode of the GEMV on the
    byte view of the codes with the entry [lo - 7.5, hi - 7.5, 0, 0]: exact integer arithmetic"""
    cb = _cb(Q, "HI")
    rng = np.random.default_rng(n + k)
    q = rng.integers(-2 ** 31, 2 ** 31 - 1, size=(n, k // 8), dtype=np.int64).astype(np.int32)
    x = rng.standard_normal((1, k)).astype(np.float16)
    xd, qd = torch.from_numpy(x).to(DEV), torch.from_numpy(q).to(DEV)
    planes = torch.ops.quip_lib.had_transform_planes_fused(xd, k, 1, None, True, None, 1.0 / np.sqrt(k), None, 1e-5,
                                                           None, cb.planes_resid_scale)
    y = cb.mm_planes(planes, qd).cpu().numpy().astype(np.float64)
    xh = O.fwht(x.astype(np.float64)) / np.sqrt(k)
    W64 = O.decompress_hi(q).astype(np.float64)
    ref = xh @ W64.T
    wx = np.abs(xh) @ np.abs(W64).T
    assert np.all(np.abs(y - ref) <= 2.0 ** -10 * np.abs(ref) + 2.0 ** -20 * wx + 1e-6), np.abs(y - ref).max()


@pytest.mark.parametrize("n,k,M", [(4096, 4096, 5), (512, 4096, 1), (11008, 4096, 4), (4096, 11008, 2), (4096, 11008, 5),
                                   (1024, 8192, 3), (28672, 8192, 3), (8192, 28672, 2), (48, 1024, 5), (4100, 2048, 5)])
def test_e8p_gemv_planes_rows_equal_single_row_gemv(n, k, M):
    """rows-mode GEMV (quip_e8p_gemv_planes_rows): row r of the result is bit identical to the bs=1 GEMV on
    row r's planes, and the plane images of had_transform_planes_rows equal the single-row transform's"""
    import quip_for_all_amd as Q
    rng = np.random.default_rng(n + k + M)
    qidx = torch.from_numpy(rng.integers(-32768, 32768, (n, k // 8), dtype=np.int16)).to(DEV)
    x = torch.from_numpy((rng.standard_normal((M, k)) * rng.uniform(0.1, 4.0, (M, 1))).astype(np.float16)).to(DEV)
    su = torch.from_numpy(rng.choice([-1.0, 1.0], k).astype(np.float16)).to(DEV)
    grid = Q.codebook.codebook_id["E8P12"](inference=True).to(DEV).grid_packed_abs
    op = torch.ops.quip_lib
    from quip_for_all_amd.quant import get_hadK
    had, K, _ = get_hadK(k, True)
    had = None if had is None else had.to(DEV).half().contiguous()
    planes = op.had_transform_planes_rows(x, k, K, had, True, su, 0.37, None, 1e-5, None)
    y = op.e8p_gemv_planes_rows(planes, qidx, grid)
    assert y.shape == (M, n)
    for r in range(M):
        pr = op.had_transform_planes_fused(x[r:r + 1], k, K, had, True, su, 0.37, None, 1e-5, None)
        used = 3 * ((k + 511) // 512 * 512) + 4          # planes + shift word (12 bytes of padding follow)
        assert torch.equal(planes[r][:used], pr[:used]), r
        assert torch.equal(y[r:r + 1], op.e8p_gemv_planes(pr, qidx, grid)), r


@pytest.mark.parametrize("n,rows", [(28672, 40), (14336, 33), (5120, 70), (1536, 64), (12288, 32), (3584, 100)])
@pytest.mark.parametrize("side", ["in", "out", "both"])
def test_wide_small_k_hadamard_batches_equal_single_rows(n, rows, side):
    """prefill batches of K = 3, 5, 7 times a power of two >= 512 (had_wide_batch_kernel: one workgroup per token row
    reads the K sub-rows once) give bit for bit what the row-parallel launch gives row by row"""
    test_tall_hadamard_batches_equal_single_rows(n, rows, side)


@pytest.mark.parametrize("n,rows", [(11008, 40), (11008, 300), (5120, 64), (2752, 33)])
@pytest.mark.parametrize("side", ["in", "out"])
def test_tall_hadamard_batches_with_table_factors(n, rows, side):
    """the +-1 factors of get_hadK(use_rand=False): 11008 = 172 x 64 (eleven row tiles of H in the batch kernel),
    5120 = 20 x 256, 2752 = 172 x 16 (not the batch kernel's: row-parallel launch) -- batch rows equal single rows"""
    test_tall_hadamard_batches_equal_single_rows(n, rows, side, use_rand=False)


def _same_or_close(batch_row, single_row, what):
    """the batch kernels that run the same operations as the single-row launch give the same bits; had_tall_batch_kernel
    runs its K-mix on the fp16 matrix cores (hi + lo split, exact products, fp32 accumulation in another order than the
    single-row kernel's fp32 MFMA): every element within one fp16 ulp of max(|y|, rms(y)), at most a few per cent differ"""
    if torch.equal(batch_row, single_row):
        return
    a, b = batch_row.float(), single_row.float()
    scale = torch.maximum(b.abs(), b.pow(2).mean().sqrt().expand_as(b))
    ulp = torch.exp2(torch.floor(torch.log2(scale)) - 10)
    assert torch.all((a - b).abs() <= ulp), (what, float(((a - b).abs() / ulp).max()))
    assert float((a != b).float().mean()) < 0.06, (what, float((a != b).float().mean()))


@pytest.mark.parametrize("n,rows", [(11008, 70), (2752, 33), (5504, 64), (11008, 600), (688 * 4, 40)])
@pytest.mark.parametrize("side", ["in", "out", "both"])
def test_tall_hadamard_batches_equal_single_rows(n, rows, side, use_rand=True):
    """prefill batches of the tall transform (one workgroup per row, K-mix in place on the fp16 matrix cores:
    had_tall_batch_kernel) against the latency-shaped launch row by row (_same_or_close), and a row's result does not
    depend on the batch it is in; input side with gate / pre-scale, output side with post-scale / bias / residual and a
    ragged out_features"""
    from quip_for_all_amd.quant import get_hadK
    torch.manual_seed(n + rows)
    had, K, qn = get_hadK(n, use_rand)
    assert qn == n and K > 1
    hd = had.to(DEV).half().contiguous()
    op = torch.ops.quip_lib
    x = torch.randn(rows, n, device=DEV).half()
    v1 = torch.randn(n, device=DEV).half()
    sub = slice(min(rows // 4, rows - 32), min(rows // 4, rows - 32) + 32)      # 32 rows: still the batch kernel
    if side == "in":
        g = torch.randn(rows, n, device=DEV).half()
        full = op.had_transform_fused(x, n, n, K, hd, True, v1, None, None, None, 0.37, None, None, 1e-5, g)
        for r in (0, 1, rows // 2, rows - 1):
            one = op.had_transform_fused(x[r:r + 1], n, n, K, hd, True, v1, None, None, None, 0.37, None, None, 1e-5, g[r:r + 1])
            _same_or_close(full[r:r + 1], one, r)
        part = op.had_transform_fused(x[sub].contiguous(), n, n, K, hd, True, v1, None, None, None, 0.37, None, None, 1e-5, g[sub].contiguous())
        assert torch.equal(part, full[sub]), "a row's result does not depend on the batch it is in"
    elif side == "both":
        # vectors of both sides in one launch: not the batch kernel's case (it keeps one side's vectors in registers),
        # the launch takes the row-parallel kernel -- same bits
        post = torch.randn(n, device=DEV).half()
        full = op.had_transform_fused(x, n, n, K, hd, True, v1, None, post, None, 0.37, None, None, 1e-5, None)
        for r in (0, rows // 2, rows - 1):
            one = op.had_transform_fused(x[r:r + 1], n, n, K, hd, True, v1, None, post, None, 0.37, None, None, 1e-5, None)
            assert torch.equal(full[r:r + 1], one), r
    else:
        out_f = n - 24
        bias = torch.randn(out_f, device=DEV).half()
        res = torch.randn(rows, out_f, device=DEV).half()
        post = v1[:out_f].contiguous()
        full = op.had_transform_fused(x, out_f, n, K, hd, False, None, None, post, bias, 0.11, res, None, 1e-5, None)
        for r in (0, 1, rows // 2, rows - 1):
            one = op.had_transform_fused(x[r:r + 1], out_f, n, K, hd, False, None, None, post, bias, 0.11, res[r:r + 1], None, 1e-5, None)
            _same_or_close(full[r:r + 1], one, r)
        part = op.had_transform_fused(x[sub].contiguous(), out_f, n, K, hd, False, None, None, post, bias, 0.11, res[sub].contiguous(), None, 1e-5, None)
        assert torch.equal(part, full[sub]), "a row's result does not depend on the batch it is in"
        assert full.shape == (rows, out_f)


@pytest.mark.parametrize("n,rows,out_f", [(4096, 100, 4096), (1024, 33, 1000), (2048, 64, 2048), (4096, 70000, 4096),
                                          (8192, 48, 8192)])
def test_kone_hadamard_batches_equal_single_rows(n, rows, out_f):
    """prefill batches of the power-of-two transform (had_kone_batch_kernel: 64 VGPRs, eight rows resident per CU)
    give bit for bit what the latency-shaped launch gives row by row (input side with gate / SU, output side with
    SV / bias / residual and a ragged out_features)"""
    torch.manual_seed(n + rows)
    op = torch.ops.quip_lib
    x = torch.randn(rows, n, device=DEV).half()
    su = torch.randn(n, device=DEV).half()
    g = torch.randn(rows, n, device=DEV).half()
    full_in = op.had_transform_fused(x, n, n, 1, None, True, su, None, None, None, 0.37, None, None, 1e-5, g)
    bias = torch.randn(out_f, device=DEV).half()
    res = torch.randn(rows, out_f, device=DEV).half()
    post = su[:out_f].contiguous()
    full_out = op.had_transform_fused(x, out_f, n, 1, None, False, None, None, post, bias, 0.11, res, None, 1e-5, None)
    for r in (0, 1, rows // 2, rows - 1):
        one = op.had_transform_fused(x[r:r + 1], n, n, 1, None, True, su, None, None, None, 0.37, None, None, 1e-5, g[r:r + 1])
        assert torch.equal(full_in[r:r + 1], one), r
        one = op.had_transform_fused(x[r:r + 1], out_f, n, 1, None, False, None, None, post, bias, 0.11, res[r:r + 1], None, 1e-5, None)
        assert torch.equal(full_out[r:r + 1], one), r


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32, torch.float16])
@pytest.mark.parametrize("n,rows", [(2, 3), (64, 5), (1024, 7), (4096, 3), (32768, 2)])
def test_hadamard_op_all_dtypes(Q, dtype, n, rows):
    """quip_lib::hadamard takes the dtypes of the reference's op (fast_hadamard_transform: fp16 / bf16 / fp32,
    register_lib.py:10-20): fp32 arithmetic inside, ONE rounding to the I/O dtype.  Bound: the float64 transform of the
    same (already rounded) input, plus fp32 accumulation noise, plus half an ulp of the output type."""
    rng = np.random.default_rng(n + rows)
    x = torch.from_numpy(rng.standard_normal((rows, n)).astype(np.float32)).to(dtype)
    s = 1.0 / np.sqrt(n)
    y = torch.ops.quip_lib.hadamard(x.to(DEV), s)
    assert y.dtype == dtype and y.shape == x.shape
    ref = O.fwht(x.double().numpy()) * s
    ulp = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8, torch.float32: 2.0 ** -24}[dtype]
    scale = np.abs(x.double().numpy()).sum(-1, keepdims=True) * s     # sum |x_i| / sqrt(n): accumulation error scale
    tol = ulp * np.abs(ref) + 2.0 ** -21 * scale + 1e-30
    assert np.all(np.abs(y.double().cpu().numpy() - ref) <= tol)
    # 3-D, non-contiguous input like the reference's call sites (quant.py:78-82)
    x3 = x.to(DEV).reshape(rows, 1, n).expand(rows, 2, n).transpose(0, 1)
    y3 = torch.ops.quip_lib.hadamard(x3, s)
    assert torch.equal(y3[0], y) and torch.equal(y3[1], y)


def test_transform_vector_lengths_are_checked(Q):
    """short SU / SV / bias / had / rms_weight vectors would be read out of bounds by the kernels: the ops refuse"""
    x = torch.zeros(1, 256, dtype=torch.float16, device=DEV)
    ok = torch.ones(256, dtype=torch.float16, device=DEV)
    short = torch.ones(100, dtype=torch.float16, device=DEV)
    op = torch.ops.quip_lib
    op.had_transform_fused(x, 256, 256, 1, None, True, ok, None, ok, ok, 1.0, None, ok, 1e-5, None)
    for kw in (dict(pre=short), dict(post=short), dict(bias=short), dict(rms=short)):
        with pytest.raises((ValueError, RuntimeError)):
            op.had_transform_fused(x, 256, 256, 1, None, True, kw.get("pre", ok), None, kw.get("post", ok),
                                   kw.get("bias", ok), 1.0, None, kw.get("rms", ok), 1e-5, None)
    with pytest.raises((ValueError, RuntimeError)):
        op.had_transform_planes_fused(x, 256, 1, None, True, short, 1.0, None, 1e-5, None)
    x688 = torch.zeros(1, 688, dtype=torch.float16, device=DEV)
    with pytest.raises((ValueError, RuntimeError)):      # (K, K) factor too small
        op.had_transform_fused(x688, 688, 688, 43, torch.ones(10, 10, dtype=torch.float16, device=DEV), True, None, None,
                               None, None, 1.0, None, None, 1e-5, None)
    # a CPU-resident grid must not reach the kernel as a pointer
    cb = _cb(Q, "E8P12")
    planes = op.had_transform_planes_fused(x, 256, 1, None, True, ok, 1.0, None, 1e-5, None)
    Qd = torch.zeros(64, 32, dtype=torch.int16, device=DEV)
    with pytest.raises((ValueError, RuntimeError)):
        op.e8p_gemv_planes_group([planes], [Qd], cb.grid_packed_abs.cpu())
    # QuantLinear refuses an input of the wrong width (the reference fails in x * SU)
    layer = Q.qlinear.QuantLinear(256, 64, cb, bias=False).to(DEV).eval()
    with pytest.raises(RuntimeError):
        layer(torch.zeros(1, 128, dtype=torch.float16, device=DEV))
