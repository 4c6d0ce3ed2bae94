"""Module-level GPU parity: QuantLinear.forward on the HIP path vs (a) the golden
outputs of the reference's own QuantLinear.forward and (b) the fp64 oracle."""
import numpy as np
import pytest
import torch

from oracle import quip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _layer(P):
    import quip_for_all_amd as Q
    return Q.QuantLinear.from_params(P).to(DEV).eval()


def test_reference_module_goldens(golden, golden_meta):
    for case in golden_meta["module_cases"]:
        P = O.make_layer(case["codebook"], case["in_features"], case["out_features"], seed=case["seed"],
                         bias=case["bias"], per_channel=case["per_channel"], resid_scale=case["resid_scale"])
        layer = _layer(P)
        What = O.qlinear_dense_weight(P)
        for M in case["Ms"]:
            x = golden[f"mod{case['idx']}_M{M}_x"]
            yref = golden[f"mod{case['idx']}_M{M}_y"].astype(np.float64)
            with torch.no_grad():
                y = layer(torch.from_numpy(x).to(DEV)).cpu().numpy().astype(np.float64)
            yexact = O.qlinear_forward(P, x, "exact", What)
            tol = O.parity_bound(P, x, What)          # envelope that also holds the reference's fp16-staged pipeline
            assert np.all(np.abs(y - yexact) <= O.ulp_bound(P, x, What)), (case, M, np.abs(y - yexact).max())
            # and directly against the reference's output: both are within tol of exact
            assert np.all(np.abs(y - yref) <= 2 * tol), (case, M)


@pytest.mark.parametrize("cbid", ["E8P12", "E8P12RVQ4B", "E8P12RVQ3B", "D4", "HI"])
@pytest.mark.parametrize("fin,fout", [(4096, 4096), (4096, 11008), (11008, 4096)])
def test_llama7b_layer_shapes(cbid, fin, fout):
    P = O.make_layer(cbid, fin, fout, seed=fin + fout, bias=False)
    layer = _layer(P)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, fin)).astype(np.float16)
    with torch.no_grad():
        y = layer(torch.from_numpy(x).to(DEV)).cpu().numpy().astype(np.float64)
    What = O.qlinear_dense_weight(P)
    yexact = O.qlinear_forward(P, x, "exact", What)
    tol = O.ulp_bound(P, x, What)                 # the stated bound: 4 fp16 ulps of max(|y|, rms(y))
    assert np.all(np.abs(y - yexact) <= tol), np.abs(y - yexact).max()


def test_config1_golden_on_gpu(golden, golden_meta):
    c = golden_meta["cfg1"]
    P = O.make_layer(c["codebook"], c["in_features"], c["out_features"], seed=c["seed"])
    layer = _layer(P)
    x = golden["cfg1_x"]
    with torch.no_grad():
        y = layer(torch.from_numpy(x).to(DEV)).cpu().numpy().astype(np.float64)
    What = O.qlinear_dense_weight(P)
    tol = O.parity_bound(P, x, What)
    assert np.all(np.abs(y - O.qlinear_forward(P, x, "exact", What)) <= O.ulp_bound(P, x, What))
    assert np.all(np.abs(y - golden["cfg1_y"].astype(np.float64)) <= 2 * tol)


def test_forward_3d_input_bf16_and_merged_suv():
    P = O.make_layer("E8P12", 1024, 1024, seed=3, bias=True)
    layer = _layer(P)
    rng = np.random.default_rng(1)
    x = torch.from_numpy(rng.standard_normal((2, 3, 1024)).astype(np.float32)).to(DEV)
    with torch.no_grad():
        y16 = layer(x.half())
        ybf = layer(x.bfloat16())
    assert y16.shape == (2, 3, 1024) and ybf.dtype == torch.bfloat16
    assert torch.allclose(y16.float(), ybf.float(), atol=0.15, rtol=0.05)
    # merge_suv checkpoints carry SU = SV = None (qlinear.py:117-131)
    P2 = O.make_layer("E8P12", 1024, 1024, seed=3, bias=True)
    P2.SU, P2.SV = None, None
    l2 = _layer(P2)
    with torch.no_grad():
        y2 = l2(x.half()).cpu().numpy().astype(np.float64)
    ex = O.qlinear_forward(P2, x.half().cpu().numpy(), "exact")
    assert np.all(np.abs(y2 - ex) <= O.parity_bound(P2, x.half().cpu().numpy()))


def test_training_branch_equals_eval_branch():
    """path identity: x @ calc_weight() == fused eval forward"""
    P = O.make_layer("E8P12", 1376, 512, seed=4)   # K_left = 43
    layer = _layer(P)
    with torch.no_grad():
        layer.Wscale.fill_(P.wscale_float)
    x = torch.from_numpy(np.random.default_rng(2).standard_normal((4, 1376)).astype(np.float16)).to(DEV)
    with torch.no_grad():
        ye = layer(x)
        layer.train()
        yt = layer(x)
        layer.eval()
    scale = ye.float().abs().max().item()
    assert (ye.float() - yt.float()).abs().max().item() <= 2 ** -7 * scale


def test_full_size_70b_rows_via_properties():
    """Llama-2-70B gate/up shape (28672 x 8192) and down shape (8192 x 28672): too big for the
    numpy oracle in seconds, so check size-independent properties: sampled rows against the
    oracle, linearity in x, and agreement of the fused GEMV with decompress + fp32 GEMM."""
    import quip_for_all_amd as Q
    for (n, k) in ((28672, 8192), (8192, 28672)):
        g = torch.Generator().manual_seed(n)
        Qi = torch.randint(-32768, 32767, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16)
        cb = Q.codebook.codebook_id["E8P12"](inference=True).to(DEV)
        Qd = Qi.to(DEV)
        x1 = torch.randn(1, k, generator=g).half().to(DEV)
        x2 = torch.randn(1, k, generator=g).half().to(DEV)
        y1, y2 = cb.mm(x1, Qd), cb.mm(x2, Qd)
        # sampled rows vs oracle
        rows = np.array([0, 1, 63, 64, 777, n // 2, n - 2, n - 1])
        W64 = O.decompress_e8p(Qi.numpy()[rows]).astype(np.float64)
        x64 = x1.cpu().numpy().astype(np.float64)
        ref = x64 @ W64.T
        err = np.abs(y1.cpu().numpy().astype(np.float64)[:, rows] - ref)
        assert np.all(err <= 2.0 ** -10 * np.abs(ref) + 2.0 ** -21 * (np.abs(x64) @ np.abs(W64).T))
        # fused GEMV == decompress + fp32 GEMM on the GPU (decompress is bit-exact tested)
        Wd = cb.decompress_weight(Qd).float()
        yd = (x1.float() @ Wd.T)
        assert (y1.float() - yd).abs().max() <= 2 ** -9 * yd.abs().max()
        # linearity: f(x1) + f(x2) == f(x1 + x2) up to fp16 rounding of inputs/outputs
        xs = (x1.float() + x2.float())
        ys = (xs @ Wd.T)
        assert ((y1.float() + y2.float()) - ys).abs().max() <= 2 ** -8 * ys.abs().max()


@pytest.mark.parametrize("cbid", ["E8P12RVQ4B", "E8P12RVQ3B"])
@pytest.mark.parametrize("fin,fout,M", [(28672, 512, 1), (28672, 512, 2), (16384, 256, 1), (14336, 1000, 1)])
def test_rvq_rows_longer_than_28672_on_matrix_core_path(cbid, fin, fout, M):
    """E8P12RVQ4B / RVQ3B at Llama-2-70B's down_proj width: the 2k = 57344-wide virtual row is beyond the first GEMV
    kernel's LDS budget and is taken by the K-splitting kernel (csrc/e8p_gemv_v2.hip; RVQ3: its third-table mode on
    the 3-byte codes) -- module forward against the oracle, rows bit identical to bs=1"""
    import quip_for_all_amd as Q
    P = O.make_layer(cbid, fin, fout, seed=fin + fout)
    layer = _layer(P)
    assert layer.codebook.planes_supported(layer.q_out_features, layer.q_in_features)
    x = np.random.default_rng(M).standard_normal((M, fin)).astype(np.float16)
    xd = torch.from_numpy(x).to(DEV)
    with torch.no_grad():
        y = layer(xd)
        for r in range(M):
            assert torch.equal(y[r:r + 1], layer(xd[r:r + 1]))
    What = O.qlinear_dense_weight(P)
    x64 = x.astype(np.float64)
    ref = O.qlinear_forward(P, x64, "exact", What)
    assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= O.ulp_bound(P, x64, What))


@pytest.mark.parametrize("fin,fout,M", [(28672, 512, 1), (28672, 8192, 1), (16384, 256, 1), (14336, 1000, 1), (28672, 512, 3)])
def test_hi_rows_longer_than_28672_on_matrix_core_path(fin, fout, M):
    """HI at Llama-2-70B's down_proj width: the 2k = 57344-wide virtual D4 row is beyond the first GEMV kernel and is
    taken by the K-splitting kernel's D4 table mode (csrc/e8p_gemv_v2.hip: T2[low code byte] = (4w, 0), T1[high code byte]
    = (0, 4w)) -- module forward against the oracle, rows bit identical to bs=1, and (where both kernels take the shape)
    bit identical to the first kernel"""
    P = O.make_layer("HI", fin, fout, seed=fin + fout)
    layer = _layer(P)
    assert layer.codebook.planes_supported(layer.q_out_features, layer.q_in_features)
    assert layer.regime(1) == "gemv_planes"
    x = np.random.default_rng(M).standard_normal((M, fin)).astype(np.float16)
    xd = torch.from_numpy(x).to(DEV)
    with torch.no_grad():
        y = layer(xd)
        for r in range(M):
            assert torch.equal(y[r:r + 1], layer(xd[r:r + 1]))
    What = O.qlinear_dense_weight(P)
    x64 = x.astype(np.float64)
    ref = O.qlinear_forward(P, x64, "exact", What)
    assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= O.ulp_bound(P, x64, What))


@pytest.mark.parametrize("cbid,n,k", [("D4", 512, 4096), ("D4", 4096, 11008), ("D4", 100, 28672), ("HI", 512, 4096), ("HI", 300, 11008)])
def test_v2_d4_table_mode_equals_first_kernel(cbid, n, k):
    """the K-splitting kernel's D4 table mode against the first kernel's on shapes both take: exact integer sums, one
    rounding -- bit identical"""
    import ctypes
    import math
    from quip_for_all_amd import capi
    from quip_for_all_amd.register_lib import _gemv_workspace, _stream
    P = O.make_layer(cbid, k, n, seed=n + k)
    layer = _layer(P)
    cb = layer.codebook
    xd = torch.from_numpy(np.random.default_rng(n).standard_normal((1, k)).astype(np.float16)).to(DEV)
    L_in = layer.q_in_features // layer.K_left
    with torch.no_grad():
        planes = torch.ops.quip_lib.had_transform_planes_fused(
            xd, layer.q_in_features, layer.K_left, layer._had("had_left"), True, layer._vec(layer.SU),
            layer.wscale_float / math.sqrt(L_in), None, 1e-5, None, getattr(cb, "planes_resid_scale", 0.0))
        y1 = cb.mm_planes(planes, layer.Qidxs)
    q8 = layer.Qidxs.view(torch.uint8)
    grid = cb.grid if cbid == "D4" else cb._virtual_grid(q8.device)
    kv = q8.shape[1] * 4
    y2 = torch.empty_like(y1)
    ws = _gemv_workspace(q8.device, n)
    # (the K-splitting kernel directly: QUIP_GEMV_V2 does not reach the D4 entry points)
    rc = capi.lib().quip_d4_gemv_planes_v2(planes.data_ptr(), q8.data_ptr(), grid.data_ptr(), y2.data_ptr(), n, kv,
                                           ws.data_ptr(), ws.numel() * 4, _stream(q8))
    assert rc == 0, rc
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("cbid,fin,fout", [("E8P12", 4096, 4096), ("E8P12", 1408, 512), ("D4", 1024, 1024),
                                           ("E8P12", 4096, 11008), ("E8P12", 11008, 4096),
                                           ("E8P12RVQ4B", 4096, 11008), ("E8P12RVQ4B", 11008, 4096),
                                           ("D4", 11008, 4096), ("HI", 4096, 4096), ("HI", 11008, 4096),
                                           ("E8P12RVQ3B", 4096, 4096), ("E8P12RVQ3B", 4096, 11008),
                                           ("E8P12RVQ3B", 11008, 4096), ("E8P12RVQ3B", 256, 688)])
@pytest.mark.parametrize("M", [1, 3])
def test_forward_fused_glue(cbid, fin, fout, M):
    """RMSNorm / SiLU*mul / residual folded into the Hadamard launches == doing them separately"""
    P = O.make_layer(cbid, fin, fout, seed=fin + fout + M)
    layer = _layer(P)
    rng = np.random.default_rng(M)
    x = rng.standard_normal((M, fin)).astype(np.float16)
    g = rng.standard_normal((M, fin)).astype(np.float16)
    w = (1 + 0.1 * rng.standard_normal(fin)).astype(np.float16)
    res = rng.standard_normal((M, fout)).astype(np.float16)
    xd, gd, wd, rd = (torch.from_numpy(t).to(DEV) for t in (x, g, w, res))
    with torch.no_grad():
        y1 = layer.forward_fused(xd, rms_weight=wd, rms_eps=1e-5).cpu().numpy().astype(np.float64)
        y2 = layer.forward_fused(xd, gate=gd, residual=rd).cpu().numpy().astype(np.float64)
    x64, g64 = x.astype(np.float64), g.astype(np.float64)
    xn = x64 / np.sqrt((x64 ** 2).mean(axis=1, keepdims=True) + 1e-5) * w.astype(np.float64)
    What = O.qlinear_dense_weight(P)
    r1 = O.qlinear_forward(P, xn, "exact", What)
    xs = g64 / (1 + np.exp(-g64)) * x64
    r2 = O.qlinear_forward(P, xs, "exact", What) + res.astype(np.float64)
    assert np.all(np.abs(y1 - r1) <= O.parity_bound(P, xn, What)), np.abs(y1 - r1).max()
    assert np.all(np.abs(y2 - r2) <= O.parity_bound(P, xs, What) + 2.0 ** -10 * np.abs(r2)), np.abs(y2 - r2).max()


@pytest.mark.parametrize("fin,fouts", [(4096, (4096, 4096, 4096)), (4096, (11008, 11008)), (8192, (8192, 1024, 1024)),
                                       (1408, (512, 256)), (11008, (4096, 4096)), (4096, (4096,)),
                                       (8192, (28672, 28672))])
def test_forward_group_equals_single_calls(fin, fouts):
    """grouped launches (q/k/v, gate/up) give exactly what the modules give one by one"""
    from quip_for_all_amd.qlinear import forward_group
    layers = [_layer(O.make_layer("E8P12", fin, fo, seed=fin + fo + i)) for i, fo in enumerate(fouts)]
    rng = np.random.default_rng(fin)
    x = torch.from_numpy(rng.standard_normal((1, fin)).astype(np.float16)).to(DEV)
    w = torch.from_numpy((1 + 0.1 * rng.standard_normal(fin)).astype(np.float16)).to(DEV)
    res = [torch.from_numpy(rng.standard_normal((1, fo)).astype(np.float16)).to(DEV) for fo in fouts]
    with torch.no_grad():
        for kw in ({}, {"rms_weight": w}):
            single = [l.forward_fused(x, **kw) for l in layers]
            group = forward_group(layers, x, **kw)
            for a, b in zip(single, group):
                assert torch.equal(a, b)
        single = [l.forward_fused(x, residual=r) for l, r in zip(layers, res)]
        group = forward_group(layers, x, residual=res)
        for a, b in zip(single, group):
            assert torch.equal(a, b)


@pytest.mark.parametrize("k,fouts,with_prev,with_rms", [
    (4096, (4096, 4096, 4096), True, True), (4096, (4096,), False, False), (4096, (11008, 11008), True, True),
    (1024, (1024, 512, 512), True, True), (8192, (8192, 1024, 1024), False, True), (8192, (8192,), False, False),
    (2048, (2048, 640), True, False), (8192, (28672, 28672), True, True)])
def test_gemv_fused_prologue_bit_identical(k, fouts, with_prev, with_rms):
    """quip_e8p_gemv_fused == output transform of the producer -> grouped input transforms ->
    grouped GEMV, bit for bit (raw GEMV outputs and the producer's finished output)"""
    from quip_for_all_amd import qlinear as QL
    layers = [_layer(O.make_layer("E8P12", k, fo, seed=k + fo + i)) for i, fo in enumerate(fouts)]
    if not QL.fused_in_supported(layers):
        pytest.skip("shape not served by the fused prologue (LDS budget)")
    rng = np.random.default_rng(k + len(fouts))
    t = lambda a: torch.from_numpy(a.astype(np.float16)).to(DEV)  # noqa: E731
    w = t(1 + 0.1 * rng.standard_normal(k)) if with_rms else None
    with torch.no_grad():
        if with_prev:
            prev = _layer(O.make_layer("E8P12", 1024, k, seed=k + 7))
            assert QL.fused_in_supported(layers, prev=prev)
            z = t(rng.standard_normal((1, k)) * 8)
            res = t(rng.standard_normal((1, k)))
            h, zs = QL.gemv_fused(layers, prev=prev, z=z, residual=res, rms_weight=w)
            (h_ref,) = QL.out_transform_group([prev], [z], residual=[res])
            assert torch.equal(h, h_ref)
            x = h_ref
        else:
            x = t(rng.standard_normal((1, k)))
            h, zs = QL.gemv_fused(layers, x=x, rms_weight=w)
            assert h is None
        planes = torch.ops.quip_lib.had_transform_planes_group(
            x, k, 1, [None] * len(layers), True, [l._vec(l.SU) for l in layers],
            [l.wscale_float / np.sqrt(k) for l in layers], w, 1e-5, None)
        for l, pl, zf in zip(layers, planes, zs):
            assert torch.equal(zf, torch.ops.quip_lib.e8p_gemv_planes(pl, l.Qidxs, l.codebook.grid_packed_abs))


@pytest.mark.parametrize("k,fouts,with_rms,with_res", [(4096, (4096, 4096, 4096), True, True), (4096, (11008, 11008), True, True),
                                                       (1024, (1024, 512, 512), True, False), (8192, (8192,), False, True),
                                                       (256, (256, 128), True, True)])
def test_gemv_chain_bit_identical(k, fouts, with_rms, with_res):
    """Hadamard chain launch (producer's output side + consumers' input side) + grouped GEMV ==
    the separate launches, bit for bit"""
    from quip_for_all_amd import qlinear as QL
    layers = [_layer(O.make_layer("E8P12", k, fo, seed=k + fo + i)) for i, fo in enumerate(fouts)]
    prev = _layer(O.make_layer("E8P12", 512, k, seed=k + 11))
    assert QL.chain_supported(layers, prev)
    rng = np.random.default_rng(k)
    t = lambda a: torch.from_numpy(a.astype(np.float16)).to(DEV)  # noqa: E731
    w = t(1 + 0.1 * rng.standard_normal(k)) if with_rms else None
    z = t(rng.standard_normal((1, k)) * 8)
    res = t(rng.standard_normal((1, k))) if with_res else None
    with torch.no_grad():
        h, zs = QL.gemv_chain(layers, prev, z, residual=res, rms_weight=w)
        (h_ref,) = QL.out_transform_group([prev], [z], residual=[res])
        assert torch.equal(h, h_ref)
        planes = torch.ops.quip_lib.had_transform_planes_group(
            h_ref, k, 1, [None] * len(layers), True, [l._vec(l.SU) for l in layers],
            [l.wscale_float / np.sqrt(k) for l in layers], w, 1e-5, None)
        for l, pl, zf in zip(layers, planes, zs):
            assert torch.equal(zf, torch.ops.quip_lib.e8p_gemv_planes(pl, l.Qidxs, l.codebook.grid_packed_abs))


@pytest.mark.parametrize("cbid,fin,fouts,M", [("E8P12RVQ4B", 4096, (4096, 4096, 4096), 1), ("D4", 1408, (512, 688), 1),
                                              ("HI", 256, (256, 688), 3), ("E8P12", 4096, (11008, 11008), 4),
                                              ("E8P12RVQ3B", 256, (256, 256, 128), 1)])
def test_forward_group_generic_codebooks_and_batches(cbid, fin, fouts, M):
    """grouped transforms around the per-module codebook product (any codebook, any batch) give exactly
    what the modules give one by one"""
    from quip_for_all_amd.qlinear import forward_group
    layers = [_layer(O.make_layer(cbid, fin, fo, seed=fin + fo + i)) for i, fo in enumerate(fouts)]
    rng = np.random.default_rng(fin + M)
    x = torch.from_numpy(rng.standard_normal((M, fin)).astype(np.float16)).to(DEV)
    w = torch.from_numpy((1 + 0.1 * rng.standard_normal(fin)).astype(np.float16)).to(DEV)
    res = [torch.from_numpy(rng.standard_normal((M, fo)).astype(np.float16)).to(DEV) for fo in fouts]
    with torch.no_grad():
        single = [l.forward_fused(x, rms_weight=w, residual=r) for l, r in zip(layers, res)]
        group = forward_group(layers, x, rms_weight=w, residual=res)
        for a, b in zip(single, group):
            assert torch.equal(a, b)


@pytest.mark.parametrize("fin,fout,M", [(4096, 4096, 2), (4096, 11008, 3), (11008, 4096, 2), (1408, 512, 3),
                                        (4096, 4096, 5), (4096, 4096, 4), (4096, 11008, 5), (11008, 4096, 5),
                                        (4096, 4096, 13), (8192, 1024, 7), (8192, 8192, 3), (1024, 8192, 16),
                                        (256, 688, 5), (688, 256, 6), (4096, 28672, 2)])
def test_skinny_rows_on_matrix_core_path(fin, fout, M, cbid="E8P12"):
    """2..16 rows go through per-row digit planes + the rows-mode GEMV ((row, plane) pairs in the MFMA's A
    rows: up to 5 rows per pass over the codes); each row must equal the bs=1 result of that row bit for
    bit (same integer arithmetic) and sit inside the oracle bound"""
    P = O.make_layer(cbid, fin, fout, seed=fin + fout + M)
    layer = _layer(P)
    layer.skinny_exact = True       # the exact integer path for every row count (QUIP_SKINNY_EXACT=1)
    rng = np.random.default_rng(M + fin)
    x = rng.standard_normal((M, fin)).astype(np.float16)
    xd = torch.from_numpy(x).to(DEV)
    with torch.no_grad():
        y = layer(xd)
        rows = [layer(xd[i:i + 1]) for i in range(M)]
    for i in range(M):
        assert torch.equal(y[i:i + 1], rows[i])
    What = O.qlinear_dense_weight(P)
    ref = O.qlinear_forward(P, x.astype(np.float64), "exact", What)
    assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= O.ulp_bound(P, x.astype(np.float64), What))
    # default dispatch: as long as one exact pass carries the rows nothing changes; beyond that E8P12 / E8P12RVQ4B take the
    # single-pass fp16 skinny kernel -- inside the same stated bound, but not bit identical to bs=1 any more
    layer.skinny_exact = False
    with torch.no_grad():
        yd = layer(xd)
    if layer.regime(M) == "rows_exact":
        assert torch.equal(yd, y)
    else:
        assert np.all(np.abs(yd.cpu().numpy().astype(np.float64) - ref) <= O.ulp_bound(P, x.astype(np.float64), What))


@pytest.mark.parametrize("fin,fout", [(4096, 4096), (4096, 11008), (11008, 4096), (8192, 8192), (8192, 28672),
                                      (28672, 8192), (256, 688), (1408, 512)])
@pytest.mark.parametrize("M", [6, 16, 31, 32, 33, 100, 257])
def test_single_pass_skinny_kernel(fin, fout, M):
    """6 <= M <= 32 rows (and more: chunks of 32 in one launch) through the fp16-MFMA skinny product (csrc/e8p_skinny_gemm.hip; the 1 < M < 32 use of the
    reference's tinygemm kernel, origin_order.cu:388-555): the op against the float64 product of the same fp16
    operands (one fp16 rounding + fp32 accumulation), the module inside the stated ulp bound, and a row's result
    independent of the batch it sits in"""
    import quip_for_all_amd as Q
    P = O.make_layer("E8P12", fin, fout, seed=fin + fout)
    layer = _layer(P)
    cb = layer.codebook
    rng = np.random.default_rng(M + fout)
    k, n = layer.q_in_features, layer.q_out_features
    if not cb.skinny_supported(M, n, k):
        pytest.skip("k % 128 != 0: rows mode / generic path")
    xh = torch.from_numpy(rng.standard_normal((M, k)).astype(np.float16)).to(DEV)
    z = torch.ops.quip_lib.e8p_mm_skinny(xh, layer.Qidxs, cb.grid_packed_abs)
    Wq = O.decompress_e8p(P.Qidxs).astype(np.float64)
    x64 = xh.cpu().numpy().astype(np.float64)
    z64 = x64 @ Wq.T
    tol = 2.0 ** -10 * np.abs(z64) + 2.0 ** -21 * (np.abs(x64) @ np.abs(Wq).T) + 1e-7
    assert np.all(np.abs(z.cpu().numpy().astype(np.float64) - z64) <= tol)
    z2 = torch.ops.quip_lib.e8p_mm_skinny(xh[3:3 + 2].contiguous(), layer.Qidxs, cb.grid_packed_abs)
    assert torch.equal(z[3:5], z2), "a row's result does not depend on the other rows"
    if M > 40:   # ... nor on the chunk it falls into
        z3 = torch.ops.quip_lib.e8p_mm_skinny(xh[M - 7:].contiguous(), layer.Qidxs, cb.grid_packed_abs)
        assert torch.equal(z[M - 7:], z3)
    if M < 32:
        x = torch.from_numpy(rng.standard_normal((M, fin)).astype(np.float16)).to(DEV)
        with torch.no_grad():
            y = layer(x)
        What = O.qlinear_dense_weight(P)
        x64m = x.cpu().numpy().astype(np.float64)
        ref = O.qlinear_forward(P, x64m, "exact", What)
        assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= O.ulp_bound(P, x64m, What))



@pytest.mark.parametrize("fin,fout,M", [(4096, 4096, 6), (4096, 4096, 16), (4096, 4096, 31), (4096, 11008, 17), (11008, 4096, 9),
                                        (256, 688, 12), (4096, 4096, 100), (1024, 512, 40), (128, 64, 2)])
def test_single_pass_skinny_kernel_rvq4(fin, fout, M):
    """E8P12RVQ4B through the fp16-MFMA skinny kernel's RVQ4 mode (csrc/e8p_skinny_gemm.hip; the 1 < M < 32 use of the
    reference's tinygemm kernel with BLayout_E8RVQ4, origin_order.cu:337-385): weights = the dense W the reference's
    decompress writes (bit for bit: checked through the op), product against float64 of the same fp16 operands, the module
    inside the stated ulp bound, rows independent of their batch"""
    P = O.make_layer("E8P12RVQ4B", fin, fout, seed=fin + fout + 1)
    layer = _layer(P)
    cb = layer.codebook
    rng = np.random.default_rng(M + fout)
    k, n = layer.q_in_features, layer.q_out_features
    if not cb.skinny_supported(M, n, k):
        pytest.skip("k % 128 != 0: rows mode / generic path")
    xh = torch.from_numpy(rng.standard_normal((M, k)).astype(np.float16)).to(DEV)
    z = torch.ops.quip_lib.e8prvq4_mm_skinny(xh, layer.Qidxs, cb.grid_packed_abs, cb.opt_resid_scale)
    Wd = cb.decompress_weight(layer.Qidxs)
    Wq = O.decompress_e8prvq4(P.Qidxs, cb.opt_resid_scale).astype(np.float64)
    assert np.array_equal(Wd.cpu().numpy().astype(np.float64), Wq)
    x64 = xh.cpu().numpy().astype(np.float64)
    z64 = x64 @ Wq.T
    tol = 2.0 ** -10 * np.abs(z64) + 2.0 ** -21 * (np.abs(x64) @ np.abs(Wq).T) + 1e-7
    assert np.all(np.abs(z.cpu().numpy().astype(np.float64) - z64) <= tol)
    # an identity row picks single weights out: the kernel's weights ARE the dense W's
    e = torch.zeros(min(M, 8), k, dtype=torch.float16, device=DEV)
    cols = [0, 1, 7, 8, k // 2 + 3, k - 9, k - 8, k - 1][:e.shape[0]]
    for i, c in enumerate(cols):
        e[i, c] = 1.0
    ze = torch.ops.quip_lib.e8prvq4_mm_skinny(e, layer.Qidxs, cb.grid_packed_abs, cb.opt_resid_scale)
    assert torch.equal(ze, Wd[:, cols].T.contiguous())
    if M >= 5:
        z2 = torch.ops.quip_lib.e8prvq4_mm_skinny(xh[3:3 + 2].contiguous(), layer.Qidxs, cb.grid_packed_abs, cb.opt_resid_scale)
        assert torch.equal(z[3:5], z2), "a row's result does not depend on the other rows"
    if M > 40:   # ... nor on the chunk it falls into
        z3 = torch.ops.quip_lib.e8prvq4_mm_skinny(xh[M - 7:].contiguous(), layer.Qidxs, cb.grid_packed_abs, cb.opt_resid_scale)
        assert torch.equal(z[M - 7:], z3)
    x = torch.from_numpy(rng.standard_normal((M, fin)).astype(np.float16)).to(DEV)
    with torch.no_grad():
        y = layer(x)
    if 5 < M < 32:
        assert layer.regime(M) == "skinny_fp16"
    What = O.qlinear_dense_weight(P)
    x64m = x.cpu().numpy().astype(np.float64)
    ref = O.qlinear_forward(P, x64m, "exact", What)
    assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= O.ulp_bound(P, x64m, What))


@pytest.mark.parametrize("cbid", ["D4", "HI", "E8P12RVQ3B"])
@pytest.mark.parametrize("fin,fout,M", [(4096, 4096, 6), (4096, 4096, 31), (4096, 11008, 17), (11008, 4096, 9), (256, 688, 12),
                                        (4096, 4096, 100), (1024, 512, 40)])
def test_single_pass_skinny_kernel_d4_hi(cbid, fin, fout, M):
    """D4, HI and E8P12RVQ3B through the skinny kernel's table / nibble / 3-byte modes: the weights are the dense W of the reference's decompress
    bit for bit (identity rows), the product agrees with float64 of the same fp16 operands, the module sits inside the stated
    bound, rows do not depend on their batch"""
    P = O.make_layer(cbid, fin, fout, seed=fin + fout + 2)
    layer = _layer(P)
    cb = layer.codebook
    rng = np.random.default_rng(M + fout)
    k, n = layer.q_in_features, layer.q_out_features
    if not cb.skinny_supported(M, n, k):
        pytest.skip("k % 128 != 0: rows mode / generic path")
    xh = torch.from_numpy(rng.standard_normal((M, k)).astype(np.float16)).to(DEV)
    z = cb.mm_skinny(xh, layer.Qidxs)
    Wd = cb.decompress_weight(layer.Qidxs)
    Wq = O.decompress(cbid, P.Qidxs, getattr(cb, "opt_resid_scale", 0.0)).astype(np.float64)
    assert np.array_equal(Wd.cpu().numpy().astype(np.float64), Wq)
    x64 = xh.cpu().numpy().astype(np.float64)
    z64 = x64 @ Wq.T
    tol = 2.0 ** -10 * np.abs(z64) + 2.0 ** -21 * (np.abs(x64) @ np.abs(Wq).T) + 1e-7
    assert np.all(np.abs(z.cpu().numpy().astype(np.float64) - z64) <= tol)
    e = torch.zeros(8, k, dtype=torch.float16, device=DEV)
    cols = [0, 1, 3, 4, 7, k // 2 + 5, k - 8, k - 1]
    for i, c in enumerate(cols):
        e[i, c] = 1.0
    assert torch.equal(cb.mm_skinny(e, layer.Qidxs), Wd[:, cols].T.contiguous())
    z2 = cb.mm_skinny(xh[3:3 + 2].contiguous(), layer.Qidxs)
    assert torch.equal(z[3:5], z2), "a row's result does not depend on the other rows"
    if M > 40:
        assert torch.equal(z[M - 7:], cb.mm_skinny(xh[M - 7:].contiguous(), layer.Qidxs))
    x = torch.from_numpy(rng.standard_normal((M, fin)).astype(np.float16)).to(DEV)
    with torch.no_grad():
        y = layer(x)
    if 5 < M < (24 if cbid == "D4" else 32):
        assert layer.regime(M) == "skinny_fp16"
    What = O.qlinear_dense_weight(P)
    x64m = x.cpu().numpy().astype(np.float64)
    ref = O.qlinear_forward(P, x64m, "exact", What)
    assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - ref) <= O.ulp_bound(P, x64m, What))


@pytest.mark.parametrize("fin,fout,M", [(4096, 4096, 2), (4096, 4096, 7), (256, 688, 3), (4096, 11008, 4)])
def test_skinny_rows_rvq4_on_matrix_core_path(fin, fout, M):
    """E8P12RVQ4B rows: the same rows-mode GEMV on the int16 view of the codes (virtual 2k-wide rows)"""
    test_skinny_rows_on_matrix_core_path(fin, fout, M, cbid="E8P12RVQ4B")


@pytest.mark.parametrize("cbid,fin,fout,M", [("D4", 4096, 4096, 5), ("D4", 4096, 11008, 9), ("D4", 11008, 4096, 3),
                                              ("HI", 4096, 4096, 4), ("HI", 256, 688, 7), ("E8P12RVQ3B", 4096, 4096, 4),
                                              ("E8P12RVQ3B", 256, 688, 5), ("E8P12RVQ3B", 4096, 11008, 2)])
def test_skinny_rows_other_codebooks_on_matrix_core_path(cbid, fin, fout, M):
    """D4 / HI (D4 table mode) and E8P12RVQ3B (E81B table mode) rows: same rows-mode GEMV, each row bit identical to
    its bs=1 result"""
    test_skinny_rows_on_matrix_core_path(fin, fout, M, cbid=cbid)


@pytest.mark.parametrize("fin,fouts", [(4096, (4096, 4096, 4096)), (4096, (11008, 11008)), (256, (256, 688))])
def test_rvq3_grouped_planes_path_equals_single_calls(fin, fouts):
    """E8P12RVQ3B bs=1: grouped launches on the matrix-core GEMV (the checkpoint's 3-byte codes + E81B table mode)
    give exactly what the modules give one by one; the kernel reads Qidxs itself, so an in-place update of the
    codes is seen by the next call"""
    from quip_for_all_amd.qlinear import forward_group
    layers = [_layer(O.make_layer("E8P12RVQ3B", fin, fo, seed=fin + fo + i)) for i, fo in enumerate(fouts)]
    assert all(l.codebook.planes_supported(l.q_out_features, l.q_in_features) for l in layers)
    rng = np.random.default_rng(fin)
    x = torch.from_numpy(rng.standard_normal((1, fin)).astype(np.float16)).to(DEV)
    with torch.no_grad():
        single = [l(x) for l in layers]
        group = forward_group(layers, x)
        for a, b in zip(single, group):
            assert torch.equal(a, b)
        l0 = layers[0]
        l0.Qidxs.copy_(layers[1].Qidxs[:l0.Qidxs.shape[0]] if layers[1].Qidxs.shape == l0.Qidxs.shape else l0.Qidxs.flip(0))
        y_new = l0(x)
        l0.train()
        y_dense = l0(x)          # training branch: x @ calc_weight() from the CURRENT codes
        l0.eval()
    assert not torch.equal(y_new, single[0])
    tol = 6 * 2.0 ** -11 * (y_dense.float().abs() + 4 * y_dense.float().pow(2).mean().sqrt()) + 1e-3
    assert torch.all((y_new.float() - y_dense.float()).abs() <= tol)


def test_out_transform_group_mixed_widths_equals_single():
    """grouped-query shapes: q_proj (8192 wide) and k / v_proj (1024 wide) output transforms in ONE launch give
    exactly what the per-module launches give (also 4096 + 1024 + 512)"""
    from quip_for_all_amd.qlinear import out_transform_group
    for fin, fouts in ((1024, (8192, 1024, 1024)), (512, (4096, 1024, 512))):
        layers = [_layer(O.make_layer("E8P12", fin, fo, seed=fin + fo + i)) for i, fo in enumerate(fouts)]
        rng = np.random.default_rng(fin)
        zs = [torch.from_numpy(rng.standard_normal((1, fo)).astype(np.float16)).to(DEV) for fo in fouts]
        res = [torch.from_numpy(rng.standard_normal((1, fo)).astype(np.float16)).to(DEV) for fo in fouts]
        with torch.no_grad():
            grouped = out_transform_group(layers, zs, residual=res)
            for l, z, r, g in zip(layers, zs, res, grouped):
                (single,) = out_transform_group([l], [z], residual=[r])
                assert torch.equal(single, g)


@pytest.mark.parametrize("n,k", [(4096, 4096), (4096, 11008), (11008, 4096), (64, 2048)])
def test_single_pass_skinny_kernel_repeated_launches(n, k):
    """the same codes, fresh activations, 40 launches per row count back to back with other kernels in between
    (whatever those leave behind in LDS and in registers): every launch against the fp32 product.  K = 11008 gives
    the 16 K-slices of a workgroup unequal lengths (5 and 6 units)."""
    import quip_for_all_amd as Q
    cb = Q.codebook.codebook_id["E8P12"](inference=True).to(DEV)
    g = torch.Generator(device="cpu").manual_seed(n + k)
    q = torch.randint(-32768, 32767, (n, k // 8), dtype=torch.int32, generator=g).to(torch.int16).to(DEV)
    W = cb.decompress_weight(q).float()
    for M in (2, 6, 16, 17, 31):
        for it in range(40):
            x = torch.randn(M, k, generator=g).half().to(DEV)
            y = torch.ops.quip_lib.e8p_mm_skinny(x, q, cb.grid_packed_abs).float()
            ref = x.float() @ W.T
            assert bool(((y - ref).abs() <= 2e-3 * ref.abs().max()).all()), (M, it)


@pytest.mark.parametrize("cbid", ["E8P12", "E8P12RVQ4B", "E8P12RVQ3B", "D4", "HI"])
@pytest.mark.parametrize("M", [1, 7, 40])
def test_reference_op_sequence_stays_selectable(cbid, M):
    """QuantLinear.reference_ops (QUIP_REFERENCE_OPS=1): the forward through the reference's op sequence -- fp16
    Hadamard, `*_mm_origorder` / decompress + dense GEMM -- equals composing those ops by hand, sits inside the
    stated bound, and the default (digit-plane / skinny / fused) forward agrees with it to the same bound"""
    import math
    P = O.make_layer(cbid, 1408 if cbid != "E8P12RVQ3B" else 1024, 688, seed=77)
    layer = _layer(P)
    x = np.random.default_rng(M).standard_normal((M, layer.in_features)).astype(np.float16)
    xd = torch.from_numpy(x).to(DEV)
    with torch.no_grad():
        y_fast = layer(xd)
        layer.reference_ops = True
        y_ref = layer(xd)
        layer.reference_ops = False
        cb = layer.codebook
        L_in = layer.q_in_features // layer.K_left
        xh = torch.ops.quip_lib.had_transform_fused(
            xd, layer.q_in_features, layer.q_in_features, layer.K_left, layer._had("had_left"), True,
            layer._vec(layer.SU), None, None, None, layer.wscale_float / math.sqrt(L_in), None, None, 1e-5, None)
        z = cb.mm(xh, layer.Qidxs) if M < cb.mm_threshold else xh @ cb.decompress_weight(layer.Qidxs).T
        L_out = layer.q_out_features // layer.K_right
        y_hand = torch.ops.quip_lib.had_transform_fused(
            z, layer.out_features, layer.q_out_features, layer.K_right, layer._had("had_right"), False, None,
            layer._vec(layer.Wscale) if layer.per_channel else None, layer._vec(layer.SV), layer._vec(layer.bias),
            1.0 / math.sqrt(L_out), None, None, 1e-5, None)
    assert torch.equal(y_ref, y_hand)
    What = O.qlinear_dense_weight(P)
    x64 = x.astype(np.float64)
    ref = O.qlinear_forward(P, x64, "exact", What)
    bound = O.ulp_bound(P, x64, What)
    assert np.all(np.abs(y_ref.cpu().numpy().astype(np.float64) - ref) <= bound)
    assert np.all(np.abs(y_fast.cpu().numpy().astype(np.float64) - ref) <= bound)


def test_empty_batch_returns_empty():
    """zero rows: QuantLinear.forward and quip_lib::hadamard return empty tensors of the right shape, no launch"""
    P = O.make_layer("E8P12", 256, 688, seed=2)
    layer = _layer(P)
    with torch.no_grad():
        y = layer(torch.empty(0, 256, device=DEV, dtype=torch.float16))
        y3 = layer(torch.empty(2, 0, 256, device=DEV, dtype=torch.bfloat16))
    assert tuple(y.shape) == (0, 688) and y.dtype == torch.float16
    assert tuple(y3.shape) == (2, 0, 688) and y3.dtype == torch.bfloat16
    h = torch.ops.quip_lib.hadamard(torch.empty(0, 512, device=DEV, dtype=torch.float16), 1.0)
    assert tuple(h.shape) == (0, 512)


@pytest.mark.parametrize("cb,fin,fout", [("E8P12", 1024, 512), ("E8P12", 688, 256), ("D4", 1024, 512),
                                         ("E8P12RVQ4B", 1024, 512), ("E8P12RVQ3B", 1024, 512), ("HI", 1024, 512)])
def test_non_finite_activation_rows_stay_non_finite(cb, fin, fout):
    """an inf or a NaN in an activation row makes that whole output row non-finite in every kernel regime (the
    transform spreads it over the row, as in the reference's fp path) and leaves the other rows alone: the
    digit-plane paths carry it in the block exponent word, where the integer arithmetic would lose it"""
    P = O.make_layer(cb, fin, fout, seed=5)
    layer = _layer(P)
    g = torch.Generator(device="cpu").manual_seed(3)
    for M in (1, 3, 8, 40):
        for bad in (float("nan"), float("inf"), -float("inf")):
            x = torch.randn(M, fin, generator=g).half().to(DEV)
            row = M // 2
            x[row, fin - 3] = bad
            with torch.no_grad():
                y = layer(x).float()
                clean = layer(torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0)).float()
            assert not torch.isfinite(y[row]).any(), (M, bad)
            keep = [r for r in range(M) if r != row]
            assert torch.isfinite(y[keep]).all(), (M, bad)
            assert torch.equal(y[keep], clean[keep]), (M, bad)
