"""Second-generation matrix-core GEMV (csrc/e8p_gemv_v2.hip) on the GPU: against the float64 oracle (the bound of
tests/test_gpu_ops.py::_mm_tol), bit identity with the first kernel, K split (workspace left zeroed), grouped
launches, ragged shapes, and the public workspace entry points / dispatcher."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import quip_oracle as O
from tests.test_gpu_ops import DEV, _cb, _mm_tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def Q():
    assert torch.cuda.is_available()
    import quip_for_all_amd as Q
    return Q


def _setup(Q, n, k, seed):
    from quip_for_all_amd import capi
    L = capi.lib()
    P = O.make_layer("E8P12", k, n, seed=seed)
    x = torch.from_numpy(np.random.default_rng(seed).standard_normal((1, k)).astype(np.float16)).to(DEV)
    Qd = torch.from_numpy(P.Qidxs).to(DEV)
    planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    assert L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, st) == 0
    return L, P, x, Qd, planes, st


# (rep, slots, blocks, ksplit, max_waves, runlen); 0 = automatic
VARIANTS = [(0, 0, 0, 0, 0, 0), (32, 1, 0, 0, 0, 1), (24, 2, 0, 0, 12, 0), (16, 3, 0, 0, 0, 1), (32, 2, 0, 2, 0, 2),
            (32, 4, 64, 3, 0, 0), (16, 2, 300, 0, 8, 3),
            # rep 4: nibble mode (csrc/e8p_gemv_v2n.hip: row octets, 512-k segments, two accumulators + the K range's constant part)
            (4, 0, 0, 0, 0, 0), (4, 3, 0, 2, 12, 1), (4, 4, 100, 3, 0, 2), (4, 4, 0, 0, 0, 0), (4, 3, 300, 0, 8, 3)]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n,k", [(1000, 4096), (513, 11008), (96, 8192), (37, 28672), (4, 128), (257, 1152),
                                 (2048, 14336)])
def test_v2_against_oracle_and_first_kernel(Q, variant, n, k):
    L, P, x, Qd, planes, st = _setup(Q, n, k, seed=n + k)
    rep, slots, blocks, ksplit, maxw, runlen = variant
    if ksplit > (k + 1023) // 1024:
        pytest.skip("more K parts than 1024-k segments")
    grid = _cb(Q, "E8P12").grid_packed_abs
    ws = torch.zeros(L.quip_e8p_gemv_v2_workspace_bytes(n) // 4, dtype=torch.int32, device=DEV)
    y = torch.full((1, n), float("nan"), dtype=torch.float16, device=DEV)
    rc = L.quip_e8p_gemv_v2_tuned(planes.data_ptr(), Qd.data_ptr(), grid.data_ptr(), y.data_ptr(), ws.data_ptr(), n, k,
                                  rep, slots, blocks, ksplit, maxw, runlen, None, st)
    assert rc == 0
    W64 = O.decompress_e8p(P.Qidxs).astype(np.float64)
    x64 = x.cpu().numpy().astype(np.float64)
    y64 = x64 @ W64.T
    err = np.abs(y.cpu().numpy().astype(np.float64) - y64)
    assert np.all(err <= _mm_tol(x64, W64, y64)), err.max()
    assert int(ws.abs().max()) == 0, "workspace must be left zeroed"
    y1 = torch.empty_like(y)
    assert L.quip_e8p_gemv_tuned(planes.data_ptr(), Qd.data_ptr(), grid.data_ptr(), y1.data_ptr(), n, k, 4, 0, 0, 0, 0,
                                 0, 0, None, st) == 0
    assert torch.equal(y.view(torch.int16), y1.view(torch.int16)), "exact integer sums: same bits as the first kernel"
    # a second launch on the same workspace (the zero invariant holds) gives the same bits
    y2 = torch.empty_like(y)
    assert L.quip_e8p_gemv_v2_tuned(planes.data_ptr(), Qd.data_ptr(), grid.data_ptr(), y2.data_ptr(), ws.data_ptr(), n,
                                    k, rep, slots, blocks, ksplit, maxw, runlen, None, st) == 0
    assert torch.equal(y.view(torch.int16), y2.view(torch.int16))


# (16, 20000): a small first problem next to a large one -- more row blocks than the first problem has accumulator
# words: the arrival counters sit behind ALL accumulators (round 2 put them behind the first problem's)
@pytest.mark.parametrize("ns,k", [((512, 64, 64), 8192), ((1000, 1000), 4096), ((300, 8, 1024), 2048), ((96, 96), 28672),
                                  ((16, 20000), 8192), ((8, 8, 12000), 4096)])
@pytest.mark.parametrize("variant", [(0, 0, 0, 0, 0, 0), (16, 2, 0, 0, 0, 0), (32, 2, 0, 2, 0, 0), (4, 0, 0, 0, 0, 0), (4, 3, 0, 2, 0, 0)])
def test_v2_group_equals_single_launches(Q, ns, k, variant):
    from quip_for_all_amd import capi
    L = capi.lib()
    grid = _cb(Q, "E8P12").grid_packed_abs
    rep, slots, blocks, ksplit, maxw, runlen = variant
    st = torch.cuda.current_stream().cuda_stream
    cnt = len(ns)
    sets = [_setup(Q, n, k, seed=100 + i + n) for i, n in enumerate(ns)]
    ys = [torch.full((1, n), float("nan"), dtype=torch.float16, device=DEV) for n in ns]
    ws = torch.zeros(sum(L.quip_e8p_gemv_v2_workspace_bytes(n) for n in ns) // 4, dtype=torch.int32, device=DEV)
    vp = ctypes.c_void_p * cnt
    rc = L.quip_e8p_gemv_v2_group_tuned(vp(*[s[4].data_ptr() for s in sets]), vp(*[s[3].data_ptr() for s in sets]),
                                        grid.data_ptr(), vp(*[y.data_ptr() for y in ys]), ws.data_ptr(),
                                        (ctypes.c_int32 * cnt)(*ns), cnt, k, rep, slots, blocks, ksplit, maxw, runlen,
                                        None, st)
    assert rc == 0
    assert int(ws.abs().max()) == 0
    for (L_, P, x, Qd, planes, _), y, n in zip(sets, ys, ns):
        y1 = torch.empty_like(y)
        assert L.quip_e8p_gemv_tuned(planes.data_ptr(), Qd.data_ptr(), grid.data_ptr(), y1.data_ptr(), n, k, 4, 0, 0, 0,
                                     0, 0, 0, None, st) == 0
        assert torch.equal(y.view(torch.int16), y1.view(torch.int16))


def test_public_entry_dispatches_and_matches(Q):
    """quip_lib::e8p_gemv_planes (workspace entry, dispatcher) at a Llama-70B layer shape and at a small one: the same
    bits as the first kernel; rows longer than the first kernel supports (2 x 28672) run on the second"""
    from quip_for_all_amd import capi
    L = capi.lib()
    grid = _cb(Q, "E8P12").grid_packed_abs
    for n, k in ((8192, 28672), (28672, 8192), (4096, 4096)):
        g = torch.Generator().manual_seed(n)
        Qd = torch.randint(-32768, 32768, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(DEV)
        x = torch.randn(1, k, generator=g).half().to(DEV)
        planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=DEV)
        st = torch.cuda.current_stream().cuda_stream
        assert L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, st) == 0
        y = torch.ops.quip_lib.e8p_gemv_planes(planes, Qd, grid)
        y1 = torch.empty_like(y)
        assert L.quip_e8p_gemv_tuned(planes.data_ptr(), Qd.data_ptr(), grid.data_ptr(), y1.data_ptr(), n, k, 4, 0, 0, 0,
                                     0, 0, 0, None, st) == 0
        assert torch.equal(y.view(torch.int16), y1.view(torch.int16))
        ref = (x.float() @ torch.ops.quip_lib.decompress_e8p_origorder(Qd[:64], grid).float().T)
        assert torch.allclose(y[:, :64].float(), ref, rtol=2e-3, atol=2e-2)
    # k = 57344: only the second-generation kernel takes it (K split, workspace from the op's cache)
    n, k = 256, 57344
    P = O.make_layer("E8P12", k, n, seed=3)
    x = torch.from_numpy(np.random.default_rng(3).standard_normal((1, k)).astype(np.float16)).to(DEV)
    Qd = torch.from_numpy(P.Qidxs).to(DEV)
    planes = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=DEV)
    assert L.quip_e8p_x_to_planes(x.data_ptr(), planes.data_ptr(), k, torch.cuda.current_stream().cuda_stream) == 0
    y = torch.ops.quip_lib.e8p_gemv_planes(planes, Qd, grid)
    W64 = O.decompress_e8p(P.Qidxs).astype(np.float64)
    x64 = x.cpu().numpy().astype(np.float64)
    y64 = x64 @ W64.T
    assert np.all(np.abs(y.cpu().numpy().astype(np.float64) - y64) <= _mm_tol(x64, W64, y64))
    from quip_for_all_amd import register_lib
    ws = register_lib._GEMV_WS[("cuda", 0, torch.cuda.current_stream().cuda_stream)]
    assert int(ws.abs().max()) == 0
    # without a workspace the C entry refuses what needs a K split and no kernel can serve
    y0 = torch.empty_like(y)
    rc = L.quip_e8p_gemv_planes(planes.data_ptr(), Qd.data_ptr(), grid.data_ptr(), y0.data_ptr(), n, k,
                                torch.cuda.current_stream().cuda_stream)
    assert rc in (-1, -5)


def test_two_streams_run_k_split_gemvs_at_the_same_time(Q):
    """quip_lib::e8p_gemv_planes keeps one K-split workspace per (device, stream): launches issued on two streams at
    the same time neither share accumulators nor arrival counters (round 2: one per device)"""
    from quip_for_all_amd import capi, register_lib as R
    L = capi.lib()
    grid = _cb(Q, "E8P12").grid_packed_abs
    n, k = 8192, 28672                       # K-split kernel
    g = torch.Generator().manual_seed(5)
    Qs = [torch.randint(-32768, 32768, (n, k // 8), generator=g, dtype=torch.int32).to(torch.int16).to(DEV) for _ in range(2)]
    planes = []
    for i in range(2):
        x = torch.randn(1, k, generator=g).half().to(DEV)
        pl = torch.empty(L.quip_e8p_planes_bytes(k), dtype=torch.uint8, device=DEV)
        assert L.quip_e8p_x_to_planes(x.data_ptr(), pl.data_ptr(), k, torch.cuda.current_stream().cuda_stream) == 0
        planes.append(pl)
    ref = [torch.ops.quip_lib.e8p_gemv_planes(planes[i], Qs[i], grid).clone() for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    for it in range(30):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[i].append(torch.ops.quip_lib.e8p_gemv_planes(planes[i], Qs[i], grid))
    torch.cuda.synchronize()
    for i in range(2):
        for y in outs[i]:
            assert torch.equal(y.view(torch.int16), ref[i].view(torch.int16))
    keys = [k_ for k_ in R._GEMV_WS if k_[2] in (streams[0].cuda_stream, streams[1].cuda_stream)]
    assert len(keys) == 2 and R._GEMV_WS[keys[0]].data_ptr() != R._GEMV_WS[keys[1]].data_ptr()
    for k_ in keys:
        assert int(R._GEMV_WS[k_].abs().max()) == 0
