"""Persistent decode engine, stage 1 (csrc/decode_engine.hip): the MLP half of a decoder block in one launch
against (a) the four stage-wise launches it replaces and (b) a float64 evaluation of the same chain through the
CPU oracle's decoder and transforms (qlinear.py:87-115 for gate / up / down, the SiLU product of the block)."""
import math

import numpy as np
import pytest
import torch

from oracle import quip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (hidden, n_ffn): n_ffn = 11 x 256, 43 x 128, 43 x 256 (the last one is Llama-2-7B's block)
SHAPES = [(1024, 2816), (2048, 5504), (4096, 11008)]


def _mlp(hidden, ffn, seed):
    from quip_for_all_amd.decode import random_quant_linear
    g = torch.Generator().manual_seed(seed)
    gate, up = random_quant_linear(hidden, ffn, "E8P12", g, DEV), random_quant_linear(hidden, ffn, "E8P12", g, DEV)
    down = random_quant_linear(ffn, hidden, "E8P12", g, DEV)
    x = (torch.randn(1, hidden, generator=g) * 1.5).to(torch.float16).to(DEV)
    return gate, up, down, x


def _planes(layers, x):
    l0 = layers[0]
    return list(torch.ops.quip_lib.had_transform_planes_group(
        x, l0.q_in_features, l0.K_left, [l._had("had_left") for l in layers], True, [l._vec(l.SU) for l in layers],
        [l.wscale_float / math.sqrt(l0.q_in_features // l0.K_left) for l in layers], None, 1e-5, None, 0.0))


def _stagewise(gate, up, down, planes):
    from quip_for_all_amd.qlinear import _gemv_planes_grouped, gemv_unfused, out_transform_group
    zgu = _gemv_planes_grouped([gate, up], planes)
    g, u = out_transform_group([gate, up], zgu)
    return gemv_unfused(down, u, gate=g)


def _ulps(a, b):
    """difference in fp16 units in the last place of max(|value|, rms of the vector) (the module bound's unit,
    oracle.ulp_bound)"""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    m = np.maximum(np.maximum(np.abs(a64), np.abs(b64)), np.sqrt(np.mean(b64 * b64)))
    ulp = 2.0 ** (np.floor(np.log2(m)) - 10)
    return np.abs(a64 - b64) / ulp


@pytest.mark.parametrize("hidden,ffn", SHAPES)
def test_engine_against_stagewise_launches(hidden, ffn):
    from quip_for_all_amd.qlinear import ffn_engine, ffn_engine_ok
    from quip_for_all_amd.register_lib import ffn_engine_status, ffn_engine_workspace
    gate, up, down, x = _mlp(hidden, ffn, 11)
    assert ffn_engine_ok(gate, up, down)
    ws = ffn_engine_workspace(ffn, gate.K_right, DEV)
    planes = _planes([gate, up], x)
    zs = _stagewise(gate, up, down, planes).cpu().numpy().reshape(-1)
    ze = ffn_engine(gate, up, down, planes, ws).cpu().numpy().reshape(-1)
    assert ffn_engine_status(ws) == 0
    assert np.isfinite(ze).all()
    d = _ulps(ze, zs)
    # same exact integer products; down's input planes come from the same transform with its two commuting factors in
    # the other order and a finer block exponent (exact maximum instead of the norm bound): values on an fp16 rounding
    # boundary may fall to the other side
    print(f"engine vs stage-wise ({hidden}, {ffn}): max {d.max():.2f} ulp, differing {np.mean(d > 0) * 100:.2f} %")
    assert d.max() <= 1.0 and np.mean(d > 0) < 0.25


def _dequant_planes(pl, k):
    kp = (k + 511) // 512 * 512
    b = pl.cpu().numpy()
    dg = b[:3 * kp].reshape(3, kp).view(np.int8).astype(np.int64)
    sh = int(b[3 * kp:3 * kp + 4].view(np.int32)[0])
    X = dg[0] * 65536 + dg[1] * 256 + dg[2]
    return (X[:k].astype(np.float64)) * 2.0 ** (-sh)


@pytest.mark.parametrize("hidden,ffn", SHAPES[:2])
def test_engine_against_float64(hidden, ffn):
    from quip_for_all_amd.qlinear import ffn_engine
    from quip_for_all_amd.register_lib import ffn_engine_status, ffn_engine_workspace
    gate, up, down, x = _mlp(hidden, ffn, 5)
    ws = ffn_engine_workspace(ffn, gate.K_right, DEV)
    planes = _planes([gate, up], x)
    ze = ffn_engine(gate, up, down, planes, ws).float().cpu().numpy().reshape(-1).astype(np.float64)
    assert ffn_engine_status(ws) == 0
    r16 = lambda a: a.astype(np.float16).astype(np.float64)  # noqa: E731
    K = gate.K_right
    outs = []
    for m, pl in zip((gate, up), planes):
        xg = _dequant_planes(pl, hidden)                       # what the GEMV multiplies: exact
        W = O.decompress("E8P12", m.Qidxs.cpu().numpy()).astype(np.float64)
        z = r16(W @ xg)                                        # the mm op's fp16 output (e8p12.py:147-150)
        y = O.matmul_hadU(z[None], m.had_right.cpu().numpy().astype(np.float64), K, ffn)[0]
        outs.append(r16(y * m.SV.detach().cpu().numpy().astype(np.float64)))   # the module's fp16 output
    g, u = outs
    e = u * (g / (1.0 + np.exp(-g))) * down.SU.detach().cpu().numpy().astype(np.float64)
    xd = O.matmul_hadU(e[None], down.had_left.cpu().numpy().astype(np.float64), K, ffn, down.wscale_float, transpose=True)[0]
    Wd = O.decompress("E8P12", down.Qidxs.cpu().numpy()).astype(np.float64)
    zd = Wd @ xd
    rms = np.sqrt(np.mean(zd * zd))
    err = np.abs(ze - zd)
    # One fp16 rounding of the result.  fp32 transforms, the fp16 hi + lo split of the rows and the 22-bit digits are far
    # below it; what shows is the fp16 rounding of g / u (the modules' output type): ~4 ulps of fp32 error ahead of it put
    # about n_ffn * 2 * 2^-11 ~ 10 of those values on the other side than in float64, each moving every output by
    # ~2^-16 rms * |e_i| / rms(e).  Their sum stays below 2^-11.5 rms (observed maxima: 2^-12.2 rms).
    tol = 2.0 ** -11 * np.abs(zd) * 1.001 + 2.0 ** -11.5 * rms
    print(f"engine vs float64 ({hidden}, {ffn}): max err / tol {np.max(err / tol):.3f}, rms {rms:.3f}")
    assert np.all(err <= tol)


def test_engine_launches_share_a_workspace_and_repeat_bit_for_bit():
    """two blocks take turns on one workspace (as the decoder's layers do); every launch gives its block's bits"""
    from quip_for_all_amd.qlinear import ffn_engine
    from quip_for_all_amd.register_lib import ffn_engine_status, ffn_engine_workspace
    hidden, ffn = 4096, 11008
    blocks = [_mlp(hidden, ffn, s) for s in (1, 2)]
    ws = ffn_engine_workspace(ffn, 43, DEV)
    planes = [_planes([b[0], b[1]], b[3]) for b in blocks]
    first = [ffn_engine(b[0], b[1], b[2], p, ws).clone() for b, p in zip(blocks, planes)]
    for it in range(40):
        i = it & 1
        z = ffn_engine(blocks[i][0], blocks[i][1], blocks[i][2], planes[i], ws)
        assert torch.equal(z, first[i]), f"launch {it}"
    assert ffn_engine_status(ws) == 0
