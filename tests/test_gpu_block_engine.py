"""Persistent decode engine, stage 2 (csrc/decode_block.hip): all decoder blocks of a token in one launch, against the
stage-wise step of the same model.  Through round 4 the two executed the same rounded operations in the same order and
agreed bit for bit.  Round 5 restructured the launch's 4096-wide edges (gather -> fwd -> strided layout -> rev -> planes,
fht_wg512x.hip.h; block exponents from the norm bound instead of the exact maximum): `rev` adds in another order and the
digits carry up to four bits fewer of the 22, so the launch now sits within a stated number of fp16 ulps of rms(logits) of
the stage-wise step -- like the 8192-wide launch (tests/test_gpu_block_engine_gqa.py) -- and the distance to the float64
model is what tests/test_gpu_decode.py bounds (one block, eight blocks)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _decoder(layers, block_engine, ffn_engine=True, max_len=48, seed=3, codebook="E8P12"):
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=4096, ffn=11008, layers=layers, heads=32, kv_heads=32, vocab=2048)
    old = {k: os.environ.get(k) for k in ("QUIP_BLOCK_ENGINE", "QUIP_FFN_ENGINE")}
    os.environ["QUIP_BLOCK_ENGINE"] = "1" if block_engine else "0"
    os.environ["QUIP_FFN_ENGINE"] = "1" if ffn_engine else "0"
    np.random.seed(1234 + seed)       # (the K x K factors come from scipy's / numpy's global generator: the same model in every run)
    try:
        dec = D.LlamaDecoder(shape, codebook, max_len=max_len, device=DEV, seed=seed, device_init=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return dec


def _same_weights(dst, src):
    """the K x K factors are drawn from scipy's global generator (get_hadK, quant.py:26-39): copy them over"""
    with torch.no_grad():
        for Ld, Ls in zip(dst.layers, src.layers):
            for k in ("gate", "up", "down"):
                for name in ("had_left", "had_right"):
                    if getattr(Ls[k], name) is not None:
                        getattr(Ld[k], name).copy_(getattr(Ls[k], name))
                if hasattr(Ld[k], "_eng_had3"):
                    del Ld[k]._eng_had3
    if getattr(dst, "block_eng", False):
        dst._init_block_engine()


def _ulps(la, lb):
    """max |la - lb| in fp16 ulps of rms(lb)"""
    rms = lb.float().pow(2).mean().sqrt().item()
    return (la.float() - lb.float()).abs().max().item() / 2.0 ** (np.floor(np.log2(rms)) - 10)


@pytest.mark.parametrize("layers", [1, 3])
def test_block_engine_against_stagewise_step(layers):
    """teacher forced (both decoders are fed the stage-wise step's tokens): logits within 2 (4 sqrt(layers) + 2) fp16 ulps of
    rms(logits) of the stage-wise step -- both sit within 4 sqrt(L) + 2 of the float64 model (tests/test_gpu_decode.py:
    deep_bound_ulps) -- observed 6 / 12 for 1 / 3 blocks; new cache rows within 2^-7 of their maximum"""
    a = _decoder(layers, True)
    b = _decoder(layers, False)
    _same_weights(b, a)
    assert a.block_eng and not b.block_eng and b.ffn_eng
    for dec in (a, b):
        dec.reset(first_token=7)
    worst = 0.0
    with torch.no_grad():
        for t in range(6):
            la = a.step().clone()
            lb = b.step().clone()
            assert a.engine_status() == 0 and b.engine_status() == 0
            worst = max(worst, _ulps(la, lb))
            a.tok.copy_(b.tok)
    print(f"{layers} block(s): logits within {worst:.2f} fp16 ulps of rms(logits) of the stage-wise step")
    assert worst <= 2.0 * (4.0 * np.sqrt(layers) + 2.0), worst
    for ca, cb_ in ((a.kcache, b.kcache), (a.vcache, b.vcache)):
        ra = torch.stack([c[:, :6] for c in ca]).float() if isinstance(ca, (list, tuple)) else ca[:, :, :6].float()
        rb = torch.stack([c[:, :6] for c in cb_]).float() if isinstance(cb_, (list, tuple)) else cb_[:, :, :6].float()
        assert (ra - rb).abs().max().item() <= 2.0 ** -7 * rb.abs().max().item()


def test_block_engine_captured_generation_matches_plain_stagewise_tokens():
    """the captured step on the persistent launch against the plain stage-wise step (no engine at all): logits agree
    to the MLP edge's rounding, the greedy tokens of a short generation are the same"""
    a = _decoder(2, True, max_len=40)
    c = _decoder(2, False, ffn_engine=False, max_len=40)
    _same_weights(c, a)
    assert a.block_eng and not c.block_eng and not c.ffn_eng
    ta = a.generate(24, first_token=5, use_graph=True).cpu().numpy()
    la = a.step_logits.float().cpu().numpy()
    tc = c.generate(24, first_token=5, use_graph=True).cpu().numpy()
    lc = c.step_logits.float().cpu().numpy()
    assert a.engine_status() == 0
    same = int((ta == tc).sum())
    print(f"greedy tokens equal: {same} / {len(ta)}; last-step logits max diff {np.abs(la - lc).max():.4f} (|logit| max {np.abs(lc).max():.2f})")
    first = int(np.argmax(ta != tc)) if same < len(ta) else len(ta)
    assert first >= 8, (ta, tc)              # (a near tie may go the other way later on and the sequences part there)
    if same == len(ta):
        assert np.abs(la - lc).max() <= 2.0 ** -8 * np.abs(lc).max()


@pytest.mark.parametrize("pos0", [126, 127, 128, 300, 1021])
def test_block_engine_long_context_split_attention(pos0):
    """from 128 positions on the eight workgroups of a head share its attention (every eighth position each, partial
    softmax states merged at the head's first workgroup through one more hand-off): same caches, logits within the
    rounding of a different summation order of the stage-wise step, status 0; positions on both sides of the threshold"""
    a = _decoder(2, True, max_len=1100)
    b = _decoder(2, False, max_len=1100)
    _same_weights(b, a)
    g = torch.Generator(device=DEV).manual_seed(pos0)
    # a plausible history: random K / V rows for positions < pos0, the same in both decoders
    for dec in (a, b):
        dec.reset(first_token=7)
    with torch.no_grad():
        kc = (torch.randn(a.kcache[..., :pos0, :].shape, generator=g, device=DEV) * 0.5).half()
        vc = (torch.randn(a.vcache[..., :pos0, :].shape, generator=g, device=DEV) * 0.5).half()
        for dec in (a, b):
            dec.kcache[..., :pos0, :].copy_(kc)
            dec.vcache[..., :pos0, :].copy_(vc)
            dec.pos.fill_(pos0)
        for t in range(3):
            la = a.step().float().clone()
            lb = b.step().float().clone()
            assert a.engine_status() == 0
            p = pos0 + t
            # (the stage-wise attention launch splits long contexts its own way: the sums run in another order on either
            #  side of the engine's threshold, so this is a bound, not an identity -- that one holds for the first positions,
            #  test_block_engine_equals_stagewise_step_bit_for_bit)
            rms = lb.pow(2).mean().sqrt().item()
            ulp = 2.0 ** (np.floor(np.log2(rms)) - 10)
            err = (la - lb).abs().max().item() / ulp
            print(f"position {p}: max |logit difference| = {err:.2f} fp16 ulps of rms(logits)")
            assert err <= 16.0, (p, err)          # (twice the largest observed: 8.0)
            # the new cache rows: written once, by the workgroup whose turn the position is
            # (block 0's rows see identical inputs; block 1's inherit the few-ulp difference of block 0's attention)
            for ca, cb_ in ((a.kcache, b.kcache), (a.vcache, b.vcache)):
                d0 = (ca[0, :, p].float() - cb_[0, :, p].float()).abs().max().item()
                assert d0 <= 2.0 ** -8 * cb_[0, :, p].float().abs().max().item(), d0      # (the planes' rounding only)
                d = (ca[1, :, p].float() - cb_[1, :, p].float()).abs().max().item()
                assert d <= 2.0 ** -6 * cb_[1, :, p].float().abs().max().item(), d
            with torch.no_grad():
                a.tok.copy_(b.tok)                        # keep the two on the same token whatever a near tie decides


@pytest.mark.parametrize("codebook,code", [("D4", 1), ("E8P12RVQ4B", 2), ("HI", 3), ("E8P12RVQ3B", 4)])
def test_block_engine_d4_matches_stagewise(codebook, code):
    """the D4 codebook (one table of 256 x 4 bytes, a private copy per lane) and E8P12RVQ4B (virtual rows of twice the
    width against x' = [s x_g | x_g]: twice the digits and items) and HI (a code byte as a D4 code of a row of twice the width)
    and E8P12RVQ3B (the 3-byte codes as RVQ4-style virtual rows whose low codes read the E81B table)
    through the same persistent launch: against the plain
    stage-wise step of the same model -- logits to the MLP edge's rounding, the same greedy tokens"""
    a = _decoder(2, True, max_len=40, codebook=codebook)
    c = _decoder(2, False, ffn_engine=False, max_len=40, codebook=codebook)
    _same_weights(c, a)
    assert a.block_eng and a.eng_codebook == code and not c.block_eng and not c.ffn_eng
    for dec in (a, c):
        dec.reset(first_token=5)
    with torch.no_grad():
        for t in range(4):
            la = a.step().float().clone()
            lc = c.step().float().clone()
            assert a.engine_status() == 0
            assert (la - lc).abs().max().item() <= 2.0 ** -8 * lc.abs().max().item(), (t, (la - lc).abs().max().item())
            a.tok.copy_(c.tok)
    ta = a.generate(24, first_token=5, use_graph=True).cpu().numpy()
    tc = c.generate(24, first_token=5, use_graph=True).cpu().numpy()
    assert a.engine_status() == 0
    same = int((ta == tc).sum())
    print(f"{codebook}: greedy tokens equal: {same} / {len(ta)}")
    assert same >= len(ta) - 2          # (a near tie may go the other way: the MLP edge rounds its block exponent differently)


def _decoder_g8(layers, block_engine, max_len=48, seed=3):
    """Llama-3-8B / Mistral-7B shape: hidden 4096, 32 heads on 8 KV heads, n_ffn 14336 = 7 x 2048"""
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=4096, ffn=14336, layers=layers, heads=32, kv_heads=8, vocab=2048)
    old = os.environ.get("QUIP_BLOCK_ENGINE")
    os.environ["QUIP_BLOCK_ENGINE"] = "1" if block_engine else "0"
    np.random.seed(4321 + seed)
    try:
        return D.LlamaDecoder(shape, "E8P12", max_len=max_len, device=DEV, seed=seed, device_init=True)
    finally:
        if old is None:
            os.environ.pop("QUIP_BLOCK_ENGINE", None)
        else:
            os.environ["QUIP_BLOCK_ENGINE"] = old


@pytest.mark.parametrize("layers", [1, 3])
def test_g8_block_engine_against_stagewise_step(layers):
    """the launch compiled for the grouped-query 4096-wide shape (decode_block_g8.hip: 56 x 256 view of the 14336-wide MLP,
    one or two q / k / v row blocks per workgroup, four query heads per KV head) against the stage-wise step of the same
    model, teacher forced: logits within 2 (4 sqrt(layers) + 2) fp16 ulps of rms(logits), cache rows within 2^-7"""
    a = _decoder_g8(layers, True)
    b = _decoder_g8(layers, False)
    _same_weights(b, a)
    assert a.block_eng and a.eng_shape == 2 and not b.block_eng
    for dec in (a, b):
        dec.reset(first_token=7)
    worst = 0.0
    with torch.no_grad():
        for t in range(6):
            la = a.step().clone()
            lb = b.step().clone()
            assert a.engine_status() == 0
            worst = max(worst, _ulps(la, lb))
            a.tok.copy_(b.tok)
    print(f"G8, {layers} block(s): logits within {worst:.2f} fp16 ulps of rms(logits) of the stage-wise step")
    assert worst <= 2.0 * (4.0 * np.sqrt(layers) + 2.0), worst
    for ca, cb_ in ((a.kcache, b.kcache), (a.vcache, b.vcache)):
        assert (ca[:, :, :6].float() - cb_[:, :, :6].float()).abs().max().item() <= 2.0 ** -7 * cb_[:, :, :6].float().abs().max().item()


@pytest.mark.parametrize("pos0", [127, 300])
def test_g8_block_engine_long_context_split_attention(pos0):
    a = _decoder_g8(2, True, max_len=400)
    b = _decoder_g8(2, False, max_len=400)
    _same_weights(b, a)
    g = torch.Generator(device=DEV).manual_seed(pos0)
    for dec in (a, b):
        dec.reset(first_token=7)
    with torch.no_grad():
        kc = (torch.randn(a.kcache[..., :pos0, :].shape, generator=g, device=DEV) * 0.5).half()
        vc = (torch.randn(a.vcache[..., :pos0, :].shape, generator=g, device=DEV) * 0.5).half()
        for dec in (a, b):
            dec.kcache[..., :pos0, :].copy_(kc)
            dec.vcache[..., :pos0, :].copy_(vc)
            dec.pos.fill_(pos0)
        for t in range(3):
            la = a.step().float().clone()
            lb = b.step().float().clone()
            assert a.engine_status() == 0
            err = _ulps(la, lb)
            print(f"G8 position {pos0 + t}: max |logit difference| = {err:.2f} fp16 ulps of rms(logits)")
            assert err <= 18.0, (pos0 + t, err)
            a.tok.copy_(b.tok)


@pytest.mark.parametrize("g8", [False, True])
def test_norm_bound_planes_take_spiky_activations(g8):
    """the launches round their digit planes against NORM bounds (|H x|_inf <= sqrt(n) |x|_2; |M r|_inf <= |row|_2 |r|_2) instead
    of the exact maximum: tight for a flat vector (|H x|_inf = sqrt(n) |x|_2 when every element agrees in sign with a row of H), log2
    sqrt(n) = 6 bits loose for a one-hot one, typically 4.  A one-hot embedding row and RMSNorm
    weights with a few channels 50 times the rest (the outlier channels of real checkpoints) put both ends through every edge:
    an exponent that let a digit overflow its 22 bits would show as garbage, too loose a one as lost precision -- logits stay
    within the usual bound of the stage-wise step (which takes exact maxima)."""
    mk = _decoder_g8 if g8 else _decoder
    a = mk(2, True)
    b = mk(2, False)
    _same_weights(b, a)
    with torch.no_grad():
        for dec in (a, b):
            dec.embed[7].zero_()
            dec.embed[7, 123] = 8.0
            dec.embed[9].mul_(0.02)                       # a tiny row: far below the norm of the others
            for L in dec.layers:
                for k in ("ln1", "ln2"):
                    L[k][torch.tensor([5, 777, 3000], device=DEV)] *= 50.0
    for dec in (a, b):
        dec.reset(first_token=7)                          # (rebuilds the launch's descriptors: ln was edited)
    assert a.block_eng and not b.block_eng
    worst = 0.0
    with torch.no_grad():
        for t, tok in enumerate([7, 9, 7, 11]):
            a.tok.fill_(tok)
            b.tok.fill_(tok)
            la = a.step().clone()
            lb = b.step().clone()
            assert a.engine_status() == 0
            assert torch.isfinite(la).all()
            worst = max(worst, _ulps(la, lb))
    print(f"spiky activations ({'G8' if g8 else '7B'} launch): logits within {worst:.2f} fp16 ulps of rms(logits) of the stage-wise step")
    assert worst <= 2.0 * (4.0 * np.sqrt(2) + 2.0), worst
