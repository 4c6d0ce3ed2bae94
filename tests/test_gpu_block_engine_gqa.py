"""Persistent decode engine for the grouped-query 8192-wide shape (csrc/decode_block_gqa.hip: Llama-2-70B -- hidden 8192,
64 heads of 128 on 8 KV heads, n_ffn = 7 x 4096): all decoder blocks of a token in one launch, against the stage-wise step
of the same model and against the float64 model.  The launch runs its 8192-point input transforms in another order of the
additions than the stand-alone kernels (strided -> natural, fht_wg512x.hip.h), mixes the 7 x 7 factors of the MLP in fp32 and
takes the block exponent of down's input from a bound: the two paths agree to a few fp16 ulps of rms(logits), not bit for bit
(the bounds below are twice the largest values observed on MI355X)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _decoder(layers, block_engine, max_len=48, seed=3, vocab=2048):
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=8192, ffn=28672, layers=layers, heads=64, kv_heads=8, vocab=vocab)
    old = os.environ.get("QUIP_BLOCK_ENGINE")
    os.environ["QUIP_BLOCK_ENGINE"] = "1" if block_engine else "0"
    np.random.seed(1234 + seed)       # (the K x K factors come from scipy's / numpy's global generator: the same model in every run)
    try:
        dec = D.LlamaDecoder(shape, "E8P12", max_len=max_len, device=DEV, seed=seed, device_init=True)
    finally:
        if old is None:
            os.environ.pop("QUIP_BLOCK_ENGINE", None)
        else:
            os.environ["QUIP_BLOCK_ENGINE"] = old
    return dec


def _same_weights(dst, src):
    """the 7 x 7 factors are drawn from scipy's global generator (get_hadK, quant.py:26-39): copy them over"""
    with torch.no_grad():
        for Ld, Ls in zip(dst.layers, src.layers):
            for k in ("gate", "up", "down"):
                for name in ("had_left", "had_right"):
                    if getattr(Ls[k], name) is not None:
                        getattr(Ld[k], name).copy_(getattr(Ls[k], name))
    if getattr(dst, "block_eng", False):
        dst._init_block_engine()


def _ulps(la, lb):
    rms = lb.pow(2).mean().sqrt().item()
    return (la - lb).abs().max().item() / 2.0 ** (np.floor(np.log2(rms)) - 10)


@pytest.mark.parametrize("layers", [1, 3])
def test_gqa_block_engine_matches_stagewise_step(layers):
    a = _decoder(layers, True)
    b = _decoder(layers, False)
    _same_weights(a, b)
    assert a.block_eng and a.eng_shape == 1 and not b.block_eng
    for dec in (a, b):
        dec.reset(first_token=7)
    worst = 0.0
    with torch.no_grad():
        for t in range(6):
            la = a.step().float().clone()
            lb = b.step().float().clone()
            assert a.engine_status() == 0
            assert torch.isfinite(la).all()
            worst = max(worst, _ulps(la, lb))
            # (a near tie of the top two logits may go either way: keep the two decoders on the same token)
            a.tok.copy_(b.tok)
    print(f"{layers} block(s): max |logit difference| over 6 tokens = {worst:.2f} fp16 ulps of rms(logits)")
    assert worst <= 20.0, worst                      # observed: 8 (1 block), 10 (3 blocks)
    for ca, cb_ in ((a.kcache, b.kcache), (a.vcache, b.vcache)):
        d = (ca[:, :, :6].float() - cb_[:, :, :6].float()).abs().max().item()
        assert d <= 2.0 ** -6 * cb_[:, :, :6].float().abs().max().item(), d


@pytest.mark.parametrize("pos0", [126, 127, 128, 300, 1021])
def test_gqa_block_engine_long_context_split_attention(pos0):
    """from 128 positions on the four workgroups of a head share its attention (every fourth position each, partial softmax
    states merged at the head's first workgroup through one more hand-off); positions on both sides of the threshold"""
    a = _decoder(2, True, max_len=1100)
    b = _decoder(2, False, max_len=1100)
    _same_weights(a, b)
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
            p = pos0 + t
            err = _ulps(la, lb)
            print(f"position {p}: max |logit difference| = {err:.2f} fp16 ulps of rms(logits)")
            assert err <= 20.0, (p, err)                 # observed: 8
            for ca, cb_ in ((a.kcache, b.kcache), (a.vcache, b.vcache)):      # the new rows: written once, by one workgroup
                d = (ca[:, :, p].float() - cb_[:, :, p].float()).abs().max().item()
                assert d <= 2.0 ** -6 * cb_[:, :, p].float().abs().max().item(), d
            a.tok.copy_(b.tok)
    # nothing beyond the rows of the three new positions was written
    assert torch.equal(a.kcache[..., :pos0, :], kc) and torch.equal(a.vcache[..., pos0 + 3:, :], b.vcache[..., pos0 + 3:, :])


def test_gqa_block_engine_captured_generation_matches_stagewise_tokens():
    a = _decoder(2, True, max_len=40)
    b = _decoder(2, False, max_len=40)
    _same_weights(a, b)
    ta = a.generate(24, first_token=5, use_graph=True).cpu().numpy()
    tb = b.generate(24, first_token=5, use_graph=True).cpu().numpy()
    assert a.engine_status() == 0
    same = int((ta == tb).sum())
    print(f"greedy tokens equal: {same} / {len(ta)}")
    first = int(np.argmax(ta != tb)) if same < len(ta) else len(ta)
    assert first >= 8, (ta, tb)              # (a near tie may go the other way later on and the sequences part there)


def test_full_size_70b_block_against_float64_model():
    """ONE Llama-2-70B-shaped decoder block (hidden 8192, 64 / 8 heads, n_ffn 28672; random init, reduced vocabulary) for 3
    decode steps through the captured step on the persistent launch, against the float64 model whose projections go through
    the CPU oracle (qlinear.py:87-115, example_generate.py:28-33)"""
    from tests.test_gpu_decode import _ref_logits, _ulps_of_rms
    np.random.seed(7)
    dec = _decoder(1, True, max_len=16, seed=5, vocab=1024)
    assert dec.block_eng and dec.eng_shape == 1
    toks = dec.generate(3, first_token=9, use_graph=True).cpu().numpy()
    got = dec.step_logits.float().cpu().numpy()[0].astype(np.float64)
    assert dec.engine_status() == 0
    ref = _ref_logits(dec, [9, int(toks[0]), int(toks[1])])
    u = _ulps_of_rms(got, ref)
    print(f"70B-shaped block (E8P12), logits of step 3 vs float64: max {u:.2f} fp16 ulps of rms(logits) = {np.sqrt(np.mean(ref * ref)):.3f}")
    assert u <= 6.0, u                        # observed: 2.58


def test_four_full_size_70b_blocks_against_float64_model():
    """FOUR Llama-2-70B-shaped blocks for 3 decode steps on the persistent launch against the layer-major
    float64 model (one block's 6.8 GB of float64 weights at a time).  Bound: deep_bound_ulps(4) = 4 sqrt(4) + 2 = 10 fp16
    ulps of rms(logits) -- the per-block error (one block: 2.58) growing in quadrature."""
    from tests.test_gpu_decode import _ref_logits_deep, _ulps_of_rms, deep_bound_ulps
    np.random.seed(19)
    dec = _decoder(4, True, max_len=16, seed=8, vocab=1024)
    assert dec.block_eng and dec.eng_shape == 1
    toks = dec.generate(3, first_token=9, use_graph=True).cpu().numpy()
    got = dec.step_logits.float().cpu().numpy()[0].astype(np.float64)
    assert dec.engine_status() == 0
    ref = _ref_logits_deep(dec, [9, int(toks[0]), int(toks[1])])
    u = _ulps_of_rms(got, ref)
    print(f"4 x 70B-shaped blocks (E8P12), logits of step 3 vs float64: max {u:.2f} fp16 ulps of rms(logits) = "
          f"{np.sqrt(np.mean(ref * ref)):.3f} (bound {deep_bound_ulps(4):.1f})")
    assert u <= deep_bound_ulps(4), u


def test_twelve_full_size_70b_blocks_against_float64_model():
    """the quadrature growth behind deep_bound_ulps was measured on 4 of the 80 blocks only.  TWELVE Llama-2-70B-
    shaped blocks (two decode steps: the second one's attention reads a cached row) on the persistent launch against the layer-
    major float64 model: 4 sqrt(12) + 2 = 15.9 fp16 ulps of rms(logits)."""
    from tests.test_gpu_decode import _ref_logits_deep, _ulps_of_rms, deep_bound_ulps
    np.random.seed(23)
    dec = _decoder(12, True, max_len=16, seed=9, vocab=1024)
    assert dec.block_eng and dec.eng_shape == 1
    toks = dec.generate(2, first_token=9, use_graph=True).cpu().numpy()
    got = dec.step_logits.float().cpu().numpy()[0].astype(np.float64)
    assert dec.engine_status() == 0
    ref = _ref_logits_deep(dec, [9, int(toks[0])])
    u = _ulps_of_rms(got, ref)
    print(f"12 x 70B-shaped blocks (E8P12), logits of step 2 vs float64: max {u:.2f} fp16 ulps of rms(logits) = "
          f"{np.sqrt(np.mean(ref * ref)):.3f} (bound {deep_bound_ulps(12):.1f})")
    assert u <= deep_bound_ulps(12), u


def test_norm_bound_planes_of_the_8192_wide_launch_take_spiky_activations():
    """the 8192-wide launch rounds its digit planes against norm bounds too (|H x|_inf <= sqrt(8192) |x|_2: up
    to log2 sqrt(8192) = 6.5 of the 22 bits idle on a one-hot vector, none on a flat one).  A one-hot embedding row, a tiny row and
    RMSNorm weights with a few channels 50 times the rest put both ends through every edge of the launch: logits finite and within
    the usual bound of the stage-wise step (which takes exact maxima)."""
    a = _decoder(2, True)
    b = _decoder(2, False)
    _same_weights(a, b)
    with torch.no_grad():
        for dec in (a, b):
            dec.embed[7].zero_()
            dec.embed[7, 123] = 8.0
            dec.embed[9].mul_(0.02)
            for L in dec.layers:
                for k in ("ln1", "ln2"):
                    L[k][torch.tensor([5, 777, 3000, 8000], device=DEV)] *= 50.0
    for dec in (a, b):
        dec.reset(first_token=7)                          # (rebuilds the launch's descriptors: ln was edited)
    assert a.block_eng and a.eng_shape == 1 and not b.block_eng
    worst = 0.0
    with torch.no_grad():
        for tok in [7, 9, 7, 11]:
            a.tok.fill_(tok)
            b.tok.fill_(tok)
            la = a.step().float().clone()
            lb = b.step().float().clone()
            assert a.engine_status() == 0
            assert torch.isfinite(la).all()
            worst = max(worst, _ulps(la, lb))
    print(f"spiky activations (8192-wide launch): logits within {worst:.2f} fp16 ulps of rms(logits) of the stage-wise step")
    assert worst <= 2.0 * (4.0 * np.sqrt(2) + 2.0), worst


def test_single_copy_mode_keeps_one_copy_of_the_codes_and_still_serves_prompts_and_the_fallback():
    """LlamaDecoder(single_copy=True) drops the modules' row-major code matrices once the tiled copies the launch
    streams exist.  Same tokens and bit-identical logits as the two-copy decoder on the launch; a batched prompt pass and the
    stage-wise fallback step (block engine off) get their matrices back through quip_untile_codes and give the two-copy decoder's
    results bit for bit; a module used outside those paths fails loudly (its Qidxs is None)."""
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=8192, ffn=28672, layers=2, heads=64, kv_heads=8, vocab=2048)
    decs = []
    for single in (False, True):
        np.random.seed(4321)
        decs.append(D.LlamaDecoder(shape, "E8P12", max_len=64, device=DEV, seed=11, device_init=True, single_copy=single))
    two, one = decs
    assert two.block_eng and one.block_eng and one.eng_shape == 1
    assert all(L[k].Qidxs is None for L in one.layers for k in ("q", "k", "v", "o", "gate", "up", "down"))
    assert all(L[k].Qidxs is not None for L in two.layers for k in ("q", "k", "v", "o", "gate", "up", "down"))
    assert one.algorithmic_bytes_per_token() == two.algorithmic_bytes_per_token()
    # 1. the tiled copy is the inverse image: untile(tile(Q)) == Q
    m1, m2 = one.layers[1]["down"], two.layers[1]["down"]
    back = torch.empty_like(m2.Qidxs)
    D.untile_codes(m1._qidxs_tiled, m2.Qidxs.shape[0], m2.Qidxs.shape[1] * m2.Qidxs.element_size(), back)
    assert torch.equal(back, m2.Qidxs)
    # 2. the launch: same logits
    for dec in decs:
        dec.reset(first_token=7)
    with torch.no_grad():
        for _ in range(3):
            la, lb = one.step().clone(), two.step().clone()
            assert one.engine_status() == 0
            assert torch.equal(la, lb)
    # 3. a batched prompt pass (row-major operators on matrices materialised block by block)
    prompt = torch.tensor([5, 9, 1, 33, 7, 2], device=DEV)
    with torch.no_grad():
        pa, pb = one.prefill(prompt).clone(), two.prefill(prompt).clone()
    assert torch.equal(pa, pb)
    assert torch.equal(one.kcache[0][:, :6], two.kcache[0][:, :6])
    assert all(L[k].Qidxs is None for L in one.layers for k in ("q", "k", "v", "o", "gate", "up", "down"))      # (handed back)
    # 4. the stage-wise fallback step
    for dec in decs:
        dec.block_eng = False
        dec.reset(first_token=7)
    with torch.no_grad():
        sa, sb = one.step().clone(), two.step().clone()
    assert torch.equal(sa, sb)
    # 5. outside those paths a module has no codes to read: loud failure, no garbage
    with pytest.raises(Exception):
        one.layers[0]["q"](torch.zeros(1, 8192, dtype=torch.float16, device=DEV))


@pytest.mark.gpu
@pytest.mark.parametrize("rows,row_bytes", [(16, 64), (48, 2048), (256, 7168), (32, 192)])
def test_tile_codes_is_the_stated_permutation(rows, row_bytes):
    """quip_tile_codes (the layout the shape-1 launch streams; include/quip_mi355.h): tiled[rb][c][q][n] = bytes
    [64 c + 16 q, +16) of row 16 rb + n -- bit exact against the index arithmetic written out"""
    from quip_for_all_amd import decode as D
    g = torch.Generator().manual_seed(rows * 7 + row_bytes)
    src = torch.randint(0, 256, (rows, row_bytes), dtype=torch.uint8, generator=g)
    got = D.tile_codes(src.view(torch.int16).cuda()).cpu().numpy()
    ref = np.empty(rows * row_bytes, dtype=np.uint8)
    a = src.numpy()
    i = 0
    for rb in range(rows // 16):
        for c in range(row_bytes // 64):
            for q in range(4):
                for n in range(16):
                    ref[i:i + 16] = a[16 * rb + n, 64 * c + 16 * q:64 * c + 16 * q + 16]
                    i += 16
    assert np.array_equal(got, ref)
    # and a permutation: every byte of the source exactly once (sorted bytes agree)
    assert np.array_equal(np.sort(got), np.sort(a.reshape(-1)))
