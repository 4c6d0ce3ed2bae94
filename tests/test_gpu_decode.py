"""Model-level GPU test: the captured bs=1 decode step of a tiny random-init Llama equals (a) its
own eager step token for token and (b) a float64 numpy re-implementation whose linear layers go
through the CPU oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import quip_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(m):
    """oracle parameter record from a QuantLinear's buffers"""
    g = lambda t: None if t is None else t.detach().cpu().numpy()  # noqa: E731
    return O.QLinearParams(codebook=m.codebook.id, in_features=m.in_features, out_features=m.out_features,
                           Qidxs=g(m.Qidxs), SU=g(m.SU), SV=g(m.SV), Wscale=g(m.Wscale),
                           wscale_float=m.wscale_float, had_left=g(m.had_left), had_right=g(m.had_right),
                           K_left=m.K_left, K_right=m.K_right, q_in=m.q_in_features, q_out=m.q_out_features,
                           bias=None, per_channel=False,
                           resid_scale=float(getattr(m.codebook, "opt_resid_scale", 0.0)))


def _ref_logits(dec, tokens):
    """float64 reference of the decoder for a token sequence; returns logits of the last position"""
    s = dec.s
    P = [{k: (_params(v) if hasattr(v, "Qidxs") else v.float().cpu().numpy().astype(np.float64))
          for k, v in L.items()} for L in dec.layers]
    W = [{k: O.qlinear_dense_weight(p) for k, p in L.items() if isinstance(p, O.QLinearParams)} for L in P]
    emb = dec.embed.float().cpu().numpy().astype(np.float64)
    cos, sin = dec.cos.cpu().numpy().astype(np.float64), dec.sin.cpu().numpy().astype(np.float64)

    def rms(h, w):
        return h / np.sqrt((h * h).mean() + s.rms_eps) * w

    def rope(x, p):
        d = x.shape[-1] // 2
        rot = np.concatenate([-x[..., d:], x[..., :d]], -1)
        return x * cos[p] + rot * sin[p]
    ks = [[] for _ in P]
    vs = [[] for _ in P]
    for p, t in enumerate(tokens):
        h = emb[t]
        for i, L in enumerate(P):
            x = rms(h, L["ln1"])
            lin = lambda name, v: O.qlinear_forward(L[name], v[None], "exact", W[i][name])[0]  # noqa: E731
            q = rope(lin("q", x).reshape(s.heads, s.head_dim), p)
            k = rope(lin("k", x).reshape(s.kv_heads, s.head_dim), p)
            v = lin("v", x).reshape(s.kv_heads, s.head_dim)
            ks[i].append(k)
            vs[i].append(v)
            lo = max(0, p + 1 - dec.window) if getattr(dec, "window", 0) else 0     # sliding window: keys (p - window, p]
            K, V = np.stack(ks[i][lo:], 1), np.stack(vs[i][lo:], 1)          # (kv, T, d)
            rep = s.heads // s.kv_heads
            out = np.empty((s.heads, s.head_dim))
            for hh in range(s.heads):
                sc = K[hh // rep] @ q[hh] / np.sqrt(s.head_dim)
                w = np.exp(sc - sc.max())
                out[hh] = (w / w.sum()) @ V[hh // rep]
            h = h + lin("o", out.reshape(-1))
            x = rms(h, L["ln2"])
            gte = lin("gate", x)
            h = h + lin("down", gte / (1 + np.exp(-gte)) * lin("up", x))
        logits = rms(h, dec.final_norm.float().cpu().numpy().astype(np.float64)) @ \
            dec.lm_head.float().cpu().numpy().astype(np.float64).T
    return logits


def _ref_logits_deep(dec, tokens):
    """the same float64 model, LAYER-major (all positions through block i before block i + 1: a block's attention only reads
    its own keys of earlier positions), so that only ONE block's dense float64 weights exist at a time -- what lets the model
    run over several full-size blocks (1.6 GB of float64 per Llama-2-7B block, 6.8 GB per Llama-2-70B block)"""
    s = dec.s
    emb = dec.embed.float().cpu().numpy().astype(np.float64)
    cos, sin = dec.cos.cpu().numpy().astype(np.float64), dec.sin.cpu().numpy().astype(np.float64)

    def rms(h, w):
        return h / np.sqrt((h * h).mean() + s.rms_eps) * w

    def rope(x, p):
        d = x.shape[-1] // 2
        rot = np.concatenate([-x[..., d:], x[..., :d]], -1)
        return x * cos[p] + rot * sin[p]
    hs = [emb[t].copy() for t in tokens]
    rep = s.heads // s.kv_heads
    for L in dec.layers:
        Pl = {k: (_params(v) if hasattr(v, "Qidxs") else v.float().cpu().numpy().astype(np.float64)) for k, v in L.items()}
        Wl = {k: O.qlinear_dense_weight(p) for k, p in Pl.items() if isinstance(p, O.QLinearParams)}
        lin = lambda name, v: O.qlinear_forward(Pl[name], v[None], "exact", Wl[name])[0]  # noqa: E731
        ks, vs = [], []
        for p in range(len(tokens)):
            h = hs[p]
            x = rms(h, Pl["ln1"])
            q = rope(lin("q", x).reshape(s.heads, s.head_dim), p)
            ks.append(rope(lin("k", x).reshape(s.kv_heads, s.head_dim), p))
            vs.append(lin("v", x).reshape(s.kv_heads, s.head_dim))
            K, V = np.stack(ks, 1), np.stack(vs, 1)
            out = np.empty((s.heads, s.head_dim))
            for hh in range(s.heads):
                sc = K[hh // rep] @ q[hh] / np.sqrt(s.head_dim)
                w = np.exp(sc - sc.max())
                out[hh] = (w / w.sum()) @ V[hh // rep]
            h = h + lin("o", out.reshape(-1))
            x = rms(h, Pl["ln2"])
            gte = lin("gate", x)
            hs[p] = h + lin("down", gte / (1 + np.exp(-gte)) * lin("up", x))
        del Wl, Pl
    return rms(hs[-1], dec.final_norm.float().cpu().numpy().astype(np.float64)) @ \
        dec.lm_head.float().cpu().numpy().astype(np.float64).T


def deep_bound_ulps(layers):
    """model-level bound over `layers` full-size blocks, in fp16 ulps of rms(logits): every block adds an independent error of
    <= 4 ulps of its output's scale (the module bound, oracle.ulp_bound; one block measures 1.9 - 3.7), independent errors add
    in quadrature, + 2 for the final norm and the fp16 logits: 4 sqrt(layers) + 2"""
    return 4.0 * float(np.sqrt(layers)) + 2.0


def _ulps_of_rms(got, ref):
    """max |got - ref| in fp16 units in the last place of rms(ref) (2^-10 of the power of two below it)"""
    rms = float(np.sqrt(np.mean(ref * ref)))
    return float(np.max(np.abs(got - ref)) / 2.0 ** (np.floor(np.log2(rms)) - 10))


# model-level bounds in fp16 ulps of rms(logits): twice the maxima observed on MI355X (profiles/r03_model_ulps.txt)
# observed: tiny 5.66 / 4.87 / 5.11, 7B-shaped block on the persistent launch 3.70 (E8P12) / 2.77 - 3.30 (E8P12RVQ4B) / 1.91 (D4) / 2.36 (HI)
_TINY_ULPS = {"E8P12": 12.0, "E8P12RVQ4B": 10.0, "D4": 11.0}
_BLOCK_ULPS = {"E8P12": 8.0, "E8P12RVQ4B": 7.0, "D4": 4.0, "HI": 5.0, "E8P12RVQ3B": 7.0}


@pytest.mark.parametrize("codebook", ["E8P12", "E8P12RVQ4B", "D4", "HI", "E8P12RVQ3B"])
def test_full_size_block_against_float64_model(codebook):
    """ONE Llama-2-7B-shaped decoder block (hidden 4096, 32 heads, n_ffn 11008; random init) for 3 decode steps through
    the captured step -- the persistent block launch for each of the five codebooks it takes -- against the float64
    model whose projections go through the CPU oracle (qlinear.py:87-115, example_generate.py:28-33): 1.6 GB of float64
    weights, the largest model-level check the oracle carries"""
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=4096, ffn=11008, layers=1, heads=32, kv_heads=32, vocab=1024)
    np.random.seed(7)
    dec = D.LlamaDecoder(shape, codebook, max_len=16, device="cuda:0", seed=5, device_init=True)
    assert dec.block_eng
    toks = dec.generate(3, first_token=9, use_graph=True).cpu().numpy()
    got = dec.step_logits.float().cpu().numpy()[0].astype(np.float64)
    assert dec.engine_status() == 0
    ref = _ref_logits(dec, [9, int(toks[0]), int(toks[1])])
    u = _ulps_of_rms(got, ref)
    print(f"7B-shaped block ({codebook}), logits of step 3 vs float64: max {u:.2f} fp16 ulps of rms(logits) = {np.sqrt(np.mean(ref * ref)):.3f}")
    assert u <= _BLOCK_ULPS[codebook], u


def test_full_size_g8_block_against_float64_model():
    """ONE Llama-3-8B-shaped block (hidden 4096, 32 / 8 heads, n_ffn 14336; random init) for 3 decode steps on the persistent
    launch compiled for that shape (decode_block_g8.hip) against the float64 model"""
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=4096, ffn=14336, layers=1, heads=32, kv_heads=8, vocab=1024)
    np.random.seed(23)
    dec = D.LlamaDecoder(shape, "E8P12", max_len=16, device="cuda:0", seed=9, device_init=True)
    assert dec.block_eng and dec.eng_shape == 2
    toks = dec.generate(3, first_token=9, use_graph=True).cpu().numpy()
    got = dec.step_logits.float().cpu().numpy()[0].astype(np.float64)
    assert dec.engine_status() == 0
    ref = _ref_logits(dec, [9, int(toks[0]), int(toks[1])])
    u = _ulps_of_rms(got, ref)
    print(f"8B-shaped block (E8P12, grouped-query launch), logits of step 3 vs float64: max {u:.2f} fp16 ulps of rms(logits) = {np.sqrt(np.mean(ref * ref)):.3f}")
    assert u <= 8.0, u


def test_eight_full_size_blocks_against_float64_model():
    """EIGHT Llama-2-7B-shaped blocks (no test carried the float64 model past 3 blocks) for 3 decode steps
    through the captured step on the persistent launch against the layer-major float64 model.  Bound: deep_bound_ulps(8) =
    4 sqrt(8) + 2 = 13.3 fp16 ulps of rms(logits): the statement of how the per-block error may grow with depth."""
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=4096, ffn=11008, layers=8, heads=32, kv_heads=32, vocab=1024)
    np.random.seed(17)
    dec = D.LlamaDecoder(shape, "E8P12", max_len=16, device="cuda:0", seed=6, device_init=True)
    assert dec.block_eng
    toks = dec.generate(3, first_token=9, use_graph=True).cpu().numpy()
    got = dec.step_logits.float().cpu().numpy()[0].astype(np.float64)
    assert dec.engine_status() == 0
    ref = _ref_logits_deep(dec, [9, int(toks[0]), int(toks[1])])
    u = _ulps_of_rms(got, ref)
    print(f"8 x 7B-shaped blocks (E8P12), logits of step 3 vs float64: max {u:.2f} fp16 ulps of rms(logits) = "
          f"{np.sqrt(np.mean(ref * ref)):.3f} (bound {deep_bound_ulps(8):.1f})")
    assert u <= deep_bound_ulps(8), u


def test_all_thirty_two_blocks_of_the_7b_launch_against_float64_model():
    """the depth the bench runs at.  The FULL Llama-2-7B-shaped stack (32 blocks, E8P12, random init) for two
    decode steps through the captured persistent launch against the layer-major float64 model (one block's 1.6 GB of float64
    weights at a time): deep_bound_ulps(32) = 4 sqrt(32) + 2 = 24.6 fp16 ulps of rms(logits) -- the quadrature-growth assumption
    measured where it is used (bench.py's step-vs-step parity bound is twice this)."""
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=4096, ffn=11008, layers=32, heads=32, kv_heads=32, vocab=1024)
    np.random.seed(29)
    dec = D.LlamaDecoder(shape, "E8P12", max_len=16, device="cuda:0", seed=10, device_init=True)
    assert dec.block_eng
    toks = dec.generate(2, first_token=9, use_graph=True).cpu().numpy()
    got = dec.step_logits.float().cpu().numpy()[0].astype(np.float64)
    assert dec.engine_status() == 0
    ref = _ref_logits_deep(dec, [9, int(toks[0])])
    u = _ulps_of_rms(got, ref)
    print(f"32 x 7B-shaped blocks (E8P12), logits of step 2 vs float64: max {u:.2f} fp16 ulps of rms(logits) = "
          f"{np.sqrt(np.mean(ref * ref)):.3f} (bound {deep_bound_ulps(32):.1f})")
    assert u <= deep_bound_ulps(32), u


@pytest.mark.parametrize("codebook", ["E8P12", "E8P12RVQ4B", "D4"])
def test_tiny_llama_decode(codebook):
    from quip_for_all_amd import decode as D
    np.random.seed(11)                 # the K x K factors come from scipy's / numpy's global generator
    dec = D.LlamaDecoder(D.TINY, codebook, max_len=32, device="cuda:0", seed=3)
    eager = dec.generate(12, first_token=5, use_graph=False).cpu().numpy()
    graph = dec.generate(12, first_token=5, use_graph=True).cpu().numpy()
    np.testing.assert_array_equal(eager, graph)
    # logits of step 3 (tokens 5, eager[0], eager[1]) against the float64 reference
    dec.reset(5)
    with torch.no_grad():
        for _ in range(3):
            logits = dec.step()
    ref = _ref_logits(dec, [5, int(eager[0]), int(eager[1])])
    got = logits.float().cpu().numpy()[0].astype(np.float64)
    u = _ulps_of_rms(got, ref)
    print(f"tiny decoder ({codebook}), logits of step 3 vs float64: max {u:.2f} fp16 ulps of rms")
    assert u <= _TINY_ULPS[codebook], u
    assert int(np.argmax(ref)) == int(eager[2]) or np.sort(ref)[-1] - np.sort(ref)[-2] < 0.05


def test_decoder_byte_count_matches_survey():
    from quip_for_all_amd import decode as D
    s = D.LLAMA2_7B
    # 1.619 GB Qidxs + 5 MB SU/SV + 262 MB lm_head = 1.886 GB per token
    q = s.layers * (4 * s.hidden * s.hidden + 3 * s.hidden * s.ffn) // 4
    suv = s.layers * 2 * (4 * 2 * s.hidden + 2 * (s.hidden + s.ffn) + (s.ffn + s.hidden))
    assert abs((q + suv + s.vocab * s.hidden * 2) / 1e9 - 1.886) < 0.002


@pytest.mark.parametrize("heads,kv_heads,hd,pos", [(32, 32, 128, 0), (32, 32, 128, 37), (8, 2, 128, 200),
                                                   (4, 2, 64, 5), (4, 4, 64, 130), (64, 8, 128, 1000),
                                                   (32, 32, 128, 256), (8, 8, 64, 2047), (16, 2, 128, 4100)])
def test_rope_attn_decode_matches_torch(heads, kv_heads, hd, pos):
    """fused rope + cache append + single-query attention == the eager torch ops it replaces"""
    import quip_for_all_amd  # noqa: F401
    import torch.nn.functional as F
    dev = "cuda"
    g = torch.Generator().manual_seed(pos + heads)
    max_len = pos + 9
    q = torch.randn(heads, hd, generator=g).half().to(dev)
    k = torch.randn(kv_heads, hd, generator=g).half().to(dev)
    v = torch.randn(kv_heads, hd, generator=g).half().to(dev)
    kc = torch.randn(kv_heads, max_len, hd, generator=g).half().to(dev)
    vc = torch.randn(kv_heads, max_len, hd, generator=g).half().to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.arange(max_len, dtype=torch.float32)[:, None] * inv[None, :]
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(dev)
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(dev)
    p = torch.tensor([pos], device=dev)
    kc2, vc2 = kc.clone(), vc.clone()
    out = torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc2, vc2, None)
    # split mode (workspace given; used from 256 positions on): same result up to the merge order, and the
    # workspace is reusable (arrival counters are reset by the merging workgroup)
    from quip_for_all_amd.register_lib import rope_attn_workspace
    ws = rope_attn_workspace(heads, hd, dev)
    for _ in range(3):
        kc3, vc3 = kc.clone(), vc.clone()
        out_s = torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc3, vc3, ws)
        assert torch.equal(kc3, kc2) and torch.equal(vc3, vc2)
        assert (out_s.float() - out.float()).abs().max().item() <= 2e-3 * max(1.0, out.float().abs().max().item())

    def rope(x):
        d = hd // 2
        rot = torch.cat([-x[..., d:], x[..., :d]], -1)
        return (x.float() * cos[pos] + rot.float() * sin[pos]).half()
    qr, kr = rope(q), rope(k)
    kc[:, pos] = kr
    vc[:, pos] = v
    assert torch.equal(kc2, kc) and torch.equal(vc2, vc)          # cache append is bit exact
    rep = heads // kv_heads
    K = kc[:, :pos + 1].double().repeat_interleave(rep, 0)
    V = vc[:, :pos + 1].double().repeat_interleave(rep, 0)
    s = torch.einsum("hd,htd->ht", qr.double(), K) / hd ** 0.5
    ref = torch.einsum("ht,htd->hd", torch.softmax(s, -1), V)
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err


def test_fused_prologue_step_equals_unfused_step():
    """SMALL (hidden 1024) takes the fused-prologue decode step; it must produce exactly the logits
    and tokens of the step built from separate launches (same arithmetic, had_device.hip.h)"""
    from quip_for_all_amd import decode as D
    dec = D.LlamaDecoder(D.SMALL, "E8P12", max_len=32, device="cuda:0", seed=11)
    assert dec.fused_prologue
    # the MLP half on its persistent launch (csrc/decode_engine.hip) differs from the four stage-wise launches by the
    # block exponent and the factor order of down's input transform: compared below with a bound, not bit for bit
    engine = dec.ffn_eng
    dec.ffn_eng = False
    fused_tokens = dec.generate(10, first_token=7, use_graph=False)
    dec.reset(7)
    with torch.no_grad():
        lf = [dec.step().clone() for _ in range(4)]
    assert dec.chain
    dec.chain = False          # the same step with every transform inside the GEMV prologues
    prologue_tokens = dec.generate(10, first_token=7, use_graph=False)
    assert torch.equal(prologue_tokens, fused_tokens)
    dec.fused_prologue = False
    plain_tokens = dec.generate(10, first_token=7, use_graph=False)
    dec.reset(7)
    with torch.no_grad():
        lp = [dec.step().clone() for _ in range(4)]
    assert torch.equal(fused_tokens, plain_tokens)
    for a, b in zip(lf, lp):
        assert torch.equal(a, b)
    dec.fused_prologue = dec.chain = True
    dec.qkv_fused = dec.o_fused = False       # the shapes a 70B model takes: no GEMV-prologue transforms at all
    assert torch.equal(dec.generate(10, first_token=7, use_graph=False), fused_tokens)
    dec.qkv_fused = dec.o_fused = True
    graph_tokens = dec.generate(10, first_token=7, use_graph=True)
    assert torch.equal(graph_tokens, fused_tokens)
    ref = _ref_logits(dec, [7, int(fused_tokens[0]), int(fused_tokens[1])])
    got = lf[2].float().cpu().numpy()[0].astype(np.float64)
    assert np.max(np.abs(got - ref)) <= 0.03 * (np.abs(ref).max() + 1.0)
    if engine:
        dec.ffn_eng = True
        dec.graph = None
        dec.reset(7)
        with torch.no_grad():
            le = [dec.step().clone() for _ in range(3)]
        assert dec.engine_status() == 0
        for a, b in zip(le, lf):
            d = (a.float() - b.float()).abs().max().item()
            assert d <= 2.0 ** -8 * b.float().abs().max().item(), d
        assert torch.equal(dec.generate(10, first_token=7, use_graph=True), fused_tokens)


@pytest.mark.parametrize("n", [1, 7, 512, 32000, 32001, 128256])
def test_argmax_step_matches_torch(n):
    """greedy tail kernel: first index of the maximum (ties!), pos += 1"""
    torch.manual_seed(n)
    lg = (torch.randn(1, n, device="cuda:0") * 3).half()
    if n > 16:
        lg[0, n // 3] = lg.max() + 1          # a clear winner ...
        lg[0, n // 2] = lg[0, n // 3]         # ... tied with a later index: the first one wins
    tok = torch.zeros(1, dtype=torch.long, device="cuda:0")
    pos = torch.full((1,), 41, dtype=torch.long, device="cuda:0")
    torch.ops.quip_lib.argmax_step(lg, tok, pos)
    assert int(tok[0]) == int(lg.float().argmax(-1)[0]) and int(pos[0]) == 42
    if n > 16:
        assert int(tok[0]) == n // 3


def test_llama3_8b_shaped_decoder_graph_equals_eager():
    """grouped-query attention, ffn = 7 x 2048 (K = 7 wide transforms), q / k / v of different widths in one output
    launch, 128 K vocabulary: the captured step equals the eager step and the plain (unfused) step"""
    from quip_for_all_amd.decode import LlamaDecoder, LlamaShape
    shape = LlamaShape(hidden=4096, ffn=14336, layers=2, heads=32, kv_heads=8, vocab=128256)
    dec = LlamaDecoder(shape, max_len=32, device="cuda:0", seed=5, device_init=True)
    a = dec.generate(6, first_token=11, use_graph=True)
    b = dec.generate(6, first_token=11, use_graph=False)
    assert torch.equal(a, b)
    dec.fused_prologue = False
    dec.graph = None
    c = dec.generate(6, first_token=11, use_graph=False)
    assert torch.equal(a, c)


@pytest.mark.parametrize("heads,kv_heads,hd,pos,window", [(32, 32, 128, 37, 8), (8, 2, 128, 200, 64), (4, 4, 64, 130, 131),
                                                          (4, 4, 64, 130, 500), (32, 8, 128, 1000, 300), (8, 8, 64, 2047, 1),
                                                          (16, 2, 128, 4100, 4096), (8, 8, 128, 300, 5)])
def test_rope_attn_decode_sliding_window(heads, kv_heads, hd, pos, window):
    """sliding-window attention (HF config.sliding_window; Mistral): the softmax runs over keys (pos - window, pos] of the
    linear cache; with and without the split-mode workspace; a window that covers the context is the plain launch bit for bit"""
    import quip_for_all_amd  # noqa: F401
    from quip_for_all_amd.register_lib import rope_attn_workspace
    dev = "cuda"
    g = torch.Generator().manual_seed(pos + heads + window)
    max_len = pos + 9
    q = torch.randn(heads, hd, generator=g).half().to(dev)
    k = torch.randn(kv_heads, hd, generator=g).half().to(dev)
    v = torch.randn(kv_heads, hd, generator=g).half().to(dev)
    kc = torch.randn(kv_heads, max_len, hd, generator=g).half().to(dev)
    vc = torch.randn(kv_heads, max_len, hd, generator=g).half().to(dev)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.arange(max_len, dtype=torch.float32)[:, None] * inv[None, :]
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(dev)
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(dev)
    p = torch.tensor([pos], device=dev)
    kc2, vc2 = kc.clone(), vc.clone()
    out = torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc2, vc2, None, window)
    ws = rope_attn_workspace(heads, hd, dev)
    for _ in range(2):
        kc3, vc3 = kc.clone(), vc.clone()
        out_s = torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc3, vc3, ws, window)
        assert torch.equal(kc3, kc2) and torch.equal(vc3, vc2)
        assert (out_s.float() - out.float()).abs().max().item() <= 2e-3 * max(1.0, out.float().abs().max().item())
    if window > pos:
        kc4, vc4 = kc.clone(), vc.clone()
        assert torch.equal(out, torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc4, vc4, None))
    lo = max(0, pos + 1 - window)
    rep = heads // kv_heads
    K = kc2[:, lo:pos + 1].double().repeat_interleave(rep, 0)
    V = vc2[:, lo:pos + 1].double().repeat_interleave(rep, 0)
    d = hd // 2
    qr = (q.float() * cos[pos] + torch.cat([-q[..., d:], q[..., :d]], -1).float() * sin[pos]).half()
    s = torch.einsum("hd,htd->ht", qr.double(), K) / hd ** 0.5
    ref = torch.einsum("ht,htd->hd", torch.softmax(s, -1), V)
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("shape_name,window,plen", [("TINY", 4, 0), ("TINY", 5, 9), ("SMALL", 7, 12)])
def test_decoder_with_sliding_window_against_float64_model(shape_name, window, plen):
    """LlamaDecoder(window=W): captured step (and the batched prompt pass) against the float64 model whose attention sees
    the last W positions only; the window changes the logits (a decoder without one differs)"""
    from quip_for_all_amd import decode as D
    shape = getattr(D, shape_name)
    dec = D.LlamaDecoder(shape, "E8P12", max_len=32, device="cuda", seed=5, window=window)
    assert dec.window == window and not dec.block_eng
    n_new = 6
    if plen:
        prompt = [3, 9, 1, 14, 7, 2, 30, 11, 5, 8, 21, 13][:plen]
        toks = dec.generate(n_new, prompt=prompt, use_graph=True)
        seq = list(prompt) + [int(t) for t in toks]
    else:
        toks = dec.generate(n_new, first_token=3, use_graph=True)
        seq = [3] + [int(t) for t in toks]
    eager = dec.generate(n_new, prompt=seq[:plen] if plen else None, first_token=3, use_graph=False)
    assert [int(t) for t in eager] == [int(t) for t in toks]
    # teacher-forced: the logits of the float64 windowed model at the last consumed position pick the same token
    # (up to fp16 noise at near ties)
    hist = seq[:-1]
    ref = _ref_logits(dec, hist)
    nxt = seq[-1]
    assert ref.max() - ref[nxt] <= 0.03 * (np.abs(ref).max() + 1.0)
    dec.window = 0            # the float64 model of the same weights without the window
    ref_f = _ref_logits(dec, hist)
    dec.window = window
    ref_w = ref
    assert np.abs(ref_w - ref_f).max() > 1e-3, "the window must matter at this length"


@pytest.mark.parametrize("bad_pos", [-1, 64, 10 ** 12])
def test_rope_attn_decode_refuses_positions_outside_the_cache(bad_pos):
    """one step past max_len (or a corrupted position) must not touch memory outside cos / sin / the cache: nothing
    is appended, the output is NaN (visible downstream), also in split mode with a workspace"""
    from quip_for_all_amd.register_lib import rope_attn_workspace
    heads, kvh, hd, max_len = 8, 2, 128, 64
    g = torch.Generator().manual_seed(0)
    q = torch.randn(heads, hd, generator=g).half().cuda()
    k = torch.randn(kvh, hd, generator=g).half().cuda()
    v = torch.randn(kvh, hd, generator=g).half().cuda()
    cos = torch.rand(max_len, hd, generator=g).cuda()
    sin = torch.rand(max_len, hd, generator=g).cuda()
    guard = 4096      # canaries around the caches
    buf = torch.full((2, guard + kvh * max_len * hd + guard,), 1.0, dtype=torch.float16, device="cuda")
    kc = buf[0, guard:guard + kvh * max_len * hd].view(kvh, max_len, hd)
    vc = buf[1, guard:guard + kvh * max_len * hd].view(kvh, max_len, hd)
    before = buf.clone()
    p = torch.tensor([bad_pos], dtype=torch.int64, device="cuda")
    for ws in (None, rope_attn_workspace(heads, hd, "cuda")):
        out = torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc, vc, ws)
        torch.cuda.synchronize()
        assert bool(torch.isnan(out).all())
        assert torch.equal(buf, before)
    # the next valid call is unaffected (split counters untouched)
    p.fill_(3)
    out = torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc, vc, None)
    assert bool(torch.isfinite(out).all())


@pytest.mark.parametrize("shape_name,plen", [("TINY", 6), ("TINY", 45), ("SMALL", 40), ("SMALL", 3)])
def test_batched_prompt_prefill_matches_token_by_token(shape_name, plen):
    """LlamaDecoder.prefill (example_generate.py:36-47: the prompt in ONE batched pass, M >= 32 rows through the fused
    dequant GEMM) fills the KV cache and gives the last-token logits that feeding the prompt token by token through
    the decode step gives (different but equivalent kernels: batch Hadamard / GEMM / SDPA vs bs=1 transforms / GEMV /
    fused attention -> fp16-noise agreement), and generation continues identically from there"""
    from quip_for_all_amd import decode as D
    shape = getattr(D, shape_name)
    dec = D.LlamaDecoder(shape, "E8P12", max_len=64, device="cuda:0", seed=3)
    g = torch.Generator().manual_seed(plen)
    prompt = torch.randint(0, shape.vocab, (plen,), generator=g).cuda()
    # token by token (teacher forced), eager
    dec.reset(int(prompt[0]))
    with torch.no_grad():
        for t in range(plen):
            logits_step = dec.step().float().clone()
            if t + 1 < plen:
                dec.tok.copy_(prompt[t + 1:t + 2].view_as(dec.tok))
    kc_step, vc_step = dec.kcache.clone(), dec.vcache.clone()
    # batched
    dec.kcache.zero_(); dec.vcache.zero_()
    logits_b = dec.prefill(prompt).float()
    assert int(dec.pos) == plen
    scale = float(logits_step.abs().max())
    assert float((logits_b - logits_step).abs().max()) <= 0.03 * (scale + 1.0)
    kerr = (dec.kcache[:, :, :plen].float() - kc_step[:, :, :plen].float()).abs().max()
    verr = (dec.vcache[:, :, :plen].float() - vc_step[:, :, :plen].float()).abs().max()
    assert float(kerr) <= 0.03 * (float(kc_step.abs().max()) + 1.0) and float(verr) <= 0.03 * (float(vc_step.abs().max()) + 1.0)
    # generate(): batched prefill and token-by-token prefill pick tokens that are (within fp16 noise) each other's
    # arg max -- checked teacher forced on the batched run's own tokens
    toks = dec.generate(6, prompt=prompt, batched_prefill=True)
    dec.reset(int(prompt[0]))
    seq = torch.cat([prompt, toks])
    with torch.no_grad():
        for t in range(plen + 5):
            lg = dec.step().float()[0]
            dec.tok.copy_(seq[t + 1:t + 2].view_as(dec.tok))
            if t >= plen - 1:
                tok = int(seq[t + 1])
                assert float(lg.max() - lg[tok]) <= 0.03 * (float(lg.abs().max()) + 1.0), (t, tok)


@pytest.mark.parametrize("heads,kvh,hd,pos,max_len", [(32, 32, 128, 0, 64), (32, 32, 128, 37, 64), (32, 32, 128, 300, 512),
                                                     (8, 8, 128, 5, 32), (16, 16, 64, 11, 32), (4, 4, 64, 3, 16),
                                                     (64, 8, 128, 0, 64), (64, 8, 128, 37, 64), (64, 8, 128, 300, 512),
                                                     (32, 8, 128, 5, 32), (32, 8, 128, 301, 512)])
def test_rope_attn_decode_z_equals_transform_then_attention(heads, kvh, hd, pos, max_len):
    """the attention launch that runs the q / k / v output transforms in its prologue (quip_rope_attn_decode_z_f16)
    against the two launches it replaces: same output, same cache rows, bit for bit -- short contexts and the
    split mode (pos >= 256); multi-head attention and grouped queries (Llama-2-70B: 64 heads on 8 KV heads -- an
    8192-point transform next to two 1024-point ones in one workgroup)"""
    import quip_for_all_amd  # noqa: F401
    from quip_for_all_amd.register_lib import rope_attn_workspace, rope_attn_decode_z_supported
    assert rope_attn_decode_z_supported(heads, kvh, hd)
    ns = [heads * hd, kvh * hd, kvh * hd]
    g = torch.Generator().manual_seed(heads * 1000 + pos)
    zs = [(torch.randn(1, n, generator=g) * 3).half().to(DEV) for n in ns]
    svs = [(torch.randint(0, 2, (n,), generator=g).float() * 2 - 1).mul(torch.rand(n, generator=g) + 0.5).half().to(DEV)
           for n in ns]
    scales = [1.0 / np.sqrt(n) for n in ns]
    ang = torch.arange(max_len, dtype=torch.float32)[:, None] * (1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd)))[None]
    cos = torch.cat([ang.cos(), ang.cos()], -1).to(DEV).contiguous()
    sin = torch.cat([ang.sin(), ang.sin()], -1).to(DEV).contiguous()
    kc0 = torch.randn(kvh, max_len, hd, generator=g).half().to(DEV)
    vc0 = torch.randn(kvh, max_len, hd, generator=g).half().to(DEV)
    p = torch.tensor([pos], dtype=torch.long, device=DEV)
    outs = [torch.ops.quip_lib.had_transform_group([z], [n], n, 1, [None], False, [None], [sv], [None], [sc], [None], [None],
                                                   None, 1e-5, None)[0] for z, n, sv, sc in zip(zs, ns, svs, scales)]
    q, k, v = outs[0].view(heads, hd), outs[1].view(kvh, hd), outs[2].view(kvh, hd)
    kc1, vc1, kc2, vc2 = kc0.clone(), vc0.clone(), kc0.clone(), vc0.clone()
    ws1, ws2 = rope_attn_workspace(heads, hd, DEV), rope_attn_workspace(heads, hd, DEV)
    ref = torch.ops.quip_lib.rope_attn_decode(q, k, v, cos, sin, p, kc1, vc1, ws1)
    got = torch.ops.quip_lib.rope_attn_decode_z(zs, svs, scales, cos, sin, p, kc2, vc2, ws2)
    assert torch.equal(got, ref)
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    assert not torch.equal(kc2[:, pos], kc0[:, pos])          # the row was appended
    # the workspace counters are back to zero: a second launch gives the same result
    got2 = torch.ops.quip_lib.rope_attn_decode_z(zs, svs, scales, cos, sin, p, kc2, vc2, ws2)
    assert torch.equal(got2, ref)


@pytest.mark.parametrize("hidden,ffn,heads,kv_heads", [(1024, 2816, 8, 8), (4096, 2816, 32, 8), (8192, 3584, 64, 8)])
def test_decoder_step_with_transforms_in_the_attention_launch(hidden, ffn, heads, kv_heads):
    """a multi-head model (heads == kv_heads, hidden a power of two), or one with 32 / 64 heads on 8 KV heads (Llama-3-8B /
    Mistral-7B / Llama-2-70B attention shapes), takes the 9-launch block; tokens and logits equal the 10-launch block's"""
    from quip_for_all_amd import decode as D
    shape = D.LlamaShape(hidden=hidden, ffn=ffn, layers=2, heads=heads, kv_heads=kv_heads, vocab=1024)
    dec = D.LlamaDecoder(shape, "E8P12", max_len=32, device="cuda:0", seed=5)
    assert dec.attn_z and dec.fused_prologue and not getattr(dec, "block_eng", False)
    t9 = dec.generate(10, first_token=3, use_graph=False)
    dec.reset(3)
    with torch.no_grad():
        l9 = [dec.step().clone() for _ in range(3)]
    g9 = dec.generate(10, first_token=3, use_graph=True)
    dec.attn_z = False
    t10 = dec.generate(10, first_token=3, use_graph=False)
    dec.reset(3)
    with torch.no_grad():
        l10 = [dec.step().clone() for _ in range(3)]
    assert torch.equal(t9, t10) and torch.equal(g9, t10)
    for a, b in zip(l9, l10):
        assert torch.equal(a, b)


def test_prefill_graph_equals_eager_prefill():
    """the prompt pass replayed from a hipGraph (captured per prompt length): logits, cache rows and position equal
    the eager pass, also for a second prompt of the same length"""
    from quip_for_all_amd import decode as D
    dec = D.LlamaDecoder(D.SMALL, "E8P12", max_len=64, device="cuda:0", seed=9)
    g = torch.Generator().manual_seed(1)
    for rep in range(2):
        toks = torch.randint(0, D.SMALL.vocab, (40,), generator=g)
        dec.reset()
        with torch.no_grad():
            le = dec.prefill(toks).clone()
        ke, ve, pe = dec.kcache.clone(), dec.vcache.clone(), int(dec.pos)
        dec.reset()
        dec.kcache.zero_(); dec.vcache.zero_()
        lg = dec.prefill_graph(toks).clone()
        assert torch.equal(lg, le) and int(dec.pos) == pe == 40
        assert torch.equal(dec.kcache, ke) and torch.equal(dec.vcache, ve)


def test_sampling_step_matches_reference_sampler_semantics():
    """example_generate.py:9-26: temperature + top-k + exponential-race arg-max inside the captured step.
    top_k=1 keeps every logit that is not below the largest one: it is greedy UNLESS the two largest fp16 logits tie,
    where it is a fair race between the tied tokens (so is the reference's sampler).  Round 2's intermittent failure of
    this test was exactly that: 0.55 % of the greedy steps of this random-init model have tied top-2 logits
    (tools/dbg/sampler_ties.py: 7 of 1280 steps, 2 of 80 runs diverged), and the K x K Hadamard factors of the model
    come from scipy's process-global generator, so every process drew another model.  Here the generator is seeded and
    the comparison stops at the first tie.  With top_k=5 every sampled token is one of the 5 largest logits of its own
    step, and the draws differ between steps (the graph-safe generator advances on replay)."""
    from quip_for_all_amd.decode import LlamaDecoder, SMALL as LLAMA_TINY
    np.random.seed(20260929)            # get_hadK(use_rand=True) draws from scipy's / numpy's global generator
    dec = LlamaDecoder(LLAMA_TINY, max_len=64, device="cuda:0", seed=3)
    # greedy tokens and the first step whose two largest logits tie (eager steps: the logits of every step are read)
    dec.reset(7)
    first_tie, greedy_eager = 16, []
    with torch.no_grad():
        for i in range(16):
            lg = dec.step().float()[0]
            top = torch.topk(lg, 2).values
            if float(top[0]) == float(top[1]) and first_tie == 16:
                first_tie = i
            greedy_eager.append(int(dec.tok[0]))
    greedy = dec.generate(16, first_token=7)
    assert greedy.tolist() == greedy_eager
    k1 = dec.generate(16, first_token=7, temperature=0.6, top_k=1)
    assert torch.equal(greedy[:first_tie], k1[:first_tie]), (first_tie, greedy.tolist(), k1.tolist())
    torch.manual_seed(0)
    dec.set_sampling(0.6, 5)
    dec.reset(7)
    dec.capture()
    dec.reset(7)
    toks, n_not_top1 = [], 0
    for _ in range(48):
        dec.graph.replay()
        lg = dec.step_logits.float()[0]
        t = int(dec.tok[0])
        top = torch.topk(lg, 5)
        assert lg[t] >= top.values[-1], (t, float(lg[t]), top.values.tolist())   # ties with the 5th logit stay candidates (logits < pivot are cut)
        n_not_top1 += bool(lg[t] < top.values[0])
        toks.append(t)
    assert n_not_top1 > 0, "48 draws at T=0.6 over 5 candidates never left the arg-max: sampler is not sampling"
    # back to greedy: re-captures and reproduces the greedy tokens
    again = dec.generate(16, first_token=7)
    assert torch.equal(again, greedy), (again.tolist(), greedy.tolist())


def test_greedy_step_as_first_gpu_work_of_a_fresh_process():
    """the captured greedy step as the FIRST GPU work of a fresh process gives the tokens of a warm process (three fresh
    subprocesses here; tools/fresh_process_greedy.py runs twenty and keeps the log)"""
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fresh_process_greedy.py")
    r = subprocess.run([sys.executable, tool, "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "all equal: True" in r.stdout, r.stdout[-2000:]
