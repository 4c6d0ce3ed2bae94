"""Drop-in check on the GPU: a checkpoint directory in the reference's format is loaded with
`load_quantized_model`, moved to the GPU and run through the stock HF `generate` loop (the
north-star claim); logits and greedy tokens must match the same HF model with every QuantLinear
replaced by a dense nn.Linear holding `calc_weight()` (qlinear.py:144-159 identity)."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
transformers = pytest.importorskip("transformers")

from tests.test_quantizer_host import _fill_random, _tiny_config   # noqa: E402


@pytest.mark.parametrize("codebook", ["E8P12", "E8P12RVQ4B"])
def test_hf_generate_with_loaded_quantized_model(tmp_path, codebook):
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer, load_quantized_model, get_layers
    from quip_for_all_amd.qlinear import QuantLinear
    torch.manual_seed(0)
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    qz = QuipQuantizer(codebook=codebook, inference=True, ft_epochs=0)
    qz.convert_model(model)
    _fill_random(model, seed=3)
    qz.save(model, str(tmp_path))
    q = load_quantized_model(str(tmp_path), device_map={"": "cuda:0"})
    assert q.is_quantized and next(q.parameters()).is_cuda
    # dense twin: QuantLinear -> nn.Linear(calc_weight)
    dense = copy.deepcopy(q)
    dense.is_quantized = False
    for name, layer in get_layers(dense, [QuantLinear]).items():
        W = layer.calc_weight(cache=False).float()          # (q_in, q_out): y = x_pad @ W
        lin = torch.nn.Linear(layer.in_features, layer.out_features, bias=False, device="cuda:0", dtype=torch.float32)
        Wd = W[:layer.in_features, :layer.out_features]
        su = layer.SU.float() if layer.SU is not None else 1.0
        sv = layer.SV.float() if layer.SV is not None else 1.0
        lin.weight.data = ((Wd * su[:, None] if torch.is_tensor(su) else Wd) * (sv[None, :] if torch.is_tensor(sv) else 1.0)).T.contiguous()
        parent = dense
        *path, leaf = name.split(".")
        for p in path:
            parent = parent[int(p)] if p.isdigit() else getattr(parent, p)
        setattr(parent, leaf, lin)
    dense = dense.float()
    ids = torch.tensor([[1, 17, 42, 99, 7, 250]], device="cuda:0")
    with torch.no_grad():
        lq = q(ids).logits.float()
        ld = dense(ids).logits.float()
    err = (lq - ld).abs().max().item()
    assert err <= 0.03 * (ld.abs().max().item() + 1.0), err
    with torch.no_grad():
        out_q = q.generate(ids, max_new_tokens=8, do_sample=False)
        assert out_q.shape == (1, 14) and torch.equal(out_q[:, :6], ids)
        # teacher-forced check of the greedy path: every token the quantized model picked is (within
        # fp16 noise) the arg-max of the dense twin at that position.  Comparing two free-running
        # greedy paths would diverge at the first near-tie of this random-weight model.
        ld = dense(out_q).logits.float()[0]
    for t in range(5, 13):
        tok = int(out_q[0, t + 1])
        margin = (ld[t].max() - ld[t, tok]).item()
        assert margin <= 0.03 * (ld[t].abs().max().item() + 1.0), (t, tok, margin)


def test_fast_decoder_from_loaded_hf_model(tmp_path):
    """load_quantized_model -> LlamaDecoder.from_hf: the captured fast decode loop on the loaded checkpoint
    follows the stock HF forward of the same model (teacher-forced arg-max check, fp16 noise margin)"""
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer, load_quantized_model
    from quip_for_all_amd.decode import LlamaDecoder
    torch.manual_seed(1)
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    _fill_random(model, seed=9)
    qz.save(model, str(tmp_path))
    q = load_quantized_model(str(tmp_path), device_map={"": "cuda:0"})
    dec = LlamaDecoder.from_hf(q, max_len=64)
    toks = dec.generate(12, first_token=5, use_graph=True)
    eager = dec.generate(12, first_token=5, use_graph=False)
    assert torch.equal(toks, eager)
    seq = torch.cat([torch.tensor([5], device="cuda:0"), toks])[None]
    with torch.no_grad():
        logits = q(seq).logits.float()[0]
    for t in range(12):
        tok = int(toks[t])
        margin = (logits[t].max() - logits[t, tok]).item()
        assert margin <= 0.03 * (logits[t].abs().max().item() + 1.0), (t, tok, margin)
    # with a prompt (batched prefill of all but its last token; 40 tokens: the M >= 32 fused GEMM path)
    for prompt in (torch.tensor([5, 17, 3, 99, 42], device="cuda:0"),
                   torch.randint(0, 320, (40,), generator=torch.Generator().manual_seed(4)).cuda()):
        _check_prompt(q, dec, prompt)


def _check_prompt(q, dec, prompt):
    if True:
        ptoks = dec.generate(8, prompt=prompt)
        seq = torch.cat([prompt, ptoks])[None]
        with torch.no_grad():
            logits = q(seq).logits.float()[0]
        for t in range(8):
            row = logits[prompt.numel() - 1 + t]
            margin = (row.max() - row[int(ptoks[t])]).item()
            assert margin <= 0.03 * (row.abs().max().item() + 1.0), (t, margin)


def test_fast_decoder_takes_the_models_rotary_embedding(tmp_path):
    """Llama-3.1-style rope scaling: LlamaDecoder.from_hf must use the model's scaled inv_freq (not plain
    theta) -- its logits follow the HF forward -- and must refuse what it does not implement."""
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer, load_quantized_model
    from quip_for_all_amd.decode import LlamaDecoder
    cfg = _tiny_config()
    scaling = {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
               "original_max_position_embeddings": 16, "rope_theta": 10000.0}
    if hasattr(cfg, "rope_parameters"):
        cfg.rope_parameters = scaling
    else:
        cfg.rope_scaling = scaling
    torch.manual_seed(2)
    model = AutoModelForCausalLM.from_config(cfg, dtype=torch.float16)
    plain = 1.0 / (10000.0 ** (torch.arange(0, 64, 2, dtype=torch.float32) / 64))
    assert not torch.allclose(model.model.rotary_emb.inv_freq.float().cpu(), plain), "config did not scale the rope"
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    _fill_random(model, seed=11)
    qz.save(model, str(tmp_path))
    q = load_quantized_model(str(tmp_path), device_map={"": "cuda:0"})
    dec = LlamaDecoder.from_hf(q, max_len=48)
    assert torch.allclose(dec.cos[:, :32].cpu(), torch.cos(torch.arange(48.)[:, None] * q.model.rotary_emb.inv_freq.float().cpu()[None]), atol=1e-6)
    prompt = torch.tensor([5, 17, 3, 99, 42, 7, 1, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 43], device="cuda:0")
    toks = dec.generate(10, prompt=prompt)
    seq = torch.cat([prompt, toks])[None]
    with torch.no_grad():
        logits = q(seq).logits.float()[0]
    for t in range(10):
        row = logits[prompt.numel() - 1 + t]
        margin = (row.max() - row[int(toks[t])]).item()
        assert margin <= 0.03 * (row.abs().max().item() + 1.0), (t, margin)
    # unsupported options raise instead of decoding something else
    q.config.sliding_window, q.config.layer_types = 16, ["sliding_attention", "chunked_attention"]
    q.config.model_type = "mistral"
    with pytest.raises(NotImplementedError):
        LlamaDecoder.from_hf(q, max_len=48)
    q.config.model_type = "somethingelse"      # an architecture nobody vouches for
    with pytest.raises(NotImplementedError):
        LlamaDecoder.from_hf(q, max_len=48)
    # ... a window that is never shorter than the context is full attention (Mistral-7B's 4096): same tokens
    q.config.model_type = "mistral"
    q.config.sliding_window, q.config.layer_types = 48, ["sliding_attention"] * q.config.num_hidden_layers
    dec2 = LlamaDecoder.from_hf(q, max_len=48)
    assert dec2.window == 0 and torch.equal(dec2.generate(10, prompt=prompt), toks)


@pytest.mark.parametrize("window,plen", [(6, 20), (5, 3), (16, 40)])
def test_fast_decoder_sliding_window_follows_hf_mistral(tmp_path, window, plen):
    """A Mistral-architecture checkpoint whose sliding window is shorter than the context: LlamaDecoder.from_hf bounds the
    attention walk (rope_attn_decode's window, the band mask of the batched prompt pass) and follows the stock HF forward
    of the same model -- which masks keys at distance >= window -- token by token (teacher forced, fp16 noise margin)"""
    from transformers import AutoModelForCausalLM, MistralConfig
    from quip_for_all_amd.quantizer import QuipQuantizer, load_quantized_model
    from quip_for_all_amd.decode import LlamaDecoder
    cfg = MistralConfig(hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4,
                        num_key_value_heads=2, vocab_size=320, max_position_embeddings=64, tie_word_embeddings=False,
                        sliding_window=window)
    torch.manual_seed(3)
    model = AutoModelForCausalLM.from_config(cfg, dtype=torch.float16)
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    _fill_random(model, seed=13)
    qz.save(model, str(tmp_path))
    q = load_quantized_model(str(tmp_path), device_map={"": "cuda:0"})
    dec = LlamaDecoder.from_hf(q, max_len=64)
    assert dec.window == window and not dec.block_eng
    prompt = torch.randint(0, 320, (plen,), generator=torch.Generator().manual_seed(plen)).cuda()
    _check_prompt(q, dec, prompt)
    toks = dec.generate(8, prompt=prompt)
    assert torch.equal(toks, dec.generate(8, prompt=prompt, use_graph=False))


def test_fast_decoder_qwen2_architecture_with_qkv_bias(tmp_path):
    """A Qwen2-architecture checkpoint (Llama-shaped blocks, bias on q / k / v_proj, `sliding_window` present but unused):
    LlamaDecoder.from_hf follows the stock HF forward of the same model, prompt pass and captured steps"""
    from transformers import AutoModelForCausalLM, Qwen2Config
    from quip_for_all_amd.quantizer import QuipQuantizer, load_quantized_model
    from quip_for_all_amd.decode import LlamaDecoder
    cfg = Qwen2Config(hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=320, max_position_embeddings=64, tie_word_embeddings=False)
    torch.manual_seed(4)
    model = AutoModelForCausalLM.from_config(cfg, dtype=torch.float16)
    assert model.model.layers[0].self_attn.q_proj.bias is not None
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)          # (module replacement only: the biases of a checkpoint arrive with its state dict)
    _fill_random(model, seed=17)
    with torch.no_grad():
        for blk in model.model.layers:
            for m in (blk.self_attn.q_proj, blk.self_attn.k_proj, blk.self_attn.v_proj):
                assert m.bias is not None, "QuantLinear keeps the bias of the layer it replaces (quantizer.py:147-160)"
                m.bias.copy_(0.5 * torch.randn(m.bias.shape).to(m.bias.dtype))
    qz.save(model, str(tmp_path))
    q = load_quantized_model(str(tmp_path), device_map={"": "cuda:0"})
    assert q.model.layers[0].self_attn.q_proj.bias is not None and float(q.model.layers[0].self_attn.q_proj.bias.abs().max()) > 0
    dec = LlamaDecoder.from_hf(q, max_len=64)
    assert dec.window == 0
    for prompt in (torch.tensor([5, 17, 3, 99, 42], device="cuda:0"),
                   torch.randint(0, 320, (40,), generator=torch.Generator().manual_seed(6)).cuda()):
        _check_prompt(q, dec, prompt)
    toks = dec.generate(8, prompt=prompt)
    assert torch.equal(toks, dec.generate(8, prompt=prompt, use_graph=False))


def test_fast_decoder_refuses_blocks_that_are_not_llamas(tmp_path):
    """module names alone do not make a Llama block: a Gemma-architecture model (scaled embeddings, (1 + w) norms, GELU gate)
    has the same q / k / v / o / gate / up / down modules and must be refused, not decoded with Llama's arithmetic; a
    Llama config with biases everywhere (attention_bias, mlp_bias) is served and follows the HF forward"""
    from transformers import AutoModelForCausalLM, GemmaConfig, LlamaConfig
    from quip_for_all_amd.quantizer import QuipQuantizer
    from quip_for_all_amd.decode import LlamaDecoder
    from quip_for_all_amd.qlinear import QuantLinear
    cfg = GemmaConfig(hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=64, vocab_size=320, max_position_embeddings=64)
    model = AutoModelForCausalLM.from_config(cfg, dtype=torch.float16)
    QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0).convert_model(model)
    _fill_random(model, seed=1)
    model = model.to("cuda:0")
    with pytest.raises(NotImplementedError):
        LlamaDecoder.from_hf(model, max_len=32)
    with pytest.raises(NotImplementedError):     # vouched for, but the norm check still sees (1 + w) / the GELU gate
        LlamaDecoder.from_hf(model, max_len=32, assume_llama_like=True)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=688, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=320, max_position_embeddings=64, tie_word_embeddings=False, attention_bias=True, mlp_bias=True)
    torch.manual_seed(8)
    model = AutoModelForCausalLM.from_config(cfg, dtype=torch.float16)
    QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0).convert_model(model)
    _fill_random(model, seed=2)
    nb = 0
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, QuantLinear):
                assert m.bias is not None
                m.bias.copy_(0.3 * torch.randn(m.bias.shape).to(m.bias.dtype))
                nb += 1
    assert nb == 14
    model = model.to("cuda:0").eval()
    dec = LlamaDecoder.from_hf(model, max_len=64)
    for prompt in (torch.tensor([5, 17, 3, 99, 42], device="cuda:0"),
                   torch.randint(0, 320, (40,), generator=torch.Generator().manual_seed(7)).cuda()):
        _check_prompt(model, dec, prompt)


def test_hf_static_cache_step_captured_in_a_graph_equals_eager():
    """the reference's harness shape (example_generate.py:28-33, 62-70): the stock HF forward on a StaticCache, one token
    per call -- eager, and with the single-token step captured in a hipGraph (what mode="reduce-overhead" buys the
    reference); both decode the same greedy tokens, which are the stock `generate` loop's"""
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer
    from quip_for_all_amd.hf_static import HFStaticDecoder
    torch.manual_seed(0)
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    _fill_random(model, seed=3)
    model = model.to("cuda:0").eval()
    model.generation_config.eos_token_id = None
    ids = torch.tensor([[1, 17, 42, 99, 7, 250]], device="cuda:0")
    eager, _ = HFStaticDecoder(model, max_cache_len=64).generate(ids, 16, "eager")
    dec = HFStaticDecoder(model, max_cache_len=64)
    graph, _ = dec.generate(ids, 16, "graph")
    assert dec.graph is not None
    assert torch.equal(eager, graph), (eager, graph)
    again, _ = dec.generate(ids, 16, "graph")           # the captured step replays from a reset cache
    assert torch.equal(again, graph)
    ref = model.generate(ids, max_new_tokens=16, do_sample=False)[0, ids.shape[1]:]
    n = min(len(ref), 16)
    assert n >= 8 and torch.equal(ref[:n], graph[:n]), (ref, graph)
    # the reference's own call on the step: torch.compile(mode="reduce-overhead", fullgraph=True) (example_generate.py:69-70)
    comp = HFStaticDecoder(model, max_cache_len=64)
    comp.compile(fullgraph=True)
    ct, _ = comp.generate(ids, 16, "compile")
    ct, _ = comp.generate(ids, 16, "compile")
    assert torch.equal(ct, eager), (ct, eager)
    # ... and HF's generate with a static cache, which compiles the forward the same way, on the stock modules
    st = model.generate(ids, max_new_tokens=16, do_sample=False, cache_implementation="static")[0, ids.shape[1]:]
    assert torch.equal(st[:n], ref[:n]), (st, ref)


def test_fast_decode_wrapper_routes_static_cache_steps_through_the_decoder():
    """hf_fast.enable_fast_decode: single-token calls on an initialised StaticCache run LlamaDecoder.step() on the cache
    object's own tensors; the prompt and everything else stays on the stock forward.  Same greedy tokens as the stock
    harness (a near tie may flip one: logits are compared too), cache rows and lengths as StaticLayer.update leaves them,
    and `model.generate(cache_implementation="static")` goes through the wrapper as well."""
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer
    from quip_for_all_amd.hf_static import HFStaticDecoder
    from quip_for_all_amd.hf_fast import enable_fast_decode, disable_fast_decode
    torch.manual_seed(0)
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    _fill_random(model, seed=3)
    model = model.to("cuda:0").eval()
    model.generation_config.eos_token_id = None
    ids = torch.tensor([[1, 17, 42, 99, 7, 250]], device="cuda:0")
    stock = HFStaticDecoder(model, max_cache_len=64)
    want, _ = stock.generate(ids, 16, "eager")
    enable_fast_decode(model)
    fd = model._quip_fast_decode
    fast = HFStaticDecoder(model, max_cache_len=64)
    got, _ = fast.generate(ids, 16, "eager")
    assert fd.disabled is None and fd.fast_steps == 15, (fd.disabled, fd.fast_steps)
    assert int((got == want).sum()) >= 14, (got, want)
    # cache bookkeeping and rows: as the stock path leaves them
    n = ids.shape[1] + 15
    for Ls, Lf in zip(stock.cache.layers, fast.cache.layers):
        assert int(Lf.cumulative_length) == n == int(Ls.cumulative_length)
    if torch.equal(got, want):
        for Ls, Lf in zip(stock.cache.layers, fast.cache.layers):
            dk = (Lf.keys[:, :, :n].float() - Ls.keys[:, :, :n].float()).abs().max().item()
            dv = (Lf.values[:, :, :n].float() - Ls.values[:, :, :n].float()).abs().max().item()
            assert dk <= 2.0 ** -5 * Ls.keys.float().abs().max().item() and dv <= 2.0 ** -5 * Ls.values.float().abs().max().item(), (dk, dv)
    # one step's logits against the stock forward on the same cache state
    a, b = HFStaticDecoder(model, max_cache_len=64), HFStaticDecoder(model, max_cache_len=64)
    a.prefill(ids)
    disable_fast_decode(model)
    b.prefill(ids)
    lb = b._forward(b.tok, b.pos).float()
    enable_fast_decode(model)
    la = a._forward(a.tok, a.pos).float()
    assert (la - lb).abs().max().item() <= 2.0 ** -6 * lb.abs().max().item(), (la - lb).abs().max().item()
    # the captured step (the reference's reduce-overhead counterpart) on the wrapper, and a second cache object (re-bound)
    g = HFStaticDecoder(model, max_cache_len=64)
    gt, _ = g.generate(ids, 16, "graph")
    assert int((gt == want).sum()) >= 14, (gt, want)
    # the reference's own call, torch.compile(step, mode="reduce-overhead", fullgraph=True): the wrapper is ONE operator in
    # the traced graph (quip_lib::hf_decode_step), so fullgraph holds, and the decoder it builds during the compiled
    # function's warm-up run does not live in the cudagraph trees' pool
    c = HFStaticDecoder(model, max_cache_len=64)
    c.compile(fullgraph=True)
    steps_c = model._quip_fast_decode.fast_steps
    ct, _ = c.generate(ids, 16, "compile")
    ct, _ = c.generate(ids, 16, "compile")
    assert model._quip_fast_decode.fast_steps > steps_c and int((ct == want).sum()) >= 14, (ct, want)
    # HF's own generate with a static cache compiles the forward and passes an attention mask, whose values a trace cannot
    # look at: the (eager) prompt pass has, and left its finding on the cache object -- unpadded here, so the operator runs
    steps0 = model._quip_fast_decode.fast_steps
    ref = model.generate(ids, max_new_tokens=12, do_sample=False, cache_implementation="static")[0, ids.shape[1]:]
    assert model._quip_fast_decode.fast_steps > steps0
    assert int((ref[:12] == want[:12]).sum()) >= 10, (ref, want)
    disable_fast_decode(model)
    assert model.forward.__self__ is model


def test_fast_decode_wrapper_serves_the_default_dynamic_cache():
    """model.generate() with its default DynamicCache: after the stock prompt pass the wrapper imports the rows into its own
    static buffers, decodes on them and hands the cache object views one row longer per step; tokens as the stock loop's,
    the cache object keeps answering get_seq_length(), can go back to the stock forward, and two cache objects never alias"""
    from transformers import AutoModelForCausalLM, DynamicCache
    from quip_for_all_amd.quantizer import QuipQuantizer
    from quip_for_all_amd.hf_fast import enable_fast_decode, disable_fast_decode
    torch.manual_seed(0)
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
    qz.convert_model(model)
    _fill_random(model, seed=3)
    model = model.to("cuda:0").eval()
    model.generation_config.eos_token_id = None
    model.generation_config.pad_token_id = 0
    ids = torch.tensor([[1, 17, 42, 99, 7, 250]], device="cuda:0")
    want = model.generate(ids, max_new_tokens=16, do_sample=False)[0, ids.shape[1]:]
    enable_fast_decode(model)
    fd = model._quip_fast_decode
    got = model.generate(ids, max_new_tokens=16, do_sample=False)[0, ids.shape[1]:]
    assert fd.disabled is None and fd.fast_steps == 15 and fd.fast_prefills == 1, (fd.disabled, fd.fast_steps, fd.fast_prefills)
    assert int((got == want).sum()) >= 14, (got, want)
    # the prompt pass alone: last-token logits of the decoder's batched prefill against the stock forward's
    from transformers import DynamicCache as _DC
    with torch.no_grad():
        lf = model(ids, past_key_values=_DC(config=model.config), use_cache=True, logits_to_keep=1).logits.float()
        model.forward = fd.orig_forward
        ls = model(ids, past_key_values=_DC(config=model.config), use_cache=True, logits_to_keep=1).logits.float()
        model.forward = fd
    assert fd.fast_prefills == 2 and lf.shape == ls.shape == (1, 1, model.config.vocab_size)
    assert (lf - ls).abs().max().item() <= 2.0 ** -6 * ls.abs().max().item(), (lf - ls).abs().max().item()

    # by hand: two cache objects interleaved, one of them sent back to the stock forward in between
    def prompt(cache, p):
        with torch.no_grad():
            return model(p, past_key_values=cache, use_cache=True).logits[:, -1].argmax(-1, keepdim=True)

    def step(cache, tok):
        with torch.no_grad():
            return model(tok, past_key_values=cache, use_cache=True).logits[:, -1].argmax(-1, keepdim=True)
    ids2 = torch.tensor([[3, 9, 200, 41]], device="cuda:0")
    ca, cb = DynamicCache(config=model.config), DynamicCache(config=model.config)
    ta, tb = prompt(ca, ids), prompt(cb, ids2)
    seq_a, seq_b = [int(ta)], [int(tb)]
    for i in range(6):
        ta = step(ca, ta); seq_a.append(int(ta))          # noqa: E702
        tb = step(cb, tb); seq_b.append(int(tb))          # noqa: E702
        assert ca.get_seq_length() == ids.shape[1] + i + 1 and cb.get_seq_length() == ids2.shape[1] + i + 1
    steps_before = fd.fast_steps
    disable_fast_decode(model)
    ta = step(ca, ta); seq_a.append(int(ta))              # noqa: E702  (stock forward on a cache whose layers were views)
    enable_fast_decode(model)
    fd = model._quip_fast_decode
    ta = step(ca, ta); seq_a.append(int(ta))              # noqa: E702
    assert fd.fast_steps == 1 and steps_before == 15 + 12
    disable_fast_decode(model)
    ra = model.generate(ids, max_new_tokens=9, do_sample=False)[0, ids.shape[1]:].tolist()
    rb = model.generate(ids2, max_new_tokens=7, do_sample=False)[0, ids2.shape[1]:].tolist()
    assert sum(x == y for x, y in zip(seq_a, ra)) >= 8 and sum(x == y for x, y in zip(seq_b, rb)) >= 6, (seq_a, ra, seq_b, rb)


def _random_quantized_hf_llama(cfg, device="cuda:0", seed=0):
    """an HF Llama of any size without its dense weights ever existing: parameters on the meta device -> convert_model ->
    to_empty on the GPU -> random codes / signs / scales / norms, rotary buffers rebuilt (what bench.py does for the 7B shape)"""
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer
    from quip_for_all_amd.qlinear import QuantLinear
    with torch.device("meta"):
        model = AutoModelForCausalLM.from_config(cfg, dtype=torch.float16)
    QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0).convert_model(model)
    # (to_empty replaces EVERY tensor by uninitialised memory, also the real ones convert_model made -- the codebooks' tables)
    real = {n: t.detach().clone() for n, t in list(model.named_parameters()) + list(model.named_buffers()) if not t.is_meta}
    model.to_empty(device=device)
    with torch.no_grad():
        live = dict(list(model.named_parameters()) + list(model.named_buffers()))
        for n, t in real.items():
            live[n].copy_(t)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, prm in list(model.named_parameters()) + list(model.named_buffers()):
            if prm.is_floating_point() and "inv_freq" not in name:
                prm.copy_((torch.randn(prm.shape, generator=g, device=device, dtype=torch.float32) * 0.02).to(prm.dtype))
        for m in model.modules():
            if isinstance(m, QuantLinear):
                m.Qidxs.copy_(torch.randint(-32768, 32768, m.Qidxs.shape, generator=g, device=device, dtype=torch.int32).to(m.Qidxs.dtype))
                m.SU.copy_((torch.randint(0, 2, m.SU.shape, generator=g, device=device) * 2 - 1).half())
                m.SV.copy_((torch.randint(0, 2, m.SV.shape, generator=g, device=device) * 2 - 1).half())
                m.Wscale.fill_(1.0 / 64.0)
                for nm in ("had_left", "had_right"):
                    h = getattr(m, nm)
                    if h is not None:
                        h.copy_(torch.linalg.qr(torch.randn(h.shape, generator=g, device=device))[0].half())
            if m.__class__.__name__.endswith("RMSNorm"):
                m.weight.fill_(1.0)
        rot = model.model.rotary_emb
        model.model.rotary_emb = type(rot)(config=cfg, device=device)
        for m in model.modules():
            if isinstance(m, QuantLinear):
                m.wscale_float = float(m.Wscale.mean().item())
    return model.eval()


def test_fast_decode_wrapper_on_a_70b_shaped_hf_model_runs_the_grouped_query_launch():
    """two blocks of the Llama-2-70B shape (hidden 8192, 64 heads on 8 KV heads, n_ffn 28672) as an HF model: the wrapper's
    decoder takes the grouped-query persistent launch (shape 1) on the StaticCache's (1, 8, len, 128) tensors; logits of a
    step within 2^-6 of the stock forward's maximum, greedy tokens equal up to a near tie"""
    from transformers import LlamaConfig
    from quip_for_all_amd.hf_static import HFStaticDecoder
    from quip_for_all_amd.hf_fast import enable_fast_decode, disable_fast_decode
    cfg = LlamaConfig(hidden_size=8192, intermediate_size=28672, num_hidden_layers=2, num_attention_heads=64,
                      num_key_value_heads=8, vocab_size=2048, max_position_embeddings=64, rms_norm_eps=1e-5,
                      tie_word_embeddings=False)
    model = _random_quantized_hf_llama(cfg)
    ids = torch.tensor([[1, 17, 42, 99, 7, 250]], device="cuda:0")
    enable_fast_decode(model)
    fd = model._quip_fast_decode
    a, b = HFStaticDecoder(model, max_cache_len=64), HFStaticDecoder(model, max_cache_len=64)
    a.prefill(ids)                                      # (the prompt pass is the stock forward either way)
    b.prefill(ids)
    same = 0
    for t in range(8):                                  # teacher-forced on the stock path's tokens: a near tie cannot fork the two
        model.forward = fd.orig_forward
        lb = b._forward(b.tok, b.pos).float()
        model.forward = fd
        la = a._forward(a.tok, a.pos).float()
        assert (la - lb).abs().max().item() <= 2.0 ** -6 * lb.abs().max().item(), (t, (la - lb).abs().max().item())
        nxt = lb[:, -1].argmax(-1, keepdim=True)
        same += int(la[:, -1].argmax(-1).item() == nxt.item())
        for h in (a, b):
            h.tok.copy_(nxt)
            h.pos += 1
    assert fd.disabled is None and fd.fast_steps == 8 and fd.dec.block_eng and fd.dec.eng_shape == 1 and fd.dec.engine_status() == 0
    assert same >= 7, same
    n = ids.shape[1] + 8
    for La, Lb in zip(a.cache.layers, b.cache.layers):  # the grouped-query cache rows (1, 8, len, 128) the launch wrote
        assert int(La.cumulative_length) == n == int(Lb.cumulative_length)
        dk = (La.keys[:, :, :n].float() - Lb.keys[:, :, :n].float()).abs().max().item()
        assert dk <= 2.0 ** -5 * Lb.keys.float().abs().max().item(), dk
    disable_fast_decode(model)


def test_fast_decode_wrapper_on_a_llama3_8b_shaped_hf_model_runs_the_shape_2_launch():
    """two blocks of the Llama-3-8B / Mistral-7B shape (hidden 4096, 32 heads on 8 KV heads, n_ffn 14336) as an HF model: the
    wrapper's decoder takes the persistent launch compiled for that shape (shape 2, decode_block_g8.hip) on the StaticCache's
    (1, 8, len, 128) tensors; logits of a step within 2^-6 of the stock forward's maximum, greedy tokens equal up to a near tie"""
    from transformers import LlamaConfig
    from quip_for_all_amd.hf_static import HFStaticDecoder
    from quip_for_all_amd.hf_fast import enable_fast_decode, disable_fast_decode
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=2048, max_position_embeddings=64, rms_norm_eps=1e-5,
                      tie_word_embeddings=False)
    model = _random_quantized_hf_llama(cfg)
    ids = torch.tensor([[1, 17, 42, 99, 7, 250]], device="cuda:0")
    enable_fast_decode(model)
    fd = model._quip_fast_decode
    a, b = HFStaticDecoder(model, max_cache_len=64), HFStaticDecoder(model, max_cache_len=64)
    a.prefill(ids)
    b.prefill(ids)
    same = 0
    for t in range(8):                                  # teacher-forced on the stock path's tokens
        model.forward = fd.orig_forward
        lb = b._forward(b.tok, b.pos).float()
        model.forward = fd
        la = a._forward(a.tok, a.pos).float()
        assert (la - lb).abs().max().item() <= 2.0 ** -6 * lb.abs().max().item(), (t, (la - lb).abs().max().item())
        nxt = lb[:, -1].argmax(-1, keepdim=True)
        same += int(la[:, -1].argmax(-1).item() == nxt.item())
        for h in (a, b):
            h.tok.copy_(nxt)
            h.pos += 1
    assert fd.disabled is None and fd.fast_steps == 8 and fd.dec.block_eng and fd.dec.eng_shape == 2 and fd.dec.engine_status() == 0
    assert same >= 7, same
    n = ids.shape[1] + 8
    for La, Lb in zip(a.cache.layers, b.cache.layers):
        assert int(La.cumulative_length) == n == int(Lb.cumulative_length)
        dk = (La.keys[:, :, :n].float() - Lb.keys[:, :, :n].float()).abs().max().item()
        assert dk <= 2.0 ** -5 * Lb.keys.float().abs().max().item(), dk
    disable_fast_decode(model)


def test_fast_decode_wrapper_leaves_a_padded_sequence_to_the_stock_forward():
    """a left-padded single sequence (attention_mask with a hole, position_ids behind the cache length): the wrapper checks
    once when it takes a cache object over and stays out; the tokens are the stock forward's"""
    from transformers import AutoModelForCausalLM
    from quip_for_all_amd.quantizer import QuipQuantizer
    from quip_for_all_amd.hf_fast import enable_fast_decode, disable_fast_decode
    torch.manual_seed(0)
    model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
    QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0).convert_model(model)
    _fill_random(model, seed=3)
    model = model.to("cuda:0").eval()
    model.generation_config.eos_token_id = None
    model.generation_config.pad_token_id = 0
    ids = torch.tensor([[0, 0, 17, 42, 99, 7]], device="cuda:0")
    mask = torch.tensor([[0, 0, 1, 1, 1, 1]], device="cuda:0")
    want = model.generate(ids, attention_mask=mask, max_new_tokens=8, do_sample=False)[0, ids.shape[1]:]
    enable_fast_decode(model)
    fd = model._quip_fast_decode
    got = model.generate(ids, attention_mask=mask, max_new_tokens=8, do_sample=False)[0, ids.shape[1]:]
    assert fd.fast_steps == 0 and torch.equal(got, want), (fd.fast_steps, got, want)
    got2 = model.generate(ids, attention_mask=mask, max_new_tokens=8, do_sample=False, cache_implementation="static")[0, ids.shape[1]:]
    assert fd.fast_steps == 0 and int((got2 == want).sum()) >= 7, (fd.fast_steps, got2, want)
    # the same (reused) static cache object, now with an unpadded prompt: its own graph, through the operator -- and padded again
    u_want = None
    for rounds in range(2):
        steps0 = fd.fast_steps
        u = model.generate(ids[:, 2:], max_new_tokens=8, do_sample=False, cache_implementation="static")[0, 4:]
        u_want = u if u_want is None else u_want
        assert fd.fast_steps > steps0 or rounds == 1, (rounds, fd.fast_steps, steps0)     # (a replayed graph runs no Python)
        assert torch.equal(u, u_want)
        steps0 = fd.fast_steps
        got3 = model.generate(ids, attention_mask=mask, max_new_tokens=8, do_sample=False, cache_implementation="static")[0, ids.shape[1]:]
        assert fd.fast_steps == steps0 and torch.equal(got3, got2), (got3, got2)
    fd.fast_steps = 0
    # ... and an unpadded one still goes through
    model.generate(ids[:, 2:], max_new_tokens=4, do_sample=False)
    assert fd.fast_steps == 3
    disable_fast_decode(model)
