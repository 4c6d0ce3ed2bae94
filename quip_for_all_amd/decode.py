"""bs=1 greedy decode harness for Llama-architecture models whose linear layers are
QuantLinear (the metric driver of the reference, example_generate.py:9-59,62-110:
static KV cache + one captured single-token step, sampling on the device).

The reference relies on HF `StaticCache` + `torch.compile(mode="reduce-overhead")`;
`_setup_cache` is a transformers-4.38 API that no longer exists, and there is no tracing
compiler in this stack by design, so the step is written once with static shapes and
captured in a hipGraph (torch.cuda.CUDAGraph): one graph replay per token, no host sync
inside the loop (the next token id stays on the device).

Only the linear layers are the QuIP# hot path; embeddings, norms, RoPE, attention over the
cache and the fp16 lm_head use stock torch ops (they are what remains once the GEMVs are
fast: SURVEY.md 8f rank 1)."""
import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from .codebook import codebook_id
from .qlinear import (QuantLinear, chain_planes, chain_supported, ffn_engine, ffn_engine_ok, forward_group,
                      fused_in_supported, gemv_chain, gemv_fused, gemv_group_unfused, gemv_unfused, out_transform_group)


@dataclass
class LlamaShape:
    hidden: int = 4096
    ffn: int = 11008
    layers: int = 32
    heads: int = 32
    kv_heads: int = 32
    vocab: int = 32000
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0

    @property
    def head_dim(self):
        return self.hidden // self.heads


LLAMA2_7B = LlamaShape()
LLAMA2_70B = LlamaShape(hidden=8192, ffn=28672, layers=80, heads=64, kv_heads=8)
LLAMA3_8B = LlamaShape(hidden=4096, ffn=14336, layers=32, heads=32, kv_heads=8, vocab=128256)   # also Mistral-7B's block (vocab 32000)
SMALL = LlamaShape(hidden=1024, ffn=2816, layers=2, heads=8, kv_heads=4, vocab=1024)   # takes the fused-prologue path
TINY = LlamaShape(hidden=256, ffn=688, layers=2, heads=4, kv_heads=2, vocab=512)


def random_quant_linear(in_f, out_f, codebook="E8P12", generator=None, device="cuda", **cb_kwargs):
    """QuantLinear with uniformly random codes (every code is a valid lattice point), +-1 SU/SV,
    random orthogonal had factors and a scale that keeps |y| ~ |x| (random-init analogue of a
    quantised checkpoint)."""
    cb = codebook_id[codebook](inference=True, **cb_kwargs)
    layer = QuantLinear(in_f, out_f, cb, bias=False, use_rand=True)
    g = generator
    dev = torch.device(device)
    on_dev = g is not None and g.device.type == dev.type == "cuda"   # large models: draw the codes on the GPU
    gdev = dev if on_dev else "cpu"
    with torch.no_grad():
        q = layer.Qidxs
        if q.dtype == torch.int16:
            codes = torch.randint(-32768, 32768, q.shape, generator=g, dtype=torch.int32, device=gdev).to(torch.int16)
        elif q.dtype == torch.uint8:
            codes = torch.randint(0, 256, q.shape, generator=g, dtype=torch.int32, device=gdev).to(torch.uint8)
        else:
            codes = torch.randint(-2 ** 31, 2 ** 31 - 1, q.shape, generator=g, dtype=torch.int64,
                                  device=gdev).to(torch.int32)
        su = (torch.randint(0, 2, (in_f,), generator=g, device=gdev) * 2 - 1).to(torch.float16)
        sv = (torch.randint(0, 2, (out_f,), generator=g, device=gdev) * 2 - 1).to(torch.float16)
        if on_dev:
            layer = layer.to(dev)
        layer.Qidxs.copy_(codes)
        layer.SU.copy_(su)
        layer.SV.copy_(sv)
        wrms = {"E8P12": 1.03, "E8P12RVQ3B": 1.2, "E8P12RVQ4B": 1.2, "D4": 1.21, "HI": 4.6}[codebook]
        layer.Wscale.fill_(1.0 / (wrms * math.sqrt(in_f)))
    layer.wscale_float = float(layer.Wscale)      # quantizer.py:836-837
    return layer.to(device).eval()


def tile_codes(qidxs):
    """The persistent 8192-wide launch's layout of a code matrix (include/quip_mi355.h: quip_tile_codes; decode_block_gqa.hip):
    inside every aligned block of 16 rows the 64-byte pieces of the 16 rows lie side by side, piece after piece --
    tiled[rb][c][q][n] = bytes [64 c + 16 q, +16) of row 16 rb + n -- so that one load instruction of the product (16 rows x
    64 bytes in the checkpoint's layout; origin_order.cu:388-555 walks it row-major) covers 1 KB of consecutive bytes.
    Returns a flat uint8 tensor of the same size on the same device."""
    from . import capi
    b = qidxs.detach().contiguous().view(torch.uint8).reshape(qidxs.shape[0], -1)
    rows, rb = b.shape
    out = torch.empty(rows * rb, dtype=torch.uint8, device=b.device)
    if not b.is_cuda:
        raise capi.QuipNativeError("tile_codes: the codes must be on the GPU (there is no CPU path)")
    with torch.cuda.device(b.device):
        capi.check(capi.lib().quip_tile_codes(b.data_ptr(), out.data_ptr(), rows, rb, torch.cuda.current_stream(b.device).cuda_stream),
                   "quip_tile_codes")
    return out


def untile_codes(tiled, rows, row_bytes, out):
    """the inverse of tile_codes (quip_untile_codes): `tiled` (flat uint8) -> the row-major matrix, written into `out` (any dtype,
    rows * row_bytes bytes)"""
    from . import capi
    with torch.cuda.device(tiled.device):
        capi.check(capi.lib().quip_untile_codes(tiled.data_ptr(), out.data_ptr(), rows, row_bytes,
                                                torch.cuda.current_stream(tiled.device).cuda_stream), "quip_untile_codes")
    return out


def qidxs_nbytes(m):
    """bytes of a module's code matrix (its row-major tensor, or -- LlamaDecoder(single_copy=True) -- the tiled copy that replaced it)"""
    return m.Qidxs.numel() * m.Qidxs.element_size() if m.Qidxs is not None else m._qidxs_tiled.numel()


class LlamaDecoder:
    """Random-init Llama with QuantLinear projections, static KV cache, bs=1.

    single_copy (or QUIP_SINGLE_COPY=1; the 8192-wide persistent launch only -- it streams a RE-TILED copy of the codes,
    tile_codes()): keep ONE copy of the code matrices.  The modules' row-major `Qidxs` are dropped (set to None: every operator that
    would read them fails loudly, and `state_dict()` no longer holds them -- a decode-only serving mode, not one to save checkpoints
    from) once the tiled copies exist; the prompt pass and the stage-wise fallback get a matrix back in the checkpoint's layout,
    one decoder block at a time, in scratch buffers (untile_codes: 2 x 214 MB of traffic per block at HBM speed).  Llama-2-70B
    E8P12: 17.1 GB of codes resident instead of 34.2 GB."""

    def __init__(self, shape: LlamaShape = LLAMA2_7B, codebook="E8P12", max_len=256, device="cuda", seed=0,
                 device_init=False, window=0, single_copy=None, **cb_kwargs):
        import os
        self.single_copy = bool(int(os.environ.get("QUIP_SINGLE_COPY", "0"))) if single_copy is None else bool(single_copy)
        self.s, self.dev, self.max_len = shape, torch.device(device), max_len
        self.window = int(window) if 0 < int(window) < max_len else 0     # sliding-window attention (HF config.sliding_window)
        g = torch.Generator().manual_seed(seed)
        gq = torch.Generator(device=self.dev).manual_seed(seed) if device_init else g
        s = shape
        kv = s.kv_heads * s.head_dim

        def ql(i, o):
            return random_quant_linear(i, o, codebook, gq, device, **cb_kwargs)

        def vec(n):
            return (1.0 + 0.02 * torch.randn(n, generator=g)).to(torch.float16).to(self.dev)

        self.embed = (0.5 * torch.randn(s.vocab, s.hidden, generator=g)).to(torch.float16).to(self.dev)
        self.lm_head = (torch.randn(s.vocab, s.hidden, generator=g) / math.sqrt(s.hidden)).to(torch.float16).to(self.dev)
        self.final_norm = vec(s.hidden)
        self.layers = []
        for _ in range(s.layers):
            self.layers.append(dict(
                ln1=vec(s.hidden), ln2=vec(s.hidden),
                q=ql(s.hidden, s.hidden), k=ql(s.hidden, kv), v=ql(s.hidden, kv), o=ql(s.hidden, s.hidden),
                gate=ql(s.hidden, s.ffn), up=ql(s.hidden, s.ffn), down=ql(s.ffn, s.hidden)))
        self._init_runtime()

    # architectures whose decoder block is Llama's: RMSNorm (weight only) -> q / k / v (+ bias) -> rotary -> softmax attention
    # -> o -> residual -> RMSNorm -> SiLU-gated MLP -> residual, embeddings and residuals unscaled
    LLAMA_LIKE = ("llama", "mistral", "qwen2")

    @classmethod
    def from_hf(cls, model, max_len=256, device=None, assume_llama_like=False, kv_cache=None):
        """Fast bs=1 decoder around a Llama-architecture HF model whose linear layers are QuantLinear
        (what `load_quantized_model` returns): shares the modules / weights, adds the static KV cache and
        the captured step.  Needs `model.model.{embed_tokens, layers, norm}` and `model.lm_head`.
        `kv_cache` = (keys, values): per-layer fp16 tensors (kv_heads, max_len, head_dim) to use as the static cache instead of
        allocating one (hf_fast.py binds the tensors of a transformers StaticCache this way; bind_kv() swaps them later).
        Architectures outside LLAMA_LIKE are refused (Gemma scales embeddings and norms, StableLM uses LayerNorm, ...:
        the module names match, the arithmetic does not) unless `assume_llama_like` vouches for them; such checkpoints run
        through `load_quantized_model` + the stock HF `generate`."""
        cfg = model.config
        dev = torch.device(device) if device is not None else next(model.parameters()).device
        self = cls.__new__(cls)
        mt = getattr(cfg, "model_type", "")
        if mt not in cls.LLAMA_LIKE and not assume_llama_like:
            raise NotImplementedError(f"model_type {mt!r}: only {cls.LLAMA_LIKE} are known to have Llama's decoder block "
                                      "(pass assume_llama_like=True if this one does)")
        if getattr(cfg, "hidden_act", "silu") != "silu":
            raise NotImplementedError(f"hidden_act {cfg.hidden_act!r}: the MLP here is SiLU-gated")
        # the normalisation must be y = x / sqrt(mean(x^2) + eps) * weight: checked on numbers, not on a class name
        n0 = model.model.layers[0].input_layernorm
        with torch.no_grad():
            xt = torch.linspace(-2.0, 3.0, cfg.hidden_size, dtype=torch.float32, device=n0.weight.device)[None]
            want = xt / torch.sqrt((xt * xt).mean() + cfg.rms_norm_eps) * n0.weight.float()
            got = n0(xt.to(n0.weight.dtype)).float()
        if getattr(n0, "bias", None) is not None or not torch.allclose(got, want, rtol=2e-2, atol=2e-2):
            raise NotImplementedError(f"{type(n0).__name__} is not RMSNorm(x) * weight with eps = rms_norm_eps")
        heads = cfg.num_attention_heads
        rope = getattr(cfg, "rope_theta", None)
        rp = getattr(cfg, "rope_parameters", None) or getattr(cfg, "rope_scaling", None) or {}
        if rope is None:
            rope = rp.get("rope_theta", 10000.0) if isinstance(rp, dict) else 10000.0
        # what this decoder does not implement must not pass silently (the HF forward of the same model would differ)
        head_dim = getattr(cfg, "head_dim", None) or cfg.hidden_size // heads
        if head_dim * heads != cfg.hidden_size:
            raise NotImplementedError(f"head_dim {head_dim} x {heads} heads != hidden_size {cfg.hidden_size}")
        prf = rp.get("partial_rotary_factor", None) if isinstance(rp, dict) else None
        prf = getattr(cfg, "partial_rotary_factor", prf)
        if prf not in (None, 1, 1.0):
            raise NotImplementedError(f"partial rotary embeddings (factor {prf}) are not supported")
        self.window = 0
        # (HF's Llama modelling code never reads `sliding_window`: on such a config the attribute changes nothing there,
        #  so it changes nothing here)
        if (getattr(cfg, "sliding_window", None) and getattr(cfg, "use_sliding_window", True)
                and getattr(cfg, "model_type", "") != "llama"):
            types = getattr(cfg, "layer_types", None)
            if not types:
                # older Qwen2-style configs have no layer_types: HF applies the window to layers >= max_window_layers only
                # (modeling_qwen2.py), so the per-layer pattern follows from that field
                mwl = getattr(cfg, "max_window_layers", None)
                nl = cfg.num_hidden_layers
                types = ["sliding_attention"] if mwl is None else \
                    ["full_attention" if i < int(mwl) else "sliding_attention" for i in range(nl)]
            if any(t not in ("full_attention", "sliding_attention") for t in types) or len(set(types)) > 1:
                raise NotImplementedError(f"layer_types {sorted(set(types))}: per-layer attention patterns are not supported")
            # a window that is never shorter than the context is full attention (Mistral-7B: 4096 on a 4096 cache); a
            # shorter one bounds the attention walk from below -- the cache stays linear (row t = position t)
            if types[0] == "sliding_attention" and max_len > int(cfg.sliding_window):
                self.window = int(cfg.sliding_window)
        # rotary frequencies and attention scaling as the model computes them (llama3 / linear / yarn scaling change
        # inv_freq at every position; taking only rope_theta from the config would silently give other logits)
        self._inv_freq, self._att_scale = None, 1.0
        rot = getattr(model.model, "rotary_emb", None)
        if rot is not None and hasattr(rot, "inv_freq"):
            rope_type = getattr(rot, "rope_type", None) or (rp.get("rope_type", "default") if isinstance(rp, dict) else "default")
            if rope_type in ("dynamic", "longrope"):
                raise NotImplementedError(f"rope_type {rope_type!r}: frequencies depend on the sequence length")
            inv = rot.inv_freq.detach().to(torch.float32).cpu()
            if inv.numel() != head_dim // 2:
                raise NotImplementedError(f"rotary_emb.inv_freq has {inv.numel()} entries for head_dim {head_dim}")
            self._inv_freq, self._att_scale = inv, float(getattr(rot, "attention_scaling", 1.0) or 1.0)
        self.s = LlamaShape(hidden=cfg.hidden_size, ffn=cfg.intermediate_size, layers=cfg.num_hidden_layers,
                            heads=heads, kv_heads=getattr(cfg, "num_key_value_heads", heads) or heads,
                            vocab=cfg.vocab_size, rms_eps=cfg.rms_norm_eps, rope_theta=float(rope))
        self.dev, self.max_len = dev, max_len
        self._external_kv = kv_cache
        h16 = lambda t: t.detach().to(device=dev, dtype=torch.float16).contiguous()  # noqa: E731
        self.embed = h16(model.model.embed_tokens.weight)
        self.lm_head = h16(model.lm_head.weight)
        self.final_norm = h16(model.model.norm.weight)
        self.layers = []
        for blk in model.model.layers:
            a, m = blk.self_attn, blk.mlp
            mods = dict(q=a.q_proj, k=a.k_proj, v=a.v_proj, o=a.o_proj, gate=m.gate_proj, up=m.up_proj, down=m.down_proj)
            for name, mod in mods.items():
                if not isinstance(mod, QuantLinear):
                    raise TypeError(f"{name}_proj is {type(mod).__name__}, expected QuantLinear")
                mod.to(dev).eval()
            self.layers.append(dict(ln1=h16(blk.input_layernorm.weight), ln2=h16(blk.post_attention_layernorm.weight),
                                    **mods))
        self._init_runtime()
        return self

    def _init_runtime(self):
        s, max_len = self.s, self.max_len
        ext = getattr(self, "_external_kv", None)
        if ext is not None:
            self.kcache, self.vcache = self._check_kv(*ext)
        else:
            self.kcache = torch.zeros(s.layers, s.kv_heads, max_len, s.head_dim, dtype=torch.float16, device=self.dev)
            self.vcache = torch.zeros_like(self.kcache)
        inv = getattr(self, "_inv_freq", None)
        if inv is None:
            inv = 1.0 / (s.rope_theta ** (torch.arange(0, s.head_dim, 2, dtype=torch.float32) / s.head_dim))
        att = float(getattr(self, "_att_scale", 1.0))
        ang = torch.arange(max_len, dtype=torch.float32)[:, None] * inv[None, :]
        self.cos = (torch.cat([ang.cos(), ang.cos()], -1) * att).to(self.dev)     # (max_len, head_dim) fp32
        self.sin = (torch.cat([ang.sin(), ang.sin()], -1) * att).to(self.dev)
        self.arange = torch.arange(max_len, device=self.dev)
        # static step I/O (graph capture): current token id, its position, next token id
        self.tok = torch.zeros(1, dtype=torch.long, device=self.dev)
        self.pos = torch.zeros(1, dtype=torch.long, device=self.dev)
        self.graph = None
        self.sampling = None
        self.fused_attention = s.head_dim in (64, 128)
        self.window = int(getattr(self, "window", 0) or 0)    # sliding-window attention: keys (pos - window, pos]; 0 = all
        from .register_lib import rope_attn_workspace
        self.attn_ws = rope_attn_workspace(s.heads, s.head_dim, self.dev) if self.fused_attention else None
        L0 = self.layers[0]
        import os
        qkv0, gu0 = [L0["q"], L0["k"], L0["v"]], [L0["gate"], L0["up"]]
        planes_ok = all(hasattr(m.codebook, "mm_planes") and m.codebook.planes_supported(m.q_out_features, m.q_in_features)
                        for m in L0.values() if isinstance(m, QuantLinear))
        # stage-wise step (10 launches per block): chain launches where the shapes allow, else the GEMV prologue
        self.chain = (planes_ok and os.environ.get("QUIP_CHAIN", "1") != "0"
                      and chain_supported(qkv0, L0["down"]) and chain_supported(gu0, L0["o"]))
        prologue_ok = (planes_ok and fused_in_supported(qkv0, prev=L0["down"]) and fused_in_supported(gu0, prev=L0["o"]))
        self.o_fused = planes_ok and fused_in_supported([L0["o"]])
        self.qkv_fused = planes_ok and fused_in_supported(qkv0)
        self.fused_prologue = os.environ.get("QUIP_FUSED_PROLOGUE", "1") != "0" and (self.chain or prologue_ok)
        # the MLP half of a block (GEMV[gate, up], two transforms, GEMV[down]) as ONE persistent launch
        # (csrc/decode_engine.hip); QUIP_FFN_ENGINE=0 keeps the four stage-wise launches
        self.ffn_eng = (self.fused_prologue and self.chain and os.environ.get("QUIP_FFN_ENGINE", "1") != "0"
                        and all(ffn_engine_ok(L["gate"], L["up"], L["down"]) for L in self.layers))
        self.ffn_ws = None
        if self.ffn_eng:
            from .register_lib import ffn_engine_workspace
            self.ffn_ws = ffn_engine_workspace(s.ffn, L0["gate"].K_right, self.dev)
        # all blocks of a token as ONE persistent launch (csrc/decode_block.hip); QUIP_BLOCK_ENGINE=0 keeps the stage-wise step
        # (E8P12; D4 through the same kernel's one-table mode; E8P12RVQ4B, E8P12RVQ3B and HI as rows of twice the virtual width)
        self.block_eng = False
        d4 = all(getattr(m.codebook, "id", None) in ("D4", "E8P12RVQ4B", "HI", "E8P12RVQ3B") for m in L0.values() if isinstance(m, QuantLinear))
        gqa_shape = self.fused_prologue and self.chain and s.kv_heads != s.heads and s.hidden in (8192, 4096)   # (csrc/decode_block_gqa.hip; decode_block_g8.hip)
        if ((self.ffn_eng or gqa_shape or (d4 and self.fused_prologue and self.chain and os.environ.get("QUIP_FFN_ENGINE", "1") != "0"))
                and os.environ.get("QUIP_BLOCK_ENGINE", "1") != "0" and not self.window):   # (its attention walks [0, pos])
            self._init_block_engine()
        # q / k / v output transforms inside the attention launch (multi-head attention with a power-of-two hidden <= 4096,
        # or 64 / 32 heads on 8 KV heads -- Llama-2-70B, Llama-3, Mistral-7B; plain SV output side)
        from .register_lib import rope_attn_decode_z_supported
        self.attn_z = (self.fused_prologue and self.fused_attention and os.environ.get("QUIP_ATTN_Z", "1") != "0"
                       and rope_attn_decode_z_supported(s.heads, s.kv_heads, s.head_dim)
                       and all(l.K_right == 1 and not l.per_channel and l.bias is None and l.SV is not None
                               and l.q_out_features == l.out_features == (s.heads if i == 0 else s.kv_heads) * s.head_dim
                               for i, l in enumerate(qkv0)))

    def _init_block_engine(self):
        """descriptors + workspace of the persistent block launch, when every block qualifies"""
        import numpy as np
        from .qlinear import _engine_had3
        from .register_lib import block_engine_supported, block_engine_workspace
        s = self.s
        L0 = self.layers[0]
        names = ("q", "k", "v", "o", "gate", "up", "down")

        cbid = getattr(L0["q"].codebook, "id", None)
        if cbid not in ("E8P12", "D4", "E8P12RVQ4B", "HI", "E8P12RVQ3B"):
            return

        def plain(m, n_in, n_out):
            return (getattr(m.codebook, "id", None) == cbid and not m.per_channel and m.bias is None and not m.training
                    and m.SU is not None and m.SV is not None and m.in_features == m.q_in_features == n_in
                    and m.out_features == m.q_out_features == n_out)
        from .register_lib import block_engine_gqa_supported
        with torch.cuda.device(self.dev):           # (the support queries ask the CURRENT device for its CU count)
            return self._init_block_engine_on_device(L0, names, cbid, plain, block_engine_supported, block_engine_gqa_supported,
                                                     block_engine_workspace, _engine_had3)

    def _engine_signature(self):
        """(data_ptr, version) of every tensor the engine descriptors were built from: a load_state_dict / .to() / in-place
        edit of a module after the descriptors were baked shows up here (reset() rebuilds them then)"""
        sig = []
        for L in self.layers:
            for k in ("q", "k", "v", "o", "gate", "up", "down"):
                m = L[k]
                for t in (m.Qidxs if m.Qidxs is not None else getattr(m, "_qidxs_tiled", None), m.SU, m.SV, m.had_left, m.had_right):
                    if t is not None:
                        sig.append((t.data_ptr(), t._version))
                sig.append(float(m.wscale_float))
            sig += [(L["ln1"].data_ptr(), L["ln1"]._version), (L["ln2"].data_ptr(), L["ln2"]._version)]
        return tuple(sig)

    def _init_block_engine_on_device(self, L0, names, cbid, plain, block_engine_supported, block_engine_gqa_supported,
                                     block_engine_workspace, _engine_had3):
        import numpy as np
        s = self.s
        gqa = block_engine_gqa_supported(s.hidden, s.heads, s.kv_heads, s.head_dim, s.ffn, L0["gate"].K_right) and cbid == "E8P12"
        import os
        from .register_lib import block_engine_g8_supported
        # shape 2 (round 5): Llama-3-8B / Mistral-7B -- the shape-0 launch compiled for 32 / 8 heads and n_ffn = 14336 = 56 x 256
        g8 = (not gqa and cbid == "E8P12" and os.environ.get("QUIP_BLOCK_ENGINE_G8", "1") != "0" and not self.window
              and block_engine_g8_supported(s.hidden, s.heads, s.kv_heads, s.head_dim, s.ffn, L0["gate"].K_right))
        ok = ((gqa or g8 or block_engine_supported(s.hidden, s.heads, s.kv_heads, s.head_dim, s.ffn, L0["gate"].K_right))
              and len(self.layers) <= 146)      # (the launch's hand-off counter: 7 per block in 10 bits)
        kvw = s.kv_heads * s.head_dim
        for L in self.layers:
            ok = ok and all(plain(L[k], s.hidden, s.hidden) and L[k].K_left == 1 and L[k].K_right == 1 for k in "qo")
            ok = ok and all(plain(L[k], s.hidden, kvw) and L[k].K_left == 1 and L[k].K_right == 1 for k in "kv")
            ok = ok and all(plain(L[k], s.hidden, s.ffn) and L[k].K_left == 1 for k in ("gate", "up"))
            ok = ok and plain(L["down"], s.ffn, s.hidden) and L["down"].K_right == 1
        if not ok:
            return
        keep, rec = [], np.zeros((len(self.layers), 32), dtype=np.uint64)
        # (a rebuild -- reset() after the modules were edited: the previous descriptors' tensors, for shape 1 a whole tiled copy of
        #  the codes, go BEFORE the new ones are made, not after: no third copy of the weights in between.  ADVICE r5)
        self._eng_keep = None
        self.eng_layers = None
        for i, L in enumerate(self.layers):
            mods = [L[k] for k in names]
            vec = lambda t: t.detach().to(torch.float16).contiguous()     # noqa: E731
            # shape 1 reads the vectors it multiplies in the strided layout of its 512-thread transforms pre-permuted:
            # p[16 t + k] = v[t + 512 k]
            perm = lambda t: vec(t).reshape(16, 512).t().contiguous().reshape(-1)     # noqa: E731
            if gqa:
                su = [perm(m.SU) if k in ("q", "k", "v", "gate", "up") else vec(m.SU) for k, m in zip(names, mods)]
                sv = [perm(m.SV) if k in ("o", "down") else vec(m.SV) for k, m in zip(names, mods)]
                ln = [perm(L["ln1"]), perm(L["ln2"])]
                K = L["gate"].K_right
                mix = torch.zeros(3, K, 8, dtype=torch.float32, device=self.dev)
                mix[0, :, :K] = L["gate"].had_right.detach().float()
                mix[1, :, :K] = L["up"].had_right.detach().float()
                mix[2, :, :K] = L["down"].had_left.detach().float().t()
                had3 = mix.contiguous()
            else:
                # shape 0 (round 5): the 4096-wide edges work in the strided layout too (8 elements per thread):
                # p[8 t + k] = v[t + 512 k] for the vectors they multiply -- ln, SU of q k v gate up, SV of o and down
                perm8 = lambda t: vec(t).reshape(8, 512).t().contiguous().reshape(-1)     # noqa: E731
                su = [perm8(m.SU) if k in ("q", "k", "v", "gate", "up") else vec(m.SU) for k, m in zip(names, mods)]
                sv = [perm8(m.SV) if k in ("o", "down") else vec(m.SV) for k, m in zip(names, mods)]
                ln = [perm8(L["ln1"]), perm8(L["ln2"])]
                if g8:
                    # (R_7 (x) H_2048) / sqrt 2048 on the (7, 2048) view = ((R_7 (x) H_8) (x) H_256) / (sqrt 8 sqrt 256) on the
                    # (56, 256) view: the launch's K = 56 factors are R_7 (x) H_8 (entries +-R_7: exact in fp16); it folds the
                    # 1 / sqrt 8 into its scales (decode_block.hip: kMixScale; sc[6] below is wscale / sqrt 2048 already)
                    h8 = torch.tensor([[1.0 - 2.0 * (bin(i & j).count("1") & 1) for j in range(8)] for i in range(8)], device=self.dev)
                    k56 = lambda r: torch.kron(r.detach().float(), h8)      # noqa: E731
                    had3 = torch.zeros(2 * 3136 + 64 * 72, dtype=torch.float16, device=self.dev)      # (down's: rows of 72: bank spread)
                    had3[:3136] = k56(L["gate"].had_right).to(torch.float16).reshape(-1)
                    had3[3136:6272] = k56(L["up"].had_right).to(torch.float16).reshape(-1)
                    hdT = torch.zeros(64, 72, dtype=torch.float16, device=self.dev)
                    hdT[:56, :56] = k56(L["down"].had_left).to(torch.float16).T
                    had3[6272:] = hdT.reshape(-1)
                else:
                    had3 = _engine_had3(L["gate"], L["up"], L["down"])
            keep += su + sv + ln + [had3]
            if gqa:
                # shape 1 streams a re-tiled copy of the codes (decode_block_gqa.hip: full-line requests): one more copy of the
                # weights in HBM, made once per model; the checkpoint's tensors stay what the stage-wise step and prefill read
                wq = []
                for m in mods:
                    t = getattr(m, "_qidxs_tiled", None)          # (a rebuild in single-copy mode: the tiled copy is all there is)
                    if m.Qidxs is not None:
                        t = tile_codes(m.Qidxs)
                    if getattr(self, "single_copy", False):
                        if m.Qidxs is not None:
                            m._qidxs_meta = (tuple(m.Qidxs.shape), m.Qidxs.dtype)
                        m._qidxs_tiled = t
                        m.Qidxs = None
                    wq.append(t)
                keep += wq
            else:
                wq = [m.Qidxs for m in mods]
            ptrs = ([t.data_ptr() for t in wq] + [t.data_ptr() for t in ln] + [t.data_ptr() for t in su]
                    + [t.data_ptr() for t in sv] + [had3.data_ptr(), self.kcache[i].data_ptr(), self.vcache[i].data_ptr()])
            rec[i, :26] = np.array(ptrs, dtype=np.uint64)
            sc = [m.wscale_float / math.sqrt(m.q_in_features // m.K_left) for m in mods]
            rec[i, 26:].view(np.float32)[:7] = np.array(sc, dtype=np.float32)
        self._eng_keep = keep
        self.eng_layers = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(self.dev)
        self.eng_shape = 1 if gqa else (2 if g8 else 0)
        self.eng_ws = block_engine_workspace(self.dev, self.eng_shape)
        self.eng_codebook = {"E8P12": 0, "D4": 1, "E8P12RVQ4B": 2, "HI": 3, "E8P12RVQ3B": 4}[cbid]
        cb0 = L0["q"].codebook
        self.eng_grid = cb0.grid if cbid == "D4" else (cb0._virtual_grid(self.dev) if cbid == "HI" else cb0.grid_packed_abs)
        self.eng_resid_scale = (float(getattr(L0["q"].codebook, "planes_resid_scale", 0.0))
                                if cbid in ("E8P12RVQ4B", "E8P12RVQ3B") else 0.0)
        self.eng_grid2 = cb0._e81b_i8(self.dev) if cbid == "E8P12RVQ3B" else None      # int8 (256, 8): 4 x the E81B entries
        self._eng_sig = self._engine_signature()
        self._kv_ptr_host = torch.empty(len(self.layers), 2, dtype=torch.int64).pin_memory()      # (bind_kv's staging buffer)
        self.block_eng = True

    def engine_status(self):
        """0, or the code of a wait that gave up inside a persistent launch (synchronises)"""
        from .register_lib import ffn_engine_status
        for ws in (getattr(self, "eng_ws", None), getattr(self, "ffn_ws", None)):
            if ws is not None and ffn_engine_status(ws) != 0:
                return ffn_engine_status(ws)
        return 0

    def engine_fail_position(self):
        """position of the first token a persistent block launch did not compute (workspace word 2 - 1), or None"""
        ws = getattr(self, "eng_ws", None)
        if ws is None:
            return None
        v = int(ws[8:12].view(torch.int32).item())
        return v - 1 if v > 0 else None

    def _check_kv(self, keys, values):
        s = self.s
        keys, values = list(keys), list(values)
        if len(keys) != s.layers or len(values) != s.layers:
            raise ValueError(f"kv_cache: {s.layers} key and value tensors expected")
        for t in keys + values:
            if (tuple(t.shape) != (s.kv_heads, self.max_len, s.head_dim) or t.dtype != torch.float16 or not t.is_contiguous()
                    or t.device != self.dev):
                raise ValueError(f"kv_cache tensors: contiguous fp16 ({s.kv_heads}, {self.max_len}, {s.head_dim}) on {self.dev}")
        return keys, values

    def bind_kv(self, keys, values):
        """use other cache tensors (same shapes) from now on: the stage-wise step reads self.kcache[i] at every call; the
        block launch's descriptors hold the row pointers (words 24, 25 of a 256-byte descriptor) and are patched in place;
        a captured step is dropped (its launches hold the old pointers)"""
        self.kcache, self.vcache = self._check_kv(keys, values)
        if getattr(self, "block_eng", False):
            # through a pinned staging buffer that lives as long as the decoder: the copy is legal inside a stream capture
            # (a re-recording torch.compile graph meets a new cache object there) and replays read the same host memory
            host = getattr(self, "_kv_ptr_host", None)
            if host is None:
                host = self._kv_ptr_host = torch.empty(len(self.layers), 2, dtype=torch.int64).pin_memory()
            for i, (k, v) in enumerate(zip(self.kcache, self.vcache)):
                host[i, 0], host[i, 1] = k.data_ptr(), v.data_ptr()
            rec = self.eng_layers.view(torch.int64).view(len(self.layers), 32)
            rec[:, 24:26].copy_(host, non_blocking=True)
        self.graph = None

    def engine_reset(self):
        """after a launch that gave up: workspaces back to their allocation state (generation 0, no granules, no code)"""
        for ws in (getattr(self, "eng_ws", None), getattr(self, "ffn_ws", None)):
            if ws is not None:
                ws.zero_()

    # ---- single-copy mode: a block's matrices in the checkpoint's layout, for the operators that read it ------------------------
    def _rm_enter(self, L):
        """the seven modules of block L get their row-major `Qidxs` back (in scratch buffers shared by all blocks, filled by
        untile_codes on the current stream) until _rm_exit; a no-op unless the modules hold tiled copies only"""
        if not getattr(self, "single_copy", False):
            return
        scr = self.__dict__.setdefault("_rm_scratch", {})
        for k in ("q", "k", "v", "o", "gate", "up", "down"):
            m = L[k]
            if m.Qidxs is None and getattr(m, "_qidxs_tiled", None) is not None:
                shape, dtype = m._qidxs_meta
                buf = scr.get(k)
                if buf is None or tuple(buf.shape) != shape or buf.dtype != dtype:
                    buf = scr[k] = torch.empty(shape, dtype=dtype, device=self.dev)
                untile_codes(m._qidxs_tiled, shape[0], buf.numel() * buf.element_size() // shape[0], buf)
                m.Qidxs = buf
                m._qidxs_borrowed = True

    def _rm_exit(self, L):
        if not getattr(self, "single_copy", False):
            return
        for k in ("q", "k", "v", "o", "gate", "up", "down"):
            m = L[k]
            if getattr(m, "_qidxs_borrowed", False):
                m.Qidxs = None
                m._qidxs_borrowed = False

    # ---- model bytes the decode step has to stream (roofline denominator) -----------
    def algorithmic_bytes_per_token(self):
        b = 0
        for L in self.layers:
            for k in ("q", "k", "v", "o", "gate", "up", "down"):
                m = L[k]
                b += qidxs_nbytes(m) + 2 * (m.in_features + m.out_features)
        return b + self.lm_head.numel() * 2

    def _rope(self, x, cos, sin):
        d = x.shape[-1] // 2
        rot = torch.cat([-x[..., d:], x[..., :d]], -1)
        return (x.float() * cos + rot.float() * sin).to(x.dtype)

    def _attention(self, i, q, k, v, cos, sin, mask):
        s = self.s
        if self.fused_attention:
            # rope + cache append + attention over [0, pos]: one launch
            return torch.ops.quip_lib.rope_attn_decode(
                q.view(s.heads, s.head_dim), k.view(s.kv_heads, s.head_dim), v.view(s.kv_heads, s.head_dim),
                self.cos, self.sin, self.pos, self.kcache[i], self.vcache[i], self.attn_ws, self.window)
        q = self._rope(q.view(1, s.heads, 1, s.head_dim), cos, sin)
        k = self._rope(k.view(1, s.kv_heads, 1, s.head_dim), cos, sin)
        self.kcache[i].index_copy_(1, self.pos, k[0])
        self.vcache[i].index_copy_(1, self.pos, v.view(s.kv_heads, 1, s.head_dim))
        return F.scaled_dot_product_attention(q, self.kcache[i][None], self.vcache[i][None], attn_mask=mask,
                                              enable_gqa=(s.kv_heads != s.heads))

    def step(self):
        """one token: reads self.tok / self.pos, writes the greedy next token into self.tok and
        advances self.pos (all on the device)"""
        s = self.s
        h = self.embed[self.tok]                                   # (1, hidden)
        cos = sin = mask = None
        if not self.fused_attention:
            cos, sin = self.cos[self.pos], self.sin[self.pos]          # (1, head_dim)
            mask = (self.arange[None, None, None, :] <= self.pos)      # (1,1,1,max_len) keys <= current
            if self.window:
                mask = mask & (self.arange[None, None, None, :] > self.pos - self.window)
        if getattr(self, "block_eng", False):
            h = torch.ops.quip_lib.block_engine(self.eng_layers, h.reshape(-1), self.pos, self.cos, self.sin,
                                                self.eng_grid, self.eng_ws, len(self.layers), self.max_len, s.rms_eps,
                                                1.0 / math.sqrt(s.head_dim), None, -1, self.eng_codebook, self.eng_resid_scale,
                                                getattr(self, "eng_shape", 0), getattr(self, "eng_grid2", None),
                                                *((self.kcache, self.vcache) if torch.is_tensor(self.kcache) else (None, None)))
            return self._head(h.reshape(1, -1))
        if self.fused_prologue:
            return self._step_fused(h, cos, sin, mask)
        for i, L in enumerate(self.layers):
            self._rm_enter(L)
            # q / k / v (and gate / up below): one launch per stage for the whole group; RMSNorm rides on
            # the input-side Hadamard launch
            q, k, v = forward_group([L["q"], L["k"], L["v"]], h, rms_weight=L["ln1"], rms_eps=s.rms_eps)
            a = self._attention(i, q, k, v, cos, sin, mask)
            # residual adds ride on the output-side Hadamard launch, SiLU(gate)*up on down's input side
            h = L["o"].forward_fused(a.reshape(1, s.hidden), residual=h)
            g, u = forward_group([L["gate"], L["up"]], h, rms_weight=L["ln2"], rms_eps=s.rms_eps)
            h = L["down"].forward_fused(u, gate=g, residual=h)
            self._rm_exit(L)
        return self._head(h)

    def _step_fused(self, h, cos, sin, mask):
        """8 launches per block: the GEMV launches of q/k/v, o and gate/up compute their own input
        transform (RMSNorm, SU, Hadamard) and the output transform + residual of the module before
        them (down of the previous block, o) in their prologue."""
        s = self.s
        zd = prev_down = None
        for i, L in enumerate(self.layers):
            self._rm_enter(L)
            qkv = [L["q"], L["k"], L["v"]]
            if zd is None and self.qkv_fused:
                _, zs = gemv_fused(qkv, x=h, rms_weight=L["ln1"], rms_eps=s.rms_eps)
            elif zd is None:
                zs = gemv_group_unfused(qkv, h, rms_weight=L["ln1"], rms_eps=s.rms_eps)
            else:   # finishes the previous block: h += down(...)
                h, zs = self._zx(qkv, prev_down, zd, h, L["ln1"])
            if self.attn_z:
                # the K = 1 output transforms of q / k / v in the attention launch's prologue: 9 launches per block
                a = torch.ops.quip_lib.rope_attn_decode_z(
                    list(zs), [l._vec(l.SV) for l in qkv], [1.0 / math.sqrt(l.q_out_features) for l in qkv],
                    self.cos, self.sin, self.pos, self.kcache[i], self.vcache[i], self.attn_ws, self.window)
            else:
                q, k, v = out_transform_group(qkv, zs)
                a = self._attention(i, q, k, v, cos, sin, mask)
            if self.o_fused:
                _, (zo,) = gemv_fused([L["o"]], x=a.reshape(1, s.hidden))
            else:
                zo = gemv_unfused(L["o"], a.reshape(1, s.hidden))
            if self.ffn_eng:
                h, planes = chain_planes([L["gate"], L["up"]], L["o"], zo, residual=h, rms_weight=L["ln2"],
                                         rms_eps=s.rms_eps)
                zd = ffn_engine(L["gate"], L["up"], L["down"], planes, self.ffn_ws)
            else:
                h, zgu = self._zx([L["gate"], L["up"]], L["o"], zo, h, L["ln2"])
                g, u = out_transform_group([L["gate"], L["up"]], zgu)
                zd = gemv_unfused(L["down"], u, gate=g)
            prev_down = L["down"]
            self._rm_exit(L)
        (h,) = out_transform_group([prev_down], [zd], residual=[h])
        return self._head(h)

    def _zx(self, layers, prev, z, residual, ln):
        """producer's output side + consumers' input side + GEMV: as a Hadamard chain launch (one
        workgroup per consumer, in parallel) followed by the grouped GEMV, or inside the GEMV
        prologue (every workgroup repeats all transforms one after the other)"""
        if self.chain:
            return gemv_chain(layers, prev, z, residual=residual, rms_weight=ln, rms_eps=self.s.rms_eps)
        return gemv_fused(layers, prev=prev, z=z, residual=residual, rms_weight=ln, rms_eps=self.s.rms_eps)

    def set_sampling(self, temperature=None, top_k=None):
        """greedy (default, temperature None / 0) or the reference demo's sampler
        (example_generate.py:9-26: logits / T, optional top-k cut, softmax, exponential-race arg-max --
        no host synchronisation).  The choice is part of the captured step: changing it re-captures."""
        new = None if not temperature else (float(temperature), None if top_k is None else int(top_k))
        if new != getattr(self, "sampling", None):
            self.sampling, self.graph = new, None

    def _head(self, h):
        s = self.s
        logits = F.rms_norm(h, (s.hidden,), self.final_norm, s.rms_eps) @ self.lm_head.T
        if getattr(self, "sampling", None) is None:
            if logits.dtype == torch.float16 and logits.is_cuda:
                torch.ops.quip_lib.argmax_step(logits, self.tok, self.pos)      # tok <- argmax, pos += 1: one launch
                return logits
            self.tok.copy_(logits.argmax(-1))
        else:
            temperature, top_k = self.sampling
            lg = logits.float() / max(temperature, 1e-5)
            if top_k is not None:
                v, _ = torch.topk(lg, min(top_k, lg.size(-1)))
                lg = torch.where(lg < v[..., -1:], -float("inf"), lg)
            probs = torch.softmax(lg, dim=-1)
            q = torch.empty_like(probs).exponential_(1)
            self.tok.copy_(torch.argmax(probs / q, dim=-1))
        self.pos.add_(1)
        return logits

    prefill_graph_cache_size = 8      # captured prompt lengths kept (each graph owns a memory pool)

    @torch.no_grad()
    def prefill_graph(self, tokens):
        """prefill() replayed from a hipGraph captured per prompt LENGTH (first call of a length captures: a serving
        loop would bucket its prompt lengths).  Short prompts are bound by the ~600 eager launches of the pass, not by
        the GPU: 33..256 tokens 18.7 -> 4..8 ms on the 32-layer 7B model (tools/ttft_bench.py --graph)."""
        tokens = torch.as_tensor(tokens, dtype=torch.long, device=self.dev).reshape(-1)
        P = tokens.numel()
        cache = self.__dict__.setdefault("_prefill_graphs", {})
        if P not in cache and len(cache) >= self.prefill_graph_cache_size:
            cache.pop(next(iter(cache)))              # oldest captured length out (a graph keeps its own memory pool)
        if P not in cache:
            static_tok = tokens.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                self.prefill(static_tok)                      # warm-up (attribute setup, allocator)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g):
                logits = self.prefill(static_tok)
            torch.cuda.synchronize()
            cache[P] = (g, static_tok, logits)
        g, static_tok, logits = cache[P]
        static_tok.copy_(tokens)
        g.replay()
        return logits

    def prefill(self, tokens):
        """Batched prompt pass (the reference demo's prefill, example_generate.py:36-47): all `tokens` (1-D ids) go
        through every block at once -- QuantLinear on (P, hidden) rows (M >= 32: skinny chunks or decompress + dense GEMM, fewer
        rows: the skinny paths), rotary embedding for positions 0..P-1, causal attention, K / V written to rows 0..P-1
        of the static cache -- and the position counter is left at P.  Returns the logits of the last token (1, vocab)."""
        s = self.s
        tokens = torch.as_tensor(tokens, dtype=torch.long, device=self.dev).reshape(-1)
        P = tokens.numel()
        assert 1 <= P <= self.max_len
        h = self.embed[tokens]                                          # (P, hidden)
        cos, sin = self.cos[:P], self.sin[:P]                           # (P, head_dim)
        for i, L in enumerate(self.layers):
            self._rm_enter(L)
            q, k, v = forward_group([L["q"], L["k"], L["v"]], h, rms_weight=L["ln1"], rms_eps=s.rms_eps)
            q = self._rope(q.view(P, s.heads, s.head_dim).transpose(0, 1), cos, sin)          # (heads, P, hd)
            k = self._rope(k.view(P, s.kv_heads, s.head_dim).transpose(0, 1), cos, sin)
            v = v.view(P, s.kv_heads, s.head_dim).transpose(0, 1)
            self.kcache[i][:, :P].copy_(k)
            self.vcache[i][:, :P].copy_(v)
            if self.window and P > self.window:     # causal band: key t for query p iff p - window < t <= p
                band = (self.arange[:P, None] >= self.arange[None, :P]) & (self.arange[:P, None] - self.arange[None, :P] < self.window)
                a = F.scaled_dot_product_attention(q[None], k[None], v[None], attn_mask=band,
                                                   enable_gqa=(s.kv_heads != s.heads))[0]
            else:
                a = F.scaled_dot_product_attention(q[None], k[None], v[None], is_causal=True,
                                                   enable_gqa=(s.kv_heads != s.heads))[0]     # (heads, P, hd)
            h = L["o"].forward_fused(a.transpose(0, 1).reshape(P, s.hidden), residual=h)
            g, u = forward_group([L["gate"], L["up"]], h, rms_weight=L["ln2"], rms_eps=s.rms_eps)
            h = L["down"].forward_fused(u, gate=g, residual=h)
            self._rm_exit(L)
        self.pos.fill_(P)
        return F.rms_norm(h[-1:], (s.hidden,), self.final_norm, s.rms_eps) @ self.lm_head.T

    def reset(self, first_token=1):
        if getattr(self, "block_eng", False) and getattr(self, "_eng_sig", None) != self._engine_signature():
            # a module's tensors were replaced or edited since the descriptors were baked: rebuild them (and the captured step)
            for L in self.layers:
                if hasattr(L["down"], "_eng_had3"):
                    del L["down"]._eng_had3
            self.block_eng = False
            self._init_block_engine()
            self.graph = None
        self.tok.fill_(first_token)
        self.pos.zero_()

    def capture(self):
        """warm up (kernel attribute setup, allocator) and capture one step as a hipGraph"""
        self.reset()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                self.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.step_logits = self.step()      # static output of the captured step (valid after each replay)
        torch.cuda.synchronize()
        self.reset()

    @torch.no_grad()
    def generate(self, n_tokens, first_token=1, use_graph=True, prompt=None, temperature=None, top_k=None,
                 batched_prefill=True, prefill_graph=False):
        """decode n_tokens (greedy, or sampled when a temperature is given: set_sampling); returns the
        token ids (device tensor).  `prompt` (1-D token ids): all but its last token go through ONE batched
        pass (`prefill`; batched_prefill=False feeds them token by token through the captured step instead, teacher
        forced), then decoding continues from the last prompt token; prompt length + n_tokens <= max_len + 1."""
        if prompt is not None:
            prompt = torch.as_tensor(prompt, dtype=torch.long, device=self.dev).reshape(-1)
            first_token = int(prompt[0])
        n_prompt = 0 if prompt is None else prompt.numel() - 1
        assert n_prompt + n_tokens <= self.max_len
        self.set_sampling(temperature, top_k)
        self.reset(first_token)
        if use_graph and self.graph is None:
            self.capture()
            self.reset(first_token)
        out = torch.empty(n_tokens, dtype=torch.long, device=self.dev)
        if batched_prefill and n_prompt >= 1:
            # all prompt tokens but the last in one batched pass (fills cache rows 0..P-2); the last one goes through
            # the captured step like every generated token
            (self.prefill_graph if prefill_graph else self.prefill)(prompt[:-1])   # graph: captured per prompt length
            self.tok.copy_(prompt[-1:].view_as(self.tok))
            n_prompt = 0
        pos0 = prompt.numel() - 1 if (batched_prefill and prompt is not None and prompt.numel() > 1) else 0

        replay = [bool(use_graph)]

        def run(t_from):
            for t in range(t_from, n_prompt + n_tokens):
                if replay[0]:
                    self.graph.replay()
                else:
                    self.step_logits = self.step()
                if t < n_prompt:
                    self.tok.copy_(prompt[t + 1:t + 2].view_as(self.tok))
                else:
                    out[t - n_prompt] = self.tok.reshape(-1)[0]
        run(0)
        # A persistent launch whose workgroups were not all resident (something else on the device) gives up on a hand-off
        # instead of hanging, leaves a code, answers NaN, and remembers the position of the first token it did not compute:
        # that token and everything behind it is decoded again -- on the stage-wise step, unless the code only says that the
        # workspace wants zeroing (0xE000: its launch counter is about to wrap).  Cache rows of earlier positions are results.
        for _ in range(3):
            if not (getattr(self, "block_eng", False) or getattr(self, "ffn_eng", False)):
                break
            st = self.engine_status()
            if not st:
                break
            fail = self.engine_fail_position()
            self.engine_reset()
            if st != 0xE000:
                import warnings
                warnings.warn("persistent decode launch gave up on a hand-off (code 0x%x): the device was shared with other "
                              "work; this decoder continues on the stage-wise step" % st)
                self.block_eng = self.ffn_eng = False
                self.graph = None
            t_from = 0 if fail is None else max(0, fail - pos0)
            if use_graph and self.graph is None:
                # The rest of THIS call runs eagerly: capture() warms up with two steps at positions 0 and 1, which would
                # overwrite cache rows 0 and 1 of every layer -- rows the resumed decode still attends to.  The next
                # generate() call captures the stage-wise step (its reset + prompt pass rewrite those rows anyway).
                replay[0] = False
            self.pos.fill_(pos0 + t_from)
            if t_from == 0:
                self.tok.fill_(first_token if not (batched_prefill and prompt is not None and prompt.numel() > 1) else int(prompt[-1]))
            elif t_from <= n_prompt:
                self.tok.copy_(prompt[t_from:t_from + 1].view_as(self.tok))
            else:
                self.tok.copy_(out[t_from - 1 - n_prompt].view_as(self.tok))
            run(t_from)
        return out
