"""Single-token decode steps of an HF model returned by `load_quantized_model`, routed through `LlamaDecoder`.

The reference's own decode loop (example_generate.py:28-33, 62-70) and HF `generate(cache_implementation="static")` call
`model(input_ids (1, 1), past_key_values=<StaticCache>, ...)` once per token; with the stock modules that is ~45 launches
per decoder block (three per QuantLinear plus the framework's element-wise kernels).  `enable_fast_decode(model)` wraps
`model.forward`: a call that IS such a step -- one token, batch 1, an initialised `StaticCache`, fp16, no extra outputs
asked for -- runs `LlamaDecoder.step()` on the SAME modules and on the cache object's own key / value tensors (row
`cumulative_length` is written, the lengths advance by one), i.e. the persistent block launch where the shape has one and
the fused stage-wise step elsewhere.  Every other call (the prompt, batches, dynamic caches, `output_attentions`,
`inputs_embeds`, training) goes to the original forward untouched, so the model stays a drop-in HF model.

The default cache of `model.generate()` (`DynamicCache`: key / value tensors that grow by concatenation) is served too: the
wrapper keeps ONE static buffer per layer (`max_position_embeddings` rows, at most QUIP_FAST_DECODE_MAX_LEN = 8192), copies
the prompt's rows into it at the first decode step and from then on hands the cache object VIEWS of the buffer
(`layer.keys = buffer[:, :, :n + 1]`), so the object keeps answering `get_seq_length()` and can go back to the stock
forward at any time (which concatenates into fresh tensors; the next fast step imports them again).  A second cache
object taking over the buffer first gets the previous owner's views cloned, so no two caches ever alias.

That `position_ids` equals the cache length and that `attention_mask` has no holes (an unpadded single sequence: what
batch 1 generation passes) is checked when the wrapper takes a cache object over -- one device synchronisation per
generation, not per token; a padded sequence stays on the stock forward.
Logits are those of `LlamaDecoder` (same arithmetic as the stage-wise ops, tests/test_gpu_hf_generate.py pins them against
the stock forward); the cache rows are bit-compatible (rotated keys, fp16), so fast and stock steps may be mixed freely."""
import torch

from . import register_lib as _R

# The static-cache step as ONE operator, so that a caller's torch.compile(..., fullgraph=True) -- the reference's own decode
# loop compiles its step that way (example_generate.py:69-70), HF's static-cache generate compiles the forward -- traces
# through the wrapper instead of meeting a function it may not inline: input token, the cache object's key / value tensors
# and lengths (mutated in place, declared so), an integer handle naming the wrapper; the logits (1, 1, vocab) come back.
_HANDLES = {}
import os as _os
_ASSUME_UNPADDED = _os.environ.get("QUIP_FAST_DECODE_ASSUME_UNPADDED", "0") != "0"
_PREFILL = _os.environ.get("QUIP_FAST_DECODE_PREFILL", "1") != "0"
try:
    _R._lib.define("hf_decode_step(Tensor input_ids, Tensor(a!)[] keys, Tensor(b!)[] values, Tensor(c!)[] lens, int handle) -> Tensor")
except RuntimeError:
    pass


def _wrapper_of(handle):
    fd = _HANDLES.get(handle, lambda: None)()       # (weak references: a wrapper goes away with its model)
    if fd is None:
        raise RuntimeError("quip_lib::hf_decode_step: the model this compiled graph was traced for no longer exists")
    return fd


def _hf_decode_step_cuda(input_ids, keys, values, lens, handle):
    return _wrapper_of(handle)._static_step(input_ids, keys, values, lens)


def _hf_decode_step_fake(input_ids, keys, values, lens, handle):
    return input_ids.new_empty((1, 1, int(_wrapper_of(handle).model.config.vocab_size)), dtype=torch.float16)


try:
    _R._lib.impl("hf_decode_step", _hf_decode_step_cuda, "CUDA")
    _R._reg_fake("hf_decode_step", _hf_decode_step_fake)
except RuntimeError:
    pass


def _mask_has_holes(mask, n_total):
    """does an attention mask hide any of the first n_total positions from the LAST query row?  2-D (1, len) masks of ones /
    zeros, or the 4-D (1, 1, q, kv) masks HF prepares for static caches (bool: True = attend; float: 0 = attend).
    Reads device memory."""
    if mask is None or not torch.is_tensor(mask):
        return False
    n_total = int(n_total)
    if mask.dim() == 2:
        return not bool((mask[:, :n_total] != 0).all())
    if mask.dim() == 4:
        row = mask[0, 0, -1, :n_total]
        ok = row if row.dtype == torch.bool else (row == 0)
        return not bool(ok.all())
    return True                         # (a layout this wrapper does not know: not vouched for)


def _off_thread(fn):
    """run fn() on a helper thread and hand back its result.  A decoder lives as long as the wrapper; when its first use falls
    into the warm-up run of torch.compile(mode="reduce-overhead"), the calling thread's allocations are being routed into the
    cudagraph trees' private pool, which then refuses to record because of live allocations it did not hand out.  That routing
    is per thread: what another thread allocates comes from the ordinary pool."""
    import threading
    box = {}
    dev = torch.cuda.current_device()

    def run():
        try:
            torch.cuda.set_device(dev)
            with torch.no_grad():
                box["v"] = fn()
            torch.cuda.synchronize(dev)
        except BaseException as e:      # noqa: BLE001 (re-raised on the caller's thread)
            box["e"] = e
    t = threading.Thread(target=run)
    t.start()
    t.join()
    if "e" in box:
        raise box["e"]
    return box["v"]


class _FastDecode:
    def __init__(self, model, assume_llama_like=False):
        self.model = model
        self.orig_forward = model.forward
        self.assume_llama_like = assume_llama_like
        self.dec = None              # decoder on a StaticCache's tensors
        self.bound = None            # the key / value data pointers that decoder currently uses
        self.dyn = None              # decoder on this wrapper's own buffers (DynamicCache calls)
        self.dyn_owner = None        # weakref of the cache object whose layers are views of the buffers
        self.dyn_len = 0
        self.disabled = None         # the reason LlamaDecoder refused this model, once known
        self.fast_steps = 0
        self.fast_prefills = 0
        import weakref
        _FastDecode._next_handle = getattr(_FastDecode, "_next_handle", 0) + 1
        self.handle = _FastDecode._next_handle
        _HANDLES[self.handle] = weakref.ref(self, lambda _r, h=self.handle: _HANDLES.pop(h, None))
        self._precheck()

    def _precheck(self):
        """the refusals of LlamaDecoder.from_hf that can be known without building a decoder, known NOW: inside a
        torch.compile trace the wrapper must decide between the operator and the stock forward without running anything"""
        from .decode import LlamaDecoder
        from .qlinear import QuantLinear
        cfg = self.model.config
        try:
            if getattr(cfg, "model_type", "") not in LlamaDecoder.LLAMA_LIKE and not self.assume_llama_like:
                raise NotImplementedError(f"model_type {getattr(cfg, 'model_type', '')!r}")
            if getattr(cfg, "hidden_act", "silu") != "silu":
                raise NotImplementedError(f"hidden_act {cfg.hidden_act!r}")
            rp = getattr(cfg, "rope_parameters", None) or getattr(cfg, "rope_scaling", None) or {}
            if isinstance(rp, dict) and rp.get("rope_type", "default") in ("dynamic", "longrope"):
                raise NotImplementedError(f"rope_type {rp.get('rope_type')!r}")
            for blk in self.model.model.layers:
                a, m = blk.self_attn, blk.mlp
                for mod in (a.q_proj, a.k_proj, a.v_proj, a.o_proj, m.gate_proj, m.up_proj, m.down_proj):
                    if not isinstance(mod, QuantLinear):
                        raise TypeError(f"{type(mod).__name__} where a QuantLinear is expected")
        except (NotImplementedError, TypeError, AttributeError) as e:
            self.disabled = repr(e)

    # -- eligibility: shapes and types only, nothing that reads device memory
    def _static_layers(self, cache):
        """("static" | "dynamic", layers) of a cache this wrapper serves, else None"""
        layers = getattr(cache, "layers", None)
        if not layers or len(layers) != self.model.config.num_hidden_layers:
            return None
        kind = type(layers[0]).__name__
        if kind not in ("StaticLayer", "DynamicLayer"):
            return None
        for L in layers:
            if (type(L).__name__ != kind or not getattr(L, "is_initialized", False) or L.keys.dim() != 4 or L.keys.shape[0] != 1
                    or L.keys.dtype != torch.float16 or not L.keys.is_cuda):
                return None
        if kind == "DynamicLayer":
            n = layers[0].keys.shape[-2]
            if n < 1 or n + 1 > self._dyn_capacity() or any(L.keys.shape[-2] != n for L in layers):
                return None
            return "dynamic", layers
        return "static", layers

    def _dyn_capacity(self):
        import os
        cap = int(os.environ.get("QUIP_FAST_DECODE_MAX_LEN", "8192"))
        return min(cap, int(getattr(self.model.config, "max_position_embeddings", cap) or cap))

    def _eligible(self, input_ids, past_key_values, inputs_embeds, labels, kw):
        if self.disabled is not None or input_ids is None or inputs_embeds is not None or labels is not None:
            return None
        if tuple(input_ids.shape) != (1, 1) or not input_ids.is_cuda or self.model.training:
            return None
        if kw.get("output_attentions") or kw.get("output_hidden_states"):
            return None
        return self._static_layers(past_key_values) if past_key_values is not None else None

    def _decoder(self, keys4, values4):
        from .decode import LlamaDecoder
        keys = [k[0] for k in keys4]
        values = [v[0] for v in values4]
        sig = tuple(t.data_ptr() for t in keys + values)
        max_len = keys[0].shape[-2]
        if self.dec is None or self.dec.max_len != max_len or self.dec.dev != keys[0].device:
            try:
                self.dec = _off_thread(lambda: LlamaDecoder.from_hf(self.model, max_len=max_len,
                                                                    assume_llama_like=self.assume_llama_like,
                                                                    kv_cache=(keys, values)))
            except (NotImplementedError, TypeError, ValueError) as e:
                self.disabled = repr(e)
                return None
            self.bound = sig
        elif sig != self.bound:
            self.dec.bind_kv(keys, values)
            self.bound = sig
        return self.dec

    @staticmethod
    def _unpadded(attention_mask, position_ids, n):
        """one token at cache length n of a single sequence without padding?  (reads device memory: called once per cache take-over)"""
        if torch.cuda.is_current_stream_capturing():
            return True
        n = int(n)
        if _mask_has_holes(attention_mask, n + 1):
            return False
        if position_ids is not None and position_ids.numel() >= 1 and int(position_ids.reshape(-1)[-1]) != int(n):
            return False
        return True

    @staticmethod
    def _positions_from_zero(position_ids, P):
        if position_ids is None:
            return True
        if position_ids.numel() != P:
            return False
        return bool((position_ids.reshape(-1) == torch.arange(P, device=position_ids.device)).all())

    def _step_checked(self, dec, set_inputs):
        """one decoder step; when the call is not being captured into a graph, the persistent launch's status word is read
        back (one small synchronising copy per token -- HF's generate loop synchronises per token anyway) and a launch that
        gave up on a hand-off (a shared device: NaN logits, decode.py) is answered by the same step on the stage-wise path"""
        set_inputs()
        logits = dec.step()
        if (getattr(dec, "block_eng", False) or getattr(dec, "ffn_eng", False)) and not torch.cuda.is_current_stream_capturing():
            st = dec.engine_status()
            if st:
                dec.engine_reset()
                if st != 0xE000:        # (0xE000: the workspace's launch counter was about to wrap -- zeroed, same path again)
                    import warnings
                    warnings.warn("persistent decode launch gave up on a hand-off (code 0x%x): the device was shared with "
                                  "other work; fast decode continues on the stage-wise step" % st)
                    dec.block_eng = dec.ffn_eng = False
                    dec.graph = None
                set_inputs()
                logits = dec.step()
        return logits

    def _dynamic_decoder(self, dev):
        from .decode import LlamaDecoder
        if self.dyn is None or self.dyn.dev != dev:
            try:
                self.dyn = _off_thread(lambda: LlamaDecoder.from_hf(self.model, max_len=self._dyn_capacity(),
                                                                    assume_llama_like=self.assume_llama_like))
            except (NotImplementedError, TypeError, ValueError) as e:
                self.disabled = repr(e)
                return None
            self.dyn_owner = None
        return self.dyn

    def _release_owner(self, cache):
        """a cache object other than `cache` that holds views of the buffers keeps its contents: the views become its own tensors"""
        owner = self.dyn_owner() if self.dyn_owner is not None else None
        if owner is not None and owner is not cache:
            for L in getattr(owner, "layers", []):
                if getattr(L, "is_initialized", False) and torch.is_tensor(L.keys) and L.keys.numel():
                    L.keys, L.values = L.keys.clone(), L.values.clone()

    @torch.compiler.disable
    def _dynamic_prefill(self, input_ids, cache, attention_mask, kw):
        """the prompt pass of a single unpadded sequence on an EMPTY DynamicCache (what generate() starts with), last-token
        logits only (logits_to_keep = 1): LlamaDecoder.prefill on the wrapper's buffers -- one batched pass, grouped launches --
        and the cache object gets views of the first P rows.  Anything else: None (the stock forward runs)."""
        import weakref
        if self.disabled is not None or type(cache).__name__ != "DynamicCache":
            return None
        layers = getattr(cache, "layers", None)
        if not layers or len(layers) != self.model.config.num_hidden_layers:
            return None
        if any(type(L).__name__ != "DynamicLayer" or getattr(L, "is_initialized", False) for L in layers):
            return None
        P = input_ids.shape[1]
        if P + 1 > self._dyn_capacity() or _mask_has_holes(attention_mask, P):
            return None
        dec = self._dynamic_decoder(input_ids.device)
        if dec is None or dec.window:
            return None
        with torch.no_grad():
            self._release_owner(cache)
            logits = dec.prefill(input_ids.reshape(-1))
            for i, L in enumerate(layers):
                L.lazy_initialization(dec.kcache[i][None, :, :0], dec.vcache[i][None, :, :0])
                L.keys = dec.kcache[i][None, :, :P]
                L.values = dec.vcache[i][None, :, :P]
        self.dyn_owner = weakref.ref(cache)
        self.dyn_len = P
        try:
            cache._quip_padded = False
        except Exception:       # noqa: BLE001
            pass
        self.fast_prefills += 1
        return logits.reshape(1, 1, -1)

    def _dynamic_step(self, input_ids, cache, layers, attention_mask=None, position_ids=None):
        import weakref
        from .decode import LlamaDecoder
        n = layers[0].keys.shape[-2]
        dev = layers[0].keys.device
        dec = self._dynamic_decoder(dev)
        if dec is None:
            return None
        owner = self.dyn_owner() if self.dyn_owner is not None else None
        owned = (owner is cache and n <= self.dyn_len and layers[0].keys.data_ptr() == dec.kcache[0].data_ptr()
                 and layers[-1].values.data_ptr() == dec.vcache[-1].data_ptr())
        if not owned and not self._unpadded(attention_mask, position_ids, n):
            return None
        with torch.no_grad():
            if not owned:
                self._release_owner(cache)
                for i, L in enumerate(layers):
                    dec.kcache[i][:, :n].copy_(L.keys[0])
                    dec.vcache[i][:, :n].copy_(L.values[0])
                self.dyn_owner = weakref.ref(cache)
            def set_inputs():
                dec.tok.copy_(input_ids.reshape(1))
                dec.pos.fill_(n)
            logits = self._step_checked(dec, set_inputs)
            for i, L in enumerate(layers):              # DynamicLayer.update()'s result: tensors one row longer
                L.keys = dec.kcache[i][None, :, :n + 1]
                L.values = dec.vcache[i][None, :, :n + 1]
        self.dyn_len = n + 1
        self.fast_steps += 1
        return logits.reshape(1, 1, -1)

    @torch.compiler.disable      # (an opaque eager call inside a torch.compile'd generate loop: the decoder owns its launches)
    def _fast_step(self, input_ids, cache, kind_layers, attention_mask=None, position_ids=None):
        kind, layers = kind_layers
        if kind == "dynamic":
            return self._dynamic_step(input_ids, cache, layers, attention_mask, position_ids)
        # every prompt pass through this wrapper leaves its verdict on the cache object (left padding / positions not from
        # zero): a cache that is reset and used again with a padded prompt goes to the stock forward
        if getattr(cache, "_quip_padded", None) is True:
            return None
        seen = getattr(self, "_static_checked", None)
        if seen is None or seen() is not cache:                      # a cache object seen for the first time
            import weakref
            if not self._unpadded(attention_mask, position_ids, layers[0].cumulative_length):
                return None
            self._static_checked = weakref.ref(cache)
        dec = self._decoder([L.keys for L in layers], [L.values for L in layers])
        if dec is None:
            return None
        return self._static_step(input_ids, [L.keys for L in layers], [L.values for L in layers],
                                 [L.cumulative_length for L in layers], dec)

    def _static_step(self, input_ids, keys4, values4, lens, dec=None):
        """the body of quip_lib::hf_decode_step: one token on the cache tensors, lengths advanced (StaticLayer.update's bookkeeping)"""
        if dec is None:                                  # (the compiled operator's entry; _fast_step hands its decoder over)
            dec = self._decoder(keys4, values4)
        if dec is None:
            raise RuntimeError("fast decode is not available for this model: " + str(self.disabled))
        with torch.no_grad():
            def set_inputs():
                dec.tok.copy_(input_ids.reshape(1))
                dec.pos.copy_(lens[0].reshape(1))
            logits = self._step_checked(dec, set_inputs)   # (1, vocab) fp16; row pos of every layer's cache written
            torch._foreach_add_(list(lens), 1)
        self.fast_steps += 1
        return logits.reshape(1, 1, -1)

    def __call__(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                 labels=None, use_cache=None, logits_to_keep=0, **kw):
        layers = self._eligible(input_ids, past_key_values, inputs_embeds, labels, kw)
        if layers is not None and torch.compiler.is_compiling():
            # inside somebody's torch.compile: static caches go through the operator (traceable; padding is not checked
            # there -- it would be a graph break); anything else is the stock forward, which compiles as it always did
            kind, ls = layers
            # (a mask's VALUES cannot be looked at in a trace.  The prompt pass -- eager, HF never compiles it -- has looked and
            #  left its finding on the cache object (_quip_padded, below): a plain attribute, so the trace is guarded on it and a
            #  later generation with the other finding gets its own graph.  No finding = stock forward, unless
            #  QUIP_FAST_DECODE_ASSUME_UNPADDED=1 vouches for unpadded sequences; the reference's loop passes no mask at all.)
            if kind == "static" and self.disabled is None and (
                    attention_mask is None or _ASSUME_UNPADDED or getattr(past_key_values, "_quip_padded", True) is False):
                logits = torch.ops.quip_lib.hf_decode_step(input_ids, [L.keys for L in ls], [L.values for L in ls],
                                                           [L.cumulative_length for L in ls], self.handle)
            else:
                logits = None
        else:
            if (layers is None and past_key_values is not None and input_ids is not None and input_ids.dim() == 2
                    and input_ids.shape[0] == 1 and input_ids.shape[1] > 1 and not torch.compiler.is_compiling()):
                # a prompt pass of a single sequence: does its mask have holes?  (one synchronisation per generation)
                try:
                    seen = past_key_values.get_seq_length()
                    past_key_values._quip_padded = _mask_has_holes(attention_mask, int(seen) + input_ids.shape[1])
                except Exception:       # noqa: BLE001 (an object that takes no attributes / an unknown cache: no finding)
                    pass
            logits = self._fast_step(input_ids, past_key_values, layers, attention_mask, position_ids) if layers is not None else None
            if (logits is None and layers is None and _PREFILL and past_key_values is not None and input_ids is not None
                    and input_ids.dim() == 2 and input_ids.shape[0] == 1 and input_ids.shape[1] > 1 and input_ids.is_cuda
                    and inputs_embeds is None and labels is None and isinstance(logits_to_keep, int) and logits_to_keep == 1
                    and not self.model.training and not kw.get("output_attentions") and not kw.get("output_hidden_states")
                    and not torch.compiler.is_compiling() and self._positions_from_zero(position_ids, input_ids.shape[1])):
                logits = self._dynamic_prefill(input_ids, past_key_values, attention_mask, kw)
        if logits is None:
            return self.orig_forward(input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                     past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels,
                                     use_cache=use_cache, logits_to_keep=logits_to_keep, **kw)
        if kw.get("return_dict", getattr(self.model.config, "return_dict", True)) is False:
            return (logits, past_key_values)
        from transformers.modeling_outputs import CausalLMOutputWithPast
        return CausalLMOutputWithPast(loss=None, logits=logits, past_key_values=past_key_values)


def enable_fast_decode(model, assume_llama_like=False):
    """wrap `model.forward` (see the module docstring); returns the model.  `disable_fast_decode` undoes it."""
    if getattr(model, "_quip_fast_decode", None) is None:
        fd = _FastDecode(model, assume_llama_like)
        model._quip_fast_decode = fd
        model.forward = fd
    return model


def disable_fast_decode(model):
    fd = getattr(model, "_quip_fast_decode", None)
    if fd is not None:
        model.forward = fd.orig_forward
        model._quip_fast_decode = None
    return model
