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

    def _decoder(self, layers):
        from .decode import LlamaDecoder
        keys = [L.keys[0] for L in layers]
        values = [L.values[0] for L in layers]
        sig = tuple(t.data_ptr() for t in keys + values)
        max_len = layers[0].max_cache_len
        if self.dec is None or self.dec.max_len != max_len or self.dec.dev != keys[0].device:
            try:
                self.dec = LlamaDecoder.from_hf(self.model, max_len=max_len, assume_llama_like=self.assume_llama_like,
                                                kv_cache=(keys, values))
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
        if attention_mask is not None and attention_mask.dim() == 2 and not bool((attention_mask[:, :n + 1] != 0).all()):
            return False
        if position_ids is not None and position_ids.numel() >= 1 and int(position_ids.reshape(-1)[-1]) != int(n):
            return False
        return True

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

    def _dynamic_step(self, input_ids, cache, layers, attention_mask=None, position_ids=None):
        import weakref
        from .decode import LlamaDecoder
        n = layers[0].keys.shape[-2]
        dev = layers[0].keys.device
        if self.dyn is None or self.dyn.dev != dev:
            try:
                self.dyn = LlamaDecoder.from_hf(self.model, max_len=self._dyn_capacity(), assume_llama_like=self.assume_llama_like)
            except (NotImplementedError, TypeError, ValueError) as e:
                self.disabled = repr(e)
                return None
            self.dyn_owner = None
        dec = self.dyn
        owner = self.dyn_owner() if self.dyn_owner is not None else None
        owned = (owner is cache and n <= self.dyn_len and layers[0].keys.data_ptr() == dec.kcache[0].data_ptr()
                 and layers[-1].values.data_ptr() == dec.vcache[-1].data_ptr())
        if not owned and not self._unpadded(attention_mask, position_ids, n):
            return None
        with torch.no_grad():
            if not owned:
                if owner is not None and owner is not cache:
                    # the previous owner keeps its contents: its views of the buffers become tensors of its own
                    for L in getattr(owner, "layers", []):
                        if getattr(L, "is_initialized", False) and torch.is_tensor(L.keys):
                            L.keys, L.values = L.keys.clone(), L.values.clone()
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
        seen = getattr(self, "_static_checked", None)
        if seen is None or seen() is not cache:                      # a cache object seen for the first time
            import weakref
            if not self._unpadded(attention_mask, position_ids, layers[0].cumulative_length):
                return None
            self._static_checked = weakref.ref(cache)
        dec = self._decoder(layers)
        if dec is None:
            return None
        with torch.no_grad():
            lens = [L.cumulative_length for L in layers]

            def set_inputs():
                dec.tok.copy_(input_ids.reshape(1))
                dec.pos.copy_(lens[0].reshape(1))
            logits = self._step_checked(dec, set_inputs)   # (1, vocab) fp16; row pos of every layer's cache written
            torch._foreach_add_(lens, 1)                 # StaticLayer.update()'s bookkeeping
        self.fast_steps += 1
        return logits.reshape(1, 1, -1)

    def __call__(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                 labels=None, use_cache=None, logits_to_keep=0, **kw):
        layers = self._eligible(input_ids, past_key_values, inputs_embeds, labels, kw)
        logits = self._fast_step(input_ids, past_key_values, layers, attention_mask, position_ids) if layers is not None else None
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
