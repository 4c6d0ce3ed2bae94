"""Single-token decode steps of an HF model returned by `load_quantized_model`, routed through `LlamaDecoder`.

The reference's own decode loop (example_generate.py:28-33, 62-70) and HF `generate(cache_implementation="static")` call
`model(input_ids (1, 1), past_key_values=<StaticCache>, ...)` once per token; with the stock modules that is ~45 launches
per decoder block (three per QuantLinear plus the framework's element-wise kernels).  `enable_fast_decode(model)` wraps
`model.forward`: a call that IS such a step -- one token, batch 1, an initialised `StaticCache`, fp16, no extra outputs
asked for -- runs `LlamaDecoder.step()` on the SAME modules and on the cache object's own key / value tensors (row
`cumulative_length` is written, the lengths advance by one), i.e. the persistent block launch where the shape has one and
the fused stage-wise step elsewhere.  Every other call (the prompt, batches, dynamic caches, `output_attentions`,
`inputs_embeds`, training) goes to the original forward untouched, so the model stays a drop-in HF model.

Not checked (it would cost a device synchronisation per token): that `position_ids` equals the cache length and that
`attention_mask` has no holes -- true for an unpadded single sequence, which is what batch 1 generation passes.
Logits are those of `LlamaDecoder` (same arithmetic as the stage-wise ops, tests/test_gpu_hf_generate.py pins them against
the stock forward); the cache rows are bit-compatible (rotated keys, fp16), so fast and stock steps may be mixed freely."""
import torch


class _FastDecode:
    def __init__(self, model, assume_llama_like=False):
        self.model = model
        self.orig_forward = model.forward
        self.assume_llama_like = assume_llama_like
        self.dec = None
        self.bound = None            # the key / value data pointers the decoder currently uses
        self.disabled = None         # the reason LlamaDecoder refused this model, once known
        self.fast_steps = 0

    # -- eligibility: shapes and types only, nothing that reads device memory
    def _static_layers(self, cache):
        layers = getattr(cache, "layers", None)
        if not layers or len(layers) != self.model.config.num_hidden_layers:
            return None
        for L in layers:
            if (type(L).__name__ != "StaticLayer" or not getattr(L, "is_initialized", False) or L.keys.shape[0] != 1
                    or L.keys.dtype != torch.float16 or not L.keys.is_cuda):
                return None
        return layers

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

    @torch.compiler.disable      # (an opaque eager call inside a torch.compile'd generate loop: the decoder owns its launches)
    def _fast_step(self, input_ids, layers):
        dec = self._decoder(layers)
        if dec is None:
            return None
        with torch.no_grad():
            lens = [L.cumulative_length for L in layers]
            dec.tok.copy_(input_ids.reshape(1))
            dec.pos.copy_(lens[0].reshape(1))
            logits = dec.step()                          # (1, vocab) fp16; row pos of every layer's cache written
            torch._foreach_add_(lens, 1)                 # StaticLayer.update()'s bookkeeping
        self.fast_steps += 1
        return logits.reshape(1, 1, -1)

    def __call__(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                 labels=None, use_cache=None, logits_to_keep=0, **kw):
        layers = self._eligible(input_ids, past_key_values, inputs_embeds, labels, kw)
        logits = self._fast_step(input_ids, layers) if layers is not None else None
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
