"""The reference's own decode harness shape, for a model returned by `load_quantized_model` / `convert_model`: the stock HF
Llama forward on a `StaticCache`, one token per call, greedy -- eager, or with the single-token step captured in a hipGraph
(what the reference gets from `torch.compile(decode_one_tokens, mode="reduce-overhead", fullgraph=True)` on top of
`model._setup_cache(StaticCache, 1, max_cache_len=2048)`, example_generate.py:28-33, 62-70; `_setup_cache` is a
transformers-4.38 API, the cache object is passed explicitly here), or compiled exactly like that (mode "compile": the ops
carry fake implementations, and their process-lifetime K-split workspace is raw device memory so that the cudagraph
trees' pool never sees it).

This is the drop-in path a user of the reference lands on first: every projection is a `QuantLinear.forward` (bs = 1: one
transform launch, one GEMV launch, one transform launch), everything else is the framework's.  `LlamaDecoder`
(decode.py) is the fast path for the same checkpoint; `bench.py` reports both (`extras.hf_generate_static_cache`)."""
import time

import torch


class HFStaticDecoder:
    def __init__(self, model, max_cache_len=2048):
        from transformers import StaticCache
        self.model = model.eval()
        self.dev = next(model.parameters()).device
        self.max_cache_len = max_cache_len
        self.cache = StaticCache(config=model.config, max_cache_len=max_cache_len)
        self.tok = torch.zeros(1, 1, dtype=torch.long, device=self.dev)
        self.pos = torch.zeros(1, dtype=torch.long, device=self.dev)
        self.graph = None

    def _forward(self, ids, cache_position):
        return self.model(ids, past_key_values=self.cache, cache_position=cache_position, use_cache=True, return_dict=False)[0]

    @torch.no_grad()
    def decode_one_token(self, tok, cache_position):
        """logits of one token at `cache_position` -> greedy next token (1, 1)   [example_generate.py:28-33, greedy]"""
        return self._forward(tok, cache_position)[:, -1].argmax(-1, keepdim=True)

    @torch.no_grad()
    def prefill(self, prompt_ids):
        """the prompt in one eager pass (example_generate.py:44-47); returns the first generated token"""
        self.cache.reset()
        n = prompt_ids.shape[1]
        logits = self._forward(prompt_ids, torch.arange(n, device=self.dev))
        self.tok.copy_(logits[:, -1].argmax(-1, keepdim=True))
        self.pos.fill_(n)
        return self.tok.clone()

    def capture(self):
        """the single-token step as a hipGraph on static token / position tensors"""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):
                self.decode_one_token(self.tok, self.pos)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self._next = self.decode_one_token(self.tok, self.pos)
        torch.cuda.synchronize()

    def compile(self, fullgraph=True):
        """the reference's own call: torch.compile(decode_one_tokens, mode="reduce-overhead", fullgraph=True)
        (example_generate.py:69-70)"""
        self._compiled = torch.compile(self.decode_one_token, mode="reduce-overhead", fullgraph=fullgraph)

    @torch.no_grad()
    def generate(self, prompt_ids, max_new_tokens, mode="eager"):
        """greedy; mode: "eager" | "graph" (captured step) | "compile" (torch.compile, mode="reduce-overhead").
        Returns (tokens (max_new_tokens,), seconds spent in the decode loop, synchronised at both ends)"""
        prompt_ids = torch.as_tensor(prompt_ids, dtype=torch.long, device=self.dev).reshape(1, -1)
        assert prompt_ids.shape[1] + max_new_tokens <= self.max_cache_len
        if mode == "graph" and self.graph is None:
            self.prefill(prompt_ids)                   # (the warm-up steps need an initialised cache)
            self.capture()
        if mode == "compile" and getattr(self, "_compiled", None) is None:
            self.compile()
        out = torch.empty(max_new_tokens, dtype=torch.long, device=self.dev)
        out[0] = self.prefill(prompt_ids).reshape(-1)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(1, max_new_tokens):
            if mode == "graph":
                self.graph.replay()
                nxt = self._next
            elif mode == "compile":
                torch.compiler.cudagraph_mark_step_begin()
                nxt = self._compiled(self.tok.clone(), self.pos.clone()).clone()
            else:
                nxt = self.decode_one_token(self.tok, self.pos)
            self.tok.copy_(nxt)
            self.pos += 1
            out[t] = nxt.reshape(-1)[0]
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0
