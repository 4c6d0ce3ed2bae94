import sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from transformers import DynamicCache
from quip_for_all_amd import decode as D
from quip_for_all_amd.hf_fast import enable_fast_decode
# reuse bench's model builder by calling its extra with tiny token counts is heavy; build here
from tests.test_gpu_hf_generate import _random_quantized_hf_llama
from transformers import LlamaConfig
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                  num_key_value_heads=32, vocab_size=32000, max_position_embeddings=4096, rms_norm_eps=1e-5)
model = _random_quantized_hf_llama(cfg)
enable_fast_decode(model)
fd = model._quip_fast_decode
for P in (16, 128, 1024, 2048):
    ids = torch.randint(1, 32000, (1, P), device="cuda:0")
    res = {}
    for name, fwd in (("stock", fd.orig_forward), ("fast", fd)):
        model.forward = fwd
        ts = []
        for _ in range(4):
            c = DynamicCache(config=model.config)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with torch.no_grad():
                out = model(ids, past_key_values=c, use_cache=True, logits_to_keep=1)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        res[name] = min(ts[1:]) * 1e3
    print("prompt %5d tokens: stock %.1f ms, fast %.1f ms (fast prefills %d)" % (P, res["stock"], res["fast"], fd.fast_prefills), flush=True)
