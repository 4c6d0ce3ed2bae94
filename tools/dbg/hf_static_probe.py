import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from transformers import AutoModelForCausalLM
from tests.test_quantizer_host import _fill_random, _tiny_config
from quip_for_all_amd.quantizer import QuipQuantizer
from quip_for_all_amd.hf_static import HFStaticDecoder
torch.manual_seed(0)
model = AutoModelForCausalLM.from_config(_tiny_config(), dtype=torch.float16)
qz = QuipQuantizer(codebook="E8P12", inference=True, ft_epochs=0)
qz.convert_model(model)
_fill_random(model, seed=3)
model = model.to("cuda:0").eval()
ids = torch.tensor([[1, 17, 42, 99, 7, 250]], device="cuda:0")
res = {}
for mode in ("eager", "graph"):
    try:
        d = HFStaticDecoder(model, max_cache_len=64)
        toks, dt = d.generate(ids, 24, mode)
        res[mode] = toks.cpu().tolist()
        print(mode, "%.1f tok/s" % (23 / dt), res[mode][:12], flush=True)
    except Exception as e:
        import traceback; traceback.print_exc()
        print(mode, "FAILED", repr(e)[:300], flush=True)
ref = model.generate(ids, max_new_tokens=24, do_sample=False)[0, 6:].cpu().tolist()
print("hf generate", ref[:12])
print({m: r == ref for m, r in res.items()}, res["eager"] == res["graph"], [i for i in range(24) if res["eager"][i] != ref[i]][:3])
