"""A/B of two builds / table modes of the library on a persistent launch: run `python tools/dbg/gqa_ab.py OUT.pt [layers] [tokens]
[shape]` once per build (QUIP_LIB_PATH selects a library, QUIP_ENG_REP=24 the byte tables of the 4096-wide launches; shape: 70b
(default) | 7b | g8), then `python tools/dbg/gqa_ab.py --cmp A.pt B.pt`: logits and cache rows must be EQUAL bit for bit (the nibble
mode of round 6 computes the same integers as the byte tables of rounds 3-5)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    ok = True
    for k in a:
        same = torch.equal(a[k], b[k])
        ok = ok and same
        d = (a[k].float() - b[k].float()).abs().max().item() if a[k].dtype.is_floating_point else -1
        print(f"{k}: {'EQUAL' if same else 'DIFFERENT'} (max abs diff {d})")
    print("A/B:", "bit identical" if ok else "NOT identical")
    sys.exit(0 if ok else 1)

from quip_for_all_amd import decode as D  # noqa: E402

out = sys.argv[1]
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
tokens = int(sys.argv[3]) if len(sys.argv) > 3 else 4
which = sys.argv[4] if len(sys.argv) > 4 else "70b"
shape = {"70b": D.LlamaShape(hidden=8192, ffn=28672, layers=layers, heads=64, kv_heads=8, vocab=2048),
         "7b": D.LlamaShape(hidden=4096, ffn=11008, layers=layers, heads=32, kv_heads=32, vocab=2048),
         "g8": D.LlamaShape(hidden=4096, ffn=14336, layers=layers, heads=32, kv_heads=8, vocab=2048)}[which]
import numpy as np  # noqa: E402
np.random.seed(1234)          # (get_hadK(use_rand=True) draws the 7 x 7 factors from scipy's process-global generator)
torch.manual_seed(1234)
dec = D.LlamaDecoder(shape, "E8P12", max_len=64, device="cuda:0", seed=3, device_init=True)
assert dec.block_eng and dec.eng_shape == {"70b": 1, "7b": 0, "g8": 2}[which], (dec.block_eng, getattr(dec, "eng_shape", None))
dec.reset(first_token=7)
res = {}
with torch.no_grad():
    for t in range(tokens):
        lg = dec.step().clone()
        torch.cuda.synchronize()
        assert dec.engine_status() == 0, hex(dec.engine_status())
        res[f"logits{t}"] = lg.cpu()
res["kcache"] = dec.kcache[..., :tokens, :].cpu()
res["vcache"] = dec.vcache[..., :tokens, :].cpu()
torch.save(res, out)
print("saved", out, "finite", all(bool(torch.isfinite(v.float()).all()) for v in res.values()), "lib", os.environ.get("QUIP_LIB_PATH", "default"))
