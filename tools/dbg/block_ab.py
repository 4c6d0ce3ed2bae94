import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_block_engine as T
a = T._decoder(1, True)
b = T._decoder(1, False)
print("embed equal", torch.equal(a.embed, b.embed), "lm_head equal", torch.equal(a.lm_head, b.lm_head))
for k in ("q", "k", "v", "o", "gate", "up", "down"):
    ma, mb = a.layers[0][k], b.layers[0][k]
    print(k, torch.equal(ma.Qidxs, mb.Qidxs), torch.equal(ma.SU, mb.SU), torch.equal(ma.SV, mb.SV), ma.wscale_float == mb.wscale_float,
          (ma.had_left is None or torch.equal(ma.had_left, mb.had_left)), (ma.had_right is None or torch.equal(ma.had_right, mb.had_right)))
print("ln", torch.equal(a.layers[0]["ln1"], b.layers[0]["ln1"]), torch.equal(a.final_norm, b.final_norm))
a.reset(7); b.reset(7)
with torch.no_grad():
    la = a.step().clone(); lb = b.step().clone()
print("logits diff", (la.float() - lb.float()).abs().max().item(), a.tok, b.tok, a.engine_status())
