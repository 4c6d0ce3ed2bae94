import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import quip_for_all_amd as Q
from quip_for_all_amd.quantizer import QuipQuantizer
for ors in (-1, None, 0.29):
    kw = {} if ors is None else {"opt_resid_scale": ors}
    cb = Q.codebook.codebook_id["E8P12RVQ4B"](inference=True, **kw)
    print("opt_resid_scale arg", ors, "->", getattr(cb, "opt_resid_scale", None))
    l = Q.QuantLinear(256, 256, cb, bias=False).cuda()
    l.Qidxs.copy_(torch.randint(-2**31, 2**31 - 1, l.Qidxs.shape, dtype=torch.int64).to(torch.int32))
    l.Wscale.fill_(0.02); l.wscale_float = 0.02
    W = l.calc_weight(cache=False)
    x = torch.randn(2, 256, device="cuda").half()
    print("  W finite", torch.isfinite(W).all().item(), W.abs().max().item(), " y finite", torch.isfinite(l(x)).all().item())
