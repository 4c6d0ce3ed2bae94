"""Layer-level quantiser with the reference's class name and attr contract (quip.py:17-200): accumulate
the proxy Hessian of one nn.Linear from calibration inputs, incoherence-process (random signs + randomised
Hadamard on both sides), LDLQ-round against a codebook, and hand back what QuantLinear.pack() consumes.
Host logic in torch; the codebook search inside LDLQ is the HIP kernel behind cb.quantize().  The model-level
calibration pipeline of the reference (quantizer.py:250-715) stays out of scope."""
import math

import torch
import torch.nn as nn

from .quant import LDLQ, get_hadK, matmul_hadU, matmul_hadUt


class QUIP:
    def __init__(self, layer, cb):
        assert isinstance(layer, nn.Linear), "this build quantises nn.Linear layers"
        self.layer = layer
        self.dev = layer.weight.device
        self.rows, self.columns = layer.weight.shape
        self.H = torch.zeros((self.columns, self.columns), dtype=torch.float64, device=self.dev)
        self.mu = torch.zeros((self.columns,), dtype=torch.float64, device=self.dev)
        self.nsamples = 0
        self.cb = cb.to(self.dev)

    def add_batch(self, inp, out=None):
        """running mean of 2 x x^T over the calibration rows (quip.py:40-69)"""
        x = inp.reshape(-1, inp.shape[-1]).to(torch.float64).t()            # (columns, rows)
        batch = 1 if inp.dim() == 2 else inp.shape[0]
        keep = self.nsamples / (self.nsamples + batch)
        self.H *= keep
        self.mu *= keep
        self.nsamples += batch
        self.mu += x.sum(dim=1) / self.nsamples
        x = math.sqrt(2 / self.nsamples) * x
        self.H += x @ x.t()

    def quant(self, rescale_WH=False, use_fp64=False, sigma_reg=0.01, scale_override=0, use_buffered=True,
              use_rand=True, per_channel=False, quip_tune_iters=0):
        """returns the attr dict of quip.py:176-187 and leaves the de-rotated hatW in layer.weight (the free
        end-to-end golden: nn.Linear(hatW)(x) == QuantLinear(x), SURVEY 4 identity 3)"""
        H = self.H.clone() if use_fp64 else self.H.to(torch.float32)
        if self.nsamples == 0:       # no calibration data: identity proxy (plain nearest rounding + LDL = I)
            H = torch.eye(self.columns, dtype=H.dtype, device=self.dev)
        w = self.layer.weight.data.clone().to(H.dtype)
        dead = torch.diag(H) == 0
        H[dead, dead] = 1
        w[:, dead] = 0
        H = H / torch.diag(H).mean()
        scaleWH = None
        if rescale_WH:
            H = H / H.abs().max()
            dH = torch.diag(H).clamp(min=1e-8)
            dW = torch.diag(w.T @ w).clamp(min=1e-8)
            scaleWH = (dH / dW).sqrt().sqrt().to(torch.float32).clamp(min=1e-8)
            w = w * scaleWH[None, :]
            H = H / scaleWH[None, :] / scaleWH[:, None]
        merge_su, merge_sv = hasattr(self.layer, "SU"), hasattr(self.layer, "SV")
        rsign = lambda k: (torch.randn(k, device=self.dev).sign() + 1e-5).sign().to(H.dtype)   # noqa: E731
        SU = self.layer.SU.to(H.dtype) if merge_su else rsign(self.columns)
        SV = self.layer.SV.to(H.dtype) if merge_sv else rsign(self.rows)
        lhad, lK, lN = get_hadK(self.columns, use_rand=use_rand)
        rhad, rK, rN = get_hadK(self.rows, use_rand=use_rand)
        # incoherence processing: H <- U_L (SU H SU) U_L^T,  W <- U_R (SV W SU) U_L^T   (quip.py:124-128)
        H = matmul_hadUt(matmul_hadUt(H * SU, lhad, lK, lN).T * SU, lhad, lK, lN)
        w = matmul_hadUt(matmul_hadUt(w.T * SV, rhad, rK, rN).T * SU, lhad, lK, lN)
        diag = torch.arange(H.shape[0], device=self.dev)
        for attempt in range(10):
            H[diag, diag] += sigma_reg
            try:
                L = torch.linalg.cholesky(H)
                if not torch.isnan(L).any():
                    break
            except RuntimeError:
                pass
        else:
            raise ValueError("Hessian is not invertible")
        w_scale = w.square().mean(dim=1, keepdim=True).sqrt() if per_channel else w.square().mean().sqrt()
        w_scale = w_scale / (scale_override if scale_override > 0 else self.cb.opt_scale)
        hat_w, Qidxs = LDLQ(w / w_scale, H, L, self.cb, quip_tune_iters, buf_cols=128 if use_buffered else self.cb.codesz)
        hat_w = hat_w * w_scale
        # back to the original basis (quip.py:162-167)
        deq = (matmul_hadU((matmul_hadU(hat_w, lhad, lK, lN)[..., :self.columns] * SU).T, rhad, rK, rN)
               [..., :self.rows] * SV).T
        if rescale_WH:
            deq = deq / scaleWH[None, :]
        self.layer.weight.data = deq.reshape(self.layer.weight.shape).to(self.layer.weight.dtype)
        return {
            "left_hadK": lhad.cpu() if use_rand and lhad is not None else None,
            "right_hadK": rhad.cpu() if use_rand and rhad is not None else None,
            "Qidxs": self.cb.maybe_pack_idxs(Qidxs).cpu(),
            "w_scale": w_scale.cpu(),
            "SU": SU.cpu(), "SV": SV.cpu(),
            "merge_su": merge_su, "merge_sv": merge_sv,
            "scaleWH": scaleWH.cpu() if rescale_WH else None,
        }

    def free(self):
        self.H = self.mu = None
        torch.cuda.empty_cache()
