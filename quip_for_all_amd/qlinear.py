"""QuantLinear with the reference's constructor, buffers, state-dict keys and
forward contract (qlinear.py:8-159); the eval forward runs on the HIP kernels:

    xh  = wscale * U_in(had_left^T)(SU (.) x)          1 launch  (quip_lib::had_transform[_planes])
    z   = decode(Qidxs) @ xh                           1 launch  (codebook mm / decompress+GEMM)
    y   = SV (.) U_out(had_right)(Wscale (.) z)[:out] + bias   1 launch

instead of the reference's >= 5 launches (mul, pad, FHT, [hadK.T.contiguous, matmul],
mm, [mul], FHT, [matmul], slice, mul, add).  Intermediates stay fp16 between the
three launches exactly where the reference materialises fp16 tensors; inside a
launch the arithmetic is fp32.
"""
import math
import os

import torch
import torch.nn as nn

from .quant import get_hadK, matmul_hadU_cuda


class QuantLinear(nn.Module):
    # forwards with 2 .. skinny_max_rows rows take the small-batch kernels (rows mode of the matrix-core GEMV, the
    # single-pass skinny kernel: see skinny_exact below); more rows go to the codebook's batched product (E8P12: chunked
    # skinny kernel / fused dequant MFMA GEMM; the other codebooks: decompress + dense GEMM)
    skinny_max_rows = int(os.environ.get("QUIP_SKINNY_MAX_ROWS", "31"))
    # 1 < M < 32 rows.  As long as ONE pass of rows mode carries them (5 rows for k <= 4096, 3 for k <= 8192, ...) the
    # exact integer path is used: every row bit identical to its bs=1 result.  More rows: E8P12 takes the single-pass
    # fp16-MFMA skinny kernel (csrc/e8p_skinny_gemm.hip: the reference's arithmetic -- fp16 x fp16 -> fp32 -- not bit
    # identical to bs=1); skinny_exact = True (QUIP_SKINNY_EXACT=1) keeps the exact multi-pass path for everything.
    skinny_exact = os.environ.get("QUIP_SKINNY_EXACT", "0") != "0"
    # reference_ops = True (QUIP_REFERENCE_OPS=1): the forward issues the reference's op sequence only -- fp16 Hadamard
    # output, `*_mm_origorder` (M < 32: decode + fp32 FMA on exactly the reference's weights, RVQ: main + s * resid
    # rounded to fp16 per weight as in origin_order.cu:330-385) or decompress + dense GEMM -- instead of the digit-plane
    # GEMV / skinny / fused-GEMM kernels.  For validating a checkpoint against the reference's numerics.
    reference_ops = os.environ.get("QUIP_REFERENCE_OPS", "0") != "0"

    def __init__(self, in_features, out_features, codebook, bias=True, use_rand=True,
                 per_channel=False, weight_dtype=torch.float16):
        super().__init__()
        # peft looks for infeatures / outfeatures (qlinear.py:20-23)
        self.in_features = self.infeatures = in_features
        self.out_features = self.outfeatures = out_features
        self.codebook = codebook
        self.use_rand = use_rand
        self.per_channel = per_channel
        self.weight_dtype = weight_dtype

        had_left, self.K_left, self.q_in_features = get_hadK(in_features, use_rand)
        had_right, self.K_right, self.q_out_features = get_hadK(out_features, use_rand)
        if had_left is not None:
            self.register_buffer("had_left", had_left.to(weight_dtype), persistent=use_rand)
        else:
            self.had_left = None
        if had_right is not None:
            self.register_buffer("had_right", had_right.to(weight_dtype), persistent=use_rand)
        else:
            self.had_right = None

        if codebook.pack_out:
            qshape = (self.q_out_features // codebook.packsz, self.q_in_features // codebook.codesz)
        else:
            qshape = (self.q_out_features,
                      int(self.q_in_features // (codebook.codesz * codebook.packsz)))
        self.register_buffer("Qidxs", torch.zeros(*qshape, dtype=codebook.idx_dtype))
        self.SU = nn.Parameter(torch.ones(in_features, dtype=weight_dtype), requires_grad=True)
        self.SV = nn.Parameter(torch.ones(out_features, dtype=weight_dtype), requires_grad=True)
        if per_channel:
            self.register_buffer("Wscale", torch.ones(self.q_out_features, dtype=weight_dtype))
        else:
            self.register_buffer("Wscale", torch.ones((), dtype=torch.float))
        self.wscale_float = 1.0
        # HF probes `.weight` for device / dtype: keep the reference's 0-dim stand-in
        self.register_buffer("weight", torch.zeros((), dtype=weight_dtype))
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=weight_dtype))
        else:
            self.bias = None

    # ------------------------------------------------------------------ forward
    def _rows_per_pass(self):
        """activation rows one exact rows-mode pass carries for this layer's shape (0: rows mode unavailable)"""
        r = getattr(self, "_rpp", None)
        if r is None:
            from . import capi
            cb = self.codebook
            r = 0
            if hasattr(cb, "mm_planes_rows") and cb.planes_supported(self.q_out_features, self.q_in_features):
                r = int(capi.lib().quip_e8p_gemv_max_rows(self.q_out_features, self.q_in_features)) if cb.id == "E8P12" else 5
            self._rpp = r
        return r

    def regime(self, M):
        """which product path a forward of M activation rows takes (the dispatch of forward_fused, as a name):
        reference_ops | gemv_planes (M = 1, digit planes + matrix-core GEMV) | rows_exact (2 .. one rows-mode pass:
        bit identical to bs = 1) | skinny_fp16 (single-pass fp16 MFMA skinny kernel) | codebook (the codebook's own
        product: generic mm below its threshold, batched path beyond -- see codebook.batched_regime)"""
        cb = self.codebook
        if self.reference_ops:
            return "reference_ops"
        L_in = self.q_in_features // self.K_left
        # (the transform that writes digit planes is the register-blocked one: L >= 256, or K > 1 and L >= 64 -- hadamard.hip)
        planes_ok = (hasattr(cb, "mm_planes") and cb.planes_supported(self.q_out_features, self.q_in_features)
                     and (L_in >= 256 or (self.K_left > 1 and L_in >= 64)))
        if M == 1 and planes_ok:
            return "gemv_planes"
        if (2 <= M <= self.skinny_max_rows and not self.skinny_exact and hasattr(cb, "mm_skinny")
                and cb.skinny_supported(M, self.q_out_features, self.q_in_features) and M > self._rows_per_pass()):
            return "skinny_fp16"
        if 2 <= M <= self.skinny_max_rows and hasattr(cb, "mm_planes_rows") and planes_ok:
            return "rows_exact"
        return "codebook"

    def _had(self, name):
        h = getattr(self, name)
        if h is not None and (h.dtype != torch.float16 or not h.is_contiguous()):
            h = h.to(torch.float16).contiguous()
        return h

    @staticmethod
    def _vec(v):
        if v is None:
            return None
        v = v.detach() if isinstance(v, nn.Parameter) else v
        return v if v.dtype == torch.float16 else v.to(torch.float16)

    def forward(self, input):
        return self.forward_fused(input)

    def forward_fused(self, input, rms_weight=None, rms_eps=1e-5, gate=None, residual=None):
        """forward() with optional decoder-block glue folded into the two Hadamard launches:
        input := RMSNorm(input) * rms_weight (before q/k/v/gate/up), input := silu(gate) * input
        (before down_proj), output += residual (after o_proj / down_proj).  With all of them
        None this is exactly QuantLinear.forward of the reference (qlinear.py:87-115)."""
        fused = rms_weight is not None or gate is not None or residual is not None
        if self.training:
            assert not fused
            return self._forward_dense(input)
        if input.shape[-1] != self.in_features:     # the reference fails in `x * self.SU` (qlinear.py:90-91)
            raise RuntimeError(f"QuantLinear: input has {input.shape[-1]} features, expected in_features = "
                               f"{self.in_features}")
        x = input.reshape(-1, input.shape[-1])
        if x.shape[0] == 0:          # an empty batch: nothing to launch (the eager ops of the reference return empty too)
            return input.new_empty((*input.shape[:-1], self.out_features))
        x_dtype = x.dtype
        if x_dtype != torch.float16:
            x = x.to(torch.float16)
        L_in = self.q_in_features // self.K_left
        cb = self.codebook
        regime = self.regime(x.shape[0])
        if regime == "reference_ops":
            xh = torch.ops.quip_lib.had_transform_fused(
                x, self.q_in_features, self.q_in_features, self.K_left, self._had("had_left"), True,
                self._vec(self.SU), None, None, None, self.wscale_float / math.sqrt(L_in), None,
                self._vec(rms_weight), rms_eps, None if gate is None else gate.reshape(x.shape).to(torch.float16))
            z = cb.forward_reference(xh, self.Qidxs)
        elif regime == "gemv_planes":
            # bs=1 decode: transform straight into the GEMV's int8 digit planes (no fp16 xh)
            planes = torch.ops.quip_lib.had_transform_planes_fused(
                x, self.q_in_features, self.K_left, self._had("had_left"), True, self._vec(self.SU),
                self.wscale_float / math.sqrt(L_in), self._vec(rms_weight), rms_eps,
                None if gate is None else gate.reshape(x.shape).to(torch.float16),
                getattr(cb, "planes_resid_scale", 0.0))
            z = cb.mm_planes(planes, self.Qidxs)
        elif regime == "skinny_fp16":
            # more rows than one exact pass carries: single-pass skinny product on fp16 activations
            xh = torch.ops.quip_lib.had_transform_fused(
                x, self.q_in_features, self.q_in_features, self.K_left, self._had("had_left"), True,
                self._vec(self.SU), None, None, None, self.wscale_float / math.sqrt(L_in), None,
                self._vec(rms_weight), rms_eps, None if gate is None else gate.reshape(x.shape).to(torch.float16))
            z = cb.mm_skinny(xh, self.Qidxs)
        elif regime == "rows_exact":
            # skinny GEMM on the matrix cores (E8P12, E8P12RVQ4B): every row gets its own digit planes (one
            # transform launch); the GEMV's MFMA carries (row, plane) pairs in its 16 A rows, so up to 5 rows
            # share ONE pass over the codes (more rows / longer k: several passes) -- exact integer
            # arithmetic, every row bit identical to its bs=1 result
            planes = torch.ops.quip_lib.had_transform_planes_rows(
                x, self.q_in_features, self.K_left, self._had("had_left"), True, self._vec(self.SU),
                self.wscale_float / math.sqrt(L_in), self._vec(rms_weight), rms_eps,
                None if gate is None else gate.reshape(x.shape).to(torch.float16).contiguous(),
                getattr(cb, "planes_resid_scale", 0.0))
            z = cb.mm_planes_rows(planes, self.Qidxs)
        else:     # "codebook": the codebook's own product (M < 32: generic mm; beyond: its batched path)
            xh = torch.ops.quip_lib.had_transform_fused(
                x, self.q_in_features, self.q_in_features, self.K_left, self._had("had_left"), True,
                self._vec(self.SU), None, None, None, self.wscale_float / math.sqrt(L_in), None,
                self._vec(rms_weight), rms_eps, None if gate is None else gate.reshape(x.shape).to(torch.float16))
            z = cb(xh, self.Qidxs)
        L_out = self.q_out_features // self.K_right
        y = torch.ops.quip_lib.had_transform_fused(
            z, self.out_features, self.q_out_features, self.K_right, self._had("had_right"), False,
            None, self._vec(self.Wscale) if self.per_channel else None, self._vec(self.SV),
            self._vec(self.bias), 1.0 / math.sqrt(L_out),
            None if residual is None else residual.reshape(-1, self.out_features).to(torch.float16), None, rms_eps,
            None)
        if x_dtype != torch.float16:
            y = y.to(x_dtype)
        return y.view(*input.shape[:-1], self.out_features)

    def _forward_dense(self, input):
        """training-mode branch: x @ calc_weight() (qlinear.py:93-97)."""
        x = input.reshape(-1, input.shape[-1])
        if self.SU is not None:
            x = x * self.SU
        if x.shape[-1] != self.q_in_features:
            x = torch.nn.functional.pad(x, (0, self.q_in_features - x.shape[-1]))
        W = self.W if hasattr(self, "W") else self.calc_weight(cache=False).to(x.dtype)
        out = (x @ W)[..., :self.out_features]
        if self.SV is not None:
            out = out * self.SV
        out = out.view(*input.shape[:-1], out.shape[-1])
        return out + self.bias if self.bias is not None else out

    @torch.no_grad()
    def calc_weight(self, cache=True):
        """dense (q_in, q_out) weight = U_R(U_L(decode)^T) (qlinear.py:144-159)."""
        weight = self.codebook.decompress_weight(self.Qidxs)
        wscale_float = self.Wscale.mean().float().item()
        t = matmul_hadU_cuda(weight, self.had_left, self.K_left, self.q_in_features, wscale_float)
        W = matmul_hadU_cuda(t.T.contiguous(), self.had_right, self.K_right,
                             self.q_out_features).to(self.weight_dtype)
        if self.per_channel:
            W = W * self.Wscale / self.Wscale.mean()
        if cache:
            self.register_buffer("W", W, persistent=False)
        return W

    def pack(self, linear, attr):
        """fill the buffers from a quantiser result dict (qlinear.py:117-142)."""
        scaleWH, SU, SV = attr["scaleWH"], attr.get("SU"), attr.get("SV")
        if attr["merge_su"] and scaleWH is None:
            self.SU = None
        else:
            su = scaleWH if attr["merge_su"] else (SU if scaleWH is None else SU * scaleWH)
            self.SU.data.copy_(su)
        if attr["merge_sv"]:
            self.SV = None
        else:
            self.SV.data.copy_(SV)
        self.Qidxs.copy_(attr["Qidxs"])
        self.Wscale.copy_(attr["w_scale"].squeeze() if self.per_channel else attr["w_scale"])
        for name, key in (("had_left", "left_hadK"), ("had_right", "right_hadK")):
            if attr[key] is not None:
                getattr(self, name).copy_(attr[key])
        if linear.bias is not None:
            self.bias.copy_(linear.bias / SV if attr["merge_sv"] else linear.bias)

    # ------------------------------------------------------------------ test / bench helper
    @classmethod
    def from_params(cls, P):
        """Build a layer from a plain-attribute parameter record (numpy arrays named as
        the state-dict keys); used by tests, smoke() and bench to share seeded inputs."""
        from .codebook import codebook_id
        kw = {}
        if P.codebook.startswith("E8P12RVQ"):
            kw["opt_resid_scale"] = P.resid_scale
        cb = codebook_id[P.codebook](inference=True, **kw)
        # use_rand=True draws random had matrices only to size the buffers; overwritten below
        layer = cls(P.in_features, P.out_features, cb, bias=P.bias is not None, use_rand=True,
                    per_channel=P.per_channel)
        with torch.no_grad():
            layer.Qidxs.copy_(torch.from_numpy(P.Qidxs))
            if P.SU is None:
                layer.SU = None
            else:
                layer.SU.copy_(torch.from_numpy(P.SU))
            if P.SV is None:
                layer.SV = None
            else:
                layer.SV.copy_(torch.from_numpy(P.SV))
            import numpy as np
            layer.Wscale.copy_(torch.from_numpy(np.asarray(P.Wscale)))
            if P.had_left is not None:
                layer.had_left.copy_(torch.from_numpy(P.had_left))
            if P.had_right is not None:
                layer.had_right.copy_(torch.from_numpy(P.had_right))
            if P.bias is not None:
                layer.bias.copy_(torch.from_numpy(P.bias))
        layer.wscale_float = float(P.wscale_float)
        return layer


# A grouped GEMV launch pays launch / table build / pipeline fill / drain once.  The first kernel needs room for
# every x vector in LDS (fewer table copies -> bank conflicts), and two 59 MB matrices lost there (70B gate/up: 33.4 us
# grouped vs 2 x 14.5-15.4 us); the K-splitting kernel the dispatcher picks for launches >= 16-20 MB keeps full
# tables: 70B gate/up grouped 27.3 us = 0.54 of 8 TB/s vs 2 x 15.3 us, q/k/v 11.0 vs 12.3 us (tools/gemv_v2_bench.py
# --groups), so groups are only cut off far above any Llama block.
_GROUP_MAX_BYTES = int(os.environ.get("QUIP_GROUP_MAX_MB", "256")) << 20


def _gemv_planes_grouped(layers, planes):
    cb = layers[0].codebook
    nbytes = sum(l.Qidxs.numel() * l.Qidxs.element_size() for l in layers)
    if len(layers) > 1 and nbytes <= _GROUP_MAX_BYTES:
        return cb.mm_planes_group(planes, [l.Qidxs for l in layers])
    return [cb.mm_planes(p, l.Qidxs) for l, p in zip(layers, planes)]


def forward_group(layers, input, rms_weight=None, rms_eps=1e-5, residual=None):
    """[l(input) for l in layers] for 1..3 QuantLinear modules that read the same bs=1 activation
    (q/k/v_proj, gate/up_proj), each of the three stages issued as ONE launch for the whole group
    instead of one per module; results are identical to calling the modules one by one
    (qlinear.py:87-115 per module).  Falls back to per-module calls whenever the group does not
    qualify (different input transform shapes, batch > 1, non-E8P12 codebook)."""
    x = input.reshape(-1, input.shape[-1])
    l0 = layers[0]
    cb = l0.codebook
    residual = residual if residual is not None else [None] * len(layers)
    same_in = (1 < len(layers) <= 3 and x.dtype == torch.float16 and not l0.training and x.shape[0] >= 1
               and all(type(l.codebook) is type(cb) and l.in_features == l0.in_features
                       and l.q_in_features == l0.q_in_features and l.K_left == l0.K_left for l in layers))
    planes_ok = (same_in and x.shape[0] == 1 and hasattr(cb, "mm_planes")
                 and all(cb.planes_supported(l.q_out_features, l.q_in_features) for l in layers)
                 and cb.planes_group_supported([l.q_out_features for l in layers], l0.q_in_features))
    skinny = (same_in and 2 <= x.shape[0] <= l0.skinny_max_rows and hasattr(cb, "mm_planes_rows")
              and all(cb.planes_supported(l.q_out_features, l.q_in_features) for l in layers))
    if not same_in or skinny:   # skinny batches: every module takes its rows-mode path (planes per row)
        return [l.forward_fused(input, rms_weight=rms_weight, rms_eps=rms_eps, residual=r)
                for l, r in zip(layers, residual)]
    L_in = l0.q_in_features // l0.K_left
    if planes_ok:
        planes = torch.ops.quip_lib.had_transform_planes_group(
            x, l0.q_in_features, l0.K_left, [l._had("had_left") for l in layers], True,
            [l._vec(l.SU) for l in layers], [l.wscale_float / math.sqrt(L_in) for l in layers],
            l0._vec(rms_weight), rms_eps, None, getattr(cb, "planes_resid_scale", 0.0))
        zs = _gemv_planes_grouped(layers, planes)
    else:
        # any codebook / any batch: grouped fp16 input transforms, one codebook product per module
        n = len(layers)
        xh = torch.ops.quip_lib.had_transform_group(
            [x] * n, [l0.q_in_features] * n, l0.q_in_features, l0.K_left, [l._had("had_left") for l in layers], True,
            [None] * n, [None] * n, [None] * n, [l.wscale_float / math.sqrt(L_in) for l in layers], [None] * n,
            [l._vec(l.SU) for l in layers], l0._vec(rms_weight), rms_eps)
        zs = [l.codebook(h, l.Qidxs) for l, h in zip(layers, xh)]
    # output transforms: one launch per set of modules with the same (q_out, K_right)
    ys = [None] * len(layers)
    todo = list(range(len(layers)))
    while todo:
        i0 = todo[0]
        same = [i for i in todo if layers[i].q_out_features == layers[i0].q_out_features
                and layers[i].K_right == layers[i0].K_right]
        todo = [i for i in todo if i not in same]
        ls = [layers[i] for i in same]
        L_out = ls[0].q_out_features // ls[0].K_right
        outs = torch.ops.quip_lib.had_transform_group(
            [zs[i] for i in same], [l.out_features for l in ls], ls[0].q_out_features, ls[0].K_right,
            [l._had("had_right") for l in ls], False, [l._vec(l.Wscale) if l.per_channel else None for l in ls],
            [l._vec(l.SV) for l in ls], [l._vec(l.bias) for l in ls], [1.0 / math.sqrt(L_out)] * len(ls),
            [None if residual[i] is None else residual[i].reshape(x.shape[0], -1).to(torch.float16) for i in same],
            [None] * len(ls), None, rms_eps)
        for i, o in zip(same, outs):
            ys[i] = o.view(*input.shape[:-1], layers[i].out_features)
    return ys


# ---- stage-wise execution for the bs=1 decode step ---------------------------------------------
# QuantLinear.forward = input transform -> GEMV -> output transform.  The decode loop runs the
# stages of neighbouring modules together: the GEMV launch of a group computes its own input
# transform (and, if given, the output transform + residual add of the module that produced its
# input) in its prologue; the output transform of the group is one more launch.

def _pow2(n):
    return n & (n - 1) == 0


def fused_in_supported(layers, prev=None):
    """can `gemv_fused` run these modules (and fold `prev`'s output side in)?"""
    l0 = layers[0]
    cb = l0.codebook
    ok = (hasattr(cb, "fused_supported") and cb.fused_supported(len(layers), l0.q_in_features)
          and all(type(l.codebook) is type(cb) and not l.training and l.K_left == 1 and l.SU is not None
                  and l.in_features == l.q_in_features == l0.q_in_features for l in layers))
    if ok and prev is not None:
        ok = (prev.K_right == 1 and prev.out_features == prev.q_out_features == l0.in_features
              and prev.bias is None and not prev.per_channel and prev.SV is not None)
    return ok


def gemv_fused(layers, x=None, prev=None, z=None, residual=None, rms_weight=None, rms_eps=1e-5):
    """One launch: input side + GEMV of 1..3 modules reading the same bs=1 activation.
    Either `x` (1, k) is the activation, or `z` is the raw GEMV output of module `prev` whose output
    transform (+ `residual`) is folded in.  Returns (h, [z_i]): h = prev's finished output (None
    when x was given), z_i = raw GEMV outputs, to be finished by `out_transform_group`."""
    l0 = layers[0]
    k = l0.q_in_features
    res = torch.ops.quip_lib.e8p_gemv_fused(
        None if z is not None else x.reshape(1, k), None if z is None else z.reshape(1, k),
        None if z is None else prev._vec(prev.SV), None if residual is None else residual.reshape(1, k),
        None if rms_weight is None else l0._vec(rms_weight), rms_eps,
        1.0 if z is None else 1.0 / math.sqrt(prev.q_out_features // prev.K_right),
        [l._vec(l.SU) for l in layers], [l.wscale_float / math.sqrt(k) for l in layers],
        [l.Qidxs for l in layers], l0.codebook.grid_packed_abs)
    if z is None:
        return None, list(res)
    return res[0], list(res[1:])


_MIXED_OUT = os.environ.get("QUIP_MIXED_OUT", "1") != "0"     # A/B switch for the mixed-width output launch


def out_transform_group(layers, zs, residual=None):
    """output side (qlinear.py:106-114) of 1..3 modules from their raw GEMV outputs; one launch per set of
    modules with the same (q_out, K_right) -- and ONE launch for power-of-two widths that differ (q_proj next
    to the narrower k / v_proj of a grouped-query model)"""
    residual = residual if residual is not None else [None] * len(layers)
    ys = [None] * len(layers)
    todo = list(range(len(layers)))

    def mixable(l):
        return (_MIXED_OUT and l.K_right == 1 and 256 <= l.q_out_features <= 16384 and _pow2(l.q_out_features)
                and not l.per_channel)
    while todo:
        i0 = todo[0]
        same = [i for i in todo if layers[i].q_out_features == layers[i0].q_out_features
                and layers[i].K_right == layers[i0].K_right]
        ns = None
        if mixable(layers[i0]) and any(mixable(layers[i]) and i not in same for i in todo):
            same = [i for i in todo if mixable(layers[i])]
            ns = [layers[i].q_out_features for i in same]
        todo = [i for i in todo if i not in same]
        ls = [layers[i] for i in same]
        L_out = ls[0].q_out_features // ls[0].K_right
        outs = torch.ops.quip_lib.had_transform_group(
            [zs[i] for i in same], [l.out_features for l in ls], ls[0].q_out_features, ls[0].K_right,
            [l._had("had_right") for l in ls], False, [l._vec(l.Wscale) if l.per_channel else None for l in ls],
            [l._vec(l.SV) for l in ls], [l._vec(l.bias) for l in ls],
            [1.0 / math.sqrt(l.q_out_features // l.K_right) for l in ls],
            [None if residual[i] is None else residual[i].reshape(1, -1).to(torch.float16) for i in same],
            [None] * len(ls), None, 1e-5, ns)
        for i, o in zip(same, outs):
            ys[i] = o
    return ys


def gemv_group_unfused(layers, x, rms_weight=None, rms_eps=1e-5):
    """raw GEMV outputs of 1..3 modules reading x: grouped input-transform launch + grouped GEMV"""
    l0 = layers[0]
    L_in = l0.q_in_features // l0.K_left
    planes = torch.ops.quip_lib.had_transform_planes_group(
        x.reshape(1, -1), l0.q_in_features, l0.K_left, [l._had("had_left") for l in layers], True,
        [l._vec(l.SU) for l in layers], [l.wscale_float / math.sqrt(L_in) for l in layers],
        None if rms_weight is None else l0._vec(rms_weight), rms_eps, None,
        getattr(l0.codebook, "planes_resid_scale", 0.0))
    return _gemv_planes_grouped(layers, list(planes))


def gemv_unfused(layer, x, gate=None, rms_weight=None, rms_eps=1e-5):
    """raw GEMV output of one module through the separate input-transform launch (any K_left)"""
    L_in = layer.q_in_features // layer.K_left
    planes = torch.ops.quip_lib.had_transform_planes_fused(
        x.reshape(1, -1), layer.q_in_features, layer.K_left, layer._had("had_left"), True, layer._vec(layer.SU),
        layer.wscale_float / math.sqrt(L_in), layer._vec(rms_weight), rms_eps,
        None if gate is None else gate.reshape(1, -1), getattr(layer.codebook, "planes_resid_scale", 0.0))
    return layer.codebook.mm_planes(planes, layer.Qidxs)


def chain_supported(layers, prev):
    """can `gemv_chain` fold `prev`'s output side into the input transforms of `layers`?"""
    l0 = layers[0]
    cb = l0.codebook
    n = l0.q_in_features
    return (hasattr(cb, "planes_group_supported") and 1 <= len(layers) <= 3 and _pow2(n) and 256 <= n <= 16384
            and all(type(l.codebook) is type(cb) and not l.training and l.K_left == 1 and l.SU is not None
                    and l.in_features == l.q_in_features == n
                    and cb.planes_supported(l.q_out_features, n) for l in layers)
            and cb.planes_group_supported([l.q_out_features for l in layers], n)
            and prev.K_right == 1 and prev.out_features == prev.q_out_features == n
            and prev.bias is None and not prev.per_channel and prev.SV is not None)


def gemv_chain(layers, prev, z, residual=None, rms_weight=None, rms_eps=1e-5):
    """Two launches: (1) one Hadamard launch with a workgroup per consumer module that first
    finishes the producer `prev` (output transform of its raw GEMV output `z`, + `residual`) and
    then runs its own input transform (RMSNorm, SU, Hadamard -> digit planes); (2) the grouped
    GEMV.  Returns (h, [z_i]) like gemv_fused."""
    l0 = layers[0]
    n = l0.q_in_features
    res = torch.ops.quip_lib.had_chain_planes_group(
        z.reshape(1, n), prev._vec(prev.SV), None if residual is None else residual.reshape(1, n),
        1.0 / math.sqrt(prev.q_out_features // prev.K_right), n, [l._vec(l.SU) for l in layers],
        [l.wscale_float / math.sqrt(n) for l in layers], None if rms_weight is None else l0._vec(rms_weight), rms_eps,
        getattr(l0.codebook, "planes_resid_scale", 0.0))
    h, planes = res[0], list(res[1:])
    return h, _gemv_planes_grouped(layers, planes)


# ---- persistent decode engine, stage 1: the MLP half of a block in one launch (csrc/decode_engine.hip) -----------
def ffn_engine_ok(gate, up, down):
    """can the engine run gate / up / down of a block?  E8P12 on all three, the same K x K factor size on the
    ffn side (K > 1, L = n_ffn / K a power of two <= 256), plain SV / SU sides, no padding, no bias on gate / up"""
    from .register_lib import ffn_engine_supported
    cb = gate.codebook
    n_ffn, hidden = gate.out_features, gate.in_features
    ok = (getattr(cb, "id", None) == "E8P12" and all(type(l.codebook) is type(cb) and not l.training and not l.per_channel
                                                     for l in (gate, up, down))
          and up.out_features == n_ffn and up.in_features == hidden and down.in_features == n_ffn
          and down.out_features == hidden
          and gate.q_out_features == n_ffn and up.q_out_features == n_ffn and down.q_in_features == n_ffn
          and gate.q_in_features == hidden and down.q_out_features == hidden
          and gate.K_right == up.K_right == down.K_left and gate.K_right > 1
          and gate.bias is None and up.bias is None
          and gate.SV is not None and up.SV is not None and down.SU is not None)
    return bool(ok and ffn_engine_supported(hidden, n_ffn, gate.K_right))


def _engine_had3(gate, up, down):
    """the three K x K factors as the engine reads them (cached on `down`): gate.had_right, up.had_right row major,
    each padded to K * K rounded up to 8 elements, then down.had_left transposed and zero padded to (KP16, KP16),
    KP16 = K rounded up to 16"""
    t = getattr(down, "_eng_had3", None)
    if t is None or t.device != down.Qidxs.device:
        K = gate.K_right
        kkp, kp16 = (K * K + 7) // 8 * 8, (K + 15) // 16 * 16
        t = torch.zeros(2 * kkp + kp16 * kp16, dtype=torch.float16, device=down.Qidxs.device)
        t[:K * K] = gate.had_right.detach().to(torch.float16).reshape(-1)
        t[kkp:kkp + K * K] = up.had_right.detach().to(torch.float16).reshape(-1)
        hdT = torch.zeros(kp16, kp16, dtype=torch.float16, device=t.device)
        hdT[:K, :K] = down.had_left.detach().to(torch.float16).T
        t[2 * kkp:] = hdT.reshape(-1)
        down._eng_had3 = t
    return t


def ffn_engine(gate, up, down, planes, workspace, dbg=None):
    """raw product of down_proj (1, hidden) from the digit planes of gate's / up's transformed input"""
    K = gate.K_right
    L = gate.q_out_features // K
    return torch.ops.quip_lib.ffn_engine(
        planes[0], planes[1], gate.Qidxs, up.Qidxs, down.Qidxs, _engine_had3(gate, up, down), gate._vec(gate.SV),
        up._vec(up.SV), down._vec(down.SU), gate.codebook.grid_packed_abs, workspace, 1.0 / math.sqrt(L),
        down.wscale_float / math.sqrt(L), K, dbg)


def chain_planes(layers, prev, z, residual=None, rms_weight=None, rms_eps=1e-5):
    """the Hadamard chain launch of `gemv_chain` alone: (h, [planes_i])"""
    l0 = layers[0]
    n = l0.q_in_features
    res = torch.ops.quip_lib.had_chain_planes_group(
        z.reshape(1, n), prev._vec(prev.SV), None if residual is None else residual.reshape(1, n),
        1.0 / math.sqrt(prev.q_out_features // prev.K_right), n, [l._vec(l.SU) for l in layers],
        [l.wscale_float / math.sqrt(n) for l in layers], None if rms_weight is None else l0._vec(rms_weight), rms_eps,
        getattr(l0.codebook, "planes_resid_scale", 0.0))
    return res[0], list(res[1:])
