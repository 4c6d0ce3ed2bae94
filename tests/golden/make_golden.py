#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *reference's own
Python*, imported read-only from /root/reference in the authoring container.

Run:  python tests/golden/make_golden.py            (needs /root/reference)

Nothing of the reference is copied: the fixtures are data (inputs + expected
outputs).  The reference's CUDA extensions do not exist here, so two empty
stub modules are put in ``sys.modules`` to let ``register_lib`` import, and CPU
implementations of the ``quip_lib`` ops are registered *in this harness only*,
built from the reference's own functions:

  hadamard                    <- quant.matmul_hadU (pure-torch butterfly, quant.py:42-65)
  decompress_e8p_origorder    <- codebook.e8p12.get_full_grid LUT (e8p12.py:82-103)
  decompress_d4_origorder     <- D4_codebook.grid LUT (d4.py:89-96)
  decompress_hi_origorder     <- HI grid = nibble-7.5 (hi.py:9-12) after undoing maybe_pack_idxs
  decompress_e8prvq{3,4}      <- full-grid / e81b LUTs combined with ONE fp16 fma as the kernel
                                  does (origin_order.cu:330-331,378-380); the LUTs are the
                                  reference's, the fma is our reading of the kernel and is
                                  additionally pinned by the ``quantize()`` goldens below
  *_mm_origorder              <- x.float() @ decompress(...).float().T -> fp16

With those, the reference's own ``QuantLinear.forward`` / ``calc_weight`` run
on CPU and give the module-level goldens.
"""
import hashlib
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def _import_reference():
    import torch
    for name in ("fast_hadamard_transform_cuda", "quiptools_cuda"):
        sys.modules[name] = types.ModuleType(name)
    os.chdir(REF)                      # quant.py:8 opens hadamard.safetensors relatively
    sys.path.insert(0, REF)
    import codebook.e8p12 as e8p12      # noqa
    # numpy>=2 raises on np.int8(250); the reference relies on wrap-around
    # (e8p12.py:96).  Give that module a numpy whose int8() wraps.
    class _NP:
        def __getattr__(self, k):
            return getattr(np, k)
        @staticmethod
        def int8(v):
            return np.array(v).astype(np.int64).astype(np.int8)[()]
    e8p12.np = _NP()
    import register_lib                # noqa: defines quip_lib ops (cuda impls only)
    import quant, qlinear, codebook    # noqa
    return torch, quant, qlinear, codebook, e8p12


def main():
    torch, quant, qlinear, codebook, e8p12 = _import_reference()
    import torch._custom_ops as co
    sys.path.insert(0, REPO)
    from oracle import quip_oracle as O   # only for seeded layer *inputs*

    out = {}
    torch.manual_seed(0)

    # ---- F1/F2 E8P tables --------------------------------------------------
    packed = e8p12.get_packed_abs_grid()
    full, _ = e8p12.get_full_grid(packed)
    full_i8 = (full * 4).round().to(torch.int8).numpy()
    out["e8p_grid_packed_abs"] = packed.numpy()
    rng = np.random.Generator(np.random.PCG64(1234))
    samp = np.sort(rng.choice(1 << 16, 2048, replace=False)).astype(np.int32)
    out["e8p_full_sample_idx"] = samp
    out["e8p_full_sample_i8"] = full_i8[samp]
    meta = {"e8p_full_sha256": hashlib.sha256(full_i8.tobytes()).hexdigest(),
            "e8p_packed_sha256": hashlib.sha256(packed.numpy().tobytes()).hexdigest()}

    # ---- F3 e81b ------------------------------------------------------------
    import codebook.e8p12_rvq3 as rvq3
    e81b = rvq3.get_e81bgrid()
    e81b_packed = rvq3.pack_e81b(e81b)
    out["e81b_grid"] = e81b.numpy().astype(np.float32)
    out["e81b_grid_packed"] = e81b_packed.numpy().astype(np.int32)

    # ---- F4 D4, F5 HI ---------------------------------------------------------
    import codebook.d4 as d4
    import codebook.hi as hi
    out["d4_grid"] = d4.build_D4_CB().numpy().astype(np.float32)
    hcb = hi.HI4B1C_codebook(inference=False)
    hidx = torch.from_numpy(rng.integers(0, 16, (8, 64)).astype(np.int32))
    out["hi_idx"] = hidx.numpy()
    out["hi_packed"] = hcb.maybe_pack_idxs(hidx).numpy().astype(np.int32)
    out["hi_dense"] = hcb.grid[hidx.long(), 0].numpy().astype(np.float32)

    # ---- F6 get_hadK shapes ---------------------------------------------------
    hk = {}
    for n in (256, 688, 1024, 4096, 8192, 11008, 28672):
        for ur in (True, False):
            m, K, padn = quant.get_hadK(n, ur)
            hk[f"{n}_{int(ur)}"] = [int(K), int(padn), None if m is None else list(m.shape)]
    meta["get_hadK"] = hk

    # ---- F7 FHT goldens (pure-torch butterfly of the reference) ---------------
    for (n, K) in ((256, 1), (688, 43), (4096, 1), (11008, 43), (1792, 7)):
        x = torch.from_numpy(rng.standard_normal((3, n)).astype(np.float32))
        if K > 1:
            hadK = torch.from_numpy(O.random_orthogonal(K, rng).astype(np.float32))
        else:
            hadK = None
        y = quant.matmul_hadU(x, hadK, K, n)
        yt = quant.matmul_hadUt(x, hadK, K, n)
        out[f"fht_{n}_x"] = x.numpy()
        if hadK is not None:
            out[f"fht_{n}_hadK"] = hadK.numpy()
        out[f"fht_{n}_y"] = y.numpy()
        out[f"fht_{n}_yt"] = yt.numpy()

    # ---- CPU impls of the quip_lib ops, built from reference functions --------
    full_f = full.float()
    e81b_f = e81b.float()
    d4_f = d4.build_D4_CB().float()

    def _fma16(scale, resid, main):
        s = torch.tensor(scale, dtype=torch.float32).half().double()
        return (s * resid.double() + main.double()).half()

    def dec_e8p(Q):
        idx = Q.view(torch.int16).to(torch.int32) & 0xFFFF
        return full_f[idx.long()].reshape(Q.shape[0], -1).half()

    def dec_rvq4(Q, scale):
        q = Q.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        m = full_f[(q >> 16).long()]
        r = full_f[(q & 0xFFFF).long()]
        return _fma16(scale, r, m).reshape(Q.shape[0], -1)

    def dec_rvq3(Q, scale):
        b = Q.contiguous().view(torch.uint8).reshape(Q.shape[0], -1, 3).to(torch.int64)
        main = b[..., 1] | (b[..., 2] << 8)
        return _fma16(scale, e81b_f[b[..., 0]], full_f[main]).reshape(Q.shape[0], -1)

    def dec_d4(Q):
        return d4_f[Q.view(torch.uint8).long()].reshape(Q.shape[0], -1).half()

    def dec_hi(Q):
        q = Q.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
        cols = []
        nib_of_col = {c: i for i, c in enumerate((0, 2, 4, 6, 1, 3, 5, 7))}  # hi.py:41-50
        for c in range(8):
            cols.append(((q >> (4 * nib_of_col[c])) & 0xF).float() - 7.5)
        return torch.stack(cols, dim=-1).reshape(Q.shape[0], -1).half()

    def had(x, scale):
        n = x.shape[-1]
        y = quant.matmul_hadU(x.float().reshape(-1, n), None, 1, n) * (n ** 0.5) * scale
        return y.reshape(x.shape).to(x.dtype)

    def mm(x, W):
        return (x.float() @ W.float().T).to(x.dtype)

    co.impl("quip_lib::hadamard", device_types="cpu")(had)
    co.impl("quip_lib::decompress_e8p_origorder", device_types="cpu")(lambda Q, g: dec_e8p(Q))
    co.impl("quip_lib::decompress_e8prvq4_origorder", device_types="cpu")(lambda Q, g, s: dec_rvq4(Q, s))
    co.impl("quip_lib::decompress_e8prvq3_origorder", device_types="cpu")(lambda Q, g, g2, s: dec_rvq3(Q, s))
    co.impl("quip_lib::decompress_d4_origorder", device_types="cpu")(lambda Q, g: dec_d4(Q))
    co.impl("quip_lib::decompress_hi_origorder", device_types="cpu")(lambda Q: dec_hi(Q))
    co.impl("quip_lib::e8p_mm_origorder", device_types="cpu")(lambda x, Q, g: mm(x, dec_e8p(Q)))
    co.impl("quip_lib::e8prvq4_mm_origorder", device_types="cpu")(lambda x, Q, g, s: mm(x, dec_rvq4(Q, s)))
    co.impl("quip_lib::e8prvq3_mm_origorder", device_types="cpu")(lambda x, Q, g, g2, s: mm(x, dec_rvq3(Q, s)))
    co.impl("quip_lib::d4_mm_origorder", device_types="cpu")(lambda x, Q, g: mm(x, dec_d4(Q)))
    co.impl("quip_lib::hi_mm_origorder", device_types="cpu")(lambda x, Q: mm(x, dec_hi(Q)))

    # ---- RVQ / E8P quantize() goldens: pins index packing and value formula ----
    for cbid, kw in (("E8P12", {}), ("E8P12RVQ4B", {"opt_resid_scale": None}),
                     ("E8P12RVQ3B", {"opt_resid_scale": None})):
        cb = codebook.codebook_id[cbid](inference=False, **kw)
        X = torch.from_numpy(rng.standard_normal((16, 32)).astype(np.float32)) * 1.2
        vals, idxs = cb.quantize(X.reshape(-1, 8))
        vals = vals.reshape(16, 32)
        idxs = idxs.reshape(16, 4)
        packed_idx = cb.maybe_pack_idxs(idxs.to(cb.idx_dtype) if cbid != "E8P12RVQ3B" else idxs.to(torch.int32))
        out[f"quant_{cbid}_vals"] = vals.numpy().astype(np.float32)
        out[f"quant_{cbid}_Qidxs"] = packed_idx.numpy()
        if hasattr(cb, "opt_resid_scale"):
            meta[f"quant_{cbid}_resid_scale"] = float(cb.opt_resid_scale)

    # ---- F8 module goldens -------------------------------------------------------
    cases = []
    cfgs = [("E8P12", 256, 256, False, False), ("E8P12", 256, 688, True, False),
            ("E8P12", 688, 256, False, True), ("E8P12RVQ4B", 256, 256, False, False),
            ("E8P12RVQ4B", 256, 256, False, False, -1.0),
            ("E8P12RVQ3B", 256, 256, True, False), ("D4", 256, 256, False, False),
            ("HI", 256, 256, False, False), ("E8P12", 1792, 512, False, False)]
    for ci, cfg in enumerate(cfgs):
        cbid, fin, fout, has_bias, per_ch = cfg[:5]
        rs = cfg[5] if len(cfg) > 5 else None
        P = O.make_layer(cbid, fin, fout, seed=100 + ci, bias=has_bias,
                         per_channel=per_ch, resid_scale=rs)
        kw = {}
        if cbid.startswith("E8P12RVQ"):
            kw["opt_resid_scale"] = P.resid_scale
        cb = codebook.codebook_id[cbid](inference=True, **kw)
        if cbid == "D4":
            cb.grid = cb.grid.half()
        layer = qlinear.QuantLinear(fin, fout, cb, bias=has_bias, use_rand=True,
                                    per_channel=per_ch)
        assert tuple(layer.Qidxs.shape) == P.Qidxs.shape, (layer.Qidxs.shape, P.Qidxs.shape)
        assert layer.K_left == P.K_left and layer.K_right == P.K_right
        assert layer.Qidxs.dtype == torch.from_numpy(P.Qidxs).dtype
        with torch.no_grad():
            layer.Qidxs.copy_(torch.from_numpy(P.Qidxs))
            layer.SU.copy_(torch.from_numpy(P.SU))
            layer.SV.copy_(torch.from_numpy(P.SV))
            layer.Wscale.copy_(torch.from_numpy(np.asarray(P.Wscale)))
            if P.had_left is not None:
                layer.had_left.copy_(torch.from_numpy(P.had_left))
            if P.had_right is not None:
                layer.had_right.copy_(torch.from_numpy(P.had_right))
            if has_bias:
                layer.bias.copy_(torch.from_numpy(P.bias))
        layer.wscale_float = P.wscale_float      # quantizer.py:836-837
        layer.eval()
        for M in (1, 5, 31, 32, 40):
            x = torch.from_numpy((rng.standard_normal((M, fin))).astype(np.float16))
            with torch.no_grad():
                y = layer(x)
            key = f"mod{ci}_M{M}"
            out[key + "_x"] = x.numpy()
            out[key + "_y"] = y.numpy()
        # training-mode path == x @ calc_weight (qlinear.py:93-97,144-159), scalar Wscale only
        if not per_ch:
            with torch.no_grad():
                layer.Wscale.fill_(P.wscale_float)
                W = layer.calc_weight(cache=False)
            wrows = np.unique(np.linspace(0, W.shape[0] - 1, 24).astype(np.int64))
            out[f"mod{ci}_Wrows"] = wrows          # row sample keeps the fixture small
            out[f"mod{ci}_W"] = W.numpy()[wrows]
        cases.append(dict(idx=ci, codebook=cbid, in_features=fin, out_features=fout,
                          bias=has_bias, per_channel=per_ch, seed=100 + ci,
                          resid_scale=P.resid_scale, Ms=[1, 5, 31, 32, 40]))
    meta["module_cases"] = cases

    # ---- F9 config-1 golden: E8P12 4096->4096, M=1 ---------------------------------
    P = O.make_layer("E8P12", 4096, 4096, seed=0)
    cb = codebook.codebook_id["E8P12"](inference=True)
    layer = qlinear.QuantLinear(4096, 4096, cb, bias=False, use_rand=True)
    with torch.no_grad():
        layer.Qidxs.copy_(torch.from_numpy(P.Qidxs))
        layer.SU.copy_(torch.from_numpy(P.SU))
        layer.SV.copy_(torch.from_numpy(P.SV))
        layer.Wscale.copy_(torch.from_numpy(np.asarray(P.Wscale)))
    layer.wscale_float = P.wscale_float
    layer.eval()
    x = torch.from_numpy(rng.standard_normal((1, 4096)).astype(np.float16))
    with torch.no_grad():
        y = layer(x)
    out["cfg1_x"] = x.numpy()
    out["cfg1_y"] = y.numpy()
    meta["cfg1"] = dict(codebook="E8P12", in_features=4096, out_features=4096, seed=0)

    # ---- QuantLinear buffer-layout pin (qlinear.py:10-84) ---------------------------
    layout = {}
    for cbid in ("E8P12", "E8P12RVQ3B", "E8P12RVQ4B", "D4", "HI"):
        cb = codebook.codebook_id[cbid](inference=True)
        L = qlinear.QuantLinear(11008, 4096, cb, bias=True, use_rand=True)
        layout[cbid] = {k: [list(v.shape), str(v.dtype)] for k, v in L.state_dict().items()}
        layout[cbid]["_K"] = [L.K_left, L.K_right, L.q_in_features, L.q_out_features]
    meta["state_dict_layout"] = layout

    os.chdir(HERE)
    np.savez_compressed(os.path.join(HERE, "reference_golden.npz"), **out)
    with open(os.path.join(HERE, "reference_golden.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(HERE, "reference_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
