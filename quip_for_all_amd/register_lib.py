"""`quip_lib` operator boundary: same op names and schemas as the reference's
register_lib.py:8-192, implemented by the C-ABI library (capi.py) on the
caller's current HIP stream.  Only a CUDA(=HIP) implementation and a fake
(meta) implementation are registered -- like the reference there is no CPU
kernel, and there is deliberately no CPU fallback."""
import math
from typing import Optional

import torch
from torch import Tensor

from . import capi

try:
    _lib = torch.library.Library("quip_lib", "DEF")
except RuntimeError:  # namespace already defined in this process
    _lib = torch.library.Library("quip_lib", "FRAGMENT")

_SCHEMAS = {
    "hadamard": "(Tensor x, float scale) -> Tensor",
    "e8p_mm_origorder": "(Tensor x, Tensor Qidxs, Tensor grid) -> Tensor",
    "e8prvq3_mm_origorder": "(Tensor x, Tensor Qidxs, Tensor grid, Tensor grid2, float scale) -> Tensor",
    "e8prvq4_mm_origorder": "(Tensor x, Tensor Qidxs, Tensor grid, float scale) -> Tensor",
    "d4_mm_origorder": "(Tensor x, Tensor Qidxs, Tensor grid) -> Tensor",
    "hi_mm_origorder": "(Tensor x, Tensor Qidxs) -> Tensor",
    "decompress_e8p_origorder": "(Tensor Qidxs, Tensor grid) -> Tensor",
    "decompress_e8prvq3_origorder": "(Tensor Qidxs, Tensor grid, Tensor grid2, float scale) -> Tensor",
    "decompress_e8prvq4_origorder": "(Tensor Qidxs, Tensor grid, float scale) -> Tensor",
    "decompress_d4_origorder": "(Tensor Qidxs, Tensor grid) -> Tensor",
    "decompress_hi_origorder": "(Tensor Qidxs) -> Tensor",
    # fused Hadamard side of QuantLinear.forward (no reference counterpart: it
    # replaces the x*SU / pad / hadamard / hadK@ / *SV / +bias op sequence)
    "had_transform": "(Tensor x, int out_features, int n, int K, Tensor? had, bool transpose, "
                     "Tensor? pre, Tensor? pre2, Tensor? post, Tensor? bias, float scale) -> Tensor",
    # bs=1 decode path: input-side transform straight to int8 digit planes, and the GEMV on them
    "had_transform_planes": "(Tensor x, int n, int K, Tensor? had, bool transpose, Tensor? pre, float scale) -> Tensor",
    "e8p_gemv_planes": "(Tensor planes, Tensor Qidxs, Tensor grid) -> Tensor",
    # the same two transforms with decoder-block glue folded in (RMSNorm / SiLU*mul in, residual out)
    "had_transform_fused": "(Tensor x, int out_features, int n, int K, Tensor? had, bool transpose, Tensor? pre, "
                           "Tensor? pre2, Tensor? post, Tensor? bias, float scale, Tensor? residual, "
                           "Tensor? rms_weight, float rms_eps, Tensor? gate) -> Tensor",
    # grouped launches: the three stages of up to 3 QuantLinear modules reading the same activation
    "had_transform_planes_group": "(Tensor x, int n, int K, Tensor?[] had, bool transpose, Tensor?[] pre, "
                                  "float[] scale, Tensor? rms_weight, float rms_eps, Tensor? gate, "
                                  "float resid_scale=0.0) -> Tensor[]",
    "e8p_gemv_planes_group": "(Tensor[] planes, Tensor[] Qidxs, Tensor grid) -> Tensor[]",
    # rope + KV append + attention on the RAW GEMV outputs of q / k / v_proj (their K = 1 output transforms in the
    # launch's prologue): zs / posts = [q, k, v], scales = 1 / sqrt(n)
    "rope_attn_decode_z": "(Tensor[] zs, Tensor[] posts, float[] scales, Tensor cos, Tensor sin, Tensor pos, "
                          "Tensor(a!) kcache, Tensor(b!) vcache, Tensor? workspace=None, int window=0) -> Tensor",
    # M >= 32 (prefill): fused dequant + MFMA GEMM, x (M, k) fp16 -> (M, n) fp16; no dense W (csrc/e8p_prefill_gemm.hip)
    "e8p_mm_batched": "(Tensor x, Tensor Qidxs, Tensor grid) -> Tensor",
    # ... with the other codebooks' decode (argument meaning as the *_mm_skinny ops below)
    "e8prvq4_mm_batched": "(Tensor x, Tensor Qidxs, Tensor grid, float scale) -> Tensor",
    "e8prvq3_mm_batched": "(Tensor x, Tensor Qidxs, Tensor grid, Tensor grid2, float scale) -> Tensor",
    "d4_mm_batched": "(Tensor x, Tensor Qidxs, Tensor grid) -> Tensor",
    "hi_mm_batched": "(Tensor x, Tensor Qidxs) -> Tensor",
    # 2 <= M <= 32 rows in one pass over the codes, fp16 MFMA (csrc/e8p_skinny_gemm.hip)
    "e8p_mm_skinny": "(Tensor x, Tensor Qidxs, Tensor grid) -> Tensor",
    # the same for E8P12RVQ4B: int32 codes (main << 16 | residual), weights = fp16 fma(scale, residual, main)
    "e8prvq4_mm_skinny": "(Tensor x, Tensor Qidxs, Tensor grid, float scale) -> Tensor",
    "e8prvq3_mm_skinny": "(Tensor x, Tensor Qidxs, Tensor grid, Tensor grid2, float scale) -> Tensor",   # packed 3-byte codes
    "d4_mm_skinny": "(Tensor x, Tensor Qidxs, Tensor grid) -> Tensor",      # uint8 codes (n, k/4), the fp16 (256, 4) table
    "hi_mm_skinny": "(Tensor x, Tensor Qidxs) -> Tensor",                   # int32 codes (n, k/8), eight nibbles each
    # E8P12RVQ3B on the matrix-core GEMV: Qidxs = the checkpoint's 3-byte codes (int32 (n, 3k/32)), e81b_i8 = int8 (256, 8)
    "e8prvq3_gemv_planes_group": "(Tensor[] planes, Tensor[] Qidxs, Tensor grid, Tensor e81b_i8) -> Tensor[]",
    "d4_gemv_planes": "(Tensor planes, Tensor Qidxs, Tensor grid) -> Tensor",
    "d4_gemv_planes_group": "(Tensor[] planes, Tensor[] Qidxs, Tensor grid) -> Tensor[]",
    # 2..3 activation rows against ONE weight matrix: the grouped GEMV with the same codes for every row
    "e8p_mm_planes_rows": "(Tensor[] planes, Tensor Qidxs, Tensor grid) -> Tensor",
    # skinny GEMM on the matrix cores: M rows -> (M, planes_bytes) plane images -> (M, n); the GEMV op splits
    # the rows into passes of quip_e8p_gemv_max_rows(n, k) (5 for k <= 4096) rows each
    "had_transform_planes_rows": "(Tensor x, int n, int K, Tensor? had, bool transpose, Tensor? pre, float scale, "
                                 "Tensor? rms_weight, float rms_eps, Tensor? gate, float resid_scale=0.0) -> Tensor",
    "e8p_gemv_planes_rows": "(Tensor planes, Tensor Qidxs, Tensor grid) -> Tensor",
    # the same for the other table modes: 64 = D4 table (Qidxs uint8 (n, k/4)), 40 = E8P12RVQ3B (the 3-byte codes)
    "gemv_planes_rows_mode": "(Tensor planes, Tensor Qidxs, Tensor grid, Tensor? grid2, int mode) -> Tensor",
    # quantise-time nearest E8P12 codeword: X (N, 8) fp32 -> (vals (N, 8) fp32, idx (N) int64)
    "e8p_quantize": "(Tensor X, Tensor grid) -> (Tensor, Tensor)",
    # chain: output side of the producer module (z, its SV, residual) + input transforms of 1..3 consumers;
    # returns [h] + planes
    "had_chain_planes_group": "(Tensor z, Tensor z_post, Tensor? z_residual, float z_scale, int n, Tensor[] pre, "
                              "float[] scale, Tensor? rms_weight, float rms_eps, float resid_scale=0.0) -> Tensor[]",
    "had_transform_group": "(Tensor[] x, int[] out_features, int n, int K, Tensor?[] had, bool transpose, "
                           "Tensor?[] pre2, Tensor?[] post, Tensor?[] bias, float[] scale, Tensor?[] residual, "
                           "Tensor?[] pre, Tensor? rms_weight, float rms_eps, int[]? ns=None) -> Tensor[]",
    # GEMV(s) with the input side computed in the prologue: [h_out]? + [y_i]; see quip_e8p_gemv_fused
    "e8p_gemv_fused": "(Tensor? x, Tensor? z, Tensor? post, Tensor? residual, Tensor? rms_weight, float rms_eps, "
                      "float z_scale, Tensor[] pre, float[] scale, Tensor[] Qidxs, Tensor grid) -> Tensor[]",
    # persistent decode engine, stage 1: GEMV[gate, up] -> output transforms -> SiLU product -> input transform of down
    # -> GEMV[down] in one launch (csrc/decode_engine.hip); returns down's raw product (1, hidden)
    "ffn_engine": "(Tensor planes_gate, Tensor planes_up, Tensor q_gate, Tensor q_up, Tensor q_down, Tensor had3, "
                  "Tensor sv_gate, Tensor sv_up, Tensor su_down, Tensor grid, Tensor(a!) workspace, float out_scale, "
                  "float in_scale, int K, Tensor? dbg=None) -> Tensor",
    # persistent decode engine, stage 2: n_layers decoder blocks of one token in one launch (csrc/decode_block.hip).
    # `layers` = the packed descriptors (decode.py builds them; they point at the KV caches, which the launch appends to)
    "block_engine": "(Tensor layers, Tensor h_in, Tensor pos, Tensor cos, Tensor sin, Tensor grid, Tensor(a!) workspace, "
                    "int n_layers, int max_len, float rms_eps, float attn_scale, Tensor? dbg=None, int dbg_layer=-1, "
                    "int codebook=0, float resid_scale=0.0, int shape=0, Tensor? grid2=None, "
                    # the KV caches the descriptors point into: row *pos of every block is written (declared here so that the
                    # mutation is visible to PyTorch; the kernel reaches them through the descriptors)
                    "Tensor(b!)? kcache=None, Tensor(c!)? vcache=None) -> Tensor",
    # decode-step glue between q/k/v_proj and o_proj: rope + KV-cache append + single-query attention
    "rope_attn_decode": "(Tensor q, Tensor k, Tensor v, Tensor cos, Tensor sin, Tensor pos, Tensor(a!) kcache, "
                        "Tensor(b!) vcache, Tensor(c!)? workspace, int window=0) -> Tensor",   # window > 0: the last `window` positions only
    # greedy tail of the decode step: tok <- argmax(logits) (first maximum), pos += 1
    "argmax_step": "(Tensor logits, Tensor(a!) tok, Tensor(b!) pos) -> ()",
    "had_transform_planes_fused": "(Tensor x, int n, int K, Tensor? had, bool transpose, Tensor? pre, float scale, "
                                  "Tensor? rms_weight, float rms_eps, Tensor? gate, float resid_scale=0.0) -> Tensor",
}
for _name, _schema in _SCHEMAS.items():
    try:
        _lib.define(_name + _schema)
    except RuntimeError:
        pass  # already defined (e.g. the reference's register_lib was imported first)


# QUIP_POISON_OUTPUTS=1 (debug): every output tensor an op allocates is filled with a pattern (fp16 / fp32 NaN, 0xA5 bytes)
# before the launch instead of being left as _empty() memory, so a kernel that does not write all of its output,
# or that reads its output buffer, shows up as NaN / garbage instead of depending on what the allocator recycled.
import os as _os
_POISON = _os.environ.get("QUIP_POISON_OUTPUTS", "0") != "0"


def _empty(*shape, **kw):
    t = torch.empty(*shape, **kw)
    if _POISON and not torch.cuda.is_current_stream_capturing():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        else:
            t.view(torch.uint8).fill_(0xA5)
    return t


def _stream(t: Tensor):
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


def _need(cond, msg):
    if not cond:
        raise ValueError("quip_lib: " + msg)


def _chk_x(x: Tensor):
    _need(x.dim() == 2, "x must be 2-D (M, in)")
    _need(x.dtype == torch.float16, f"x must be float16, got {x.dtype}")
    return x.contiguous()


def _chk_lens(what, in_features, out_features, n, K, had=None, pre=None, pre2=None, post=None, bias=None,
              rms_weight=None):
    """element counts of the vectors a transform launch reads (a short vector would be read out of bounds):
    pre / rms_weight multiply the `in_features` inputs, pre2 the n transformed values, post / bias the
    `out_features` outputs, had is the (K, K) factor"""
    _need(0 < in_features <= n and 0 < out_features <= n and K >= 1 and n % K == 0,
          f"{what}: in_features {in_features} / out_features {out_features} must be in (0, n = {n}], K = {K} must divide n")
    for name, t, want in (("had", had, K * K), ("pre", pre, in_features), ("rms_weight", rms_weight, in_features),
                          ("pre2", pre2, n), ("post", post, out_features), ("bias", bias, out_features)):
        _need(t is None or t.numel() == want, f"{what}: {name} has {0 if t is None else t.numel()} elements, expected {want}")
    _need(K == 1 or had is not None, f"{what}: K = {K} needs the (K, K) factor")


def _chk_q(q: Tensor, dtype):
    _need(q.dim() == 2, "Qidxs must be 2-D")
    _need(q.dtype == dtype, f"Qidxs must be {dtype}, got {q.dtype}")
    return q.contiguous()


# ---- hadamard ---------------------------------------------------------------------
_HAD_DTYPES = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}     # QUIP_DTYPE_*


def _hadamard_cuda(x: Tensor, scale: float) -> Tensor:
    """fp16 / bf16 / fp32 like fast_hadamard_transform (register_lib.py:10-20); fp32 inside, output in x's dtype"""
    _need(x.dtype in _HAD_DTYPES, f"hadamard: float16 / bfloat16 / float32, got {x.dtype}")
    n = x.shape[-1]
    _need(n & (n - 1) == 0 and 0 < n <= 32768, f"hadamard length {n} must be a power of two <= 32768")
    xc = x.contiguous()
    y = torch.empty_like(xc)
    rows = xc.numel() // n
    if rows == 0:
        return y
    with torch.cuda.device(x.device):
        if x.dtype == torch.float16:
            capi.check(capi.lib().quip_hadamard_f16(xc.data_ptr(), y.data_ptr(), rows, n, float(scale),
                                                    _stream(x)), "quip_hadamard_f16")
        else:
            capi.check(capi.lib().quip_hadamard(xc.data_ptr(), y.data_ptr(), rows, n, float(scale),
                                                _HAD_DTYPES[x.dtype], _stream(x)), "quip_hadamard")
    return y


def _had_transform_cuda(x, out_features, n, K, had, transpose, pre, pre2, post, bias, scale):
    xc = _chk_x(x)
    for t in (had, pre, pre2, post, bias):
        _need(t is None or (t.dtype == torch.float16 and t.is_contiguous() and t.device == x.device),
              "had_transform: vectors must be contiguous float16 on x's device")
    _chk_lens("had_transform", xc.shape[1], out_features, n, K, had, pre, pre2, post, bias)
    y = _empty((xc.shape[0], out_features), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().quip_had_transform_f16(
            xc.data_ptr(), y.data_ptr(), xc.shape[0], xc.shape[1], out_features, n, K, _ptr(had),
            int(bool(transpose)), _ptr(pre), _ptr(pre2), _ptr(post), _ptr(bias), float(scale),
            _stream(x)), "quip_had_transform_f16")
    return y


def _had_transform_planes_cuda(x, n, K, had, transpose, pre, scale):
    xc = _chk_x(x)
    _need(xc.shape[0] == 1, "had_transform_planes is the bs=1 path (one row)")
    for t in (had, pre):
        _need(t is None or (t.dtype == torch.float16 and t.is_contiguous() and t.device == x.device),
              "had_transform_planes: vectors must be contiguous float16 on x's device")
    _chk_lens("had_transform_planes", xc.shape[1], n, n, K, had, pre)
    L = capi.lib()
    planes = _empty(L.quip_e8p_planes_bytes(n), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        capi.check(L.quip_had_transform_planes(xc.data_ptr(), planes.data_ptr(), xc.shape[1], n, K, _ptr(had),
                                               int(bool(transpose)), _ptr(pre), float(scale), _stream(x)),
                   "quip_had_transform_planes")
    return planes


def _fusion(residual, rms_weight, rms_eps, gate, x):
    for t in (residual, rms_weight, gate):
        _need(t is None or (t.dtype == torch.float16 and t.is_contiguous() and t.device == x.device),
              "fusion tensors must be contiguous float16 on x's device")
    return capi.HadFusion(_ptr(residual), _ptr(rms_weight), _ptr(gate), float(rms_eps))


def _had_transform_fused_cuda(x, out_features, n, K, had, transpose, pre, pre2, post, bias, scale, residual,
                              rms_weight, rms_eps, gate):
    xc = _chk_x(x)
    for t in (had, pre, pre2, post, bias):
        _need(t is None or (t.dtype == torch.float16 and t.is_contiguous() and t.device == x.device),
              "had_transform: vectors must be contiguous float16 on x's device")
    _need(gate is None or gate.shape == xc.shape, "gate must have x's shape")
    _need(residual is None or tuple(residual.shape) == (xc.shape[0], out_features), "residual shape")
    _chk_lens("had_transform_fused", xc.shape[1], out_features, n, K, had, pre, pre2, post, bias, rms_weight)
    y = _empty((xc.shape[0], out_features), dtype=torch.float16, device=x.device)
    f = _fusion(residual, rms_weight, rms_eps, gate, x)
    import ctypes
    with torch.cuda.device(x.device):
        capi.check(capi.lib().quip_had_transform_fused_f16(
            xc.data_ptr(), y.data_ptr(), xc.shape[0], xc.shape[1], out_features, n, K, _ptr(had),
            int(bool(transpose)), _ptr(pre), _ptr(pre2), _ptr(post), _ptr(bias), float(scale), ctypes.byref(f),
            _stream(x)), "quip_had_transform_fused_f16")
    return y


def _had_transform_planes_fused_cuda(x, n, K, had, transpose, pre, scale, rms_weight, rms_eps, gate, resid_scale=0.0):
    if resid_scale != 0.0:     # the RVQ4 virtual-vector layout is served by the problem-struct entry point
        return _had_transform_planes_group_cuda(x, n, K, [had], transpose, [pre], [scale], rms_weight, rms_eps, gate,
                                                resid_scale)[0]
    xc = _chk_x(x)
    _need(xc.shape[0] == 1, "had_transform_planes is the bs=1 path (one row)")
    for t in (had, pre):
        _need(t is None or (t.dtype == torch.float16 and t.is_contiguous() and t.device == x.device),
              "had_transform_planes: vectors must be contiguous float16 on x's device")
    _need(gate is None or gate.shape == xc.shape, "gate must have x's shape")
    _chk_lens("had_transform_planes_fused", xc.shape[1], n, n, K, had, pre, rms_weight=rms_weight)
    L = capi.lib()
    planes = _empty(L.quip_e8p_planes_bytes(n), dtype=torch.uint8, device=x.device)
    f = _fusion(None, rms_weight, rms_eps, gate, x)
    import ctypes
    with torch.cuda.device(x.device):
        capi.check(L.quip_had_transform_planes_fused(xc.data_ptr(), planes.data_ptr(), xc.shape[1], n, K, _ptr(had),
                                                     int(bool(transpose)), _ptr(pre), float(scale), ctypes.byref(f),
                                                     _stream(x)), "quip_had_transform_planes_fused")
    return planes


def _vec_ok(t, dev):
    _need(t is None or (t.dtype == torch.float16 and t.is_contiguous() and t.device == dev),
          "vectors must be contiguous float16 on x's device")
    return _ptr(t)


def _d4_grid(grid):
    _need(grid.dtype == torch.float16 and grid.is_contiguous() and tuple(grid.shape) == (256, 4),
          "D4 grid must be the contiguous fp16 (256, 4) table")
    return grid


def _d4_gemv_planes_cuda(planes, Qidxs, grid):
    return _d4_gemv_planes_group_cuda([planes], [Qidxs], grid)[0]


def _e8prvq3_gemv_planes_group_cuda(planes, Qidxs, grid, e81b_i8):
    import ctypes
    count = len(planes)
    _need(1 <= count <= capi.MAX_GROUP and len(Qidxs) == count, "group of 1..3 problems")
    _need(Qidxs[0].shape[1] % 3 == 0, "Qidxs: the checkpoint's packed int32 (n, 3 k / 32) codes")
    k = Qidxs[0].shape[1] * 32 // 3
    dev = planes[0].device
    for pl, q in zip(planes, Qidxs):
        _need(q.dtype == torch.int32 and q.is_contiguous() and q.shape[1] * 32 == 3 * k and q.device == dev,
              "Qidxs must be the checkpoint's contiguous int32 (n, 3 k / 32) tensors (3-byte codes) with a common k")
        _need(pl.dtype == torch.uint8 and pl.is_contiguous() and pl.device == dev, "planes must be uint8")
    _need(e81b_i8.dtype == torch.int8 and tuple(e81b_i8.shape) == (256, 8) and e81b_i8.is_contiguous()
          and e81b_i8.device == dev, "e81b_i8 must be the contiguous int8 (256, 8) table")
    g = _grid_i64(grid, planes[0])
    outs = [_empty((1, q.shape[0]), dtype=torch.float16, device=dev) for q in Qidxs]
    vp = ctypes.c_void_p * count
    ns = (ctypes.c_int32 * count)(*[q.shape[0] for q in Qidxs])
    ws = _gemv_workspace(dev, sum(q.shape[0] for q in Qidxs))
    with torch.cuda.device(dev):
        capi.check(capi.lib().quip_e8prvq3_gemv_planes_group_ws(
            vp(*[p.data_ptr() for p in planes]), vp(*[q.data_ptr() for q in Qidxs]), g.data_ptr(), e81b_i8.data_ptr(),
            vp(*[o.data_ptr() for o in outs]), ns, count, k, ws.data_ptr(), ws.numel() * 4, _stream(planes[0])),
            "quip_e8prvq3_gemv_planes_group_ws")
    return outs


def _d4_gemv_planes_group_cuda(planes, Qidxs, grid):
    import ctypes
    count = len(planes)
    _need(1 <= count <= capi.MAX_GROUP and len(Qidxs) == count, "group of 1..3 problems")
    k = Qidxs[0].shape[1] * 4
    dev = planes[0].device
    for pl, q in zip(planes, Qidxs):
        _need(q.dtype == torch.uint8 and q.is_contiguous() and q.shape[1] * 4 == k and q.device == dev,
              "Qidxs must be contiguous uint8 (n, k/4) with a common k")
        _need(pl.dtype == torch.uint8 and pl.is_contiguous() and pl.device == dev, "planes must be uint8")
    outs = [_empty((1, q.shape[0]), dtype=torch.float16, device=dev) for q in Qidxs]
    vp = ctypes.c_void_p * count
    ns = (ctypes.c_int32 * count)(*[q.shape[0] for q in Qidxs])
    ws = _gemv_workspace(dev, sum(q.shape[0] for q in Qidxs))    # (rows beyond 28672: the K-splitting kernel's partial sums)
    with torch.cuda.device(dev):
        capi.check(capi.lib().quip_d4_gemv_planes_group_ws(
            vp(*[p.data_ptr() for p in planes]), vp(*[q.data_ptr() for q in Qidxs]), _d4_grid(grid).data_ptr(),
            vp(*[o.data_ptr() for o in outs]), ns, count, k, ws.data_ptr(), ws.numel() * 4, _stream(planes[0])),
            "quip_d4_gemv_planes_group_ws")
    return outs


def _had_transform_planes_rows_cuda(x, n, K, had, transpose, pre, scale, rms_weight, rms_eps, gate, resid_scale=0.0):
    xc = _chk_x(x)
    _need(gate is None or (gate.shape == xc.shape and gate.dtype == torch.float16 and gate.is_contiguous()),
          "gate must be contiguous float16 with x's shape")
    _chk_lens("had_transform_planes_rows", xc.shape[1], n, n, K, had, pre, rms_weight=rms_weight)
    L = capi.lib()
    rows = xc.shape[0]
    out = _empty((rows, _planes_numel(n, resid_scale)), dtype=torch.uint8, device=x.device)
    pr = capi.HadProblem(xc.data_ptr(), out.data_ptr(), _vec_ok(had, x.device), _vec_ok(pre, x.device), None, None,
                         None, None, _vec_ok(rms_weight, x.device), _ptr(gate), xc.shape[1], n, float(scale),
                         float(rms_eps), None, None, None, None, 1.0, *_layout(resid_scale))
    import ctypes
    with torch.cuda.device(x.device):
        capi.check(L.quip_had_transform_planes_rows(ctypes.byref(pr), rows, n, K, int(bool(transpose)), _stream(x)),
                   "quip_had_transform_planes_rows")
    return out


def _argmax_step_cuda(logits, tok, pos):
    _need(logits.dtype == torch.float16 and logits.is_contiguous() and logits.dim() <= 2
          and (logits.dim() == 1 or logits.shape[0] == 1), "argmax_step: logits must be contiguous float16 (1, n)")
    _need(tok.dtype == torch.int64 and pos.dtype == torch.int64 and tok.numel() >= 1 and pos.numel() >= 1
          and tok.device == logits.device and pos.device == logits.device, "tok / pos: int64 on the logits' device")
    with torch.cuda.device(logits.device):
        capi.check(capi.lib().quip_argmax_step_f16(logits.data_ptr(), logits.numel(), tok.data_ptr(), pos.data_ptr(),
                                                   _stream(logits)), "quip_argmax_step_f16")


def _e8p_quantize_cuda(X, grid):
    _need(X.dim() == 2 and X.shape[1] == 8 and X.dtype == torch.float32 and X.is_contiguous(),
          "e8p_quantize: X must be contiguous float32 (N, 8)")
    g = _grid_i64(grid, X)
    vals = torch.empty_like(X)
    idx = _empty(X.shape[0], dtype=torch.int64, device=X.device)
    with torch.cuda.device(X.device):
        capi.check(capi.lib().quip_e8p_quantize_f32(X.data_ptr(), X.shape[0], g.data_ptr(), vals.data_ptr(),
                                                    idx.data_ptr(), _stream(X)), "quip_e8p_quantize_f32")
    return vals, idx


def _e8p_gemv_planes_rows_cuda(planes, Qidxs, grid):
    _need(Qidxs.dtype == torch.int16 and Qidxs.is_contiguous(), "Qidxs must be contiguous int16 (n, k/8)")
    n, k = Qidxs.shape[0], Qidxs.shape[1] * 8
    L = capi.lib()
    _need(planes.dim() == 2 and planes.dtype == torch.uint8 and planes.is_contiguous()
          and planes.shape[1] == L.quip_e8p_planes_bytes(k) and planes.device == Qidxs.device,
          "planes must be the (rows, quip_e8p_planes_bytes(k)) uint8 images of had_transform_planes_rows")
    rows = planes.shape[0]
    per = L.quip_e8p_gemv_max_rows(n, k)
    g = _grid_i64(grid, Qidxs)
    out = _empty((rows, n), dtype=torch.float16, device=Qidxs.device)
    if per < 1:
        # rows longer than rows mode holds in LDS (k > 28672: E8P12RVQ4B's 2k-wide virtual rows at 70B): one bs=1
        # launch per row through the dispatcher (the K-splitting kernel) -- the same exact integer sums
        ws = _gemv_workspace(Qidxs.device, n)
        with torch.cuda.device(Qidxs.device):
            for r in range(rows):
                capi.check(L.quip_e8p_gemv_planes_ws(planes[r].data_ptr(), Qidxs.data_ptr(), g.data_ptr(),
                                                     out[r].data_ptr(), n, k, ws.data_ptr(), ws.numel() * 4,
                                                     _stream(out)), "quip_e8p_gemv_planes_ws")
        return out
    with torch.cuda.device(Qidxs.device):
        for r0 in range(0, rows, per):
            m = min(per, rows - r0)
            capi.check(L.quip_e8p_gemv_planes_rows(planes[r0].data_ptr(), Qidxs.data_ptr(), g.data_ptr(),
                                                   out[r0].data_ptr(), m, n, k, _stream(out)),
                       "quip_e8p_gemv_planes_rows")
    return out


def _gemv_planes_rows_mode_cuda(planes, Qidxs, grid, grid2, mode):
    _need(mode in (64, 40), "mode: 64 (D4 table) or 40 (E8P12RVQ3B tables)")
    _need(Qidxs.is_contiguous() and Qidxs.dtype == (torch.uint8 if mode == 64 else torch.int32),
          "Qidxs: contiguous uint8 (n, k/4) for mode 64, the checkpoint's int32 (n, 3 k / 32) 3-byte codes for mode 40")
    n = Qidxs.shape[0]
    _need(mode == 64 or Qidxs.shape[1] % 3 == 0, "mode 40: Qidxs (n, 3 k / 32)")
    k = Qidxs.shape[1] * 4 if mode == 64 else Qidxs.shape[1] * 64 // 3   # mode 40: 2 * in features virtual weights
    L = capi.lib()
    _need(planes.dim() == 2 and planes.dtype == torch.uint8 and planes.is_contiguous()
          and planes.shape[1] == L.quip_e8p_planes_bytes(k) and planes.device == Qidxs.device,
          "planes must be (rows, quip_e8p_planes_bytes(k)) uint8 images")
    if mode == 64:
        g = _d4_grid(grid)
    else:
        g = _grid_i64(grid, planes)
        _need(grid2 is not None and grid2.dtype == torch.int8 and tuple(grid2.shape) == (256, 8) and grid2.is_contiguous(),
              "grid2 must be the contiguous int8 (256, 8) E81B table")
    rows = planes.shape[0]
    per = L.quip_gemv_max_rows_mode(n, k, mode)
    if per < 1 and mode == 40:
        # virtual rows longer than rows mode holds in LDS (70B down_proj): one bs=1 launch per row (K-splitting kernel)
        return torch.cat([_e8prvq3_gemv_planes_group_cuda([planes[r]], [Qidxs], grid, grid2)[0] for r in range(rows)])
    if per < 1 and mode == 64:      # ... the same for the D4 table mode (HI's virtual rows)
        return torch.cat([_d4_gemv_planes_group_cuda([planes[r]], [Qidxs], grid)[0] for r in range(rows)])
    _need(per >= 1, "shape not supported by the matrix-core GEMV")
    out = _empty((rows, n), dtype=torch.float16, device=Qidxs.device)
    with torch.cuda.device(Qidxs.device):
        for r0 in range(0, rows, per):
            m = min(per, rows - r0)
            capi.check(L.quip_gemv_planes_rows_mode(planes[r0].data_ptr(), Qidxs.data_ptr(), g.data_ptr(), _ptr(grid2),
                                                    out[r0].data_ptr(), m, n, k, mode, _stream(out)),
                       "quip_gemv_planes_rows_mode")
    return out


def _e8p_mm_planes_rows_cuda(planes, Qidxs, grid):
    import ctypes
    count = len(planes)
    _need(1 <= count <= capi.MAX_GROUP, "1..3 rows")
    _need(Qidxs.dtype == torch.int16 and Qidxs.is_contiguous(), "Qidxs must be contiguous int16 (n, k/8)")
    n, k = Qidxs.shape[0], Qidxs.shape[1] * 8
    dev = Qidxs.device
    out = _empty((count, n), dtype=torch.float16, device=dev)
    vp = ctypes.c_void_p * count
    ns = (ctypes.c_int32 * count)(*([n] * count))
    g = _grid_i64(grid, Qidxs)
    with torch.cuda.device(dev):
        capi.check(capi.lib().quip_e8p_gemv_planes_group(
            vp(*[p.data_ptr() for p in planes]), vp(*([Qidxs.data_ptr()] * count)), g.data_ptr(),
            vp(*[out.data_ptr() + 2 * n * i for i in range(count)]), ns, count, k, _stream(out)),
            "quip_e8p_gemv_planes_group (rows)")
    return out


def _had_transform_planes_group_cuda(x, n, K, had, transpose, pre, scale, rms_weight, rms_eps, gate, resid_scale=0.0):
    xc = _chk_x(x)
    count = len(pre)
    _need(xc.shape[0] in (1, count), "had_transform_planes_group: x has one row (shared) or one row per problem")
    _need(1 <= count <= capi.MAX_GROUP and len(had) == count and len(scale) == count, "group of 1..3 problems")
    _need(gate is None or gate.shape == xc.shape, "gate must have x's shape")
    for i in range(count):
        _chk_lens("had_transform_planes_group", xc.shape[1], n, n, K, had[i], pre[i], rms_weight=rms_weight)
    L = capi.lib()
    nbytes = L.quip_e8p_planes_bytes(2 * n if resid_scale != 0.0 else n)
    outs = [_empty(nbytes, dtype=torch.uint8, device=x.device) for _ in range(count)]
    arr = (capi.HadProblem * count)()
    per_row = xc.shape[0] == count and count > 1
    gptr = _vec_ok(gate, x.device)
    for i in range(count):
        off = 2 * xc.shape[1] * i if per_row else 0
        arr[i] = capi.HadProblem(xc.data_ptr() + off, outs[i].data_ptr(), _vec_ok(had[i], x.device),
                                 _vec_ok(pre[i], x.device), None, None, None, None, _vec_ok(rms_weight, x.device),
                                 None if gptr is None else gptr + off, xc.shape[1], n, float(scale[i]), float(rms_eps),
                                 None, None, None, None, 1.0, *_layout(resid_scale))
    with torch.cuda.device(x.device):
        capi.check(L.quip_had_transform_planes_group(arr, count, n, K, int(bool(transpose)), _stream(x)),
                   "quip_had_transform_planes_group")
    return outs


def _had_chain_planes_group_cuda(z, z_post, z_residual, z_scale, n, pre, scale, rms_weight, rms_eps, resid_scale=0.0):
    zc = _chk_x(z)
    count = len(pre)
    _need(zc.shape == (1, n), "had_chain_planes_group is the bs=1 path: z must be (1, n)")
    _need(1 <= count <= capi.MAX_GROUP and len(scale) == count, "group of 1..3 problems")
    _need(z_residual is None or tuple(z_residual.shape) == (1, n), "residual shape")
    for i in range(count):
        _chk_lens("had_chain_planes_group", n, n, n, 1, None, pre[i], post=z_post, rms_weight=rms_weight)
    L = capi.lib()
    nbytes = L.quip_e8p_planes_bytes(2 * n if resid_scale != 0.0 else n)
    outs = [_empty(nbytes, dtype=torch.uint8, device=z.device) for _ in range(count)]
    h = _empty((1, n), dtype=torch.float16, device=z.device)
    arr = (capi.HadProblem * count)()
    for i in range(count):
        arr[i] = capi.HadProblem(None, outs[i].data_ptr(), None, _vec_ok(pre[i], z.device), None, None, None, None,
                                 _vec_ok(rms_weight, z.device), None, n, n, float(scale[i]), float(rms_eps),
                                 zc.data_ptr(), _vec_ok(z_post, z.device), _vec_ok(z_residual, z.device),
                                 h.data_ptr(), float(z_scale), *_layout(resid_scale))
    with torch.cuda.device(z.device):
        capi.check(L.quip_had_transform_planes_group(arr, count, n, 1, 1, _stream(z)),
                   "quip_had_transform_planes_group (chain)")
    return [h] + outs


def _had_transform_group_cuda(x, out_features, n, K, had, transpose, pre2, post, bias, scale, residual, pre,
                              rms_weight, rms_eps, ns=None):
    count = len(x)
    _need(1 <= count <= capi.MAX_GROUP and all(len(v) == count for v in (out_features, had, pre2, post, bias, scale,
                                                                          residual, pre)), "group of 1..3 problems")
    xs = [_chk_x(t) for t in x]
    rows = xs[0].shape[0]
    _need(all(t.shape[0] == rows and t.device == xs[0].device for t in xs), "group inputs must share rows / device")
    _need(ns is not None or all(t.shape == xs[0].shape for t in xs), "group inputs must share a shape (or pass ns)")
    _need(ns is None or (len(ns) == count and K == 1 and rms_weight is None),
          "ns: one width per problem, K == 1, no rms_weight")
    dev = xs[0].device
    outs = [_empty((rows, int(o)), dtype=torch.float16, device=dev) for o in out_features]
    arr = (capi.HadProblem * count)()
    for i in range(count):
        _need(residual[i] is None or tuple(residual[i].shape) == tuple(outs[i].shape), "residual shape")
        ni = int(ns[i]) if ns is not None else n
        _chk_lens("had_transform_group", xs[i].shape[1], int(out_features[i]), ni, K, had[i], pre[i], pre2[i], post[i],
                  bias[i], rms_weight)
        arr[i] = capi.HadProblem(xs[i].data_ptr(), outs[i].data_ptr(), _vec_ok(had[i], dev), _vec_ok(pre[i], dev),
                                 _vec_ok(pre2[i], dev), _vec_ok(post[i], dev), _vec_ok(bias[i], dev),
                                 _vec_ok(residual[i], dev), _vec_ok(rms_weight, dev), None,
                                 xs[i].shape[1], int(out_features[i]), float(scale[i]), float(rms_eps))
        if ns is not None:
            arr[i].n = int(ns[i])
    with torch.cuda.device(dev):
        capi.check(capi.lib().quip_had_transform_group_f16(arr, count, rows, n, K, int(bool(transpose)), _stream(xs[0])),
                   "quip_had_transform_group_f16")
    return outs


_GEMV_WS = {}        # (device type, index, stream handle) -> zeroed int32 workspace
_GEMV_WS_RETIRED = []   # superseded workspaces: captured hipGraphs may still hold their pointers, so they are never freed


class _RawWorkspace:
    """device memory from hipMalloc itself, not from torch's caching allocator: a buffer that is created on first use and lives
    for the whole process must not come out of whatever memory pool is current at that moment -- inside
    torch.compile(mode="reduce-overhead") (HF's generate with a static cache compiles the forward that way) that is the
    cudagraph trees' private pool, which refuses live allocations it did not hand out as outputs"""
    _hip = None

    def __init__(self, nbytes, dev):
        import ctypes
        if _RawWorkspace._hip is None:
            _RawWorkspace._hip = ctypes.CDLL("libamdhip64.so")
        hip = _RawWorkspace._hip
        p = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes))
            if rc != 0 or not p.value:
                raise RuntimeError(f"hipMalloc({nbytes}) failed: {rc}")
            rc = hip.hipMemset(p, 0, ctypes.c_size_t(nbytes))
            if rc != 0:
                raise RuntimeError(f"hipMemset failed: {rc}")
        self.ptr, self.nbytes = int(p.value), int(nbytes)

    def data_ptr(self):
        return self.ptr

    def numel(self):                 # int32 words, like the tensor this replaces
        return self.nbytes // 4

    def read(self):
        """the contents as an int32 CPU tensor (tests: a launch leaves its workspace zeroed); synchronises the device"""
        import ctypes
        out = torch.empty(self.nbytes // 4, dtype=torch.int32)
        torch.cuda.synchronize()
        rc = _RawWorkspace._hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(self.ptr), ctypes.c_size_t(self.nbytes), 2)
        if rc != 0:
            raise RuntimeError(f"hipMemcpy failed: {rc}")
        return out

    def abs(self):                   # (the tensor methods the tests use on a workspace)
        return self.read().abs()


def _gemv_workspace(dev, n_total):
    """zeroed int32 scratch for K-split GEMV launches: ONE PER (device, stream) -- two streams running such launches at
    the same time must not add into the same accumulators and arrival counters -- kept for the life of the process
    (captured hipGraphs hold its pointer); every launch leaves it zeroed.  A workspace that has to grow is replaced,
    never freed: the old one stays alive for the graphs that captured it.  Raw device memory (_RawWorkspace) unless a stream
    capture is under way (hipMalloc is not allowed there: torch's allocator, which knows the capture's pool, serves it)."""
    need = capi.lib().quip_e8p_gemv_workspace_bytes(int(n_total))
    key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _GEMV_WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        if ws is not None:
            _GEMV_WS_RETIRED.append(ws)
        nbytes = max(need, 4 << 20)
        if torch.cuda.is_current_stream_capturing():
            with torch.cuda.stream(torch.cuda.current_stream(dev)):
                ws = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)
        else:
            ws = _RawWorkspace(nbytes, dev)
        _GEMV_WS[key] = ws
    return ws


def _e8p_gemv_planes_group_cuda(planes, Qidxs, grid):
    import ctypes
    count = len(planes)
    _need(1 <= count <= capi.MAX_GROUP and len(Qidxs) == count, "group of 1..3 problems")
    k = Qidxs[0].shape[1] * 8
    dev = planes[0].device
    for pl, q in zip(planes, Qidxs):
        _need(q.dtype == torch.int16 and q.is_contiguous() and q.shape[1] * 8 == k and q.device == dev,
              "Qidxs must be contiguous int16 (n, k/8) with a common k")
        _need(pl.dtype == torch.uint8 and pl.is_contiguous() and pl.device == dev, "planes must be uint8")
    outs = [_empty((1, q.shape[0]), dtype=torch.float16, device=dev) for q in Qidxs]
    vp = ctypes.c_void_p * count
    ns = (ctypes.c_int32 * count)(*[q.shape[0] for q in Qidxs])
    g = _grid_i64(grid, planes[0])
    ws = _gemv_workspace(dev, sum(q.shape[0] for q in Qidxs))
    with torch.cuda.device(dev):
        capi.check(capi.lib().quip_e8p_gemv_planes_group_ws(
            vp(*[p.data_ptr() for p in planes]), vp(*[q.data_ptr() for q in Qidxs]), g.data_ptr(),
            vp(*[o.data_ptr() for o in outs]), ns, count, k, ws.data_ptr(), ws.numel() * 4, _stream(planes[0])),
            "quip_e8p_gemv_planes_group_ws")
    return outs


def _e8p_gemv_fused_cuda(x, z, post, residual, rms_weight, rms_eps, z_scale, pre, scale, Qidxs, grid):
    import ctypes
    count = len(Qidxs)
    _need(1 <= count <= capi.MAX_GROUP and len(pre) == count and len(scale) == count, "group of 1..3 problems")
    k = Qidxs[0].shape[1] * 8
    src = z if z is not None else x
    _need(src is not None and src.numel() == k and src.dtype == torch.float16 and src.is_contiguous(),
          "x / z must be a contiguous fp16 vector of k elements")
    dev = src.device
    for q in Qidxs:
        _need(q.dtype == torch.int16 and q.is_contiguous() and q.shape[1] * 8 == k and q.device == dev,
              "Qidxs must be contiguous int16 (n, k/8) with a common k")
    for t in [post, residual, rms_weight] + list(pre):
        _need(t is None or (t.numel() == k and t.dtype == torch.float16 and t.is_contiguous() and t.device == dev),
              "vectors must be contiguous fp16 of k elements")
    _need(z is None or post is not None, "post (the producer's SV) is required with z")
    outs = [_empty((1, q.shape[0]), dtype=torch.float16, device=dev) for q in Qidxs]
    h_out = _empty((1, k), dtype=torch.float16, device=dev) if z is not None else None
    fin = capi.GemvFusedIn()
    fin.x, fin.z, fin.post_scale, fin.residual = _ptr(x), _ptr(z), _ptr(post), _ptr(residual)
    fin.h_out, fin.rms_weight = _ptr(h_out), _ptr(rms_weight)
    for i in range(count):
        fin.pre_scale[i] = pre[i].data_ptr()
        fin.scale[i] = float(scale[i])
    fin.z_scale, fin.rms_eps = float(z_scale), float(rms_eps)
    vp = ctypes.c_void_p * count
    ns = (ctypes.c_int32 * count)(*[q.shape[0] for q in Qidxs])
    g = _grid_i64(grid, src)
    with torch.cuda.device(dev):
        capi.check(capi.lib().quip_e8p_gemv_fused(
            ctypes.byref(fin), vp(*[q.data_ptr() for q in Qidxs]), g.data_ptr(),
            vp(*[o.data_ptr() for o in outs]), ns, count, k, _stream(src)), "quip_e8p_gemv_fused")
    return ([h_out] if h_out is not None else []) + outs


def ffn_engine_supported(hidden, n_ffn, K):
    return bool(capi.lib().quip_ffn_engine_supported(int(hidden), int(n_ffn), int(K)))


def ffn_engine_workspace(n_ffn, K, device):
    """hand-off area of the engine launches of one decoder (zeroed once; launches sharing it must be stream ordered)"""
    return torch.zeros(capi.lib().quip_ffn_engine_workspace_bytes(int(n_ffn), int(K)), dtype=torch.uint8, device=device)


def ffn_engine_status(workspace):
    """0, or the code of the wait that gave up in an engine launch on this workspace (synchronises)"""
    return int(workspace[:8].view(torch.int32)[1].item())


def _ffn_engine_cuda(planes_gate, planes_up, q_gate, q_up, q_down, had3, sv_gate, sv_up, su_down, grid, workspace,
                     out_scale, in_scale, K, dbg=None):
    dev = planes_gate.device
    n_ffn, hidden = q_gate.shape[0], q_gate.shape[1] * 8
    for q in (q_gate, q_up, q_down):
        _need(q.dtype == torch.int16 and q.is_contiguous() and q.device == dev, "Qidxs must be contiguous int16")
    _need(q_up.shape == q_gate.shape and q_down.shape == (hidden, n_ffn // 8), "gate / up (n_ffn, hidden / 8), down (hidden, n_ffn / 8)")
    _need(ffn_engine_supported(hidden, n_ffn, K), f"ffn_engine: shape (hidden {hidden}, n_ffn {n_ffn}, K {K}) not supported")
    kp = (hidden + 511) // 512 * 512
    for pl in (planes_gate, planes_up):
        _need(pl.dtype == torch.uint8 and pl.is_contiguous() and pl.numel() == 3 * kp + 16 and pl.device == dev,
              "planes must be uint8 images of 3 Kp + 16 bytes")
    kkp, kp16 = (K * K + 7) // 8 * 8, (K + 15) // 16 * 16
    _need(had3.dtype == torch.float16 and had3.is_contiguous() and had3.numel() == 2 * kkp + kp16 * kp16
          and had3.device == dev, "had3 must be the fp16 pack of qlinear._engine_had3")
    for v in (sv_gate, sv_up, su_down):
        _need(v.dtype == torch.float16 and v.is_contiguous() and v.numel() == n_ffn and v.device == dev,
              "sv_gate / sv_up / su_down must be fp16 vectors of n_ffn elements")
    _need(workspace.dtype == torch.uint8 and workspace.device == dev
          and workspace.numel() >= capi.lib().quip_ffn_engine_workspace_bytes(n_ffn, K), "workspace too small")
    g = _grid_i64(grid, planes_gate)
    out = _empty((1, hidden), dtype=torch.float16, device=dev)
    a = capi.FfnEngineArgs(q_gate.data_ptr(), q_up.data_ptr(), q_down.data_ptr(), planes_gate.data_ptr(),
                           planes_up.data_ptr(), had3.data_ptr(), sv_gate.data_ptr(), sv_up.data_ptr(),
                           su_down.data_ptr(), out.data_ptr(), g.data_ptr(), workspace.data_ptr(), _ptr(dbg),
                           float(out_scale), float(in_scale), hidden, n_ffn, int(K))
    import ctypes
    with torch.cuda.device(dev):
        capi.check(capi.lib().quip_ffn_engine(ctypes.byref(a), _stream(planes_gate)), "quip_ffn_engine")
    return out


def block_engine_supported(hidden, heads, kv_heads, head_dim, n_ffn, K):
    return bool(capi.lib().quip_block_engine_supported(int(hidden), int(heads), int(kv_heads), int(head_dim), int(n_ffn), int(K)))


def _block_engine_ws_bytes(shape):
    L = capi.lib()
    return (L.quip_block_engine_gqa_workspace_bytes() if shape == 1 else
            L.quip_block_engine_g8_workspace_bytes() if shape == 2 else L.quip_block_engine_workspace_bytes())


def block_engine_workspace(device, shape=0):
    return torch.zeros(_block_engine_ws_bytes(shape), dtype=torch.uint8, device=device)


def block_engine_g8_supported(hidden, heads, kv_heads, head_dim, n_ffn, K):
    """the 4096-wide grouped-query shape of Llama-3-8B / Mistral-7B (32 / 8 heads of 128, n_ffn 14336, K = 7) on the persistent launch"""
    return bool(capi.lib().quip_block_engine_g8_supported(int(hidden), int(heads), int(kv_heads), int(head_dim), int(n_ffn), int(K)))


def block_engine_gqa_supported(hidden, heads, kv_heads, head_dim, n_ffn, K):
    return bool(capi.lib().quip_block_engine_gqa_supported(int(hidden), int(heads), int(kv_heads), int(head_dim), int(n_ffn), int(K)))


def _block_engine_cuda(layers, h_in, pos, cos, sin, grid, workspace, n_layers, max_len, rms_eps, attn_scale, dbg=None,
                       dbg_layer=-1, codebook=0, resid_scale=0.0, shape=0, grid2=None, kcache=None, vcache=None):
    dev = h_in.device
    lb = capi.lib().quip_block_engine_layer_bytes()
    _need(layers.dtype == torch.uint8 and layers.is_contiguous() and layers.numel() >= n_layers * lb and layers.device == dev,
          "layers must be the packed descriptors (uint8, n_layers x 256 bytes) on the device")
    _need(shape in (0, 1, 2), "shape: 0 (hidden 4096, multi-head), 1 (hidden 8192, grouped-query) or 2 (hidden 4096, grouped-query)")
    hid = 8192 if shape == 1 else 4096
    _need(h_in.dtype == torch.float16 and h_in.is_contiguous() and h_in.numel() == hid, f"h_in: fp16 [{hid}]")
    _need(pos.dtype == torch.int64 and pos.numel() == 1 and pos.device == dev, "pos: int64 device scalar")
    for t in (cos, sin):
        _need(t.dtype == torch.float32 and t.is_contiguous() and t.shape == (max_len, 128) and t.device == dev,
              "cos / sin: fp32 [max_len, 128]")
    need_ws = _block_engine_ws_bytes(shape)
    _need(workspace.dtype == torch.uint8 and workspace.device == dev and workspace.numel() >= need_ws, "workspace too small")
    if codebook in (1, 3):      # D4 / HI: the fp16 (256, 4) table (HI: of a code byte) -- 2 KB like grid_packed_abs
        g = _d4_grid_f16(grid)
        _need(g.device == dev and g.numel() == 1024, "D4 / HI grid: fp16 (256, 4) on the device")
    else:
        g = _grid_i64(grid, h_in)
    if codebook == 4:           # E8P12RVQ3B: the E81B table as int8 (256, 8) = 4 r
        _need(grid2 is not None and grid2.dtype == torch.int8 and grid2.is_contiguous() and grid2.numel() == 2048 and grid2.device == dev,
              "grid2: the E81B table as int8 (256, 8) on the device")
    out = torch.empty_like(h_in)
    a = capi.BlockEngineArgs(layers.data_ptr(), h_in.data_ptr(), out.data_ptr(), pos.data_ptr(), cos.data_ptr(),
                             sin.data_ptr(), g.data_ptr(), workspace.data_ptr(), _ptr(dbg), int(n_layers), int(max_len),
                             int(dbg_layer), float(rms_eps), float(attn_scale), int(codebook), float(resid_scale), int(shape),
                             _ptr(grid2) if codebook == 4 else None)
    import ctypes
    with torch.cuda.device(dev):
        rc = capi.lib().quip_block_engine(ctypes.byref(a), _stream(h_in))
        # (shape 1, dbg_layer == -2: the measurement mode answers QUIP_NO_RESULT = 1 -- the caller asked for exactly that)
        if not (rc == 1 and shape == 1 and int(dbg_layer) == -2):
            capi.check(rc, "quip_block_engine")
    return out


def rope_attn_workspace(heads, head_dim, device):
    """zeroed scratch for the split (long context) mode of rope_attn_decode; allocate once, reuse"""
    return torch.zeros(capi.lib().quip_rope_attn_workspace_bytes(heads, head_dim), dtype=torch.uint8, device=device)


def _rope_attn_decode_cuda(q, k, v, cos, sin, pos, kcache, vcache, workspace=None, window=0):
    """q (heads, hd), k / v (kv_heads, hd) fp16; cos / sin (max_len, hd) fp32; pos int64 device scalar;
    kcache / vcache (kv_heads, max_len, hd) fp16 (row pos is written) -> (heads, hd) fp16"""
    import math
    for t in (q, k, v, kcache, vcache):
        _need(t.dtype == torch.float16 and t.is_contiguous() and t.is_cuda, "rope_attn_decode: fp16 contiguous CUDA tensors")
    _need(cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous(),
          "cos / sin must be contiguous float32")
    _need(pos.dtype == torch.int64 and pos.numel() == 1 and pos.is_cuda, "pos must be an int64 device scalar")
    heads, hd = q.shape
    kvh, max_len = kcache.shape[0], kcache.shape[1]
    _need(tuple(k.shape) == (kvh, hd) and tuple(v.shape) == (kvh, hd) and tuple(vcache.shape) == tuple(kcache.shape)
          and kcache.shape[2] == hd and tuple(cos.shape) == (max_len, hd) and tuple(sin.shape) == (max_len, hd),
          "rope_attn_decode: shape mismatch")
    out = torch.empty_like(q)
    if workspace is not None:
        _need(workspace.dtype == torch.uint8 and workspace.is_contiguous() and workspace.device == q.device
              and workspace.numel() >= capi.lib().quip_rope_attn_workspace_bytes(heads, hd),
              "workspace: use rope_attn_workspace(heads, head_dim, device)")
    with torch.cuda.device(q.device):
        capi.check(capi.lib().quip_rope_attn_decode_window_f16(
            q.data_ptr(), k.data_ptr(), v.data_ptr(), cos.data_ptr(), sin.data_ptr(), pos.data_ptr(),
            kcache.data_ptr(), vcache.data_ptr(), out.data_ptr(), heads, kvh, hd, max_len, 1.0 / math.sqrt(hd),
            int(window), _ptr(workspace), _stream(q)), "quip_rope_attn_decode_window_f16")
    return out


def rope_attn_decode_z_supported(heads, kv_heads, head_dim):
    return bool(capi.lib().quip_rope_attn_decode_z_supported(heads, kv_heads, head_dim))


def _rope_attn_decode_z_cuda(zs, posts, scales, cos, sin, pos, kcache, vcache, workspace=None, window=0):
    """zs / posts: the raw GEMV outputs (1, n) or (n,) and SV vectors (n,) of q / k / v_proj, fp16; the rest as
    rope_attn_decode -> (heads, hd) fp16"""
    import ctypes
    import math
    _need(len(zs) == 3 and len(posts) == 3 and len(scales) == 3, "rope_attn_decode_z: q, k, v")
    kvh, max_len, hd = kcache.shape
    n = zs[0].numel()
    heads = n // hd
    for i, t in enumerate(list(zs) + list(posts)):
        _need(t.dtype == torch.float16 and t.is_contiguous() and t.is_cuda and t.numel() == (n if i % 3 == 0 else kvh * hd),
              "rope_attn_decode_z: fp16 contiguous CUDA vectors, heads * head_dim (q) / kv_heads * head_dim (k, v) long")
    for t in (kcache, vcache):
        _need(t.dtype == torch.float16 and t.is_contiguous() and t.is_cuda, "rope_attn_decode_z: fp16 caches")
    _need(heads * hd == n and rope_attn_decode_z_supported(heads, kvh, hd),
          "rope_attn_decode_z: needs heads == kv_heads and heads * head_dim a power of two in 256..4096, or 64 / 32 heads "
          "of 128 on 8 KV heads")
    _need(cos.dtype == torch.float32 and sin.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
          and tuple(cos.shape) == (max_len, hd) and tuple(sin.shape) == (max_len, hd), "cos / sin: float32 (max_len, hd)")
    _need(pos.dtype == torch.int64 and pos.numel() == 1 and pos.is_cuda, "pos must be an int64 device scalar")
    _need(tuple(vcache.shape) == tuple(kcache.shape), "cache shapes differ")
    out = _empty((heads, hd), dtype=torch.float16, device=kcache.device)
    if workspace is not None:
        _need(workspace.dtype == torch.uint8 and workspace.is_contiguous() and workspace.device == kcache.device
              and workspace.numel() >= capi.lib().quip_rope_attn_workspace_bytes(heads, hd),
              "workspace: use rope_attn_workspace(heads, head_dim, device)")
    zp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in zs])
    pp = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in posts])
    sc = (ctypes.c_float * 3)(*[float(x) for x in scales])
    with torch.cuda.device(kcache.device):
        capi.check(capi.lib().quip_rope_attn_decode_z_window_f16(
            zp, pp, sc, cos.data_ptr(), sin.data_ptr(), pos.data_ptr(), kcache.data_ptr(), vcache.data_ptr(),
            out.data_ptr(), heads, kvh, hd, max_len, 1.0 / math.sqrt(hd), int(window), _ptr(workspace), _stream(kcache)),
            "quip_rope_attn_decode_z_window_f16")
    return out


def e8p_mm_batched_supported(m, n, k):
    """shapes the fused batched product takes (quip_e8p_mm_batched)"""
    return m >= 1 and n >= 2 and n % 2 == 0 and k >= 64 and k % 64 == 0


def _e8p_mm_batched_cuda(x, Qidxs, grid):
    g = _grid_i64(grid, x)
    xc = _chk_x(x)
    Qc = _chk_q(Qidxs, torch.int16)
    m, k, n = xc.shape[0], xc.shape[1], Qc.shape[0]
    _need(Qc.shape[1] * 8 == k, f"e8p_mm_batched: x has {k} columns but Qidxs {tuple(Qidxs.shape)} encodes {Qc.shape[1] * 8}")
    _need(Qc.device == x.device, "Qidxs and x must be on the same device")
    _need(e8p_mm_batched_supported(m, n, k), f"e8p_mm_batched: shape ({m}, {n}, {k}) needs k % 64 == 0 and n % 2 == 0")
    y = _empty((m, n), dtype=torch.float16, device=x.device)
    if m == 0:
        return y
    with torch.cuda.device(x.device):
        capi.check(capi.lib().quip_e8p_mm_batched(xc.data_ptr(), Qc.data_ptr(), g.data_ptr(), y.data_ptr(), m, n, k,
                                                  _stream(x)), "quip_e8p_mm_batched")
    return y


def e8p_mm_skinny_supported(m, n, k):
    """(rows beyond 32 run as chunks of 32 in the same launch)"""
    return 1 <= m <= 32 * 65535 and n >= 2 and n % 2 == 0 and k >= 128 and k % 128 == 0


def _e8p_mm_skinny_cuda(x, Qidxs, grid):
    g = _grid_i64(grid, x)
    xc = _chk_x(x)
    Qc = _chk_q(Qidxs, torch.int16)
    m, k, n = xc.shape[0], xc.shape[1], Qc.shape[0]
    _need(Qc.shape[1] * 8 == k, f"e8p_mm_skinny: x has {k} columns but Qidxs {tuple(Qidxs.shape)} encodes {Qc.shape[1] * 8}")
    _need(Qc.device == x.device, "Qidxs and x must be on the same device")
    _need(e8p_mm_skinny_supported(m, n, k), f"e8p_mm_skinny: shape ({m}, {n}, {k}) needs k % 128 == 0, n % 2 == 0")
    y = _empty((m, n), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().quip_e8p_mm_skinny(xc.data_ptr(), Qc.data_ptr(), g.data_ptr(), y.data_ptr(), m, n, k,
                                                 _stream(x)), "quip_e8p_mm_skinny")
    return y


def _e8prvq4_mm_skinny_cuda(x, Qidxs, grid, scale):
    g = _grid_i64(grid, x)
    xc = _chk_x(x)
    Qc = _chk_q(Qidxs, torch.int32)
    m, k, n = xc.shape[0], xc.shape[1], Qc.shape[0]
    _need(Qc.shape[1] * 8 == k, f"e8prvq4_mm_skinny: x has {k} columns but Qidxs {tuple(Qidxs.shape)} encodes {Qc.shape[1] * 8}")
    _need(Qc.device == x.device, "Qidxs and x must be on the same device")
    _need(e8p_mm_skinny_supported(m, n, k), f"e8prvq4_mm_skinny: shape ({m}, {n}, {k}) needs k % 128 == 0, n % 2 == 0")
    y = _empty((m, n), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().quip_e8prvq4_mm_skinny(xc.data_ptr(), Qc.data_ptr(), g.data_ptr(), float(scale), y.data_ptr(),
                                                     m, n, k, _stream(x)), "quip_e8prvq4_mm_skinny")
    return y


def _skinny_generic(fn, what, x, Qidxs, qdtype, per_code, extra):
    xc = _chk_x(x)
    Qc = _chk_q(Qidxs, qdtype)
    m, k, n = xc.shape[0], xc.shape[1], Qc.shape[0]
    _need(Qc.shape[1] * per_code == k, f"{what}: x has {k} columns but Qidxs {tuple(Qidxs.shape)} encodes {Qc.shape[1] * per_code}")
    _need(Qc.device == x.device, "Qidxs and x must be on the same device")
    _need(e8p_mm_skinny_supported(m, n, k), f"{what}: shape ({m}, {n}, {k}) needs k % 128 == 0, n % 2 == 0")
    y = _empty((m, n), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        capi.check(getattr(capi.lib(), fn)(xc.data_ptr(), Qc.data_ptr(), *extra(), y.data_ptr(), m, n, k, _stream(x)), fn)
    return y


def _e8prvq3_mm_skinny_cuda(x, Qidxs, grid, grid2, scale):
    g = _grid_i64(grid, x)
    _need(grid2.dtype == torch.int32 and grid2.numel() == 256, "e81b_grid_packed must be int32[256]")
    g2 = grid2.contiguous()
    xc = _chk_x(x)
    Qc = _chk_q(Qidxs, torch.int32)
    m, k, n = xc.shape[0], xc.shape[1], Qc.shape[0]
    _need(Qc.shape[1] * 32 == 3 * k, f"e8prvq3_mm_skinny: x has {k} columns but Qidxs {tuple(Qidxs.shape)} encodes {Qc.shape[1] * 32 // 3}")
    _need(Qc.device == x.device and g2.device == x.device, "Qidxs, grid2 and x must be on the same device")
    _need(e8p_mm_skinny_supported(m, n, k), f"e8prvq3_mm_skinny: shape ({m}, {n}, {k}) needs k % 128 == 0, n % 2 == 0")
    y = _empty((m, n), dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        capi.check(capi.lib().quip_e8prvq3_mm_skinny(xc.data_ptr(), Qc.data_ptr(), g.data_ptr(), g2.data_ptr(), float(scale),
                                                     y.data_ptr(), m, n, k, _stream(x)), "quip_e8prvq3_mm_skinny")
    return y


def _batched_generic(fn, what, x, Qidxs, qdtype, k_of_cols, extra):
    """the fused tile kernel in another codebook's mode: x (M, k) fp16, any M >= 0"""
    xc = _chk_x(x)
    Qc = _chk_q(Qidxs, qdtype)
    m, k, n = xc.shape[0], xc.shape[1], Qc.shape[0]
    _need(k_of_cols(Qc.shape[1]) == k, f"{what}: x has {k} columns but Qidxs {tuple(Qidxs.shape)} encodes {k_of_cols(Qc.shape[1])}")
    _need(Qc.device == x.device, "Qidxs and x must be on the same device")
    _need(e8p_mm_batched_supported(max(m, 1), n, k), f"{what}: shape ({m}, {n}, {k}) needs k % 64 == 0 and n % 2 == 0")
    y = _empty((m, n), dtype=torch.float16, device=x.device)
    if m == 0:
        return y
    with torch.cuda.device(x.device):
        capi.check(getattr(capi.lib(), fn)(xc.data_ptr(), Qc.data_ptr(), *extra(), y.data_ptr(), m, n, k, _stream(x)), fn)
    return y


def _e8prvq4_mm_batched_cuda(x, Qidxs, grid, scale):
    g = _grid_i64(grid, x)
    return _batched_generic("quip_e8prvq4_mm_batched", "e8prvq4_mm_batched", x, Qidxs, torch.int32, lambda c: c * 8,
                            lambda: (g.data_ptr(), float(scale)))


def _e8prvq3_mm_batched_cuda(x, Qidxs, grid, grid2, scale):
    g = _grid_i64(grid, x)
    _need(grid2.dtype == torch.int32 and grid2.numel() == 256, "e81b_grid_packed must be int32[256]")
    g2 = grid2.contiguous()
    _need(g2.device == x.device, "grid2 and x must be on the same device")
    return _batched_generic("quip_e8prvq3_mm_batched", "e8prvq3_mm_batched", x, Qidxs, torch.int32, lambda c: c * 32 // 3,
                            lambda: (g.data_ptr(), g2.data_ptr(), float(scale)))


def _d4_mm_batched_cuda(x, Qidxs, grid):
    g = _d4_grid_f16(grid)
    return _batched_generic("quip_d4_mm_batched", "d4_mm_batched", x, Qidxs, torch.uint8, lambda c: c * 4, lambda: (g.data_ptr(),))


def _hi_mm_batched_cuda(x, Qidxs):
    return _batched_generic("quip_hi_mm_batched", "hi_mm_batched", x, Qidxs, torch.int32, lambda c: c * 8, lambda: ())


def _d4_mm_skinny_cuda(x, Qidxs, grid):
    g = _d4_grid_f16(grid)
    return _skinny_generic("quip_d4_mm_skinny", "d4_mm_skinny", x, Qidxs, torch.uint8, 4, lambda: (g.data_ptr(),))


def _hi_mm_skinny_cuda(x, Qidxs):
    return _skinny_generic("quip_hi_mm_skinny", "hi_mm_skinny", x, Qidxs, torch.int32, 8, lambda: ())


def _e8p_gemv_planes_cuda(planes, Qidxs, grid):
    g = _grid_i64(grid, Qidxs)
    Qc = _chk_q(Qidxs, torch.int16)
    n, k = Qc.shape[0], Qc.shape[1] * 8
    L = capi.lib()
    _need(planes.dtype == torch.uint8 and planes.numel() >= L.quip_e8p_planes_bytes(k), "planes buffer too small")
    y = _empty((1, n), dtype=torch.float16, device=Qidxs.device)
    ws = _gemv_workspace(Qidxs.device, n)
    with torch.cuda.device(Qidxs.device):
        capi.check(L.quip_e8p_gemv_planes_ws(planes.data_ptr(), Qc.data_ptr(), g.data_ptr(), y.data_ptr(), n, k,
                                             ws.data_ptr(), ws.numel() * 4, _stream(Qidxs)), "quip_e8p_gemv_planes_ws")
    return y


# ---- mm ops -------------------------------------------------------------------------
def _mm(fn_name, x, Q, qdtype, k_per_col_num, k_per_col_den, extra):
    xc = _chk_x(x)
    Qc = _chk_q(Q, qdtype)
    k = xc.shape[1]
    _need(Qc.shape[1] * k_per_col_num == k * k_per_col_den,
          f"{fn_name}: x has {k} columns but Qidxs {tuple(Q.shape)} encodes {Qc.shape[1] * k_per_col_num // k_per_col_den}")
    _need(Qc.device == x.device, "Qidxs and x must be on the same device")
    y = _empty((xc.shape[0], Qc.shape[0]), dtype=x.dtype, device=x.device)
    if y.numel() == 0:
        return y
    with torch.cuda.device(x.device):
        fn = getattr(capi.lib(), fn_name)
        capi.check(fn(xc.data_ptr(), Qc.data_ptr(), *extra(), y.data_ptr(), xc.shape[0], Qc.shape[0],
                      k, _stream(x)), fn_name)
    return y


def _grid_i64(grid: Tensor, x: Tensor):
    _need(grid.dtype == torch.int64 and grid.numel() == 256, "grid_packed_abs must be int64[256]")
    _need(grid.device == x.device, "grid and x must be on the same device")
    return grid.contiguous()


def _e8p_mm_cuda(x, Qidxs, grid):
    g = _grid_i64(grid, x)
    xc = _chk_x(x)
    Qc = _chk_q(Qidxs, torch.int16)
    m, k, n = xc.shape[0], xc.shape[1], Qc.shape[0]
    _need(Qc.shape[1] * 8 == k, f"e8p_mm: x has {k} columns but Qidxs {tuple(Qidxs.shape)} encodes {Qc.shape[1] * 8}")
    _need(Qc.device == x.device, "Qidxs and x must be on the same device")
    y = _empty((m, n), dtype=x.dtype, device=x.device)
    if y.numel() == 0:
        return y
    L = capi.lib()
    ws_bytes = L.quip_e8p_mm_workspace_bytes(m, n, k)
    ws = _empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
    with torch.cuda.device(x.device):
        capi.check(L.quip_e8p_mm_origorder_ws(xc.data_ptr(), Qc.data_ptr(), g.data_ptr(), y.data_ptr(), m, n, k,
                                              _ptr(ws), ws_bytes, _stream(x)), "quip_e8p_mm_origorder_ws")
    return y


def _e8prvq3_mm_cuda(x, Qidxs, grid, grid2, scale):
    g = _grid_i64(grid, x)
    _need(grid2.dtype == torch.int32 and grid2.numel() == 256, "e81b_grid_packed must be int32[256]")
    g2 = grid2.contiguous()
    return _mm("quip_e8prvq3_mm_origorder", x, Qidxs, torch.int32, 32, 3,
               lambda: (g.data_ptr(), g2.data_ptr(), float(scale)))


def _e8prvq4_mm_cuda(x, Qidxs, grid, scale):
    g = _grid_i64(grid, x)
    return _mm("quip_e8prvq4_mm_origorder", x, Qidxs, torch.int32, 8, 1,
               lambda: (g.data_ptr(), float(scale)))


def _d4_grid_f16(grid: Tensor):
    # the reference kernel reinterprets `grid` as uint64[256], i.e. it silently
    # requires fp16 (origin_order.cu:794-805); cast explicitly here.
    _need(grid.numel() == 1024, "D4 grid must be (256, 4)")
    return grid.to(torch.float16).contiguous()


def _d4_mm_cuda(x, Qidxs, grid):
    g = _d4_grid_f16(grid)
    return _mm("quip_d4_mm_origorder", x, Qidxs, torch.uint8, 4, 1, lambda: (g.data_ptr(),))


def _hi_mm_cuda(x, Qidxs):
    return _mm("quip_hi_mm_origorder", x, Qidxs, torch.int32, 8, 1, lambda: ())


# ---- decompress ops --------------------------------------------------------------------
def _dec(fn_name, Q, qdtype, k, extra):
    Qc = _chk_q(Q, qdtype)
    w = _empty((Qc.shape[0], k), dtype=torch.float16, device=Q.device)
    with torch.cuda.device(Q.device):
        fn = getattr(capi.lib(), fn_name)
        capi.check(fn(Qc.data_ptr(), *extra(), w.data_ptr(), Qc.shape[0], k, _stream(Q)), fn_name)
    return w


def _dec_e8p_cuda(Qidxs, grid):
    g = _grid_i64(grid, Qidxs)
    return _dec("quip_decompress_e8p_origorder", Qidxs, torch.int16, Qidxs.shape[1] * 8,
                lambda: (g.data_ptr(),))


def _dec_e8prvq3_cuda(Qidxs, grid, grid2, scale):
    g = _grid_i64(grid, Qidxs)
    g2 = grid2.contiguous()
    _need(g2.dtype == torch.int32 and g2.numel() == 256, "e81b_grid_packed must be int32[256]")
    return _dec("quip_decompress_e8prvq3_origorder", Qidxs, torch.int32, Qidxs.shape[1] * 32 // 3,
                lambda: (g.data_ptr(), g2.data_ptr(), float(scale)))


def _dec_e8prvq4_cuda(Qidxs, grid, scale):
    g = _grid_i64(grid, Qidxs)
    return _dec("quip_decompress_e8prvq4_origorder", Qidxs, torch.int32, Qidxs.shape[1] * 8,
                lambda: (g.data_ptr(), float(scale)))


def _dec_d4_cuda(Qidxs, grid):
    g = _d4_grid_f16(grid)
    return _dec("quip_decompress_d4_origorder", Qidxs, torch.uint8, Qidxs.shape[1] * 4,
                lambda: (g.data_ptr(),))


def _dec_hi_cuda(Qidxs):
    return _dec("quip_decompress_hi_origorder", Qidxs, torch.int32, Qidxs.shape[1] * 8, lambda: ())


_IMPLS = {
    "hadamard": _hadamard_cuda,
    "had_transform": _had_transform_cuda,
    "had_transform_planes": _had_transform_planes_cuda,
    "e8p_gemv_planes": _e8p_gemv_planes_cuda,
    "rope_attn_decode": _rope_attn_decode_cuda,
    "ffn_engine": _ffn_engine_cuda,
    "block_engine": _block_engine_cuda,
    "rope_attn_decode_z": _rope_attn_decode_z_cuda,
    "e8p_gemv_fused": _e8p_gemv_fused_cuda,
    "had_transform_planes_group": _had_transform_planes_group_cuda,
    "had_chain_planes_group": _had_chain_planes_group_cuda,
    "had_transform_group": _had_transform_group_cuda,
    "e8p_gemv_planes_group": _e8p_gemv_planes_group_cuda,
    "e8p_mm_batched": _e8p_mm_batched_cuda,
    "e8prvq4_mm_batched": _e8prvq4_mm_batched_cuda,
    "e8prvq3_mm_batched": _e8prvq3_mm_batched_cuda,
    "d4_mm_batched": _d4_mm_batched_cuda,
    "hi_mm_batched": _hi_mm_batched_cuda,
    "e8p_mm_skinny": _e8p_mm_skinny_cuda,
    "e8prvq4_mm_skinny": _e8prvq4_mm_skinny_cuda,
    "e8prvq3_mm_skinny": _e8prvq3_mm_skinny_cuda,
    "d4_mm_skinny": _d4_mm_skinny_cuda,
    "hi_mm_skinny": _hi_mm_skinny_cuda,
    "e8p_mm_planes_rows": _e8p_mm_planes_rows_cuda,
    "had_transform_planes_rows": _had_transform_planes_rows_cuda,
    "e8p_gemv_planes_rows": _e8p_gemv_planes_rows_cuda,
    "e8p_quantize": _e8p_quantize_cuda,
    "argmax_step": _argmax_step_cuda,
    "gemv_planes_rows_mode": _gemv_planes_rows_mode_cuda,
    "e8prvq3_gemv_planes_group": _e8prvq3_gemv_planes_group_cuda,
    "d4_gemv_planes": _d4_gemv_planes_cuda,
    "d4_gemv_planes_group": _d4_gemv_planes_group_cuda,
    "had_transform_fused": _had_transform_fused_cuda,
    "had_transform_planes_fused": _had_transform_planes_fused_cuda,
    "e8p_mm_origorder": _e8p_mm_cuda,
    "e8prvq3_mm_origorder": _e8prvq3_mm_cuda,
    "e8prvq4_mm_origorder": _e8prvq4_mm_cuda,
    "d4_mm_origorder": _d4_mm_cuda,
    "hi_mm_origorder": _hi_mm_cuda,
    "decompress_e8p_origorder": _dec_e8p_cuda,
    "decompress_e8prvq3_origorder": _dec_e8prvq3_cuda,
    "decompress_e8prvq4_origorder": _dec_e8prvq4_cuda,
    "decompress_d4_origorder": _dec_d4_cuda,
    "decompress_hi_origorder": _dec_hi_cuda,
}
for _name, _fn in _IMPLS.items():
    _lib.impl(_name, _fn, "CUDA")


# ---- fake (meta) implementations: shapes only, for torch.compile / export -----------------
def _fake_mm(x, Qidxs, *a):
    return x.new_empty((x.shape[0], Qidxs.shape[0]))


def _reg_fake(name, fn):
    torch.library.register_fake("quip_lib::" + name, fn, lib=_lib)


_reg_fake("hadamard", lambda x, scale: torch.empty_like(x, memory_format=torch.contiguous_format))
_reg_fake("had_transform", lambda x, out_features, n, K, had, transpose, pre, pre2, post, bias, scale:
          x.new_empty((x.shape[0], out_features)))
_reg_fake("had_transform_planes", lambda x, n, K, had, transpose, pre, scale:
          x.new_empty((3 * ((n + 511) // 512 * 512) + 16,), dtype=torch.uint8))
_reg_fake("had_transform_fused", lambda x, out_features, n, K, had, transpose, pre, pre2, post, bias, scale, residual,
          rms_weight, rms_eps, gate: x.new_empty((x.shape[0], out_features)))
HI_PLANES = float("inf")     # resid_scale value that selects the HI virtual-vector layout of the planes ops


def _layout(resid_scale):
    """(resid_scale, planes_layout) of a quip_had_problem from the ops' single float argument"""
    return (0.0, 2) if resid_scale == HI_PLANES else (float(resid_scale), 0)


def _planes_numel(n, resid_scale):
    m = 2 * n if resid_scale != 0.0 else n
    return 3 * ((m + 511) // 512 * 512) + 16


_reg_fake("had_transform_planes_fused", lambda x, n, K, had, transpose, pre, scale, rms_weight, rms_eps, gate,
          resid_scale=0.0: x.new_empty((_planes_numel(n, resid_scale),), dtype=torch.uint8))
_reg_fake("had_transform_planes_group", lambda x, n, K, had, transpose, pre, scale, rms_weight, rms_eps, gate,
          resid_scale=0.0: [x.new_empty((_planes_numel(n, resid_scale),), dtype=torch.uint8) for _ in pre])
_reg_fake("had_chain_planes_group", lambda z, z_post, z_residual, z_scale, n, pre, scale, rms_weight, rms_eps,
          resid_scale=0.0: [z.new_empty((1, n))] + [z.new_empty((_planes_numel(n, resid_scale),), dtype=torch.uint8)
                                                     for _ in pre])
_reg_fake("had_transform_group", lambda x, out_features, n, K, had, transpose, pre2, post, bias, scale, residual, pre,
          rms_weight, rms_eps, ns=None:
          [t.new_empty((t.shape[0], int(o))) for t, o in zip(x, out_features)])
_reg_fake("e8prvq3_gemv_planes_group", lambda planes, Qidxs, grid, e81b_i8:
          [q.new_empty((1, q.shape[0]), dtype=torch.float16) for q in Qidxs])
_reg_fake("d4_gemv_planes", lambda planes, Qidxs, grid: Qidxs.new_empty((1, Qidxs.shape[0]), dtype=torch.float16))
_reg_fake("d4_gemv_planes_group", lambda planes, Qidxs, grid:
          [q.new_empty((1, q.shape[0]), dtype=torch.float16) for q in Qidxs])
_reg_fake("had_transform_planes_rows", lambda x, n, K, had, transpose, pre, scale, rms_weight, rms_eps, gate,
          resid_scale=0.0: x.new_empty((x.shape[0], _planes_numel(n, resid_scale)), dtype=torch.uint8))
_reg_fake("e8p_gemv_planes_rows", lambda planes, Qidxs, grid:
          Qidxs.new_empty((planes.shape[0], Qidxs.shape[0]), dtype=torch.float16))
_reg_fake("gemv_planes_rows_mode", lambda planes, Qidxs, grid, grid2, mode:
          Qidxs.new_empty((planes.shape[0], Qidxs.shape[0]), dtype=torch.float16))
_reg_fake("argmax_step", lambda logits, tok, pos: None)
_reg_fake("e8p_quantize", lambda X, grid: (torch.empty_like(X), X.new_empty((X.shape[0],), dtype=torch.int64)))
_reg_fake("e8p_mm_planes_rows", lambda planes, Qidxs, grid:
          Qidxs.new_empty((len(planes), Qidxs.shape[0]), dtype=torch.float16))
_reg_fake("e8p_gemv_planes_group", lambda planes, Qidxs, grid:
          [p.new_empty((1, q.shape[0]), dtype=torch.float16) for p, q in zip(planes, Qidxs)])
_reg_fake("e8p_gemv_fused", lambda x, z, post, residual, rms_weight, rms_eps, z_scale, pre, scale, Qidxs, grid:
          ([z.new_empty((1, z.numel()))] if z is not None else []) +
          [q.new_empty((1, q.shape[0]), dtype=torch.float16) for q in Qidxs])
_reg_fake("ffn_engine", lambda planes_gate, planes_up, q_gate, q_up, q_down, had3, sv_gate, sv_up, su_down, grid, workspace,
          out_scale, in_scale, K, dbg=None: q_down.new_empty((1, q_down.shape[0]), dtype=torch.float16))
_reg_fake("block_engine", lambda layers, h_in, pos, cos, sin, grid, workspace, n_layers, max_len, rms_eps, attn_scale,
          dbg=None, dbg_layer=-1, codebook=0, resid_scale=0.0, shape=0, grid2=None, kcache=None, vcache=None: torch.empty_like(h_in))
_reg_fake("rope_attn_decode", lambda q, k, v, cos, sin, pos, kcache, vcache, workspace=None, window=0: torch.empty_like(q))
_reg_fake("rope_attn_decode_z", lambda zs, posts, scales, cos, sin, pos, kcache, vcache, workspace=None, window=0:
          kcache.new_empty((zs[0].numel() // kcache.shape[2], kcache.shape[2])))
_reg_fake("e8p_mm_skinny", lambda x, Q, g: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("e8prvq4_mm_skinny", lambda x, Q, g, s: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("e8prvq3_mm_skinny", lambda x, Q, g, g2, s: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("d4_mm_skinny", lambda x, Q, g: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("hi_mm_skinny", lambda x, Q: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("e8p_mm_batched", lambda x, Q, g: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("e8prvq4_mm_batched", lambda x, Q, g, s: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("e8prvq3_mm_batched", lambda x, Q, g, g2, s: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("d4_mm_batched", lambda x, Q, g: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("hi_mm_batched", lambda x, Q: x.new_empty((x.shape[0], Q.shape[0]), dtype=torch.float16))
_reg_fake("e8p_gemv_planes", lambda planes, Q, g: Q.new_empty((1, Q.shape[0]), dtype=torch.float16))
for _n in ("e8p_mm_origorder", "e8prvq3_mm_origorder", "e8prvq4_mm_origorder", "d4_mm_origorder",
           "hi_mm_origorder"):
    _reg_fake(_n, _fake_mm)
_reg_fake("decompress_e8p_origorder", lambda Q, g: Q.new_empty((Q.shape[0], Q.shape[1] * 8), dtype=torch.float16))
_reg_fake("decompress_e8prvq3_origorder",
          lambda Q, g, g2, s: Q.new_empty((Q.shape[0], Q.shape[1] * 32 // 3), dtype=torch.float16))
_reg_fake("decompress_e8prvq4_origorder",
          lambda Q, g, s: Q.new_empty((Q.shape[0], Q.shape[1] * 8), dtype=torch.float16))
_reg_fake("decompress_d4_origorder", lambda Q, g: Q.new_empty((Q.shape[0], Q.shape[1] * 4), dtype=torch.float16))
_reg_fake("decompress_hi_origorder", lambda Q: Q.new_empty((Q.shape[0], Q.shape[1] * 8), dtype=torch.float16))
