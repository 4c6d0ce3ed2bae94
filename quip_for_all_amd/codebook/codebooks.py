"""Codebook modules: tables as (non-persistent) buffers + the M-threshold
dispatch of the reference's forward() (e8p12.py:139-156, e8p12_rvq3.py:109-129,
e8p12_rvq4.py:50-67, d4.py:128-139, hi.py:52-63), and the quantise-time nearest-codeword search
`quantize` (e8p12.py:125-137, e8p12_rvq3.py:81-92, e8p12_rvq4.py:32-45, d4.py:116-123, hi.py:30-39):
the 65 536-entry E8P12 search is a structured HIP kernel (csrc/quantize.hip), the 256 / 16-entry tables
(E81B residual, D4, HI) are a small dense arg max like the reference's."""
import os
from fractions import Fraction

import torch
from torch import nn

from . import tables


class _Codebook(nn.Module):
    mm_threshold = 32  # fused mm kernel for M < threshold, else decompress + dense GEMM

    def quantize(self, X, return_idx=True):
        raise NotImplementedError

    @staticmethod
    def _round_dense(X, grid, grid_norm=None):
        """arg max_c 2 X . g_c - |g_c|^2 over a small table (the reference's round())"""
        grid = grid.to(X.dtype)
        gn = (grid * grid).sum(-1) if grid_norm is None else grid_norm.to(X.dtype)
        idx = (2 * X @ grid.T - gn).argmax(-1)
        return grid[idx], idx

    def _round_e8p(self, X):
        """nearest E8P12 codeword of every row of X (N, 8): (vals in X.dtype, idx int64)"""
        assert X.shape[-1] == 8
        vals, idx = torch.ops.quip_lib.e8p_quantize(X.detach().to(torch.float32).contiguous(), self.grid_packed_abs)
        return vals.to(X.dtype), idx

    def maybe_pack_idxs(self, idxs):
        return idxs

    def forward(self, input, Qidxs):
        if input.size(0) < self.mm_threshold:
            return self.mm(input, Qidxs)
        W = self.decompress_weight(Qidxs)
        return input @ W.T

    def batched_regime(self, m, n, k):
        """the path forward() takes, as a name (tests/test_dispatch_table.py): mm | skinny_chunks | fused_gemm | decompress_gemm"""
        return "mm" if m < self.mm_threshold else "decompress_gemm"

    def forward_reference(self, input, Qidxs):
        """the reference's own op sequence, whatever faster path exists: `*_mm_origorder` below the threshold,
        decompress + dense GEMM from it on (codebook/e8p12.py:139-156 and its siblings)"""
        return _Codebook.forward(self, input, Qidxs)


class _SkinnyMixin:
    """2 <= M < 32 beyond one exact rows-mode pass, and M >= mm_threshold up to a few hundred rows: the single-pass fp16
    skinny kernel in the codebook's mode (csrc/e8p_skinny_gemm.hip) -- the reference's arithmetic for this regime
    (origin_order.cu:388-555 with the codebook's BLayout: exact fp16 weights, fp32 accumulation), i.e. x . W of the dense W
    that decompress_weight() writes.  Subclasses give mm_skinny() and the m * n up to which chunks of 32 rows beat
    decompress + dense GEMM.  Beyond that, QUIP_BATCHED_MM=fused selects the fused dequant + MFMA tile kernel in the
    codebook's mode (csrc/e8p_prefill_gemm.hip, mm_batched()): the same product of the same dense W without the
    2 n k bytes of scratch and without the vendor GEMM; the default stays decompress + dense GEMM, the reference's shape
    (e8p12_rvq4.py:50-67, e8p12_rvq3.py:109-129, d4.py:128-139, hi.py:52-63), measured faster."""
    skinny_chunks_max_mn = int(os.environ.get("QUIP_SKINNY_MAX_MN", str(1_200_000)))

    @staticmethod
    def skinny_supported(m, q_out, q_in):
        return 1 <= m and q_out >= 2 and q_out % 2 == 0 and q_in >= 128 and q_in % 128 == 0

    def batched_regime(self, m, n, k):
        if m < self.mm_threshold:
            return "mm"
        if (E8P12_codebook.batched_mode != "reference" and m * n <= self.skinny_chunks_max_mn
                and self.skinny_supported(m, n, k)):
            return "skinny_chunks"
        if E8P12_codebook.batched_mode == "fused" and n % 2 == 0 and k % 64 == 0:
            return "fused_gemm"
        return "decompress_gemm"

    def forward(self, input, Qidxs):
        if input.size(0) >= self.mm_threshold and input.is_cuda and input.dtype == torch.float16 and input.dim() == 2:
            regime = self.batched_regime(input.shape[0], Qidxs.shape[0], input.shape[1])
            if regime == "skinny_chunks":
                return self.mm_skinny(input, Qidxs)
            if regime == "fused_gemm":
                return self.mm_batched(input, Qidxs)
        return _Codebook.forward(self, input, Qidxs)


class E8P12_codebook(_Codebook):
    def __init__(self, inference=False, **kwargs):
        super().__init__()
        self.id = "E8P12"
        self.opt_scale = 1.03
        self.codesz = 8
        self.idx_dtype = torch.int16
        self.packsz = 1
        self.pack_out = False
        self.version = 1
        self.register_buffer("grid_packed_abs", torch.from_numpy(tables.e8p_grid_packed_abs().copy()),
                             persistent=False)
        if not inference:
            g = torch.from_numpy(tables.e8p_full_grid().copy())
            self.register_buffer("grid", g, persistent=False)
            self.register_buffer("grid_norm", g.norm(dim=-1) ** 2, persistent=False)

    def round(self, X, grid=None, grid_norm=None):
        return self._round_e8p(X)

    def quantize(self, X, return_idx=True):
        vals, idx = self._round_e8p(X)
        return (vals, idx) if return_idx else vals

    def decompress_weight(self, Qidxs):
        return torch.ops.quip_lib.decompress_e8p_origorder(Qidxs, self.grid_packed_abs)

    def mm(self, input, Qidxs):
        return torch.ops.quip_lib.e8p_mm_origorder(input, Qidxs, self.grid_packed_abs)

    # M >= mm_threshold: three paths with the same arithmetic (exact fp16 weights, fp32 accumulation, one fp16 rounding):
    #   skinny_chunks    the single-pass skinny kernel on chunks of 32 rows, up to a few hundred rows;
    #   decompress_gemm  the reference's shape (e8p12.py:152-155): dense fp16 W in scratch memory, then the vendor GEMM.
    #                    Measured on MI355X (tools/prefill_crossover.py, profiles/r03_prefill_crossover.txt) it beats the
    #                    fused kernel at every M from 256 to 32768 on all three 7B shapes (1.0-2.0 x; 1.34-1.46 x at
    #                    M = 32768: 1.2-1.6 PFLOP/s), so it is the default;
    #   fused_gemm       fused dequant + MFMA GEMM (csrc/e8p_prefill_gemm.hip): no 2 n k bytes of scratch, no vendor
    #                    library in the path, 0.94-1.17 PFLOP/s.  QUIP_BATCHED_MM=fused (or 1) selects it.
    # QUIP_BATCHED_MM=0: the reference-shaped path for every M >= mm_threshold (no skinny chunks either).
    batched_mode = {"1": "fused", "0": "reference"}.get(os.environ.get("QUIP_BATCHED_MM", "auto"),
                                                        os.environ.get("QUIP_BATCHED_MM", "auto"))

    def batched_regime(self, m, n, k):
        """the path forward() takes for an (m, k) fp16 batch against (n, k / 8) codes, as a name: mm (below the
        threshold) | skinny_chunks | fused_gemm | decompress_gemm"""
        if m < self.mm_threshold:
            return "mm"
        if self.batched_mode != "reference" and n % 2 == 0 and k % 64 == 0:
            if m * n <= self.skinny_chunks_max_mn and self.skinny_supported(m, n, k):
                return "skinny_chunks"
            if self.batched_mode == "fused":
                return "fused_gemm"
        return "decompress_gemm"

    def forward(self, input, Qidxs):
        if input.size(0) < self.mm_threshold:
            return self.mm(input, Qidxs)
        if input.is_cuda and input.dtype == torch.float16 and input.dim() == 2:
            # up to a few hundred rows the single-pass skinny kernel on chunks of 32 rows (many small workgroups, a
            # few microseconds each) beats both GEMMs (tools/midm_bench.py: 4096 x 4096 at M = 64: 10 us against 37
            # for decompress + GEMM and 55 for the tile kernel; M = 256: 34 / 40 / 56)
            regime = self.batched_regime(input.shape[0], Qidxs.shape[0], input.shape[1])
            if regime == "skinny_chunks":
                return torch.ops.quip_lib.e8p_mm_skinny(input, Qidxs, self.grid_packed_abs)
            if regime == "fused_gemm":
                return torch.ops.quip_lib.e8p_mm_batched(input, Qidxs, self.grid_packed_abs)
        W = self.decompress_weight(Qidxs)
        return input @ W.T

    @staticmethod
    def planes_supported(q_out, q_in):
        """shapes the matrix-core bs=1 GEMV takes (csrc/e8p_gemv_mfma.hip)"""
        return q_in % 128 == 0 and 128 <= q_in <= 28672 and q_out >= 1

    @staticmethod
    def planes_group_supported(q_outs, q_in):
        """1..3 GEMVs of a common k in one launch: all digit planes must fit in LDS next to the tables"""
        kp = (q_in + 511) // 512 * 512
        return 1 <= len(q_outs) <= 3 and len(q_outs) * kp <= 31232

    @staticmethod
    def fused_supported(count, q_in):
        """GEMV with the input transform in its prologue (csrc/e8p_gemv_mfma.hip, FusedIn): k a power
        of two in 1024..8192, tables + count x planes + the transform buffer within 160 KB of LDS"""
        pow2 = q_in & (q_in - 1) == 0
        lds = 2 * 256 * 16 * 8 + 256 * 16 + 3 * q_in * count + (q_in + q_in // 32 + 4 + 16) * 4
        return 1 <= count <= 3 and pow2 and 1024 <= q_in <= 8192 and lds <= 160 * 1024

    def mm_planes(self, planes, Qidxs):
        """bs=1 product with x given as int8 digit planes (quip_lib::had_transform_planes)"""
        return torch.ops.quip_lib.e8p_gemv_planes(planes, Qidxs, self.grid_packed_abs)

    def mm_planes_group(self, planes, Qidxs):
        return list(torch.ops.quip_lib.e8p_gemv_planes_group(planes, Qidxs, self.grid_packed_abs))

    def mm_planes_rows(self, planes, Qidxs):
        """skinny product: planes (M, planes_bytes) -> (M, n), passes of up to 5 rows over the codes"""
        return torch.ops.quip_lib.e8p_gemv_planes_rows(planes, Qidxs, self.grid_packed_abs)

    skinny_chunks_max_mn = int(os.environ.get("QUIP_SKINNY_MAX_MN", str(1_200_000)))

    @staticmethod
    def skinny_supported(m, q_out, q_in):
        """shapes the single-pass fp16 skinny product takes (csrc/e8p_skinny_gemm.hip; rows beyond 32: chunks of 32
        in one launch)"""
        return m >= 1 and q_out % 2 == 0 and q_in % 128 == 0 and q_in >= 128

    def mm_skinny(self, xh, Qidxs):
        """(M <= 32, k) fp16 (already input-transformed) -> (M, n): one pass over the codes, fp16 MFMA"""
        return torch.ops.quip_lib.e8p_mm_skinny(xh, Qidxs, self.grid_packed_abs)


class E8P12RVQ4B_codebook(_SkinnyMixin, _Codebook):
    def __init__(self, inference=False, opt_resid_scale=None, **kwargs):
        super().__init__()
        self.id = "E8P12RVQ4B"
        self.opt_scale = 1.03
        self.codesz = 8
        self.idx_dtype = torch.int32
        self.packsz = 1
        self.pack_out = False
        self.version = 0
        # NB the reference's quantizer passes its default -1 straight through
        # (quantizer.py:69,126-127): only None selects 1/3.45 (e8p12_rvq4.py:23).
        self.opt_resid_scale = 1 / 3.45 if opt_resid_scale is None else opt_resid_scale
        self.register_buffer("grid_packed_abs", torch.from_numpy(tables.e8p_grid_packed_abs().copy()),
                             persistent=False)

    def quantize(self, X, return_idx=True):
        """two E8P12 stages: x ~ g_a + s * g_b, code = a << 16 | b (e8p12_rvq4.py:37-45)"""
        init_vals, init_idxs = self._round_e8p(X)
        resid_vals, resid_idxs = self._round_e8p((X - init_vals) / self.opt_resid_scale)
        final = init_vals + resid_vals * self.opt_resid_scale
        return (final, (init_idxs << 16) + resid_idxs) if return_idx else final

    def decompress_weight(self, Qidxs):
        return torch.ops.quip_lib.decompress_e8prvq4_origorder(Qidxs, self.grid_packed_abs,
                                                               self.opt_resid_scale)

    def mm(self, input, Qidxs):
        return torch.ops.quip_lib.e8prvq4_mm_origorder(input, Qidxs, self.grid_packed_abs,
                                                       self.opt_resid_scale)

    # bs=1 on the E8P12 matrix-core GEMV: read as 16-bit codes an RVQ4 row is an E8P12 row of 2k
    # weights whose 8-groups alternate residual / main codes, and W x = W' x' with
    # x' = [s * x_g | x_g]_g (s = the fp16 residual scale the reference uses, origin_order.cu:337-385).
    # The Hadamard launch writes the digit planes of x' (resid_scale argument); main + s * resid is
    # summed exactly instead of being rounded to fp16 per weight.
    @property
    def planes_resid_scale(self):
        return float(torch.tensor(self.opt_resid_scale, dtype=torch.float16))

    @staticmethod
    def planes_supported(q_out, q_in):
        # virtual rows longer than 28672 (70B down_proj: 2 k = 57344) go to the K-splitting kernel
        # (csrc/e8p_gemv_v2.hip) through the op's dispatcher
        return (2 * q_in) % 128 == 0 and 128 <= 2 * q_in <= 57344 and q_out >= 1

    @staticmethod
    def planes_group_supported(q_outs, q_in):
        kp = (2 * q_in + 511) // 512 * 512
        return 1 <= len(q_outs) <= 3 and len(q_outs) * kp <= 31232

    def mm_planes(self, planes, Qidxs):
        return torch.ops.quip_lib.e8p_gemv_planes(planes, Qidxs.view(torch.int16), self.grid_packed_abs)

    def mm_planes_group(self, planes, Qidxs):
        return list(torch.ops.quip_lib.e8p_gemv_planes_group(planes, [q.view(torch.int16) for q in Qidxs],
                                                             self.grid_packed_abs))

    def mm_planes_rows(self, planes, Qidxs):
        return torch.ops.quip_lib.e8p_gemv_planes_rows(planes, Qidxs.view(torch.int16), self.grid_packed_abs)

    skinny_chunks_max_mn = _SkinnyMixin.skinny_chunks_max_mn // 2     # (twice the code bytes per weight)

    def mm_skinny(self, xh, Qidxs):
        return torch.ops.quip_lib.e8prvq4_mm_skinny(xh, Qidxs, self.grid_packed_abs, self.opt_resid_scale)

    def mm_batched(self, xh, Qidxs):
        return torch.ops.quip_lib.e8prvq4_mm_batched(xh, Qidxs, self.grid_packed_abs, self.opt_resid_scale)


class E8P12RVQ3B_codebook(_SkinnyMixin, _Codebook):
    def __init__(self, inference=False, opt_resid_scale=None, **kwargs):
        super().__init__()
        self.id = "E8P12RVQ3B"
        self.opt_scale = 0.98
        self.codesz = 8
        self.idx_dtype = torch.int32
        self.packsz = Fraction(4, 3)
        self.pack_out = False
        self.version = 0
        self.opt_resid_scale = 1 / 2.04 if opt_resid_scale is None else opt_resid_scale
        self.register_buffer("grid_packed_abs", torch.from_numpy(tables.e8p_grid_packed_abs().copy()),
                             persistent=False)
        self.register_buffer("e81b_grid", torch.from_numpy(tables.e81b_grid().copy()), persistent=False)
        self.register_buffer("e81b_grid_packed", torch.from_numpy(tables.e81b_grid_packed().copy()),
                             persistent=False)

    def quantize(self, X, return_idx=True):
        """E8P12 stage + 256-entry E81B residual: code = a << 8 | b (e8p12_rvq3.py:81-92)"""
        init_vals, init_idxs = self._round_e8p(X)
        resid_vals, resid_idxs = self._round_dense((X - init_vals) / self.opt_resid_scale, self.e81b_grid)
        final = init_vals + resid_vals * self.opt_resid_scale
        return (final, (init_idxs << 8) + resid_idxs) if return_idx else final

    # bs=1 on the matrix-core GEMV: a 3-byte code [resid8, e8p_lo, e8p_hi] read behind a zero byte is the dword
    # (main16 << 16 | resid8 << 8), i.e. an RVQ4-style row of 2k virtual weights (8-groups alternate residual / main)
    # whose LOW codes index the E81B table (csrc/e8p_gemv_mfma.hip, table modes 40 / 20); x' = [s * x_g | x_g]_g as
    # for RVQ4.  The kernel streams the checkpoint's own bytes (12-byte loads = 4 codes per lane): no repacked copy.
    @property
    def planes_resid_scale(self):
        return float(torch.tensor(self.opt_resid_scale, dtype=torch.float16))

    @staticmethod
    def planes_supported(q_out, q_in):
        # virtual rows longer than 25600 (70B down_proj: 2 k = 57344): the K-splitting kernel's third-table mode
        return (2 * q_in) % 128 == 0 and 128 <= 2 * q_in <= 57344 and q_out >= 1

    @staticmethod
    def planes_group_supported(q_outs, q_in):
        kp = (2 * q_in + 511) // 512 * 512
        return 1 <= len(q_outs) <= 3 and len(q_outs) * kp <= 25600

    def _e81b_i8(self, device):
        t = getattr(self, "_e81b_i8_cache", None)
        if t is None or t.device != device:
            t = (self.e81b_grid.to(torch.float32) * 4).round().to(torch.int8).contiguous().to(device)
            self._e81b_i8_cache = t
        return t

    def mm_planes(self, planes, Qidxs):
        return self.mm_planes_group([planes], [Qidxs])[0]

    def mm_planes_group(self, planes, Qidxs):
        return list(torch.ops.quip_lib.e8prvq3_gemv_planes_group(
            planes, list(Qidxs), self.grid_packed_abs, self._e81b_i8(Qidxs[0].device)))

    def mm_planes_rows(self, planes, Qidxs):
        return torch.ops.quip_lib.gemv_planes_rows_mode(planes, Qidxs, self.grid_packed_abs,
                                                        self._e81b_i8(Qidxs.device), 40)

    skinny_chunks_max_mn = _SkinnyMixin.skinny_chunks_max_mn * 2 // 3     # (1.5 x the code bytes per weight)

    def mm_skinny(self, xh, Qidxs):
        return torch.ops.quip_lib.e8prvq3_mm_skinny(xh, Qidxs, self.grid_packed_abs, self.e81b_grid_packed, self.opt_resid_scale)

    def mm_batched(self, xh, Qidxs):
        return torch.ops.quip_lib.e8prvq3_mm_batched(xh, Qidxs, self.grid_packed_abs, self.e81b_grid_packed, self.opt_resid_scale)

    def maybe_pack_idxs(self, idxs):
        """keep the low 3 bytes of every int32 index (e8p12_rvq3.py:102-107)"""
        b = idxs.contiguous().view(torch.int8).view(idxs.shape[0], idxs.shape[1], -1)
        return b[..., :3].reshape(idxs.shape[0], -1).view(torch.int32)

    def decompress_weight(self, Qidxs):
        return torch.ops.quip_lib.decompress_e8prvq3_origorder(
            Qidxs, self.grid_packed_abs, self.e81b_grid_packed, self.opt_resid_scale)

    def mm(self, input, Qidxs):
        return torch.ops.quip_lib.e8prvq3_mm_origorder(
            input, Qidxs, self.grid_packed_abs, self.e81b_grid_packed, self.opt_resid_scale)


class D4_codebook(_SkinnyMixin, _Codebook):
    mm_threshold = 24  # d4.py:134

    def __init__(self, inference=False, **kwargs):
        super().__init__()
        self.id = "D4"
        self.codesz = 4
        self.opt_scale = 1.21
        self.idx_dtype = torch.uint8
        self.packsz = 1
        self.pack_out = False
        self.version = 0
        # the kernels need fp16 entries (8-byte rows); keep the buffer in fp16 so that
        # per-layer copies are valid whatever the caller casts
        self.register_buffer("grid", torch.from_numpy(tables.d4_grid().copy()).half(), persistent=False)

    def quantize(self, X, return_idx=True):
        assert X.shape[-1] == self.codesz
        vals, idx = self._round_dense(X, self.grid.float())
        return (vals, idx.to(self.idx_dtype)) if return_idx else vals

    def decompress_weight(self, Qidxs):
        return torch.ops.quip_lib.decompress_d4_origorder(Qidxs, self.grid)

    def mm(self, input, Qidxs):
        return torch.ops.quip_lib.d4_mm_origorder(input, Qidxs, self.grid)

    # bs=1: the same integer-domain matrix-core GEMV as E8P12 (2w of every D4 entry is an int8)
    @staticmethod
    def planes_supported(q_out, q_in):
        return q_in % 128 == 0 and 128 <= q_in <= 28672 and q_out >= 1

    @staticmethod
    def planes_group_supported(q_outs, q_in):
        kp = (q_in + 511) // 512 * 512
        return 1 <= len(q_outs) <= 3 and len(q_outs) * kp <= 31232

    def mm_planes(self, planes, Qidxs):
        return torch.ops.quip_lib.d4_gemv_planes(planes, Qidxs, self.grid)

    def mm_planes_group(self, planes, Qidxs):
        return list(torch.ops.quip_lib.d4_gemv_planes_group(planes, Qidxs, self.grid))

    def mm_planes_rows(self, planes, Qidxs):
        return torch.ops.quip_lib.gemv_planes_rows_mode(planes, Qidxs, self.grid, None, 64)

    def mm_skinny(self, xh, Qidxs):
        return torch.ops.quip_lib.d4_mm_skinny(xh, Qidxs, self.grid)

    def mm_batched(self, xh, Qidxs):
        return torch.ops.quip_lib.d4_mm_batched(xh, Qidxs, self.grid)


class HI4B1C_codebook(_SkinnyMixin, _Codebook):
    def __init__(self, inference=False, **kwargs):
        super().__init__()
        self.id = "HI"
        self.opt_scale = 2.97
        self.codesz = 1
        self.idx_dtype = torch.int32
        self.packsz = 8
        self.pack_out = False
        self.version = 0
        if not inference:
            g = (torch.arange(-8, 8) + 0.5).unsqueeze(-1)
            self.register_buffer("grid", g, persistent=False)
            self.register_buffer("grid_norm", (g @ g.T).diag(), persistent=False)

    # bs=1 on the D4 mode of the matrix-core GEMV: a code byte (two nibbles) is a "D4 code" of the
    # virtual (n_out, 2k) matrix with entry [lo - 7.5, hi - 7.5, 0, 0]; the Hadamard launch writes the
    # planes of the matching virtual vector (csrc/hadamard.hip, HI layout).  Exact: 2w is an int8.
    planes_resid_scale = float("inf")      # register_lib.HI_PLANES: selects the HI planes layout

    def _virtual_grid(self, device):
        g = getattr(self, "_vgrid", None)
        if g is None or g.device != device:
            b = torch.arange(256)
            g = torch.stack([(b & 15).float() - 7.5, (b >> 4).float() - 7.5, torch.zeros(256), torch.zeros(256)], 1)
            g = g.to(torch.float16).contiguous().to(device)
            self._vgrid = g
        return g

    @staticmethod
    def planes_supported(q_out, q_in):
        # virtual rows longer than 28672 (70B down_proj: 2 k = 57344) go to the K-splitting kernel's D4 table mode
        # (csrc/e8p_gemv_v2.hip) through the op's dispatcher
        return (2 * q_in) % 128 == 0 and 128 <= 2 * q_in <= 57344 and q_out >= 1

    @staticmethod
    def planes_group_supported(q_outs, q_in):
        kp = (2 * q_in + 511) // 512 * 512
        return 1 <= len(q_outs) <= 3 and len(q_outs) * kp <= 31232

    def mm_planes(self, planes, Qidxs):
        return torch.ops.quip_lib.d4_gemv_planes(planes, Qidxs.view(torch.uint8), self._virtual_grid(Qidxs.device))

    def mm_planes_group(self, planes, Qidxs):
        return list(torch.ops.quip_lib.d4_gemv_planes_group(planes, [q.view(torch.uint8) for q in Qidxs],
                                                            self._virtual_grid(Qidxs[0].device)))

    def quantize(self, X, return_idx=True):
        assert X.shape[-1] == self.codesz
        g = (torch.arange(-8, 8, device=X.device, dtype=X.dtype) + 0.5).unsqueeze(-1)
        vals, idx = self._round_dense(X, g)
        return (vals, idx.to(self.idx_dtype)) if return_idx else vals

    def mm_planes_rows(self, planes, Qidxs):
        return torch.ops.quip_lib.gemv_planes_rows_mode(planes, Qidxs.view(torch.uint8), self._virtual_grid(Qidxs.device),
                                                        None, 64)

    def maybe_pack_idxs(self, idxs):
        """nibble i <- column [0,2,4,6,1,3,5,7][i] of each 8-group (hi.py:41-50)"""
        out = torch.zeros(idxs.shape[0], idxs.shape[1] // 8, dtype=idxs.dtype, device=idxs.device)
        for i, col in enumerate(tables.HI_NIBBLE_COLS):
            out = out + (idxs[:, col::8] << (4 * i))
        return out

    skinny_chunks_max_mn = _SkinnyMixin.skinny_chunks_max_mn // 2     # (twice the code bytes per weight)

    def mm_skinny(self, xh, Qidxs):
        return torch.ops.quip_lib.hi_mm_skinny(xh, Qidxs)

    def mm_batched(self, xh, Qidxs):
        return torch.ops.quip_lib.hi_mm_batched(xh, Qidxs)

    def decompress_weight(self, Qidxs):
        return torch.ops.quip_lib.decompress_hi_origorder(Qidxs)

    def mm(self, input, Qidxs):
        return torch.ops.quip_lib.hi_mm_origorder(input, Qidxs)
