// Device-side primitives shared by the gfx950 kernels of the QuIP# hot path.
// CDNA4 only (wave64, v_dot2_f32_f16, DPP row_bcast, v_perm_b32); no
// portability layer on purpose.
//
// Format statements follow SURVEY.md appendix A, which restates
// /root/reference: codebook/e8p12.py:63-103, quip_cuda/origin_order.cu:211-385.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace quip {

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// streamed-once 16-byte global load (nt policy: weights are read by one CU once)
__device__ __forceinline__ uint4 ld_nt_u4(const uint4* p) {
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// ---- hand-counted weight stream ------------------------------------------------------
// hipcc's s_waitcnt insertion drains vmcnt(0) at loop headers / exec-mask joins, which
// empties a software-pipelined load queue once per iteration.  The streaming loops
// therefore issue their loads through inline asm (invisible to that pass) and wait
// with explicit counts: VMEM loads return in issue order, so "the oldest slot has
// landed" == "at most N newer loads are still outstanding".  The wait statement names
// the destination registers as read-write so no use can be scheduled above it
// (cdna_hip_programming.md section 5.7, form ii).
__device__ __forceinline__ void asm_load16_nt(u32x4& dst, const uint4* p) {
  asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void asm_load16(u32x4& dst, const uint4* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait_vmcnt_x(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e,
                                                 u32x4& f) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait_vmcnt(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
// 12-byte slots (E8P12RVQ3B's native 3-byte codes: four codes per lane and load)
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void asm_load16_nt(u32x3& dst, const uint4* p) {
  asm volatile("global_load_dwordx3 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void asm_wait_vmcnt(u32x3& a, u32x3& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

__device__ __forceinline__ f16x2 as_f16x2(uint32_t u) { return __builtin_bit_cast(f16x2, u); }
__device__ __forceinline__ uint32_t as_u32(f16x2 h) { return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }

// Block exponent of an activation row that holds an inf or a NaN: its digit planes mean nothing and the GEMV
// epilogues answer NaN for the whole row (what the Hadamard transform of such a row is in floating point).
constexpr int kShiftNotFinite = 1 << 20;
// 2^-(sh + extra): the factor that takes the integer sums of a row at block exponent sh back to floats
__device__ __forceinline__ float unscale_of(int sh, int extra) {
  return sh == kShiftNotFinite ? as_f32(0x7fc00000u) : as_f32((uint32_t)(127 - sh - extra) << 23);
}

// acc + a.lo*b.lo + a.hi*b.hi  (v_dot2_f32_f16: fp16 products exact in fp32)
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float acc) {
  return __builtin_amdgcn_fdot2(as_f16x2(a), as_f16x2(b), acc, false);
}

__device__ __forceinline__ uint32_t pack_f16(float lo, float hi) {
  f16x2 h = {(f16)lo, (f16)hi};
  return as_u32(h);
}

// ---- wave64 sum; total lands in lane 63 (DPP only, no LDS traffic) ----------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v = dpp_add<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xf>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xf>(v);  // row_mirror : every lane of a 16-row holds the row total
  v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
  return v;
}
// butterfly variant: every lane ends with the total (uses ds_swizzle/bpermute for >16)
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- E8P12 ------------------------------------------------------------------
// position p of an 8-group <- byte kByteOfPos[p] of grid_packed_abs[abs]
// sign of position p       <- bit (7 - kByteOfPos[p]) of (sign ^ parity)
__device__ __forceinline__ constexpr int e8p_byte_of_pos(int p) {
  return (p == 1) ? 2 : (p == 2) ? 1 : (p == 5) ? 6 : (p == 6) ? 5 : p;
}

// fp16 magnitude row of one abs entry in natural position order, column-7 sign
// of the table kept (16 B).  `packed` = grid_packed_abs[abs] (int8 = 4*a).
__device__ __forceinline__ uint4 e8p_abs_row_f16(uint64_t packed) {
  uint32_t r[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    int b0 = (int)(int8_t)(packed >> (8 * e8p_byte_of_pos(2 * d)));
    int b1 = (int)(int8_t)(packed >> (8 * e8p_byte_of_pos(2 * d + 1)));
    r[d] = pack_f16(0.25f * (float)b0, 0.25f * (float)b1);
  }
  return make_uint4(r[0], r[1], r[2], r[3]);
}

// XOR masks (fp16 sign bits) of a sign byte, natural position order (16 B).
__device__ __forceinline__ uint4 e8p_sign_masks(uint32_t s) {
  uint32_t par = __builtin_popcount(s & 0xff) & 1;
  uint32_t sv = (s ^ par) & 0xff;
  uint32_t r[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    uint32_t lo = (sv >> (7 - e8p_byte_of_pos(2 * d))) & 1;
    uint32_t hi = (sv >> (7 - e8p_byte_of_pos(2 * d + 1))) & 1;
    r[d] = (lo << 15) | (hi << 31);
  }
  return make_uint4(r[0], r[1], r[2], r[3]);
}

// Full decode of one code to 8 fp16 (natural order), exact.  `absrow` is the
// table row for (code >> 8).  Used by the non-hot kernels (decompress, generic
// mm); the decode GEMV uses the hoisted form (shift applied to sum(x)).
__device__ __forceinline__ uint4 e8p_decode_f16(uint32_t code, uint4 absrow) {
  uint4 m = e8p_sign_masks(code);
  uint32_t par = __builtin_popcount(code & 0xff) & 1;
  f16x2 sh = par ? f16x2{(f16)-0.25f, (f16)-0.25f} : f16x2{(f16)0.25f, (f16)0.25f};
  uint4 w;
  w.x = as_u32(as_f16x2(absrow.x ^ m.x) + sh);
  w.y = as_u32(as_f16x2(absrow.y ^ m.y) + sh);
  w.z = as_u32(as_f16x2(absrow.z ^ m.z) + sh);
  w.w = as_u32(as_f16x2(absrow.w ^ m.w) + sh);
  return w;
}

// ---- residual table of E8P12RVQ3B (origin_order.cu:908-922) --------------------
// packed nibble i = int4(2*value) of column [0,2,4,6,1,3,5,7][i]; natural
// position pairs (0,1)=(nib0,nib4) (2,3)=(nib1,nib5) (4,5)=(nib2,nib6) (6,7)=(nib3,nib7)
__device__ __forceinline__ uint4 e81b_decode_f16(uint32_t cb2) {
  uint32_t r[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    int lo = (int)((cb2 >> (4 * d)) & 0xf), hi = (int)((cb2 >> (4 * d + 16)) & 0xf);
    lo = (lo ^ 8) - 8;  // sign-extend int4
    hi = (hi ^ 8) - 8;
    r[d] = pack_f16(0.5f * (float)lo, 0.5f * (float)hi);
  }
  return make_uint4(r[0], r[1], r[2], r[3]);
}

// w = fp16(scale)*resid + main with ONE fp16 rounding (v_pk_fma_f16), as the
// reference's __hfma2 (origin_order.cu:330-331, 378-380)
__device__ __forceinline__ uint4 rvq_combine(uint4 main, uint4 resid, f16x2 scale2) {
  uint4 w;
  w.x = as_u32(__builtin_elementwise_fma(scale2, as_f16x2(resid.x), as_f16x2(main.x)));
  w.y = as_u32(__builtin_elementwise_fma(scale2, as_f16x2(resid.y), as_f16x2(main.y)));
  w.z = as_u32(__builtin_elementwise_fma(scale2, as_f16x2(resid.z), as_f16x2(main.z)));
  w.w = as_u32(__builtin_elementwise_fma(scale2, as_f16x2(resid.w), as_f16x2(main.w)));
  return w;
}

// ---- HI (4-bit half-integer, origin_order.cu:1028-1051): w = nibble - 7.5 -------
__device__ __forceinline__ uint4 hi_decode_f16(uint32_t code) {
  uint32_t r[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    int lo = (int)((code >> (4 * d)) & 0xf), hi = (int)((code >> (4 * d + 16)) & 0xf);
    r[d] = pack_f16((float)lo - 7.5f, (float)hi - 7.5f);
  }
  return make_uint4(r[0], r[1], r[2], r[3]);
}

}  // namespace quip
