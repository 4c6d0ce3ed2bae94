// Device-side pieces of the randomised-Hadamard kernels shared by hadamard.hip (stand-alone
// launches) and e8p_gemv_mfma.hip (the same transforms executed in the GEMV prologue).  Everything
// numerically relevant lives here, written with explicitly rounded operations (no fp contraction),
// so that a transform gives bit-identical results wherever it runs.
#pragma once
#include "quip_device.hip.h"

namespace quip {
namespace had {

// Individually rounded operations.  HIP's __fmul_rn / __fadd_rn are plain operators and may still
// be contracted into an fma after inlining (one context yes, another no -> 1 ulp differences), so
// contraction is switched off lexically here.
__device__ __forceinline__ float fmul(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float fadd(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float fsub(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// g * sigmoid(g) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division
// (~10 instructions): the tall kernels evaluate it for the whole row in every workgroup
__device__ __forceinline__ float silu(float g) {
#pragma clang fp contract(off)
  return g * __builtin_amdgcn_rcpf(1.f + __expf(-g));
}

// Workgroup barrier between LDS phases.  RAW = false: __syncthreads() (a workgroup fence: s_waitcnt vmcnt(0) lgkmcnt(0) +
// s_barrier).  RAW = true: only the LDS counter is drained before the s_barrier -- for kernels that keep vector-memory
// loads in flight across their LDS phases on purpose (the persistent decode engine's weight requests) and exchange
// nothing through global memory inside the workgroup at that point.
template <bool RAW>
__device__ __forceinline__ void wg_barrier() {
  if constexpr (RAW) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else __syncthreads();
}

// LDS index with one pad word per 32 (keeps the strided pass reads off a single bank)
__device__ __forceinline__ int pad(int i) { return i + (i >> 5); }
__host__ __device__ constexpr int buf_floats(int elems) { return elems + (elems >> 5) + 4; }

// element index held in register r of thread t during pass p (4 new index bits per pass; the
// last pass may have nb < 4 new bits, the spare register bits then reuse index bits [0, 4 - nb))
__device__ __forceinline__ int pass_index(int t, int r, int p, int nb) {
  const int sh_lo = 4 - nb;                       // register high bits -> index bits [0, sh_lo)
  const int lo_bits = 4 * p - sh_lo;              // thread low bits -> index bits [sh_lo, 4p)
  const int t_lo = t & ((1 << lo_bits) - 1), t_hi = t >> lo_bits;
  return (t_hi << (4 * p + nb)) | ((r & ((1 << nb) - 1)) << (4 * p)) | (t_lo << sh_lo) | (r >> nb);
}

// wave64 reduction on DPP only; the result lands in lane 63.  The max variant is for
// non-negative values (rows masked out of a row_bcast step contribute 0).
template <bool IS_MAX, int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_step(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false);
  const float w = __builtin_bit_cast(float, t);
  return IS_MAX ? fmaxf(v, w) : fadd(v, w);
}
template <bool IS_MAX>
__device__ __forceinline__ float wave_reduce_to_lane63(float v) {
  v = dpp_step<IS_MAX, 0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v = dpp_step<IS_MAX, 0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v = dpp_step<IS_MAX, 0x141, 0xf>(v);  // row_half_mirror
  v = dpp_step<IS_MAX, 0x140, 0xf>(v);  // row_mirror: every lane of a 16-row holds the row result
  v = dpp_step<IS_MAX, 0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
  v = dpp_step<IS_MAX, 0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

// Workgroup reduction over the first nt threads (nt a multiple of 64 or < 64); EVERY thread of the
// workgroup must call it (barriers), threads >= nt contribute nothing.  Fixed order: DPP tree
// inside a wave, then waves 0, 1, 2, ... -> the same value wherever the same data is reduced.
// is_max: values must be >= 0.
// lead_barrier = false: the caller guarantees that nobody can still be reading `red` (a slot
// used once per kernel), which saves one barrier.
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red, int tid, int nt,
                                              bool lead_barrier = true) {
  v = is_max ? wave_reduce_to_lane63<true>(v) : wave_reduce_to_lane63<false>(v);
  if (lead_barrier) __syncthreads();
  if ((tid & 63) == 63 && tid < nt) red[tid >> 6] = v;
  if (nt < 64 && tid == nt - 1) red[0] = v;   // (not used by the blocked kernels: nt >= 64 there)
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < ((nt + 63) >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : fadd(r, red[w]);
  return r;
}

// A sum (over the first nt threads) and a maximum of non-negative values (over the first nta <= nt threads) in
// ONE pass: same trees, hence the same two values, as block_reduce(sum, false, red, ..) and
// block_reduce(mx, true, red + 16, ..) -- one barrier and one LDS round trip instead of two.  red: 32 floats
// nobody else is using (no leading barrier).
__device__ __forceinline__ void block_reduce_sum_max(float& sum, float& mx, float* red, int tid, int nt, int nta) {
  sum = wave_reduce_to_lane63<false>(sum);
  mx = wave_reduce_to_lane63<true>(mx);
  if ((tid & 63) == 63 && tid < nt) red[tid >> 6] = sum;
  if ((tid & 63) == 63 && tid < nta) red[16 + (tid >> 6)] = mx;
  if (nt < 64 && tid == nt - 1) red[0] = sum;
  if (nta < 64 && tid == nta - 1) red[16] = mx;
  __syncthreads();
  float r = red[0], q = red[16];
  for (int w = 1; w < ((nt + 63) >> 6); ++w) r = fadd(r, red[w]);
  for (int w = 1; w < ((nta + 63) >> 6); ++w) q = fmaxf(q, red[16 + w]);
  sum = r;
  mx = q;
}

// 8 fp16 of a 16-byte piece -> fp32
__device__ __forceinline__ void unpack8(const uint4& u, float o[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 h = as_f16x2(w[i]);
    o[2 * i] = (float)h.x;
    o[2 * i + 1] = (float)h.y;
  }
}
__device__ __forceinline__ void sumsq8(const float o[8], float& ss) {
#pragma unroll
  for (int i = 0; i < 8; ++i) ss = __fmaf_rn(o[i], o[i], ss);
}
// Element-wise products and sums two at a time (v_pk_mul_f32 / v_pk_add_f32 on register pairs): the IEEE operations of
// fmul / fadd, half the issue slots.  Q: uint4 or a 4 x uint32 vector.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <typename Q>
__device__ __forceinline__ void unpack8p(const Q& u, f32x2 o[4]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 h = as_f16x2(w[i]);
    o[i] = f32x2{(float)h.x, (float)h.y};
  }
}
template <typename Q>
__device__ __forceinline__ void mul8p(f32x2 o[4], const Q& u) {
#pragma clang fp contract(off)
  f32x2 t[4];
  unpack8p(u, t);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = o[i] * t[i];
}
template <typename Q>
__device__ __forceinline__ void add8p(f32x2 o[4], const Q& u) {
#pragma clang fp contract(off)
  f32x2 t[4];
  unpack8p(u, t);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = o[i] + t[i];
}
// 4 pairs -> 8 fp16 (round to nearest even, v_cvt_pk_f16_f32)
__device__ __forceinline__ uint4 pack8p(const f32x2 o[4]) {
  uint4 pk;
  pk.x = as_u32(__builtin_convertvector(o[0], f16x2));
  pk.y = as_u32(__builtin_convertvector(o[1], f16x2));
  pk.z = as_u32(__builtin_convertvector(o[2], f16x2));
  pk.w = as_u32(__builtin_convertvector(o[3], f16x2));
  return pk;
}
__device__ __forceinline__ void mul8(float o[8], const uint4& piece) {
#pragma clang fp contract(off)
  f32x2 t[4];
  unpack8p(piece, t);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 r = f32x2{o[2 * i], o[2 * i + 1]} * t[i];
    o[2 * i] = r.x;
    o[2 * i + 1] = r.y;
  }
}
__device__ __forceinline__ void silu_mul8(float o[8], const uint4& gate_piece) {
#pragma clang fp contract(off)
  f32x2 t[4];
  unpack8p(gate_piece, t);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2 r = f32x2{o[2 * i], o[2 * i + 1]} * f32x2{silu(t[i].x), silu(t[i].y)};
    o[2 * i] = r.x;
    o[2 * i + 1] = r.y;
  }
}

// RMSNorm factor folded into the transform scale
__device__ __forceinline__ float rms_scale(float scale, float sumsq, int n, float eps) {
#pragma clang fp contract(off)
  const float mean = sumsq / (float)n;
  return fmul(scale, rsqrtf(fadd(mean, eps)));
}

// Length-2^LOGL Walsh-Hadamard transform of E = 16 * (#active threads) values viewed as rows of
// 2^LOGL: thread t holds the 16 consecutive elements [16 t, 16 t + 16).  4 butterfly stages per
// pass in registers, re-shuffle through LDS (buf: buf_floats(E) floats) between passes.  Every
// thread of the workgroup must call it; only `active` threads (t < E / 16) touch data.
// LOGL is a compile-time constant: the element index of register r in pass P is
// (thread bits) | (register bits) with disjoint bit fields, and pad(a | b) = pad(a) + pad(b) for
// disjoint fields, so every LDS access is `thread base + immediate offset`.
template <int LOGL, int P>
struct FhtPass {
  static constexpr int NB = (LOGL - 4 * P) < 4 ? (LOGL - 4 * P) : 4;   // new index bits of this pass
  static constexpr int SH_LO = 4 - NB;
  static constexpr int LO_BITS = 4 * P - SH_LO;
  __device__ static constexpr int rpart(int r) { return ((r & ((1 << NB) - 1)) << (4 * P)) | (r >> NB); }
  __device__ static constexpr int rpad(int r) { return rpart(r) + (rpart(r) >> 5); }
  __device__ static __forceinline__ int tbase(int t) {
    const int t_lo = t & ((1 << LO_BITS) - 1), t_hi = t >> LO_BITS;
    const int tp = (t_hi << (4 * P + NB)) | (t_lo << SH_LO);
    return tp + (tp >> 5);
  }
  // Two butterflies per instruction (v_pk_add_f32 on the register pairs (v[2i], v[2i + 1])): the same IEEE additions
  // as fadd / fsub one at a time, half the VALU issue slots -- the batch transforms are bound by those.
  __device__ static __forceinline__ void butterflies(float v[16]) {
#pragma clang fp contract(off)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    if (NB >= 1) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {     // partners inside a pair: (a + b, a - b)
        const f32x2 a = {v[r], v[r]}, b = {v[r + 1], -v[r + 1]};
        const f32x2 c = a + b;
        v[r] = c.x;
        v[r + 1] = c.y;
      }
    }
#pragma unroll
    for (int s = 1; s < NB; ++s) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        if (!(r & (1 << s))) {
          const int q = r | (1 << s);
          const f32x2 x0 = {v[r], v[r + 1]}, x1 = {v[q], v[q + 1]};
          const f32x2 p = x0 + x1, m = x0 - x1;
          v[r] = p.x;
          v[r + 1] = p.y;
          v[q] = m.x;
          v[q + 1] = m.y;
        }
      }
    }
  }
  __device__ static __forceinline__ void load(float v[16], const float* buf, int t) {
    const float* b = buf + tbase(t);
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = b[rpad(r)];
  }
  __device__ static __forceinline__ void store(const float v[16], float* buf, int t) {
    float* b = buf + tbase(t);
#pragma unroll
    for (int r = 0; r < 16; ++r) b[rpad(r)] = v[r];
  }
};

// Butterflies on index bits 4..7 = lane bits 0..3 (thread t holds elements [16 t, 16 t + 16), so these partners sit
// in the same DPP row of 16 lanes): the value lane ^ (1 << S) holds comes over by a DPP move (bits 0, 1: quad
// permutes; bit 3: row rotate by 8) or ds_swizzle (bit 2; the LDS crossbar, no memory), and
//   bit clear: x0 + x1 = partner + own,   bit set: x0 - x1 = partner - own
// is one packed fma with (+-1, +-1) (exact product): the same IEEE additions as a pass through LDS, without its
// 32 LDS accesses per thread and two workgroup barriers.
template <int S>
__device__ __forceinline__ float lane_xor(float v) {
  const int i = __builtin_bit_cast(int, v);
  if constexpr (S == 0) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0xB1, 0xf, 0xf, true));        // quad_perm [1,0,3,2]
  else if constexpr (S == 1) return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  else if constexpr (S == 2) return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(i, 0x101F));              // bit mode: xor 4
  else return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(i, 0x128, 0xf, 0xf, true));                        // row_ror:8
}
// xor 4 without the LDS crossbar: row_half_mirror (lane ^ 7 within 8) then quad_perm [3, 2, 1, 0] (lane ^ 3)
__device__ __forceinline__ float lane_xor4_dpp(float v) {
  const int a = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true);
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(a, 0x1B, 0xf, 0xf, true));
}
// s + s(lane ^ 1), then ^ 2, ^ 4, ^ 8: the sum over an aligned group of 16 lanes in every lane of the group -- the additions
// of `for (o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o)`, operand for operand, on DPP moves instead of four dependent
// ds_bpermute round trips (the attention loops' critical path: ~100 clocks each)
template <int N = 16>
__device__ __forceinline__ float sum16_xor(float s) {      // N = 16 or 8 lanes per group
#pragma clang fp contract(off)
  static_assert(N == 16 || N == 8, "groups of 8 or 16 lanes");
  s = s + lane_xor<0>(s);
  s = s + lane_xor<1>(s);
  s = s + lane_xor4_dpp(s);
  if constexpr (N == 16) s = s + lane_xor<3>(s);
  return s;
}
template <int S>
__device__ __forceinline__ void lane_stage(float v[16], int t) {
#pragma clang fp contract(off)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const float sg = ((t >> S) & 1) ? -1.f : 1.f;
  const f32x2 sg2 = {sg, sg};
#pragma unroll
  for (int r = 0; r < 16; r += 2) {
    const f32x2 own = {v[r], v[r + 1]}, par = {lane_xor<S>(v[r]), lane_xor<S>(v[r + 1])};
    const f32x2 w = __builtin_elementwise_fma(own, sg2, par);
    v[r] = w.x;
    v[r + 1] = w.y;
  }
}
// index bits 0 .. min(LOGL, 8) - 1 of the transform, entirely in registers and between lanes
template <int LOGL>
__device__ __forceinline__ void fht16_lanes(float v[16], int t) {
  FhtPass<LOGL, 0>::butterflies(v);
  if constexpr (LOGL > 4) lane_stage<0>(v, t);
  if constexpr (LOGL > 5) lane_stage<1>(v, t);
  if constexpr (LOGL > 6) lane_stage<2>(v, t);
  if constexpr (LOGL > 7) lane_stage<3>(v, t);
}

// PP (ping-pong): pass P stores to half (P & 1) of a buffer of 2 * buf_floats(E) floats, so that a
// pass needs ONE barrier (store -> barrier -> load) instead of two; one more barrier at entry
// protects the buffer from earlier readers.  Same data movement, same results.
template <int LOGL, int P, bool PP, bool RAW = false>
__device__ __forceinline__ void fht16_passes(float v[16], float* buf, int stride, int t, bool active) {
  constexpr int NPASS = (LOGL + 3) / 4;
  if constexpr (P == 0 && LOGL > 4) {
    // passes 0 and 1 without LDS (fht16_lanes); longer transforms hand over in natural order -- pass 0's store
    // mapping -- in the half pass 1 would have written
    if constexpr (NPASS <= 2) wg_barrier<RAW>();   // callers order their own LDS data (reduction slots, staged rows) across this call
    fht16_lanes<LOGL>(v, t);
    if constexpr (NPASS <= 2) wg_barrier<RAW>();
    if constexpr (NPASS > 2) {
      float* cur = PP ? buf + stride : buf;
      wg_barrier<RAW>();   // the buffer's earlier readers are done
      if (active) FhtPass<LOGL, 0>::store(v, cur, t);
      wg_barrier<RAW>();
      fht16_passes<LOGL, 2, PP, RAW>(v, buf, stride, t, active);
    }
  } else if constexpr (P < NPASS) {
    using Pass = FhtPass<LOGL, P>;
    float* cur = PP ? buf + (P & 1) * stride : buf;
    float* prev = PP ? buf + ((P + 1) & 1) * stride : buf;
    if (P > 0 && active) Pass::load(v, prev, t);
    Pass::butterflies(v);
    if (NPASS > 1) {
      if (!PP || P == 0) wg_barrier<RAW>();  // !PP: everyone has read its pass-P inputs; PP: entry barrier
      if (active) Pass::store(v, cur, t);
      wg_barrier<RAW>();
    }
    fht16_passes<LOGL, P + 1, PP, RAW>(v, buf, stride, t, active);
  }
}

template <int LOGL, bool PP, bool RAW = false>
__device__ __forceinline__ void fht16_fixed(float v[16], float* buf, int stride, int t, bool active) {
  fht16_passes<LOGL, 0, PP, RAW>(v, buf, stride, t, active);
  constexpr int NPASS = (LOGL + 3) / 4;
  if (NPASS > 2 && active) {  // back to 16 consecutive elements per thread (up to 2^8 they never left)
    const float* b = buf + (PP ? ((NPASS - 1) & 1) * stride : 0) + (16 * t + ((16 * t) >> 5));
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = b[r];
  }
}

// run-time dispatch over the lengths the kernels take (2^6 .. 2^14).  pp_stride = 0: single buffer;
// otherwise the distance (floats) between the two halves of a ping-pong buffer.
template <bool PP>
__device__ __forceinline__ void fht16_dispatch(float v[16], float* buf, int stride, int t, int logL, bool active) {
  switch (logL) {
    case 6: fht16_fixed<6, PP>(v, buf, stride, t, active); break;
    case 7: fht16_fixed<7, PP>(v, buf, stride, t, active); break;
    case 8: fht16_fixed<8, PP>(v, buf, stride, t, active); break;
    case 9: fht16_fixed<9, PP>(v, buf, stride, t, active); break;
    case 10: fht16_fixed<10, PP>(v, buf, stride, t, active); break;
    case 11: fht16_fixed<11, PP>(v, buf, stride, t, active); break;
    case 12: fht16_fixed<12, PP>(v, buf, stride, t, active); break;
    case 13: fht16_fixed<13, PP>(v, buf, stride, t, active); break;
    default: fht16_fixed<14, PP>(v, buf, stride, t, active); break;
  }
}
__device__ __forceinline__ void fht16(float v[16], float* buf, int t, int logL, bool active, int pp_stride = 0) {
  if (pp_stride) fht16_dispatch<true>(v, buf, pp_stride, t, logL, active);
  else fht16_dispatch<false>(v, buf, 0, t, logL, active);
}

// digit split of a block fixed point value (balanced int8 digits, see e8p_gemv_i8.hip)
__device__ __forceinline__ void digits_of(int X, int& h, int& m, int& l) {
  l = (X << 24) >> 24;
  const int X1 = (X - l) >> 8;
  m = (X1 << 24) >> 24;
  h = (X1 - m) >> 8;
}

// shift for |v| <= bound: bound < 2^(E+1) => |rint(v * 2^sh)| < 2^22 with sh = 21 - E
// A bound that is not finite (an inf or NaN activation; absmax16() turns a NaN into +inf so that fmaxf cannot
// drop it) gives kShiftNotFinite, which unscale_of() turns into a NaN output row like the fp path's.
__device__ __forceinline__ int shift_for(float bound) {
  if (!(bound < __builtin_inff())) return kShiftNotFinite;
  int E = (int)((as_u32(bound) >> 23) & 0xff) - 127;
  E = max(-60, min(60, E));
  return 21 - E;
}

__device__ __forceinline__ float absmax16(const float v[16], float scale) {
#pragma clang fp contract(off)
  float mx = 0.f;
#pragma unroll
  for (int r = 0; r < 16; r += 2) {   // the products two at a time (v_pk_mul_f32: fmul's rounding)
    const f32x2 p = f32x2{v[r], v[r + 1]} * f32x2{scale, scale};
    const float a = fabsf(p.x), b = fabsf(p.y);
    mx = fmaxf(mx, a == a ? a : __builtin_inff());
    mx = fmaxf(mx, b == b ? b : __builtin_inff());
  }
  return mx;
}

// 16 transformed values -> three 16-byte digit pieces (h, m, l planes) at block exponent sh.
// Balanced digits without sign extension: l = sext8(X) means X - l = 256 * floor((X + 128) / 256), so
// X1 = (X + 128) >> 8 and h = (X1 + 128) >> 8 (arithmetic shifts), and the plane bytes are the low bytes of
// X, X1, h -- gathered four at a time with v_perm_b32 (same bytes as digits_of(), ~half the instructions:
// this runs on one wave per SIMD at the end of a latency-bound launch).
__device__ __forceinline__ uint32_t low_bytes4(int a, int b, int c, int d) {
  const uint32_t lo = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0400u);   // [a0, b0, 0, 0]
  const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x04000c0cu);   // [0, 0, c0, d0]
  return lo | hi;
}
__device__ __forceinline__ void planes16(const float v[16], float scale, int sh, uint4 out[3]) {
#pragma clang fp contract(off)
  const float s2 = fmul(scale, as_f32((uint32_t)(sh + 127) << 23));
  uint32_t dg[3][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    int X[4], X1[4], H[4];
    const f32x2 p01 = f32x2{v[4 * g], v[4 * g + 1]} * f32x2{s2, s2}, p23 = f32x2{v[4 * g + 2], v[4 * g + 3]} * f32x2{s2, s2};
    const float pr[4] = {p01.x, p01.y, p23.x, p23.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      X[i] = (int)__builtin_rintf(pr[i]);
      X1[i] = (X[i] + 128) >> 8;
      H[i] = (X1[i] + 128) >> 8;
    }
    dg[0][g] = low_bytes4(H[0], H[1], H[2], H[3]);
    dg[1][g] = low_bytes4(X1[0], X1[1], X1[2], X1[3]);
    dg[2][g] = low_bytes4(X[0], X[1], X[2], X[3]);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) out[d] = make_uint4(dg[d][0], dg[d][1], dg[d][2], dg[d][3]);
}

// output-side element: ((v * scale) * post + bias) + residual, rounded once to fp16
__device__ __forceinline__ f16 out_elem(float v, float scale, bool has_post, float post, bool has_bias, float bias,
                                        bool has_res, float res) {
  float w = fmul(v, scale);
  if (has_post) w = fmul(w, post);
  if (has_bias) w = fadd(w, bias);
  if (has_res) w = fadd(w, res);
  return (f16)w;
}

}  // namespace had
}  // namespace quip
