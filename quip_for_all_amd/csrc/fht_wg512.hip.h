// 4096-point Walsh-Hadamard transforms across a whole 512-thread workgroup, 8 elements per thread, for the
// persistent decode engine (decode_block.hip): NT independent transforms go through ONE LDS exchange and ONE pair of
// barriers.  The register-blocked transform of had_device.hip.h (16 elements per thread on 256 threads, three passes,
// four barriers, 64 LDS accesses per thread) costs ~3.5K clocks per call inside the engine, where every workgroup runs
// seven of them per decoder block on the critical path; this one is ~1.3K (NT = 1) .. 2.7K (NT = 3).
//
// Same numbers, bit for bit: a Walsh-Hadamard butterfly network applied in ascending order of the index bit computes
// every element by the same additions whatever thread holds it -- stage b replaces (x0, x1), the pair that differs in
// bit b, by (x0 + x1, x0 - x1) -- and had_device.hip.h applies bits 0..11 in that order too.
//
//   in : thread t holds x[8 t + r], r = 0..7                    (index bits 0..2 register, 3..8 lane, 9..11 wave)
//   stages 0..2 in registers, 3..6 by DPP inside 16-lane rows, 7 by ds_swizzle (xor 16), 8 by ds_bpermute (xor 32)
//   exchange through LDS (padded: one word per 32), stages 9..11 in registers
//   out: thread t holds X[t + 512 k], k = 0..7
#pragma once
#include "had_device.hip.h"

namespace quip {
namespace had8 {

constexpr int kN = 4096, kThreads = 512;
constexpr int kBufFloats = kN + (kN >> 5);          // exchange buffer of one transform

template <int STRIDE>
__device__ __forceinline__ void reg_stage(float v[8]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    if (!(r & STRIDE)) {
      const float x0 = v[r], x1 = v[r | STRIDE];
      v[r] = x0 + x1;
      v[r | STRIDE] = x0 - x1;
    }
  }
}

// partner's value of a lane stage: lane ^ 1, 2, 4, 8 (DPP / swizzle as in had_device.hip.h), 16 (swizzle), 32 (bpermute)
template <int S>
__device__ __forceinline__ float lane_partner(float v, int lane) {
  if constexpr (S < 4) return had::lane_xor<S>(v);
  else if constexpr (S == 4)
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));   // bit mode: xor 16
  else
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
}
#ifndef QUIP_FHT8_HALF_SWAP
#define QUIP_FHT8_HALF_SWAP 1
#endif
template <int S>
__device__ __forceinline__ void lane_stage(float v[8], int lane) {
#pragma clang fp contract(off)
  if constexpr (S == 5 && QUIP_FHT8_HALF_SWAP) {
    // lane bit 5 (the partner is 32 lanes away) on gfx950's half swap instead of ds_bpermute (round 4, measured on the
    // 8192-point transforms: -0.5K of 5K clocks): for a PAIR of registers (x, y)
    //   swap(x, y) -> x' = (x lower half | y lower half), y' = (x upper half | y upper half);  s = x' + y', d = x' - y';
    //   swap(s, d) -> (x_lo + x_hi | x_lo - x_hi), the same of y
    // -- own + partner where the bit is clear, partner - own where it is set: the values of the fma form, bit for bit
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
      const auto t = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[r]), __builtin_bit_cast(unsigned, v[r + 1]), false, false);
      const float x1 = __builtin_bit_cast(float, (unsigned)t[0]), y1 = __builtin_bit_cast(float, (unsigned)t[1]);
      const float sm = x1 + y1, df = x1 - y1;
      const auto u = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, sm), __builtin_bit_cast(unsigned, df), false, false);
      v[r] = __builtin_bit_cast(float, (unsigned)u[0]);
      v[r + 1] = __builtin_bit_cast(float, (unsigned)u[1]);
    }
  } else {
    const float sg = ((lane >> S) & 1) ? -1.f : 1.f;     // bit clear: x0 + x1 = own + partner; bit set: x0 - x1 = partner - own
    float par[8];                                        // (the partners in a batch: one wait for the crossbar, fht_wg512x.hip.h)
#pragma unroll
    for (int r = 0; r < 8; ++r) par[r] = lane_partner<S>(v[r], lane);
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = __builtin_fmaf(v[r], sg, par[r]);
  }
}

// RAW: barriers that do not wait for vector-memory loads in flight (had::wg_barrier)
template <int NT, bool RAW>
__device__ __forceinline__ void fht4096(float (&v)[NT][8], float* xbuf, int tid) {
  const int lane = tid & 63;
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    reg_stage<1>(v[i]);
    reg_stage<2>(v[i]);
    reg_stage<4>(v[i]);
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) lane_stage<0>(v[i], lane);
#pragma unroll
  for (int i = 0; i < NT; ++i) lane_stage<1>(v[i], lane);
#pragma unroll
  for (int i = 0; i < NT; ++i) lane_stage<2>(v[i], lane);
#pragma unroll
  for (int i = 0; i < NT; ++i) lane_stage<3>(v[i], lane);
#pragma unroll
  for (int i = 0; i < NT; ++i) lane_stage<4>(v[i], lane);
#pragma unroll
  for (int i = 0; i < NT; ++i) lane_stage<5>(v[i], lane);
  had::wg_barrier<RAW>();                              // whatever the buffer held (the transform's input, usually) has been read
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    float* b = xbuf + i * kBufFloats + (8 * tid + (tid >> 2));      // pad(8 t + r) = 8 t + r + (t >> 2), r < 8
#pragma unroll
    for (int r = 0; r < 8; ++r) b[r] = v[i][r];
  }
  had::wg_barrier<RAW>();
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const float* b = xbuf + i * kBufFloats + (tid + (tid >> 5));    // pad(t + 512 k) = t + (t >> 5) + 528 k
#pragma unroll
    for (int k = 0; k < 8; ++k) v[i][k] = b[528 * k];
    reg_stage<1>(v[i]);
    reg_stage<2>(v[i]);
    reg_stage<4>(v[i]);
  }
}

// ---- reductions that reproduce had_device.hip.h's values ---------------------------------------------------------------
// Sum of squares of a 4096-vector held 8 consecutive elements per thread (element 8 t + r), with the summation order of
// the 256-thread kernels: a 16-element fma chain per pair of threads (sumsq8 twice), the DPP tree over 64 such pairs
// (bits 0..5 of the pair index), then ((w0 + w1) + w2) + w3 over the four waves of those kernels.  red: 16 floats of LDS.
template <bool RAW>
__device__ __forceinline__ float sumsq4096(const float e[8], float* red, int tid) {
#pragma clang fp contract(off)
  const int lane = tid & 63, wave = tid >> 6;
  // the chain of the even thread, continued by the odd one
  float ss = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) ss = __builtin_fmaf(e[r], e[r], ss);
  const float lower = had::lane_xor<0>(ss);           // odd lane: the even lane's partial chain
  if (lane & 1) {
    ss = lower;
#pragma unroll
    for (int r = 0; r < 8; ++r) ss = __builtin_fmaf(e[r], e[r], ss);
  }
  // tree over pair-index bits 0..4 = lane bits 1..5 (the odd lanes hold the pairs' sums; even lanes carry copies along)
  ss = had::fadd(ss, lane_partner<1>(ss, lane));
  ss = had::fadd(ss, lane_partner<2>(ss, lane));
  ss = had::fadd(ss, lane_partner<3>(ss, lane));
  ss = had::fadd(ss, lane_partner<4>(ss, lane));
  ss = had::fadd(ss, lane_partner<5>(ss, lane));
  had::wg_barrier<RAW>();
  if (lane == 63) red[wave] = ss;
  had::wg_barrier<RAW>();
  // pair-index bit 5 = this layout's wave bit 0, then the four 256-thread-kernel waves in order
  const float w0 = had::fadd(red[0], red[1]), w1 = had::fadd(red[2], red[3]);
  const float w2 = had::fadd(red[4], red[5]), w3 = had::fadd(red[6], red[7]);
  return had::fadd(had::fadd(had::fadd(w0, w1), w2), w3);
}

// maximum of non-negative values over the workgroup (any order gives the same value)
template <bool RAW>
__device__ __forceinline__ float max4096(float mx, float* red, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  mx = had::wave_reduce_to_lane63<true>(mx);
  had::wg_barrier<RAW>();
  if (lane == 63) red[wave] = mx;
  had::wg_barrier<RAW>();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) r = fmaxf(r, red[w]);
  return r;
}

// |v * scale| maximum of the thread's 8 values (absmax16's arithmetic: the products rounded on their own, NaN -> +inf)
__device__ __forceinline__ float absmax8(const float v[8], float scale) {
#pragma clang fp contract(off)
  float mx = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const float a = fabsf(v[r] * scale);
    mx = fmaxf(mx, a == a ? a : __builtin_inff());
  }
  return mx;
}

// digit planes of the transformed vector: this thread's values are X[t + 512 k]; plane d at planes + d * 4096 (planes16's
// arithmetic per element)
__device__ __forceinline__ void planes_scatter(const float v[8], float scale, int sh, uint8_t* planes, int tid) {
#pragma clang fp contract(off)
  const float s2 = had::fmul(scale, as_f32((uint32_t)(sh + 127) << 23));
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int X = (int)__builtin_rintf(v[k] * s2);
    const int X1 = (X + 128) >> 8;
    const int H = (X1 + 128) >> 8;
    uint8_t* p = planes + tid + 512 * k;
    p[0] = (uint8_t)H;
    p[kN] = (uint8_t)X1;
    p[2 * kN] = (uint8_t)X;
  }
}

// E8P12RVQ4B: the digits of x' = [s x_g | x_g]_g (hadamard.hip, rvq_scale): element e's residual-side digits at 2 (e & ~7) +
// (e & 7), its main-side digits 8 further; plane d at planes + d * 2 * kN; planes16's arithmetic with scale * rs / scale
__device__ __forceinline__ void planes_scatter_rvq(const float v[8], float scale, float rs, int sh, uint8_t* planes, int tid) {
#pragma clang fp contract(off)
  const float p2 = as_f32((uint32_t)(sh + 127) << 23);
  const float s2m = had::fmul(scale, p2), s2r = had::fmul(had::fmul(scale, rs), p2);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = tid + 512 * k;
    uint8_t* p = planes + 2 * (e & ~7) + (e & 7);
    const int Xm = (int)__builtin_rintf(v[k] * s2m), Xr = (int)__builtin_rintf(v[k] * s2r);
    const int X1m = (Xm + 128) >> 8, X1r = (Xr + 128) >> 8;
    const int Hm = (X1m + 128) >> 8, Hr = (X1r + 128) >> 8;
    p[0] = (uint8_t)Hr;  p[8] = (uint8_t)Hm;
    p[2 * kN] = (uint8_t)X1r;  p[2 * kN + 8] = (uint8_t)X1m;
    p[4 * kN] = (uint8_t)Xr;  p[4 * kN + 8] = (uint8_t)Xm;
  }
}

// HI: the digits of x' = [x0 x2 0 0 | x4 x6 0 0 | x1 x3 0 0 | x5 x7 0 0]_g (hadamard.hip, HI layout): element i of an
// 8-group at virtual position 8 (i & 1) + 4 (i >> 2) + ((i >> 1) & 1) of its 16-group, a zero digit two positions further;
// plane d at planes + d * 2 * kN; planes16's arithmetic
__device__ __forceinline__ void planes_scatter_hi(const float v[8], float scale, int sh, uint8_t* planes, int tid) {
#pragma clang fp contract(off)
  const float s2 = had::fmul(scale, as_f32((uint32_t)(sh + 127) << 23));
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int e = tid + 512 * k, i = e & 7;
    uint8_t* p = planes + 2 * (e & ~7) + (((i & 1) << 3) | ((i >> 2) << 2) | ((i >> 1) & 1));
    const int X = (int)__builtin_rintf(v[k] * s2);
    const int X1 = (X + 128) >> 8;
    const int H = (X1 + 128) >> 8;
    p[0] = (uint8_t)H;  p[2] = 0;
    p[2 * kN] = (uint8_t)X1;  p[2 * kN + 2] = 0;
    p[4 * kN] = (uint8_t)X;  p[4 * kN + 2] = 0;
  }
}

}  // namespace had8
}  // namespace quip
