// Walsh-Hadamard transforms of 4096 (8 elements per thread) or 8192 (16 per thread) points across a 512-thread workgroup
// with ONE LDS exchange, in both directions, for the persistent decode engine of the grouped-query / 8192-wide shapes
// (decode_block_gqa.hip).  Generalises fht_wg512.hip.h (4096 points, natural -> strided only):
//
//   fwd: thread t holds x[EPT t + r]  ->  thread t holds X[t + 512 k]     (index bits in ascending order: the additions of
//        had_device.hip.h's transforms, so the same bits as the stand-alone kernels)
//   rev: thread t holds x[t + 512 k]  ->  thread t holds X[EPT t + r]     (bits 9.. first, then 0..8: same transform,
//        another order of the additions)
//
// The pair lets an edge of the decoder block run  gather (natural) -> fwd -> element-wise in the strided layout
// (residual, RMSNorm, SU: the static vectors are stored pre-permuted by the host) -> rev -> digit planes as 16-byte
// pieces, without a layout change through LDS in between.  wave_fht1024: 1024 points inside one wave, no LDS at all.
#pragma once
#include "fht_wg512.hip.h"

namespace quip {
namespace hadw {

// two butterflies per instruction (v_pk_add_f32 on the register pairs (v[2 i], v[2 i + 1])): the IEEE additions of the
// one-at-a-time form (had_device.hip.h, FhtPass::butterflies), half the issue slots -- the transforms are VALU bound
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int N, int STRIDE>
__device__ __forceinline__ void reg_stage(float (&v)[N]) {
#pragma clang fp contract(off)
  if constexpr (STRIDE == 1) {
#pragma unroll
    for (int r = 0; r < N; r += 2) {      // partners inside a pair: (a + b, a - b)
      const f32x2 a = {v[r], v[r]}, b = {v[r + 1], -v[r + 1]};
      const f32x2 c = a + b;
      v[r] = c.x;
      v[r + 1] = c.y;
    }
  } else {
#pragma unroll
    for (int r = 0; r < N; r += 2) {
      if (!(r & STRIDE)) {
        const int q = r | STRIDE;
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
template <int N, int FROM>
__device__ __forceinline__ void reg_stages(float (&v)[N]) {      // strides FROM, 2 FROM, .. < N
  if constexpr (FROM < N) {
    reg_stage<N, FROM>(v);
    reg_stages<N, 2 * FROM>(v);
  }
}
#ifndef QUIP_FHT_SWAP_STAGES
#define QUIP_FHT_SWAP_STAGES 1
#endif
#ifndef QUIP_FHT_SWAP_FROM
#define QUIP_FHT_SWAP_FROM 5      // measured: the half swap (bit 5) beats ds_bpermute, the row swap (bit 4) loses to ds_swizzle
#endif
template <int N, int S>
__device__ __forceinline__ void lane_stage(float (&v)[N], int lane) {
#pragma clang fp contract(off)
  if constexpr (QUIP_FHT_SWAP_STAGES && S >= QUIP_FHT_SWAP_FROM && N % 2 == 0) {
    // lane bits 4 / 5 (partners 16 / 32 lanes away) on gfx950's row / half swaps instead of the LDS crossbar (ds_swizzle,
    // ds_bpermute): for a PAIR of registers (x, y)
    //   swap(x, y)        -> x' = x's even rows | y's even rows interleaved, y' = the odd rows (S = 4); lower | upper halves (S = 5)
    //   s = x' + y', d = x' - y'
    //   swap(s, d)        -> x'' = (x_lo + x_hi | x_lo - x_hi), y'' = the same of y
    // i.e. own + partner where the bit is clear, partner - own where it is set: the same two IEEE operations as the fma form
#pragma unroll
    for (int r = 0; r < N; r += 2) {
      const unsigned a = __builtin_bit_cast(unsigned, v[r]), b = __builtin_bit_cast(unsigned, v[r + 1]);
      const auto t = S == 4 ? __builtin_amdgcn_permlane16_swap(a, b, false, false) : __builtin_amdgcn_permlane32_swap(a, b, false, false);
      const float x1 = __builtin_bit_cast(float, (unsigned)t[0]), y1 = __builtin_bit_cast(float, (unsigned)t[1]);
      const float sm = x1 + y1, df = x1 - y1;
      const auto u = S == 4 ? __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, sm), __builtin_bit_cast(unsigned, df), false, false)
                            : __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, sm), __builtin_bit_cast(unsigned, df), false, false);
      v[r] = __builtin_bit_cast(float, (unsigned)u[0]);
      v[r + 1] = __builtin_bit_cast(float, (unsigned)u[1]);
    }
  } else {
    const float sg = ((lane >> S) & 1) ? -1.f : 1.f;     // bit clear: own + partner; bit set: partner - own
    const f32x2 sg2 = {sg, sg};
    // all the partners first, then the arithmetic: lane bits 2 and 4 go through the LDS crossbar (ds_swizzle), and written
    // pair by pair the compiler waits for every pair of swizzles on its own (lgkmcnt(0) four to eight times per stage, ~60
    // clocks each: profiles/r05_block_stamps.txt) -- in a batch they are one wait
    float par[N];
#pragma unroll
    for (int r = 0; r < N; ++r) par[r] = had8::lane_partner<S>(v[r], lane);
#pragma unroll
    for (int r = 0; r < N; r += 2) {
      const f32x2 own = {v[r], v[r + 1]}, pp = {par[r], par[r + 1]};
      const f32x2 w = __builtin_elementwise_fma(own, sg2, pp);
      v[r] = w.x;
      v[r + 1] = w.y;
    }
  }
}
template <int N, int S, int END>
__device__ __forceinline__ void lane_stages(float (&v)[N], int lane) {     // lane bits S .. END - 1
  if constexpr (S < END) {
    lane_stage<N, S>(v, lane);
    lane_stages<N, S + 1, END>(v, lane);
  }
}

template <int LOGN>
struct Geo {
  static constexpr int N = 1 << LOGN, EPT = N / 512, RB = LOGN - 9;
  static constexpr int kBufFloats = N + (N >> 5) + 4;
  static_assert(EPT == 8 || EPT == 16, "4096 or 8192 points");
  // natural element EPT t + r  ->  padded index (one word per 32)
  __device__ static __forceinline__ int nat(int t) { return EPT * t + ((EPT * t) >> 5); }
  // strided element t + 512 k  ->  t + (t >> 5) + 528 k
  __device__ static __forceinline__ int str(int t) { return t + (t >> 5); }
};

// natural -> strided (ascending index bits)
template <int LOGN, int NT, bool RAW>
__device__ __forceinline__ void fwd(float (&v)[NT][Geo<LOGN>::EPT], float* xbuf, int tid) {
  using G = Geo<LOGN>;
  constexpr int E = G::EPT;
  const int lane = tid & 63;
#pragma unroll
  for (int i = 0; i < NT; ++i) reg_stages<E, 1>(v[i]);
#pragma unroll
  for (int i = 0; i < NT; ++i) lane_stages<E, 0, 6>(v[i], lane);          // index bits RB .. RB + 5
  had::wg_barrier<RAW>();
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    float* b = xbuf + i * G::kBufFloats + G::nat(tid);
#pragma unroll
    for (int r = 0; r < E; ++r) b[r] = v[i][r];
  }
  had::wg_barrier<RAW>();
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const float* b = xbuf + i * G::kBufFloats + G::str(tid);
#pragma unroll
    for (int k = 0; k < E; ++k) v[i][k] = b[528 * k];
    // pending index bits RB + 6 .. LOGN - 1 = bits (RB - 3) .. of k
    reg_stages<E, (1 << (G::RB - 3))>(v[i]);
  }
}

// strided -> natural.  SYNC_AFTER_READ: one more barrier right behind the LDS reads -- the caller writes over the exchange
// buffer as soon as rev returns (digit planes), and the barrier's skew hides under the stages that follow it here
template <int LOGN, int NT, bool RAW, bool SYNC_AFTER_READ = false>
__device__ __forceinline__ void rev(float (&v)[NT][Geo<LOGN>::EPT], float* xbuf, int tid) {
  using G = Geo<LOGN>;
  constexpr int E = G::EPT;
  const int lane = tid & 63;
#pragma unroll
  for (int i = 0; i < NT; ++i) reg_stages<E, 1>(v[i]);                      // index bits 9 .. LOGN - 1 (all bits of k)
  had::wg_barrier<RAW>();
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    float* b = xbuf + i * G::kBufFloats + G::str(tid);
#pragma unroll
    for (int k = 0; k < E; ++k) b[528 * k] = v[i][k];
  }
  had::wg_barrier<RAW>();
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const float* b = xbuf + i * G::kBufFloats + G::nat(tid);
#pragma unroll
    for (int r = 0; r < E; ++r) v[i][r] = b[r];
  }
  if constexpr (SYNC_AFTER_READ) had::wg_barrier<RAW>();
#pragma unroll
  for (int i = 0; i < NT; ++i) reg_stages<E, 1>(v[i]);                      // index bits 0 .. RB - 1
#pragma unroll
  for (int i = 0; i < NT; ++i) lane_stages<E, 0, 9 - G::RB>(v[i], lane);    // index bits RB .. 8
}

// 1024 points inside ONE wave: lane l holds x[16 l + r] before and X[16 l + r] after; no LDS, no barrier
__device__ __forceinline__ void wave_fht1024(float (&v)[16], int lane) {
  reg_stages<16, 1>(v);
  lane_stages<16, 0, 6>(v, lane);
}

// The three balanced digits of X (had::digits_of) as bytes, without the shifts: l = byte 0 of X, m = byte 0 of (X + 128) >> 8 =
// byte 1 of X + 0x80, h = byte 0 of (((X + 128) >> 8) + 128) >> 8 = byte 2 of X + 0x8080.  Byte B of four values -> one word.
template <int BYTE>
__device__ __forceinline__ uint32_t bytes4(int a, int b, int c, int d) {
  constexpr uint32_t lo = 0x0c0c0400u + BYTE * 0x0101u, hi = 0x04000c0cu + BYTE * 0x01010000u;
  return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, lo) | __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, hi);
}
__device__ __forceinline__ void digit_words(const int (&X)[4], uint32_t& h, uint32_t& m, uint32_t& l) {
  l = bytes4<0>(X[0], X[1], X[2], X[3]);
  m = bytes4<1>(X[0] + 0x80, X[1] + 0x80, X[2] + 0x80, X[3] + 0x80);
  h = bytes4<2>(X[0] + 0x8080, X[1] + 0x8080, X[2] + 0x8080, X[3] + 0x8080);
}

// The same digits straight from fp32: W = bits(fma(v, s, 1.5 * 2^23)) = 0x4B400000 + X with X = rint(v s) (the fma rounds to
// the integer grid, ties to even, for |v s| < 2^22 -- what rintf + v_cvt_i32_f32 compute in two more instructions per value):
// byte 0 of W = byte 0 of X, byte 1 of W + 0x80 = byte 1 of X + 0x80, byte 2 of W + 0xC08080 = byte 2 of X + 0x8080
// (0xC0 takes the 0x40 of the bias out of byte 2, modulo 256).
constexpr float kMagic = 12582912.f;      // 1.5 * 2^23
__device__ __forceinline__ void digit_words_magic(const float (&v)[4], float s, uint32_t& h, uint32_t& m, uint32_t& l) {
#pragma clang fp contract(off)
  int W[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) W[i] = (int)__builtin_bit_cast(uint32_t, __builtin_fmaf(v[i], s, kMagic));
  l = bytes4<0>(W[0], W[1], W[2], W[3]);
  m = bytes4<1>(W[0] + 0x80, W[1] + 0x80, W[2] + 0x80, W[3] + 0x80);
  h = bytes4<2>(W[0] + 0xC08080, W[1] + 0xC08080, W[2] + 0xC08080, W[3] + 0xC08080);
}

// sum of squares / maximum over the workgroup, a fixed order (every workgroup of a launch computes the same value from the
// same data): the thread's chain, the wave's DPP tree, the eight waves in order.  red: 8 floats per call site.
template <int N, bool RAW>
__device__ __forceinline__ float sumsq(const float (&e)[N], float* red, int tid) {
#pragma clang fp contract(off)
  const int lane = tid & 63, wave = tid >> 6;
  float ss = 0.f;
#pragma unroll
  for (int r = 0; r < N; ++r) ss = __builtin_fmaf(e[r], e[r], ss);
  ss = had::wave_reduce_to_lane63<false>(ss);
  had::wg_barrier<RAW>();
  if (lane == 63) red[wave] = ss;
  had::wg_barrier<RAW>();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < 8; ++w) r = had::fadd(r, red[w]);
  return r;
}
template <int N>
__device__ __forceinline__ float absmax(const float (&v)[N], float scale) {
#pragma clang fp contract(off)
  float mx = 0.f;
#pragma unroll
  for (int r = 0; r < N; ++r) {
    const float a = fabsf(v[r] * scale);
    mx = fmaxf(mx, a == a ? a : __builtin_inff());
  }
  return mx;
}

}  // namespace hadw
}  // namespace quip
