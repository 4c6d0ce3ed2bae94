// Randomised-Hadamard side of QuantLinear.forward on gfx950.
//
//   y = residual + post (.) ( scale * (H (x) H_L) v )[:out_features] + bias,
//   v = rms(x) * rms_w (.) act(gate) (.) pre (.) pre2 (.) x      (zero padded to n = K * L)
//
// where the padded row is viewed row-major as (K, L), L a power of two, H_L is the Sylvester
// Walsh-Hadamard matrix applied along L and H is the K x K factor `had` (or had^T) applied
// along K -- quant.py:72-88 (matmul_hadU_cuda / matmul_hadUt_cuda) fused with the element-wise
// ops around it in qlinear.py:90-91 (x*SU), :106-107 (per-channel Wscale), :108-114 (slice,
// *SV, +bias).  With K == 1 and no vectors this is quip_lib::hadamard (register_lib.py:10-20).
// The optional rms / gate / residual hooks fuse the decoder-block glue around a QuantLinear
// (RMSNorm before q/k/v/gate/up, SiLU(gate)*up before down, residual add after o/down).
//
// (H (x) H_L) = (I (x) H_L)(H (x) I): workgroup (kp, row) first forms row kp of the K-mix,
// t[j] = sum_k H[kp,k] v[k*L + j] (independent per j), then runs the length-L transform.
// K workgroups per token row run in parallel: the 43x43 / 7x7 factors of 11008 / 28672 cost
// one extra pass over the (L2 resident) input row per workgroup.
//
// The length-L transform is register blocked: a thread owns 16 elements, does 4 butterfly
// stages in registers, and the workgroup re-shuffles through LDS between passes (3 passes for
// L = 4096).  Lengths < 256 or > 16384 take the simple LDS radix-2 kernel.
//
// Output either fp16, or (input side of the bs=1 decode path) the block fixed point int8
// digit planes the matrix-core GEMV consumes (e8p_gemv_mfma.hip): planes[d][Kp], d = h, m, l
// digits of X = rint(v * 2^sh), |X| < 2^22, then the int shift word.  No fp16 rounding happens
// between the transform and the GEMV.  The shift comes from a bound every workgroup can
// compute alone: K == 1: the exact max |v| of the transformed row; K > 1: |v_i| <= ||v||_2 =
// scale * sqrt(L) * ||H||_2 * ||input||_2 with H ~ orthogonal.
#include "had_device.hip.h"
#include "quip_internal.h"

namespace quip {

#ifdef QUIP_HAD_STAMPS
__device__ unsigned long long g_had_stamps[16];
#define HSTAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0) g_had_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int quip_had_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_had_stamps), sizeof(g_had_stamps)) == hipSuccess ? 0 : -1;
}
#else
#define HSTAMP(i) do {} while (0)
#endif

namespace {

struct HadArgs {
  const f16* x;
  const f16* had;       // (K, K) or null
  const f16* pre;       // [in_features] or null   (SU)
  const f16* pre2;      // [n] or null              (per-channel Wscale)
  const f16* post;      // [out_features] or null   (SV)
  const f16* bias;      // [out_features] or null
  const f16* residual;  // [rows, out_features] or null, added to the output
  const f16* rms_w;     // [in_features] or null: RMSNorm weight, v *= rsqrt(mean(x^2) + eps) * rms_w
  const f16* gate;      // [rows, in_features] or null: v *= silu(gate)
  f16* y;               // fp16 output [rows, out_features] (planes == null)
  uint8_t* planes;      // digit planes output (one image of 3 Kp + 16 bytes per token row) or null
  int in_features, out_features, n, Kp, K, L, logL, transpose;
  int vec, vec_out;     // 16-byte vector loads / stores allowed (alignment + multiple-of-8 sizes)
  // chain: the input row is itself the output side of the producer module, computed here first:
  //   x = z_post (.) (z_scale * H_n z) + z_res   (rounded to fp16, stored to h_out by the z == 0 problem)
  const f16* z;         // [rows, n] or null
  const f16* z_post;    // [n]
  const f16* z_res;     // [rows, n] or null
  f16* h_out;           // [rows, n]
  float z_scale;
  int pp;               // floats between the two halves of the ping-pong shuffle buffer (0: single buffer)
  float rvq_scale;      // != 0: planes of the E8P12RVQ4B virtual vector (see the PLANES epilogue)
  int hi_layout;        // != 0: planes of the HI virtual vector (see the PLANES epilogue)
  int tgroups;          // wide K > 1: thread groups that split the k range (partials combined through LDS)
  int part_off;         // floats from buf to the partial-sum area
  float scale, rms_eps;
};

constexpr int kMaxGroup = 3;   // problems per launch (q/k/v, gate/up)
struct HadGroup { HadArgs p[kMaxGroup]; };

using had::block_reduce;
using had::f32x2;
using had::unpack8p;
using had::mul8p;
using had::add8p;
using had::pad;
using had::silu;

// input element idx of the current row, all element-wise pre-ops applied (not the rms factor)
__device__ __forceinline__ float in_val(const HadArgs& a, const f16* xr, const f16* gr, int idx) {
  if (idx >= a.in_features) return 0.f;  // F.pad (quant.py:73-74)
  float v = (float)xr[idx];
  if (a.gate) v *= silu((float)gr[idx]);
  if (a.rms_w) v *= (float)a.rms_w[idx];
  if (a.pre) v *= (float)a.pre[idx];
  if (a.pre2) v *= (float)a.pre2[idx];
  return v;
}

// 8 consecutive fp16 -> fp32 (16-byte load)
__device__ __forceinline__ void ld8(const f16* p, float o[8]) { had::unpack8(*reinterpret_cast<const uint4*>(p), o); }
__device__ __forceinline__ uint4 ldp(const f16* p) { return *reinterpret_cast<const uint4*>(p); }

// 16 consecutive input elements [idx0, idx0 + 16) of the current row with the element-wise
// pre-ops applied; ss_x accumulates the raw x^2 (RMSNorm statistic).  vec: in_features % 8 == 0
// and every vector 16-byte aligned (checked on the host), so an 8-chunk is all in or all out.
__device__ __forceinline__ void in_vals16(const HadArgs& a, const f16* xr, const f16* gr, int idx0, float e[16],
                                          float& ss_x) {
  if (a.vec) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = idx0 + 8 * h;
      float* o = e + 8 * h;
      if (c < a.in_features) {
        ld8(xr + c, o);
        if (a.rms_w) {
          had::sumsq8(o, ss_x);
          had::mul8(o, ldp(a.rms_w + c));
        }
        if (a.gate) had::silu_mul8(o, ldp(gr + c));
        if (a.pre) had::mul8(o, ldp(a.pre + c));
        if (a.pre2) had::mul8(o, ldp(a.pre2 + c));
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int idx = idx0 + r;
      e[r] = in_val(a, xr, gr, idx);
      if (a.rms_w && idx < a.in_features) { const float xv = (float)xr[idx]; ss_x = __builtin_fmaf(xv, xv, ss_x); }
    }
  }
}

// Two-phase version of in_vals16 for the vec layout: issue every 16-byte load of a chunk first
// (raw_load16), unpack / multiply later (raw_math16), so that several chunks share one memory
// round trip instead of paying one per chunk.
struct Raw16 { uint4 d[5][2]; };   // x, rms_w, gate, pre, pre2

__device__ __forceinline__ void raw_load16(const HadArgs& a, const f16* xr, const f16* gr, int idx0, Raw16& r) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = idx0 + 8 * h;
    const int cc = c < a.in_features ? c : 0;
    r.d[0][h] = *reinterpret_cast<const uint4*>(xr + cc);
    if (a.rms_w) r.d[1][h] = *reinterpret_cast<const uint4*>(a.rms_w + cc);
    if (a.gate) r.d[2][h] = *reinterpret_cast<const uint4*>(gr + cc);
    if (a.pre) r.d[3][h] = *reinterpret_cast<const uint4*>(a.pre + cc);
    if (a.pre2) r.d[4][h] = *reinterpret_cast<const uint4*>(a.pre2 + cc);
  }
}

__device__ __forceinline__ void raw_math16(const HadArgs& a, int idx0, const Raw16& r, float e[16], float& ss_x) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float* o = e + 8 * h;
    const float keep = (idx0 + 8 * h) < a.in_features ? 1.f : 0.f;   // F.pad zeros
    had::unpack8(r.d[0][h], o);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] *= keep;
    if (a.rms_w) {
      had::sumsq8(o, ss_x);
      had::mul8(o, r.d[1][h]);
    }
    if (a.gate) had::silu_mul8(o, r.d[2][h]);
    if (a.pre) had::mul8(o, r.d[3][h]);
    if (a.pre2) had::mul8(o, r.d[4][h]);
  }
}

// One workgroup transforms E = R * L elements (R rows kp of the (K, L) view), 16 per thread.
//   wide (TALL == false): R = 1, thread owns 16 consecutive columns, K-mix by looping over k with
//                         16-byte loads (K == 1, or long rows: 28672 = 7 x 4096);
//   tall (TALL == true):  64 <= L <= 256 < n, 256 threads, R = 16 * (256 / L): thread owns one
//                         column and 16 rows of the K-mix (11008 = 43 x 256 or 172 x 64), the H
//                         tile sits in LDS and is read as broadcasts.
// KONE: instantiation for K == 1 only (no K-mix code: far fewer registers, so that many token rows
// of a prefill batch are resident per CU)
template <bool PLANES, bool TALL, int MAXT, bool KONE = false>
__global__ __launch_bounds__(MAXT) void had_fast_kernel(HadGroup grp) {
  HadArgs a = grp.p[blockIdx.z];
  if constexpr (PLANES) a.planes += (size_t)blockIdx.y * ((size_t)3 * a.Kp + 16);   // one plane image per token row
  extern __shared__ __attribute__((aligned(16))) float buf[];
  __shared__ float red[32];     // two slots, each used once: no barrier before their writes
  const int tid = threadIdx.x, nt = blockDim.x;   // nt == E / 16
  const int64_t row = blockIdx.y;
  const f16* xr = a.x + row * a.in_features;
  const f16* gr = a.gate ? a.gate + row * a.in_features : nullptr;
  const int L = a.L, K = a.K, logL = a.logL;
  // tall: the tile is 4096 elements = 256 "tile threads" x 16; the workgroup has 4 x 256 threads that
  // share the row staging and split the k range of the K-mix; threads >= 256 then only keep barriers
  // wide with K > 1: a.tgroups groups of L / 16 threads each take every tgroups-th k of the K-mix
  constexpr int kTile = 256;
  const int nta = TALL ? kTile : nt / a.tgroups;   // threads that hold transform data ("tile threads")
  const int tgrp = tid / nta, tt = tid - tgrp * nta;
  const bool act = tgrp == 0;
  const int R = TALL ? (16 * kTile) >> logL : 1;
  const int kp0 = blockIdx.x * R;
  const int e0 = tt * 16;                          // first of this thread's 16 elements of [R][L]
  const int kp = kp0 + (e0 >> logL), j0 = e0 & (L - 1);

  HSTAMP(0);
  // (1) load + K-mix; sums for rms / the planes bound
  float v[16];
  float ss_x = 0.f, ss_in = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = 0.f;
  if constexpr (!TALL) {
    if ((KONE || K == 1) && a.z) {
      // producer's output transform + residual first (host guarantees vec, in_features == n == L)
      const f16* zr = a.z + row * a.n;
      float tp[16], tr[16];
      ld8(zr + j0, v); ld8(zr + j0 + 8, v + 8);
      ld8(a.z_post + j0, tp); ld8(a.z_post + j0 + 8, tp + 8);
      if (a.z_res) { ld8(a.z_res + row * a.n + j0, tr); ld8(a.z_res + row * a.n + j0 + 8, tr + 8); }
      // the consumer's element-wise vectors are requested now, one memory round trip with z (requested after
      // the transform they cost a second, fully exposed one: measured 0.8 us of a 6.3 us launch)
      uint4 qw[2], qg[2], qp[2], qp2[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int c = j0 + 8 * h;
        if (a.rms_w) qw[h] = ldp(a.rms_w + c);
        if (a.gate) qg[h] = ldp(gr + c);
        if (a.pre) qp[h] = ldp(a.pre + c);
        if (a.pre2) qp2[h] = ldp(a.pre2 + c);
      }
      HSTAMP(7);
      had::fht16(v, buf, tt, logL, true, a.pp);
      HSTAMP(8);
      f16 o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o[r] = had::out_elem(v[r], a.z_scale, true, tp[r], false, 0.f, a.z_res != nullptr, a.z_res ? tr[r] : 0.f);
      if (blockIdx.z == 0) {
        uint4* dst = reinterpret_cast<uint4*>(a.h_out + row * a.n + j0);
        dst[0] = *reinterpret_cast<uint4*>(&o[0]);
        dst[1] = *reinterpret_cast<uint4*>(&o[8]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = (float)o[r];
      // then the element-wise input ops of this module, as in_vals16 applies them
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float* e = v + 8 * h;
        if (a.rms_w) {
          had::sumsq8(e, ss_x);
          had::mul8(e, qw[h]);
        }
        if (a.gate) had::silu_mul8(e, qg[h]);
        if (a.pre) had::mul8(e, qp[h]);
        if (a.pre2) had::mul8(e, qp2[h]);
      }
      __syncthreads();   // the shuffle buffer is reused by the transform below
      HSTAMP(9);
    } else if (KONE || K == 1) {
      in_vals16(a, xr, gr, kp * L + j0, v, ss_x);
    } else if constexpr (!KONE) {
      constexpr int U = MAXT <= 256 ? 2 : 1;       // k values per memory round trip
      const int TG = a.tgroups;
      for (int k0 = tgrp; k0 < K; k0 += U * TG) {
        float h[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int k = min(k0 + u * TG, K - 1);
          h[u] = (k0 + u * TG) < K ? (float)(a.transpose ? a.had[k * K + kp] : a.had[kp * K + k]) : 0.f;
        }
        if (a.vec) {
          Raw16 raw[U];
#pragma unroll
          for (int u = 0; u < U; ++u) raw_load16(a, xr, gr, min(k0 + u * TG, K - 1) * L + j0, raw[u]);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            float e[16];
            float sx = 0.f;
            raw_math16(a, min(k0 + u * TG, K - 1) * L + j0, raw[u], e, sx);
            if (k0 + u * TG < K) {
              ss_x += sx;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                ss_in = __builtin_fmaf(e[r], e[r], ss_in);
                v[r] = __builtin_fmaf(h[u], e[r], v[r]);
              }
            }
          }
        } else {
          for (int u = 0; u < U && k0 + u * TG < K; ++u) {
            float e[16];
            in_vals16(a, xr, gr, (k0 + u * TG) * L + j0, e, ss_x);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              ss_in = __builtin_fmaf(e[r], e[r], ss_in);
              v[r] = __builtin_fmaf(h[u], e[r], v[r]);
            }
          }
        }
      }
      if (TG > 1) {   // combine the groups' partial sums in a fixed order: 0 + 1 + 2 + 3
        float* part = buf + a.part_off;
        if (tgrp > 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r) part[((tgrp - 1) * 16 + r) * nta + tt] = v[r];
        }
        __syncthreads();
        if (act) {
          for (int o = 1; o < TG; ++o) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = had::fadd(v[r], part[((o - 1) * 16 + r) * nta + tt]);
          }
        }
      }
    }
  } else {
    float* hs = buf + had::buf_floats(16 * kTile);        // [K][R] tile of H (rows kp0..kp0+R)
    for (int i = tid; i < ((K + 3) & ~3) * R; i += nt) {
      const int k = i / R, rr = i - k * R, kq = kp0 + rr;
      hs[i] = (kq < K && k < K) ? (float)(a.transpose ? a.had[k * K + kq] : a.had[kq * K + k]) : 0.f;
    }
    HSTAMP(1);
    // stage the pre-processed input row in LDS (one memory round trip for the whole row)
    float* xs = hs + ((K + 3) & ~3) * R;                  // [K][L + 8] (padded rows: fewer bank conflicts)
    const int xstride = L + 8;
    if (a.vec) {
      for (int c = tid; c * 16 < a.n; c += nt) {
        Raw16 raw;
        raw_load16(a, xr, gr, c * 16, raw);
        float e[16];
        raw_math16(a, c * 16, raw, e, ss_x);
#pragma unroll
        for (int r = 0; r < 16; ++r) ss_in = __builtin_fmaf(e[r], e[r], ss_in);
        float* dst = xs + ((c * 16) >> logL) * xstride + ((c * 16) & (L - 1));
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(dst + 4 * q) = make_float4(e[4 * q], e[4 * q + 1], e[4 * q + 2], e[4 * q + 3]);
      }
    } else {
      for (int i = tid; i < a.n; i += nt) {
        const float e = in_val(a, xr, gr, i);
        if (a.rms_w && i < a.in_features) { const float xv = (float)xr[i]; ss_x = __builtin_fmaf(xv, xv, ss_x); }
        ss_in = __builtin_fmaf(e, e, ss_in);
        xs[(i >> logL) * xstride + (i & (L - 1))] = e;
      }
    }
    __syncthreads();
    HSTAMP(2);
    // K-mix on the matrix cores (fp32 in, fp32 accumulate): out[r][j] = sum_k H[kp0 + r][k] x[k][j] is
    // (R / 16) x (L / 16) = 16 tiles of 16 x 16, one per wave; v_mfma_f32_16x16x4_f32 takes
    // A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15].  (Read through LDS broadcasts the H
    // tile costs 64 lanes x 16 B of LDS return bandwidth per k and row group: measured 5.3K cycles.)
    {
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const int wave = tid >> 6, lane = tid & 63, nwv = nt >> 6;
      const int ctiles = L >> 4;
      const int lr = lane & 15, lq = lane >> 4;
      for (int tile = wave; tile < 16; tile += nwv) {      // 16 waves: one tile each; 8 waves (batches): two
        const int rt = tile / ctiles, ct = tile - rt * ctiles;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* hp = hs + rt * 16 + lr;                 // + k * R
        const float* xp = xs + ct * 16 + lr;                 // + k * xstride
        for (int k0 = 0; k0 < K; k0 += 4) {
          const int k = k0 + lq;
          const float av = hp[k * R];                        // rows of H past K are zero filled below
          const float bv = xp[min(k, K - 1) * xstride];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
        }
        // D[row = 4 * (l >> 4) + i][col = l & 15] -> [row][col] image of the tile
#pragma unroll
        for (int i = 0; i < 4; ++i)
          buf[pad(((rt * 16 + 4 * lq + i) << logL) + ct * 16 + lr)] = acc[i];
      }
    }
    HSTAMP(3);
    __syncthreads();
    if (act) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = buf[pad(e0 + r)];
    }
    __syncthreads();
  }
  float scale = a.scale;
  // planes of a K == 1 transform: the RMSNorm statistic is only needed with the planes' bound, after the
  // transform -- both workgroup reductions then share one pass (see the epilogue)
  const bool defer_rms = PLANES && !TALL && (KONE || K == 1) && a.rms_w != nullptr;
  if (a.rms_w && !defer_rms) {
    // every workgroup sees the whole input row (K == 1: it is the row; K > 1: the k loop)
    const float tot = block_reduce(ss_x, false, red, tid, nt, false);
    scale = had::rms_scale(a.scale, tot, a.in_features, a.rms_eps);
  }

  HSTAMP(4);
  // (2) length-L transform: 4 index bits per pass in registers, LDS re-shuffle in between
  had::fht16(v, buf, tt, logL, act, a.pp);
  HSTAMP(5);
  const bool live = act && kp < K;                // rows past K in the last tall workgroup

  // (3) epilogue
  if constexpr (PLANES) {
    float bound;
    if (defer_rms) {
      // max_r |fl(v_r * scale)| == fl(max_r |v_r| * |scale|) (rounding is monotonic and sign symmetric): reduce the
      // unscaled maximum together with the sum of squares, then scale -- the same bits as two separate passes
      float tot = ss_x, mx = act ? had::absmax16(v, 1.f) : 0.f;
      had::block_reduce_sum_max(tot, mx, red, tid, nt, nta);
      scale = had::rms_scale(a.scale, tot, a.in_features, a.rms_eps);
      bound = had::fmul(mx, fabsf(scale));
    } else if (K == 1) {
      bound = block_reduce(act ? had::absmax16(v, scale) : 0.f, true, red + 16, tid, nta, false);
    } else {
      bound = sqrtf(block_reduce(ss_in, false, red + 16, tid, nt, false)) * sqrtf((float)L) * fabsf(scale) * 1.0625f;
    }
    if (a.hi_layout) {
      // HI (4-bit scalar, w = nibble - 7.5, hi.py:41-63): a code byte holds the nibbles of columns
      // (0,2) (4,6) (1,3) (5,7) of its 8-group.  Read as D4 codes (one byte = 4 weights) with the table
      // entry [lo - 7.5, hi - 7.5, 0, 0], the row is a D4 matrix with 2n columns, and its product with
      // x' = [x0 x2 0 0 | x4 x6 0 0 | x1 x3 0 0 | x5 x7 0 0]_g is the HI product: the D4 mode of the
      // matrix-core GEMV runs unchanged on the planes of x' (length 2n), written here.
      const int sh = had::shift_for(bound);
      if (blockIdx.x == 0 && tid == 0) *reinterpret_cast<int*>(a.planes + (size_t)3 * a.Kp) = sh;
      float vp[2][16];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const float* e = v + 8 * g;
        const float o16[16] = {e[0], e[2], 0.f, 0.f, e[4], e[6], 0.f, 0.f, e[1], e[3], 0.f, 0.f, e[5], e[7], 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 16; ++i) vp[g][i] = o16[i];
      }
      uint4 d0[3], d1[3];
      had::planes16(vp[0], scale, sh, d0);
      had::planes16(vp[1], scale, sh, d1);
      const int idx = 2 * (kp * L + j0);
      if (live) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          uint4* dst = reinterpret_cast<uint4*>(a.planes + (size_t)d * a.Kp + idx);
          dst[0] = d0[d];
          dst[1] = d1[d];
        }
      }
      if (blockIdx.x == 0)  // zero the k padding [2n, Kp)
        for (int i = 2 * a.n + tid * 16; i < a.Kp; i += nt * 16)
#pragma unroll
          for (int d = 0; d < 3; ++d) *reinterpret_cast<uint4*>(a.planes + (size_t)d * a.Kp + i) = make_uint4(0, 0, 0, 0);
    } else if (a.rvq_scale != 0.f) {
      // E8P12RVQ4B: a code is (main16 << 16 | resid16) and w = E8P(main) + s * E8P(resid)
      // (e8p12_rvq4.py:37-67).  Read as 16-bit E8P codes, the row is a matrix with 2n columns whose
      // 8-groups alternate resid, main; its product with x' = [s * x_g | x_g]_g is the RVQ4 product.
      // So the E8P12 GEMV runs unchanged on the planes of x' (length 2n), written here.
      const float rs = a.rvq_scale;
      const int sh = had::shift_for(bound * fmaxf(1.f, fabsf(rs)));
      if (blockIdx.x == 0 && tid == 0) *reinterpret_cast<int*>(a.planes + (size_t)3 * a.Kp) = sh;
      uint4 dm[3], dr[3];
      had::planes16(v, scale, sh, dm);
      had::planes16(v, had::fmul(scale, rs), sh, dr);
      const int idx = 2 * (kp * L + j0);
      if (live) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          uint4* dst = reinterpret_cast<uint4*>(a.planes + (size_t)d * a.Kp + idx);
          dst[0] = make_uint4(dr[d].x, dr[d].y, dm[d].x, dm[d].y);
          dst[1] = make_uint4(dr[d].z, dr[d].w, dm[d].z, dm[d].w);
        }
      }
      if (blockIdx.x == 0)  // zero the k padding [2n, Kp)
        for (int i = 2 * a.n + tid * 16; i < a.Kp; i += nt * 16)
#pragma unroll
          for (int d = 0; d < 3; ++d) *reinterpret_cast<uint4*>(a.planes + (size_t)d * a.Kp + i) = make_uint4(0, 0, 0, 0);
    } else {
    const int sh = had::shift_for(bound);
    if (blockIdx.x == 0 && tid == 0) *reinterpret_cast<int*>(a.planes + (size_t)3 * a.Kp) = sh;
    uint4 dg[3];
    had::planes16(v, scale, sh, dg);
    const int idx = kp * L + j0;
    if (live) {
#pragma unroll
      for (int d = 0; d < 3; ++d) *reinterpret_cast<uint4*>(a.planes + (size_t)d * a.Kp + idx) = dg[d];
    }
    if (blockIdx.x == 0)  // zero the k padding [n, Kp)
      for (int i = a.n + tid * 16; i < a.Kp; i += nt * 16)
#pragma unroll
        for (int d = 0; d < 3; ++d) *reinterpret_cast<uint4*>(a.planes + (size_t)d * a.Kp + i) = make_uint4(0, 0, 0, 0);
    }
  } else {
    f16* yr = a.y + row * a.out_features;
    const f16* rr = a.residual ? a.residual + row * a.out_features : nullptr;
    const int idx0 = kp * L + j0;
    if (!live) {
    } else if (idx0 + 16 <= a.out_features && a.vec_out) {
      // out_elem() two elements at a time, the optional vectors behind uniform branches: requested together, applied
      // in out_elem()'s order (scale, post, bias, residual; one rounding each, then fp16)
      uint4 qp[2], qb[2], qr[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (a.post) qp[h] = ldp(a.post + idx0 + 8 * h);
        if (a.bias) qb[h] = ldp(a.bias + idx0 + 8 * h);
        if (rr) qr[h] = ldp(rr + idx0 + 8 * h);
      }
      const f32x2 sc = {scale, scale};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
#pragma clang fp contract(off)
        f32x2 o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f32x2{v[8 * h + 2 * r], v[8 * h + 2 * r + 1]} * sc;
        if (a.post) mul8p(o, qp[h]);
        if (a.bias) add8p(o, qb[h]);
        if (rr) add8p(o, qr[h]);
        reinterpret_cast<uint4*>(yr + idx0)[h] = had::pack8p(o);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = idx0 + r;
        if (idx < a.out_features)
          yr[idx] = had::out_elem(v[r], scale, a.post != nullptr, a.post ? (float)a.post[idx] : 0.f,
                                  a.bias != nullptr, a.bias ? (float)a.bias[idx] : 0.f, rr != nullptr,
                                  rr ? (float)rr[idx] : 0.f);
      }
    }
  }
  HSTAMP(6);
}

// Tall transform for BATCHES (prefill: thousands of token rows), fp16 output.  had_fast_kernel<*, TALL> is shaped
// for latency: three workgroups per row, each staging the whole row.  Here ONE 256-thread workgroup owns a whole
// row at a time and walks over many rows (grid.x < rows), two workgroups per CU:
//   (1) the pre-processed row goes to LDS in the padded [k][j] layout.  Its raw 16-byte pieces were requested
//       while the PREVIOUS row was in phases 2 and 3 (registers), so no memory latency is exposed here;
//   (2) wave w runs the K-mix of column tiles w, w + 4, ... for all (<= 3) row tiles on the matrix cores (one B
//       read feeds three MFMAs, H sits in LDS as fp16) and writes the result over its own input columns;
//   (3) the length-L transform, 4096 elements at a time: a thread holds 16 consecutive columns, so index bits
//       0..3 are register butterflies and bits 4..LOGL-1 are butterflies between the L / 16 <= 16 lanes of one
//       DPP row (had::fht16_lanes: no LDS traffic, no barrier), then the packed epilogue and 16-byte stores.
// The per-tile MFMA sequence, the butterfly order (index bits 0, 1, ... LOGL-1; x0 + x1 and x0 - x1 with x0 the
// element whose bit is clear) and the element-wise operations are those of had_fast_kernel, so the results are bit
// identical to it.  LDS: roundup4(K) (L + L / 32) floats + H = 51 KB at 43 x 256; ~230 VGPRs (the prefetched row and
// one side's vectors stay in registers).
// Requires 64 <= L <= 256, K <= 48 (L = 64: K <= 176), n <= 12288, vector access, no RMSNorm statistic, vectors of one side only (host-checked).
// LDS traffic of this workgroup is complete and visible, nothing else is waited for (the row prefetch stays in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// "these loads have landed" as far as the compiler's wait bookkeeping goes: it places its own s_waitcnt before this
// point and none later.  gfx9 counts loads and stores in ONE counter, so a wait for a load that is placed after
// stores is a wait for those stores as well (a few microseconds per row here) -- the row loop therefore waits for
// everything it has requested BEFORE its first store, at a point where the requests are a whole phase old.
__device__ __forceinline__ void landed(const u32x4 (&q)[3][2]) {
  asm volatile("" ::"v"(q[0][0]), "v"(q[0][1]), "v"(q[1][0]), "v"(q[1][1]), "v"(q[2][0]), "v"(q[2][1]));
}
// row-invariant vectors stay PACKED in their registers: without this the compiler converts them to fp32 once,
// outside the row loop, and keeps twice the registers
__device__ __forceinline__ void keep_packed(u32x4 (&q)[3][2]) {
  asm volatile("" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[2][0]), "+v"(q[2][1]));
}
// SIDE 0: input side (element-wise work before the transform: gate, pre; fp16 = scale * transform after it);
// SIDE 1: output side (post, bias, residual after it).  One or the other per launch (host-checked): each side keeps
// its vectors in registers, both together do not fit two workgroups per CU.
template <int LOGL, int SIDE, int RT, bool PAIR>
__global__ __launch_bounds__(PAIR ? 512 : 256, PAIR ? 1 : 2) void had_tall_batch_kernel(HadGroup grp, int rows) {
#pragma clang fp contract(off)
  const HadArgs a = grp.p[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) float buf[];
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int nt = 256, L = 1 << LOGL, nw = 4;
  constexpr int ctiles = L >> 4, tpw = ctiles / nw;   // column tiles per wave: 4 (L = 256), 2 (128), 1 (64)
  constexpr int rr = (nt * 16) >> LOGL;             // rows per transform round (4096 elements)
  constexpr int round_floats = nt * 16 + ((nt * 16) >> 5);
  // PAIR: two groups of 256 threads in one workgroup, each with its own row buffer, sharing H (K > 48: two workgroups
  // with their own 62 KB of H do not fit a CU); group 1 runs one phase behind group 0 (see the row loop)
  constexpr int G = PAIR ? 2 : 1;
  const int tid = threadIdx.x & 255, half = PAIR ? threadIdx.x >> 8 : 0, K = a.K;
  const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int ksteps = (K + 3) >> 2;
  const int BR = (K + 3) & ~3;                     // buffer rows: inputs k < K, outputs kp < BR (kp >= K are zero)
  // H as the MFMA's A operand, A[row kq][k] at hs[kq * KP + k] (fp16 as stored; a lane's four k of a step are one 8-byte
  // read), zero outside (K, K).  RT row tiles of
  // 16: 3 (K <= 48) or 11 (K <= 176: 11008 = 172 x 64 with the table factors of get_hadK(use_rand=False); 62 KB of H
  // next to the 45 KB row: one workgroup per CU)
  constexpr int KP = RT * 16;
  const int RowF = (had::buf_floats(BR * L) + 3) & ~3;
  float* const gbuf = buf + half * RowF;
  f16* hs = reinterpret_cast<f16*>(buf + G * RowF);
  for (int i = threadIdx.x; i < KP * KP; i += G * nt) {
    const int kq = i / KP, k = i - kq * KP;
    hs[i] = (kq < K && k < K) ? (a.transpose ? a.had[k * K + kq] : a.had[kq * K + k]) : (f16)0.f;
  }
  const int j0 = (tid * 16) & (L - 1);
  // LDS addresses as `lane base + compile-time offset` (pad(a + b) = pad(a) + pad(b) when a is a multiple of 32 --
  // written out, because the compiler keeps every pad(...) of the loops below in a register of its own otherwise)
  constexpr int RS = L + (L >> 5);                       // floats per buffer row
  float* const stage = gbuf + tid * 16 + (tid >> 1);      // pad(16 tid); + round_floats per 4096 elements
  const int cb = wave * 16 + lr;                         // column of this lane in its first tile; tile t: + 64 t
  float* const mixcol = gbuf + cb + (cb >> 5);            // pad(cb); tile t: + 66 t, row k: + k RS
  // this thread's pieces of a row, input side: 16-element chunks tid, tid + 256, tid + 512 (n <= 48 x 256 =
  // 3 x 4096) at 32-bit byte offsets (requests are `uniform row base + lane offset`); output side: the 16 columns
  // from j0 of row rd rr + (16 tid >> LOGL), round rd = 0, 1, 2
  u32x4 xq[3][2], gq[3][2], pq[3][2];                    // x (next row), gate (next row), pre (all rows)
  u32x4 qpost[3][2], qbias[3][2], qres[3][2];            // post, bias (all rows), residual (this row)
  uint32_t off[3][2];
  const bool padded = a.in_features < a.n;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c = (tid + i * nt) * 16 + 8 * h;
      off[i][h] = (uint32_t)(c < a.in_features ? c : 0) * 2u;
      if (SIDE == 0 && a.pre && (tid + i * nt) * 16 < a.n)
        pq[i][h] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(a.pre) + off[i][h]);
    }
  bool whole[3];
#pragma unroll
  for (int rd = 0; rd < 3; ++rd) {
    const int idx0 = (rd * rr + ((tid * 16) >> LOGL)) * L + j0;
    whole[rd] = rd * rr + ((tid * 16) >> LOGL) < K && idx0 + 16 <= a.out_features;
    if (SIDE == 1 && whole[rd]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (a.post) qpost[rd][h] = *reinterpret_cast<const u32x4*>(a.post + idx0 + 8 * h);
        if (a.bias) qbias[rd][h] = *reinterpret_cast<const u32x4*>(a.bias + idx0 + 8 * h);
      }
    }
  }
  auto fetch = [&](int row) {
    const char* xr = reinterpret_cast<const char*>(a.x + (int64_t)row * a.in_features);
    const char* gr = reinterpret_cast<const char*>(a.gate ? a.gate + (int64_t)row * a.in_features : nullptr);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if ((tid + i * nt) * 16 < a.n) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          xq[i][h] = *reinterpret_cast<const u32x4*>(xr + off[i][h]);
          if (SIDE == 0 && a.gate) gq[i][h] = *reinterpret_cast<const u32x4*>(gr + off[i][h]);
        }
      }
    }
  };
  // rows vw, vw + stride, ... of this group; both groups run the same number of iterations (barriers are workgroup wide)
  const int vw = blockIdx.x * G + half, stride = G * gridDim.x, iters = (rows + stride - 1) / stride;
  if (vw < rows) fetch(vw);
  // the loop starts with nothing pending (see landed())
  landed(xq);
  if (SIDE == 0) { landed(gq); landed(pq); } else { landed(qpost); landed(qbias); }
#ifdef QUIP_HAD_STAMPS
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tl = __builtin_amdgcn_s_memtime();
#define TSTAMP(i) do { const unsigned long long now = __builtin_amdgcn_s_memtime(); tacc[i] += now - tl; tl = now; } while (0)
#else
#define TSTAMP(i) do {} while (0)
#endif
  // Phases per row: (1) stage, (2) K-mix on the matrix cores, (3) transform + epilogue, a barrier after each.  In a
  // PAIR group 1 enters one barrier late, so its phase p runs beside group 0's phase p + 1.  (That the K-mix of one
  // group -- matrix cores, hardly any VALU -- would hide behind the butterflies of the other did not happen: at K = 43
  // the paired form is as fast as two independent workgroups per CU, 0.59 / 0.67 ms; the counters of either show the
  // matrix cores busy 42 %, VALU 36 %, LDS 28 % of the time, adding up instead of overlapping.)
  if (PAIR && half) lds_barrier();
  for (int it = 0; it < iters; ++it) {
    const int row = vw + it * stride;
    const bool valid = row < rows;
    f16* yr = a.y + (int64_t)row * a.out_features;
    const f16* rr_ = a.residual ? a.residual + (int64_t)row * a.out_features : nullptr;
    if (SIDE == 1 && rr_ && valid) {
#pragma unroll
      for (int rd = 0; rd < 3; ++rd)
        if (whole[rd]) {
          const int idx0 = (rd * rr + ((tid * 16) >> LOGL)) * L + j0;
          const char* rb = reinterpret_cast<const char*>(rr_);   // uniform row base + 32-bit lane offset
          qres[rd][0] = *reinterpret_cast<const u32x4*>(rb + (uint32_t)idx0 * 2u);
          qres[rd][1] = *reinterpret_cast<const u32x4*>(rb + (uint32_t)idx0 * 2u + 16u);
        }
    }
    // (1) the pre-processed row, [k][j] in the padded layout
    if (SIDE == 0) keep_packed(pq);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx0 = (tid + i * nt) * 16;
      if (idx0 < a.n && valid) {
        float* dst = stage + i * round_floats;   // pad(idx0): 16 elements inside one 32-block, constant pad offset
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          f32x2 o[4];
          unpack8p(xq[i][h], o);
          if (padded) {                          // F.pad zeros
            const float keep = (idx0 + 8 * h) < a.in_features ? 1.f : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = o[r] * f32x2{keep, keep};
          }
          if (SIDE == 0 && a.gate) {
            f32x2 t[4];
            unpack8p(gq[i][h], t);
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = o[r] * f32x2{silu(t[r].x), silu(t[r].y)};
          }
          if (SIDE == 0 && a.pre) mul8p(o, pq[i][h]);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            dst[8 * h + 2 * r] = o[r].x;
            dst[8 * h + 2 * r + 1] = o[r].y;
          }
          __builtin_amdgcn_sched_barrier(0);   // one piece at a time: interleaved, the six pieces do not fit the registers
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // (hoisted above the arithmetic, the requests would need a second set of registers)
    if (row + stride < rows) fetch(row + stride);   // lands during phase 2
    TSTAMP(0);
    lds_barrier();
    TSTAMP(1);
    // (2) K-mix on the matrix cores, in place per column tile.  A wave runs its (<= 4) column tiles TOGETHER: per
    //     k step one A read per row tile feeds all of them, and the 3 x tiles MFMAs of a step (>= 96 cycles of
    //     matrix-core time) cover the LDS latency of the next step's operands.
    if (valid) {
      // v_mfma_f32_16x16x16_f16, k steps of 16: A = four consecutive k of a row of H (fp16 as stored), B = four rows of
      // the staged columns as fp16.  Output side: the staged values ARE fp16 numbers (raw products, times 0 / 1), one MFMA
      // per step; input side (x * pre, silu(gate) * x: fp32): hi + lo = the value to 22 bits, two MFMAs.  Products of two
      // fp16 numbers are exact in the fp32 accumulator either way.  (The fp32 MFMA this replaces, 16x16x4, ran at the
      // vector rate: 132 instructions of 32 cycles per row and wave at K = 43, against 36 / 72 of 16 here.  Results differ
      // from the single-row kernel's fp32 K-mix in the order of the additions: tests/test_gpu_ops.py states the bound.)
      typedef _Float16 f16x4v __attribute__((ext_vector_type(4)));
      f32x4 acc[tpw][RT];
#pragma unroll
      for (int t = 0; t < tpw; ++t)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[t][rt] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int ksteps16 = RT;                           // KP / 16
      f16x4v ah[RT];
      float bv[tpw][4];
      auto operands = [&](int ks, f16x4v (&ah_)[RT], float (&bv_)[tpw][4]) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          ah_[rt] = *reinterpret_cast<const f16x4v*>(hs + (rt * 16 + lr) * KP + 16 * ks + 4 * lq);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int k = min(16 * ks + 4 * lq + i, K - 1);   // (rows past K: any finite value, their column of H is zero)
#pragma unroll
          for (int t = 0; t < tpw; ++t) bv_[t][i] = mixcol[k * RS + 66 * t];
        }
      };
      auto step = [&](const f16x4v (&ah_)[RT], const float (&bv_)[tpw][4]) {
#pragma unroll
        for (int t = 0; t < tpw; ++t) {
          const auto h01 = __builtin_amdgcn_cvt_pkrtz(bv_[t][0], bv_[t][1]), h23 = __builtin_amdgcn_cvt_pkrtz(bv_[t][2], bv_[t][3]);
          const f16x4v bh = __builtin_bit_cast(f16x4v, uint2{__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23)});
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah_[rt], bh, acc[t][rt], 0, 0, 0);
          if (SIDE == 0) {
            const auto l01 = __builtin_amdgcn_cvt_pkrtz(bv_[t][0] - (float)h01[0], bv_[t][1] - (float)h01[1]);
            const auto l23 = __builtin_amdgcn_cvt_pkrtz(bv_[t][2] - (float)h23[0], bv_[t][3] - (float)h23[1]);
            const f16x4v bl = __builtin_bit_cast(f16x4v, uint2{__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23)});
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[t][rt] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah_[rt], bl, acc[t][rt], 0, 0, 0);
          }
        }
      };
      if constexpr (RT <= 3) {
        // three steps, straight line: the scheduler overlaps a step's reads with the previous step's MFMAs, and the other
        // workgroup of the CU covers the rest (operands a step ahead do not fit the registers at L = 256)
#pragma unroll
        for (int ks = 0; ks < ksteps16; ++ks) {
          operands(ks, ah, bv);
          step(ah, bv);
        }
      } else {
        operands(0, ah, bv);
#pragma unroll 1
        for (int ks = 0; ks < ksteps16; ++ks) {
          f16x4v an[RT];
          float bn[tpw][4];
          operands(min(ks + 1, ksteps16 - 1), an, bn);
          __builtin_amdgcn_sched_barrier(0);
          step(ah, bv);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) ah[rt] = an[rt];
#pragma unroll
          for (int t = 0; t < tpw; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) bv[t][i] = bn[t][i];
        }
      }
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        float* const orow4 = mixcol + 4 * lq * RS;             // output rows rt 16 + 4 lq + i
        if (rt * 16 + 4 * lq < BR) {                           // BR is a multiple of 4: all four rows or none
#pragma unroll
          for (int t = 0; t < tpw; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) orow4[(rt * 16 + i) * RS + 66 * t] = acc[t][rt][i];
        }
      }
    }
    TSTAMP(2);
    lds_barrier();
    TSTAMP(3);
    // everything requested so far has had phase 2 to arrive; no load is waited for after the first store below
    landed(xq);
    if (SIDE == 0) landed(gq); else { landed(qres); keep_packed(qpost); keep_packed(qbias); }
    // (3) + (4): length-L transform of the K rows and the epilogue, 4096 elements at a time
#pragma unroll
    for (int rd = 0; rd < 3; ++rd) {
      if (rd * rr < K && valid) {
        const int kp = rd * rr + ((tid * 16) >> LOGL);
        const bool live = kp < K;
        const float* hb = live ? stage + rd * round_floats : stage;
        const int idx0 = kp * L + j0;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = hb[r];   // (rows >= K of the last round: never stored)
        had::fht16_lanes<LOGL>(v, lane);
        if (whole[rd]) {
          const f32x2 sc = {a.scale, a.scale};
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            f32x2 o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = f32x2{v[8 * h + 2 * r], v[8 * h + 2 * r + 1]} * sc;
            if (SIDE == 1) {
              if (a.post) mul8p(o, qpost[rd][h]);
              if (a.bias) add8p(o, qbias[rd][h]);
              if (rr_) add8p(o, qres[rd][h]);
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<char*>(yr) + ((uint32_t)idx0 * 2u + 16u * h)) = had::pack8p(o);
          }
        } else if (live) {   // the ragged end of out_features
#pragma unroll 1
          for (int r = 0; r < 16; ++r) {
            const int idx = idx0 + r;
            if (idx < a.out_features)
              yr[idx] = had::out_elem(v[r], a.scale, a.post != nullptr, a.post ? (float)a.post[idx] : 0.f,
                                      a.bias != nullptr, a.bias ? (float)a.bias[idx] : 0.f, rr_ != nullptr,
                                      rr_ ? (float)rr_[idx] : 0.f);
          }
        }
      }
    }
    TSTAMP(4);
    lds_barrier();   // the buffer is restaged for the next row
    TSTAMP(5);
  }
  if (PAIR && !half) lds_barrier();   // (group 1's last phase)
#ifdef QUIP_HAD_STAMPS
  if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0)
    for (int i = 0; i < 6; ++i) g_had_stamps[8 + i] = tacc[i];
#endif
}

// K == 1 transform for BATCHES (prefill), fp16 output, L = 2^LOGL in {1024, 2048, 4096, 8192}: the same arithmetic as
// had_fast_kernel<false, false, 256, true> without its decode-only paths (chain, planes, RMSNorm statistic), so
// that it fits 64 VGPRs: eight row-workgroups resident per CU instead of four -- a batch of rows is bound by the
// bytes in flight, not by the latency of one row.
template <int LOGL>
__global__ __launch_bounds__((LOGL > 12 ? 512 : 256), (LOGL > 12 ? 4 : 8)) void had_kone_batch_kernel(HadGroup grp) {
  const HadArgs a = grp.p[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) float buf[];
  constexpr int L = 1 << LOGL;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.y;
  const f16* xr = a.x + row * a.in_features;
  const f16* gr = a.gate ? a.gate + row * a.in_features : nullptr;
  const int j0 = tid * 16;
  float v[16];
  float ss = 0.f;
  in_vals16(a, xr, gr, j0, v, ss);
  had::fht16_fixed<LOGL, false>(v, buf, 0, tid, true);
  f16* yr = a.y + row * a.out_features;
  const f16* rr = a.residual ? a.residual + row * a.out_features : nullptr;
  if (j0 + 16 <= a.out_features) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {   // out_elem() two elements at a time, optional vectors behind uniform branches
#pragma clang fp contract(off)
      uint4 qp, qb, qr;
      if (a.post) qp = ldp(a.post + j0 + 8 * h);
      if (a.bias) qb = ldp(a.bias + j0 + 8 * h);
      if (rr) qr = ldp(rr + j0 + 8 * h);
      const f32x2 sc = {a.scale, a.scale};
      f32x2 o[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = f32x2{v[8 * h + 2 * r], v[8 * h + 2 * r + 1]} * sc;
      if (a.post) mul8p(o, qp);
      if (a.bias) add8p(o, qb);
      if (rr) add8p(o, qr);
      reinterpret_cast<uint4*>(yr + j0)[h] = had::pack8p(o);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int idx = j0 + r;
      if (idx < a.out_features)
        yr[idx] = had::out_elem(v[r], a.scale, a.post != nullptr, a.post ? (float)a.post[idx] : 0.f,
                                a.bias != nullptr, a.bias ? (float)a.bias[idx] : 0.f, rr != nullptr,
                                rr ? (float)rr[idx] : 0.f);
    }
  }
  (void)L;
}

// Small odd K (3, 5, 7: 5120 = 5 x 1024, 14336 = 7 x 2048, 28672 = 7 x 4096) times a long power of two, for BATCHES
// (prefill), fp16 output.  The row-parallel launch gives every (kp, row) its own workgroup, which reads all K input
// sub-rows: K times the input traffic (28672-wide rows ran at 0.8-1.0 TB/s r+w).  Here one workgroup of L / 16 threads
// owns a token row: a thread reads its 16 columns of the K sub-rows ONCE (K x 16 fp32 in registers), and for
// kp = 0 .. K-1 forms the K-mix (the single-row launch's partial fma chains and their order), runs the length-L
// transform and stores -- the same functions, so the same bits as the row alone.
template <int LOGL, int K>
__global__ __launch_bounds__(1 << (LOGL - 4)) void had_wide_batch_kernel(HadGroup grp) {
  const HadArgs a = grp.p[blockIdx.z];
  extern __shared__ __attribute__((aligned(16))) float buf[];
  constexpr int L = 1 << LOGL;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.y;
  const f16* xr = a.x + row * a.in_features;
  const f16* gr = a.gate ? a.gate + row * a.in_features : nullptr;
  const int j0 = tid * 16;
  float e[K][16];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) in_vals16(a, xr, gr, k * L + j0, e[k], ss);
  f16* yr = a.y + row * a.out_features;
  const f16* rr = a.residual ? a.residual + row * a.out_features : nullptr;
#pragma unroll
  for (int kp = 0; kp < K; ++kp) {
    // the single-row launch splits the k loop over TG = min(K, 4) thread groups (k = g, g + TG, ...: one fma chain
    // each) and adds the partial sums in the order ((p0 + p1) + p2) + p3: the same here, so that a row of a batch and
    // the row alone agree bit for bit
    constexpr int TG = K < 4 ? K : 4;
    float v[16];
#pragma unroll
    for (int g = 0; g < TG; ++g) {
      float p[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) p[r] = 0.f;
#pragma unroll
      for (int k = g; k < K; k += TG) {
        const float h = (float)(a.transpose ? a.had[k * K + kp] : a.had[kp * K + k]);
#pragma unroll
        for (int r = 0; r < 16; ++r) p[r] = __builtin_fmaf(h, e[k][r], p[r]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = g == 0 ? p[r] : had::fadd(v[r], p[r]);
    }
    had::fht16_fixed<LOGL, false>(v, buf, 0, tid, true);
    const int idx0 = kp * L + j0;
    if (idx0 + 16 <= a.out_features) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {   // out_elem() two elements at a time, optional vectors behind uniform branches
#pragma clang fp contract(off)
        uint4 qp, qb, qr;
        if (a.post) qp = ldp(a.post + idx0 + 8 * h);
        if (a.bias) qb = ldp(a.bias + idx0 + 8 * h);
        if (rr) qr = ldp(rr + idx0 + 8 * h);
        const f32x2 sc = {a.scale, a.scale};
        f32x2 o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = f32x2{v[8 * h + 2 * r], v[8 * h + 2 * r + 1]} * sc;
        if (a.post) mul8p(o, qp);
        if (a.bias) add8p(o, qb);
        if (rr) add8p(o, qr);
        reinterpret_cast<uint4*>(yr + idx0)[h] = had::pack8p(o);
      }
    } else {
#pragma unroll 1
      for (int r = 0; r < 16; ++r) {
        const int idx = idx0 + r;
        if (idx < a.out_features)
          yr[idx] = had::out_elem(v[r], a.scale, a.post != nullptr, a.post ? (float)a.post[idx] : 0.f,
                                  a.bias != nullptr, a.bias ? (float)a.bias[idx] : 0.f, rr != nullptr,
                                  rr ? (float)rr[idx] : 0.f);
      }
    }
  }
}

// simple LDS radix-2 version for lengths the blocked kernel does not take
template <bool PLANES>
__global__ __launch_bounds__(256) void had_small_kernel(HadGroup grp) {
  HadArgs a = grp.p[blockIdx.z];
  if constexpr (PLANES) a.planes += (size_t)blockIdx.y * ((size_t)3 * a.Kp + 16);   // one plane image per token row
  extern __shared__ __attribute__((aligned(16))) float buf[];
  __shared__ float red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int kp = blockIdx.x;
  const int64_t row = blockIdx.y;
  const f16* xr = a.x + row * a.in_features;
  const f16* gr = a.gate ? a.gate + row * a.in_features : nullptr;
  const int L = a.L, K = a.K;
  float ss_x = 0.f, ss_in = 0.f;
  for (int j = tid; j < L; j += nt) {
    float acc = 0.f;
    for (int k = 0; k < K; ++k) {
      const float h = (K == 1) ? 1.f : (float)(a.transpose ? a.had[k * K + kp] : a.had[kp * K + k]);
      const int idx = (K == 1 ? kp : k) * L + j;
      const float e = in_val(a, xr, gr, idx);
      if (a.rms_w && idx < a.in_features) { const float xv = (float)xr[idx]; ss_x += xv * xv; }
      ss_in += e * e;
      acc = __builtin_fmaf(h, e, acc);
    }
    buf[j] = acc;
  }
  float scale = a.scale;
  if (a.rms_w) scale = had::rms_scale(a.scale, block_reduce(ss_x, false, red, tid, nt), a.in_features, a.rms_eps);
  __syncthreads();
  for (int h = 1; h < L; h <<= 1) {
    for (int i = tid; i < (L >> 1); i += nt) {
      const int i0 = ((i & ~(h - 1)) << 1) | (i & (h - 1));
      const float x0 = buf[i0], x1 = buf[i0 + h];
      buf[i0] = x0 + x1;
      buf[i0 + h] = x0 - x1;
    }
    __syncthreads();
  }
  if constexpr (PLANES) {
    float bound;
    if (K == 1) {
      float mx = 0.f;
      for (int j = tid; j < L; j += nt) {
        const float m = fabsf(had::fmul(buf[j], scale));
        mx = fmaxf(mx, m == m ? m : __builtin_inff());   // fmaxf would drop a NaN
      }
      bound = block_reduce(mx, true, red, tid, nt);
    } else {
      bound = sqrtf(block_reduce(ss_in, false, red, tid, nt)) * sqrtf((float)L) * fabsf(scale) * 1.0625f;
    }
    const int sh = had::shift_for(bound);
    const float s2 = had::fmul(scale, as_f32((uint32_t)(sh + 127) << 23));
    if (kp == 0 && tid == 0) *reinterpret_cast<int*>(a.planes + (size_t)3 * a.Kp) = sh;
    for (int j4 = tid * 4; j4 < L; j4 += nt * 4) {
      uint32_t dg[3] = {0, 0, 0};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int h, m, l;
        had::digits_of((int)__builtin_rintf(had::fmul(buf[j4 + e], s2)), h, m, l);
        dg[0] |= (uint32_t)(h & 0xff) << (8 * e);
        dg[1] |= (uint32_t)(m & 0xff) << (8 * e);
        dg[2] |= (uint32_t)(l & 0xff) << (8 * e);
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) *reinterpret_cast<uint32_t*>(a.planes + (size_t)d * a.Kp + kp * L + j4) = dg[d];
    }
    if (kp == 0)
      for (int i = a.n + tid * 4; i < a.Kp; i += nt * 4)
#pragma unroll
        for (int d = 0; d < 3; ++d) *reinterpret_cast<uint32_t*>(a.planes + (size_t)d * a.Kp + i) = 0u;
  } else {
    f16* yr = a.y + row * a.out_features;
    const f16* rr = a.residual ? a.residual + row * a.out_features : nullptr;
    for (int j = tid; j < L; j += nt) {
      const int idx = kp * L + j;
      if (idx < a.out_features) {
        yr[idx] = had::out_elem(buf[j], scale, a.post != nullptr, a.post ? (float)a.post[idx] : 0.f,
                                a.bias != nullptr, a.bias ? (float)a.bias[idx] : 0.f, rr != nullptr,
                                rr ? (float)rr[idx] : 0.f);
      }
    }
  }
}

template <typename Kern>
int launch_one(Kern kern, DynLdsCache& configured, const HadGroup& g, dim3 grid, int threads, int lds,
               hipStream_t stream) {
  if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds, stream, g);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// `count` problems of the same (n, K, L) and kind (planes / fp16) in one launch (grid.z)
int launch(HadGroup& g, int count, int64_t rows, hipStream_t stream) {
  const int L = g.p[0].L, K = g.p[0].K, n = g.p[0].n;
  const bool planes = g.p[0].planes != nullptr;
  for (int i = 0; i < count; ++i) {
    HadArgs& a = g.p[i];
    a.vec = (a.in_features % 8 == 0) && aligned16(a.x) && aligned16(a.gate) && aligned16(a.rms_w) &&
            aligned16(a.pre) && aligned16(a.pre2);
    a.vec_out = (a.out_features % 8 == 0) && aligned16(a.y) && aligned16(a.post) && aligned16(a.bias) &&
                aligned16(a.residual);
  }
  static DynLdsCache cfg[8];   // per kernel instantiation, per device
  if (K > 1 && L >= 64 && L <= 256) {   // tall: 256 threads, R = 4096 / L rows per workgroup
    const int R = 4096 / L;
    // [shuffle buffer | H tile | x rows | second half of the ping-pong buffer]
    // decode launches: 1024 threads, ping-pong buffer (latency); batches: 512 threads, single buffer, so
    // that two workgroups share a CU (one at a time made prefill throughput = workgroup latency)
    const bool batch = rows > 8;
    {   // prefill batches: one workgroup per row at a time (had_tall_batch_kernel)
      bool ok = rows >= 32 && (K <= 48 || (K <= 176 && L == 64)) && n <= 12288 && !planes;
      for (int i = 0; i < count; ++i)
        ok = ok && g.p[i].vec && g.p[i].vec_out && !g.p[i].rms_w && !g.p[i].pre2 && !g.p[i].z && g.p[i].out_features % 8 == 0;
      // element-wise vectors of one side only (input: gate, pre; output: post, bias, residual), see the kernel
      bool in_side = false, out_side = false;
      for (int i = 0; i < count; ++i) {
        in_side = in_side || g.p[i].gate || g.p[i].pre;
        out_side = out_side || g.p[i].post || g.p[i].bias || g.p[i].residual;
      }
      ok = ok && !(in_side && out_side);
      if (ok) {
        static DynLdsCache cfgb[8];
        const int BR = (K + 3) & ~3;
        const int KP = K <= 48 ? 48 : 176;
        const bool pair = K > 48;   // two rows in flight in ONE workgroup per CU (they share the 62 KB of H); else two workgroups
        const int lds = (pair ? 2 : 1) * ((had::buf_floats(BR * L) + 3) & ~3) * 4 + KP * KP * 2;
        const int threads = pair ? 512 : 256;
        const int64_t want = (pair ? 1 : 2) * (int64_t)device_cu_count(), units = pair ? (rows + 1) / 2 : rows;
        const dim3 grid((unsigned)(units < want ? units : want), 1, count);
        auto go = [&](auto kern, DynLdsCache& cache) {
          if (ensure_dyn_lds(cache, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return (int)QUIP_ERR_LAUNCH;
          hipLaunchKernelGGL(kern, grid, dim3(threads), lds, stream, g, (int)rows);
          return hipGetLastError() == hipSuccess ? (int)QUIP_OK : (int)QUIP_ERR_LAUNCH;
        };
        const int logL = g.p[0].logL;
        if (K > 48) return in_side ? go(had_tall_batch_kernel<6, 0, 11, true>, cfgb[6]) : go(had_tall_batch_kernel<6, 1, 11, true>, cfgb[7]);
        if (!in_side) return logL == 8 ? go(had_tall_batch_kernel<8, 1, 3, false>, cfgb[0]) : logL == 7 ? go(had_tall_batch_kernel<7, 1, 3, false>, cfgb[1])
                                                                                                  : go(had_tall_batch_kernel<6, 1, 3, false>, cfgb[2]);
        return logL == 8 ? go(had_tall_batch_kernel<8, 0, 3, false>, cfgb[3]) : logL == 7 ? go(had_tall_batch_kernel<7, 0, 3, false>, cfgb[4])
                                                                                    : go(had_tall_batch_kernel<6, 0, 3, false>, cfgb[5]);
      }
    }
    const int pp = had::buf_floats(4096) + ((K + 3) & ~3) * R + K * (L + 8);
    const int lds = (pp + (batch ? 0 : had::buf_floats(4096))) * 4;
    for (int i = 0; i < count; ++i) g.p[i].pp = batch ? 0 : pp;
    const dim3 grid((K + R - 1) / R, (unsigned)rows, count);
    const int threads = batch ? 512 : 1024;
    return planes ? launch_one(had_fast_kernel<true, true, 1024>, cfg[0], g, grid, threads, lds, stream)
                  : launch_one(had_fast_kernel<false, true, 1024>, cfg[1], g, grid, threads, lds, stream);
  }
  const dim3 grid(K, (unsigned)rows, count);
  bool mixed = false;
  for (int i = 1; i < count; ++i) mixed = mixed || g.p[i].L != L;
  if (mixed) {
    // K == 1 problems of different power-of-two widths (host-checked: fp16 output, no RMSNorm statistic) in one
    // launch: the workgroup has the threads of the widest problem, a narrower problem keeps its first L / 16
    // threads active (thread groups > 0 only take part in the barriers)
    int Lmax = L;
    for (int i = 1; i < count; ++i) Lmax = g.p[i].L > Lmax ? g.p[i].L : Lmax;
    const bool batch = rows > 8;
    for (int i = 0; i < count; ++i) {
      g.p[i].tgroups = Lmax / g.p[i].L;
      g.p[i].part_off = 0;
      g.p[i].pp = batch ? 0 : had::buf_floats(g.p[i].L);
    }
    const int lds = (batch ? 1 : 2) * had::buf_floats(Lmax) * 4;
    static DynLdsCache cm[2];
    return Lmax <= 4096 ? launch_one(had_fast_kernel<false, false, 256, true>, cm[0], g, grid, Lmax / 16, lds, stream)
                        : launch_one(had_fast_kernel<false, false, 1024>, cm[1], g, grid, Lmax / 16, lds, stream);
  }
  if (L >= 256 && L <= 16384) {
    // ping-pong shuffle buffer for the latency-bound decode launches; batches (prefill) take the single
    // buffer so that more rows are resident per CU
    const bool batch = rows > 8;
    const int lds = (batch ? 1 : 2) * had::buf_floats(L) * 4;
    for (int i = 0; i < count; ++i) g.p[i].pp = batch ? 0 : had::buf_floats(L);
    for (int i = 0; i < count; ++i) { g.p[i].tgroups = 1; g.p[i].part_off = 0; }
    if (L <= 4096 && K > 1 && rows <= 8) {
      // latency-bound decode launch of a long row (28672 = 7 x 4096): up to 4 thread groups split the k loop
      int tg = K < 4 ? K : 4;
      while ((L / 16) * tg > 1024) --tg;
      if (tg > 1) {
        const int part = 2 * had::buf_floats(L);
        const int lds2 = (part + (tg - 1) * 16 * (L / 16)) * 4;
        for (int i = 0; i < count; ++i) { g.p[i].tgroups = tg; g.p[i].part_off = part; g.p[i].pp = had::buf_floats(L); }
        static DynLdsCache c2[2];
        return planes ? launch_one(had_fast_kernel<true, false, 1024>, c2[0], g, grid, (L / 16) * tg, lds2, stream)
                      : launch_one(had_fast_kernel<false, false, 1024>, c2[1], g, grid, (L / 16) * tg, lds2, stream);
      }
    }
    if ((K == 3 || K == 5 || K == 7) && !planes && rows > 8 && L >= 512 && L <= 4096) {   // batches, small odd K
      bool ok = true;
      for (int i = 0; i < count; ++i)
        ok = ok && g.p[i].vec && g.p[i].vec_out && !g.p[i].rms_w && !g.p[i].z && g.p[i].out_features % 8 == 0;
      if (ok) {
        const int lds1 = had::buf_floats(L) * 4;
        const dim3 grid1(1, (unsigned)rows, count);
        const int logL = g.p[0].logL;
        auto go = [&](auto kern) {
          hipLaunchKernelGGL(kern, grid1, dim3(L / 16), lds1, stream, g);
          return hipGetLastError() == hipSuccess ? (int)QUIP_OK : (int)QUIP_ERR_LAUNCH;
        };
#define QUIP_WIDE_BATCH(KK)                                                                               \
        return logL == 12 ? go(had_wide_batch_kernel<12, KK>) : logL == 11 ? go(had_wide_batch_kernel<11, KK>) \
             : logL == 10 ? go(had_wide_batch_kernel<10, KK>) : go(had_wide_batch_kernel<9, KK>)
        if (K == 3) { QUIP_WIDE_BATCH(3); }
        if (K == 5) { QUIP_WIDE_BATCH(5); }
        QUIP_WIDE_BATCH(7);
#undef QUIP_WIDE_BATCH
      }
    }
    if (K == 1 && !planes && rows >= 32 && (L == 1024 || L == 2048 || L == 4096 || L == 8192)) {   // prefill batches
      bool ok = true;
      for (int i = 0; i < count; ++i)
        ok = ok && g.p[i].vec && g.p[i].vec_out && !g.p[i].rms_w && !g.p[i].z && g.p[i].out_features % 8 == 0;
      if (ok) {
        const int lds1 = had::buf_floats(L) * 4;
        if (L == 8192) hipLaunchKernelGGL(had_kone_batch_kernel<13>, grid, dim3(L / 16), lds1, stream, g);
        else if (L == 4096) hipLaunchKernelGGL(had_kone_batch_kernel<12>, grid, dim3(L / 16), lds1, stream, g);
        else if (L == 2048) hipLaunchKernelGGL(had_kone_batch_kernel<11>, grid, dim3(L / 16), lds1, stream, g);
        else hipLaunchKernelGGL(had_kone_batch_kernel<10>, grid, dim3(L / 16), lds1, stream, g);
        return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
      }
    }
    if (L <= 4096 && K == 1) {
      static DynLdsCache c1[2];
      return planes ? launch_one(had_fast_kernel<true, false, 256, true>, c1[0], g, grid, L / 16, lds, stream)
                    : launch_one(had_fast_kernel<false, false, 256, true>, c1[1], g, grid, L / 16, lds, stream);
    }
    if (L == 8192 && K == 1) {   // Llama-2-70B's hidden size: the K == 1 instantiation (no K-mix code) on 512 threads
      static DynLdsCache c8[2];
      return planes ? launch_one(had_fast_kernel<true, false, 512, true>, c8[0], g, grid, L / 16, lds, stream)
                    : launch_one(had_fast_kernel<false, false, 512, true>, c8[1], g, grid, L / 16, lds, stream);
    }
    if (L <= 4096)
      return planes ? launch_one(had_fast_kernel<true, false, 256>, cfg[2], g, grid, L / 16, lds, stream)
                    : launch_one(had_fast_kernel<false, false, 256>, cfg[3], g, grid, L / 16, lds, stream);
    return planes ? launch_one(had_fast_kernel<true, false, 1024>, cfg[6], g, grid, L / 16, lds, stream)
                  : launch_one(had_fast_kernel<false, false, 1024>, cfg[7], g, grid, L / 16, lds, stream);
  }
  const int threads = L >= 512 ? 256 : 64;
  return planes ? launch_one(had_small_kernel<true>, cfg[4], g, grid, threads, L * 4, stream)
                : launch_one(had_small_kernel<false>, cfg[5], g, grid, threads, L * 4, stream);
}

int check_shape(int in_features, int out_features, int n, int K, const void* had, int& L, int& logL) {
  if (K < 1 || n % K != 0) return QUIP_ERR_BAD_SHAPE;
  L = n / K;
  if (L < 1 || (L & (L - 1)) != 0 || L > 32768) return QUIP_ERR_BAD_SHAPE;
  if (in_features > n || out_features > n || in_features < 1 || out_features < 1) return QUIP_ERR_BAD_SHAPE;
  if (K > 1 && !had) return QUIP_ERR_NULL_POINTER;
  logL = 0;
  while ((1 << logL) < L) ++logL;
  return QUIP_OK;
}

int fill(HadArgs& a, const HadProblem& pr, bool planes, int n, int K, int transpose) {
  int rc = check_shape(pr.in_features, planes ? n : pr.out_features, n, K, pr.had, a.L, a.logL);
  if (rc != QUIP_OK) return rc;
  if ((!pr.x && !pr.z) || !pr.out) return QUIP_ERR_NULL_POINTER;
  a.x = reinterpret_cast<const f16*>(pr.x ? pr.x : pr.z);
  if (planes) a.planes = reinterpret_cast<uint8_t*>(pr.out); else a.y = reinterpret_cast<f16*>(pr.out);
  a.had = reinterpret_cast<const f16*>(pr.had);
  a.pre = reinterpret_cast<const f16*>(pr.pre);
  a.pre2 = reinterpret_cast<const f16*>(pr.pre2);
  a.post = reinterpret_cast<const f16*>(pr.post);
  a.bias = reinterpret_cast<const f16*>(pr.bias);
  a.residual = reinterpret_cast<const f16*>(pr.residual);
  a.rms_w = reinterpret_cast<const f16*>(pr.rms_weight);
  a.gate = reinterpret_cast<const f16*>(pr.gate);
  a.rms_eps = pr.rms_eps;
  a.z = reinterpret_cast<const f16*>(pr.z);
  a.z_post = reinterpret_cast<const f16*>(pr.z_post);
  a.z_res = reinterpret_cast<const f16*>(pr.z_residual);
  a.h_out = reinterpret_cast<f16*>(pr.h_out);
  a.z_scale = pr.z_scale;
  a.tgroups = 1;
  a.part_off = 0;
  if (pr.z) {   // chain: plain power-of-two width, blocked kernel, vector access
    if (K != 1 || pr.in_features != n || a.L < 256 || a.L > 16384) return QUIP_ERR_UNSUPPORTED;
    if (!pr.z_post || !pr.h_out) return QUIP_ERR_NULL_POINTER;
    if (pr.h_out == pr.z_residual) return QUIP_ERR_BAD_SHAPE;
    const uintptr_t al = reinterpret_cast<uintptr_t>(pr.z) | reinterpret_cast<uintptr_t>(pr.z_post) |
                         reinterpret_cast<uintptr_t>(pr.z_residual) | reinterpret_cast<uintptr_t>(pr.h_out);
    if (al & 15) return QUIP_ERR_MISALIGNED;
  }
  a.in_features = pr.in_features; a.out_features = planes ? n : pr.out_features; a.n = n; a.K = K;
  a.Kp = (n + 511) & ~511;
  a.rvq_scale = planes ? pr.resid_scale : 0.f;
  a.hi_layout = planes && pr.planes_layout == 2;
  if (a.hi_layout) a.rvq_scale = 0.f;
  if (planes && pr.planes_layout != 0 && pr.planes_layout != 1 && pr.planes_layout != 2) return QUIP_ERR_BAD_SHAPE;
  if (a.rvq_scale != 0.f || a.hi_layout) {
    if (!(a.L >= 256 || (K > 1 && a.L >= 64))) return QUIP_ERR_UNSUPPORTED;   // blocked kernels only
    a.Kp = (2 * n + 511) & ~511;
  }
  a.transpose = transpose; a.scale = pr.scale;
  return QUIP_OK;
}

}  // namespace

int had_transform_group_launch(const HadProblem* problems, int count, bool planes, int64_t rows, int n, int K,
                               int transpose, hipStream_t stream) {
  if (count < 1 || count > kMaxGroup || !problems) return QUIP_ERR_BAD_SHAPE;
  HadGroup g{};
  for (int i = 0; i < count; ++i) {
    const int ni = problems[i].n > 0 ? problems[i].n : n;
    if (ni != n && (planes || K != 1 || problems[i].rms_weight || problems[i].z)) return QUIP_ERR_UNSUPPORTED;
    const int rc = fill(g.p[i], problems[i], planes, ni, K, transpose);
    if (rc != QUIP_OK) return rc;
    if (ni != n && (g.p[i].L < 256 || g.p[i].L > 16384)) return QUIP_ERR_UNSUPPORTED;
    if (planes && (reinterpret_cast<uintptr_t>(problems[i].out) & 15) != 0) return QUIP_ERR_MISALIGNED;
  }
  if (planes && (g.p[0].L < 4 || n % 16 != 0)) return QUIP_ERR_BAD_SHAPE;
  if (rows <= 0) return QUIP_OK;
  // grid.y carries the token rows (<= 65535 per launch): longer batches go out in slices
  constexpr int64_t kMaxRows = 65535;
  for (int64_t r0 = 0; r0 < rows; r0 += kMaxRows) {
    const int64_t nr = rows - r0 < kMaxRows ? rows - r0 : kMaxRows;
    HadGroup part = g;
    for (int i = 0; i < count; ++i) {
      HadArgs& a = part.p[i];
      a.x += r0 * a.in_features;
      if (a.gate) a.gate += r0 * a.in_features;
      if (a.y) a.y += r0 * a.out_features;
      if (a.planes) a.planes += r0 * ((int64_t)3 * a.Kp + 16);
      if (a.residual) a.residual += r0 * a.out_features;
      if (a.z) { a.z += r0 * a.n; a.h_out += r0 * a.n; if (a.z_res) a.z_res += r0 * a.n; }
    }
    const int rc = launch(part, count, nr, stream);
    if (rc != QUIP_OK) return rc;
  }
  return QUIP_OK;
}

int had_transform_launch(const void* x, void* y, int64_t rows, int in_features, int out_features,
                         int n, int K, const void* had, int transpose, const void* pre,
                         const void* pre2, const void* post, const void* bias, float scale,
                         hipStream_t stream, const HadFusion* fuse) {
  HadProblem pr;
  pr.x = x; pr.out = y; pr.had = had; pr.pre = pre; pr.pre2 = pre2; pr.post = post; pr.bias = bias;
  pr.in_features = in_features; pr.out_features = out_features; pr.scale = scale;
  if (fuse) { pr.residual = fuse->residual; pr.rms_weight = fuse->rms_weight; pr.gate = fuse->gate; pr.rms_eps = fuse->rms_eps; }
  if (rows <= 0) {   // shape errors still reported for empty batches
    HadArgs a{};
    return check_shape(in_features, out_features, n, K, had, a.L, a.logL);
  }
  return had_transform_group_launch(&pr, 1, false, rows, n, K, transpose, stream);
}

int had_transform_planes_launch(const void* x, void* planes, int in_features, int n, int K,
                                const void* had, int transpose, const void* pre, float scale,
                                hipStream_t stream, const HadFusion* fuse) {
  HadProblem pr;
  pr.x = x; pr.out = planes; pr.had = had; pr.pre = pre;
  pr.in_features = in_features; pr.out_features = n; pr.scale = scale;
  if (fuse) { pr.rms_weight = fuse->rms_weight; pr.gate = fuse->gate; pr.rms_eps = fuse->rms_eps; }
  return had_transform_group_launch(&pr, 1, true, 1, n, K, transpose, stream);
}

}  // namespace quip
