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

__device__ __forceinline__ float silu(float g) {
#pragma clang fp contract(off)
  return g / (1.f + __expf(-g));
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

// Workgroup reduction over the first nt threads (nt a multiple of 64 or < 64); EVERY thread of the
// workgroup must call it (barriers), threads >= nt contribute nothing.  Fixed order: lanes by
// butterfly, then waves 0, 1, 2, ... -> the same value wherever the same data is reduced.
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red, int tid, int nt) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float w = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, w) : fadd(v, w);
  }
  __syncthreads();
  if ((tid & 63) == 0 && tid < nt) red[tid >> 6] = v;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < ((nt + 63) >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : fadd(r, red[w]);
  return r;
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
__device__ __forceinline__ void mul8(float o[8], const uint4& piece) {
  float t[8];
  unpack8(piece, t);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmul(o[i], t[i]);
}
__device__ __forceinline__ void silu_mul8(float o[8], const uint4& gate_piece) {
  float t[8];
  unpack8(gate_piece, t);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = fmul(o[i], silu(t[i]));
}

// RMSNorm factor folded into the transform scale
__device__ __forceinline__ float rms_scale(float scale, float sumsq, int n, float eps) {
#pragma clang fp contract(off)
  const float mean = sumsq / (float)n;
  return fmul(scale, rsqrtf(fadd(mean, eps)));
}

// Length-2^logL Walsh-Hadamard transform of E = 16 * (#active threads) values viewed as rows of
// 2^logL: thread t holds the 16 consecutive elements [16 t, 16 t + 16).  4 butterfly stages per
// pass in registers, re-shuffle through LDS (buf: buf_floats(E) floats) between passes.  Every
// thread of the workgroup must call it; only `active` threads (t < E / 16) touch data.
__device__ __forceinline__ void fht16(float v[16], float* buf, int t, int logL, bool active) {
  const int npass = (logL + 3) >> 2;
  for (int p = 0; p < npass; ++p) {
    const int nb = min(4, logL - 4 * p);
    if (p > 0 && active) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = buf[pad(pass_index(t, r, p, nb))];
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (!(r & (1 << s))) {
            const float x0 = v[r], x1 = v[r | (1 << s)];
            v[r] = fadd(x0, x1);
            v[r | (1 << s)] = fsub(x0, x1);
          }
        }
      }
    }
    if (npass > 1) {
      __syncthreads();  // everyone has read its pass-p inputs
      if (active) {
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[pad(pass_index(t, r, p, nb))] = v[r];
      }
      __syncthreads();
    }
  }
  if (npass > 1 && active) {  // back to 16 consecutive elements per thread
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = buf[pad(t * 16 + r)];
  }
}

// digit split of a block fixed point value (balanced int8 digits, see e8p_gemv_i8.hip)
__device__ __forceinline__ void digits_of(int X, int& h, int& m, int& l) {
  l = (X << 24) >> 24;
  const int X1 = (X - l) >> 8;
  m = (X1 << 24) >> 24;
  h = (X1 - m) >> 8;
}

// shift for |v| <= bound: bound < 2^(E+1) => |rint(v * 2^sh)| < 2^22 with sh = 21 - E
__device__ __forceinline__ int shift_for(float bound) {
  int E = (int)((as_u32(bound) >> 23) & 0xff) - 127;
  E = max(-60, min(60, E));
  return 21 - E;
}

__device__ __forceinline__ float absmax16(const float v[16], float scale) {
  float mx = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(fmul(v[r], scale)));
  return mx;
}

// 16 transformed values -> three 16-byte digit pieces (h, m, l planes) at block exponent sh
__device__ __forceinline__ void planes16(const float v[16], float scale, int sh, uint4 out[3]) {
  const float s2 = fmul(scale, as_f32((uint32_t)(sh + 127) << 23));
  uint32_t dg[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int h, m, l;
    digits_of((int)__builtin_rintf(fmul(v[r], s2)), h, m, l);
    dg[0][r >> 2] |= (uint32_t)(h & 0xff) << (8 * (r & 3));
    dg[1][r >> 2] |= (uint32_t)(m & 0xff) << (8 * (r & 3));
    dg[2][r >> 2] |= (uint32_t)(l & 0xff) << (8 * (r & 3));
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
