// Randomised-Hadamard side of QuantLinear.forward on gfx950.
//
//   y = post (.) ( scale * (H (x) H_L) (pre (.) pre2 (.) x) )[:out_features] + bias
//
// where the padded row (n = K * L, L a power of two) is viewed row-major as
// (K, L), H_L is the Sylvester Walsh-Hadamard matrix applied along L and H is
// the K x K factor `had` (or had^T) applied along K  -- quant.py:72-88
// (matmul_hadU_cuda / matmul_hadUt_cuda) fused with the element-wise ops around
// it in qlinear.py:90-91 (x*SU), :106-107 (per-channel Wscale), :108-114
// (slice, *SV, +bias).  With K == 1 and no vectors this is quip_lib::hadamard
// (register_lib.py:10-20).
//
// (H (x) H_L) = (I (x) H_L)(H (x) I): workgroup (kp, row) first forms row kp of the
// K-mix, t[j] = sum_k H[kp,k] v[k*L + j] (independent per j), then runs the
// length-L transform in LDS.  K workgroups per token row run in parallel, which
// is what makes the 43x43 / 7x7 factors of 11008 / 28672 cheap.
#include "quip_device.hip.h"
#include "quip_internal.h"

namespace quip {

__global__ __launch_bounds__(256) void had_transform_kernel(
    const f16* __restrict__ x, f16* __restrict__ y, int in_features, int out_features, int K, int L,
    const f16* __restrict__ had, int transpose, const f16* __restrict__ pre,
    const f16* __restrict__ pre2, const f16* __restrict__ post, const f16* __restrict__ bias,
    float scale) {
  extern __shared__ __attribute__((aligned(16))) float buf[];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int kp = blockIdx.x;
  const int64_t row = blockIdx.y;
  const f16* xr = x + row * in_features;

  auto in_val = [&](int idx) -> float {
    if (idx >= in_features) return 0.f;  // F.pad (quant.py:73-74)
    float v = (float)xr[idx];
    if (pre) v *= (float)pre[idx];
    if (pre2) v *= (float)pre2[idx];
    return v;
  };

  if (K == 1) {
    for (int j = tid; j < L; j += nt) buf[j] = in_val(j);
  } else {
    for (int j = tid; j < L; j += nt) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) {
        const float h = (float)(transpose ? had[k * K + kp] : had[kp * K + k]);
        acc = __builtin_fmaf(h, in_val(k * L + j), acc);
      }
      buf[j] = acc;
    }
  }
  __syncthreads();
  for (int h = 1; h < L; h <<= 1) {
    for (int i = tid; i < (L >> 1); i += nt) {
      const int i0 = ((i & ~(h - 1)) << 1) | (i & (h - 1));
      const float a = buf[i0], b = buf[i0 + h];
      buf[i0] = a + b;
      buf[i0 + h] = a - b;
    }
    __syncthreads();
  }
  f16* yr = y + row * out_features;
  for (int j = tid; j < L; j += nt) {
    const int idx = kp * L + j;
    if (idx < out_features) {
      float v = buf[j] * scale;
      if (post) v *= (float)post[idx];
      if (bias) v += (float)bias[idx];
      yr[idx] = (f16)v;
    }
  }
}

// Input-side variant for the bs=1 decode path: same transform, but the result is written
// as the block fixed point int8 digit planes the matrix-core GEMV consumes
// (e8p_gemv_mfma.hip: planes[d][Kp], d = 0..2 = h, m, l digits of X = rint(v * 2^sh),
// |X| < 2^22, followed by the int shift word at byte 3 * Kp).  No fp16 rounding happens
// between the transform and the GEMV: typical elements keep >= 14 significant bits
// (fp16 has 11).  The shift comes from a bound every workgroup can compute alone:
//   K == 1: the exact max |v| of the transformed row;
//   K  > 1: |v_i| <= ||v||_2 = scale * sqrt(L) * ||H||_2 * ||pre (.) x||_2, H ~ orthogonal
//           (each of the K workgroups reads the whole input row anyway).
__global__ __launch_bounds__(256) void had_transform_planes_kernel(
    const f16* __restrict__ x, uint8_t* __restrict__ planes, int in_features, int n, int Kp, int K,
    int L, const f16* __restrict__ had, int transpose, const f16* __restrict__ pre, float scale) {
  extern __shared__ __attribute__((aligned(16))) float buf[];
  __shared__ float red[8];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int kp = blockIdx.x;
  auto in_val = [&](int idx) -> float {
    if (idx >= in_features) return 0.f;
    float v = (float)x[idx];
    if (pre) v *= (float)pre[idx];
    return v;
  };
  auto block_reduce = [&](float v, bool is_max) -> float {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const float w = __shfl_xor(v, o, 64);
      v = is_max ? fmaxf(v, w) : v + w;
    }
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int w = 1; w < (nt >> 6); ++w) r = is_max ? fmaxf(r, red[w]) : r + red[w];
    return r;
  };
  float bound;
  if (K == 1) {
    for (int j = tid; j < L; j += nt) buf[j] = in_val(j);
  } else {
    float ss = 0.f;
    for (int idx = tid; idx < in_features; idx += nt) { const float v = in_val(idx); ss += v * v; }
    for (int j = tid; j < L; j += nt) {
      float acc = 0.f;
      for (int k = 0; k < K; ++k) {
        const float h = (float)(transpose ? had[k * K + kp] : had[kp * K + k]);
        acc = __builtin_fmaf(h, in_val(k * L + j), acc);
      }
      buf[j] = acc;
    }
    bound = sqrtf(block_reduce(ss, false)) * sqrtf((float)L) * fabsf(scale) * 1.0625f;
  }
  __syncthreads();
  for (int h = 1; h < L; h <<= 1) {
    for (int i = tid; i < (L >> 1); i += nt) {
      const int i0 = ((i & ~(h - 1)) << 1) | (i & (h - 1));
      const float a = buf[i0], b = buf[i0 + h];
      buf[i0] = a + b;
      buf[i0 + h] = a - b;
    }
    __syncthreads();
  }
  if (K == 1) {
    float mx = 0.f;
    for (int j = tid; j < L; j += nt) mx = fmaxf(mx, fabsf(buf[j] * scale));
    bound = block_reduce(mx, true);
  }
  // |v| <= bound < 2^(E+1)  =>  |rint(v * 2^sh)| < 2^22 with sh = 21 - E  (bound == 0: any shift)
  int E = (int)((as_u32(bound) >> 23) & 0xff) - 127;
  E = max(-60, min(60, E));
  const int sh = 21 - E;
  const float s2 = scale * as_f32((uint32_t)(sh + 127) << 23);
  if (kp == 0 && tid == 0) *reinterpret_cast<int*>(planes + (size_t)3 * Kp) = sh;
  for (int j4 = tid * 4; j4 < L; j4 += nt * 4) {
    uint32_t dg[3] = {0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int X = (int)__builtin_rintf(buf[j4 + e] * s2);
      const int l = (X << 24) >> 24;
      const int X1 = (X - l) >> 8;
      const int m = (X1 << 24) >> 24;
      const int hh = (X1 - m) >> 8;
      dg[0] |= (uint32_t)(hh & 0xff) << (8 * e);
      dg[1] |= (uint32_t)(m & 0xff) << (8 * e);
      dg[2] |= (uint32_t)(l & 0xff) << (8 * e);
    }
    const int idx = kp * L + j4;
#pragma unroll
    for (int d = 0; d < 3; ++d) *reinterpret_cast<uint32_t*>(planes + (size_t)d * Kp + idx) = dg[d];
  }
  if (kp == 0)  // zero the k padding [n, Kp)
    for (int i = n + tid * 4; i < Kp; i += nt * 4)
#pragma unroll
      for (int d = 0; d < 3; ++d) *reinterpret_cast<uint32_t*>(planes + (size_t)d * Kp + i) = 0u;
}

int had_transform_planes_launch(const void* x, void* planes, int in_features, int n, int K,
                                const void* had, int transpose, const void* pre, float scale,
                                hipStream_t stream) {
  if (K < 1 || n % K != 0) return QUIP_ERR_BAD_SHAPE;
  const int L = n / K;
  if (L < 4 || (L & (L - 1)) != 0 || L > 32768 || n % 4 != 0) return QUIP_ERR_BAD_SHAPE;
  if (in_features > n || in_features < 1) return QUIP_ERR_BAD_SHAPE;
  if (K > 1 && !had) return QUIP_ERR_NULL_POINTER;
  const int kp = (n + 511) & ~511;
  const int lds = L * 4;
  static int configured = 0;
  if (lds > 48 * 1024 && lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(had_transform_planes_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return QUIP_ERR_LAUNCH;
    configured = lds;
  }
  const int threads = L >= 1024 ? 256 : 64;
  hipLaunchKernelGGL(had_transform_planes_kernel, dim3(K), dim3(threads), lds, stream,
                     reinterpret_cast<const f16*>(x), reinterpret_cast<uint8_t*>(planes), in_features, n,
                     kp, K, L, reinterpret_cast<const f16*>(had), transpose,
                     reinterpret_cast<const f16*>(pre), scale);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

int had_transform_launch(const void* x, void* y, int64_t rows, int in_features, int out_features,
                         int n, int K, const void* had, int transpose, const void* pre,
                         const void* pre2, const void* post, const void* bias, float scale,
                         hipStream_t stream) {
  if (K < 1 || n % K != 0) return QUIP_ERR_BAD_SHAPE;
  const int L = n / K;
  if (L < 1 || (L & (L - 1)) != 0 || L > 32768) return QUIP_ERR_BAD_SHAPE;
  if (in_features > n || out_features > n || in_features < 1 || out_features < 1) return QUIP_ERR_BAD_SHAPE;
  if (K > 1 && !had) return QUIP_ERR_NULL_POINTER;
  if (rows <= 0) return QUIP_OK;
  if (rows > 65535) return QUIP_ERR_BAD_SHAPE;  // TODO(round 2): fold rows into grid.x for prefill
  const int lds = L * 4;
  static int configured = 0;
  if (lds > 64 * 1024 && lds > configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(had_transform_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return QUIP_ERR_LAUNCH;
    configured = lds;
  }
  const int threads = L >= 512 ? 256 : (L >= 128 ? 64 : 64);
  hipLaunchKernelGGL(had_transform_kernel, dim3(K, (unsigned)rows), dim3(threads), lds, stream,
                     reinterpret_cast<const f16*>(x), reinterpret_cast<f16*>(y), in_features,
                     out_features, K, L, reinterpret_cast<const f16*>(had), transpose,
                     reinterpret_cast<const f16*>(pre), reinterpret_cast<const f16*>(pre2),
                     reinterpret_cast<const f16*>(post), reinterpret_cast<const f16*>(bias), scale);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

}  // namespace quip
