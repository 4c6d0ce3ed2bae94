// quip_lib::hadamard for the dtypes the fused fp16 kernels do not take: bf16 and fp32 (the reference's op is
// fast_hadamard_transform_cuda.fast_hadamard_transform, register_lib.py:10-20, which accepts fp16 / bf16 / fp32 and
// computes in fp32).  y[r, :] = scale * H_n x[r, :], Sylvester order, n a power of two <= 32768.
//
// Not on the decode path (QuantLinear always transforms fp16 through hadamard.hip); this is the plain form: a row
// lives in LDS as fp32 (<= 128 KiB), log2(n) radix-2 passes, one workgroup per row.  HBM-bound for large batches,
// latency-bound for one row -- like the op it replaces.
#include <hip/hip_bf16.h>

#include "quip_device.hip.h"
#include "quip_internal.h"

namespace quip {

namespace {

template <typename T>
__device__ __forceinline__ float load_as_f32(const T* p, int64_t i);
template <>
__device__ __forceinline__ float load_as_f32<float>(const float* p, int64_t i) { return p[i]; }
template <>
__device__ __forceinline__ float load_as_f32<uint16_t>(const uint16_t* p, int64_t i) {   // bf16 bits
  return as_f32((uint32_t)p[i] << 16);
}
template <>
__device__ __forceinline__ float load_as_f32<f16>(const f16* p, int64_t i) { return (float)p[i]; }

__device__ __forceinline__ void store_from_f32(float* p, int64_t i, float v) { p[i] = v; }
__device__ __forceinline__ void store_from_f32(f16* p, int64_t i, float v) { p[i] = (f16)v; }
__device__ __forceinline__ void store_from_f32(uint16_t* p, int64_t i, float v) {   // round to nearest even bf16
  uint32_t u = as_u32(v);
  if ((u & 0x7fffffffu) > 0x7f800000u) {   // NaN stays NaN
    p[i] = (uint16_t)((u >> 16) | 0x40u);
    return;
  }
  u += 0x7fffu + ((u >> 16) & 1u);
  p[i] = (uint16_t)(u >> 16);
}

template <typename T>
__global__ __launch_bounds__(1024) void hadamard_generic_kernel(const T* __restrict__ x, T* __restrict__ y, int n,
                                                                int logn, float scale) {
  extern __shared__ float row[];
  const int64_t base = (int64_t)blockIdx.x * n;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < n; i += nt) row[i] = load_as_f32<T>(x, base + i);
  __syncthreads();
  for (int s = 0; s < logn; ++s) {
    const int h = 1 << s;
    for (int b = tid; b < (n >> 1); b += nt) {
      const int lo = ((b >> s) << (s + 1)) | (b & (h - 1));
      const float u = row[lo], v = row[lo + h];
      row[lo] = u + v;
      row[lo + h] = u - v;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += nt) store_from_f32(y, base + i, row[i] * scale);
}

template <typename T>
int launch_generic(const void* x, void* y, int64_t rows, int n, float scale, hipStream_t stream) {
  int logn = 0;
  while ((1 << logn) < n) ++logn;
  const int threads = n >= 2048 ? 1024 : (n >= 128 ? n / 2 : 64);
  const int lds = n * 4;
  auto kern = hadamard_generic_kernel<T>;
  static DynLdsCache configured;   // per instantiation, per device
  if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
  for (int64_t r0 = 0; r0 < rows; r0 += 1 << 30) {   // grid.x limit
    const int64_t m = rows - r0 < (1 << 30) ? rows - r0 : (1 << 30);
    hipLaunchKernelGGL(kern, dim3((unsigned)m), dim3(threads), lds, stream, reinterpret_cast<const T*>(x) + r0 * n,
                       reinterpret_cast<T*>(y) + r0 * n, n, logn, scale);
  }
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

}  // namespace

int hadamard_generic_launch(const void* x, void* y, int64_t rows, int n, float scale, int dtype, hipStream_t stream) {
  if (n < 1 || n > 32768 || (n & (n - 1)) != 0 || rows < 0) return QUIP_ERR_BAD_SHAPE;
  if (rows == 0) return QUIP_OK;
  if (dtype == QUIP_DTYPE_BF16) return launch_generic<uint16_t>(x, y, rows, n, scale, stream);
  if (dtype == QUIP_DTYPE_F32) return launch_generic<float>(x, y, rows, n, scale, stream);
  if (dtype == QUIP_DTYPE_F16) return launch_generic<f16>(x, y, rows, n, scale, stream);
  return QUIP_ERR_UNSUPPORTED;
}

}  // namespace quip
