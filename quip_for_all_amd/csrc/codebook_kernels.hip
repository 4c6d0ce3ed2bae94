// Generic (correctness-first) kernels for all five codebooks of the reference:
//   * decompress_*_origorder  (origin_order.cu:794-1074)
//   * *_mm_origorder for shapes / codebooks the specialised kernels do not cover
//     (origin_order.cu:388-555 with BLayout_{D4,HI,E8,E8RVQ3,E8RVQ4}, :143-385)
// One decode functor per codebook yields 8 fp16 weights (16 B) of a packed row.
#include "quip_device.hip.h"
#include "quip_internal.h"

namespace quip {

// ---- decode functors ----------------------------------------------------------
struct DecE8P {
  static constexpr int kLds = 4096;
  const uint64_t* grid;
  __device__ void fill(char* smem, int tid, int nt) const {
    for (int e = tid; e < 256; e += nt) reinterpret_cast<uint4*>(smem)[e] = e8p_abs_row_f16(grid[e]);
  }
  __host__ __device__ static int64_t row_bytes(int k) { return k / 4; }
  __device__ uint4 operator()(const uint8_t* qrow, int u, const char* smem) const {
    const uint32_t c = reinterpret_cast<const uint16_t*>(qrow)[u];
    return e8p_decode_f16(c, reinterpret_cast<const uint4*>(smem)[c >> 8]);
  }
};

struct DecE8PRVQ4 {
  static constexpr int kLds = 4096;
  const uint64_t* grid;
  float scale;
  __device__ void fill(char* smem, int tid, int nt) const {
    for (int e = tid; e < 256; e += nt) reinterpret_cast<uint4*>(smem)[e] = e8p_abs_row_f16(grid[e]);
  }
  __host__ __device__ static int64_t row_bytes(int k) { return k / 2; }
  __device__ uint4 operator()(const uint8_t* qrow, int u, const char* smem) const {
    const uint32_t c = reinterpret_cast<const uint32_t*>(qrow)[u];
    const uint4* t = reinterpret_cast<const uint4*>(smem);
    const uint4 main = e8p_decode_f16(c >> 16, t[c >> 24]);
    const uint4 res = e8p_decode_f16(c & 0xffffu, t[(c >> 8) & 0xff]);
    const f16 s = (f16)scale;  // __float2half2_rn(scale)
    return rvq_combine(main, res, f16x2{s, s});
  }
};

struct DecE8PRVQ3 {
  static constexpr int kLds = 4096 + 1024;
  const uint64_t* grid;
  const uint32_t* grid2;
  float scale;
  __device__ void fill(char* smem, int tid, int nt) const {
    for (int e = tid; e < 256; e += nt) {
      reinterpret_cast<uint4*>(smem)[e] = e8p_abs_row_f16(grid[e]);
      reinterpret_cast<uint32_t*>(smem + 4096)[e] = grid2[e];
    }
  }
  __host__ __device__ static int64_t row_bytes(int k) { return (int64_t)k * 3 / 8; }
  __device__ uint4 operator()(const uint8_t* qrow, int u, const char* smem) const {
    const uint8_t* p = qrow + 3 * u;  // byte0 residual idx, byte1 sign byte, byte2 abs idx
    const uint32_t r = p[0], c = (uint32_t)p[1] | ((uint32_t)p[2] << 8);
    const uint4 main = e8p_decode_f16(c, reinterpret_cast<const uint4*>(smem)[c >> 8]);
    const uint4 res = e81b_decode_f16(reinterpret_cast<const uint32_t*>(smem + 4096)[r]);
    const f16 s = (f16)scale;
    return rvq_combine(main, res, f16x2{s, s});
  }
};

struct DecD4 {
  static constexpr int kLds = 2048;
  const uint2* grid;  // 256 x 4 fp16
  __device__ void fill(char* smem, int tid, int nt) const {
    for (int e = tid; e < 256; e += nt) reinterpret_cast<uint2*>(smem)[e] = grid[e];
  }
  __host__ __device__ static int64_t row_bytes(int k) { return k / 4; }
  __device__ uint4 operator()(const uint8_t* qrow, int u, const char* smem) const {
    const uint2* t = reinterpret_cast<const uint2*>(smem);
    const uint2 a = t[qrow[2 * u]], b = t[qrow[2 * u + 1]];
    return make_uint4(a.x, a.y, b.x, b.y);
  }
};

struct DecHI {
  static constexpr int kLds = 16;
  __device__ void fill(char*, int, int) const {}
  __host__ __device__ static int64_t row_bytes(int k) { return k / 2; }
  __device__ uint4 operator()(const uint8_t* qrow, int u, const char*) const {
    return hi_decode_f16(reinterpret_cast<const uint32_t*>(qrow)[u]);
  }
};

// ---- decompress: one thread per 8 weights, 16-byte stores ------------------------
template <class Dec>
__global__ __launch_bounds__(256) void decompress_kernel(const uint8_t* __restrict__ q, Dec dec,
                                                         uint4* __restrict__ w, int64_t rows,
                                                         int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dec.fill(smem, threadIdx.x, blockDim.x);
  __syncthreads();
  const int units = k >> 3;
  const int64_t total = rows * units;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / units;
    const int u = (int)(i - row * units);
    w[i] = dec(q + row * Dec::row_bytes(k), u, smem);
  }
}

// ---- generic mm: wave per output row, MT token rows per pass -----------------------
template <class Dec, int MT>
__global__ __launch_bounds__(256) void generic_mm_kernel(const f16* __restrict__ x,
                                                         const uint8_t* __restrict__ q, Dec dec,
                                                         f16* __restrict__ y, int m, int n, int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  dec.fill(smem, threadIdx.x, blockDim.x);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= n) return;
  const int units = k >> 3;
  const uint8_t* qrow = q + (int64_t)row * Dec::row_bytes(k);
  for (int m0 = 0; m0 < m; m0 += MT) {
    float acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = 0.f;
    for (int u = lane; u < units; u += 64) {
      const uint4 w = dec(qrow, u, smem);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        if (m0 + i < m) {
          const uint4 xv = *reinterpret_cast<const uint4*>(x + (int64_t)(m0 + i) * k + u * 8);
          acc[i] = dot2(w.x, xv.x, dot2(w.y, xv.y, dot2(w.z, xv.z, dot2(w.w, xv.w, acc[i]))));
        }
      }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float tot = wave_sum_all(acc[i]);
      if (lane == 0 && m0 + i < m) y[(int64_t)(m0 + i) * n + row] = (f16)tot;
    }
  }
}

template <class Dec>
static int launch_mm(const Dec& dec, const void* x, const void* q, void* y, int m, int n, int k,
                     hipStream_t s) {
  const int waves = 4;
  dim3 grid((n + waves - 1) / waves), block(64 * waves);
  hipLaunchKernelGGL((generic_mm_kernel<Dec, 8>), grid, block, Dec::kLds, s,
                     reinterpret_cast<const f16*>(x), reinterpret_cast<const uint8_t*>(q), dec,
                     reinterpret_cast<f16*>(y), m, n, k);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

template <class Dec>
static int launch_dec(const Dec& dec, const void* q, void* w, int64_t rows, int k, hipStream_t s) {
  const int64_t total = rows * (k >> 3);
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = (int64_t)device_cu_count() * 16;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((decompress_kernel<Dec>), dim3((unsigned)blocks), dim3(256), Dec::kLds, s,
                     reinterpret_cast<const uint8_t*>(q), dec, reinterpret_cast<uint4*>(w), rows, k);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

int generic_mm_launch(CodebookId cb, const void* x, const void* q, const CodebookArgs& a, void* y,
                      int m, int n, int k, hipStream_t s) {
  switch (cb) {
    case kE8P: return launch_mm(DecE8P{(const uint64_t*)a.grid}, x, q, y, m, n, k, s);
    case kE8PRVQ4: return launch_mm(DecE8PRVQ4{(const uint64_t*)a.grid, a.scale}, x, q, y, m, n, k, s);
    case kE8PRVQ3:
      return launch_mm(DecE8PRVQ3{(const uint64_t*)a.grid, (const uint32_t*)a.grid2, a.scale}, x, q, y, m, n, k, s);
    case kD4: return launch_mm(DecD4{(const uint2*)a.grid}, x, q, y, m, n, k, s);
    case kHI: return launch_mm(DecHI{}, x, q, y, m, n, k, s);
  }
  return QUIP_ERR_UNSUPPORTED;
}

int decompress_launch(CodebookId cb, const void* q, const CodebookArgs& a, void* w, int64_t rows,
                      int k, hipStream_t s) {
  switch (cb) {
    case kE8P: return launch_dec(DecE8P{(const uint64_t*)a.grid}, q, w, rows, k, s);
    case kE8PRVQ4: return launch_dec(DecE8PRVQ4{(const uint64_t*)a.grid, a.scale}, q, w, rows, k, s);
    case kE8PRVQ3:
      return launch_dec(DecE8PRVQ3{(const uint64_t*)a.grid, (const uint32_t*)a.grid2, a.scale}, q, w, rows, k, s);
    case kD4: return launch_dec(DecD4{(const uint2*)a.grid}, q, w, rows, k, s);
    case kHI: return launch_dec(DecHI{}, q, w, rows, k, s);
  }
  return QUIP_ERR_UNSUPPORTED;
}

}  // namespace quip
