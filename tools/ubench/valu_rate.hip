// VALU / MFMA issue-rate micro-benchmark for gfx950: cycles per wave-instruction per SIMD
// for the instructions the decode GEMV is built from.  One workgroup of W waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef long i64;

#define CHAINS 8
#define ITERS 512

template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t a[CHAINS], b = seed | 1, c = threadIdx.x * 2654435761u;
  float fa[CHAINS];
  i32x4 macc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) { a[i] = c + i; fa[i] = (float)i; }
  __syncthreads();
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
      if (OP == 0) a[i] = __builtin_amdgcn_sdot4((int)b, (int)c, (int)a[i], false);          // v_dot4c_i32_i8
      if (OP == 1) fa[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, b), __builtin_bit_cast(f16x2, c), fa[i], false);  // v_dot2c_f32_f16
      if (OP == 2) a[i] = a[i] ^ b;                                                          // v_xor_b32
      if (OP == 3) a[i] = __builtin_amdgcn_perm(a[i], b, 0x0c020500u);                        // v_perm_b32
      if (OP == 4) fa[i] = __builtin_fmaf(fa[i], 1.0001f, 0.5f);                              // v_fma_f32
      if (OP == 5) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(f16x2, a[i]), __builtin_bit_cast(f16x2, b), __builtin_bit_cast(f16x2, c)));  // v_pk_fma_f16
      if (OP == 6) a[i] = (a[i] & b) | c;                                                     // v_and_or_b32
      if (OP == 7) a[i] = a[i] + b;                                                           // v_add_u32
      if (OP == 8) a[i] = __builtin_amdgcn_udot4(b, c, a[i], false);                          // v_dot4c? (u8)
      if (OP == 9) a[i] = __builtin_popcount(a[i]) + b;                                       // v_bcnt_u32_b32
      if (OP == 10) a[i] = __builtin_amdgcn_sdot8((int)b, (int)c, (int)a[i], false);          // v_dot8c_i32_i4
      if (OP == 11) a[i] = __builtin_amdgcn_sdot2(__builtin_bit_cast(short __attribute__((ext_vector_type(2))), b), __builtin_bit_cast(short __attribute__((ext_vector_type(2))), c), (int)a[i], false);  // v_dot2_i32_i16
    }
    if (OP == 20) {  // v_mfma_i32_16x16x64_i8, two independent accumulators
      i32x4 A = {(int)b, (int)c, (int)b, (int)c}, B = {(int)c, (int)b, (int)c, (int)b};
#pragma unroll
      for (int i = 0; i < CHAINS; ++i) macc[i & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, macc[i & 1], 0, 0, 0);
    }
  }
  uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; ++i) r ^= a[i] ^ __builtin_bit_cast(uint32_t, fa[i]);
  r ^= macc[0].x ^ macc[1].y;
  if (r == 0x13572468u) out[0] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves) {
  uint32_t* out; uint64_t* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<256, 64 * waves>>>(out, cyc, 12345);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<256, 64 * waves>>>(out, cyc, 12345);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= 256;
  const double insts_per_simd = (double)ITERS * CHAINS * (waves / 4.0);
  printf("%-22s waves/CU %2d: %8.0f ticks  -> %.2f ticks per wave-instr per SIMD ; wall %.1f us -> %.2f ns per wave-instr per SIMD\n",
         name, waves, avg, avg / insts_per_simd, ms * 1e3, ms * 1e6 / insts_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {4, 8, 16}) {
    run<0>("v_dot4c_i32_i8", w);
    run<1>("v_dot2c_f32_f16", w);
    run<2>("v_xor_b32", w);
    run<3>("v_perm_b32", w);
    run<4>("v_fma_f32", w);
    run<5>("v_pk_fma_f16", w);
    run<6>("v_and_or_b32", w);
    run<7>("v_add_u32", w);
    run<9>("v_bcnt_u32_b32", w);
    run<10>("v_dot8c_i32_i4", w);
    run<11>("v_dot2_i32_i16", w);
    run<20>("v_mfma_i32_16x16x64_i8", w);
  }
  return 0;
}
