// Issue cost of the instructions the E8P decode is made of, on gfx950, as the decode runs them: 1 / 2 / 4 waves per SIMD, blocks
// of 16 independent instructions written in asm (nothing for the compiler to fold), s_memtime around 2000 blocks.
// Reported: ticks per wave-instruction as seen by ONE wave (time / instructions of a wave) and per SIMD (time / all instructions
// issued on the SIMD).  build: hipcc --offload-arch=gfx950 -O3 -o issue_rate issue_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { PERM, SDWA_MOV, AND_OR, XOR, BITOP3, LSHL_OR, BFE, DSREAD, PERM_DS, SDWA_DS, XOR_DS, MIX_NIB, MIX_SDWA, MFMA, MIX_NIB_MFMA, NKINDS };
static const char* kNames[NKINDS] = {"v_perm_b32", "v_mov_b32_sdwa (byte -> byte 1, preserve)", "v_and_or_b32", "v_xor_b32", "v_bitop3_b32", "v_lshl_or_b32",
                                     "v_bfe_u32", "ds_read_b32 (conflict free)", "v_perm_b32 + ds_read_b32 pairs", "v_mov_sdwa + ds_read_b32 pairs",
                                     "v_xor_b32 + ds_read_b32 pairs", "nibble item mix: 2 perm 2 ds 1 xor 1 bitop3", "same with sdwa movs for perms",
                                     "v_mfma_i32_16x16x64_i8 (2 chains)", "nibble item mix + 1 mfma per 2 groups"};
static const int kPerBlock[NKINDS] = {16, 16, 16, 16, 16, 16, 16, 16, 32, 32, 32, 48, 48, 2, 52};

template <int KIND, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void kern(int iters, uint64_t* cyc, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 1024 / 4; i += 64 * WAVES) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u;
  __syncthreads();
  uint32_t a[16], b[16];
  uint32_t x = (uint32_t)tid * 747796405u + 2891336453u;
  const uint32_t lane_c = ((uint32_t)lane & 31u) << 2;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    x = x * 1664525u + 1013904223u;
    a[j] = x;                                   // "codes"
    b[j] = ((x >> 9) & 0xff00u) | lane_c;       // valid look-up addresses (byte 1 = index)
  }
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  i32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, A = {(int)x, (int)x, 1, 2};
  const uint32_t sel = 0x0c0c0500u, m0f = 0x0f0f0f0fu;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == PERM) {
#define X(j) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(b[j]) : "v"(a[j]), "v"(lane_c), "s"(sel));
      R16(X)
#undef X
    } else if constexpr (KIND == SDWA_MOV) {
#define X(j) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(b[j]) : "v"(a[j]));
      R16(X)
#undef X
    } else if constexpr (KIND == AND_OR) {
#define X(j) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(b[j]) : "v"(a[j]), "s"(0xff00u), "v"(lane_c));
      R16(X)
#undef X
    } else if constexpr (KIND == XOR) {
#define X(j) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(b[j]) : "v"(a[j]), "v"(a[(j + 1) & 15]));
      R16(X)
#undef X
    } else if constexpr (KIND == BITOP3) {
#define X(j) asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x48" : "=v"(b[j]) : "v"(a[j]), "s"(m0f), "v"(a[(j + 1) & 15]));
      R16(X)
#undef X
    } else if constexpr (KIND == LSHL_OR) {
#define X(j) asm volatile("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(b[j]) : "v"(a[j]), "v"(lane_c));
      R16(X)
#undef X
    } else if constexpr (KIND == BFE) {
#define X(j) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(b[j]) : "v"(a[j]));
      R16(X)
#undef X
    } else if constexpr (KIND == DSREAD) {
#define X(j) asm volatile("ds_read_b32 %0, %1" : "=v"(a[j]) : "v"(b[j]) : "memory");
      R16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == PERM_DS || KIND == SDWA_DS || KIND == XOR_DS) {
      uint32_t r[16];
#define X(j) asm volatile("ds_read_b32 %0, %1" : "=v"(r[j]) : "v"(b[j]) : "memory");
      R16(X)
#undef X
      if constexpr (KIND == PERM_DS) {
#define X(j) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(b[j]) : "v"(a[j]), "v"(lane_c), "s"(sel));
        R16(X)
#undef X
      } else if constexpr (KIND == SDWA_DS) {
#define X(j) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(b[j]) : "v"(a[j]));
        R16(X)
#undef X
      } else {
#define X(j) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(a[j]) : "v"(a[j]), "v"(a[(j + 1) & 15]));
        R16(X)
#undef X
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define X(j) asm volatile("" :: "v"(r[j]));
      R16(X)
#undef X
    } else if constexpr (KIND == MIX_NIB || KIND == MIX_SDWA || KIND == MIX_NIB_MFMA) {
      // 8 groups of (2 addresses, 2 look-ups, xor, masked xor): half an item
      uint32_t r[16], u[8], m[8];
#define X(j)                                                                                                              \
  if constexpr (KIND == MIX_SDWA) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3" : "+v"(b[j]) : "v"(a[j])); \
  else asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(b[j]) : "v"(a[j]), "v"(lane_c), "s"(sel));
      R16(X)
#undef X
#define X(j) asm volatile("ds_read_b32 %0, %1" : "=v"(r[j]) : "v"(b[j]) : "memory");
      R16(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        asm volatile("v_xor_b32 %0, %1, %2" : "=v"(u[g]) : "v"(r[2 * g]), "v"(r[2 * g + 1]));
        asm volatile("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x48" : "=v"(m[g]) : "v"(r[2 * g]), "s"(m0f), "v"(r[2 * g + 1]));
      }
      if constexpr (KIND == MIX_NIB_MFMA) {
        const i32x4 B0 = {(int)u[0], (int)u[1], (int)u[2], (int)u[3]}, B1 = {(int)m[0], (int)m[1], (int)m[2], (int)m[3]};
        const i32x4 B2 = {(int)u[4], (int)u[5], (int)u[6], (int)u[7]}, B3 = {(int)m[4], (int)m[5], (int)m[6], (int)m[7]};
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B0, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B1, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B2, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B3, acc1, 0, 0, 0);
      } else {
#pragma unroll
        for (int g = 0; g < 8; ++g) asm volatile("" :: "v"(u[g]), "v"(m[g]));
      }
    } else if constexpr (KIND == MFMA) {
      const i32x4 B0 = {(int)a[0], (int)a[1], (int)a[2], (int)a[3]};
      acc0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B0, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(a[j]), "+v"(b[j]));
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  uint32_t acc = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc ^= a[j] ^ b[j];
  acc ^= (uint32_t)(acc0.x ^ acc1.y);
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x1234567u) sink[0] = acc;
}

static uint64_t* g_cyc;
static uint32_t* g_sink;

template <int KIND, int WAVES>
static double run(int iters) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern<KIND, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((kern<KIND, WAVES>), dim3(64), dim3(64 * WAVES), 64 * 1024, 0, iters, g_cyc, g_sink);
    CHECK(hipDeviceSynchronize());
    uint64_t h[64];
    CHECK(hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < 64; ++i) mean += (double)h[i];
    mean /= 64;
    if (mean < best) best = mean;
  }
  return best;
}

template <int KIND>
static void report() {
  const int iters = 2000;
  const double n = (double)iters * kPerBlock[KIND];
  const double t4 = run<KIND, 4>(iters), t8 = run<KIND, 8>(iters), t16 = run<KIND, 16>(iters);
  printf("%-48s per wave %6.2f %6.2f %6.2f | per SIMD %6.2f %6.2f %6.2f   (1 / 2 / 4 waves per SIMD)\n", kNames[KIND], t4 / n, t8 / n, t16 / n, t4 / n,
         t8 / n / 2, t16 / n / 4);
}

int main() {
  CHECK(hipMalloc(&g_cyc, 8 * 64));
  CHECK(hipMalloc(&g_sink, 4));
  printf("s_memtime ticks per instruction (a block's instruction count includes every instruction named in the row)\n");
  report<PERM>(); report<SDWA_MOV>(); report<AND_OR>(); report<XOR>(); report<BITOP3>(); report<LSHL_OR>(); report<BFE>(); report<DSREAD>();
  report<PERM_DS>(); report<SDWA_DS>(); report<XOR_DS>(); report<MIX_NIB>(); report<MIX_SDWA>(); report<MFMA>(); report<MIX_NIB_MFMA>();
  return 0;
}
