// What a table look-up of the E8P decode costs the LDS, isolated (round-5 review, "weak" 4: tools/ubench/lds_b64_groups.hip could
// not see it -- every index there came from a dependent VALU multiply, so it timed the VALU chain).  Here every lane holds 16
// PRECOMPUTED random addresses in registers, a loop iteration is 16 ds_read in flight behind ONE `s_waitcnt lgkmcnt(0)`, nothing
// else issues, and 4 / 8 / 16 waves per CU run it (one workgroup per CU, 256 workgroups).  Reported: s_memtime ticks per
// wave-instruction and CU (= workgroup time / (waves x instructions per wave)), against MI355X_MICROARCH.md (LDS): 2 cycles for a
// conflict-free ds_read_b32 / ds_read_b64, N x for N-way conflicts, 4 for ds_read_b128.
//
// Part 2: the decode of ONE ITEM (16 rows x 512 k: 32 look-ups + A fragments + MFMAs) exactly as the kernels do it
// (e8p_gemv_core.hip.h), codes from registers, no HBM: byte tables 16 / 16, 32 / 16, 32 / 32 copies against the nibble mode.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o lds_lookup_rate lds_lookup_rate.hip      run: ./lds_lookup_rate
// counters: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -- ./lds_lookup_rate   (kernel names carry the case)
#include "../../quip_for_all_amd/csrc/e8p_gemv_core.hip.h"
#include <cstdio>
#include <cstdlib>
using namespace quip;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

enum { B64_16 = 0, B64_32 = 1, B32_32_ROW256 = 2, B32_32_DENSE = 3, B32_16 = 4, B128_FRAG = 5, B64_16_SAMEPAR = 6 };

template <int MODE>
__device__ __forceinline__ uint32_t address_of(uint32_t idx, int lane) {
  switch (MODE) {
    case B64_16: return idx * 128u + ((uint32_t)lane & 15u) * 8u;          // rounds 1-5, sign table (and abs table of the 70B launch)
    case B64_16_SAMEPAR: return (idx & ~1u) * 128u + ((uint32_t)lane & 15u) * 8u;   // every index even: lanes l, l + 16 always collide
    case B64_32: return idx * 256u + ((uint32_t)lane & 31u) * 8u;          // 32 copies: conflict free
    case B32_32_ROW256: return idx * 256u + ((uint32_t)lane & 31u) * 4u;   // nibble mode: T1n | T2n interleaved in 256-byte rows
    case B32_32_DENSE: return idx * 128u + ((uint32_t)lane & 31u) * 4u;    // 4-byte entries, 32 copies, 128-byte rows
    case B32_16: return idx * 64u + ((uint32_t)lane & 15u) * 4u;           // 4-byte entries, 16 copies: two-way
    default: return (idx & 63u) * 1024u + ((uint32_t)lane >> 4) * 64u + (uint32_t)min(lane & 15, 2) * 16u * 17u;   // A fragments: 3 planes + broadcast rows
  }
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lookups(int iters, uint64_t* cyc, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 1024 / 4; i += 64 * WAVES) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u;
  __syncthreads();
  uint32_t ad[16];
  // (the A fragments of an item are the same K slice for every lane: a wave-uniform index there)
  uint32_t x = (uint32_t)((MODE == B128_FRAG ? (tid >> 6) : tid) + 977 * blockIdx.x) * 747796405u + 2891336453u;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    x = x * 1664525u + 1013904223u;
    ad[j] = address_of<MODE>((x >> 13) & 255u, lane);
  }
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == B128_FRAG) {
      u32x4 v[16];
      asm volatile(
          "ds_read_b128 %0, %16\n\tds_read_b128 %1, %17\n\tds_read_b128 %2, %18\n\tds_read_b128 %3, %19\n\t"
          "ds_read_b128 %4, %20\n\tds_read_b128 %5, %21\n\tds_read_b128 %6, %22\n\tds_read_b128 %7, %23\n\t"
          "ds_read_b128 %8, %24\n\tds_read_b128 %9, %25\n\tds_read_b128 %10, %26\n\tds_read_b128 %11, %27\n\t"
          "ds_read_b128 %12, %28\n\tds_read_b128 %13, %29\n\tds_read_b128 %14, %30\n\tds_read_b128 %15, %31\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
            "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
          : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "v"(ad[8]), "v"(ad[9]),
            "v"(ad[10]), "v"(ad[11]), "v"(ad[12]), "v"(ad[13]), "v"(ad[14]), "v"(ad[15])
          : "memory");
      if (it == iters - 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j].x ^ v[j].w;
      }
    } else if constexpr (MODE == B64_16 || MODE == B64_32 || MODE == B64_16_SAMEPAR) {
      u32x2 v[16];
      asm volatile(
          "ds_read_b64 %0, %16\n\tds_read_b64 %1, %17\n\tds_read_b64 %2, %18\n\tds_read_b64 %3, %19\n\t"
          "ds_read_b64 %4, %20\n\tds_read_b64 %5, %21\n\tds_read_b64 %6, %22\n\tds_read_b64 %7, %23\n\t"
          "ds_read_b64 %8, %24\n\tds_read_b64 %9, %25\n\tds_read_b64 %10, %26\n\tds_read_b64 %11, %27\n\t"
          "ds_read_b64 %12, %28\n\tds_read_b64 %13, %29\n\tds_read_b64 %14, %30\n\tds_read_b64 %15, %31\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
            "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
          : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "v"(ad[8]), "v"(ad[9]),
            "v"(ad[10]), "v"(ad[11]), "v"(ad[12]), "v"(ad[13]), "v"(ad[14]), "v"(ad[15])
          : "memory");
      if (it == iters - 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j].x ^ v[j].y;
      }
    } else {
      uint32_t v[16];
      asm volatile(
          "ds_read_b32 %0, %16\n\tds_read_b32 %1, %17\n\tds_read_b32 %2, %18\n\tds_read_b32 %3, %19\n\t"
          "ds_read_b32 %4, %20\n\tds_read_b32 %5, %21\n\tds_read_b32 %6, %22\n\tds_read_b32 %7, %23\n\t"
          "ds_read_b32 %8, %24\n\tds_read_b32 %9, %25\n\tds_read_b32 %10, %26\n\tds_read_b32 %11, %27\n\t"
          "ds_read_b32 %12, %28\n\tds_read_b32 %13, %29\n\tds_read_b32 %14, %30\n\tds_read_b32 %15, %31\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]),
            "=&v"(v[9]), "=&v"(v[10]), "=&v"(v[11]), "=&v"(v[12]), "=&v"(v[13]), "=&v"(v[14]), "=&v"(v[15])
          : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "v"(ad[8]), "v"(ad[9]),
            "v"(ad[10]), "v"(ad[11]), "v"(ad[12]), "v"(ad[13]), "v"(ad[14]), "v"(ad[15])
          : "memory");
      if (it == iters - 1) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j];
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  const uint64_t t2 = __builtin_amdgcn_s_memtime();
  if (tid == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = t2 - t0; }
  if (acc == 0x1234567u) sink[0] = acc;
}

// ---- part 2: one item's decode ----------------------------------------------------------------------------------------------
// REP: 16 / 24 / 32 = the byte tables (e8p_gemv_core.hip.h Lds<REP>), 4 = nibble mode.  Every wave decodes and multiplies `iters`
// items whose codes are a register ring of four random (q0, q1) pairs; the A fragments come from an LDS image of garbage digits.
template <int REP, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void items(const uint64_t* grid, int iters, uint64_t* cyc, uint32_t* sink) {
  using namespace quip;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = Lds<REP>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int kX = T::kNib ? kNibTableBytes : T::kT3;
  for (int i = tid; i < 24 * 1024 / 4; i += 64 * WAVES) reinterpret_cast<uint32_t*>(smem + kX)[i] = (uint32_t)i * 2654435761u;
  if (wave < 8) {
    if constexpr (T::kNib) {
      const uint2 s = *table_source_ptr_nib(grid, lane, wave);
      fill_tables_nib(u32x2{s.x, s.y}, lane, wave);
    } else {
      const uint2 s = *table_source_ptr(grid, lane, wave);
      fill_tables_from_lane<REP>(smem, u32x2{s.x, s.y}, lane, wave);
    }
  }
  __syncthreads();
  u32x4 qa[4], qb[4];
  uint32_t x = (uint32_t)(tid + 977 * blockIdx.x) * 747796405u + 2891336453u;
  for (int j = 0; j < 4; ++j) {
    uint32_t r[8];
    for (int k = 0; k < 8; ++k) { x = x * 1664525u + 1013904223u; r[k] = x ^ (x >> 15); }
    qa[j] = u32x4{r[0], r[1], r[2], r[3]};
    qb[j] = u32x4{r[4], r[5], r[6], r[7]};
  }
  const int n = lane & 15, q = lane >> 4;
  uint32_t lane_c, lane_c2 = 0;
  if constexpr (T::kNib) lane_c = nib_lane_const(lane);
  else {
    lane_c = (T::kRep1 == 32) ? ((((uint32_t)lane & 31u) << 3) | (REP == 32 ? 0x00010000u : 0u)) : ((((uint32_t)lane & 15u) << 3) | (uint32_t)T::kT1);
    lane_c2 = (((uint32_t)lane & 15u) << 3) | (uint32_t)T::kT2;
  }
  i32x4 accr = {0, 0, 0, 0}, accm = {0, 0, 0, 0}, sx = {0, 0, 0, 0};
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ItemAddr ad;
      asm volatile("" : "+v"(qa[j]), "+v"(qb[j]));
      if constexpr (T::kNib) {
        item_addresses_nib(qa[j], qb[j], lane_c, ad);
        const uint32_t xa = (uint32_t)kX + (uint32_t)(n < 4 ? min(n, 2) * (8192 + 64) + 4096 + 16 : min(n - 4, 2) * (8192 + 64)) + (uint32_t)q * 32u +
                            (uint32_t)((wave + j) & 7) * 256u;
        i32x4 A[4];
        item_fragments_nib(xa, A);
        if (j == 0) item_digit_sums_nib(A, sx);      // (once per two items in the 70B launch's worst product, once per 14 in its best)
        item_mfma_nib(ad, A, accr, accm);
      } else {
        item_addresses<REP>(qa[j], qb[j], lane_c, lane_c2, ad, 0u);
        const uint32_t xa = (uint32_t)kX + (uint32_t)min(n, 2) * (8192u + 16u) + (uint32_t)q * 64u + (uint32_t)((wave + j) & 7) * 512u;
        const i32x4 r = item_mfma(ad, xa);
        accr.x += r.x; accr.y += r.y; accr.z += r.z;
      }
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  const uint64_t t2 = __builtin_amdgcn_s_memtime();
  if (tid == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = t2 - t0; }
  if ((accr.x ^ accr.y ^ accr.z ^ accm.x ^ accm.y ^ sx.x) == 0x1234567) sink[0] = 1;
}

// ---- part 3: variants of the nibble item (what bounds it?) ------------------------------------------------------------------
// VAR 0: as shipped (two accumulators)   1: four accumulators   2: all 32 look-ups first, then the arithmetic
//     3: addresses by v_and_or / v_lshrrev instead of v_perm   4: no MFMAs (look-ups + VALU)   5: no look-ups (VALU + MFMAs on the codes)
//     6: look-ups only (addresses + 32 ds_read, results xor-folded)   7: two items interleaved (A B A B ...)
template <int VAR, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void nibvar(const uint64_t* grid, int iters, uint64_t* cyc, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int kX = kNibTableBytes;
  for (int i = tid; i < 24 * 1024 / 4; i += 64 * WAVES) reinterpret_cast<uint32_t*>(smem + kX)[i] = (uint32_t)i * 2654435761u;
  if (wave < 8) {
    const uint2 s = *table_source_ptr_nib(grid, lane, wave);
    fill_tables_nib(u32x2{s.x, s.y}, lane, wave);
  }
  __syncthreads();
  u32x4 qa[4], qb[4];
  uint32_t x = (uint32_t)(tid + 977 * blockIdx.x) * 747796405u + 2891336453u;
  for (int j = 0; j < 4; ++j) {
    uint32_t r[8];
    for (int k = 0; k < 8; ++k) { x = x * 1664525u + 1013904223u; r[k] = x ^ (x >> 15); }
    qa[j] = u32x4{r[0], r[1], r[2], r[3]};
    qb[j] = u32x4{r[4], r[5], r[6], r[7]};
  }
  const int n = lane & 15, q = lane >> 4;
  const uint32_t lane_c = nib_lane_const(lane);
  i32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  const uint32_t xa0 = (uint32_t)kX + (uint32_t)(n < 4 ? min(n, 2) * (8192 + 64) + 4096 + 16 : min(n - 4, 2) * (8192 + 64)) + (uint32_t)q * 32u;
  i32x4 A[4];
  item_fragments_nib(xa0 + (uint32_t)wave * 256u, A);
  auto addresses = [&](const u32x4& a, const u32x4& b, ItemAddr& ad) {
    if constexpr (VAR == 3) {
      const uint32_t d[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const uint32_t c1 = lane_c & 0xffu, c2 = (lane_c >> 16) & 0xffu;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        ad.a1l[t] = (d[t] & 0xff00u) | c1;
        ad.a2l[t] = ((d[t] << 8) & 0xff00u) | c2;
        ad.a1h[t] = ((d[t] >> 16) & 0xff00u) | c1;
        ad.a2h[t] = ((d[t] >> 8) & 0xff00u) | c2;
      }
    } else {
      item_addresses_nib(a, b, lane_c, ad);
    }
  };
  auto one = [&](const ItemAddr& ad, i32x4& r0, i32x4& m0, i32x4& r1, i32x4& m1) {
    uint32_t o[4][8];
    auto lk = [&](int s) {
      o[s][0] = lds_read4(ad.a1l[2 * s]); o[s][1] = lds_read4(ad.a2l[2 * s]);
      o[s][2] = lds_read4(ad.a1h[2 * s]); o[s][3] = lds_read4(ad.a2h[2 * s]);
      o[s][4] = lds_read4(ad.a1l[2 * s + 1]); o[s][5] = lds_read4(ad.a2l[2 * s + 1]);
      o[s][6] = lds_read4(ad.a1h[2 * s + 1]); o[s][7] = lds_read4(ad.a2h[2 * s + 1]);
    };
    if constexpr (VAR == 5) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) o[s][j] = (j & 1) ? ad.a2l[2 * s + (j >> 2)] : ad.a1h[2 * s + (j >> 2)];
    } else if constexpr (VAR == 2) {
      lk(0); lk(1); lk(2); lk(3);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(o[s][j]));
    } else {
      lk(0); lk(1);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (VAR != 2 && VAR != 5) { if (s + 2 < 4) lk(s + 2); }
      const i32x4 Br = {(int)(o[s][0] ^ o[s][1]), (int)(o[s][2] ^ o[s][3]), (int)(o[s][4] ^ o[s][5]), (int)(o[s][6] ^ o[s][7])};
      const i32x4 Bm = {Br.x & 0x0f0f0f0f, Br.y & 0x0f0f0f0f, Br.z & 0x0f0f0f0f, Br.w & 0x0f0f0f0f};
      if constexpr (VAR == 4 || VAR == 6) {
        r0.x ^= Br.x ^ Br.y ^ Br.z ^ Br.w;
        if constexpr (VAR == 4) m0.x ^= Bm.x ^ Bm.y ^ Bm.z ^ Bm.w;
      } else if constexpr (VAR == 1) {
        i32x4& rr = (s & 1) ? r1 : r0;
        i32x4& mm = (s & 1) ? m1 : m0;
        rr = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s], Br, rr, 0, 0, 0);
        mm = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s], Bm, mm, 0, 0, 0);
      } else {
        r0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s], Br, r0, 0, 0, 0);
        m0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s], Bm, m0, 0, 0, 0);
      }
    }
  };
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asm volatile("s_nop 0" : "+v"(qa[j]), "+v"(qb[j]) : : "memory");        // (the kernels' items are fenced by their asm waits / requests)
      ItemAddr ad;
      addresses(qa[j], qb[j], ad);
      one(ad, acc[0], acc[1], acc[2], acc[3]);
    }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  const uint64_t t2 = __builtin_amdgcn_s_memtime();
  if (tid == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = t2 - t0; }
  if ((acc[0].x ^ acc[0].y ^ acc[1].z ^ acc[2].x ^ acc[3].y) == 0x1234567) sink[0] = 1;
}

static uint64_t* g_cyc;
static uint32_t* g_sink;
static const int kWg = 256;

template <class K, class... Args>
static void run(const char* name, K kernel, int waves, size_t lds, double per_wave_instr, Args... args) {
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double best_mean = 1e30, best_ms = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(kWg), dim3(64 * waves), lds, 0, args..., g_cyc, g_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t h[2 * kWg];
    CHECK(hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mean = 0;
    for (int i = 0; i < kWg; ++i) mean += (double)h[2 * i + 1];
    mean /= kWg;
    if (mean < best_mean) best_mean = mean;
    if (ms < best_ms) best_ms = ms;
  }
  printf("  %-46s %2d waves  %7.2f ticks per wave-instruction and CU   (launch %.3f ms)\n", name, waves, best_mean / (per_wave_instr * waves), best_ms);
}

int main() {
  CHECK(hipMalloc(&g_cyc, 16 * kWg));
  CHECK(hipMalloc(&g_sink, 4));
  uint64_t* grid;
  CHECK(hipMalloc(&grid, 2048));
  {
    // any table will do for the timing: bytes in {2, 6, 10, 14}
    uint64_t h[256];
    for (int i = 0; i < 256; ++i) {
      uint64_t v = 0;
      for (int b = 0; b < 8; ++b) v |= (uint64_t)(2 + 4 * ((i >> (b & 3)) & 3)) << (8 * b);
      h[i] = v;
    }
    CHECK(hipMemcpy(grid, h, sizeof(h), hipMemcpyHostToDevice));
  }
  const int iters = 4000;
  const double n1 = iters * 16.0;
  printf("part 1: random look-ups, 16 in flight per lgkmcnt wait, addresses precomputed; 256 workgroups (one per CU)\n");
#define CASE(MODE, NAME)                                                       \
  run(NAME, lookups<MODE, 4>, 4, 64 * 1024, n1, iters);                        \
  run(NAME, lookups<MODE, 8>, 8, 64 * 1024, n1, iters);                        \
  run(NAME, lookups<MODE, 16>, 16, 64 * 1024, n1, iters);
  CASE(B64_16, "ds_read_b64, 16 copies (idx*128 + (l&15)*8)")
  CASE(B64_16_SAMEPAR, "ds_read_b64, 16 copies, all indices even")
  CASE(B64_32, "ds_read_b64, 32 copies (idx*256 + (l&31)*8)")
  CASE(B32_32_ROW256, "ds_read_b32, 32 copies, 256-byte rows (nibble)")
  CASE(B32_32_DENSE, "ds_read_b32, 32 copies, 128-byte rows")
  CASE(B32_16, "ds_read_b32, 16 copies (idx*64 + (l&15)*4)")
  CASE(B128_FRAG, "ds_read_b128, A-fragment pattern")
#undef CASE
  printf("part 2: the decode of one item (32 look-ups, A fragments, MFMAs) from registers: ticks per ITEM and CU (all waves' items / waves)\n");
  const int it2 = 2000;
#define ICASE(REP, NAME)                                                                             \
  run(NAME, items<REP, 8>, 8, (size_t)(quip::Lds<REP>::kNib ? quip::kNibTableBytes : quip::Lds<REP>::kT3) + 24 * 1024, (double)it2, grid, it2); \
  run(NAME, items<REP, 16>, 16, (size_t)(quip::Lds<REP>::kNib ? quip::kNibTableBytes : quip::Lds<REP>::kT3) + 24 * 1024, (double)it2, grid, it2);
  ICASE(16, "byte tables 16 / 16 copies (rounds 4-5, 70B)")
  ICASE(24, "byte tables 32 / 16 copies (7B launch)")
  ICASE(32, "byte tables 32 / 32 copies")
  ICASE(4, "nibble mode (round 6)")
#undef ICASE
  printf("part 3: variants of the nibble item, 8 waves per CU (ticks per item and CU; x 8 = ticks per item and wave)\n");
#define VCASE(V, NAME) run(NAME, nibvar<V, 8>, 8, (size_t)quip::kNibTableBytes + 24 * 1024, (double)it2, grid, it2);
  VCASE(0, "as shipped, items fenced")
  VCASE(1, "four accumulators")
  VCASE(2, "all 32 look-ups first")
  VCASE(3, "addresses without v_perm_b32")
  VCASE(4, "no MFMAs")
  VCASE(5, "no look-ups")
  VCASE(6, "look-ups only")
#undef VCASE
  printf("(an item = 2 KB of codes: at T ticks per item and CU the decode sustains 2048 / T bytes per clock and CU;\n"
         " the 70B launch's ring alone delivers 0.87 x 8 TB/s = 13.3 B per clock and CU at 2.05 GHz)\n");
  return 0;
}
