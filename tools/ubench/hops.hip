// Hand-off micro-benchmark for the persistent decode engine (quip_for_all_amd/csrc/engine_sync.hip.h):
// cost per dependency edge INSIDE one launch of 256 x 512-thread workgroups (one per CU), with the exact
// protocols the engine uses, and a checksum of every word that crossed.
//
//   mode A  all-gather: every workgroup publishes T/NWG granules, every workgroup sweeps all T granules
//           (T = 2048: an fp16 vector of 4096 as 2-value granules; 4096: an fp32 vector of 4096)
//   mode C  the FFN edge: 256 column owners -> 43 row owners (512 granules each) -> every workgroup
//           (43 KB of fp32 rows behind 43 flags)
//   mode L  reference: the same number of dependent trivial launches (kernel boundary)
//
// build: hipcc --offload-arch=gfx950 -O3 -o hops hops.hip      run: ./hops [iters]
#include "../../quip_for_all_amd/csrc/engine_sync.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace quip::esync;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ uint32_t val_of(uint32_t g, uint32_t it) { return g * 2654435761u + it * 40503u + 17u; }

// ---- mode A ---------------------------------------------------------------------------------
// buf: 2 x T granules (double buffer by iteration parity: a workgroup that is one hop ahead must not
// overwrite granules a slower one is still waiting for)
template <int PAIRS>   // 16-byte pieces (2 granules) per thread per sweep: T = 512 * 2 * PAIRS
__global__ __launch_bounds__(512) void allgather_kernel(uint64_t* buf, int per_wg, int iters, uint32_t* err,
                                                        uint64_t* cyc, uint32_t* chk) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, w = blockIdx.x;
  constexpr int T = 512 * 2 * PAIRS;
  uint32_t acc = 0;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t epoch = (uint32_t)it + 1u;
    uint64_t* b = buf + (size_t)(it & 1) * T;
    if (tid < per_wg) {
      const uint32_t g = (uint32_t)(w * per_wg + tid);
      st_granule(b + g, val_of(g, it), epoch);
    }
    u32x4_t v[PAIRS];
    uint32_t spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) ld16(v[j], b + 2 * (tid + 512 * j));
      drain();
      bool ok = true;
#pragma unroll
      for (int j = 0; j < PAIRS; ++j) { own(v[j]); ok = ok && v[j].y == epoch && v[j].w == epoch; }
      if (spin_step(ok, spins, err, 0x100u + w)) break;
    }
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) acc += v[j].x + v[j].z;
    reinterpret_cast<uint32_t*>(smem)[tid] = acc;   // the engine stages the gathered vector in LDS
    __syncthreads();
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[w] = t1 - t0;
  atomicAdd(&chk[w], acc);
}

// ---- mode C ---------------------------------------------------------------------------------
// inbox: [NR rows][2][NWG] granules; frow: [NR][256] fp32; flags: [NR]
constexpr int NR = 43;
__global__ __launch_bounds__(512) void ffn_edge_kernel(uint64_t* inbox, uint32_t* frow, uint32_t* flags, int iters,
                                                       uint32_t* err, uint64_t* cyc, uint32_t* chk, uint64_t* stamps) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, w = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = gridDim.x;   // 256
  uint32_t acc = 0;
  uint64_t ts[4] = {0, 0, 0, 0};
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t epoch = (uint32_t)it + 1u;
    const uint64_t s0 = __builtin_amdgcn_s_memtime();
    // hop 1: 86 granules of this column owner, one to each (row, matrix) inbox
    if (tid < 2 * NR) {
      const int r = tid % NR, m = tid / NR;
      const uint32_t g = (uint32_t)((r * 2 + m) * nwg + w);
      st_granule(inbox + g, val_of(g, it), epoch);
    }
    // row owners: wave 0, lanes 0..31: 16 granules each (lanes 0..15 matrix 0, 16..31 matrix 1)
    if (w < NR && wave == 0) {
      u32x4_t v[8];
      uint32_t spins = 0;
      const uint64_t* src = inbox + (size_t)((w * 2 + ((lane >> 4) & 1)) * nwg + (lane & 15) * 16);
      for (;;) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ld16(v[j], src + 2 * j);
        drain();
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 8; ++j) { own(v[j]); ok = ok && v[j].y == epoch && v[j].w == epoch; }
        if (spin_step(ok || lane >= 32, spins, err, 0x200u + w)) break;
      }
      // stand-in for the two length-256 transforms: every output word = a value of matrix 0 + the
      // matching value of matrix 1 (brought over from lane + 16)
      u32x4_t o[4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t a = v[j].x + (uint32_t)__shfl((int)v[j].x, lane + 16, 64);
        const uint32_t b = v[j].z + (uint32_t)__shfl((int)v[j].z, lane + 16, 64);
        if (j & 1) { o[j >> 1].z = a; o[j >> 1].w = b; } else { o[j >> 1].x = a; o[j >> 1].y = b; }
      }
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) st_payload16(frow + (size_t)w * 256 + lane * 16 + 4 * j, o[j]);
      }
      drain();
      if (lane == 0) st_word(flags + w, epoch);
    }
    const uint64_t s1 = __builtin_amdgcn_s_memtime();
    // hop 2: everybody waits for the 43 flags (one wave polls), then reads the 43 rows
    if (wave == 0) {
      uint32_t spins = 0;
      for (;;) {
        uint32_t f = epoch;
        if (lane < NR) ld4(f, flags + lane);
        drain();
        own(f);
        if (spin_step(f == epoch, spins, err, 0x300u + w)) break;
      }
    }
    __syncthreads();
    const uint64_t s2 = __builtin_amdgcn_s_memtime();
    constexpr int PIECES = NR * 256 / 4;            // 2752 16-byte pieces
    u32x4_t p[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int i = tid + 512 * j;
      ld16(p[j], frow + 4 * (size_t)(i < PIECES ? i : 0));
    }
    drain();
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      own(p[j]);
      if (tid + 512 * j < PIECES) acc += p[j].x + p[j].y + p[j].z + p[j].w;
    }
    reinterpret_cast<uint32_t*>(smem)[tid] = acc;
    __syncthreads();
    const uint64_t s3 = __builtin_amdgcn_s_memtime();
    ts[0] += s1 - s0; ts[1] += s2 - s1; ts[2] += s3 - s2;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) {
    cyc[w] = t1 - t0;
    stamps[w * 4 + 0] = ts[0]; stamps[w * 4 + 1] = ts[1]; stamps[w * 4 + 2] = ts[2];
  }
  atomicAdd(&chk[w], acc);
}


// ---- 4-byte self-tagged granules: an fp32 whose low 4 mantissa bits carry (epoch & 15) ---------
__device__ __forceinline__ uint32_t tagged(uint32_t payload, uint32_t epoch) { return (payload & ~15u) | (epoch & 15u); }
__device__ __forceinline__ bool tag_ok(uint32_t v, uint32_t epoch) { return (v & 15u) == (epoch & 15u); }

// mode A4: all-gather of N dwords (N / NWG per workgroup, N % 2048 == 0), double buffered
template <int LOADS>   // dwordx4 loads per thread per sweep: N = 512 * 4 * LOADS
__global__ __launch_bounds__(512) void allgather4_kernel(uint32_t* buf, int per_wg, int iters, uint32_t* err,
                                                         uint64_t* cyc, uint32_t* chk) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, w = blockIdx.x;
  constexpr int N = 512 * 4 * LOADS;
  uint32_t acc = 0;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t epoch = (uint32_t)it + 1u;
    uint32_t* b = buf + (size_t)(it & 1) * N;
    for (int i = tid; i < per_wg; i += 512) {
      const uint32_t g = (uint32_t)(w * per_wg + i);
      st_word(b + g, tagged(val_of(g, it), epoch));
    }
    u32x4_t v[LOADS];
    uint32_t spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < LOADS; ++j) ld16(v[j], b + 4 * (tid + 512 * j));
      drain();
      bool ok = true;
#pragma unroll
      for (int j = 0; j < LOADS; ++j) {
        own(v[j]);
        ok = ok && tag_ok(v[j].x, epoch) && tag_ok(v[j].y, epoch) && tag_ok(v[j].z, epoch) && tag_ok(v[j].w, epoch);
      }
      if (spin_step(ok, spins, err, 0x400u + w)) break;
    }
#pragma unroll
    for (int j = 0; j < LOADS; ++j) acc += (v[j].x & ~15u) + (v[j].y & ~15u) + (v[j].z & ~15u) + (v[j].w & ~15u);
    reinterpret_cast<uint32_t*>(smem)[tid] = acc;
    __syncthreads();
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[w] = t1 - t0;
  atomicAdd(&chk[w], acc);
}

// mode C4: the FFN edge on self-tagged dwords.  inbox4: [NR][2][NWG] dwords; frow4: 2 x [NR][256] dwords (double
// buffered: every workgroup sweeps it)
__global__ __launch_bounds__(512) void ffn_edge4_kernel(uint32_t* inbox4, uint32_t* frow4, int iters, uint32_t* err,
                                                        uint64_t* cyc, uint32_t* chk, uint64_t* stamps, int delay_wg,
                                                        int delay_iters) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, w = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nwg = gridDim.x;
  uint32_t acc = 0;
  uint64_t ts[4] = {0, 0, 0, 0};
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const uint32_t epoch = (uint32_t)it + 1u;
    if (w == delay_wg + (it % 7)) {   // uneven load: one workgroup arrives late
      for (int d = 0; d < delay_iters; ++d) __builtin_amdgcn_s_sleep(16);
    }
    const uint64_t s0 = __builtin_amdgcn_s_memtime();
    if (tid < 2 * NR) {
      const int r = tid % NR, m = tid / NR;
      const uint32_t g = (uint32_t)((r * 2 + m) * nwg + w);
      st_word(inbox4 + g, tagged(val_of(g, it), epoch));
    }
    uint32_t* fb = frow4 + (size_t)(it & 1) * NR * 256;
    if (w < NR && wave < 2) {
      // wave m gathers matrix m: 256 dwords = one dwordx4 per lane
      u32x4_t v;
      uint32_t spins = 0;
      const uint32_t* src = inbox4 + (size_t)((w * 2 + wave) * nwg + lane * 4);
      for (;;) {
        ld16(v, src);
        drain();
        own(v);
        const bool ok = tag_ok(v.x, epoch) && tag_ok(v.y, epoch) && tag_ok(v.z, epoch) && tag_ok(v.w, epoch);
        if (spin_step(ok, spins, err, 0x500u + w)) break;
      }
      reinterpret_cast<u32x4_t*>(smem + 4096 + wave * 1024)[lane] = u32x4_t{v.x & ~15u, v.y & ~15u, v.z & ~15u, v.w & ~15u};
    }
    if (w < NR) {
      __syncthreads();
      if (wave == 0) {
        const u32x4_t a = reinterpret_cast<u32x4_t*>(smem + 4096)[lane], b = reinterpret_cast<u32x4_t*>(smem + 4096 + 1024)[lane];
        const u32x4_t o = {tagged(a.x + b.x, epoch), tagged(a.y + b.y, epoch), tagged(a.z + b.z, epoch), tagged(a.w + b.w, epoch)};
        st_payload16(fb + (size_t)w * 256 + lane * 4, o);
      }
    }
    const uint64_t s1 = __builtin_amdgcn_s_memtime();
    constexpr int PIECES = NR * 256 / 4;            // 2752 16-byte pieces
    u32x4_t p[6];
    uint32_t spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int i = tid + 512 * j;
        ld16(p[j], fb + 4 * (size_t)(i < PIECES ? i : 0));
      }
      drain();
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        own(p[j]);
        ok = ok && tag_ok(p[j].x, epoch) && tag_ok(p[j].y, epoch) && tag_ok(p[j].z, epoch) && tag_ok(p[j].w, epoch);
      }
      if (spin_step(ok, spins, err, 0x600u + w)) break;
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (tid + 512 * j < PIECES) acc += (p[j].x & ~15u) + (p[j].y & ~15u) + (p[j].z & ~15u) + (p[j].w & ~15u);
    reinterpret_cast<uint32_t*>(smem)[tid] = acc;
    __syncthreads();
    const uint64_t s2 = __builtin_amdgcn_s_memtime();
    ts[0] += s1 - s0; ts[1] += s2 - s1;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) {
    cyc[w] = t1 - t0;
    stamps[w * 4 + 0] = ts[0]; stamps[w * 4 + 1] = ts[1];
  }
  atomicAdd(&chk[w], acc);
}

__global__ void trivial_kernel(uint32_t* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

static double ms_between(hipEvent_t a, hipEvent_t b) { float ms; CHECK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 200;
  const int NWG = 256;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  uint32_t *err, *chk, *flags, *frow;
  uint64_t *cyc, *buf, *inbox, *stamps;
  CHECK(hipMalloc(&err, 64)); CHECK(hipMalloc(&chk, NWG * 4)); CHECK(hipMalloc(&cyc, NWG * 8));
  CHECK(hipMalloc(&buf, 2 * 8192 * 8)); CHECK(hipMalloc(&inbox, NR * 2 * NWG * 8)); CHECK(hipMalloc(&frow, NR * 256 * 4 + 64));
  CHECK(hipMalloc(&flags, 256)); CHECK(hipMalloc(&stamps, NWG * 4 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int lds = 140 * 1024;   // one workgroup per CU, like the engine
  std::vector<uint32_t> hchk(NWG); std::vector<uint64_t> hcyc(NWG);
  uint32_t herr = 0;

  auto report = [&](const char* name, double ms, uint32_t expect) {
    CHECK(hipMemcpy(hchk.data(), chk, NWG * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hcyc.data(), cyc, NWG * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    int bad = 0; uint64_t cmax = 0;
    for (int w = 0; w < NWG; ++w) { bad += hchk[w] != expect; cmax = hcyc[w] > cmax ? hcyc[w] : cmax; }
    printf("%-34s %8.3f us/iter (events)  %8.0f ticks/iter (s_memtime max)  err=0x%x  bad checksums=%d\n", name,
           ms * 1e3 / iters, (double)cmax / iters, herr, bad);
  };

  // ---- mode A
  for (int pairs = 2; pairs <= 4; pairs += 2) {
    const int T = 512 * 2 * pairs, per = T / NWG;
    uint32_t expect = 0;
    for (int it = 0; it < iters; ++it) for (int g = 0; g < T; ++g) expect += (uint32_t)g * 2654435761u + (uint32_t)it * 40503u + 17u;
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(buf, 0, 2 * 8192 * 8)); CHECK(hipMemset(err, 0, 64)); CHECK(hipMemset(chk, 0, NWG * 4));
      CHECK(hipEventRecord(e0));
      if (pairs == 2) { CHECK(hipFuncSetAttribute((const void*)allgather_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); allgather_kernel<2><<<NWG, 512, lds>>>(buf, per, iters, err, cyc, chk); }
      else { CHECK(hipFuncSetAttribute((const void*)allgather_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds)); allgather_kernel<4><<<NWG, 512, lds>>>(buf, per, iters, err, cyc, chk); }
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      char nm[64]; snprintf(nm, sizeof nm, "A all-gather %d granules (%d KB)", T, T * 8 / 1024);
      report(nm, ms_between(e0, e1), expect);
    }
  }
  // ---- mode C
  {
    // expected per-workgroup checksum: sum over rows r, columns c of (g0 + g1 values), every iteration
    uint32_t expect = 0;
    for (int it = 0; it < iters; ++it)
      for (int r = 0; r < NR; ++r)
        for (int c = 0; c < NWG; ++c)
          for (int m = 0; m < 2; ++m) expect += (uint32_t)((r * 2 + m) * NWG + c) * 2654435761u + (uint32_t)it * 40503u + 17u;
    CHECK(hipFuncSetAttribute((const void*)ffn_edge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipMemset(inbox, 0, NR * 2 * NWG * 8)); CHECK(hipMemset(flags, 0, 256)); CHECK(hipMemset(err, 0, 64));
      CHECK(hipMemset(chk, 0, NWG * 4));
      CHECK(hipEventRecord(e0));
      ffn_edge_kernel<<<NWG, 512, lds>>>(inbox, frow, flags, iters, err, cyc, chk, stamps);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      report("C ffn edge (256->43->256, 43 KB)", ms_between(e0, e1), expect);
      std::vector<uint64_t> st(NWG * 4);
      CHECK(hipMemcpy(st.data(), stamps, NWG * 4 * 8, hipMemcpyDeviceToHost));
      double a = 0, b = 0, c = 0;
      for (int w = 0; w < NWG; ++w) { a += st[w * 4]; b += st[w * 4 + 1]; c += st[w * 4 + 2]; }
      printf("    mean ticks/iter: publish+row work %.0f, flag wait %.0f, row read %.0f (100 MHz ticks: x10 = ns)\n",
             a / NWG / iters, b / NWG / iters, c / NWG / iters);
    }
  }

  // ---- mode A4
  {
    uint32_t* buf4; CHECK(hipMalloc(&buf4, 2 * 32768 * 4));
    auto runA4 = [&](auto kern, int loads) {
      const int N = 512 * 4 * loads, per = N / NWG;
      uint32_t expect = 0;
      for (int it = 0; it < iters; ++it) for (int g = 0; g < N; ++g) expect += ((uint32_t)g * 2654435761u + (uint32_t)it * 40503u + 17u) & ~15u;
      CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipMemset(buf4, 0, 2 * 32768 * 4)); CHECK(hipMemset(err, 0, 64)); CHECK(hipMemset(chk, 0, NWG * 4));
        CHECK(hipEventRecord(e0));
        kern<<<NWG, 512, lds>>>(buf4, per, iters, err, cyc, chk);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        char nm[64]; snprintf(nm, sizeof nm, "A4 all-gather %d dwords (%d KB)", N, N * 4 / 1024);
        report(nm, ms_between(e0, e1), expect);
      }
    };
    runA4(allgather4_kernel<2>, 2); runA4(allgather4_kernel<4>, 4); runA4(allgather4_kernel<6>, 6);
    runA4(allgather4_kernel<8>, 8); runA4(allgather4_kernel<12>, 12); runA4(allgather4_kernel<16>, 16);
  }
  // ---- mode C4
  {
    uint32_t *inbox4, *frow4;
    CHECK(hipMalloc(&inbox4, NR * 2 * NWG * 4)); CHECK(hipMalloc(&frow4, 2 * NR * 256 * 4 + 64));
    uint32_t expect = 0;
    for (int it = 0; it < iters; ++it)
      for (int r = 0; r < NR; ++r)
        for (int c = 0; c < NWG; ++c) {
          uint32_t s = 0;
          for (int m = 0; m < 2; ++m) s += ((uint32_t)((r * 2 + m) * NWG + c) * 2654435761u + (uint32_t)it * 40503u + 17u) & ~15u;
          expect += s & ~15u;
        }
    CHECK(hipFuncSetAttribute((const void*)ffn_edge4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int rep = 0; rep < 4; ++rep) {
      const int delay_iters = rep < 2 ? 0 : 40;
      CHECK(hipMemset(inbox4, 0, NR * 2 * NWG * 4)); CHECK(hipMemset(frow4, 0, 2 * NR * 256 * 4)); CHECK(hipMemset(err, 0, 64));
      CHECK(hipMemset(chk, 0, NWG * 4));
      CHECK(hipEventRecord(e0));
      ffn_edge4_kernel<<<NWG, 512, lds>>>(inbox4, frow4, iters, err, cyc, chk, stamps, 100, delay_iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      report(delay_iters ? "C4 ffn edge, one late workgroup" : "C4 ffn edge (self-tagged dwords)", ms_between(e0, e1), expect);
      std::vector<uint64_t> st(NWG * 4);
      CHECK(hipMemcpy(st.data(), stamps, NWG * 4 * 8, hipMemcpyDeviceToHost));
      double a = 0, b = 0;
      for (int w = 0; w < NWG; ++w) { a += st[w * 4]; b += st[w * 4 + 1]; }
      printf("    mean ticks/iter: publish + row work %.0f, row sweep %.0f\n", a / NWG / iters, b / NWG / iters);
    }
  }
  // ---- mode L: kernel boundaries
  {
    CHECK(hipMemset(chk, 0, 4));
    for (int rep = 0; rep < 2; ++rep) {
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) trivial_kernel<<<NWG, 512>>>(chk);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      printf("%-34s %8.3f us/iter (events)\n", "L dependent trivial launches", ms_between(e0, e1) * 1e3 / iters);
    }
  }
  return 0;
}
