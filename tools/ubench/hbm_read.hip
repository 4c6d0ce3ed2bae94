// What a pure read stream reaches on this box, next to the request pattern the E8P GEMV's B operand dictates -- the
// yardstick for roofline.frac of the decode launches (the 8192-wide launch's products sustain 0.61-0.63 of the
// 8 TB/s peak; its ring alone, decode and MFMAs taken out, 0.72).
//
//   mode L  full lines: a wave's load instruction covers 1 KB of CONSECUTIVE bytes (64 lanes x 16 B)
//   mode R  the GEMV's pattern: a load instruction covers 16 rows x 64 B (lane = 16 q + n reads 16 B at row n, byte 16 q of
//           the row's current 64-byte piece; the second instruction of an item the next 64 B: e8p_gemv_core.hip.h)
//   both persistent (256 workgroups x 512 threads, as the launches), DEPTH instructions per wave in flight, `nt` loads,
//   buffer >> the 256 MB of MALL so that every byte comes from HBM.
//
// build: hipcc --offload-arch=gfx950 -O3 -o hbm_read hbm_read.hip      run: ./hbm_read [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, bool ROWS>
__global__ __launch_bounds__(512) void read_kernel(const uint4* __restrict__ src, size_t bytes, size_t row_bytes, uint32_t* out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t nwaves = (size_t)gridDim.x * 8, wid = (size_t)blockIdx.x * 8 + wave;
  uint32_t acc = 0;
  u32x4 v[DEPTH];
  if constexpr (!ROWS) {
    // wave `wid` reads the 1 KB chunks wid, wid + nwaves, ...
    const size_t chunks = bytes / 1024;
    const char* base = reinterpret_cast<const char*>(src) + (size_t)lane * 16;
    for (size_t c = wid; c + (DEPTH - 1) * nwaves < chunks; c += DEPTH * nwaves) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* p = base + (c + d * nwaves) * 1024;
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[d]) : "v"(p) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) { asm volatile("" : "+v"(v[d])); acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w; }
    }
  } else {
    // a "matrix" of rows of row_bytes: a workgroup owns 16 consecutive rows at a time (a row block), its wave w the row's
    // bytes [w * row_bytes / 8, +row_bytes / 8) in 64-byte pieces: instruction = 16 rows x 64 B
    const size_t rows = bytes / row_bytes, blocks = rows / 16, slice = row_bytes / 8, pieces = slice / 64;
    const int n = lane & 15, q = lane >> 4;
    const size_t nb = (blocks - blockIdx.x + gridDim.x - 1) / gridDim.x, total = nb * pieces;      // (block, piece) pairs of this workgroup
    const char* base = reinterpret_cast<const char*>(src) + (size_t)n * row_bytes + (size_t)wave * slice + q * 16;
    for (size_t i = 0; i + DEPTH <= total; i += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const size_t b = blockIdx.x + ((i + d) / pieces) * gridDim.x, pc = (i + d) % pieces;
        const char* p = base + b * 16 * row_bytes + pc * 64;
        asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[d]) : "v"(p) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) { asm volatile("" : "+v"(v[d])); acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w; }
    }
  }
  if (acc == 0x12345678u) out[0] = acc;      // (keeps the loads)
}

template <int DEPTH, bool ROWS>
static void run(const uint4* src, size_t bytes, size_t row_bytes, uint32_t* out, const char* name) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int it = 0; it < 5; ++it) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((read_kernel<DEPTH, ROWS>), dim3(256), dim3(512), 0, 0, src, bytes, row_bytes, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (it > 0 && ms < best) best = ms;
  }
  const double tbs = (double)bytes / (best * 1e-3) / 1e12;
  printf("%-44s depth %2d: %8.3f ms  %6.3f TB/s  %.3f of 8 TB/s\n", name, DEPTH, best, tbs, tbs / 8.0);
}

int main(int argc, char** argv) {
  const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
  const size_t bytes = gib << 30;
  uint4* src;
  uint32_t* out;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(src, 1, bytes));
  CHECK(hipDeviceSynchronize());
  printf("pure read streams over %zu GiB, 256 x 512 threads persistent (hipEvent, best of 4 after one warm-up)\n", gib);
  run<4, false>(src, bytes, 0, out, "full lines (1 KB consecutive / instruction)");
  run<8, false>(src, bytes, 0, out, "full lines (1 KB consecutive / instruction)");
  run<18, false>(src, bytes, 0, out, "full lines (1 KB consecutive / instruction)");
  // rows of 2048 B = the codes of a K = 8192 row (2 bits per weight); 7168 B = K = 28672
  run<8, true>(src, bytes, 2048, out, "GEMV pattern, K = 8192 (16 rows x 64 B)");
  run<18, true>(src, bytes, 2048, out, "GEMV pattern, K = 8192 (16 rows x 64 B)");
  run<14, true>(src, bytes, 7168, out, "GEMV pattern, K = 28672 (16 rows x 64 B)");
  CHECK(hipFree(src));
  CHECK(hipFree(out));
  return 0;
}
