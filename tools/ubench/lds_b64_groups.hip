// Which lanes of a wave does the LDS serve in the same cycle of a ds_read_b64?  The sign table of the E8P decode has 16 copies
// (entry idx of copy c at idx * 128 + 8 c: a copy owns one bank pair), and with copy = lane & 15 its look-ups show bank conflicts
// that 32 copies do not -- so the 16 lanes of a cycle are not 16 consecutive lanes.  This times a stream of
// random look-ups for several lane -> copy maps; a map whose 16 lanes of every cycle hit 16 different copies runs at the rate
// of 32 copies without their 32 KB.
//
// build: hipcc --offload-arch=gfx950 -O3 -o lds_b64_groups lds_b64_groups.hip      run: ./lds_b64_groups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__device__ __forceinline__ int copy_of(int map, int lane) {
  switch (map) {
    case 0: return lane & 15;                                   // today's
    case 1: return (lane & 7) | ((lane >> 1) & 8);              // bit 4 -> bit 3: lanes l, l + 8 share
    case 2: return (lane >> 1) & 15;                            // neighbours share: bits 1..4
    case 3: return (lane >> 2) & 15;                            // bits 2..5
    case 4: return (lane & 7) | ((lane >> 2) & 8);              // bit 5 -> bit 3: lanes l, l + 8 and l + 16 ... share
    case 5: return ((lane & 3) | ((lane >> 2) & 12));           // bits 0, 1, 4, 5
    case 6: return ((lane & 1) | ((lane >> 2) & 14));           // bits 0, 3, 4, 5
    case 7: return (lane & 31);                                 // 32 copies (the reference point; stride 256)
    default: return 0;
  }
}

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lookups(int map, int iters, uint64_t* cyc, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 1024 / 4; i += 64 * WAVES) reinterpret_cast<uint32_t*>(smem)[i] = (uint32_t)i * 2654435761u;
  __syncthreads();
  const int stride = map == 7 ? 256 : 128;
  const uint32_t base = (uint32_t)copy_of(map, lane) * 8u;
  uint32_t x = (uint32_t)tid * 747796405u + 2891336453u, acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    uint2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      x = x * 1664525u + 1013904223u;
      const uint32_t idx = (x >> 13) & 255u;
      v[j] = *reinterpret_cast<const uint2*>(smem + idx * stride + base);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j].x ^ v[j].y;
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x1234567u) sink[0] = acc;
}

int main() {
  uint64_t* cyc;
  uint32_t* sink;
  CHECK(hipMalloc(&cyc, 8 * 256));
  CHECK(hipMalloc(&sink, 4));
  const int iters = 2000;
  const char* names[8] = {"lane & 15 (today)", "bits 0-2, 4", "bits 1-4", "bits 2-5", "bits 0-2, 5", "bits 0, 1, 4, 5", "bits 0, 3, 4, 5", "32 copies (lane & 31)"};
  for (int waves = 1; waves <= 8; waves *= 8) {
    printf("%d wave(s) per workgroup, one workgroup per CU, %d x 8 random ds_read_b64 per lane: s_memtime ticks per look-up instruction\n", waves, iters);
    for (int map = 0; map < 8; ++map) {
      uint64_t best = ~0ull;
      for (int rep = 0; rep < 3; ++rep) {
        if (waves == 1) hipLaunchKernelGGL((lookups<1>), dim3(8), dim3(64), 64 * 1024, 0, map, iters, cyc, sink);
        else hipLaunchKernelGGL((lookups<8>), dim3(8), dim3(512), 64 * 1024, 0, map, iters, cyc, sink);
        CHECK(hipDeviceSynchronize());
        uint64_t h[8];
        CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i) if (h[i] < best) best = h[i];
      }
      printf("  copy = %-24s %8.2f\n", names[map], (double)best / (iters * 8.0));
    }
  }
  return 0;
}
