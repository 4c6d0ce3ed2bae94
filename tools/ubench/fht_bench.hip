// Clocks per call of the 512-thread transforms of the persistent decode engines, in the engines' geometry (256 workgroups x 512
// threads, one per CU by LDS footprint), data dependent from call to call.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iquip_for_all_amd/csrc -Iinclude -o tools/ubench/fht_bench tools/ubench/fht_bench.hip
#include "fht_wg512x.hip.h"
#include <cstdio>
#include <vector>
using namespace quip;

template <int VAR>
__global__ __launch_bounds__(512) void bench(float* out, unsigned long long* t, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xbuf = reinterpret_cast<float*>(smem);
  const int tid = threadIdx.x;
  float v[2][16];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[i][r] = (float)((tid * 16 + r + i) % 7) - 3.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (VAR == 0) { float (&w)[1][16] = *reinterpret_cast<float (*)[1][16]>(&v[0]); hadw::fwd<13, 1, true>(w, xbuf, tid); }
    if constexpr (VAR == 1) { float (&w)[1][16] = *reinterpret_cast<float (*)[1][16]>(&v[0]); hadw::rev<13, 1, true>(w, xbuf, tid); }
    if constexpr (VAR == 2) hadw::rev<13, 2, true>(v, xbuf, tid);
    if constexpr (VAR == 3) hadw::fwd<13, 2, true>(v, xbuf, tid);
    if constexpr (VAR == 4) { float w[1][8]; for (int r = 0; r < 8; ++r) w[0][r] = v[0][r]; had8::fht4096<1, true>(w, xbuf, tid); for (int r = 0; r < 8; ++r) v[0][r] = w[0][r]; }
    if constexpr (VAR == 5) { float w[1][8]; for (int r = 0; r < 8; ++r) w[0][r] = v[0][r]; hadw::fwd<12, 1, true>(w, xbuf, tid); for (int r = 0; r < 8; ++r) v[0][r] = w[0][r]; }
    if constexpr (VAR == 6) hadw::wave_fht1024(v[0], tid & 63);
#pragma unroll
    for (int r = 0; r < 16; ++r) { v[0][r] *= 0.01f; v[1][r] *= 0.01f; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) t[blockIdx.x] = t1 - t0;
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += v[0][r] + v[1][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int VAR>
void run(const char* name, float* out, unsigned long long* t) {
  const int iters = 200, lds = 150 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(bench<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 2; ++rep) bench<VAR><<<256, 512, lds>>>(out, t, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), t, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto x : h) s += (double)x;
  printf("%-34s %8.0f ticks per call\n", name, s / 256 / iters);
}

int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 8);
  run<0>("fwd 8192 x1", out, t); run<1>("rev 8192 x1", out, t); run<2>("rev 8192 x2", out, t); run<3>("fwd 8192 x2", out, t);
  run<4>("had8::fht4096 x1", out, t); run<5>("fwd 4096 x1", out, t); run<6>("wave_fht1024 (every wave)", out, t);
  return 0;
}
