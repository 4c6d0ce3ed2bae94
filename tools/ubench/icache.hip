// Instruction-fetch probe for gfx950: a straight-line body of N dependent-free VALU instructions (4 bytes each) executed in a
// loop by 256 x 512 threads (the persistent decode engine's geometry: two waves per SIMD, every CU busy).  Reports cycles per
// instruction for bodies of 16 KB .. 256 KB: what a kernel whose per-block code exceeds the instruction cache pays.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/icache tools/ubench/icache.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KB>
__global__ __launch_bounds__(512) void body(float* out, unsigned long long* t, int iters) {
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, b = 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    // four independent accumulators: no dependency stalls, 4 bytes per instruction
    asm volatile(".rept %5\n\tv_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t.endr"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "n"(KB * 1024 / 16));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 512 + threadIdx.x] = a0 + a1 + a2 + a3;
}

template <int KB>
void run(float* out, unsigned long long* t, int iters) {
  body<KB><<<256, 512>>>(out, t, iters);
  body<KB><<<256, 512>>>(out, t, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), t, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto x : h) s += (double)x;
  const double per = s / 256 / iters / (KB * 1024 / 4);
  printf("body %4d KB: %.3f ticks per instruction per wave-pair-slot (%.0f ticks per pass)\n", KB, per, s / 256 / iters);
}

int main() {
  float* out; unsigned long long* t;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256 * 8);
  run<16>(out, t, 64); run<32>(out, t, 64); run<48>(out, t, 64); run<64>(out, t, 64); run<80>(out, t, 32); run<96>(out, t, 32); run<112>(out, t, 32);

  return 0;
}
