// Quantise-time nearest-codeword search in the E8P12 codebook.
//
// Replaces E8P12_codebook.round / quantize (e8p12.py:125-137; also the first stage of
// e8p12_rvq3.py:81-92 and both stages of e8p12_rvq4.py:32-45):
//     idx = argmax_c ( 2 x . g_c - |g_c|^2 )  over the 65 536 codewords,   vals = g_idx
// which the reference evaluates as a dense (N, 8) x (8, 65 536) GEMM per LDLQ step (quant.py:103-135).
//
// The codebook is structured (origin_order.cu:211-253 read backwards): with b_e the abs-table row
// (eight half-integers, column 7 negative when the row's sum is odd, e8p12.py:72-78),
//     g = D b_e + s 1,   D = diag(+-1) with an EVEN number of -1,   s = +1/4 (par = 0) or -1/4 (par = 1),
//     code = e << 8 | (sv ^ par),  bit (7 - [0,2,1,3,4,6,5,7][j]) of sv <=> D_j = -1.
// For fixed (e, s), with y = x - s 1:
//     2 x.g - |g|^2 = 2 sum_j D_j b_j y_j + 2 s sum_j x_j - |b_e|^2 - 8 s^2
// is maximised by D_j = sign(b_j y_j); if that takes an odd number of -1 the cheapest repair flips the
// column with the smallest |b_j y_j|.  So the exact arg max needs 2 x 256 candidates of ~25 operations
// instead of 65 536 dot products: kLanes = 8 lanes per vector, each scanning 32 table rows (table in LDS),
// then an 8-lane shuffle arg max -- ~14 K operations per vector, and an LDLQ step (N = out_features vectors,
// issued thousands of times in sequence) has the latency of 64 candidates, not of a GEMM.
//
// Ties (a measure-zero set: some b_j y_j == 0, or two candidates with equal score) resolve to the
// lowest (e, par, column) in scan order, which need not be the lowest code; near-ties inside fp32
// rounding of the score may resolve differently from a GEMM-based arg max.  Either way the returned point
// is a nearest codeword to within that rounding (tests/test_gpu_quantize.py states the bound).
#include <hip/hip_runtime.h>

#include "quip_device.hip.h"
#include "quip_internal.h"

namespace quip {

namespace {

template <int kLanes>
__global__ __launch_bounds__(256) void e8p_quantize_kernel(const float* __restrict__ x, int64_t nvec,
                                                          const uint64_t* __restrict__ grid_packed_abs,
                                                          float* __restrict__ vals, int64_t* __restrict__ idx) {
  __shared__ float tb[256][8];    // |b_e|, natural column order
  __shared__ float tn[256];       // |b_e|^2
  __shared__ int tneg[256];       // column 7 of b_e is negative
  {
    const int e = threadIdx.x;
    const uint64_t p = grid_packed_abs[e];
    float n2 = 0.f;
    int neg = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // byte j of the packed entry = 4 * value of column [0,2,1,3,4,6,5,7][j] (self-inverse permutation)
      const int col = e8p_byte_of_pos(j);
      const int v4 = (int)(int8_t)((p >> (8 * j)) & 0xff);
      const float v = 0.25f * (float)v4;
      tb[e][col] = fabsf(v);
      if (col == 7 && v4 < 0) neg = 1;
      n2 += v * v;
    }
    tn[e] = n2;
    tneg[e] = neg;
  }
  __syncthreads();
  const int sub = threadIdx.x & (kLanes - 1);
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kLanes;
  const bool valid = i < nvec;
  i = valid ? i : nvec - 1;     // keep every lane in the shuffles below
  float xv[8];
  {
    const float4 a = reinterpret_cast<const float4*>(x + i * 8)[0];
    const float4 b = reinterpret_cast<const float4*>(x + i * 8)[1];
    xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; xv[4] = b.x; xv[5] = b.y; xv[6] = b.z; xv[7] = b.w;
  }
  float ay[2][8];     // |x_j - s| for s = +1/4, -1/4
  int ypar[2];        // parity of the number of negative (x_j - s)
  float cs[2];        // 2 s sum x - 8 s^2
  float sx = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) sx += xv[j];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const float s = p ? -0.25f : 0.25f;
    int par = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float y = xv[j] - s;
      ay[p][j] = fabsf(y);
      par ^= (y < 0.f) ? 1 : 0;
    }
    ypar[p] = par;
    cs[p] = 2.f * s * sx - 0.5f;
  }
  float best = -3.0e38f;
  int best_e = 0, best_p = 0;
  for (int e = sub * (256 / kLanes); e < (sub + 1) * (256 / kLanes); ++e) {
    float b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = tb[e][j];
    const float n2 = tn[e];
    const int neg = tneg[e];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      float sum = 0.f, mn = 3.0e38f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = b[j] * ay[p][j];
        sum += t;
        mn = fminf(mn, t);
      }
      // number of -1 in D relative to b_e (whose column 7 may itself be negative): negatives of y, plus one
      // if b_7 < 0; odd -> give up the cheapest column
      const int odd = ypar[p] ^ neg;
      const float score = 2.f * (sum - (odd ? 2.f * mn : 0.f)) + cs[p] - n2;
      if (score > best) { best = score; best_e = e; best_p = p; }
    }
  }
  // arg max over the vector's kLanes lanes: higher score, then the earlier (e, par) like a sequential scan
  int key = best_e * 2 + best_p;
#pragma unroll
  for (int off = 1; off < kLanes; off <<= 1) {
    const float os = __shfl_xor(best, off, 64);
    const int ok = __shfl_xor(key, off, 64);
    if (os > best || (os == best && ok < key)) { best = os; key = ok; }
  }
  best_e = key >> 1;
  best_p = key & 1;
  if (!valid || sub != 0) return;
  // rebuild the winner: signs, the repaired column, the code and the values
  const float s = best_p ? -0.25f : 0.25f;
  float b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) b[j] = tb[best_e][j];
  const int neg7 = tneg[best_e];
  int flip[8];     // D_j == -1 relative to the SIGNED b_e
  float mn = 3.0e38f;
  int jm = 0, cnt = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float y = xv[j] - s;
    const int want_neg = y < 0.f ? 1 : 0;                 // sign of the codeword's (g_j - s)
    const int base_neg = (j == 7 && neg7) ? 1 : 0;        // sign of b_e's column
    flip[j] = want_neg ^ base_neg;
    cnt += flip[j];
    const float t = b[j] * fabsf(y);
    if (t < mn) { mn = t; jm = j; }
  }
  if (cnt & 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j == jm) flip[j] ^= 1;
  }
  int sv = 0;
  float out[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sv |= flip[j] << (7 - e8p_byte_of_pos(j));   // the sign bit of column j is the one of its packed byte
    const float signed_b = ((j == 7 && neg7) ? -b[j] : b[j]);
    out[j] = (flip[j] ? -signed_b : signed_b) + s;
  }
  idx[i] = (int64_t)((best_e << 8) | (sv ^ best_p));
  reinterpret_cast<float4*>(vals + i * 8)[0] = make_float4(out[0], out[1], out[2], out[3]);
  reinterpret_cast<float4*>(vals + i * 8)[1] = make_float4(out[4], out[5], out[6], out[7]);
}

}  // namespace

int e8p_quantize_launch(const void* x, int64_t nvec, const void* grid_packed_abs, void* vals, void* idx,
                        hipStream_t stream) {
  if (nvec <= 0) return QUIP_OK;
  const int threads = 256;   // == table rows: thread e builds row e
  // LDLQ steps (N = out_features, one after the other): 8 lanes per vector for latency; big batches: one
  // lane per vector for throughput (measured 262 144 vectors: 97 us vs 178 us; 4096: 43 vs 16 us)
  const int lanes = nvec < 65536 ? 8 : 1;
  const int64_t blocks = (nvec * lanes + threads - 1) / threads;
  if (blocks > 0x7fffffff) return QUIP_ERR_BAD_SHAPE;
  if (lanes == 8)
    hipLaunchKernelGGL(e8p_quantize_kernel<8>, dim3((unsigned)blocks), dim3(threads), 0, stream,
                       reinterpret_cast<const float*>(x), nvec, reinterpret_cast<const uint64_t*>(grid_packed_abs),
                       reinterpret_cast<float*>(vals), reinterpret_cast<int64_t*>(idx));
  else
    hipLaunchKernelGGL(e8p_quantize_kernel<1>, dim3((unsigned)blocks), dim3(threads), 0, stream,
                       reinterpret_cast<const float*>(x), nvec, reinterpret_cast<const uint64_t*>(grid_packed_abs),
                       reinterpret_cast<float*>(vals), reinterpret_cast<int64_t*>(idx));
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

}  // namespace quip
