/*
 * quip_oracle.c -- plain-C restatement of the reference's QuantLinear eval forward for the
 * E8P12 codebook (CPU).  TEST INFRASTRUCTURE ONLY: it is the checker for the HIP path and the
 * timed `cpu_baseline` of bench.py; nothing in the product links or calls it.
 *
 * Follows (paths relative to the reference checkout):
 *   decode            codebook/e8p12.py:82-103 (get_full_grid) == quip_cuda/origin_order.cu:211-231
 *   Hadamard          quant.py:42-65 (matmul_hadU butterfly) and :72-88 (the (K, n/K) view, hadK)
 *   forward order     qlinear.py:87-115 (x*SU, hadUt * wscale, codebook mm, hadU, [:out], *SV, +bias)
 * Parity pin: tests/test_oracle_c.py checks every function against oracle/quip_oracle.py, which is
 * itself pinned to the reference-generated fixtures under tests/golden/.
 *
 * The reference has no CPU inference path (quantizer.py:799-801 raises without a GPU); its
 * "decompress + matmul" branch (e8p12.py:152-155) materialises W and calls a GEMM.  Here the
 * decode is fused into the row dot product and rows are spread over OpenMP threads, i.e. a
 * favourable CPU implementation: the baseline it produces is, if anything, too fast.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const int kByteOfPos[8] = {0, 2, 1, 3, 4, 6, 5, 7};

int quip_oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* 8 weights (times 4, as small integers) of one E8P12 code */
static inline void e8p_decode4(uint16_t code, const int64_t* grid, int out4[8]) {
  const unsigned sign = code & 0xff, absi = code >> 8;
  const int par = __builtin_popcount(sign) & 1;
  const unsigned sv = sign ^ (unsigned)par;
  const int64_t packed = grid[absi];
  for (int p = 0; p < 8; ++p) {
    const int b = kByteOfPos[p];
    int v = (int)(int8_t)((packed >> (8 * b)) & 0xff);
    if ((sv >> (7 - b)) & 1) v = -v;
    out4[p] = v + (par ? -1 : 1);
  }
}

/* dense decode: w (rows, cols*8) float = decoded weights */
void quip_oracle_decompress_e8p(const uint16_t* q, const int64_t* grid, float* w, long rows, long cols) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < rows; ++r)
    for (long c = 0; c < cols; ++c) {
      int d[8];
      e8p_decode4(q[r * cols + c], grid, d);
      for (int p = 0; p < 8; ++p) w[(r * cols + c) * 8 + p] = 0.25f * (float)d[p];
    }
}

/* in-place unnormalised Sylvester Walsh-Hadamard transform, n a power of two */
void quip_oracle_fwht(float* x, long n) {
  for (long h = 1; h < n; h <<= 1)
    for (long i = 0; i < n; i += 2 * h)
      for (long j = i; j < i + h; ++j) {
        const float a = x[j], b = x[j + h];
        x[j] = a + b;
        x[j + h] = a - b;
      }
}

/* y = scale * (hadK (x) H_{n/K}) x on the row-major (K, n/K) view; hadK NULL when K == 1 */
void quip_oracle_hadU(const float* x, float* y, long n, long K, const float* hadK, int transpose, float scale) {
  const long L = n / K;
  float* t = (float*)malloc(sizeof(float) * (size_t)n);
  memcpy(t, x, sizeof(float) * (size_t)n);
  for (long k = 0; k < K; ++k) quip_oracle_fwht(t + k * L, L);
  if (K == 1) {
    for (long i = 0; i < n; ++i) y[i] = scale * t[i];
  } else {
    for (long kp = 0; kp < K; ++kp)
      for (long j = 0; j < L; ++j) {
        double acc = 0.0;
        for (long k = 0; k < K; ++k) acc += (double)(transpose ? hadK[k * K + kp] : hadK[kp * K + k]) * t[k * L + j];
        y[kp * L + j] = scale * (float)acc;
      }
  }
  free(t);
}

/* z[n] = sum_k decode(q)[n,k] * x[k]  (fp32 accumulation) */
void quip_oracle_e8p_gemv(const uint16_t* q, const int64_t* grid, const float* x, float* z, long n, long k) {
  const long cols = k / 8;
#pragma omp parallel for schedule(static)
  for (long r = 0; r < n; ++r) {
    float acc = 0.f;
    const uint16_t* qr = q + r * cols;
    for (long c = 0; c < cols; ++c) {
      int d[8];
      e8p_decode4(qr[c], grid, d);
      const float* xc = x + c * 8;
      float s = 0.f;
      for (int p = 0; p < 8; ++p) s += (float)d[p] * xc[p];
      acc += 0.25f * s;
    }
    z[r] = acc;
  }
}

/* QuantLinear.forward, eval branch, one token row (all vectors float; SU/SV/bias may be NULL) */
void quip_oracle_qlinear_e8p(const float* x, float* y, long in_f, long out_f, long q_in, long q_out,
                             const uint16_t* q, const int64_t* grid, const float* SU, const float* SV,
                             const float* bias, float wscale, long K_left, const float* had_left,
                             long K_right, const float* had_right) {
  float* a = (float*)calloc((size_t)q_in, sizeof(float));
  float* b = (float*)malloc(sizeof(float) * (size_t)q_in);
  float* z = (float*)malloc(sizeof(float) * (size_t)q_out);
  float* u = (float*)malloc(sizeof(float) * (size_t)q_out);
  for (long i = 0; i < in_f; ++i) a[i] = SU ? x[i] * SU[i] : x[i];
  quip_oracle_hadU(a, b, q_in, K_left, had_left, 1, wscale / sqrtf((float)(q_in / K_left)));
  quip_oracle_e8p_gemv(q, grid, b, z, q_out, q_in);
  quip_oracle_hadU(z, u, q_out, K_right, had_right, 0, 1.0f / sqrtf((float)(q_out / K_right)));
  for (long i = 0; i < out_f; ++i) y[i] = (SV ? u[i] * SV[i] : u[i]) + (bias ? bias[i] : 0.f);
  free(a); free(b); free(z); free(u);
}

/* fp16-weight dense GEMV stand-in for the lm_head (weights given as float) */
void quip_oracle_dense_gemv(const float* w, const float* x, float* y, long n, long k) {
#pragma omp parallel for schedule(static)
  for (long r = 0; r < n; ++r) {
    float acc = 0.f;
    for (long c = 0; c < k; ++c) acc += w[r * k + c] * x[c];
    y[r] = acc;
  }
}
