/*
 * quip_mi355.h -- C ABI of the MI355X-native QuIP# inference hot path.
 *
 * This is the drop-in boundary: one `extern "C"` entry point per native function
 * the reference binds for this path.  The reference's native boundary is the
 * pybind11 module `quiptools_cuda` (quip_cuda/quiptools_wrapper.cpp:87-100) plus
 * the third-party `fast_hadamard_transform_cuda.fast_hadamard_transform`
 * (register_lib.py:5,18-20); its Python operator boundary is the `quip_lib`
 * torch.library namespace (register_lib.py:8-192).
 *
 * Contract (differs from the reference where the reference had none):
 *   - plain pointers + sizes, no torch / ATen types; every pointer is a DEVICE
 *     pointer to a contiguous row-major buffer owned by the caller;
 *   - the library allocates nothing and keeps no state: safe under hipGraph
 *     capture and torch.compile; re-entrant; callable with or without the GIL;
 *   - `stream` is a hipStream_t (NULL = the legacy default stream); every call
 *     is asynchronous and stream ordered, no host synchronisation;
 *   - returns 0 on success or a negative QUIP_ERR_* code; never aborts
 *     (the reference only had `assert`s, compiled out in release builds:
 *     origin_order.cu:816-822, 868-874).
 *
 * Shapes use the reference's names: x is (m, k) fp16, Qidxs is (n, k/codesz/packsz)
 * in the codebook's index dtype, the product is (m, n) fp16 with
 * y[i, j] = sum_t x[i, t] * W[j, t], W = decode(Qidxs) (n, k), fp32 accumulation,
 * round-to-nearest fp16 store (origin_order.cu:388-555).
 */
#ifndef QUIP_MI355_H_
#define QUIP_MI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QUIP_ABI_VERSION 8

typedef void* quip_stream_t; /* hipStream_t */

enum {
  QUIP_OK = 0,
  QUIP_ERR_NULL_POINTER = -1,
  QUIP_ERR_BAD_SHAPE = -2,   /* dimension not supported by the packed format   */
  QUIP_ERR_MISALIGNED = -3,  /* pointer not aligned for vector access (16 B)    */
  QUIP_ERR_LAUNCH = -4,      /* HIP launch error; see hipGetLastError           */
  QUIP_ERR_UNSUPPORTED = -5, /* valid request this build cannot serve           */
  /* positive: the launch ran, but NOT as the operator the entry point stands for.  quip_block_engine with shape 1 and
   * dbg_layer == -2 (measurement mode: the products of every block without edges, attention and hand-offs) answers this instead
   * of QUIP_OK, so that a caller who checks `!= 0` never takes the contents of h_out for a hidden state. */
  QUIP_NO_RESULT = 1
};

int quip_abi_version(void);
const char* quip_strerror(int code);

/* Number of compute units of the current device (used by callers to size
 * rotating buffers); <0 on error. */
int quip_device_cu_count(void);

/* ---- quip_lib::hadamard ----------------------------------------------------
 * y[r, :] = scale * H_n x[r, :], H_n the Sylvester-ordered +-1 Walsh-Hadamard
 * matrix, n a power of two (<= 32768).  fp16 in / fp16 out, fp32 inside.
 * Replaces fast_hadamard_transform_cuda.fast_hadamard_transform
 * (register_lib.py:18-20; call sites quant.py:78,82).  x and y may alias. */
int quip_hadamard_f16(const void* x, void* y, int64_t rows, int32_t n, float scale,
                      quip_stream_t stream);
/* The same transform for every dtype the reference's op accepts (fast_hadamard_transform takes fp16 / bf16 / fp32,
 * register_lib.py:10-20; fp32 inside, one rounding to the I/O type).  dtype: QUIP_DTYPE_*.  fp16 goes to
 * quip_hadamard_f16's kernels. */
enum { QUIP_DTYPE_F16 = 0, QUIP_DTYPE_BF16 = 1, QUIP_DTYPE_F32 = 2 };
int quip_hadamard(const void* x, void* y, int64_t rows, int32_t n, float scale, int32_t dtype,
                  quip_stream_t stream);

/* ---- quip_lib::*_mm_origorder ---------------------------------------------
 * Replace quiptools_cuda.{e8p,e8prvq3,e8prvq4,d4,hi}_mm_origorder
 * (quiptools_wrapper.cpp:88-92; origin_order.cu:557-788).  The reference
 * allocates and returns C; here the caller passes y (m, n) fp16.
 * Requirements: k % 8 == 0 (E8P*, HI), k % 32 == 0 (E8P12RVQ3B), k % 4 == 0 (D4);
 * m >= 1 (any m is accepted; the codebook modules call this for m < 32). */
int quip_e8p_mm_origorder(const void* x, const void* qidxs /* int16 (n, k/8) */,
                          const void* grid_packed_abs /* int64[256] */, void* y,
                          int32_t m, int32_t n, int32_t k, quip_stream_t stream);
/* Skinny E8P12 product, up to 32 rows in ONE pass over the codes (the 1 < M < 32 use of the reference's
 * tinygemm_m16n8k16_chunk_kernel, origin_order.cu:388-555; e8p12.py:147-150) with the reference's arithmetic: fp16
 * activations x exact fp16 weights, fp32 accumulation on the matrix cores, one fp16 rounding.  m > 32: chunks of 32
 * rows in the same launch (one pass over the codes per chunk; faster than quip_e8p_mm_batched up to m * n of about
 * 1.8e6).  Not bit identical to the exact integer path of quip_e8p_mm_origorder_ws / quip_e8p_gemv_planes_rows (which
 * stays available); a row's result does not depend on the other rows of the batch.  k % 128 == 0, n % 2 == 0, else
 * QUIP_ERR_UNSUPPORTED. */
int quip_e8p_mm_skinny(const void* x, const void* qidxs /* int16 (n, k/8) */, const void* grid_packed_abs, void* y,
                       int32_t m, int32_t n, int32_t k, quip_stream_t stream);
/* The same for E8P12RVQ4B (replaces the 1 < M < 32 use of tinygemm_m16n8k16_chunk_kernel<.., BLayout_E8RVQ4, ..>,
 * origin_order.cu:337-385, 388-555; e8p12_rvq4.py:47-63): qidxs int32 (n, k/8), code = main << 16 | residual, weight =
 * fma(resid_scale, w_residual, w_main) rounded once to fp16 -- exactly the dense W of quip_decompress_e8prvq4_origorder
 * -- fp16 activations, fp32 accumulation.  Same shape rules. */
int quip_e8prvq4_mm_skinny(const void* x, const void* qidxs /* int32 (n, k/8) */, const void* grid_packed_abs,
                           float resid_scale, void* y, int32_t m, int32_t n, int32_t k, quip_stream_t stream);
/* ... for E8P12RVQ3B (BLayout_E8RVQ3, origin_order.cu:287-335; e8p12_rvq3.py:109-129): qidxs = the checkpoint's packed 3-byte
 * codes (n, 3 k / 8 bytes; k % 32 == 0), e81b_packed = the uint32 [256] residual table as the reference passes it. */
int quip_e8prvq3_mm_skinny(const void* x, const void* qidxs, const void* grid_packed_abs, const void* e81b_packed,
                           float resid_scale, void* y, int32_t m, int32_t n, int32_t k, quip_stream_t stream);
/* ... for D4 (BLayout_D4; d4.py:134-151): qidxs uint8 (n, k/4), grid_f16 = the fp16 (256, 4) table as the reference holds it;
 * and for HI (BLayout_HI; hi.py:52-66): qidxs int32 (n, k/8), eight nibbles per code, w = nibble - 7.5. */
int quip_d4_mm_skinny(const void* x, const void* qidxs, const void* grid_f16, void* y, int32_t m, int32_t n, int32_t k,
                      quip_stream_t stream);
int quip_hi_mm_skinny(const void* x, const void* qidxs, void* y, int32_t m, int32_t n, int32_t k, quip_stream_t stream);
/* Batched E8P12 product for M >= 32 (prompt prefill): fused dequant + MFMA GEMM, y (m, n) = x (m, k) @ W^T with
 * W = decode(qidxs), fp16 in / fp32 accumulation / fp16 out -- the arithmetic of the reference's M >= 32 path
 * (e8p12.py:152-155: decompress_e8p_origorder + `x @ W.T`; origin_order.cu:837-885) without ever materialising W.
 * Any m >= 1 is computed correctly; k % 64 == 0, n % 2 == 0.  QUIP_ERR_UNSUPPORTED otherwise (callers then use
 * quip_decompress_e8p_origorder + a dense GEMM). */
int quip_e8p_mm_batched(const void* x, const void* qidxs /* int16 (n, k/8) */, const void* grid_packed_abs,
                        void* y, int64_t m, int32_t n, int32_t k, quip_stream_t stream);
/* The same fused tile kernel with the other codebooks' decode -- what the reference serves at M >= 32 with
 * decompress_* + `x @ W.T` (e8p12_rvq4.py:50-67, e8p12_rvq3.py:109-129, d4.py:128-139, hi.py:52-63), computed on the
 * exact fp16 weights those decompress ops write (E8P12RVQ4B / 3B: fma(resid_scale, w_residual, w_main), one fp16 rounding
 * per weight, origin_order.cu:330-331,378-380), without materialising W.  Code layouts and tables as for the *_mm_skinny
 * entry points above; same shape rules as quip_e8p_mm_batched. */
int quip_e8prvq4_mm_batched(const void* x, const void* qidxs /* int32 (n, k/8) */, const void* grid_packed_abs,
                            float resid_scale, void* y, int64_t m, int32_t n, int32_t k, quip_stream_t stream);
int quip_e8prvq3_mm_batched(const void* x, const void* qidxs /* 3 k / 8 bytes per row */, const void* grid_packed_abs,
                            const void* e81b_packed, float resid_scale, void* y, int64_t m, int32_t n, int32_t k,
                            quip_stream_t stream);
int quip_d4_mm_batched(const void* x, const void* qidxs /* uint8 (n, k/4) */, const void* grid_f16, void* y, int64_t m,
                       int32_t n, int32_t k, quip_stream_t stream);
int quip_hi_mm_batched(const void* x, const void* qidxs /* int32 (n, k/8) */, void* y, int64_t m, int32_t n, int32_t k,
                       quip_stream_t stream);
/* Workspace variant of the E8P12 product.  For 1 <= m < 32 (the range the codebook module sends to this
 * op, e8p12.py:147-150) the fast path first rewrites every row of x as block fixed point int8 digit planes
 * (one small launch, one workgroup per row) and then runs the integer-domain matrix-core GEMV -- m == 1:
 * the bs=1 kernel; m > 1: rows mode, passes of quip_e8p_gemv_max_rows(n, k) rows over the codes (see
 * quip_e8p_gemv_planes_rows).  The planes live in caller-provided scratch (m plane images), so the library
 * still allocates nothing.  quip_e8p_mm_origorder() without workspace stays valid (m == 1: converts x
 * inside every workgroup; m > 1: the generic fused decode + fp32 FMA kernel): self-contained but slower.
 * m >= 32 or unsupported shapes: workspace unused (bytes == 0).
 * A row of x with an inf or a NaN gives a NaN output row (shift word 1 << 20 in its plane image), never zeros. */
size_t quip_e8p_mm_workspace_bytes(int32_t m, int32_t n, int32_t k);
int quip_e8p_mm_origorder_ws(const void* x, const void* qidxs, const void* grid_packed_abs, void* y,
                             int32_t m, int32_t n, int32_t k, void* workspace,
                             size_t workspace_bytes, quip_stream_t stream);
/* The two halves of the above, for callers that produce the planes themselves (the fused
 * Hadamard kernel does): planes buffer = quip_e8p_planes_bytes(k) bytes, 16-byte aligned;
 * layout documented in csrc/e8p_gemv_i8.hip.  k % 64 == 0. */
size_t quip_e8p_planes_bytes(int32_t k);
int quip_e8p_x_to_planes(const void* x /* fp16[k] */, void* planes, int32_t k, quip_stream_t stream);
int quip_e8p_gemv_planes(const void* planes, const void* qidxs, const void* grid_packed_abs,
                         void* y /* fp16[n] */, int32_t n, int32_t k, quip_stream_t stream);

int quip_e8prvq3_mm_origorder(const void* x, const void* qidxs /* int32 (n, k*3/32) */,
                              const void* grid_packed_abs /* int64[256] */,
                              const void* e81b_grid_packed /* int32[256] */, float scale,
                              void* y, int32_t m, int32_t n, int32_t k,
                              quip_stream_t stream);
int quip_e8prvq4_mm_origorder(const void* x, const void* qidxs /* int32 (n, k/8) */,
                              const void* grid_packed_abs, float scale, void* y,
                              int32_t m, int32_t n, int32_t k, quip_stream_t stream);
int quip_d4_mm_origorder(const void* x, const void* qidxs /* uint8 (n, k/4) */,
                         const void* grid_f16 /* fp16[256*4] */, void* y, int32_t m,
                         int32_t n, int32_t k, quip_stream_t stream);
int quip_hi_mm_origorder(const void* x, const void* qidxs /* int32 (n, k/8) */, void* y,
                         int32_t m, int32_t n, int32_t k, quip_stream_t stream);

/* ---- quip_lib::decompress_*_origorder -------------------------------------
 * Replace quiptools_cuda.decompress_{e8p,e8prvq3,e8prvq4,d4,hi}_origorder
 * (quiptools_wrapper.cpp:93-97; origin_order.cu:794-1074): dense fp16
 * W (rows, cols_out) written into caller-allocated `w`, `cols_out` = number of
 * weights per row (k). */
int quip_decompress_e8p_origorder(const void* qidxs, const void* grid_packed_abs, void* w,
                                  int64_t rows, int32_t k, quip_stream_t stream);
int quip_decompress_e8prvq3_origorder(const void* qidxs, const void* grid_packed_abs,
                                      const void* e81b_grid_packed, float scale, void* w,
                                      int64_t rows, int32_t k, quip_stream_t stream);
int quip_decompress_e8prvq4_origorder(const void* qidxs, const void* grid_packed_abs,
                                      float scale, void* w, int64_t rows, int32_t k,
                                      quip_stream_t stream);
int quip_decompress_d4_origorder(const void* qidxs, const void* grid_f16, void* w,
                                 int64_t rows, int32_t k, quip_stream_t stream);
int quip_decompress_hi_origorder(const void* qidxs, void* w, int64_t rows, int32_t k,
                                 quip_stream_t stream);

/* ---- fused pieces of QuantLinear.forward (qlinear.py:87-115) ----------------
 * Input side:  xh = wscale * U_{q_in}(had_left^T) (SU (.) x)      (qlinear.py:90-100,
 *              quant.py:72-88 with transpose=True)
 * Output side: y  = SV (.) U_{q_out}(had_right)(Wscale_vec (.) z)[:out] + bias
 *                                                                   (qlinear.py:106-114)
 * with U_n(hadK) = (hadK (x) H_{n/K}) / sqrt(n/K) acting on the row-major
 * (K, n/K) view (quant.py:81-84).  All vectors fp16; had is (K, K) fp16 row
 * major or NULL when K == 1; su / sv / wscale_vec / bias may be NULL.
 * `features` = unpadded width (in_features / out_features), n = padded width
 * (q_in / q_out), n == K * 2^e.  `transpose` selects hadK^T (input side). */
int quip_had_transform_f16(const void* x, void* y, int64_t rows, int32_t in_features,
                           int32_t out_features, int32_t n, int32_t K,
                           const void* had /* (K,K) fp16 or NULL */, int32_t transpose,
                           const void* pre_scale /* fp16[in_features] or NULL  */,
                           const void* pre_scale2 /* fp16[n] or NULL (Wscale)   */,
                           const void* post_scale /* fp16[out_features] or NULL */,
                           const void* bias /* fp16[out_features] or NULL */,
                           float scale, quip_stream_t stream);

/* Input side for the bs=1 decode path: the same transform as quip_had_transform_f16 (one row,
 * no output slice / post scale), written directly as the digit planes quip_e8p_gemv_planes()
 * consumes (quip_e8p_planes_bytes(n) bytes).  Removes the fp16 round trip of xh and the
 * separate x -> planes launch: QuantLinear.forward at bs=1 is had_transform_planes ->
 * e8p_gemv_planes -> had_transform_f16, three launches. */
int quip_had_transform_planes(const void* x, void* planes, int32_t in_features, int32_t n, int32_t K,
                              const void* had, int32_t transpose, const void* pre_scale, float scale,
                              quip_stream_t stream);

/* Fused variants: the decoder-block glue around a QuantLinear folded into the same launches
 * (no reference counterpart; they replace separate RMSNorm / SiLU*mul / residual-add kernels of
 * the surrounding HF decoder layer).  Every pointer of quip_had_fusion may be NULL. */
typedef struct {
  const void* residual;   /* fp16 [rows, out_features]: added to the output (f16 variant only)   */
  const void* rms_weight; /* fp16 [in_features]: input is RMSNorm(x) * rms_weight                 */
  const void* gate;       /* fp16 [rows, in_features]: input is silu(gate) * x                    */
  float rms_eps;
} quip_had_fusion;
int quip_had_transform_fused_f16(const void* x, void* y, int64_t rows, int32_t in_features,
                                 int32_t out_features, int32_t n, int32_t K, const void* had,
                                 int32_t transpose, const void* pre_scale, const void* pre_scale2,
                                 const void* post_scale, const void* bias, float scale,
                                 const quip_had_fusion* fusion, quip_stream_t stream);
int quip_had_transform_planes_fused(const void* x, void* planes, int32_t in_features, int32_t n,
                                    int32_t K, const void* had, int32_t transpose,
                                    const void* pre_scale, float scale,
                                    const quip_had_fusion* fusion, quip_stream_t stream);

/* ---- grouped launches -------------------------------------------------------------------------
 * Several QuantLinear modules that read the same activation (q/k/v_proj, gate/up_proj) are
 * independent reference calls (qlinear.py:87-115 once per module); at bs = 1 each launch costs
 * more than its math, so the three stages of up to QUIP_MAX_GROUP modules can each be issued as
 * one launch.  All problems of a group share n, K, transpose (Hadamard) resp. k (GEMV). */
#define QUIP_MAX_GROUP 3
typedef struct quip_had_problem {
  const void* x;           /* fp16 [rows, in_features] */
  void* out;               /* fp16 [rows, out_features], or digit planes (planes group) */
  const void* had;         /* (K, K) fp16 or NULL */
  const void* pre_scale;   /* fp16 [in_features] or NULL */
  const void* pre_scale2;  /* fp16 [n] or NULL */
  const void* post_scale;  /* fp16 [out_features] or NULL */
  const void* bias;        /* fp16 [out_features] or NULL */
  const void* residual;    /* fusion hooks as in quip_had_fusion, all optional */
  const void* rms_weight;
  const void* gate;
  int32_t in_features, out_features;
  float scale, rms_eps;
  /* chain (optional; K == 1 and in_features == n): the input row is the finished output of the
   * PRODUCER module, computed first from its raw GEMV output z:
   *   x = z_post_scale (.) (z_scale * H_n z) + z_residual    (qlinear.py:108-114 of the producer)
   * rounded to fp16 and also stored to h_out (by the first problem of the group).  x is ignored. */
  const void* z;             /* fp16 [rows, n] or NULL */
  const void* z_post_scale;  /* fp16 [n] */
  const void* z_residual;    /* fp16 [rows, n] or NULL */
  void* h_out;               /* fp16 [rows, n], must not alias z_residual */
  float z_scale;
  /* planes only.  != 0: write the planes of the E8P12RVQ4B virtual vector instead: per 8-group g
   * [resid_scale * x_g | x_g] (2n digits, quip_e8p_planes_bytes(2n) bytes).  An RVQ4 row read as
   * 16-bit E8P codes is a (n_out, 2k) E8P12 matrix whose 8-groups alternate residual / main codes
   * (code = main << 16 | resid, e8p12_rvq4.py:37-67), so quip_e8p_gemv_planes(planes, qidxs, ..., n_out,
   * 2k) IS the RVQ4 product W x with W = E8P(main) + resid_scale * E8P(resid), summed exactly. */
  float resid_scale;
  /* planes only.  2: write the planes of the HI virtual vector (2n digits): per 8-group
   * [x0 x2 0 0 | x4 x6 0 0 | x1 x3 0 0 | x5 x7 0 0].  An HI row (nibble i = column [0,2,4,6,1,3,5,7][i],
   * w = nibble - 7.5, hi.py:41-63) read as one-byte codes is a D4-style matrix with 2n columns and the
   * table entry [lo - 7.5, hi - 7.5, 0, 0], so quip_d4_gemv_planes(planes, qidxs, that_table, y, n_out, 2k)
   * IS the HI product.  0 (or 1): plain / RVQ4 (resid_scale). */
  int32_t planes_layout;
  /* fp16 group launches with K == 1 only.  != 0: this problem's own transform width (a power of two, 256..16384)
   * instead of the launch's n, so that output transforms of different widths (q_proj next to the narrower k / v of
   * a grouped-query model) share one launch; no rms_weight in such a group.  0: the launch's n. */
  int32_t n;
} quip_had_problem;
int quip_had_transform_group_f16(const quip_had_problem* problems, int32_t count, int64_t rows,
                                 int32_t n, int32_t K, int32_t transpose, quip_stream_t stream);
int quip_had_transform_planes_group(const quip_had_problem* problems, int32_t count, int32_t n,
                                    int32_t K, int32_t transpose, quip_stream_t stream);
/* ---- skinny GEMM on the matrix cores: 2..5 activation rows per pass over the codes ----------------
 * Replaces the 1 < M < 32 use of e8p_mm_origorder (origin_order.cu:388-555, 604-648; e8p12.py:147-150).
 * The MFMA of the bs=1 GEMV has 16 A rows and a single activation row fills 3 of them (its digit
 * planes); rows mode fills up to 15 with (activation row, plane) pairs, so `rows` activation rows are
 * multiplied in the ONE pass over the codes that bs=1 needs -- same exact integer sums per row, i.e. row
 * r of y is bit identical to quip_e8p_gemv_planes on row r alone.
 *   quip_had_transform_planes_rows: problem->x (rows, in_features) fp16 -> problem->out: `rows` plane
 *     images back to back, quip_e8p_planes_bytes(n) bytes each (same fusion fields as the planes group,
 *     row-wise; no chain fields).
 *   quip_e8p_gemv_max_rows(n, k): rows one launch takes for this shape (LDS budget: 5 for k <= 4096,
 *     3 for k <= 8192, 2 for k <= 15360, else 1); 0 if the shape is unsupported.
 *   quip_e8p_gemv_planes_rows: y (rows, n) fp16 row major; rows <= quip_e8p_gemv_max_rows(n, k). */
int quip_had_transform_planes_rows(const quip_had_problem* problem, int64_t rows, int32_t n, int32_t K,
                                   int32_t transpose, quip_stream_t stream);
int32_t quip_e8p_gemv_max_rows(int32_t n, int32_t k);
int quip_e8p_gemv_planes_rows(const void* planes, const void* qidxs, const void* grid_packed_abs, void* y,
                              int32_t rows, int32_t n, int32_t k, quip_stream_t stream);

/* ---- quantise-time codebook search (E8P12_codebook.round / quantize, e8p12.py:125-137) -------------
 * idx[i] = arg max_c (2 x_i . g_c - |g_c|^2) over the 65 536 E8P12 codewords, vals[i] = g_idx[i]; x, vals:
 * fp32 (nvec, 8) row major, 16-byte aligned; idx: int64 [nvec] (the reference's arg max dtype).  The
 * reference evaluates a (nvec, 8) x (8, 65 536) GEMM per LDLQ step (quant.py:103-135); this uses the
 * codebook's structure (abs row, even sign flips, +-1/4 shift): 512 candidates per vector, exact up to
 * fp32 rounding of the score (ties / near-ties may resolve to a different, equally near codeword). */
int quip_e8p_quantize_f32(const void* x, int64_t nvec, const void* grid_packed_abs, void* vals, void* idx,
                          quip_stream_t stream);

/* Rows mode for the other table modes of the same kernel.  mode 0: E8P12 (== quip_e8p_gemv_planes_rows);
 * mode 64: D4 (qidxs uint8 (n, k/4), grid = fp16 (256, 4) table; HI through its virtual 2k layout, see
 * quip_had_problem.planes_layout); mode 40: E8P12RVQ3B (qidxs = the checkpoint's 3-byte codes, read as 2k virtual
 * weights -- pass k = 2 * in features --, grid = grid_packed_abs, grid2 = e81b_i8, see
 * quip_e8prvq3_gemv_planes_group).  grid2 is ignored otherwise. */
int32_t quip_gemv_max_rows_mode(int32_t n, int32_t k, int32_t mode);
int quip_gemv_planes_rows_mode(const void* planes, const void* qidxs, const void* grid, const void* grid2, void* y,
                               int32_t rows, int32_t n, int32_t k, int32_t mode, quip_stream_t stream);

/* count GEMVs y[i] = W[i] x[i] (W[i]: (ns[i], k) E8P12 codes, x[i] as digit planes) */
int quip_e8p_gemv_planes_group(const void* const* planes, const void* const* qidxs,
                               const void* grid_packed_abs, void* const* ys, const int32_t* ns,
                               int32_t count, int32_t k, quip_stream_t stream);
/* Workspace variants of the two launches above.  Rows longer than the LDS image of x allows (k > ~20K, e.g. the
 * Llama-70B down projection, or E8P12RVQ4B's 2k-wide virtual rows) are split along k over workgroups; the integer
 * partial sums meet in `workspace` (agent-scope atomics; the last workgroup of a row block converts and stores y).
 * The workspace must be ZERO when first used and is left zero by every launch, so one buffer serves all launches
 * of a stream: quip_e8p_gemv_workspace_bytes(sum of ns) bytes, 16-byte aligned.  Without a workspace (NULL) such
 * shapes run on the first-generation kernel where it supports them.  Results are bit identical either way. */
size_t quip_e8p_gemv_workspace_bytes(int32_t n_total);
int quip_e8p_gemv_planes_ws(const void* planes, const void* qidxs, const void* grid_packed_abs, void* y, int32_t n,
                            int32_t k, void* workspace, size_t workspace_bytes, quip_stream_t stream);
int quip_e8p_gemv_planes_group_ws(const void* const* planes, const void* const* qidxs, const void* grid_packed_abs,
                                  void* const* ys, const int32_t* ns, int32_t count, int32_t k, void* workspace,
                                  size_t workspace_bytes, quip_stream_t stream);

/* D4 codebook (d4.py:26-96, origin_order.cu:143-168) on the same integer-domain matrix-core GEMV:
 * qidxs uint8 (n, k/4), grid_f16 the fp16 (256, 4) table; 2w of every entry is an int8, so the
 * arithmetic is exact like E8P12's.  x as digit planes (quip_e8p_x_to_planes / quip_had_transform_planes). */
int quip_d4_gemv_planes(const void* planes, const void* qidxs, const void* grid_f16, void* y /* fp16[n] */,
                        int32_t n, int32_t k, quip_stream_t stream);
int quip_d4_gemv_planes_group(const void* const* planes, const void* const* qidxs, const void* grid_f16,
                              void* const* ys, const int32_t* ns, int32_t count, int32_t k,
                              quip_stream_t stream);
/* ... with the zeroed workspace of quip_e8p_gemv_workspace_bytes(sum of ns): rows longer than 28672 (HI's virtual rows at the
 * Llama-2-70B down_proj width, 2 k = 57344) go to the K-splitting kernel in its D4 table mode; grid_f16 64-byte aligned there. */
/* (the K-splitting kernel's D4 table mode on any shape it takes, k % 128 == 0: for A/B tests against the first kernel) */
int quip_d4_gemv_planes_v2(const void* planes, const void* qidxs, const void* grid_f16, void* y, int32_t n, int32_t k,
                           void* workspace, size_t workspace_bytes, quip_stream_t stream);
int quip_d4_gemv_planes_group_ws(const void* const* planes, const void* const* qidxs, const void* grid_f16,
                                 void* const* ys, const int32_t* ns, int32_t count, int32_t k, void* workspace,
                                 size_t workspace_bytes, quip_stream_t stream);

/* E8P12RVQ3B (3-byte codes: resid8 | e8p16 << 8, e8p12_rvq3.py:81-107) on the matrix-core GEMV.  qidxs is the
 * checkpoint's own packed tensor (n rows of 3 k / 8 bytes, the int32 (n, 3 k / 32) Qidxs of the reference; rows need
 * 4-byte alignment only): a lane loads 12 bytes = 4 codes, and a code read behind a zero byte is the dword
 * (main16 << 16 | resid8 << 8) -- as 16-bit codes an RVQ4-style row of 2k virtual weights (8-groups alternate
 * residual / main), W x = W' x' with x' = [s x_g | x_g]_g: the planes of x' come from the Hadamard launch with
 * quip_had_problem.resid_scale = s, exactly as for E8P12RVQ4B.  The low code of every pair indexes the E81B
 * table instead of the E8P tables: e81b_i8 = int8 [256][8] = 4 * e81b_grid (natural column order, 8-byte
 * aligned).  main + s * resid is summed exactly (the reference rounds it to fp16 per weight,
 * origin_order.cu:287-335).  k = in features (the launch runs on 2k), k % 32 == 0; 2k <= 25600 (count * 2k for
 * groups) without a workspace.  (ABI 7: up to version 6 this entry point took codes repacked to 4 bytes.) */
int quip_e8prvq3_gemv_planes_group(const void* const* planes, const void* const* qidxs,
                                   const void* grid_packed_abs, const void* e81b_i8, void* const* ys,
                                   const int32_t* ns, int32_t count, int32_t k, quip_stream_t stream);
/* The same with a caller workspace (quip_e8p_gemv_workspace_bytes(n), zeroed once, see quip_e8p_gemv_planes_ws): a
 * single problem whose virtual row is longer than 25600 (Llama-2-70B down_proj: 2k = 57344) is then taken by the
 * K-splitting kernel with a third-table mode; everything else runs as above. */
int quip_e8prvq3_gemv_planes_group_ws(const void* const* planes, const void* const* qidxs,
                                      const void* grid_packed_abs, const void* e81b_i8, void* const* ys,
                                      const int32_t* ns, int32_t count, int32_t k, void* workspace,
                                      size_t workspace_bytes, quip_stream_t stream);

/* ---- GEMV with the input side of the layer(s) computed in its prologue (bs = 1, K_left == 1) ----
 * For `count` (1..3) E8P12 modules reading the same activation of width k (a power of two,
 * 1024..8192, in_features == q_in_features == k):
 *   h        = z ? post (.) (z_scale * H_k z) + residual : x        (z: the PRODUCER module's GEMV
 *              output whose output transform qlinear.py:108-114 is thereby folded in; h is also
 *              stored to h_out)
 *   ys[i]    = W_i . ( scale[i] * rms(h) * H_k (h (.) rms_weight (.) pre_scale[i]) )   [ns[i]]
 * i.e. qlinear.py:90-100 + e8p12.py:147-150 of every module, with the decoder block's RMSNorm in
 * front, in ONE launch.  ys are the raw GEMV outputs (before the modules' own output transform).
 * Bit identical to quip_had_transform_fused_f16 -> quip_had_transform_planes_group ->
 * quip_e8p_gemv_planes_group.  QUIP_ERR_UNSUPPORTED when the shape does not qualify. */
typedef struct quip_gemv_fused_in {
  const void* x;           /* fp16 [k] (used when z == NULL) */
  const void* z;           /* fp16 [k] or NULL */
  const void* post_scale;  /* fp16 [k]: producer's SV (z != NULL) */
  const void* residual;    /* fp16 [k] or NULL */
  void* h_out;             /* fp16 [k] (z != NULL), must not alias residual */
  const void* rms_weight;  /* fp16 [k] or NULL */
  const void* pre_scale[QUIP_MAX_GROUP];   /* SU of every module */
  float scale[QUIP_MAX_GROUP];             /* wscale_float / sqrt(k) of every module */
  float z_scale, rms_eps;
} quip_gemv_fused_in;
int quip_e8p_gemv_fused(const quip_gemv_fused_in* in, const void* const* qidxs,
                        const void* grid_packed_abs, void* const* ys, const int32_t* ns,
                        int32_t count, int32_t k, quip_stream_t stream);

/* ---- decode-step glue between q/k/v_proj and o_proj (bs = 1) -------------------------------
 * Rotary embedding of q and k at position *pos, append of (k, v) to the static KV cache and
 * single-query softmax attention over positions [0, *pos], one launch.  Replaces, for the
 * metric driver only, what example_generate.py:9-59 gets from HF LlamaAttention + StaticCache
 * under torch.compile; it is not a quip_cuda entry point.
 *   q [heads, head_dim], k / v [kv_heads, head_dim], out [heads, head_dim]: fp16
 *   cos / sin [max_len, head_dim] fp32 (HF half-rotation tables), pos: device int64 scalar
 *   kcache / vcache [kv_heads, max_len, head_dim] fp16, row *pos is written
 * head_dim 64 or 128; scale is the softmax scale (1/sqrt(head_dim)).
 * workspace (optional): quip_rope_attn_workspace_bytes(heads, head_dim) bytes, zeroed ONCE by the
 * caller and then reused by every call on that stream: with it, contexts beyond 256 positions are
 * split over 8 workgroups per head whose partial softmax states are merged by the last one to finish
 * (the one-workgroup-per-head walk is latency bound: 89 us at 2048 positions).  NULL: never split. */
size_t quip_rope_attn_workspace_bytes(int32_t heads, int32_t head_dim);
/* Greedy tail of the decode step: tok[0] = first index of the largest of the n fp16 logits (torch.argmax's tie
 * rule), pos[0] += 1 (both int64 on the device) -- one launch instead of reduce + copy + add. */
int quip_argmax_step_f16(const void* logits, int32_t n, void* tok, void* pos, quip_stream_t stream);
int quip_rope_attn_decode_f16(const void* q, const void* k, const void* v, const float* cos,
                              const float* sin, const int64_t* pos, void* kcache, void* vcache,
                              void* out, int32_t heads, int32_t kv_heads, int32_t head_dim,
                              int32_t max_len, float scale, void* workspace, quip_stream_t stream);
/* ... with sliding-window attention (HF `config.sliding_window`, e.g. Mistral-7B): the softmax runs over the last
 * `window` positions (pos - window, pos] only; the cache stays linear (row t = position t).  window == 0: all of [0, pos]
 * (quip_rope_attn_decode_f16 is this call with window 0). */
int quip_rope_attn_decode_window_f16(const void* q, const void* k, const void* v, const float* cos,
                                     const float* sin, const int64_t* pos, void* kcache, void* vcache,
                                     void* out, int32_t heads, int32_t kv_heads, int32_t head_dim,
                                     int32_t max_len, float scale, int32_t window, void* workspace,
                                     quip_stream_t stream);

/* The same launch on the RAW GEMV outputs of q / k / v_proj: their output-side transforms (SV (.) H z * scales[i],
 * qlinear.py:106-114 with K = 1, no bias) run in the launch's prologue -- same bits as quip_had_transform_f16 followed
 * by quip_rope_attn_decode_f16, one dependent launch less per decoder block.  z[3] / post[3]: fp16 [heads * head_dim]
 * each (q, k, v); needs heads == kv_heads and heads * head_dim a power of two in 256..4096
 * (quip_rope_attn_decode_z_supported). */
int quip_rope_attn_decode_z_supported(int32_t heads, int32_t kv_heads, int32_t head_dim);
int quip_rope_attn_decode_z_f16(const void* const* z, const void* const* post, const float* scales, const float* cos,
                                const float* sin, const int64_t* pos, void* kcache, void* vcache, void* out,
                                int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t max_len, float scale,
                                void* workspace, quip_stream_t stream);
int quip_rope_attn_decode_z_window_f16(const void* const* z, const void* const* post, const float* scales,
                                       const float* cos, const float* sin, const int64_t* pos, void* kcache, void* vcache,
                                       void* out, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t max_len,
                                       float scale, int32_t window, void* workspace, quip_stream_t stream);

/* Which kernel a bs=1 E8P12 GEMV launch of `count` matrices (ns[i] rows, common k) is dispatched to by default, without
 * launching anything: 1 = e8p_gemv_mfma_kernel, 2 = e8p_gemv_v2_kernel (needs the workspace of the *_ws entry points when
 * it splits K), negative = QUIP_ERR_*.  Lets a host-side test pin the regime table of the shapes it cares about. */
int quip_e8p_gemv_kernel_choice(const int32_t* ns, int32_t count, int32_t k);

/* ---- persistent decode engine, stage 1: the MLP half of a decoder block in ONE launch -------------------------
 * z_down = raw product of down_proj on  SU_d (.) silu(g) (.) u,  g / u = the finished outputs of gate_proj / up_proj
 * computed from the digit planes of their transformed input (quip_had_transform_planes_group): what
 * quip_e8p_gemv_planes_group + quip_had_transform_group_f16 + quip_had_transform_planes_fused + quip_e8p_gemv_planes
 * do in four launches (qlinear.py:103-114 for gate / up, :90-103 for down; origin_order.cu:388-555 at m = 1).
 * n_ffn = K * L with K x K the orthogonal factor of quant.py:26-39 (had_right of gate / up, had_left of down) and L a
 * power of two in 16..256; the launch has L workgroups that exchange data through `workspace` and must all be
 * resident, so L <= the number of CUs and NOTHING else may occupy the device while it runs (one engine launch at a
 * time per device; launches that share a workspace must be stream ordered).  Every wait inside the launch is bounded:
 * a launch that gives up writes a non-zero code to workspace word 1 (quip_ffn_engine_status) instead of hanging.
 * workspace: quip_ffn_engine_workspace_bytes(n_ffn, K) bytes, zeroed ONCE at allocation, then owned by the engine. */
typedef struct quip_ffn_engine_args {
  const void* w_gate;       /* Qidxs (n_ffn, hidden / 8) int16 */
  const void* w_up;
  const void* w_down;       /* Qidxs (hidden, n_ffn / 8) int16 */
  const void* planes_gate;  /* 3 * Kp + 16 bytes each (Kp = hidden rounded up to 512) */
  const void* planes_up;
  const void* had3;         /* fp16: gate.had_right, up.had_right (K x K row major, each padded to a multiple of 8 elements), down.had_left transposed, zero padded to (K16, K16), K16 = K rounded up to 16 */
  const void* sv_gate;      /* fp16 [n_ffn] */
  const void* sv_up;
  const void* su_down;      /* fp16 [n_ffn] */
  void* z_down;             /* fp16 [hidden] */
  const void* grid_packed_abs;
  void* workspace;
  void* dbg;                /* NULL, or 16 uint64 clock stamps per workgroup (bench) */
  float out_scale;          /* 1 / sqrt(L) */
  float in_scale;           /* down.wscale_float / sqrt(L) */
  int32_t hidden, n_ffn, K;
} quip_ffn_engine_args;
int quip_ffn_engine_supported(int32_t hidden, int32_t n_ffn, int32_t K);
size_t quip_ffn_engine_workspace_bytes(int32_t n_ffn, int32_t K);
int quip_ffn_engine(const quip_ffn_engine_args* args, quip_stream_t stream);

/* ---- persistent decode engine, stage 2: consecutive decoder blocks in ONE launch -------------------------------
 * One token (bs = 1) through n_layers Llama decoder blocks whose seven projections are E8P12 QuantLinear modules:
 * per block RMSNorm, q / k / v_proj, rotary embedding, KV-cache append at *pos, attention over [0, *pos], o_proj,
 * residual, RMSNorm, gate / up_proj, SiLU product, down_proj, residual (the HF LlamaDecoderLayer of the reference's
 * metric driver, example_generate.py:28-33, with qlinear.py:87-115 for every projection) -- the nine launches per block
 * of the stage-wise step.  Shape: hidden 4096, 32 heads of 128 (multi-head), n_ffn = 43 x 256 (Llama-2-7B);
 * quip_block_engine_supported says so.  256 workgroups exchange data through `workspace` and must all be resident:
 * nothing else may occupy the device while the launch runs; launches sharing a workspace must be stream ordered.
 * Every wait is bounded; a launch that gives up leaves a non-zero code in workspace word 1, answers all-NaN in h_out, and workspace
 * word 2 keeps (*pos + 1) of the first such launch (a host can replay from there on another path after zeroing the workspace).
 * Code 0xE000 in word 1 = the launch counter is about to wrap (after 2^22 launches on one workspace): zero the workspace.
 * The launch is refused (QUIP_ERR_UNSUPPORTED) when the device cannot hold all 256 workgroups at once (fewer CUs, e.g. a CPX
 * partition; occupancy query).
 * layers: n_layers descriptors of quip_block_engine_layer_bytes() = 256 bytes each, in device memory:
 *   uint64 W[7]   Qidxs of q, k, v, o, gate, up, down        uint64 ln[2]  input / post-attention RMSNorm weights (fp16)
 *   uint64 su[7]  SU of the same seven modules (fp16)          uint64 sv[7]  SV (fp16)
 *   uint64 had3   the K x K factors packed as for quip_ffn_engine
 *   uint64 kcache, vcache   fp16 [heads, max_len, 128], row *pos is written
 *   float  sc[7]  wscale_float / sqrt(L_in) (L_in = 4096; down: 256), then 5 floats of padding
 * workspace: quip_block_engine_workspace_bytes() bytes, zeroed once at allocation.  1 <= n_layers <= 146 per launch (longer
 * models: several launches).  From 128 positions on the eight workgroups of a head share its attention. */
typedef struct quip_block_engine_args {
  const void* layers;
  const void* h_in;          /* fp16 [4096]: embedding row of the token */
  void* h_out;               /* fp16 [4096]: hidden state after the last block */
  const void* pos;           /* int64 device scalar */
  const float* cos;          /* fp32 [max_len, 128] */
  const float* sin;
  const void* grid_packed_abs;
  void* workspace;
  void* dbg;                 /* NULL, or 32 uint64 clock stamps per workgroup of block dbg_layer */
                             /* (shape 1, dbg_layer == -2: MEASUREMENT MODE -- the products of every block without the edges,
                                the attention and the hand-offs; h_out holds no result and the call returns QUIP_NO_RESULT, not
                                QUIP_OK.  bench.py: gemv_stream_in_launch) */
  int32_t n_layers, max_len, dbg_layer;
  float rms_eps, attn_scale;
  int32_t codebook;          /* 0: E8P12; 1: D4 (uint8 codes, grid_packed_abs = the fp16 (256, 4) table, d4.py:26-96);
                              * 2: E8P12RVQ4B (int32 codes, e8p12_rvq4.py:37-45); 3: HI (int32 codes = 8 nibbles,
                              * hi.py:41-63; grid_packed_abs = the fp16 (256, 4) table [lo - 7.5, hi - 7.5, 0, 0] of a
                              * code BYTE: the row reads as a D4 row of twice the width); 4: E8P12RVQ3B (the checkpoint's
                              * 3-byte codes, int32 (n, 3 k / 32), e8p12_rvq3.py:81-107; grid2 = the E81B table) */
  float resid_scale;         /* codebooks 2, 4: the residual scale rounded to fp16 (origin_order.cu:337-385), else ignored */
  int32_t shape;             /* 0: hidden 4096, 32 heads, n_ffn 43 x 256 (Llama-2-7B); 1: hidden 8192, 64 heads on 8 KV heads,
                              * n_ffn 7 x 4096 (Llama-2-70B; E8P12 only) -- see quip_block_engine_gqa_* below */
  const void* grid2;         /* codebook 4: the E81B residual table as int8 (256, 8) = 4 r (as for quip_e8prvq3_gemv_planes_group),
                              * 8-byte aligned; else ignored */
} quip_block_engine_args;
int quip_block_engine_supported(int32_t hidden, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t n_ffn, int32_t K);
size_t quip_block_engine_workspace_bytes(void);
size_t quip_block_engine_layer_bytes(void);
int quip_block_engine(const quip_block_engine_args* args, quip_stream_t stream);
/* shape 1 (grouped-query attention, hidden 8192; csrc/decode_block_gqa.hip): the same call with args->shape = 1, h_in / h_out
 * fp16 [8192], kcache / vcache fp16 [kv_heads, max_len, 128], its own workspace size, and descriptors of the same 256-byte
 * layout whose static vectors are stored the way the launch reads them:
 *   ln[0], ln[1], su of q, k, v, gate, up and sv of o, down:  PERMUTED  p[16 t + k] = v[t + 512 k]  (t < 512, k < 16: the
 *     strided layout a 512-thread transform of 8192 points leaves its values in);  su of o, down and sv of q, k, v, gate, up: as stored;
 *   had3 -> float [3][7][8]: rows of gate.had_right, up.had_right and of down.had_left TRANSPOSED (7 x 7, padded to 8);
 *   sc[i] = wscale_float / sqrt(8192) (down: / 64).
 * One weight stream per wave runs through the whole launch (a ring of nine 2 KB requests always ahead of the products);
 * 7 workgroups own the 4096-point chunks of the MLP edge; from 128 positions on the four workgroups of a head share its
 * attention.  Same liveness rules as shape 0 (256 resident workgroups, bounded waits, workspace word 1). */
/* Test helper (not part of the replaced interface): nwg single-wave workgroups that each hold lds_bytes of LDS -- above 80 KB a
 * whole CU, as far as the persistent launches are concerned -- for `ticks` shader clocks, on `stream`.  The tests use it to
 * share the device with a persistent launch and check that the launch gives up cleanly instead of hanging. */
int quip_debug_occupy(int32_t nwg, int32_t lds_bytes, int64_t ticks, void* sink, quip_stream_t stream);
int quip_block_engine_gqa_supported(int32_t hidden, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t n_ffn, int32_t K);
size_t quip_block_engine_gqa_workspace_bytes(void);
/* shape 1 reads its seven code matrices W[0..6] RE-TILED (round 5), not in the checkpoint's row-major layout: inside every aligned
 * block of 16 rows the 64-byte pieces of the 16 rows lie side by side, piece after piece --
 *   tiled[rb][c][q][n] (16 bytes) = bytes [64 c + 16 q, +16) of row 16 rb + n,   rb < rows / 16, c < row_bytes / 64, q < 4, n < 16
 * -- so that one load instruction of the product (16 rows x 64 bytes of a row-major matrix: origin_order.cu:388-555 walks the rows)
 * covers 1 KB of consecutive bytes: a pure read stream in the row-major pattern tops at 0.72 of the HBM peak on this part, in full
 * lines at 0.85-0.88 (tools/ubench/hbm_read.hip).  quip_tile_codes writes that copy (same size; not in place; rows % 16 == 0,
 * row_bytes % 64 == 0, both pointers 16-byte aligned): once per matrix at model load.  Shapes 0 and 2 read the checkpoint's layout. */
int quip_tile_codes(const void* qidxs, void* tiled, int64_t rows, int64_t row_bytes, quip_stream_t stream);
/* the inverse (round 6): the checkpoint's row-major matrix back from its tiled copy (same constraints).  A decode-only server keeps
 * ONE copy of the codes of a shape-1 model -- the tiled one -- and materialises a matrix in the layout the other operators read
 * (decompress_*_origorder, *_mm_origorder: register_lib.py:8-192 of the reference) into a scratch buffer only where a prompt pass
 * or the stage-wise fallback needs it (decode.py: LlamaDecoder(single_copy=True)). */
int quip_untile_codes(const void* tiled, void* qidxs, int64_t rows, int64_t row_bytes, quip_stream_t stream);
/* shape 2 (round 5): hidden 4096, 32 heads of 128 on 8 KV heads, n_ffn = 14336 = 7 x 2048 (Llama-3-8B, Mistral-7B; E8P12 only): the
 * shape-0 launch compiled for this shape.  Descriptors as for shape 0, except had3 = the 56 x 56 factors R_7 (x) H_8 of gate.had_right,
 * up.had_right (row major, 3136 fp16 each) and of down.had_left TRANSPOSED (64 rows of 72 fp16, zero padded): see decode_block.hip, QUIP_BLOCK_G8. */
int quip_block_engine_g8_supported(int32_t hidden, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t n_ffn, int32_t K);
size_t quip_block_engine_g8_workspace_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* QUIP_MI355_H_ */
