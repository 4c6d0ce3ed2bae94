// extern "C" entry points of libquip_mi355.so (see include/quip_mi355.h).
// Argument validation lives here; the reference validated nothing
// (origin_order.cu:557-788 have no dtype / shape / contiguity checks).
#include <stdlib.h>

#include "quip_internal.h"

namespace quip {

int device_cu_count() {
  static int cached[16] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
  if (cached[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    cached[dev] = v;
  }
  return cached[dev];
}

int device_cu_count_strict() {
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 0;
  return v;
}

bool persistent_grid_fits(ResidencyCache& cache, const void* kernel, int threads, int lds, int nwg) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const bool slot = dev >= 0 && dev < 16;
  if (slot && cache.ok[dev] != 0) return cache.ok[dev] > 0;
  bool fits = false;
  const int cus = device_cu_count_strict();
  int per_cu = 0;
  // (one workgroup per CU is the design -- the LDS footprint admits no second one --, so a device with fewer CUs than
  //  workgroups, e.g. a CPX partition or a CU-masked queue, cannot hold the grid whatever the occupancy query says)
  if (cus >= nwg && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, (size_t)lds) == hipSuccess)
    fits = (long long)per_cu * cus >= nwg;
  if (slot) cache.ok[dev] = fits ? 1 : -1;
  return fits;
}

int ensure_dyn_lds(DynLdsCache& cache, const void* kernel, int lds) {
  if (lds <= 48 * 1024) return QUIP_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return QUIP_ERR_LAUNCH;
  const bool slot = dev >= 0 && dev < 16;
  if (slot && lds <= cache.bytes[dev]) return QUIP_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return QUIP_ERR_LAUNCH;
  if (slot) cache.bytes[dev] = lds;
  return QUIP_OK;
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned64(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 63u) == 0; }

static int mm_common(CodebookId cb, const void* x, const void* q, const CodebookArgs& a, void* y,
                     int m, int n, int k, int kdiv, hipStream_t s) {
  if (!x || !q || !y) return QUIP_ERR_NULL_POINTER;
  if (m < 0 || n < 0 || k <= 0 || k % kdiv != 0 || k % 8 != 0) return QUIP_ERR_BAD_SHAPE;
  if (m == 0 || n == 0) return QUIP_OK;
  if (!aligned16(x)) return QUIP_ERR_MISALIGNED;
  return generic_mm_launch(cb, x, q, a, y, m, n, k, s);
}

}  // namespace quip

using namespace quip;

extern "C" {

int quip_abi_version(void) { return QUIP_ABI_VERSION; }

const char* quip_strerror(int code) {
  switch (code) {
    case QUIP_OK: return "ok";
    case QUIP_ERR_NULL_POINTER: return "null pointer argument";
    case QUIP_ERR_BAD_SHAPE: return "shape not supported by the packed format";
    case QUIP_ERR_MISALIGNED: return "pointer not 16-byte aligned";
    case QUIP_ERR_LAUNCH: return "HIP kernel launch failed";
    case QUIP_ERR_UNSUPPORTED: return "request not supported by this build";
    case QUIP_NO_RESULT: return "measurement launch: the outputs hold no result";
    default: return "unknown quip error";
  }
}

int quip_device_cu_count(void) { return device_cu_count(); }

namespace {
// test helper: `nwg` workgroups that hold `lds` bytes of LDS each (i.e. a CU each above 80 KB) for `ticks` shader clocks
__global__ void occupy_kernel(long long ticks, unsigned* sink) {
  extern __shared__ char smem_occ[];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned acc = 0;
  while ((long long)(__builtin_amdgcn_s_memtime() - t0) < ticks) {
    __builtin_amdgcn_s_sleep(32);
    acc += (unsigned)smem_occ[threadIdx.x];
  }
  if (acc == 0x12345678u) *sink = acc;
}
}  // namespace

int quip_debug_occupy(int32_t nwg, int32_t lds_bytes, int64_t ticks, void* sink, quip_stream_t stream) {
  if (nwg < 1 || nwg > 4096 || lds_bytes < 0 || lds_bytes > 160 * 1024 || ticks < 0 || !sink) return QUIP_ERR_BAD_SHAPE;
  static quip::DynLdsCache configured;
  if (quip::ensure_dyn_lds(configured, reinterpret_cast<const void*>(occupy_kernel), lds_bytes) != QUIP_OK) return QUIP_ERR_LAUNCH;
  hipLaunchKernelGGL(occupy_kernel, dim3(nwg), dim3(64), lds_bytes, (hipStream_t)stream, (long long)ticks, reinterpret_cast<unsigned*>(sink));
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

int quip_hadamard_f16(const void* x, void* y, int64_t rows, int32_t n, float scale,
                      quip_stream_t stream) {
  if (!x || !y) return QUIP_ERR_NULL_POINTER;
  return had_transform_launch(x, y, rows, n, n, n, 1, nullptr, 0, nullptr, nullptr, nullptr,
                              nullptr, scale, (hipStream_t)stream);
}

int quip_hadamard(const void* x, void* y, int64_t rows, int32_t n, float scale, int32_t dtype,
                  quip_stream_t stream) {
  if (!x || !y) return QUIP_ERR_NULL_POINTER;
  if (dtype == QUIP_DTYPE_F16) return quip_hadamard_f16(x, y, rows, n, scale, stream);
  return hadamard_generic_launch(x, y, rows, n, scale, dtype, (hipStream_t)stream);
}

int quip_had_transform_f16(const void* x, void* y, int64_t rows, int32_t in_features,
                           int32_t out_features, int32_t n, int32_t K, const void* had,
                           int32_t transpose, const void* pre_scale, const void* pre_scale2,
                           const void* post_scale, const void* bias, float scale,
                           quip_stream_t stream) {
  if (!x || !y) return QUIP_ERR_NULL_POINTER;
  return had_transform_launch(x, y, rows, in_features, out_features, n, K, had, transpose,
                              pre_scale, pre_scale2, post_scale, bias, scale, (hipStream_t)stream);
}

int quip_had_transform_planes(const void* x, void* planes, int32_t in_features, int32_t n, int32_t K,
                              const void* had, int32_t transpose, const void* pre_scale, float scale,
                              quip_stream_t stream) {
  if (!x || !planes) return QUIP_ERR_NULL_POINTER;
  if (!aligned16(planes)) return QUIP_ERR_MISALIGNED;
  return had_transform_planes_launch(x, planes, in_features, n, K, had, transpose, pre_scale, scale,
                                     (hipStream_t)stream);
}

static HadFusion to_fusion(const quip_had_fusion* f) {
  HadFusion h;
  if (f) { h.residual = f->residual; h.rms_weight = f->rms_weight; h.gate = f->gate; h.rms_eps = f->rms_eps; }
  return h;
}

int quip_had_transform_fused_f16(const void* x, void* y, int64_t rows, int32_t in_features,
                                 int32_t out_features, int32_t n, int32_t K, const void* had,
                                 int32_t transpose, const void* pre_scale, const void* pre_scale2,
                                 const void* post_scale, const void* bias, float scale,
                                 const quip_had_fusion* fusion, quip_stream_t stream) {
  if (!x || !y) return QUIP_ERR_NULL_POINTER;
  const HadFusion h = to_fusion(fusion);
  return had_transform_launch(x, y, rows, in_features, out_features, n, K, had, transpose, pre_scale,
                              pre_scale2, post_scale, bias, scale, (hipStream_t)stream, &h);
}

int quip_had_transform_planes_fused(const void* x, void* planes, int32_t in_features, int32_t n,
                                    int32_t K, const void* had, int32_t transpose,
                                    const void* pre_scale, float scale,
                                    const quip_had_fusion* fusion, quip_stream_t stream) {
  if (!x || !planes) return QUIP_ERR_NULL_POINTER;
  if (!aligned16(planes)) return QUIP_ERR_MISALIGNED;
  const HadFusion h = to_fusion(fusion);
  return had_transform_planes_launch(x, planes, in_features, n, K, had, transpose, pre_scale, scale,
                                     (hipStream_t)stream, &h);
}

static int group_had(const quip_had_problem* problems, int32_t count, bool planes, int64_t rows, int32_t n,
                     int32_t K, int32_t transpose, quip_stream_t stream) {
  if (!problems) return QUIP_ERR_NULL_POINTER;
  if (count < 1 || count > QUIP_MAX_GROUP) return QUIP_ERR_BAD_SHAPE;
  HadProblem pr[QUIP_MAX_GROUP];
  for (int i = 0; i < count; ++i) {
    const quip_had_problem& s = problems[i];
    if ((!s.x && !s.z) || !s.out) return QUIP_ERR_NULL_POINTER;
    pr[i].z = s.z; pr[i].z_post = s.z_post_scale; pr[i].z_residual = s.z_residual; pr[i].h_out = s.h_out;
    pr[i].z_scale = s.z_scale;
    pr[i].resid_scale = s.resid_scale;
    pr[i].planes_layout = s.planes_layout;
    pr[i].n = s.n;
    pr[i].x = s.x; pr[i].out = s.out; pr[i].had = s.had; pr[i].pre = s.pre_scale; pr[i].pre2 = s.pre_scale2;
    pr[i].post = s.post_scale; pr[i].bias = s.bias; pr[i].residual = s.residual; pr[i].rms_weight = s.rms_weight;
    pr[i].gate = s.gate; pr[i].in_features = s.in_features; pr[i].out_features = s.out_features;
    pr[i].scale = s.scale; pr[i].rms_eps = s.rms_eps;
  }
  return had_transform_group_launch(pr, count, planes, rows, n, K, transpose, (hipStream_t)stream);
}

int quip_had_transform_group_f16(const quip_had_problem* problems, int32_t count, int64_t rows,
                                 int32_t n, int32_t K, int32_t transpose, quip_stream_t stream) {
  return group_had(problems, count, false, rows, n, K, transpose, stream);
}

int quip_had_transform_planes_group(const quip_had_problem* problems, int32_t count, int32_t n,
                                    int32_t K, int32_t transpose, quip_stream_t stream) {
  return group_had(problems, count, true, 1, n, K, transpose, stream);
}

int quip_had_transform_planes_rows(const quip_had_problem* problem, int64_t rows, int32_t n, int32_t K,
                                   int32_t transpose, quip_stream_t stream) {
  if (rows < 1) return rows == 0 ? QUIP_OK : QUIP_ERR_BAD_SHAPE;
  return group_had(problem, 1, true, rows, n, K, transpose, stream);
}

int quip_e8p_quantize_f32(const void* x, int64_t nvec, const void* grid_packed_abs, void* vals, void* idx,
                          quip_stream_t stream) {
  if (nvec < 0) return QUIP_ERR_BAD_SHAPE;
  if (nvec == 0) return QUIP_OK;
  if (!x || !grid_packed_abs || !vals || !idx) return QUIP_ERR_NULL_POINTER;
  if (!aligned16(x) || !aligned16(vals) || (reinterpret_cast<uintptr_t>(idx) & 7) != 0) return QUIP_ERR_MISALIGNED;
  return e8p_quantize_launch(x, nvec, grid_packed_abs, vals, idx, (hipStream_t)stream);
}

int32_t quip_e8p_gemv_max_rows(int32_t n, int32_t k) { return (n > 0 && k > 0) ? e8p_gemv_mfma_max_rows(n, k) : 0; }

int quip_e8p_gemv_planes_rows(const void* planes, const void* qidxs, const void* grid_packed_abs, void* y,
                              int32_t rows, int32_t n, int32_t k, quip_stream_t stream) {
  if (!planes || !qidxs || !grid_packed_abs || !y) return QUIP_ERR_NULL_POINTER;
  if (rows < 1 || n < 1 || k < 1 || k % 8 != 0) return QUIP_ERR_BAD_SHAPE;
  if (!aligned16(planes) || !aligned16(qidxs)) return QUIP_ERR_MISALIGNED;
  return e8p_gemv_mfma_rows_launch(planes, qidxs, grid_packed_abs, y, rows, n, k, GemvTune{}, (hipStream_t)stream);
}

// rows mode with another table mode: 64 = D4 table (fp16 (256, 4) grid; HI through its virtual layout),
// 40 = E8P12RVQ3B (the 3-byte codes, k = 2 * in features, grid2 = e81b_i8)
int32_t quip_gemv_max_rows_mode(int32_t n, int32_t k, int32_t mode) {
  return (n > 0 && k > 0 && (mode == 0 || mode == 64 || mode == 40)) ? e8p_gemv_mfma_max_rows(n, k, mode) : 0;
}

int quip_gemv_planes_rows_mode(const void* planes, const void* qidxs, const void* grid, const void* grid2, void* y,
                               int32_t rows, int32_t n, int32_t k, int32_t mode, quip_stream_t stream) {
  if (!planes || !qidxs || !grid || !y) return QUIP_ERR_NULL_POINTER;
  if (rows < 1 || n < 1 || k < 1 || k % 8 != 0) return QUIP_ERR_BAD_SHAPE;
  if (mode != 0 && mode != 64 && mode != 40) return QUIP_ERR_BAD_SHAPE;
  if (mode == 40 && k % 64 != 0) return QUIP_ERR_BAD_SHAPE;   // rows of 3 (k / 2) / 8 bytes, dword aligned
  if (!aligned16(planes) || !aligned16(qidxs)) return QUIP_ERR_MISALIGNED;
  GemvTune t;
  t.rep = mode;
  t.grid2 = grid2;
  return e8p_gemv_mfma_rows_launch(planes, qidxs, grid, y, rows, n, k, t, (hipStream_t)stream);
}

// Which of the two matrix-core GEMVs serves a launch (measured on MI355X, tools/gemv_v2_bench.py): the second
// generation (whole-line loads, K split) wins from Llama-70B sizes on -- K >= 8192 with >= 16 MB of codes, or >= 20 MB
// of codes in the launch (7B gate / up group) -- and is the only one for rows longer than 28672 (E8P12RVQ4B's 2k-wide virtual rows at 70B);
// short launches stay on the first kernel (one-shot loads on 8 waves).  QUIP_GEMV_V2=0 / 1 forces one of them.
static int gemv_v2_mode() {
  static int mode = -2;
  if (mode == -2) {
    const char* e = getenv("QUIP_GEMV_V2");
    mode = e ? atoi(e) : -1;
  }
  return mode;
}

// which kernel a bs=1 E8P12 GEMV launch of `count` matrices takes first: the K-splitting kernel (e8p_gemv_v2.hip) when
// the first one (e8p_gemv_mfma.hip) does not take the shape, or for long rows (k >= 8192) from 16 MB of codes, or from
// 20 MB; QUIP_GEMV_V2 = 0 / 1 forces either
static bool gemv_prefers_v2(const int* ns, int count, int k, bool* v1_ok_out) {
  size_t bytes = 0;
  for (int i = 0; i < count; ++i) bytes += (size_t)ns[i] * (size_t)k / 4;
  const bool v1_ok = e8p_gemv_mfma_group_supported(ns, count, k);
  const int mode = gemv_v2_mode();
  if (v1_ok_out) *v1_ok_out = v1_ok;
  return mode == 1 || !v1_ok || (mode != 0 && ((k >= 8192 && bytes >= ((size_t)16 << 20)) || bytes >= ((size_t)20 << 20)));
}

int quip_e8p_gemv_kernel_choice(const int32_t* ns, int32_t count, int32_t k) {
  if (!ns) return QUIP_ERR_NULL_POINTER;
  if (count < 1 || count > QUIP_MAX_GROUP || k < 1 || k % 8 != 0) return QUIP_ERR_BAD_SHAPE;
  int n32[QUIP_MAX_GROUP];
  for (int i = 0; i < count; ++i) {
    if (ns[i] < 1) return QUIP_ERR_BAD_SHAPE;
    n32[i] = ns[i];
  }
  bool v1_ok = false;
  const bool v2 = gemv_prefers_v2(n32, count, k, &v1_ok);
  if (v2) {
    for (int i = 0; i < count; ++i)
      if (!e8p_gemv_v2_supported(n32[i], k)) return v1_ok ? 1 : QUIP_ERR_UNSUPPORTED;
    return 2;
  }
  return 1;
}

static int e8p_gemv_dispatch(const void* const* planes, const void* const* qidxs, const void* grid, void* const* ys,
                             const int* ns, int count, int k, void* ws, size_t ws_bytes, hipStream_t stream) {
  size_t need = 0;
  for (int i = 0; i < count; ++i) need += e8p_gemv_v2_workspace_words(ns[i]) * 4;
  if (ws && ws_bytes < need) ws = nullptr;
  bool v1_ok = false;
  const bool v2 = gemv_prefers_v2(ns, count, k, &v1_ok);
  if (v2) {
    const int rc = e8p_gemv_v2_group_launch(planes, qidxs, grid, ys, ws, ns, count, k, GemvTune{}, stream);
    if (rc == QUIP_OK || !v1_ok || (rc != QUIP_ERR_NULL_POINTER && rc != QUIP_ERR_UNSUPPORTED)) return rc;
    // needs a K split but no workspace was given: the first kernel takes it
  }
  return e8p_gemv_mfma_group_launch(planes, qidxs, grid, ys, ns, count, k, GemvTune{}, stream);
}

size_t quip_e8p_gemv_workspace_bytes(int32_t n_total) {
  return n_total < 1 ? 0 : (e8p_gemv_v2_workspace_words(n_total) + 2 * 64) * 4;
}

static int gemv_group_common(const void* const* planes, const void* const* qidxs, const void* grid_packed_abs,
                             void* const* ys, const int32_t* ns, int32_t count, int32_t k, void* ws, size_t ws_bytes,
                             quip_stream_t stream) {
  if (!planes || !qidxs || !grid_packed_abs || !ys || !ns) return QUIP_ERR_NULL_POINTER;
  if (count < 1 || count > QUIP_MAX_GROUP) return QUIP_ERR_BAD_SHAPE;
  int n32[QUIP_MAX_GROUP];
  for (int i = 0; i < count; ++i) {
    if (!planes[i] || !qidxs[i] || !ys[i]) return QUIP_ERR_NULL_POINTER;
    if (!aligned16(planes[i]) || !aligned16(qidxs[i])) return QUIP_ERR_MISALIGNED;
    if (ns[i] < 1) return QUIP_ERR_BAD_SHAPE;
    n32[i] = ns[i];
  }
  if (k < 1 || k % 8 != 0) return QUIP_ERR_BAD_SHAPE;
  if (!aligned64(grid_packed_abs) || (ws && !aligned16(ws))) return QUIP_ERR_MISALIGNED;
  return e8p_gemv_dispatch(planes, qidxs, grid_packed_abs, ys, n32, count, k, ws, ws_bytes, (hipStream_t)stream);
}

int quip_e8p_gemv_planes_group_ws(const void* const* planes, const void* const* qidxs, const void* grid_packed_abs,
                                  void* const* ys, const int32_t* ns, int32_t count, int32_t k, void* workspace,
                                  size_t workspace_bytes, quip_stream_t stream) {
  return gemv_group_common(planes, qidxs, grid_packed_abs, ys, ns, count, k, workspace, workspace_bytes, stream);
}

int quip_e8p_gemv_planes_ws(const void* planes, const void* qidxs, const void* grid, void* y, int32_t n, int32_t k,
                            void* workspace, size_t workspace_bytes, quip_stream_t stream) {
  if (n == 0) return QUIP_OK;
  return gemv_group_common(&planes, &qidxs, grid, &y, &n, 1, k, workspace, workspace_bytes, stream);
}

int quip_e8p_gemv_planes_group(const void* const* planes, const void* const* qidxs,
                               const void* grid_packed_abs, void* const* ys, const int32_t* ns,
                               int32_t count, int32_t k, quip_stream_t stream) {
  return gemv_group_common(planes, qidxs, grid_packed_abs, ys, ns, count, k, nullptr, 0, stream);
}

// E8P12RVQ3B on the matrix-core GEMV: a 3-byte code behind a zero byte is (main16 << 16 | resid8 << 8), i.e. an RVQ4-style row of
// 2k virtual weights whose low 16-bit codes index the E81B table (T3) instead of the E8P tables
static int e8prvq3_group_common(const void* const* planes, const void* const* qidxs, const void* grid_packed_abs,
                                const void* e81b_i8, void* const* ys, const int32_t* ns, int32_t count, int32_t k,
                                void* ws, size_t ws_bytes, quip_stream_t stream) {
  if (!planes || !qidxs || !grid_packed_abs || !e81b_i8 || !ys || !ns) return QUIP_ERR_NULL_POINTER;
  if (count < 1 || count > QUIP_MAX_GROUP) return QUIP_ERR_BAD_SHAPE;
  int n32[QUIP_MAX_GROUP];
  size_t need = 0;
  for (int i = 0; i < count; ++i) {
    if (!planes[i] || !qidxs[i] || !ys[i]) return QUIP_ERR_NULL_POINTER;
    if (!aligned16(planes[i]) || (reinterpret_cast<uintptr_t>(qidxs[i]) & 3) != 0) return QUIP_ERR_MISALIGNED;
    if (ns[i] < 1) return QUIP_ERR_BAD_SHAPE;
    n32[i] = ns[i];
    need += e8p_gemv_v2_workspace_words(ns[i]) * 4;
  }
  if (k < 1 || k % 32 != 0) return QUIP_ERR_BAD_SHAPE;   // rows of 3 k / 8 bytes, dword aligned
  if ((reinterpret_cast<uintptr_t>(e81b_i8) & 7) != 0 || (ws && !aligned16(ws))) return QUIP_ERR_MISALIGNED;
  GemvTune t;
  t.rep = 40;
  t.grid2 = e81b_i8;
  const int rc = e8p_gemv_mfma_group_launch(planes, qidxs, grid_packed_abs, ys, n32, count, 2 * k, t, (hipStream_t)stream);
  if (rc != QUIP_ERR_UNSUPPORTED) return rc;
  // virtual rows beyond the first kernel's LDS budget (70B down_proj: 2k = 57344): the K-splitting kernel, one problem
  if (count != 1) return QUIP_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < need) return QUIP_ERR_NULL_POINTER;
  return e8p_gemv_v2_group_launch(planes, qidxs, grid_packed_abs, ys, ws, n32, 1, 2 * k, t, (hipStream_t)stream);
}

int quip_e8prvq3_gemv_planes_group(const void* const* planes, const void* const* qidxs,
                                   const void* grid_packed_abs, const void* e81b_i8, void* const* ys,
                                   const int32_t* ns, int32_t count, int32_t k, quip_stream_t stream) {
  return e8prvq3_group_common(planes, qidxs, grid_packed_abs, e81b_i8, ys, ns, count, k, nullptr, 0, stream);
}

int quip_e8prvq3_gemv_planes_group_ws(const void* const* planes, const void* const* qidxs,
                                      const void* grid_packed_abs, const void* e81b_i8, void* const* ys,
                                      const int32_t* ns, int32_t count, int32_t k, void* workspace,
                                      size_t workspace_bytes, quip_stream_t stream) {
  return e8prvq3_group_common(planes, qidxs, grid_packed_abs, e81b_i8, ys, ns, count, k, workspace, workspace_bytes,
                              stream);
}

// D4 table mode of the matrix-core GEMVs: the first kernel wherever it takes the shape (k <= 28672), else the
// K-splitting kernel in its D4 table mode (HI's virtual rows at the 70B down_proj width: 2 k = 57344)
static int d4_group_common(const void* const* planes, const void* const* qidxs, const void* grid_f16,
                           void* const* ys, const int32_t* ns, int32_t count, int32_t k, void* ws, size_t ws_bytes,
                           quip_stream_t stream) {
  if (!planes || !qidxs || !grid_f16 || !ys || !ns) return QUIP_ERR_NULL_POINTER;
  if (count < 1 || count > QUIP_MAX_GROUP) return QUIP_ERR_BAD_SHAPE;
  int n32[QUIP_MAX_GROUP];
  size_t need = 0;
  for (int i = 0; i < count; ++i) {
    if (!planes[i] || !qidxs[i] || !ys[i]) return QUIP_ERR_NULL_POINTER;
    if (!aligned16(planes[i]) || !aligned16(qidxs[i])) return QUIP_ERR_MISALIGNED;
    if (ns[i] < 1) return QUIP_ERR_BAD_SHAPE;
    n32[i] = ns[i];
    need += e8p_gemv_v2_workspace_words(ns[i]) * 4;
  }
  if (k < 1 || k % 8 != 0) return QUIP_ERR_BAD_SHAPE;
  if (ws && !aligned16(ws)) return QUIP_ERR_MISALIGNED;
  if (ws && ws_bytes < need) ws = nullptr;
  GemvTune t;
  t.rep = 64;
  const int rc = e8p_gemv_mfma_group_launch(planes, qidxs, grid_f16, ys, n32, count, k, t, (hipStream_t)stream);
  if (rc != QUIP_ERR_UNSUPPORTED) return rc;
  return e8p_gemv_v2_group_launch(planes, qidxs, grid_f16, ys, ws, n32, count, k, t, (hipStream_t)stream);
}

int quip_d4_gemv_planes(const void* planes, const void* qidxs, const void* grid_f16, void* y, int32_t n,
                        int32_t k, quip_stream_t stream) {
  return d4_group_common(&planes, &qidxs, grid_f16, &y, &n, 1, k, nullptr, 0, stream);
}

int quip_d4_gemv_planes_group(const void* const* planes, const void* const* qidxs, const void* grid_f16,
                              void* const* ys, const int32_t* ns, int32_t count, int32_t k,
                              quip_stream_t stream) {
  return d4_group_common(planes, qidxs, grid_f16, ys, ns, count, k, nullptr, 0, stream);
}

int quip_d4_gemv_planes_v2(const void* planes, const void* qidxs, const void* grid_f16, void* y, int32_t n, int32_t k,
                           void* workspace, size_t workspace_bytes, quip_stream_t stream) {
  if (!planes || !qidxs || !grid_f16 || !y) return QUIP_ERR_NULL_POINTER;
  if (n < 1 || k < 1 || k % 8 != 0) return QUIP_ERR_BAD_SHAPE;
  if (!aligned16(planes) || !aligned16(qidxs) || (workspace && !aligned16(workspace))) return QUIP_ERR_MISALIGNED;
  if (workspace && workspace_bytes < e8p_gemv_v2_workspace_words(n) * 4) workspace = nullptr;
  GemvTune t;
  t.rep = 64;
  return e8p_gemv_v2_launch(planes, qidxs, grid_f16, y, workspace, n, k, t, (hipStream_t)stream);
}

int quip_d4_gemv_planes_group_ws(const void* const* planes, const void* const* qidxs, const void* grid_f16,
                                 void* const* ys, const int32_t* ns, int32_t count, int32_t k, void* workspace,
                                 size_t workspace_bytes, quip_stream_t stream) {
  return d4_group_common(planes, qidxs, grid_f16, ys, ns, count, k, workspace, workspace_bytes, stream);
}

int quip_e8p_gemv_fused(const quip_gemv_fused_in* in, const void* const* qidxs,
                        const void* grid_packed_abs, void* const* ys, const int32_t* ns,
                        int32_t count, int32_t k, quip_stream_t stream) {
  if (!in || !qidxs || !grid_packed_abs || !ys || !ns) return QUIP_ERR_NULL_POINTER;
  if (count < 1 || count > QUIP_MAX_GROUP) return QUIP_ERR_BAD_SHAPE;
  GemvFusedIn f;
  f.x = in->x; f.z = in->z; f.post = in->post_scale; f.residual = in->residual; f.h_out = in->h_out;
  f.rms_w = in->rms_weight; f.z_scale = in->z_scale; f.rms_eps = in->rms_eps;
  if (!f.z && !f.x) return QUIP_ERR_NULL_POINTER;
  if (f.z && (!f.post || !f.h_out)) return QUIP_ERR_NULL_POINTER;
  if (f.z && f.h_out == f.residual) return QUIP_ERR_BAD_SHAPE;
  if (!aligned16(f.x) || !aligned16(f.z) || !aligned16(f.post) || !aligned16(f.residual) || !aligned16(f.h_out) ||
      !aligned16(f.rms_w))
    return QUIP_ERR_MISALIGNED;
  int n32[QUIP_MAX_GROUP];
  for (int i = 0; i < count; ++i) {
    if (!qidxs[i] || !ys[i] || !in->pre_scale[i]) return QUIP_ERR_NULL_POINTER;
    if (!aligned16(qidxs[i]) || !aligned16(in->pre_scale[i])) return QUIP_ERR_MISALIGNED;
    if (ns[i] < 1) return QUIP_ERR_BAD_SHAPE;
    f.pre[i] = in->pre_scale[i];
    f.scale[i] = in->scale[i];
    n32[i] = ns[i];
  }
  return e8p_gemv_mfma_fused_launch(f, qidxs, grid_packed_abs, ys, n32, count, k, GemvTune{}, (hipStream_t)stream);
}

int quip_argmax_step_f16(const void* logits, int32_t n, void* tok, void* pos, quip_stream_t stream) {
  if (!logits || !tok || !pos) return QUIP_ERR_NULL_POINTER;
  if (!aligned16(logits) || (reinterpret_cast<uintptr_t>(tok) & 7) || (reinterpret_cast<uintptr_t>(pos) & 7))
    return QUIP_ERR_MISALIGNED;
  return argmax_step_launch(logits, n, tok, pos, (hipStream_t)stream);
}

int quip_ffn_engine_supported(int32_t hidden, int32_t n_ffn, int32_t K) {
  return ffn_engine_supported(hidden, n_ffn, K) ? 1 : 0;
}

size_t quip_ffn_engine_workspace_bytes(int32_t n_ffn, int32_t K) {
  return (n_ffn > 0 && K > 0 && n_ffn % K == 0) ? ffn_engine_workspace_bytes(n_ffn, K) : 0;
}

int quip_ffn_engine(const quip_ffn_engine_args* in, quip_stream_t stream) {
  if (!in) return QUIP_ERR_NULL_POINTER;
  if (!in->w_gate || !in->w_up || !in->w_down || !in->planes_gate || !in->planes_up || !in->had3 || !in->sv_gate ||
      !in->sv_up || !in->su_down || !in->z_down || !in->grid_packed_abs || !in->workspace)
    return QUIP_ERR_NULL_POINTER;
  if (!aligned16(in->w_gate) || !aligned16(in->w_up) || !aligned16(in->w_down) || !aligned16(in->planes_gate) ||
      !aligned16(in->planes_up) || !aligned16(in->had3) || !aligned16(in->sv_gate) || !aligned16(in->sv_up) ||
      !aligned16(in->su_down) || !aligned16(in->workspace) || (reinterpret_cast<uintptr_t>(in->grid_packed_abs) & 63u) != 0)
    return QUIP_ERR_MISALIGNED;
  if (in->hidden < 1 || in->n_ffn < 1 || in->K < 1) return QUIP_ERR_BAD_SHAPE;
  FfnEngineArgs a;
  a.w_gate = in->w_gate; a.w_up = in->w_up; a.w_down = in->w_down;
  a.planes_gate = in->planes_gate; a.planes_up = in->planes_up; a.had3 = in->had3;
  a.sv_gate = in->sv_gate; a.sv_up = in->sv_up; a.su_down = in->su_down;
  a.z_down = in->z_down; a.grid = in->grid_packed_abs; a.workspace = in->workspace; a.dbg = in->dbg;
  a.out_scale = in->out_scale; a.in_scale = in->in_scale;
  a.hidden = in->hidden; a.n_ffn = in->n_ffn; a.K = in->K;
  return ffn_engine_launch(a, (hipStream_t)stream);
}

int quip_block_engine_supported(int32_t hidden, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t n_ffn, int32_t K) {
  return block_engine_supported(hidden, heads, kv_heads, head_dim, n_ffn, K) ? 1 : 0;
}

size_t quip_block_engine_workspace_bytes(void) { return block_engine_workspace_bytes(); }
size_t quip_block_engine_layer_bytes(void) { return block_engine_layer_bytes(); }

int quip_block_engine(const quip_block_engine_args* in, quip_stream_t stream) {
  if (!in) return QUIP_ERR_NULL_POINTER;
  if (!in->layers || !in->h_in || !in->h_out || !in->pos || !in->cos || !in->sin || !in->grid_packed_abs || !in->workspace)
    return QUIP_ERR_NULL_POINTER;
  if (!aligned16(in->layers) || !aligned16(in->h_in) || !aligned16(in->h_out) || !aligned16(in->workspace) ||
      (reinterpret_cast<uintptr_t>(in->grid_packed_abs) & 63u) != 0)
    return QUIP_ERR_MISALIGNED;
  if (in->n_layers < 1 || in->max_len < 1) return QUIP_ERR_BAD_SHAPE;
  BlockEngineArgs a;
  a.layers = in->layers; a.h_in = in->h_in; a.h_out = in->h_out; a.pos = in->pos; a.cos = in->cos; a.sin = in->sin;
  a.grid = in->grid_packed_abs; a.workspace = in->workspace; a.dbg = in->dbg;
  a.n_layers = in->n_layers; a.max_len = in->max_len; a.dbg_layer = in->dbg_layer;
  a.rms_eps = in->rms_eps; a.attn_scale = in->attn_scale; a.codebook = in->codebook; a.resid_scale = in->resid_scale;
  a.grid2 = in->grid2;
  if (in->codebook == 4 && (!in->grid2 || (reinterpret_cast<uintptr_t>(in->grid2) & 7u) != 0))
    return in->grid2 ? QUIP_ERR_MISALIGNED : QUIP_ERR_NULL_POINTER;
  if (in->shape == 1) return block_engine_gqa_launch(a, (hipStream_t)stream);
  if (in->shape == 2) return block_engine_g8_launch(a, (hipStream_t)stream);
  if (in->shape != 0) return QUIP_ERR_UNSUPPORTED;
  return block_engine_launch(a, (hipStream_t)stream);
}

namespace {
// tiled[rb][c][q][n] = bytes [64 c + 16 q, +16) of row 16 rb + n: one 16-byte piece per thread, destination order
__global__ void tile_codes_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long pieces, int row_u4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pieces) return;
  const int n = (int)(i & 15), q = (int)((i >> 4) & 3);
  const long long cc = i >> 6;                         // (row block, 64-byte piece) pairs
  const int pieces_per_row = row_u4 / 4;
  const long long rb = cc / pieces_per_row;
  const int c = (int)(cc - rb * pieces_per_row);
  dst[i] = src[(rb * 16 + n) * row_u4 + c * 4 + q];
}
// the inverse: piece i of the tiled copy back to its place in the row-major matrix
__global__ void untile_codes_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long pieces, int row_u4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pieces) return;
  const int n = (int)(i & 15), q = (int)((i >> 4) & 3);
  const long long cc = i >> 6;
  const int pieces_per_row = row_u4 / 4;
  const long long rb = cc / pieces_per_row;
  const int c = (int)(cc - rb * pieces_per_row);
  dst[(rb * 16 + n) * row_u4 + c * 4 + q] = src[i];
}
}  // namespace

int quip_tile_codes(const void* qidxs, void* tiled, int64_t rows, int64_t row_bytes, quip_stream_t stream) {
  if (!qidxs || !tiled) return QUIP_ERR_NULL_POINTER;
  if (rows < 0 || row_bytes <= 0 || rows % 16 != 0 || row_bytes % 64 != 0 || row_bytes > (1 << 24)) return QUIP_ERR_BAD_SHAPE;
  if (rows == 0) return QUIP_OK;
  if (!aligned16(qidxs) || !aligned16(tiled)) return QUIP_ERR_MISALIGNED;
  {
    // not in place, and no partial overlap either: the kernel gathers from the source while it scatters
    const uintptr_t s0 = reinterpret_cast<uintptr_t>(qidxs), d0 = reinterpret_cast<uintptr_t>(tiled);
    const uintptr_t bytes = (uintptr_t)rows * (uintptr_t)row_bytes;
    if (s0 < d0 + bytes && d0 < s0 + bytes) return QUIP_ERR_UNSUPPORTED;
  }
  const long long pieces = rows * (row_bytes / 16);
  hipLaunchKernelGGL(tile_codes_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint4*>(qidxs), reinterpret_cast<uint4*>(tiled), pieces, (int)(row_bytes / 16));
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

int quip_untile_codes(const void* tiled, void* qidxs, int64_t rows, int64_t row_bytes, quip_stream_t stream) {
  if (!qidxs || !tiled) return QUIP_ERR_NULL_POINTER;
  if (rows < 0 || row_bytes <= 0 || rows % 16 != 0 || row_bytes % 64 != 0 || row_bytes > (1 << 24)) return QUIP_ERR_BAD_SHAPE;
  if (rows == 0) return QUIP_OK;
  if (!aligned16(qidxs) || !aligned16(tiled)) return QUIP_ERR_MISALIGNED;
  {
    const uintptr_t s0 = reinterpret_cast<uintptr_t>(tiled), d0 = reinterpret_cast<uintptr_t>(qidxs);
    const uintptr_t bytes = (uintptr_t)rows * (uintptr_t)row_bytes;
    if (s0 < d0 + bytes && d0 < s0 + bytes) return QUIP_ERR_UNSUPPORTED;
  }
  const long long pieces = rows * (row_bytes / 16);
  hipLaunchKernelGGL(untile_codes_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const uint4*>(tiled), reinterpret_cast<uint4*>(qidxs), pieces, (int)(row_bytes / 16));
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

int quip_block_engine_gqa_supported(int32_t hidden, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t n_ffn, int32_t K) {
  return block_engine_gqa_supported(hidden, heads, kv_heads, head_dim, n_ffn, K) ? 1 : 0;
}
size_t quip_block_engine_gqa_workspace_bytes(void) { return block_engine_gqa_workspace_bytes(); }
int quip_block_engine_g8_supported(int32_t hidden, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t n_ffn, int32_t K) {
  return block_engine_g8_supported(hidden, heads, kv_heads, head_dim, n_ffn, K) ? 1 : 0;
}
size_t quip_block_engine_g8_workspace_bytes(void) { return block_engine_g8_workspace_bytes(); }

size_t quip_rope_attn_workspace_bytes(int32_t heads, int32_t head_dim) {
  return heads > 0 && head_dim > 0 ? rope_attn_workspace_bytes(heads, head_dim) : 0;
}

int quip_rope_attn_decode_f16(const void* q, const void* k, const void* v, const float* cos,
                              const float* sin, const int64_t* pos, void* kcache, void* vcache,
                              void* out, int32_t heads, int32_t kv_heads, int32_t head_dim,
                              int32_t max_len, float scale, void* workspace, quip_stream_t stream) {
  return quip_rope_attn_decode_window_f16(q, k, v, cos, sin, pos, kcache, vcache, out, heads, kv_heads, head_dim, max_len,
                                          scale, 0, workspace, stream);
}

int quip_rope_attn_decode_window_f16(const void* q, const void* k, const void* v, const float* cos,
                                     const float* sin, const int64_t* pos, void* kcache, void* vcache,
                                     void* out, int32_t heads, int32_t kv_heads, int32_t head_dim,
                                     int32_t max_len, float scale, int32_t window, void* workspace,
                                     quip_stream_t stream) {
  if (!q || !k || !v || !cos || !sin || !pos || !kcache || !vcache || !out) return QUIP_ERR_NULL_POINTER;
  if (!aligned16(q) || !aligned16(k) || !aligned16(v) || !aligned16(kcache) || !aligned16(vcache))
    return QUIP_ERR_MISALIGNED;
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return QUIP_ERR_MISALIGNED;
  if (window < 0) return QUIP_ERR_BAD_SHAPE;
  return rope_attn_decode_launch(q, k, v, cos, sin, pos, kcache, vcache, out, heads, kv_heads, head_dim,
                                 max_len, scale, (hipStream_t)stream, workspace, window);
}

int quip_rope_attn_decode_z_supported(int32_t heads, int32_t kv_heads, int32_t head_dim) {
  return rope_attn_decode_z_supported(heads, kv_heads, head_dim) ? 1 : 0;
}

int quip_rope_attn_decode_z_f16(const void* const* z, const void* const* post, const float* scales, const float* cos,
                                const float* sin, const int64_t* pos, void* kcache, void* vcache, void* out,
                                int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t max_len, float scale,
                                void* workspace, quip_stream_t stream) {
  return quip_rope_attn_decode_z_window_f16(z, post, scales, cos, sin, pos, kcache, vcache, out, heads, kv_heads, head_dim,
                                            max_len, scale, 0, workspace, stream);
}

int quip_rope_attn_decode_z_window_f16(const void* const* z, const void* const* post, const float* scales,
                                       const float* cos, const float* sin, const int64_t* pos, void* kcache, void* vcache,
                                       void* out, int32_t heads, int32_t kv_heads, int32_t head_dim, int32_t max_len,
                                       float scale, int32_t window, void* workspace, quip_stream_t stream) {
  if (!z || !post || !scales || !cos || !sin || !pos || !kcache || !vcache || !out) return QUIP_ERR_NULL_POINTER;
  for (int i = 0; i < 3; ++i) {
    if (!z[i] || !post[i]) return QUIP_ERR_NULL_POINTER;
    if (!aligned16(z[i]) || !aligned16(post[i])) return QUIP_ERR_MISALIGNED;
  }
  if (!aligned16(kcache) || !aligned16(vcache)) return QUIP_ERR_MISALIGNED;
  if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return QUIP_ERR_MISALIGNED;
  if (window < 0) return QUIP_ERR_BAD_SHAPE;
  return rope_attn_decode_z_launch(z, post, scales, cos, sin, pos, kcache, vcache, out, heads, kv_heads, head_dim,
                                   max_len, scale, (hipStream_t)stream, workspace, window);
}

int quip_e8p_mm_origorder(const void* x, const void* qidxs, const void* grid, void* y, int32_t m,
                          int32_t n, int32_t k, quip_stream_t stream) {
  if (!grid) return QUIP_ERR_NULL_POINTER;
  if (x && qidxs && y && m == 1 && n > 0 && e8p_gemv_i8_supported(n, k) && aligned16(x) &&
      aligned16(qidxs))
    return e8p_gemv_i8_launch(x, 1, qidxs, grid, y, n, k, GemvTune{}, (hipStream_t)stream);
  CodebookArgs a;
  a.grid = grid;
  return mm_common(kE8P, x, qidxs, a, y, m, n, k, 8, (hipStream_t)stream);
}


size_t quip_e8p_planes_bytes(int32_t k) { return k > 0 ? e8p_gemv_mfma_planes_bytes(k) : 0; }

size_t quip_e8p_mm_workspace_bytes(int32_t m, int32_t n, int32_t k) {
  return (m >= 1 && m < 32 && e8p_gemv_mfma_supported(n, k)) ? (size_t)m * e8p_gemv_mfma_planes_bytes(k) : 0;
}

int quip_e8p_x_to_planes(const void* x, void* planes, int32_t k, quip_stream_t stream) {
  if (!x || !planes) return QUIP_ERR_NULL_POINTER;
  if (!aligned16(x) || !aligned16(planes)) return QUIP_ERR_MISALIGNED;
  return x_to_planes_linear_launch(x, planes, k, (hipStream_t)stream);
}

int quip_e8p_gemv_planes(const void* planes, const void* qidxs, const void* grid, void* y, int32_t n,
                         int32_t k, quip_stream_t stream) {
  if (!planes || !qidxs || !grid || !y) return QUIP_ERR_NULL_POINTER;
  if (n < 0) return QUIP_ERR_BAD_SHAPE;
  if (n == 0) return QUIP_OK;
  if (!aligned16(planes) || !aligned16(qidxs) || !aligned64(grid)) return QUIP_ERR_MISALIGNED;
  return gemv_group_common(&planes, &qidxs, grid, &y, &n, 1, k, nullptr, 0, stream);
}

int quip_e8p_mm_skinny(const void* x, const void* qidxs, const void* grid, void* y, int32_t m, int32_t n, int32_t k,
                       quip_stream_t stream) {
  if (!x || !qidxs || !grid || !y) return QUIP_ERR_NULL_POINTER;
  if (m < 0 || n < 1 || k < 8) return QUIP_ERR_BAD_SHAPE;
  if (m == 0) return QUIP_OK;
  if (!aligned16(x) || !aligned16(qidxs) || (reinterpret_cast<uintptr_t>(y) & 3u) || (reinterpret_cast<uintptr_t>(grid) & 7u))
    return QUIP_ERR_MISALIGNED;
  return e8p_skinny_gemm_launch(x, qidxs, grid, y, m, n, k, (hipStream_t)stream);
}

int quip_e8prvq4_mm_skinny(const void* x, const void* qidxs, const void* grid, float resid_scale, void* y, int32_t m,
                           int32_t n, int32_t k, quip_stream_t stream) {
  if (!x || !qidxs || !grid || !y) return QUIP_ERR_NULL_POINTER;
  if (m < 0 || n < 1 || k < 8) return QUIP_ERR_BAD_SHAPE;
  if (m == 0) return QUIP_OK;
  if (!aligned16(x) || !aligned16(qidxs) || (reinterpret_cast<uintptr_t>(y) & 3u) || (reinterpret_cast<uintptr_t>(grid) & 7u))
    return QUIP_ERR_MISALIGNED;
  return e8prvq4_skinny_gemm_launch(x, qidxs, grid, resid_scale, y, m, n, k, (hipStream_t)stream);
}

int quip_e8prvq3_mm_skinny(const void* x, const void* qidxs, const void* grid, const void* e81b_packed, float resid_scale,
                           void* y, int32_t m, int32_t n, int32_t k, quip_stream_t stream) {
  if (!x || !qidxs || !grid || !e81b_packed || !y) return QUIP_ERR_NULL_POINTER;
  if (m < 0 || n < 1 || k < 8) return QUIP_ERR_BAD_SHAPE;
  if (m == 0) return QUIP_OK;
  if (!aligned16(x) || (reinterpret_cast<uintptr_t>(qidxs) & 3u) || (reinterpret_cast<uintptr_t>(y) & 3u) ||
      (reinterpret_cast<uintptr_t>(grid) & 7u) || (reinterpret_cast<uintptr_t>(e81b_packed) & 7u))
    return QUIP_ERR_MISALIGNED;
  return e8prvq3_skinny_gemm_launch(x, qidxs, grid, e81b_packed, resid_scale, y, m, n, k, (hipStream_t)stream);
}

int quip_d4_mm_skinny(const void* x, const void* qidxs, const void* grid_f16, void* y, int32_t m, int32_t n, int32_t k,
                      quip_stream_t stream) {
  if (!x || !qidxs || !grid_f16 || !y) return QUIP_ERR_NULL_POINTER;
  if (m < 0 || n < 1 || k < 8) return QUIP_ERR_BAD_SHAPE;
  if (m == 0) return QUIP_OK;
  if (!aligned16(x) || !aligned16(qidxs) || (reinterpret_cast<uintptr_t>(y) & 3u) || (reinterpret_cast<uintptr_t>(grid_f16) & 7u))
    return QUIP_ERR_MISALIGNED;
  return d4_skinny_gemm_launch(x, qidxs, grid_f16, y, m, n, k, (hipStream_t)stream);
}

int quip_hi_mm_skinny(const void* x, const void* qidxs, void* y, int32_t m, int32_t n, int32_t k, quip_stream_t stream) {
  if (!x || !qidxs || !y) return QUIP_ERR_NULL_POINTER;
  if (m < 0 || n < 1 || k < 8) return QUIP_ERR_BAD_SHAPE;
  if (m == 0) return QUIP_OK;
  if (!aligned16(x) || !aligned16(qidxs) || (reinterpret_cast<uintptr_t>(y) & 3u)) return QUIP_ERR_MISALIGNED;
  return hi_skinny_gemm_launch(x, qidxs, y, m, n, k, (hipStream_t)stream);
}

int quip_e8p_mm_batched(const void* x, const void* qidxs, const void* grid, void* y, int64_t m, int32_t n, int32_t k,
                        quip_stream_t stream) {
  if (!x || !qidxs || !grid || !y) return QUIP_ERR_NULL_POINTER;
  if (m < 0 || n < 1 || k < 8) return QUIP_ERR_BAD_SHAPE;
  if (m == 0) return QUIP_OK;
  if (!aligned16(x) || !aligned16(qidxs) || !aligned16(y) || (reinterpret_cast<uintptr_t>(grid) & 7u)) return QUIP_ERR_MISALIGNED;
  return e8p_prefill_gemm_launch(x, qidxs, grid, y, m, n, k, (hipStream_t)stream);
}

// the other codebooks on the fused tile kernel (e8p_prefill_gemm.hip, MODE 1..4)
static int batched_args_ok(const void* x, const void* qidxs, const void* y, int64_t m, int32_t n, int32_t k) {
  if (!x || !qidxs || !y) return QUIP_ERR_NULL_POINTER;
  if (m < 0 || n < 1 || k < 8) return QUIP_ERR_BAD_SHAPE;
  if (!aligned16(x) || !aligned16(qidxs) || !aligned16(y)) return QUIP_ERR_MISALIGNED;
  return QUIP_OK;
}

int quip_e8prvq4_mm_batched(const void* x, const void* qidxs, const void* grid, float resid_scale, void* y, int64_t m,
                            int32_t n, int32_t k, quip_stream_t stream) {
  if (!grid) return QUIP_ERR_NULL_POINTER;
  if (const int st = batched_args_ok(x, qidxs, y, m, n, k)) return st;
  if (reinterpret_cast<uintptr_t>(grid) & 7u) return QUIP_ERR_MISALIGNED;
  if (m == 0) return QUIP_OK;
  return e8prvq4_prefill_gemm_launch(x, qidxs, grid, resid_scale, y, m, n, k, (hipStream_t)stream);
}

int quip_e8prvq3_mm_batched(const void* x, const void* qidxs, const void* grid, const void* e81b_packed, float resid_scale,
                            void* y, int64_t m, int32_t n, int32_t k, quip_stream_t stream) {
  if (!grid || !e81b_packed) return QUIP_ERR_NULL_POINTER;
  if (const int st = batched_args_ok(x, qidxs, y, m, n, k)) return st;
  if ((reinterpret_cast<uintptr_t>(grid) & 7u) || (reinterpret_cast<uintptr_t>(e81b_packed) & 3u)) return QUIP_ERR_MISALIGNED;
  if (m == 0) return QUIP_OK;
  return e8prvq3_prefill_gemm_launch(x, qidxs, grid, e81b_packed, resid_scale, y, m, n, k, (hipStream_t)stream);
}

int quip_d4_mm_batched(const void* x, const void* qidxs, const void* grid_f16, void* y, int64_t m, int32_t n, int32_t k,
                       quip_stream_t stream) {
  if (!grid_f16) return QUIP_ERR_NULL_POINTER;
  if (const int st = batched_args_ok(x, qidxs, y, m, n, k)) return st;
  if (reinterpret_cast<uintptr_t>(grid_f16) & 7u) return QUIP_ERR_MISALIGNED;
  if (m == 0) return QUIP_OK;
  return d4_prefill_gemm_launch(x, qidxs, grid_f16, y, m, n, k, (hipStream_t)stream);
}

int quip_hi_mm_batched(const void* x, const void* qidxs, void* y, int64_t m, int32_t n, int32_t k, quip_stream_t stream) {
  if (const int st = batched_args_ok(x, qidxs, y, m, n, k)) return st;
  if (m == 0) return QUIP_OK;
  return hi_prefill_gemm_launch(x, qidxs, y, m, n, k, (hipStream_t)stream);
}

int quip_e8p_mm_origorder_ws(const void* x, const void* qidxs, const void* grid, void* y, int32_t m,
                             int32_t n, int32_t k, void* workspace, size_t workspace_bytes,
                             quip_stream_t stream) {
  if (x && qidxs && grid && y && workspace && m >= 1 && m < 32 && n > 0 && e8p_gemv_mfma_supported(n, k) &&
      aligned16(x) && aligned16(qidxs) && aligned16(workspace) && aligned64(grid) &&
      workspace_bytes >= (size_t)m * e8p_gemv_mfma_planes_bytes(k)) {
    // 1 <= M < 32 on the matrix cores: digit planes of every row, then passes of up to
    // e8p_gemv_mfma_max_rows rows over the codes (M == 1: the plain GEMV)
    const int rc = x_to_planes_linear_launch(x, workspace, k, (hipStream_t)stream, m);
    if (rc != QUIP_OK) return rc;
    if (m == 1) return e8p_gemv_mfma_launch(workspace, qidxs, grid, y, n, k, GemvTune{}, (hipStream_t)stream);
    const int per = e8p_gemv_mfma_max_rows(n, k);
    const size_t pstride = e8p_gemv_mfma_planes_bytes(k);
    for (int r0 = 0; r0 < m; r0 += per) {
      const int mr = m - r0 < per ? m - r0 : per;
      const int rc2 = e8p_gemv_mfma_rows_launch(reinterpret_cast<const char*>(workspace) + (size_t)r0 * pstride, qidxs,
                                                grid, reinterpret_cast<char*>(y) + (size_t)r0 * n * 2, mr, n, k,
                                                GemvTune{}, (hipStream_t)stream);
      if (rc2 != QUIP_OK) return rc2;
    }
    return QUIP_OK;
  }
  return quip_e8p_mm_origorder(x, qidxs, grid, y, m, n, k, stream);
}

int quip_e8p_x_to_planes_laneorder(const void* x, void* planes, int32_t k, quip_stream_t stream) {
  if (!x || !planes) return QUIP_ERR_NULL_POINTER;
  return x_to_planes_launch(x, planes, k, (hipStream_t)stream);
}

int quip_e8p_gemv_tuned(const void* x, const void* qidxs, const void* grid, void* y, int32_t n,
                        int32_t k, int32_t kernel, int32_t rep, int32_t rows, int32_t blocks,
                        int32_t waves_g, int32_t max_waves, int32_t digits, void* dbg,
                        quip_stream_t stream) {
  if (!x || !qidxs || !grid || !y) return QUIP_ERR_NULL_POINTER;
  GemvTune t;
  t.rep = rep; t.rows = rows; t.blocks = blocks; t.waves_g = waves_g;
  t.max_waves = max_waves; t.digits = digits; t.dbg = dbg;
  if (kernel == 2) return stream_probe_launch(qidxs, y, n, k, t, (hipStream_t)stream);
  if (kernel == 5) return pattern_probe_launch(qidxs, y, n, k, t, (hipStream_t)stream);
  if (kernel == 6) return shape_probe_launch(qidxs, y, n, k, t, (hipStream_t)stream);
  if (kernel == 4) return e8p_gemv_mfma_launch(x, qidxs, grid, y, n, k, t, (hipStream_t)stream);
  // kernel 4: matrix-core GEMV on linear digit planes; kernel 0: VALU integer GEMV on lane-ordered
  // digit planes; kernel 3: the same converting fp16 x itself
  return e8p_gemv_i8_launch(x, kernel == 3 ? 1 : 0, qidxs, grid, y, n, k, t, (hipStream_t)stream);
}

int quip_e8p_gemv_v2_tuned(const void* planes, const void* qidxs, const void* grid, void* y, void* ws, int32_t n,
                           int32_t k, int32_t rep2, int32_t slots, int32_t blocks, int32_t ksplit,
                           int32_t max_waves, int32_t runlen, void* dbg, quip_stream_t stream) {
  if (!planes || !qidxs || !grid || !y) return QUIP_ERR_NULL_POINTER;
  GemvTune t;
  t.rep = rep2; t.rows = slots; t.blocks = blocks; t.waves_g = ksplit; t.max_waves = max_waves; t.dbg = dbg;
  t.digits = runlen;
  return e8p_gemv_v2_launch(planes, qidxs, grid, y, ws, n, k, t, (hipStream_t)stream);
}

int quip_e8p_gemv_v2_group_tuned(const void* const* planes, const void* const* qidxs, const void* grid,
                                 void* const* ys, void* ws, const int32_t* ns, int32_t count, int32_t k, int32_t rep,
                                 int32_t slots, int32_t blocks, int32_t ksplit, int32_t max_waves, int32_t runlen,
                                 void* dbg, quip_stream_t stream) {
  if (!planes || !qidxs || !grid || !ys || !ns) return QUIP_ERR_NULL_POINTER;
  if (count < 1 || count > QUIP_MAX_GROUP) return QUIP_ERR_UNSUPPORTED;
  GemvTune t;
  t.rep = rep; t.rows = slots; t.blocks = blocks; t.waves_g = ksplit; t.max_waves = max_waves; t.dbg = dbg;
  t.digits = runlen;
  int n32[QUIP_MAX_GROUP];
  for (int i = 0; i < count; ++i) n32[i] = ns[i];
  return e8p_gemv_v2_group_launch(planes, qidxs, grid, ys, ws, n32, count, k, t, (hipStream_t)stream);
}

size_t quip_e8p_gemv_v2_workspace_bytes(int32_t n) { return e8p_gemv_v2_workspace_words(n) * 4; }

int quip_e8p_gemv_fused_tuned(const quip_gemv_fused_in* in, const void* const* qidxs, const void* grid,
                              void* const* ys, const int32_t* ns, int32_t count, int32_t k, void* dbg,
                              quip_stream_t stream) {
  GemvFusedIn f;
  f.x = in->x; f.z = in->z; f.post = in->post_scale; f.residual = in->residual; f.h_out = in->h_out;
  f.rms_w = in->rms_weight; f.z_scale = in->z_scale; f.rms_eps = in->rms_eps;
  int n32[QUIP_MAX_GROUP];
  for (int i = 0; i < count && i < QUIP_MAX_GROUP; ++i) {
    f.pre[i] = in->pre_scale[i]; f.scale[i] = in->scale[i]; n32[i] = ns[i];
  }
  GemvTune t;
  t.dbg = dbg;
  return e8p_gemv_mfma_fused_launch(f, qidxs, grid, ys, n32, count, k, t, (hipStream_t)stream);
}

int quip_e8p_gemv_group_tuned(const void* const* planes, const void* const* qidxs, const void* grid,
                              void* const* ys, const int32_t* ns, int32_t count, int32_t k, int32_t rep,
                              int32_t rows, int32_t blocks, int32_t max_waves, void* dbg,
                              quip_stream_t stream) {
  GemvTune t;
  t.rep = rep; t.rows = rows; t.blocks = blocks; t.max_waves = max_waves; t.dbg = dbg;
  int n32[QUIP_MAX_GROUP];
  for (int i = 0; i < count && i < QUIP_MAX_GROUP; ++i) n32[i] = ns[i];
  return e8p_gemv_mfma_group_launch(planes, qidxs, grid, ys, n32, count, k, t, (hipStream_t)stream);
}

int quip_e8prvq3_mm_origorder(const void* x, const void* qidxs, const void* grid,
                              const void* grid2, float scale, void* y, int32_t m, int32_t n,
                              int32_t k, quip_stream_t stream) {
  if (!grid || !grid2) return QUIP_ERR_NULL_POINTER;
  CodebookArgs a;
  a.grid = grid; a.grid2 = grid2; a.scale = scale;
  return mm_common(kE8PRVQ3, x, qidxs, a, y, m, n, k, 32, (hipStream_t)stream);
}

int quip_e8prvq4_mm_origorder(const void* x, const void* qidxs, const void* grid, float scale,
                              void* y, int32_t m, int32_t n, int32_t k, quip_stream_t stream) {
  if (!grid) return QUIP_ERR_NULL_POINTER;
  CodebookArgs a;
  a.grid = grid; a.scale = scale;
  return mm_common(kE8PRVQ4, x, qidxs, a, y, m, n, k, 8, (hipStream_t)stream);
}

int quip_d4_mm_origorder(const void* x, const void* qidxs, const void* grid_f16, void* y,
                         int32_t m, int32_t n, int32_t k, quip_stream_t stream) {
  if (!grid_f16) return QUIP_ERR_NULL_POINTER;
  CodebookArgs a;
  a.grid = grid_f16;
  return mm_common(kD4, x, qidxs, a, y, m, n, k, 8, (hipStream_t)stream);
}

int quip_hi_mm_origorder(const void* x, const void* qidxs, void* y, int32_t m, int32_t n,
                         int32_t k, quip_stream_t stream) {
  return mm_common(kHI, x, qidxs, CodebookArgs{}, y, m, n, k, 8, (hipStream_t)stream);
}

static int dec_common(CodebookId cb, const void* q, const CodebookArgs& a, void* w, int64_t rows,
                      int32_t k, int kdiv, hipStream_t s) {
  if (!q || !w) return QUIP_ERR_NULL_POINTER;
  if (rows < 0 || k <= 0 || k % kdiv != 0 || k % 8 != 0) return QUIP_ERR_BAD_SHAPE;
  if (rows == 0) return QUIP_OK;
  if (!aligned16(w)) return QUIP_ERR_MISALIGNED;
  return decompress_launch(cb, q, a, w, rows, k, s);
}

int quip_decompress_e8p_origorder(const void* qidxs, const void* grid, void* w, int64_t rows,
                                  int32_t k, quip_stream_t stream) {
  if (!grid) return QUIP_ERR_NULL_POINTER;
  CodebookArgs a;
  a.grid = grid;
  return dec_common(kE8P, qidxs, a, w, rows, k, 8, (hipStream_t)stream);
}

int quip_decompress_e8prvq3_origorder(const void* qidxs, const void* grid, const void* grid2,
                                      float scale, void* w, int64_t rows, int32_t k,
                                      quip_stream_t stream) {
  if (!grid || !grid2) return QUIP_ERR_NULL_POINTER;
  CodebookArgs a;
  a.grid = grid; a.grid2 = grid2; a.scale = scale;
  return dec_common(kE8PRVQ3, qidxs, a, w, rows, k, 32, (hipStream_t)stream);
}

int quip_decompress_e8prvq4_origorder(const void* qidxs, const void* grid, float scale, void* w,
                                      int64_t rows, int32_t k, quip_stream_t stream) {
  if (!grid) return QUIP_ERR_NULL_POINTER;
  CodebookArgs a;
  a.grid = grid; a.scale = scale;
  return dec_common(kE8PRVQ4, qidxs, a, w, rows, k, 8, (hipStream_t)stream);
}

int quip_decompress_d4_origorder(const void* qidxs, const void* grid_f16, void* w, int64_t rows,
                                 int32_t k, quip_stream_t stream) {
  if (!grid_f16) return QUIP_ERR_NULL_POINTER;
  CodebookArgs a;
  a.grid = grid_f16;
  return dec_common(kD4, qidxs, a, w, rows, k, 8, (hipStream_t)stream);
}

int quip_decompress_hi_origorder(const void* qidxs, void* w, int64_t rows, int32_t k,
                                 quip_stream_t stream) {
  return dec_common(kHI, qidxs, CodebookArgs{}, w, rows, k, 8, (hipStream_t)stream);
}

}  // extern "C"
