// Internal (non-ABI) declarations shared by the translation units of libquip_mi355.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/quip_mi355.h"

namespace quip {

int device_cu_count();
// Persistent launches (decode_engine.hip, decode_block*.hip) spin across workgroups: every one of the `nwg` workgroups has to be
// resident at once.  True when the current device has at least nwg CUs (no fall-back value: a failed query says no) and the
// occupancy query admits the grid for this kernel / block size / dynamic LDS.  Cached per kernel and device by the caller.
struct ResidencyCache { signed char ok[16] = {}; };      // 0 unknown, 1 fits, -1 does not
bool persistent_grid_fits(ResidencyCache& cache, const void* kernel, int threads, int lds, int nwg);
int device_cu_count_strict();                            // 0 when the query fails

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one static DynLdsCache per kernel
// instantiation (at its launch site) remembers the largest size configured on each device of the process, so a
// process that drives several GPUs configures the kernel on each of them.  (Benign race: the attribute is idempotent.)
struct DynLdsCache { int bytes[16] = {}; };
int ensure_dyn_lds(DynLdsCache& cache, const void* kernel, int lds);   // QUIP_OK / QUIP_ERR_LAUNCH

// Tuning knobs of the E8P decode GEMV (0 = pick automatically).  Exposed through
// quip_e8p_gemv_tuned() for the micro-benchmark only; the ABI entry points use auto.
struct GemvTune {
  int rep = 0;        // LDS table replication: 1 or 32 (i8 kernel) / 1 or 16 (f16 kernel)
  int rows = 0;       // rows in flight per wave iteration: 1, 2 or 4
  int blocks = 0;     // workgroups
  int waves_g = 0;    // row-groups per workgroup (waves = waves_g * J)
  int max_waves = 0;  // cap on waves per workgroup (default 16)
  int digits = 0;     // i8 kernel: int8 digit planes of x, 2 or 3 (default 3)
  const void* grid2 = nullptr;   // matrix-core GEMV, rep == 40 (E8P12RVQ3B): the E81B table, 256 x 8 int8 (4r)
  void* dbg = nullptr;  // i8 kernel: device buffer of 8 x uint64 s_memtime stamps per workgroup
};
int stream_probe_launch(const void* qidxs, void* out, int n, int k, const GemvTune& tune,
                        hipStream_t stream);

bool e8p_gemv_i8_supported(int n, int k);
size_t e8p_gemv_planes_bytes(int k);
int x_to_planes_launch(const void* x, void* planes, int k, hipStream_t stream);
// xmode 0: xsrc = digit planes (+ shift word), xmode 1: xsrc = fp16 x (self-contained, slower)
int e8p_gemv_i8_launch(const void* xsrc, int xmode, const void* qidxs, const void* grid, void* y,
                       int n, int k, const GemvTune& tune, hipStream_t stream);
// matrix-core GEMV (default bs=1 path): planes = [3][Kp] digit bytes + shift word
bool e8p_gemv_mfma_supported(int n, int k);
size_t e8p_gemv_mfma_planes_bytes(int k);
int x_to_planes_linear_launch(const void* x, void* planes, int k, hipStream_t stream, int rows = 1);
int e8p_gemv_mfma_launch(const void* planes, const void* qidxs, const void* grid, void* y, int n, int k,
                         const GemvTune& tune, hipStream_t stream);
// second-generation matrix-core GEMV (e8p_gemv_v2.hip): whole-line loads, K split over workgroups.
// ws: zeroed int32 workspace of e8p_gemv_v2_workspace_words(n) words (needed when K is split; left zeroed).
// tune: rep = 32 / 24 / 16 -> (32, 32) / (32, 16) / (16, 16) table copies, rows = load slots per wave,
// waves_g = K split, digits = segments per run, blocks, max_waves (0: automatic)
bool e8p_gemv_v2_supported(int n, int k);
size_t e8p_gemv_v2_workspace_words(int n);
int e8p_gemv_v2_group_launch(const void* const* planes, const void* const* qidxs, const void* grid, void* const* ys,
                             void* ws, const int* ns, int count, int k, const GemvTune& tune, hipStream_t stream);
// the same kernel in nibble mode (e8p_gemv_v2n.hip; tune.rep == 4): 4-byte table entries, 64 KB of tables whatever K is
int e8p_gemv_v2n_group_launch(const void* const* planes, const void* const* qidxs, const void* grid, void* const* ys,
                              void* ws, const int* ns, int count, int k, const GemvTune& tune, hipStream_t stream);
int e8p_gemv_v2_launch(const void* planes, const void* qidxs, const void* grid, void* y, void* ws, int n, int k,
                       const GemvTune& tune, hipStream_t stream);
int shape_probe_launch(const void* qidxs, void* out, int n, int k, const GemvTune& tune, hipStream_t stream);
int pattern_probe_launch(const void* qidxs, void* out, int n, int k, const GemvTune& tune, hipStream_t stream);

enum CodebookId { kE8P = 0, kE8PRVQ3 = 1, kE8PRVQ4 = 2, kD4 = 3, kHI = 4 };

struct CodebookArgs {
  const void* grid = nullptr;   // grid_packed_abs (E8P*), fp16 grid (D4)
  const void* grid2 = nullptr;  // e81b_grid_packed (RVQ3)
  float scale = 0.f;            // residual scale (RVQ3/4)
};

int generic_mm_launch(CodebookId cb, const void* x, const void* qidxs, const CodebookArgs& a,
                      void* y, int m, int n, int k, hipStream_t stream);
int decompress_launch(CodebookId cb, const void* qidxs, const CodebookArgs& a, void* w,
                      int64_t rows, int k, hipStream_t stream);
// optional decoder-block glue fused into the Hadamard kernels (all pointers may be null)
struct HadFusion {
  const void* residual = nullptr;    // fp16 [rows, out_features], added to the output
  const void* rms_weight = nullptr;  // fp16 [in_features]: RMSNorm(x) * weight before everything else
  const void* gate = nullptr;        // fp16 [rows, in_features]: input is silu(gate) * x
  float rms_eps = 1e-5f;
};
// one Hadamard problem of a grouped launch (same n, K, transpose, kind for the whole group)
struct HadProblem {
  const void* x = nullptr;
  void* out = nullptr;               // fp16 [rows, out_features] or digit planes
  const void* had = nullptr;
  const void* pre = nullptr;
  const void* pre2 = nullptr;
  const void* post = nullptr;
  const void* bias = nullptr;
  const void* residual = nullptr;
  const void* rms_weight = nullptr;
  const void* gate = nullptr;
  int in_features = 0, out_features = 0;
  float scale = 1.f, rms_eps = 1e-5f;
  // chain (K == 1, in_features == n): x := z_post (.) (z_scale * H_n z) + z_residual, also stored to h_out
  const void* z = nullptr;
  const void* z_post = nullptr;
  const void* z_residual = nullptr;
  void* h_out = nullptr;
  float z_scale = 1.f;
  float resid_scale = 0.f;   // planes only: != 0 -> planes of the E8P12RVQ4B virtual vector [s * x_g | x_g] (2n digits)
  int planes_layout = 0;     // planes only: 0 plain (or RVQ4 when resid_scale != 0), 2 = HI virtual vector (2n digits)
  int n = 0;                 // fp16, K == 1: this problem's own width (0: the launch's n)
};
int had_transform_group_launch(const HadProblem* problems, int count, bool planes, int64_t rows, int n, int K,
                               int transpose, hipStream_t stream);
// quantise-time nearest E8P12 codeword (quantize.hip): x fp32 (nvec, 8) -> vals fp32 (nvec, 8), idx int64 (nvec)
int e8p_quantize_launch(const void* x, int64_t nvec, const void* grid_packed_abs, void* vals, void* idx,
                        hipStream_t stream);
// rows mode: up to e8p_gemv_mfma_max_rows(n, k) <= 5 activation rows against one matrix in one pass
int e8p_gemv_mfma_max_rows(int n, int k, int mode = 0);   // mode: 0 E8P12, 64 D4 / HI, 40 E8P12RVQ3B (2k virtual)
int e8p_gemv_mfma_rows_launch(const void* planes, const void* qidxs, const void* grid, void* y, int mrows, int n,
                              int k, const GemvTune& tune, hipStream_t stream);
bool e8p_gemv_mfma_group_supported(const int* ns, int count, int k);
int e8p_gemv_mfma_group_launch(const void* const* planes, const void* const* qidxs, const void* grid,
                               void* const* ys, const int* ns, int count, int k, const GemvTune& tune,
                               hipStream_t stream);
// input side of a bs=1 GEMV computed in its prologue (see FusedIn in e8p_gemv_mfma.hip)
struct GemvFusedIn {
  const void* x = nullptr;         // fp16 [k]: the activation (when z == null)
  const void* z = nullptr;         // fp16 [k]: producer's GEMV output, still to be output-transformed
  const void* post = nullptr;      // fp16 [k]: producer's SV            (z != null)
  const void* residual = nullptr;  // fp16 [k] or null: added to the producer's output
  void* h_out = nullptr;           // fp16 [k]: receives the producer's output (z != null)
  const void* rms_w = nullptr;     // fp16 [k] or null: RMSNorm weight
  const void* pre[3] = {nullptr, nullptr, nullptr};   // SU of every problem
  float scale[3] = {1.f, 1.f, 1.f};                   // wscale / sqrt(k)
  float z_scale = 1.f, rms_eps = 1e-5f;
};
bool e8p_gemv_mfma_fused_supported(const int* ns, int count, int k);
int e8p_gemv_mfma_fused_launch(const GemvFusedIn& in, const void* const* qidxs, const void* grid,
                               void* const* ys, const int* ns, int count, int k, const GemvTune& tune,
                               hipStream_t stream);
// fused E8P12 dequant + MFMA GEMM for batches (e8p_prefill_gemm.hip): y (m, n) = x (m, k) @ decode(qidxs)^T
bool e8p_prefill_gemm_supported(int64_t m, int n, int k);
int e8p_prefill_gemm_launch(const void* x, const void* qidxs, const void* grid, void* y, int64_t m, int n, int k,
                            hipStream_t stream);
// the same tile kernel with the other codebooks' B-fragment decode (argument meaning as the skinny launchers below)
int e8prvq4_prefill_gemm_launch(const void* x, const void* qidxs, const void* grid, float resid_scale, void* y, int64_t m, int n,
                                int k, hipStream_t stream);
int e8prvq3_prefill_gemm_launch(const void* x, const void* qidxs, const void* grid, const void* e81b_packed, float resid_scale,
                                void* y, int64_t m, int n, int k, hipStream_t stream);
int d4_prefill_gemm_launch(const void* x, const void* qidxs, const void* grid_f16, void* y, int64_t m, int n, int k,
                           hipStream_t stream);
int hi_prefill_gemm_launch(const void* x, const void* qidxs, void* y, int64_t m, int n, int k, hipStream_t stream);
// single-pass skinny E8P12 product, 1 <= m <= 32 rows, fp16 MFMA (e8p_skinny_gemm.hip)
bool e8p_skinny_gemm_supported(int m, int n, int k);
int e8p_skinny_gemm_launch(const void* x, const void* qidxs, const void* grid, void* y, int m, int n, int k,
                           hipStream_t stream);
// the same for E8P12RVQ4B: 32-bit codes (main << 16 | residual), weight = fma(resid_scale, w_resid, w_main) in fp16
int e8prvq4_skinny_gemm_launch(const void* x, const void* qidxs, const void* grid, float resid_scale, void* y, int m, int n,
                               int k, hipStream_t stream);
// E8P12RVQ3B: 3-byte codes (n, 3 k / 8 bytes), e81b_packed = uint32 [256] (eight int4 = 2 x value per entry)
int e8prvq3_skinny_gemm_launch(const void* x, const void* qidxs, const void* grid, const void* e81b_packed, float resid_scale,
                               void* y, int m, int n, int k, hipStream_t stream);
// D4: one-byte codes (n, k/4), grid = the fp16 (256, 4) table; HI: 32-bit codes of eight nibbles (n, k/8)
int d4_skinny_gemm_launch(const void* x, const void* qidxs, const void* grid_f16, void* y, int m, int n, int k, hipStream_t stream);
int hi_skinny_gemm_launch(const void* x, const void* qidxs, void* y, int m, int n, int k, hipStream_t stream);
// bf16 / fp32 (and plain fp16) Walsh-Hadamard transform of the last dimension (hadamard_generic.hip)
int hadamard_generic_launch(const void* x, void* y, int64_t rows, int n, float scale, int dtype, hipStream_t stream);
int had_transform_launch(const void* x, void* y, int64_t rows, int in_features, int out_features,
                         int n, int K, const void* had, int transpose, const void* pre,
                         const void* pre2, const void* post, const void* bias, float scale,
                         hipStream_t stream, const HadFusion* fuse = nullptr);
int rope_attn_decode_launch(const void* q, const void* k, const void* v, const float* cos, const float* sin,
                            const int64_t* pos, void* kcache, void* vcache, void* out, int heads, int kv_heads,
                            int head_dim, int max_len, float scale, hipStream_t stream, void* workspace = nullptr,
                            int window = 0);
size_t rope_attn_workspace_bytes(int heads, int head_dim);
bool rope_attn_decode_z_supported(int heads, int kv_heads, int head_dim);
int rope_attn_decode_z_launch(const void* const* z, const void* const* post, const float* scales, const float* cos,
                              const float* sin, const int64_t* pos, void* kcache, void* vcache, void* out, int heads,
                              int kv_heads, int head_dim, int max_len, float scale, hipStream_t stream,
                              void* workspace, int window = 0);
int argmax_step_launch(const void* logits, int n, void* tok, void* pos, hipStream_t stream);
// persistent decode engine, stage 1 (decode_engine.hip): GEMV[gate, up] -> output transforms -> SiLU product ->
// input transform of down -> GEMV[down] of one decoder block in one launch
struct FfnEngineArgs {
  const void* w_gate = nullptr;      // Qidxs (n_ffn, hidden / 8) int16
  const void* w_up = nullptr;
  const void* w_down = nullptr;      // Qidxs (hidden, n_ffn / 8) int16
  const void* planes_gate = nullptr; // digit planes of gate's / up's transformed input (3 Kp + 16 bytes each)
  const void* planes_up = nullptr;
  const void* had3 = nullptr;        // fp16: gate.had_right, up.had_right (K x K row major, each padded to a multiple of 8 elements), down.had_left transposed, zero padded to (K16, K16), K16 = K rounded up to 16
  const void* sv_gate = nullptr;     // fp16 [n_ffn]
  const void* sv_up = nullptr;
  const void* su_down = nullptr;     // fp16 [n_ffn]
  void* z_down = nullptr;            // fp16 [hidden]: raw product of down_proj
  const void* grid = nullptr;        // grid_packed_abs
  void* workspace = nullptr;         // ffn_engine_workspace_bytes(), zeroed once at allocation
  void* dbg = nullptr;               // optional: 16 uint64 s_memtime stamps per workgroup
  float out_scale = 1.f;             // 1 / sqrt(L), L = n_ffn / K
  float in_scale = 1.f;              // down.wscale_float / sqrt(L)
  int hidden = 0, n_ffn = 0, K = 0;
};
bool ffn_engine_supported(int hidden, int n_ffn, int K);
size_t ffn_engine_workspace_bytes(int n_ffn, int K);
int ffn_engine_launch(const FfnEngineArgs& in, hipStream_t stream);
// persistent decode engine, stage 2 (decode_block.hip): consecutive decoder blocks of a Llama-2-7B-shaped model, one token
struct BlockEngineArgs {
  const void* layers = nullptr;      // n_layers descriptors of block_engine_layer_bytes() bytes each (device memory)
  const void* h_in = nullptr;        // fp16 [hidden]
  void* h_out = nullptr;             // fp16 [hidden]
  const void* pos = nullptr;         // int64 device scalar
  const float* cos = nullptr;        // fp32 [max_len, head_dim]
  const float* sin = nullptr;
  const void* grid = nullptr;        // grid_packed_abs
  void* workspace = nullptr;         // block_engine_workspace_bytes(), zeroed once
  void* dbg = nullptr;
  int n_layers = 0, max_len = 0, dbg_layer = -1;
  float rms_eps = 1e-5f, attn_scale = 1.f;
  int codebook = 0;                  // 0: E8P12 (grid = grid_packed_abs), 1: D4 (grid = the fp16 (256, 4) table), 2: E8P12RVQ4B, 3: HI (grid = the byte table), 4: E8P12RVQ3B
  float resid_scale = 0.f;           // codebooks 2, 4: the fp16 residual scale
  const void* grid2 = nullptr;       // codebook 4 (E8P12RVQ3B): int8 (256, 8) E81B table
};
bool block_engine_supported(int hidden, int heads, int kv_heads, int head_dim, int n_ffn, int K);
size_t block_engine_workspace_bytes();
size_t block_engine_layer_bytes();
int block_engine_launch(const BlockEngineArgs& in, hipStream_t stream);
// the same for the grouped-query 8192-wide shape (decode_block_gqa.hip: Llama-2-70B; E8P12): its own descriptor vectors
// (permuted, see include/quip_mi355.h) and workspace
bool block_engine_gqa_supported(int hidden, int heads, int kv_heads, int head_dim, int n_ffn, int K);
size_t block_engine_gqa_workspace_bytes();
size_t block_engine_gqa_layer_bytes();
int block_engine_gqa_launch(const BlockEngineArgs& in, hipStream_t stream);
// the 4096-wide grouped-query shape (Llama-3-8B / Mistral-7B): decode_block.hip compiled with QUIP_BLOCK_G8 (decode_block_g8.hip)
bool block_engine_g8_supported(int hidden, int heads, int kv_heads, int head_dim, int n_ffn, int K);
size_t block_engine_g8_workspace_bytes();
int block_engine_g8_launch(const BlockEngineArgs& in, hipStream_t stream);
int had_transform_planes_launch(const void* x, void* planes, int in_features, int n, int K,
                                const void* had, int transpose, const void* pre, float scale,
                                hipStream_t stream, const HadFusion* fuse = nullptr);

}  // namespace quip

// kernel: 4 = matrix-core GEMV (default), 0 / 3 = VALU integer GEMV (planes / fp16 x), 2 / 5 = read probes
extern "C" int quip_e8p_gemv_tuned(const void* x, const void* qidxs, const void* grid, void* y,
                                   int32_t n, int32_t k, int32_t kernel, int32_t rep, int32_t rows,
                                   int32_t blocks, int32_t waves_g, int32_t max_waves,
                                   int32_t digits, void* dbg, quip_stream_t stream);
extern "C" int quip_e8p_gemv_fused_tuned(const quip_gemv_fused_in* in, const void* const* qidxs, const void* grid,
                                         void* const* ys, const int32_t* ns, int32_t count, int32_t k,
                                         void* dbg, quip_stream_t stream);
extern "C" int quip_e8p_gemv_group_tuned(const void* const* planes, const void* const* qidxs, const void* grid,
                                         void* const* ys, const int32_t* ns, int32_t count, int32_t k,
                                         int32_t rep, int32_t rows, int32_t blocks, int32_t max_waves,
                                         void* dbg, quip_stream_t stream);
extern "C" int quip_e8p_gemv_v2_tuned(const void* planes, const void* qidxs, const void* grid, void* y, void* ws,
                                      int32_t n, int32_t k, int32_t rep2, int32_t slots, int32_t blocks,
                                      int32_t ksplit, int32_t max_waves, int32_t runlen, void* dbg, quip_stream_t stream);
extern "C" int quip_e8p_gemv_v2_group_tuned(const void* const* planes, const void* const* qidxs, const void* grid,
                                            void* const* ys, void* ws, const int32_t* ns, int32_t count, int32_t k,
                                            int32_t rep, int32_t slots, int32_t blocks, int32_t ksplit,
                                            int32_t max_waves, int32_t runlen, void* dbg, quip_stream_t stream);
extern "C" size_t quip_e8p_gemv_v2_workspace_bytes(int32_t n);
// kernel 2 in quip_e8p_gemv_tuned = streaming-read probe (y is a 4-byte scratch)
// lane-ordered digit planes for the VALU integer GEMV (kernel 0); planes: 3*k + 16 bytes
extern "C" int quip_e8p_x_to_planes_laneorder(const void* x, void* planes, int32_t k, quip_stream_t stream);
// test hook (decode_probe.hip): every E8P12 code through the shared decode core of table mode `mode` (4 nibble | 16 | 24 | 32 byte
// tables), read back through the matrix core; codes: 64 tiles x 2 KB in item lane order, out: int8 [65536][8] = 4 w
extern "C" int quip_e8p_decode_probe(const void* grid, const void* codes, void* out, int32_t mode, quip_stream_t stream);
