// Internal (non-ABI) declarations shared by the translation units of libquip_mi355.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/quip_mi355.h"

namespace quip {

int device_cu_count();

// Tuning knobs of the E8P decode GEMV (0 = pick automatically).  Exposed through
// quip_e8p_gemv_tuned() for the micro-benchmark only; the ABI entry points use auto.
struct GemvTune {
  int rep = 0;      // LDS table replication: 1 or 16
  int rows = 0;     // rows in flight per wave iteration: 1, 2 or 4
  int blocks = 0;   // workgroups
  int waves_g = 0;  // row-groups per workgroup (waves = waves_g * J)
};

bool e8p_gemv_m1_supported(int n, int k);
int e8p_gemv_m1_launch(const void* x, const void* qidxs, const void* grid, void* y, int n, int k,
                       const GemvTune& tune, hipStream_t stream);

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
int had_transform_launch(const void* x, void* y, int64_t rows, int in_features, int out_features,
                         int n, int K, const void* had, int transpose, const void* pre,
                         const void* pre2, const void* post, const void* bias, float scale,
                         hipStream_t stream);

}  // namespace quip

extern "C" int quip_e8p_gemv_tuned(const void* x, const void* qidxs, const void* grid, void* y,
                                   int32_t n, int32_t k, int32_t rep, int32_t rows,
                                   int32_t blocks, int32_t waves_g, quip_stream_t stream);
