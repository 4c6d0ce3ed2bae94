// Test hook (not a reference interface): what the decode core the matrix-core kernels share (e8p_gemv_core.hip.h: LDS table
// build, code -> look-up addresses, the B fragments of the MFMAs) makes of every E8P12 code, read back THROUGH the matrix core --
// A operands that are one-hot rows, so that row 0 of a v_mfma_i32_16x16x64_i8 result is one byte of the B fragment per column.
// The persistent launches (decode_block*.hip) take their tables, addresses and fragments from exactly these functions; their
// results are compared with the stage-wise step and the float64 model only to a few fp16 ulps, which one mis-decoded code in
// 65 536 would not move: this pins the core itself, code by code, bit for bit, in every E8P12 table mode --
// mode 4 = nibble (round 6), 16 / 24 / 32 = the byte tables with 16 / 16, 32 / 16, 32 / 32 copies.
// tests/test_gpu_exhaustive_codes.py::test_decode_core_of_every_table_mode_decodes_every_code.
#include "e8p_gemv_core.hip.h"

namespace quip {
namespace {

// codes: 64 tiles of 2 KB in the lane order of the kernels' item loads -- tile t = 16 rows x 64 codes, lane (n = l & 15, q = l >> 4)
// holds bytes [64 c + 16 q, +16) of row n's 128-byte line in its c-th 16-byte piece (c = 0: first KB of the tile, 1: second);
// out: int8 [65536][8] = 4 w of code id (t * 16 + n) * 64 + j at out[id * 8 + position]
template <int REP>
__global__ __launch_bounds__(512) void decode_probe_kernel(const uint64_t* grid, const uint4* codes, int8_t* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = Lds<REP>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    const uint2 s = T::kNib ? *table_source_ptr_nib(grid, lane, wave) : *table_source_ptr(grid, lane, wave);
    if constexpr (T::kNib) fill_tables_nib(u32x2{s.x, s.y}, lane, wave);
    else fill_tables_from_lane<REP>(smem, u32x2{s.x, s.y}, lane, wave);
  }
  __syncthreads();
  const int tile = blockIdx.x * 8 + wave;
  const int n = lane & 15, q = lane >> 4;
  const uint4 c0 = codes[tile * 128 + lane], c1 = codes[tile * 128 + 64 + lane];
  const u32x4 qa = {c0.x, c0.y, c0.z, c0.w}, qb = {c1.x, c1.y, c1.z, c1.w};
  uint32_t lane_c, lane_c2 = 0;
  if constexpr (T::kNib) lane_c = nib_lane_const(lane);
  else {
    lane_c = (T::kRep1 == 32) ? ((((uint32_t)lane & 31u) << 3) | (REP == 32 ? 0x00010000u : 0u)) : ((((uint32_t)lane & 15u) << 3) | (uint32_t)T::kT1);
    lane_c2 = (((uint32_t)lane & 15u) << 3) | (uint32_t)T::kT2;
  }
  ItemAddr ad;
  item_addresses<REP>(qa, qb, lane_c, lane_c2, ad, 0u);
  int8_t* o = out + (size_t)(tile * 16 + n) * 64 * 8;
  // the A operand whose row 0 is one-hot at byte kb of the step (lane = row l & 15, bytes [16 (l >> 4), +16) of the 64)
  auto onehot = [&](int kb) -> i32x4 {
    i32x4 A = {0, 0, 0, 0};
    if (n == 0 && q == (kb >> 4)) {
      const int w = (kb & 15) >> 2, sh = 8 * (kb & 3);
      const int v = 1 << sh;
      A = i32x4{w == 0 ? v : 0, w == 1 ? v : 0, w == 2 ? v : 0, w == 3 ? v : 0};
    }
    return A;
  };
  if constexpr (T::kNib) {
    uint32_t raw[16];
    nib_decode(ad, raw);
    for (int s = 0; s < 4; ++s) {
      const i32x4 Br = {(int)raw[4 * s], (int)raw[4 * s + 1], (int)raw[4 * s + 2], (int)raw[4 * s + 3]};
      const i32x4 Bm = {Br.x & 0x0f0f0f0f, Br.y & 0x0f0f0f0f, Br.z & 0x0f0f0f0f, Br.w & 0x0f0f0f0f};
      for (int kb = 0; kb < 64; ++kb) {
        const i32x4 A = onehot(kb), z = {0, 0, 0, 0};
        const i32x4 r = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Br, z, 0, 0, 0);
        const i32x4 m = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Bm, z, 0, 0, 0);
        if (q == 0) {           // D row 0 of column n: the int8 byte 16 v_hi + lo_u, and lo_u = v_lo + 8
          const int vhi = (r.x - m.x) >> 4, vlo = m.x - 8;
          const int k0 = 256 * (s >> 1) + 64 * (kb >> 4) + 32 * (s & 1) + 8 * ((kb & 15) >> 2), b = kb & 3;
          o[(k0 >> 3) * 8 + 4 + b] = (int8_t)(2 * vhi + 1);
          o[(k0 >> 3) * 8 + b] = (int8_t)(2 * vlo + 1);
        }
      }
    }
  } else {
    i32x4 Bf[8];
    item_decode<false>(ad, Bf);
    for (int t = 0; t < 8; ++t)
      for (int kb = 0; kb < 64; ++kb) {
        const i32x4 A = onehot(kb), z = {0, 0, 0, 0};
        const i32x4 r = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, Bf[t], z, 0, 0, 0);
        if (q == 0) {
          const int k = 256 * (t >> 2) + 64 * (kb >> 4) + 16 * (t & 3) + (kb & 15);
          o[k] = (int8_t)r.x;                 // (k >> 3) * 8 + (k & 7) = k
        }
      }
  }
}

template <int REP>
int probe_launch(const void* grid, const void* codes, void* out, hipStream_t stream) {
  using T = Lds<REP>;
  const int lds = T::kNib ? kNibTableBytes : T::kT3;
  static DynLdsCache cache;
  if (ensure_dyn_lds(cache, reinterpret_cast<const void*>(decode_probe_kernel<REP>), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
  hipLaunchKernelGGL(decode_probe_kernel<REP>, dim3(8), dim3(512), lds, stream, reinterpret_cast<const uint64_t*>(grid),
                     reinterpret_cast<const uint4*>(codes), reinterpret_cast<int8_t*>(out));
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

}  // namespace
}  // namespace quip

extern "C" int quip_e8p_decode_probe(const void* grid, const void* codes, void* out, int32_t mode, quip_stream_t stream) {
  if (!grid || !codes || !out) return QUIP_ERR_NULL_POINTER;
  using namespace quip;
  switch (mode) {
    case 4: return probe_launch<4>(grid, codes, out, (hipStream_t)stream);
    case 16: return probe_launch<16>(grid, codes, out, (hipStream_t)stream);
    case 24: return probe_launch<24>(grid, codes, out, (hipStream_t)stream);
    case 32: return probe_launch<32>(grid, codes, out, (hipStream_t)stream);
    default: return QUIP_ERR_UNSUPPORTED;
  }
}
