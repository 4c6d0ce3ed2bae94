// E8P12 decode GEMV for gfx950 (bs=1 decode): y[n] = sum_k W[n,k] x[k],
// W = decode(Qidxs (n, k/8) int16).  Replaces the M=1 use of
// tinygemm_m16n8k16_chunk_kernel<..., BLayout_E8, ...> (origin_order.cu:388-555,
// 604-648), which pads M to 16 for mma.sync; here it is a plain wave64
// dot-product kernel, HBM-bound by design.
//
// Work decomposition (wave64, 16-byte coalesced index loads):
//   * a packed row is K/4 bytes; it is cut into J = ceil(K/4096) slices of 1 KiB,
//     one wave-wide global_load_dwordx4 each (lane = 8 codes = 64 weights);
//   * a workgroup owns a contiguous block of rows; wave (g, j) walks rows
//     g, g+G, ... of the block and always reads slice j, so the 64 x values a
//     lane multiplies with never change: they live in 32 VGPRs for the whole
//     kernel (no LDS traffic for x in the main loop);
//   * decode is table driven.  w_p = s_p * a_p + d with a = abs row (fp16, from
//     LDS table T1[code >> 8]), s = sign masks (LDS table T2[code & 255], XORed
//     onto the fp16 sign bits) and d = +-1/4 by the parity of the sign byte.
//     The +-1/4 is hoisted: sum_p w_p x_p = sum_p (s_p a_p) x_p + d * X,
//     X = sum of the 8 x values of the group (fp32, precomputed per lane);
//     per code: 2 v_perm (LDS addresses), 2 ds_read_b128, 4 v_xor, 4 v_dot2_f32_f16,
//     v_bcnt + v_lshl_or + v_fmac for the shift = 13 VALU ops per 8 weights;
//   * REP = 16 replicates every 16-byte table entry over the 16 slots of its own
//     256-byte bank row and lane l reads slot (l & 15): the four 16-lane service
//     groups of ds_read_b128 then never conflict, whatever the codes are
//     (MI355X LDS: 64 banks x 4 B, ds_read_b128 serviced in 4 groups of 16 lanes).
//     REP = 1 keeps compact 4 KiB tables (random ~3-way conflicts) for launches
//     too small to amortise the 128 KiB fill;
//   * per row one DPP wave reduction (6 v_add_f32_dpp), slices of a row are
//     summed through LDS at the end, fp32 -> fp16 RN store.
#include "quip_device.hip.h"
#include "quip_internal.h"

namespace quip {

static constexpr int kMaxPartials = 4096;  // floats of LDS for per-row slice partials

template <int REP>
struct GemvLds {
  static constexpr int kT1 = 0;
  static constexpr int kT2 = (REP == 16) ? 0x10000 : 0x1000;
  static constexpr int kPart = (REP == 16) ? 0x20000 : 0x2000;
  static constexpr int kXStage = (REP == 16) ? 0 : (kPart + kMaxPartials * 4);  // aliased with tables when REP==16
  static int bytes(int K) {
    int x_end = kXStage + K * 2;
    int t_end = kPart + kMaxPartials * 4;
    return x_end > t_end ? x_end : t_end;
  }
};

// LDS read by absolute byte address.  The kernel has no static __shared__, so the
// dynamic segment starts at LDS address 0 and table offsets are absolute; this
// keeps the v_perm-built addresses usable as they are (no base add per lookup).
typedef const __attribute__((address_space(3))) u32x4* lds_u4_ptr;
__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
  return *reinterpret_cast<lds_u4_ptr>((uintptr_t)addr);
}

// LDS addresses of the two table rows of one code (d holds two codes).
template <int REP, bool HIGH>
__device__ __forceinline__ void e8p_code_addr(uint32_t d, uint32_t lane_c, uint32_t& a1,
                                              uint32_t& a2) {
  if constexpr (REP == 16) {
    // address = table_base | idx << 8 | (lane & 15) << 4, built with one v_perm_b32:
    // result bytes {3,2,1,0} <- {0, lane_c.b2 or 0, d.byte(idx), lane_c.b0}
    a1 = __builtin_amdgcn_perm(d, lane_c, HIGH ? 0x0c0c0700u : 0x0c0c0500u);
    a2 = __builtin_amdgcn_perm(d, lane_c, HIGH ? 0x0c020600u : 0x0c020400u);
  } else {
    a1 = HIGH ? ((d >> 20) & 0xff0u) : ((d >> 4) & 0xff0u);
    a2 = (HIGH ? ((d >> 12) & 0xff0u) : ((d << 4) & 0xff0u)) | GemvLds<1>::kT2;
  }
}

// acc += <sign-applied abs row, x group> + (+-1/4) * X.  popcount(a2) has the
// parity of the sign byte up to the lane constant par_fix.
__device__ __forceinline__ float e8p_code_fma(const u32x4& a, const u32x4& m, uint32_t a2,
                                              const uint4& xg, float X, float acc,
                                              uint32_t par_fix) {
  acc = dot2(a.x ^ m.x, xg.x, acc);
  acc = dot2(a.y ^ m.y, xg.y, acc);
  acc = dot2(a.z ^ m.z, xg.z, acc);
  acc = dot2(a.w ^ m.w, xg.w, acc);
  const uint32_t cnt = __builtin_popcount(a2) + par_fix;
  const float delta = as_f32((cnt << 31) | 0x3e800000u);  // +-0.25
  return __builtin_fmaf(delta, X, acc);
}

template <int REP>
__device__ __forceinline__ float e8p_row_dot(const uint4& q, const uint4 (&xr)[8],
                                             const float (&X)[8], uint32_t lane_c,
                                             uint32_t par_fix) {
  uint32_t a1[8], a2[8];
  e8p_code_addr<REP, false>(q.x, lane_c, a1[0], a2[0]);
  e8p_code_addr<REP, true>(q.x, lane_c, a1[1], a2[1]);
  e8p_code_addr<REP, false>(q.y, lane_c, a1[2], a2[2]);
  e8p_code_addr<REP, true>(q.y, lane_c, a1[3], a2[3]);
  e8p_code_addr<REP, false>(q.z, lane_c, a1[4], a2[4]);
  e8p_code_addr<REP, true>(q.z, lane_c, a1[5], a2[5]);
  e8p_code_addr<REP, false>(q.w, lane_c, a1[6], a2[6]);
  e8p_code_addr<REP, true>(q.w, lane_c, a1[7], a2[7]);
  // LDS reads run two codes ahead of their use (ds_read latency ~64+ cycles)
  u32x4 ta[8], tm[8];
  ta[0] = lds_read16(a1[0]); tm[0] = lds_read16(a2[0]);
  ta[1] = lds_read16(a1[1]); tm[1] = lds_read16(a2[1]);
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c + 2 < 8) { ta[c + 2] = lds_read16(a1[c + 2]); tm[c + 2] = lds_read16(a2[c + 2]); }
    if (c & 1) acc1 = e8p_code_fma(ta[c], tm[c], a2[c], xr[c], X[c], acc1, par_fix);
    else       acc0 = e8p_code_fma(ta[c], tm[c], a2[c], xr[c], X[c], acc0, par_fix);
  }
  return acc0 + acc1;
}

template <int REP, int ROWS, int MAXT>
__global__ __launch_bounds__(MAXT) void e8p_gemv_m1_kernel(
    const uint4* __restrict__ W, const f16* __restrict__ x, f16* __restrict__ y,
    const uint64_t* __restrict__ grid, int N, int K, int J, int G, int rows_per_block) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = GemvLds<REP>;
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = wave % J, g = wave / J;
  const int slices = K >> 6;  // 16-byte pieces per packed row (= 64 weights each)
  const int lp = j * 64 + lane;
  const bool active = lp < slices;
  const int row0 = blockIdx.x * rows_per_block;
  const int row_end = min(N, row0 + rows_per_block);
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  const uint4* Wl = W + (active ? lp : 0);  // inactive lanes: valid address, x = 0

  // (0) first weight loads go out before anything else: their HBM latency
  //     covers the x staging and the table fill below.
  uint4 q[ROWS];
  int r = row0 + g;
#pragma unroll
  for (int i = 0; i < ROWS; ++i) {
    const int ri = r + i * G;
    q[i] = zero4;
    if (ri < row_end) q[i] = ld_nt_u4(Wl + (size_t)ri * slices);
  }

  // (1) stage x (K fp16) into LDS, transposed so that piece (c, lp) sits at slot
  //     c * slices + lp: the per-lane reads below are then bank-conflict free.
  {
    const uint4* xg = reinterpret_cast<const uint4*>(x);
    uint4* xs = reinterpret_cast<uint4*>(smem + L::kXStage);
    const int pieces = K >> 3;  // 16-byte pieces of x (8 halfs = one code group)
    for (int p = tid; p < pieces; p += nthreads) xs[(p & 7) * slices + (p >> 3)] = xg[p];
  }
  __syncthreads();
  uint4 xr[8];
  float X[8];
  {
    const uint4* xs = reinterpret_cast<const uint4*>(smem + L::kXStage);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      xr[c] = active ? xs[c * slices + lp] : zero4;
      const uint32_t ones = 0x3c003c00u;  // (1.0h, 1.0h)
      X[c] = dot2(xr[c].x, ones, dot2(xr[c].y, ones, dot2(xr[c].z, ones, dot2(xr[c].w, ones, 0.f))));
    }
  }
  if constexpr (REP == 16) __syncthreads();  // x staging aliases the tables

  // (2) build the decode tables in LDS from grid_packed_abs (2 KiB, L2 resident).
  if constexpr (REP == 16) {
    for (int t = tid; t < 1024; t += nthreads) {
      const int e = t & 255, rg = t >> 8;  // entry, replica group (4 slots each)
      const uint4 arow = e8p_abs_row_f16(grid[e]);
      const uint4 mrow = e8p_sign_masks((uint32_t)e);
      uint4* t1 = reinterpret_cast<uint4*>(smem + L::kT1 + e * 256 + rg * 64);
      uint4* t2 = reinterpret_cast<uint4*>(smem + L::kT2 + e * 256 + rg * 64);
#pragma unroll
      for (int s = 0; s < 4; ++s) { t1[s] = arow; t2[s] = mrow; }
    }
  } else {
    for (int e = tid; e < 256; e += nthreads) {
      reinterpret_cast<uint4*>(smem + L::kT1)[e] = e8p_abs_row_f16(grid[e]);
      reinterpret_cast<uint4*>(smem + L::kT2)[e] = e8p_sign_masks((uint32_t)e);
    }
  }
  __syncthreads();

  // lane constants of the v_perm address builder: byte0 = slot*16, byte2 = 0x01 (T2 base 0x10000)
  const uint32_t lane_c = (REP == 16) ? (((uint32_t)(lane & 15) << 4) | 0x00010000u) : 0u;
  // popcount(T2 address) == popcount(sign byte) + popcount(constant address bits)
  const uint32_t par_fix = (REP == 16) ? (__builtin_popcount(lane_c) & 1)
                                       : (__builtin_popcount((uint32_t)L::kT2) & 1);
  float* part = reinterpret_cast<float*>(smem + L::kPart);

  // (3) stream the rows.  ROWS independent load slots rotate: a slot is refilled
  //     as soon as its codes have been turned into LDS addresses, so ROWS-1 KiB-wide
  //     loads per wave are always in flight behind the one being decoded.  The
  //     steady-state loop has no conditional loads (hipcc would otherwise drain
  //     vmcnt(0) at every exec-mask join); the tail loop handles the last rows.
  for (; r + (2 * ROWS - 1) * G < row_end; r += ROWS * G) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const uint4 qi = q[i];
      const int ri = r + i * G;
      q[i] = ld_nt_u4(Wl + (size_t)(ri + ROWS * G) * slices);
      const float acc = e8p_row_dot<REP>(qi, xr, X, lane_c, par_fix);
      const float tot = wave_sum_to_lane63(acc);
      if (lane == 63) part[(ri - row0) * J + j] = tot;
    }
  }
  for (; r < row_end; r += ROWS * G) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const uint4 qi = q[i];
      const int ri = r + i * G;
      const int rn = ri + ROWS * G;
      if (rn < row_end) q[i] = ld_nt_u4(Wl + (size_t)rn * slices);
      if (ri < row_end) {
        const float acc = e8p_row_dot<REP>(qi, xr, X, lane_c, par_fix);
        const float tot = wave_sum_to_lane63(acc);
        if (lane == 63) part[(ri - row0) * J + j] = tot;
      }
    }
  }
  __syncthreads();

  // (4) sum the J slices of each row, round to fp16, coalesced store
  for (int t = tid; t < row_end - row0; t += nthreads) {
    float s = 0.f;
    for (int jj = 0; jj < J; ++jj) s += part[t * J + jj];
    y[row0 + t] = (f16)s;
  }
}

template <int REP, int ROWS, int MAXT>
static int launch_variant(const void* x, const void* qidxs, const void* grid, void* y, int n,
                          int k, int J, int G, int rpb, int nblocks, hipStream_t stream) {
  auto kern = e8p_gemv_m1_kernel<REP, ROWS, MAXT>;
  const int lds = GemvLds<REP>::bytes(k);
  static int configured_lds = 0;  // per instantiation; benign race (idempotent)
  if (lds > configured_lds) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return QUIP_ERR_LAUNCH;
    configured_lds = lds;
  }
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * G * J), lds, stream,
                     reinterpret_cast<const uint4*>(qidxs), reinterpret_cast<const f16*>(x),
                     reinterpret_cast<f16*>(y), reinterpret_cast<const uint64_t*>(grid), n, k, J,
                     G, rpb);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

bool e8p_gemv_m1_supported(int n, int k) {
  if (k % 64 != 0 || k < 64) return false;
  const int J = (k / 64 + 63) / 64;
  return J <= 16 && k * 2 <= 96 * 1024;
}

// tune.rep: 0 = auto, 1 or 16.  tune.rows: 0 = auto, 1/2/4.  tune.blocks: 0 = auto.
int e8p_gemv_m1_launch(const void* x, const void* qidxs, const void* grid, void* y, int n, int k,
                       const GemvTune& tune, hipStream_t stream) {
  if (!e8p_gemv_m1_supported(n, k)) return QUIP_ERR_UNSUPPORTED;
  const int slices = k / 64;
  const int J = (slices + 63) / 64;
  const int ncu = device_cu_count();
  int nblocks = tune.blocks > 0 ? tune.blocks : ncu;
  int rpb = (n + nblocks - 1) / nblocks;
  int G = tune.waves_g > 0 ? tune.waves_g : (16 / J > 0 ? 16 / J : 1);
  if (G > rpb) G = rpb;
  if (G * J > 16) G = 16 / J;
  if (G < 1) G = 1;
  while (rpb * J > kMaxPartials) { nblocks *= 2; rpb = (n + nblocks - 1) / nblocks; }
  nblocks = (n + rpb - 1) / rpb;
  const long long bytes = (long long)n * k / 4;
  int rep = tune.rep ? tune.rep : (bytes >= (8ll << 20) ? 16 : 1);
  int rows = tune.rows ? tune.rows : 4;
  const int per_wave = (rpb + G - 1) / G;
  if (!tune.rows) rows = per_wave >= 4 ? 4 : (per_wave >= 2 ? 2 : 1);
  const bool big = G * J * 64 > 512;  // >8 waves per workgroup: 128-VGPR budget
#define QUIP_GEMV_CASE(R, RW)                                                                   \
  if (rep == R && rows == RW)                                                                   \
    return big ? launch_variant<R, RW, 1024>(x, qidxs, grid, y, n, k, J, G, rpb, nblocks, stream) \
               : launch_variant<R, RW, 512>(x, qidxs, grid, y, n, k, J, G, rpb, nblocks, stream);
  QUIP_GEMV_CASE(16, 4) QUIP_GEMV_CASE(16, 2) QUIP_GEMV_CASE(16, 1)
  QUIP_GEMV_CASE(1, 4) QUIP_GEMV_CASE(1, 2) QUIP_GEMV_CASE(1, 1)
#undef QUIP_GEMV_CASE
  return QUIP_ERR_UNSUPPORTED;
}

}  // namespace quip
