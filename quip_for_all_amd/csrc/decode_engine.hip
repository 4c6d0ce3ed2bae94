// Persistent decode engine for gfx950, stage 1: the MLP half of a decoder block in ONE launch.
//
//   z_d = W_down . U_in( SU_d (.) silu(g) (.) u ),   g = SV_g (.) U_out(W_gate . x_g),  u = SV_u (.) U_out(W_up . x_u)
//
// i.e. QuantLinear.forward of gate_proj / up_proj from their digit planes on (qlinear.py:103-114), the SiLU
// product of the decoder block, and QuantLinear.forward of down_proj up to its raw product (qlinear.py:90-103)
// -- the work of four launches of the stage-wise decode step (GEMV[gate, up], output transforms, input transform
// of down, GEMV[down]; reference kernels: origin_order.cu:388-555 at m = 1, quant.py:72-88).
//
// Why one launch: every launch of the stage-wise step rebuilds 96-128 KB of decode tables per workgroup, starts
// its weight stream only after its input exists, and pays a drain + a cold prologue; the weights depend on
// nothing.  Here the tables are built once, the codes of down_proj are requested while gate / up are still being
// multiplied (they wait in registers), and the two transforms between the products run as a distributed
// computation with two hand-offs through device-coherent memory (engine_sync.hip.h) instead of two launches.
//
// Workgroup w of the L = n_ffn / K workgroups (one per CU; n_ffn = K * L, K x K the orthogonal factor of
// get_hadK, quant.py:26-39; Llama-2-7B: 11008 = 43 x 256):
//   (1) GEMV gate / up for COLUMN w of the (K, L) view of the outputs: rows k * L + w, k = 0..K-1, of both
//       matrices (exact integer sums, the decode and mapping of e8p_gemv_mfma.hip; one-shot slots).
//   (2) z -> fp16 (the reference's mm output type), then the K-mix of its column on the spot:
//       t[k'] = sum_k had[k'][k] z[k]  --  (H (x) H_L) = (I (x) H_L)(H (x) I), and H (x) I is column local.
//       The 2 K values go to the K row owners (hand-off 1: 8-byte {value, tag} granules).
//   (3) row owner r < K (one wave): length-L transforms of row r of gate and up, SV, fp16 rounding (the output
//       type of the module), silu(g) * u * SU_d, length-L transform of the result = row r of (I (x) H_L) applied to
//       down's input; published as granules (hand-off 2).
//   (4) every workgroup gathers the K x L rows and finishes down's input transform, (H^T (x) I), on the matrix
//       cores (fp32 MFMA), takes the exact maximum, and writes the digit planes straight into its LDS.
//   (5) GEMV down for rows [w * rpw, +rpw) from the registers requested in (1); z_d -> fp16.
// Numerics: the same individually rounded fp32 operations as the stand-alone kernels up to the order of the two
// commuting factors on down's input side (there: K-mix, then length-L transform; here the reverse, which is the
// reference's order, quant.py:81-84) and the exact maximum instead of the norm bound for the block exponent;
// results agree with the stage-wise path to the last fp16 bit except where a value sits on a rounding boundary
// (tests/test_gpu_engine.py states the bound).  Everything integer stays exact.
//
// Liveness: all L workgroups must be resident at once (L <= #CUs, one workgroup per CU by its LDS footprint);
// every spin is bounded (engine_sync.hip.h) and a launch that gives up leaves a code in ctl[1].
#include "e8p_gemv_core.hip.h"
#include "engine_sync.hip.h"
#include <cstdlib>

namespace quip {

namespace {

using esync::u32x4_t;

struct FfnArgs {
  const uint4* Wg;
  const uint4* Wu;
  const uint4* Wd;
  const uint8_t* planes_g;   // [3][Kp_in] digits + shift word (output of the input-transform launch)
  const uint8_t* planes_u;
  const f16* had3;           // gate.had_right, up.had_right (K x K row major, each padded to KKP = K * K rounded up to 8
                             // elements), then down.had_left TRANSPOSED and zero padded to [KP16][KP16], KP16 = K rounded up to 16
  const f16* sv_g;           // [n_ffn]
  const f16* sv_u;
  const f16* su_d;           // [n_ffn]
  f16* zd;                   // [hidden]: raw product of down_proj
  const uint64_t* grid;      // grid_packed_abs
  uint64_t* inbox;           // [K][2][L] granules
  uint64_t* frow;            // [L][KP16] granules: element (k, j) of the transformed rows at j * KP16 + k
  uint32_t* ctl;             // [0] generation (epoch of the last finished launch), [1] error code
  uint64_t* dbg;             // optional s_memtime stamps, 16 per workgroup
  float out_scale;           // 1 / sqrt(L)                    (gate / up output side)
  float in_scale;            // wscale_down / sqrt(L)          (down input side)
  int hidden;                // k of gate / up = n of down
};

constexpr int kEngWaves = 8;
constexpr int kEngThreads = 64 * kEngWaves;

// LDS map of the engine (REP: table copies as in Lds<REP>)
template <int REP, int K, int LOGL>
struct EngLds {
  using T = Lds<REP>;
  static constexpr int L = 1 << LOGL;
  static constexpr int RB = (K + 15) / 16;                 // row blocks of a column's K rows
  static constexpr int kAccRows = 2 * RB * 16 + 16;        // gate, up, down
  static constexpr int kAcc = T::kAcc;
  static constexpr int KKP = (K * K + 7) & ~7;             // elements of a padded K x K factor
  static constexpr int KP16 = (K + 15) & ~15;              // K rounded up to the k step of the fp16 MFMA
  static constexpr int kHad = kAcc + kAccRows * 16;        // fp16: gate [KKP], up [KKP] (row major), down TRANSPOSED [KP16][KP16]
  static constexpr int kHadElems = 2 * KKP + KP16 * KP16;
  static constexpr int kHadBytes = kHadElems * 2;
  static constexpr int kZ = kHad + kHadBytes;              // float [2][RB * 16]: z of this column
  static constexpr int kRed = kZ + 2 * RB * 16 * 4;        // 64 floats of reduction scratch
  static constexpr int kVec = kRed + 256;                  // row owner: SV_gate, SV_up, SU_down of its row (fp16 [3][L])
  static constexpr int kR = kVec + ((3 * L * 2 + 15) & ~15);   // region R: planes of gate / up, then the gathered rows, then down's planes
  // gathered rows, transposed: dwords (fp16 hi | fp16 lo << 16) [L][KP16], k contiguous (one 16-byte read = the four k of
  // an MFMA operand, hi and lo)
  static constexpr int kFBytes = L * KP16 * 4;
  static constexpr int KpD = (K * L + 511) & ~511;         // digits of down's input
  static constexpr int kPlaneD = (KpD / 256) * 272;        // a plane of down: 16 bytes of padding per 256 digits
  static int bytes(int kp_in) {
    int r = 2 * 3 * kp_in;
    r = r > kFBytes ? r : kFBytes;
    r = r > 3 * kPlaneD ? r : 3 * kPlaneD;
    return kR + r;
  }
};

// NGU / ND: items per wave of the gate + up product / of down's (static slot registers; the launcher picks the
// instantiation that covers the shape)
template <int REP, int K, int LOGL, int NGU, int ND, int DEPTH>
__global__ __launch_bounds__(kEngThreads) void ffn_engine_kernel(FfnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using E = EngLds<REP, K, LOGL>;
  using T = Lds<REP>;
  constexpr int L = E::L, RB = E::RB;
  constexpr int NS = NGU + ND;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = blockIdx.x;                       // column of the (K, L) view; row block of down
  const int n = lane & 15, q = lane >> 4;
#define ENG_STAMP(i) do { if (a.dbg && tid == 0) a.dbg[w * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  ENG_STAMP(0);

  const int k_in = a.hidden;
  const int kp_in = (k_in + 511) & ~511;
  const int J_in = kp_in >> 9;
  const int row_u4_in = k_in >> 6;
  const int n_gu_items = 2 * RB * J_in;
  constexpr int n_ffn = K * L;
  constexpr int J_d = E::KpD >> 9;
  constexpr int row_u4_d = n_ffn >> 6;
  const int rpw = a.hidden / L;                   // rows of down per workgroup (<= 16)

  // ---- item -> address --------------------------------------------------------------------------------------
  // unified item list of a wave: i < NGU: item (i * 8 + wave) of gate / up, then i - NGU: slice (i - NGU) * 8 + wave of down
  auto item_ptr = [&](int i, int j) -> const uint4* {
    if (i < NGU) {
      const int it = i * kEngWaves + wave;
      const int itc = it < n_gu_items ? it : 0;
      const int rbg = itc / J_in, s = itc - rbg * J_in;
      const int m = rbg / RB, rb = rbg - m * RB;
      int kr = rb * 16 + n;
      kr = kr < K ? kr : K - 1;
      int off = s * 8 + q + 4 * j;
      off = off < row_u4_in ? off : row_u4_in - 1;     // past the row: its digits are zero
      const uint4* W = m ? a.Wu : a.Wg;
      return W + ((size_t)(kr * L + w) * row_u4_in + off);
    }
    const int s0 = (i - NGU) * kEngWaves + wave;
    const int s = s0 < J_d ? s0 : 0;
    int r = n < rpw ? n : rpw - 1;
    int off = s * 8 + q + 4 * j;
    off = off < row_u4_d ? off : row_u4_d - 1;
    return a.Wd + ((size_t)(w * rpw + r) * row_u4_d + off);
  };

  // ---- (0) requests: table source, digit planes of gate / up, the first weight slots --------------------------
  // row owners (w < K) keep SV_gate / SV_up / SU_down of their row in LDS: requested first (every workgroup issues the
  // load, so the counted waits below are the same everywhere), cold in HBM at this point and needed in stage (4)
  constexpr int VPIECES = 3 * L / 8;
  static_assert(VPIECES <= kEngThreads, "one vector piece per thread");
  u32x4 vr;
  {
    const int i = tid < VPIECES ? tid : 0;
    const int vsel = i / (L / 8), piece = i - vsel * (L / 8);
    const f16* vsrc = (vsel == 0 ? a.sv_g : (vsel == 1 ? a.sv_u : a.su_d)) + (size_t)(w < K ? w : 0) * L + piece * 8;
    asm_load16(vr, reinterpret_cast<const uint4*>(vsrc));
  }
  u32x2 tsrc;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(tsrc) : "v"(table_source_ptr(a.grid, lane, wave)) : "memory");
  uint32_t gen;
  esync::ld4(gen, a.ctl);
  constexpr int HPIECES = E::kHadElems / 8;       // 16-byte pieces of the three K x K factors
  constexpr int XH = (HPIECES + kEngThreads - 1) / kEngThreads;
  u32x4 hr[XH];
#pragma unroll
  for (int j = 0; j < XH; ++j) {
    const int i = tid + j * kEngThreads;
    asm_load16(hr[j], reinterpret_cast<const uint4*>(a.had3) + (i < HPIECES ? i : 0));
  }
  constexpr int XR = 6;                            // 16-byte plane pieces per thread (2 x 3 x kp_in <= 48 KB)
  const int ppieces = 3 * (kp_in >> 4);
  const int xpieces = 2 * ppieces;
  u32x4 xr[XR];
  const int rot = (int)(((uint32_t)w * 613u) % (uint32_t)xpieces);
#pragma unroll
  for (int j = 0; j < XR; ++j) {
    const int i = tid + j * kEngThreads;
    int ic = i < xpieces ? i + rot : 0;
    ic = ic >= xpieces ? ic - xpieces : ic;
    const int p = ic >= ppieces ? 1 : 0;
    const uint8_t* src = p ? a.planes_u : a.planes_g;
    asm_load16(xr[j], reinterpret_cast<const uint4*>(src) + (ic - p * ppieces));
  }
  u32x4 qa[NS], qb[NS];
#pragma unroll
  for (int i = 0; i < (DEPTH < NS ? DEPTH : NS); ++i) {
    asm_load16_nt(qa[i], item_ptr(i, 0));
    asm_load16_nt(qb[i], item_ptr(i, 1));
  }
  ENG_STAMP(1);

  // ---- (1) accumulators, tables, had factors, planes ----------------------------------------------------------
  int* accs = reinterpret_cast<int*>(smem + E::kAcc);
  for (int i = tid; i < E::kAccRows * 4; i += kEngThreads) accs[i] = 0;
  constexpr int kAhead = 2 * (DEPTH < NS ? DEPTH : NS);
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tsrc) : "n"(1 + XH + XR + kAhead) : "memory");
  fill_tables_from_lane<REP>(smem, tsrc, lane, wave);
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(gen) : "n"(XH + XR + kAhead) : "memory");
  const uint32_t epoch = (uint32_t)__builtin_amdgcn_readfirstlane((int)gen) + 1u;
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(XR + kAhead) : "memory");
  esync::own(vr);     // older than everything waited for so far
  if (tid < VPIECES) *reinterpret_cast<u32x4*>(smem + E::kVec + tid * 16) = vr;
#pragma unroll
  for (int j = 0; j < XH; ++j) {
    esync::own(hr[j]);
    const int i = tid + j * kEngThreads;
    if (i < HPIECES) *reinterpret_cast<u32x4*>(smem + E::kHad + i * 16) = hr[j];
  }
  asm_wait_vmcnt_x<kAhead>(xr[0], xr[1], xr[2], xr[3], xr[4], xr[5]);
#pragma unroll
  for (int j = 0; j < XR; ++j) {
    const int i = tid + j * kEngThreads;
    int ic = i + rot;
    ic = ic >= xpieces ? ic - xpieces : ic;
    if (i < xpieces) *reinterpret_cast<u32x4*>(smem + E::kR + ic * 16) = xr[j];
  }
  // the shift words of the planes (uniform addresses: scalar loads)
  const int sh_g = *reinterpret_cast<const int*>(a.planes_g + (size_t)3 * kp_in);
  const int sh_u = *reinterpret_cast<const int*>(a.planes_u + (size_t)3 * kp_in);
  __syncthreads();
  ENG_STAMP(2);

  const uint32_t lane_c = (T::kRep1 == 32) ? ((((uint32_t)lane & 31u) << 3) | 0x00010000u)
                                           : ((((uint32_t)lane & 15u) << 3) | (uint32_t)T::kT1);
  const uint32_t lane_c2 = (((uint32_t)lane & 15u) << 3) | (uint32_t)T::kT2;

  // ---- (2) GEMV gate / up; the slots of down are requested on the way ------------------------------------------
  const uint32_t xlane_in = (uint32_t)E::kR + (uint32_t)min(n, 2) * (uint32_t)kp_in + (uint32_t)q * 64u;
#pragma unroll
  for (int i = 0; i < NGU; ++i) {
    // requested so far: slots 0 .. min(i + DEPTH, NS) - 1
    constexpr int dummy = 0;
    (void)dummy;
    const int ahead = (i + DEPTH < NS ? i + DEPTH : NS) - 1 - i;      // newer slots outstanding once slot i has landed
    switch (ahead) {
      case 0: asm_wait_vmcnt<0>(qa[i], qb[i]); break;
      case 1: asm_wait_vmcnt<2>(qa[i], qb[i]); break;
      case 2: asm_wait_vmcnt<4>(qa[i], qb[i]); break;
      case 3: asm_wait_vmcnt<6>(qa[i], qb[i]); break;
      default: asm_wait_vmcnt<8>(qa[i], qb[i]); break;
    }
    if (i + DEPTH < NS) {
      asm_load16_nt(qa[(i + DEPTH) < NS ? (i + DEPTH) : 0], item_ptr(i + DEPTH, 0));
      asm_load16_nt(qb[(i + DEPTH) < NS ? (i + DEPTH) : 0], item_ptr(i + DEPTH, 1));
    }
    const int it = i * kEngWaves + wave;
    if (it < n_gu_items) {   // wave uniform
      const int rbg = it / J_in, s = it - rbg * J_in;
      const int m = rbg / RB;
      ItemAddr ad;
      item_addresses<REP>(qa[i], qb[i], lane_c, lane_c2, ad, 0u);
      const i32x4 acc = item_mfma(ad, xlane_in + (uint32_t)(m * 3 * kp_in + s * 512));
      if (q == 0) {
        int* dst = accs + (rbg * 16 + n) * 4;
        __hip_atomic_fetch_add(dst + 0, acc.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 1, acc.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 2, acc.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  }
  // slots of down not requested yet (DEPTH < ND)
#pragma unroll
  for (int i = NGU + DEPTH; i < NS; ++i) {
    if (i >= DEPTH) {
      asm_load16_nt(qa[i < NS ? i : 0], item_ptr(i, 0));
      asm_load16_nt(qb[i < NS ? i : 0], item_ptr(i, 1));
    }
  }
  // every request of down has to have landed before anything else touches the vector memory queue
#pragma unroll
  for (int i = NGU; i < NS; ++i) asm_wait_vmcnt<0>(qa[i], qb[i]);
  ENG_STAMP(3);
  __syncthreads();
  ENG_STAMP(4);

  // ---- (3) z of this column -> fp16 -> K-mix -> granules to the row owners --------------------------------------
  float* zbuf = reinterpret_cast<float*>(smem + E::kZ);
  if (tid < 2 * RB * 16) {
    const int m = tid / (RB * 16);
    const int* s3 = accs + tid * 4;
    const float f = __builtin_fmaf((float)s3[0], 65536.f, __builtin_fmaf((float)s3[1], 256.f, (float)s3[2]));
    const f16 z = (f16)(f * unscale_of(m ? sh_u : sh_g, 2));
    zbuf[tid] = (float)z;
  }
  __syncthreads();
  {
    // output (m, k') = thread quad o = tid >> 2; lane p of the quad sums k = p, p + 4, ...; the quad adds its four chains
    const int o = tid >> 2, part = tid & 3;
    const int m = o >> 6, kq = o & 63;
    const bool live = kq < K;
    const f16* hs = reinterpret_cast<const f16*>(smem + E::kHad) + m * E::KKP + (live ? kq : 0) * K;
    const float* zz = zbuf + m * RB * 16;
    float t = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < (K + 3) / 4; ++k4) {
      const int k = 4 * k4 + part;
      if (k < K) t = __builtin_fmaf((float)hs[k], zz[k], t);
    }
    t += __shfl_xor(t, 1, 64);
    t += __shfl_xor(t, 2, 64);
    if (live && part == 0) esync::st_granule(a.inbox + ((size_t)(kq * 2 + m) * L + w), as_u32(t), epoch);
  }
  ENG_STAMP(5);

  // ---- (4) row owners: the length-L transforms of row w of gate and up, the SiLU product, down's length-L ------
  constexpr int TPR = L / 16;                      // lanes per row (16 elements each)
  static_assert(TPR >= 1 && TPR <= 16, "row transforms run inside 16-lane groups");
  if (w < K && wave == 0) {
    const int m = (lane >> 4) & 1, t = lane & 15;
    const bool active = lane < 32 && t < TPR;
    const uint64_t* src = a.inbox + ((size_t)(w * 2 + m) * L + (active ? t : 0) * 16);
    u32x4_t g[8];
    uint32_t spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < 8; ++j) esync::ld16(g[j], src + 2 * j);
      esync::drain();
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        esync::own(g[j]);
        ok = ok && g[j].y == epoch && g[j].w == epoch;
      }
      if (esync::spin_step(ok || !active, spins, a.ctl + 1, 0x1000u + (uint32_t)w)) break;
    }
    if (a.dbg && lane == 0) a.dbg[w * 16 + 12] = __builtin_amdgcn_s_memtime();
    float v[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[2 * j] = as_f32(g[j].x);
      v[2 * j + 1] = as_f32(g[j].z);
    }
    had::fht16_lanes<LOGL>(v, t);
    // output side of gate / up: fp16( (v * scale) * SV ), element (w, 16 t + r) of the (K, L) view
    const f16* vecs = reinterpret_cast<const f16*>(smem + E::kVec);
    const f16* sv = vecs + m * L + (active ? t : 0) * 16;
    float o[16];
    {
      float svf[16];
      had::unpack8(*reinterpret_cast<const uint4*>(sv), svf);
      had::unpack8(*reinterpret_cast<const uint4*>(sv + 8), svf + 8);
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = (float)had::out_elem(v[r], a.out_scale, true, svf[r], false, 0.f, false, 0.f);
    }
    // lanes of matrix 0 (gate) fetch u from the lane 16 above; e = (u * silu(g)) * SU_d
    float e[16];
    {
      float suf[16];
      const f16* su = vecs + 2 * L + (active ? t : 0) * 16;
      had::unpack8(*reinterpret_cast<const uint4*>(su), suf);
      had::unpack8(*reinterpret_cast<const uint4*>(su + 8), suf + 8);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float u = __shfl(o[r], (lane + 16) & 63, 64);
        e[r] = had::fmul(had::fmul(u, had::silu(o[r])), suf[r]);
      }
    }
    had::fht16_lanes<LOGL>(e, t);
    if (lane < 16 && t < TPR) {
      // element (k = w, j = 16 t + r) as fp16 hi + lo of the prescaled value (exact power of two: the unnormalised
      // length-L transform stays inside the fp16 range), stored where the consumers' operand reads want it
      constexpr float kPre = 1.f / (float)(1 << ((LOGL + 1) / 2));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = e[r] * kPre;
        const f16 hi = (f16)v;
        const f16 lo = (f16)(v - (float)hi);
        const uint32_t pair = (uint32_t)__builtin_bit_cast(uint16_t, hi) | ((uint32_t)__builtin_bit_cast(uint16_t, lo) << 16);
        esync::st_granule(a.frow + ((size_t)(16 * t + r) * E::KP16 + w), pair, epoch);
      }
    }
  }
  ENG_STAMP(6);

  // B fragments of stage (6) (had_d^T, already in LDS): read now, they do not depend on the hand-off.
  // v_mfma_f32_16x16x16_f16: A[row = l & 15][k = 4 (l >> 4) + i], B[k = 4 (l >> 4) + i][col = l & 15]
  typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
  constexpr int KSTEPS = E::KP16 / 16;
  f16x4 bfr[KSTEPS][RB];
  {
    const f16* hdT = reinterpret_cast<const f16*>(smem + E::kHad) + 2 * E::KKP;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
      for (int ct = 0; ct < RB; ++ct)
        bfr[s][ct] = *reinterpret_cast<const f16x4*>(hdT + (16 * ct + n) * E::KP16 + 16 * s + 4 * q);
  }
  // ---- (5) gather the K rows (every workgroup): the value fields go to LDS as they are --------------------------------
  {
    constexpr int KPAIRS = (K + 1) / 2;            // 16-byte pieces (2 granules = k, k + 1) per column j
    constexpr int PIECES = L * KPAIRS;
    constexpr int NP = (PIECES + kEngThreads - 1) / kEngThreads;
    uint32_t* ft = reinterpret_cast<uint32_t*>(smem + E::kR);
    // k = 2 KPAIRS .. KP16 - 1 of every column: zero (the region held digit bytes; B is zero there, but 0 * NaN is not)
    if constexpr (E::KP16 > 2 * KPAIRS) {
      for (int j = tid; j < L; j += kEngThreads)
#pragma unroll
        for (int k = 2 * KPAIRS; k < E::KP16; ++k) ft[j * E::KP16 + k] = 0u;
    }
    // wait on ONE granule per row (the last column its owner stores) with one wave; sweeping all K L granules while
    // they are still being produced would put 256 x 8 K L bytes per pass on the fabric the producers need
    if (wave == 0) {
      uint32_t spins0 = 0;
      const uint64_t* last = a.frow + ((size_t)(L - 1) * E::KP16 + (lane < K ? lane : 0));
      for (;;) {
        esync::u32x2_t f;
        esync::ld8(f, last);
        esync::drain();
        esync::own(f);
        if (esync::spin_step(f.y == epoch, spins0, a.ctl + 1, 0x3000u + (uint32_t)w)) break;
      }
    }
    __syncthreads();
    if (a.dbg && tid == 0) a.dbg[w * 16 + 13] = __builtin_amdgcn_s_memtime();
    u32x4_t p[NP];
    uint32_t spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int i = tid + kEngThreads * j;
        const int ic = i < PIECES ? i : 0;
        const int col = ic / KPAIRS, kp = ic - col * KPAIRS;
        esync::ld16(p[j], a.frow + ((size_t)col * E::KP16 + 2 * kp));
      }
      esync::drain();
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        esync::own(p[j]);
        const int i = tid + kEngThreads * j;
        const int ic = i < PIECES ? i : 0;
        const int kp = ic % KPAIRS;
        ok = ok && p[j].y == epoch && (2 * kp + 1 >= K || p[j].w == epoch);
      }
      if (esync::spin_step(ok, spins, a.ctl + 1, 0x2000u + (uint32_t)w)) break;
    }
    if (a.dbg && tid == 0) a.dbg[w * 16 + 14] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = tid + kEngThreads * j;
      if (i < PIECES) {
        const int col = i / KPAIRS, kp = i - col * KPAIRS;
        *reinterpret_cast<uint2*>(ft + col * E::KP16 + 2 * kp) = make_uint2(p[j].x, (2 * kp + 1 < K) ? p[j].z : 0u);
      }
    }
  }
  __syncthreads();
  ENG_STAMP(7);

  // ---- (6) (H^T (x) I) on the matrix cores: D[j][k'] = sum_k f[k][j] had_d[k][k'], f = hi + lo (fp16 x fp16 products are
  //      exact in the fp32 accumulator; had_d is fp16 in the checkpoint).  D[row = 4 (l >> 4) + i][col = l & 15]
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int JT = L / 16;                                   // row tiles of D (columns j of the view)
  constexpr int JTW = (JT + kEngWaves - 1) / kEngWaves;       // per wave
  f32x4 acc[JTW][RB];
#pragma unroll
  for (int jt = 0; jt < JTW; ++jt)
#pragma unroll
    for (int ct = 0; ct < RB; ++ct) acc[jt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const uint32_t* ft = reinterpret_cast<const uint32_t*>(smem + E::kR);
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
      for (int jt = 0; jt < JTW; ++jt) {
        const int tile = wave + jt * kEngWaves;
        const u32x4 d = *reinterpret_cast<const u32x4*>(ft + (16 * (tile < JT ? tile : 0) + n) * E::KP16 + 16 * s + 4 * q);
        // four (hi | lo << 16) dwords -> the hi and the lo operand (4 halves each)
        const uint2 h2 = make_uint2(__builtin_amdgcn_perm(d.y, d.x, 0x05040100u), __builtin_amdgcn_perm(d.w, d.z, 0x05040100u));
        const uint2 l2 = make_uint2(__builtin_amdgcn_perm(d.y, d.x, 0x07060302u), __builtin_amdgcn_perm(d.w, d.z, 0x07060302u));
        const f16x4 ah = __builtin_bit_cast(f16x4, h2), al = __builtin_bit_cast(f16x4, l2);
#pragma unroll
        for (int ct = 0; ct < RB; ++ct) {
          acc[jt][ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bfr[s][ct], acc[jt][ct], 0, 0, 0);
          acc[jt][ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bfr[s][ct], acc[jt][ct], 0, 0, 0);
        }
      }
    }
  }
  const float in_scale = a.in_scale * (float)(1 << ((LOGL + 1) / 2));   // undoes the prescale (exact)
  // exact maximum of |scale * x| over the whole vector -> block exponent
  float mx = 0.f;
#pragma unroll
  for (int jt = 0; jt < JTW; ++jt)
#pragma unroll
    for (int ct = 0; ct < RB; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float m = fabsf(had::fmul(acc[jt][ct][i], in_scale));
        mx = fmaxf(mx, m == m ? m : __builtin_inff());
      }
  float* red = reinterpret_cast<float*>(smem + E::kRed);
  const float bound = had::block_reduce(mx, true, red, tid, kEngThreads);   // leading barrier: everyone is done reading the rows
  const int sh_d = had::shift_for(bound);
  ENG_STAMP(8);
  // digit planes of down's input, plane stride kPlaneD, 16 bytes of padding per 256 digits; this lane holds the
  // four consecutive digits j = 16 tile + 4 q + (0..3) of row k' = 16 ct + n
  {
    uint8_t* pl = reinterpret_cast<uint8_t*>(smem + E::kR);
    const float s2 = had::fmul(in_scale, as_f32((uint32_t)(sh_d + 127) << 23));
#pragma unroll
    for (int jt = 0; jt < JTW; ++jt) {
      const int tile = wave + jt * kEngWaves;
#pragma unroll
      for (int ct = 0; ct < RB; ++ct) {
        const int kc = 16 * ct + n;
        int X[4], X1[4], H[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          X[i] = (int)__builtin_rintf(had::fmul(acc[jt][ct][i], s2));
          X1[i] = (X[i] + 128) >> 8;
          H[i] = (X1[i] + 128) >> 8;
        }
        if (tile < JT && kc < K) {
          const int kk = kc * L + 16 * tile + 4 * q;
          const int off = (kk >> 8) * 272 + (kk & 255);
          *reinterpret_cast<uint32_t*>(pl + off) = had::low_bytes4(H[0], H[1], H[2], H[3]);
          *reinterpret_cast<uint32_t*>(pl + E::kPlaneD + off) = had::low_bytes4(X1[0], X1[1], X1[2], X1[3]);
          *reinterpret_cast<uint32_t*>(pl + 2 * E::kPlaneD + off) = had::low_bytes4(X[0], X[1], X[2], X[3]);
        }
      }
    }
    // the k padding [K L, KpD) reads as zero digits
    for (int i = n_ffn + 4 * tid; i < E::KpD; i += 4 * kEngThreads) {
      const int off = (i >> 8) * 272 + (i & 255);
#pragma unroll
      for (int d = 0; d < 3; ++d) *reinterpret_cast<uint32_t*>(pl + d * E::kPlaneD + off) = 0u;
    }
  }
  __syncthreads();
  ENG_STAMP(9);

  // ---- (7) GEMV down from the registers -------------------------------------------------------------------------
  const uint32_t xlane_d = (uint32_t)E::kR + (uint32_t)min(n, 2) * (uint32_t)E::kPlaneD + (uint32_t)q * 64u;
#pragma unroll
  for (int i = 0; i < ND; ++i) {
    const int s = i * kEngWaves + wave;
    if (s < J_d) {   // wave uniform
      ItemAddr ad;
      item_addresses<REP>(qa[NGU + i], qb[NGU + i], lane_c, lane_c2, ad, 0u);
      const i32x4 d4 = item_mfma<272>(ad, xlane_d + (uint32_t)(s * 544));
      if (q == 0) {
        int* dst = accs + (2 * RB * 16 + n) * 4;
        __hip_atomic_fetch_add(dst + 0, d4.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 1, d4.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 2, d4.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  ENG_STAMP(10);
  if (tid < rpw) {
    const int* s3 = accs + (2 * RB * 16 + tid) * 4;
    const float f = __builtin_fmaf((float)s3[0], 65536.f, __builtin_fmaf((float)s3[1], 256.f, (float)s3[2]));
    a.zd[w * rpw + tid] = (f16)(f * unscale_of(sh_d, 2));
  }
  // the launch is over for every workgroup that got here: they all passed hand-off 2, so nobody still reads ctl[0]
  if (w == 0 && tid == 0) esync::st_word(a.ctl, epoch);
  ENG_STAMP(11);
#undef ENG_STAMP
}

template <int REP, int K, int LOGL, int NGU, int ND>
int launch_ffn(const FfnArgs& a, hipStream_t stream) {
  using E = EngLds<REP, K, LOGL>;
  auto kern = ffn_engine_kernel<REP, K, LOGL, NGU, ND, 2>;
  const int kp_in = (a.hidden + 511) & ~511;
  const int lds = E::bytes(kp_in);
  if (lds > 160 * 1024) return QUIP_ERR_UNSUPPORTED;
  static DynLdsCache configured;
  if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
  static ResidencyCache resident;
  if (!persistent_grid_fits(resident, reinterpret_cast<const void*>(kern), kEngThreads, lds, E::L)) return QUIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3(E::L), dim3(kEngThreads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

// the (K, log2 L, hidden) combinations with an instantiation
struct FfnShape { int K, logL, ngu, nd; };
bool ffn_shape_of(int hidden, int n_ffn, int K, FfnShape& s) {
  if (K < 2 || n_ffn % K != 0) return false;
  const int L = n_ffn / K;
  if (L < 16 || L > 256 || (L & (L - 1)) != 0) return false;
  int logL = 0;
  while ((1 << logL) < L) ++logL;
  if (hidden % L != 0 || hidden / L > 16 || hidden < 128 || hidden % 128 != 0 || n_ffn % 128 != 0) return false;
  const int kp_in = (hidden + 511) & ~511, kp_d = (n_ffn + 511) & ~511;
  const int rb = (K + 15) / 16;
  s.K = K; s.logL = logL;
  s.ngu = (2 * rb * (kp_in >> 9) + kEngWaves - 1) / kEngWaves;
  s.nd = ((kp_d >> 9) + kEngWaves - 1) / kEngWaves;
  if (2 * 3 * kp_in > 48 * 1024) return false;      // six plane pieces per thread
  return true;
}

}  // namespace

size_t ffn_engine_workspace_bytes(int n_ffn, int K) {
  // ctl (64 bytes), inbox [K][2][L] granules, transformed rows [L][K rounded up to 16] granules
  return 64 + (size_t)K * 2 * (n_ffn / K) * 8 + (size_t)(n_ffn / K) * ((K + 15) & ~15) * 8;
}

bool ffn_engine_supported(int hidden, int n_ffn, int K) {
  FfnShape s;
  if (!ffn_shape_of(hidden, n_ffn, K, s)) return false;
  if (n_ffn / K > device_cu_count_strict()) return false;       // every workgroup has to be resident
  return (s.K == 43 && s.logL == 8 && s.ngu <= 6 && s.nd <= 3) || (s.K == 11 && s.logL == 8 && s.ngu <= 1 && s.nd <= 1) ||
         (s.K == 43 && s.logL == 7 && s.ngu <= 3 && s.nd <= 2);
}

int ffn_engine_launch(const FfnEngineArgs& in, hipStream_t stream) {
  FfnShape s;
  if (!ffn_shape_of(in.hidden, in.n_ffn, in.K, s) || !ffn_engine_supported(in.hidden, in.n_ffn, in.K)) return QUIP_ERR_UNSUPPORTED;
  const int L = in.n_ffn / in.K;
  FfnArgs a;
  a.Wg = reinterpret_cast<const uint4*>(in.w_gate); a.Wu = reinterpret_cast<const uint4*>(in.w_up);
  a.Wd = reinterpret_cast<const uint4*>(in.w_down);
  a.planes_g = reinterpret_cast<const uint8_t*>(in.planes_gate); a.planes_u = reinterpret_cast<const uint8_t*>(in.planes_up);
  a.had3 = reinterpret_cast<const f16*>(in.had3);
  a.sv_g = reinterpret_cast<const f16*>(in.sv_gate); a.sv_u = reinterpret_cast<const f16*>(in.sv_up);
  a.su_d = reinterpret_cast<const f16*>(in.su_down);
  a.zd = reinterpret_cast<f16*>(in.z_down);
  a.grid = reinterpret_cast<const uint64_t*>(in.grid);
  char* ws = reinterpret_cast<char*>(in.workspace);
  a.ctl = reinterpret_cast<uint32_t*>(ws);
  a.inbox = reinterpret_cast<uint64_t*>(ws + 64);
  a.frow = a.inbox + (size_t)in.K * 2 * L;
  a.dbg = reinterpret_cast<uint64_t*>(in.dbg);
  a.out_scale = in.out_scale; a.in_scale = in.in_scale; a.hidden = in.hidden;
  static const int rep16 = [] { const char* e = getenv("QUIP_ENG_REP"); return e && atoi(e) == 16; }();
  if (s.K == 43 && s.logL == 8 && rep16) return launch_ffn<16, 43, 8, 6, 3>(a, stream);
  if (s.K == 43 && s.logL == 8) return launch_ffn<24, 43, 8, 6, 3>(a, stream);
  if (s.K == 11 && s.logL == 8) return launch_ffn<24, 11, 8, 1, 1>(a, stream);
  if (s.K == 43 && s.logL == 7) return launch_ffn<24, 43, 7, 3, 2>(a, stream);
  return QUIP_ERR_UNSUPPORTED;
}

}  // namespace quip
