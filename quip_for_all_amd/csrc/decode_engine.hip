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

namespace quip {

namespace {

using esync::u32x4_t;

struct FfnArgs {
  const uint4* Wg;
  const uint4* Wu;
  const uint4* Wd;
  const uint8_t* planes_g;   // [3][Kp_in] digits + shift word (output of the input-transform launch)
  const uint8_t* planes_u;
  const f16* had3;           // [3][KKP]: gate.had_right, up.had_right, down.had_left (K x K, row major), each padded to
                             // KKP = K * K rounded up to 8 elements (16-byte pieces)
  const f16* sv_g;           // [n_ffn]
  const f16* sv_u;
  const f16* su_d;           // [n_ffn]
  f16* zd;                   // [hidden]: raw product of down_proj
  const uint64_t* grid;      // grid_packed_abs
  uint64_t* inbox;           // [K][2][L] granules
  uint64_t* frow;            // [K][L] granules
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
  static constexpr int kHad = kAcc + kAccRows * 16;        // fp16 [3][KKP]
  static constexpr int kHadBytes = 3 * KKP * 2;
  static constexpr int kZ = kHad + kHadBytes;              // float [2][RB * 16]: z of this column
  static constexpr int kRed = kZ + 2 * RB * 16 * 4;        // 64 floats of reduction scratch
  static constexpr int kR = kRed + 256;                    // region R: planes of gate / up, then the gathered rows, then down's planes
  static constexpr int kFStride = L + 16;                  // floats per gathered row (bank spread for the MFMA operand reads)
  static constexpr int kFRows = (K + 3) & ~3;
  static constexpr int kFBytes = kFRows * kFStride * 4;
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
  u32x2 tsrc;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(tsrc) : "v"(table_source_ptr(a.grid, lane, wave)) : "memory");
  uint32_t gen;
  esync::ld4(gen, a.ctl);
  constexpr int HPIECES = 3 * E::KKP / 8;          // 16-byte pieces of the three K x K factors
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
  if (tid < 2 * 64 && (tid & 63) < K) {
    const int m = tid >> 6, kq = tid & 63;
    const f16* hs = reinterpret_cast<const f16*>(smem + E::kHad) + m * E::KKP + kq * K;
    const float* zz = zbuf + m * RB * 16;
    float t = 0.f;
#pragma unroll 4
    for (int k = 0; k < K; ++k) t = __builtin_fmaf((float)hs[k], zz[k], t);
    esync::st_granule(a.inbox + ((size_t)(kq * 2 + m) * L + w), as_u32(t), epoch);
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
    float v[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[2 * j] = as_f32(g[j].x);
      v[2 * j + 1] = as_f32(g[j].z);
    }
    had::fht16_lanes<LOGL>(v, t);
    // output side of gate / up: fp16( (v * scale) * SV ), element (w, 16 t + r) of the (K, L) view
    const int e0 = w * L + (active ? t : 0) * 16;
    const f16* sv = (m ? a.sv_u : a.sv_g) + e0;
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
      had::unpack8(*reinterpret_cast<const uint4*>(a.su_d + e0), suf);
      had::unpack8(*reinterpret_cast<const uint4*>(a.su_d + e0 + 8), suf + 8);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float u = __shfl(o[r], (lane + 16) & 63, 64);
        e[r] = had::fmul(had::fmul(u, had::silu(o[r])), suf[r]);
      }
    }
    had::fht16_lanes<LOGL>(e, t);
    if (lane < 16 && t < TPR) {
      uint64_t* dst = a.frow + ((size_t)w * L + t * 16);
#pragma unroll
      for (int j = 0; j < 8; ++j) esync::st_granule2(dst + 2 * j, as_u32(e[2 * j]), as_u32(e[2 * j + 1]), epoch);
    }
  }
  ENG_STAMP(6);

  // ---- (5) gather the K rows (every workgroup), stage as fp32 [k][L + 16] ---------------------------------------
  {
    constexpr int PIECES = K * L / 2;              // 16-byte pieces = 2 granules
    constexpr int NP = (PIECES + kEngThreads - 1) / kEngThreads;
    float* fs = reinterpret_cast<float*>(smem + E::kR);
    // rows K .. kFRows - 1 of the staging area are the zero padding of the k loop
    for (int i = tid; i < (E::kFRows - K) * E::kFStride; i += kEngThreads) fs[K * E::kFStride + i] = 0.f;
    u32x4_t p[NP];
    uint32_t spins = 0;
    for (;;) {
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int i = tid + kEngThreads * j;
        esync::ld16(p[j], a.frow + 2 * (size_t)(i < PIECES ? i : 0));
      }
      esync::drain();
      bool ok = true;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        esync::own(p[j]);
        ok = ok && p[j].y == epoch && p[j].w == epoch;
      }
      if (esync::spin_step(ok, spins, a.ctl + 1, 0x2000u + (uint32_t)w)) break;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int i = tid + kEngThreads * j;
      if (i < PIECES) {
        const int el = 2 * i, k = el >> LOGL, c = el & (L - 1);
        *reinterpret_cast<float2*>(fs + k * E::kFStride + c) = make_float2(as_f32(p[j].x), as_f32(p[j].z));
      }
    }
  }
  __syncthreads();
  ENG_STAMP(7);

  // ---- (6) (H^T (x) I) on the matrix cores: D[j][k'] = sum_k f[k][j] had_d[k][k'] ------------------------------
  //      v_mfma_f32_16x16x4_f32: A[row = l & 15][k = l >> 4], B[k = l >> 4][col = l & 15], D[row = 4 (l >> 4) + i][col = l & 15]
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int JT = L / 16;                                   // row tiles of D (columns j of the view)
  constexpr int JTW = (JT + kEngWaves - 1) / kEngWaves;       // per wave
  constexpr int KSTEPS = E::kFRows / 4;
  f32x4 acc[JTW][RB];
#pragma unroll
  for (int jt = 0; jt < JTW; ++jt)
#pragma unroll
    for (int ct = 0; ct < RB; ++ct) acc[jt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* fs = reinterpret_cast<const float*>(smem + E::kR);
    const f16* hd = reinterpret_cast<const f16*>(smem + E::kHad) + 2 * E::KKP;
    for (int s = 0; s < KSTEPS; ++s) {
      const int k = 4 * s + q;
      float bv[RB];
#pragma unroll
      for (int ct = 0; ct < RB; ++ct) {
        const int kc = 16 * ct + n;
        bv[ct] = (k < K && kc < K) ? (float)hd[k * K + kc] : 0.f;
      }
#pragma unroll
      for (int jt = 0; jt < JTW; ++jt) {
        const int tile = wave + jt * kEngWaves;
        const float av = fs[k * E::kFStride + (tile < JT ? tile : 0) * 16 + n];
#pragma unroll
        for (int ct = 0; ct < RB; ++ct) acc[jt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[ct], acc[jt][ct], 0, 0, 0);
      }
    }
  }
  // exact maximum of |scale * x| over the whole vector -> block exponent
  float mx = 0.f;
#pragma unroll
  for (int jt = 0; jt < JTW; ++jt)
#pragma unroll
    for (int ct = 0; ct < RB; ++ct)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float m = fabsf(had::fmul(acc[jt][ct][i], a.in_scale));
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
    const float s2 = had::fmul(a.in_scale, as_f32((uint32_t)(sh_d + 127) << 23));
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
  // ctl (64 bytes), inbox [K][2][L] and rows [K][L] granules
  return 64 + (size_t)K * 2 * (n_ffn / K) * 8 + (size_t)n_ffn * 8;
}

bool ffn_engine_supported(int hidden, int n_ffn, int K) {
  FfnShape s;
  if (!ffn_shape_of(hidden, n_ffn, K, s)) return false;
  if (n_ffn / K > device_cu_count()) return false;       // every workgroup has to be resident
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
  if (s.K == 43 && s.logL == 8) return launch_ffn<24, 43, 8, 6, 3>(a, stream);
  if (s.K == 11 && s.logL == 8) return launch_ffn<24, 11, 8, 1, 1>(a, stream);
  if (s.K == 43 && s.logL == 7) return launch_ffn<24, 43, 7, 3, 2>(a, stream);
  return QUIP_ERR_UNSUPPORTED;
}

}  // namespace quip
