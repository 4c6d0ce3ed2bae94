// E8P12 decode GEMV for gfx950, matrix-core variant (default bs=1 path).
//
//   y[n] = sum_k W[n,k] x[k],  W = decode(Qidxs (n, k/8) int16)
//
// Replaces the M=1 use of tinygemm_m16n8k16_chunk_kernel<.., BLayout_E8, ..>
// (origin_order.cu:388-555, 604-648).
//
// Why matrix cores for a GEMV: measured on MI355X (tools/ubench/valu_rate.hip),
// v_dot4c_i32_i8 / v_dot2c_f32_f16 / v_perm_b32 issue at ~4 cycles per wave64
// instruction, so a VALU dot-product kernel needs >= 36 issue cycles per 64 codes
// and saturates the VALU at roughly half of the HBM rate (profiles/r01_*).  One
// v_mfma_i32_16x16x64_i8 (~18 cycles) multiplies 128 decoded codes (16 weight rows
// x 64 k) with up to 16 rows of A at once; here the rows of A are the int8 digit
// planes of x, so ONE instruction yields the exact integer dot products of all
// three planes and the VALU is left with the decode only (2 address ops + 2 xor
// per code).
//
// Arithmetic (same integer statement as e8p_gemv_i8.hip, see there for the proof
// that it equals the reference's decode8weights, origin_order.cu:211-253):
//     4*w (8 x int8) = T1[code >> 8] ^ T2[code & 255]
//     x = (h*65536 + m*256 + l) * 2^-sh   (block fixed point, |X| < 2^22, balanced digits)
//     y[n] = 2^(-sh-2) * (65536*S_h + 256*S_m + S_l),  S_d = sum_k 4w[n,k] * d[k]  (int32, exact)
// Integer accumulation is exact and order independent: K-splits are combined with
// LDS integer atomics and the result is bit-reproducible whatever the schedule.
//
// Mapping (wave64, lane l: n = l & 15, q = l >> 4):
//   item   = 16 weight rows x 512 k (one 128-byte line per row): lane (n, q) loads
//            bytes [16 q, +16) and [64 + 16 q, +16) of row n's line = 2 x 8 codes (each
//            load instruction covers 64 contiguous bytes per row);
//   MFMA t (t = 0..7) takes the lane's codes 2t, 2t+1 as its 16-byte B fragment
//            (k = (t < 4 ? 0 : 256) + 64 q + 16 (t & 3) + 0..15 of the slice) and, as A fragment, the 16
//            digit bytes of plane (l & 15) at the same k, read from LDS; rows of A
//            beyond the three planes hold (finite) garbage whose outputs are never
//            read.  D[m][n]: lane n (q = 0) regs 0..2 = S_h, S_m, S_l of row n.
//   workgroup = contiguous block of rows, all of K.  x digit planes for up to
//            8192 k live in LDS at a time; longer K is processed in phases that
//            reuse the region.  Waves take items round-robin in (row-block, slice)
//            order, so concurrently running waves read neighbouring lines of the
//            same 16 rows.
// LDS: T1/T2 with REP = 32 copies (ds_read_b64 conflict free, 128 KiB), x planes
// (3 x 8 KiB), int32 accumulators [rows][4].
#include "e8p_gemv_core.hip.h"

namespace quip {

namespace {

// Shape of one weight load instruction.  0 (default): 16 rows x 64 B straight in the MFMA layout.
// 1: 8 rows x 128 B (whole cache lines; lane = (row l >> 3, 16-byte chunk l & 7)), redistributed to the
// MFMA's (row l & 15, k block l >> 4) layout with ds_bpermute after landing.  A pure read probe
// streams whole lines 8-9 % faster (tools/shape_probe.py: 11.6 vs 12.7 us for 58.7 MB), but the 16
// ds_bpermute per item go through the LDS pipe the table lookups already load: measured 4.85 vs
// 4.52 us (4096^2), 24.7 vs 20.9 us (8192 x 28672), 42.7 vs 33.4 us (2 x 28672 x 8192).  Kept for
// experiments (-DQUIP_GEMV_R8=1); parity-tested in both settings.
#ifndef QUIP_GEMV_R8
#define QUIP_GEMV_R8 0
#endif
// MFMA steps whose table / A reads are in flight ahead of the MFMA that consumes them
#ifndef QUIP_GEMV_PIPE
#define QUIP_GEMV_PIPE 4
#endif
// one-shot mode: slots requested ahead of the one being decoded.  Measured on the 7B shapes (us per
// launch, q/k/v group | gate/up group | down | 8192^2): depth 1: 8.35 10.61 6.74 7.61; 2: 8.67 10.72 7.02
// 7.53; 3: 8.86 10.75 7.77 8.21; 4: 8.79 11.33 7.61 8.62; all upfront: 8.78 12.22 7.64 8.61.
#ifndef QUIP_GEMV_DEPTH
#define QUIP_GEMV_DEPTH 1
#endif
// streaming mode with one slot per wave: items handed out dynamically (LDS counter) instead of round-robin
#ifndef QUIP_GEMV_DYNAMIC
#define QUIP_GEMV_DYNAMIC 1
#endif

// Up to G independent GEMVs of the same K in one launch (q/k/v or gate/up of a decoder block):
// workgroup b owns rows [b * rpb[p], +rpb[p]) of every problem p; its items run over all of
// them, so the tables are built once and the launch / drain cost is paid once.
template <int G>
struct GemvGroup {
  const uint4* W[G];
  const uint8_t* planes[G];
  f16* y[G];
  int N[G];
  int rpb[G];
  int boff[G];   // workgroup b serves row range ((b - boff) mod gridDim.x) of the problem: problems that need
                 // fewer workgroups than the launch has start at different workgroups (70B k / v next to q)
  const void* grid2;   // RVQ3 table modes (REP 40 / 20): the E81B residual table, 256 x 8 int8 (4r); else unused
};

// Input side of the GEMV computed in the prologue instead of by separate launches (bs = 1 decode,
// K_left == 1, n = k a power of two):
//   z != null:  h = post (.) (z_scale * H_n z) + residual      (the producer's output transform,
//               qlinear.py:108-114; workgroup 0 stores h to h_out, every workgroup keeps its copy)
//   else:       h = x
//   problem g:  planes_g = digits( scale[g] * rms(h) * H_n (h (.) rms_w (.) pre[g]) )
//               (RMSNorm + SU + input transform, qlinear.py:90-100)
// Every workgroup repeats the (tiny) transforms on its own: no extra launch, no extra HBM round
// trip.  The arithmetic is had_device.hip.h's, bit identical to the stand-alone kernels.
struct FusedIn {
  const f16* x;
  const f16* z;
  const f16* post;
  const f16* residual;
  f16* h_out;
  const f16* rms_w;
  const f16* pre[3];
  float scale[3];
  float z_scale, rms_eps;
  int n, logn;
};

// ONESHOT: the workgroup has at most SLOTS * nwaves items, so every wave requests all of its
// items up front and never reloads a slot (decode shapes of a 7B model: 8..48 items per
// workgroup); otherwise slots are reloaded in place while the stream lasts.
//
// ROWS (G == 1): `mrows` <= 5 activation rows against the ONE weight matrix (skinny GEMM, the reference's
// M < 32 use of tinygemm_m16n8k16_chunk_kernel, origin_order.cu:388-555).  The 16 A rows of the MFMA
// carry (activation row r, digit plane d) = 3 r + d, so the codes are streamed and decoded ONCE and every
// activation row costs no further instruction: D row 3 r + d = S_d of activation row r.  gp.planes[0]
// holds the rows' plane images back to back (stride 3 Kp + 16 bytes), gp.y[0] is (mrows, N) row major.
template <int REP, int SLOTS, int MAXT, int G, bool ONESHOT, bool FUSED, bool ROWS = false>
__global__ __launch_bounds__(MAXT) void e8p_gemv_mfma_kernel(
    GemvGroup<G> gp, FusedIn fi, const uint64_t* __restrict__ grid, int K, int Kp, uint64_t* __restrict__ dbg,
    int mrows) {
  static_assert(!ROWS || (G == 1 && !FUSED), "rows mode: one matrix, planes from memory");
  static_assert(!Lds<REP>::kRvq3 || !FUSED, "RVQ3 tables: planes from memory");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = Lds<REP>;
#define QUIP_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define QUIP_STAMP2(i) do { if (dbg && threadIdx.x == 0) dbg[(2048 + blockIdx.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  QUIP_STAMP(0);
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = __builtin_amdgcn_readfirstlane(nthreads >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int row_u4 = K >> 6;               // uint4 per packed row (K/4 bytes)
  const int J = Kp >> 9;                   // slices of 512 k
  int row0[G], rows_here[G], cbase[G + 1], rbase[G];   // per problem: first row, rows, item base, acc row base
  cbase[0] = 0;
#pragma unroll
  for (int p = 0; p < G; ++p) {
    int bp = (int)blockIdx.x - gp.boff[p];
    bp = bp < 0 ? bp + (int)gridDim.x : bp;
    row0[p] = bp * gp.rpb[p];
    rows_here[p] = max(0, min(gp.N[p], row0[p] + gp.rpb[p]) - row0[p]);
    cbase[p + 1] = cbase[p] + ((rows_here[p] + 15) >> 4) * J;
    rbase[p] = p == 0 ? 0 : rbase[p - 1] + gp.rpb[p - 1];
  }
  const int cnt = cbase[G];                // items of this workgroup

  // item -> (problem, row block, slice); wave uniform
  auto problem_of = [&](int it) -> int {
    int p = 0;
#pragma unroll
    for (int i = 1; i < G; ++i) p += it >= cbase[i] ? 1 : 0;
    return p;
  };
  // arr[p] for a wave-uniform p without ever indexing dynamically: a select chain the optimiser
  // recognises as arr[p] would put the array in scratch memory, and scratch accesses are VMEM
  // operations that would corrupt the hand-counted vmcnt waits below.  The empty asm keeps each
  // step opaque (and in an SGPR).
  auto pick = [&](const int* arr, int p) -> int {
    int v = arr[0];
#pragma unroll
    for (int i = 1; i < G; ++i) {
      v = p == i ? arr[i] : v;
      asm volatile("" : "+s"(v));
    }
    return v;
  };
  // lanes outside the matrix read a valid (clamped) address
  // and are neutralised by zero x digits (k padding) / never-read rows
  // Load j (0, 1) of lane (n, q) reads bytes [64 j + 16 q, +16) of row n's 128-byte slice line, so
  // each load instruction covers 64 contiguous bytes per row (measured: the 16-byte-at-32-byte-
  // stride alternative costs ~25 % of the streaming rate).  A slice that sticks out of the row
  // (K % 512 != 0) re-reads a valid piece; its x digits are zero (k padding).
  auto item_ptr = [&](int it, int j) -> const uint4* {
    const int p = problem_of(it);
    const int li = it - pick(cbase, p);
    const int rb = li / J, s = li - rb * J;
    const int N = pick(gp.N, p);
    const uint4* W = gp.W[0];
#pragma unroll
    for (int i = 1; i < G; ++i) {
      W = p == i ? gp.W[i] : W;
      asm volatile("" : "+s"(W));
    }
    int row, off;     // off: uint4 units inside the row
    if constexpr (kR8) {        // load j covers rows 8j .. 8j+7 of the row block, the slice's whole 128-byte line each
      row = pick(row0, p) + rb * 16 + 8 * j + (lane >> 3);
      off = s * 8 + (lane & 7);
      off = off < row_u4 ? off : s * 8 + (lane & 1);   // K % 512 != 0: re-read a valid piece (x digits are 0)
    } else {
      row = pick(row0, p) + rb * 16 + n;
      off = s * 8 + q + 4 * j;
      off = off < row_u4 ? off : s * 8 + q;
    }
    row = row < N ? row : N - 1;
    if constexpr (L::kRvq3)   // 12-byte pieces of the native 3-byte code stream (same piece index)
      return reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(W) + ((size_t)row * row_u4 + off) * 12);
    return W + (size_t)row * row_u4 + off;
  };

  // (0) VMEM loads return in issue order: the x digit planes (L2 hits) -- or, fused, the fp16
  //     vectors the prologue transforms -- are requested before the (TLB-cold, HBM) weight
  //     loads, all through hand-counted asm loads.
  u32x2 tsrc;   // this lane's table source entry: the first load of the kernel, so the first to land
  asm volatile("global_load_dwordx2 %0, %1, off"
               : "=v"(tsrc)
               : "v"(L::kD4 ? table_source_ptr_d4(grid, lane, wave) : table_source_ptr(grid, lane, wave))
               : "memory");
  u32x2 tsrc3;  // RVQ3: this lane's E81B entry, the second load of the kernel (any later wait covers it)
  if constexpr (L::kRvq3)
    asm volatile("global_load_dwordx2 %0, %1, off"
                 : "=v"(tsrc3)
                 : "v"(reinterpret_cast<const uint2*>(gp.grid2) + ((wave & 7) * 32 + (lane & 31)))
                 : "memory");
  constexpr int XR = 6;  // 16-byte x pieces per thread: needs nthreads >= G * 3 * Kp / 96
  const int ppieces = 3 * (Kp >> 4);       // pieces per problem (rows mode: per activation row)
  const int xpieces = (ROWS ? mrows : G) * ppieces;
  const size_t pstride = (size_t)3 * Kp + 16;   // rows mode: bytes between the rows' plane images
  u32x4 xr[XR];
  // fused: thread t < n / 16 owns elements [16 t, 16 t + 16) = pieces 2 t, 2 t + 1 of every vector
  const bool act = FUSED && tid < (fi.n >> 4);   // wave uniform (n % 1024 == 0)
  u32x4 pin[2], ppost[2], pres[2], prms[2], ppre[G][2];
  const uint4* hot;
  if constexpr (FUSED) {
    // Every thread issues every load unconditionally (threads past n / 16 re-read a valid piece,
    // absent vectors fall back to the input vector): a load under a branch would make the
    // compiler merge its destination with another value, i.e. copy a register whose load is
    // still in flight.
    const f16* in = fi.z ? fi.z : fi.x;
    const int c = (tid & ((fi.n >> 4) - 1)) * 2;
    const uint4* s_in = reinterpret_cast<const uint4*>(in) + c;
    const uint4* s_post = reinterpret_cast<const uint4*>(fi.z ? fi.post : in) + c;
    const uint4* s_res = reinterpret_cast<const uint4*>((fi.z && fi.residual) ? fi.residual : in) + c;
    const uint4* s_rms = reinterpret_cast<const uint4*>(fi.rms_w ? fi.rms_w : in) + c;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      asm_load16(pin[h], s_in + h);
      asm_load16(ppost[h], s_post + h);
      asm_load16(pres[h], s_res + h);
      asm_load16(prms[h], s_rms + h);
#pragma unroll
      for (int g = 0; g < G; ++g) asm_load16(ppre[g][h], reinterpret_cast<const uint4*>(fi.pre[g]) + c + h);
    }
    hot = reinterpret_cast<const uint4*>(fi.z ? fi.z : fi.x) + (size_t)((tid * 2) % ((fi.n >> 3) - 1));
  } else {
    // every workgroup reads the same planes: start each one at a different piece so that they do
    // not convoy on the same L2 channels
    const int rot = (int)((blockIdx.x * 613u) % (uint32_t)xpieces);
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int i = tid + j * nthreads;
      int ic = i < xpieces ? i + rot : 0;
      ic = ic >= xpieces ? ic - xpieces : ic;
      int p = 0;
#pragma unroll
      for (int g = 1; g < (ROWS ? 5 : G); ++g) p += ic >= g * ppieces ? 1 : 0;
      const uint8_t* src = gp.planes[0];   // per-lane choice (pieces of several problems in one wave)
      if constexpr (ROWS) {
        src += (size_t)p * pstride;
      } else {
#pragma unroll
        for (int g = 1; g < G; ++g) {
          src = p == g ? gp.planes[g] : src;
          asm volatile("" : "+v"(src));
        }
      }
      asm_load16(xr[j], reinterpret_cast<const uint4*>(src) + (ic - p * ppieces));
    }
    // past-the-end reloads read the L2-resident x planes (each lane its own 32 bytes)
    hot = reinterpret_cast<const uint4*>(gp.planes[0]) + (size_t)((tid * 2) % (ppieces - 1));
  }
  // Weight loads in flight per wave.  Streaming mode: all SLOTS slots, reloaded in place.  One-shot
  // mode: every item has its own slot registers but only kDepth slots are requested ahead of the
  // one being decoded (slot i + kDepth is requested when slot i has landed), so that the address /
  // TA work of the loads overlaps with the LDS-bound decode of other waves instead of preceding it.
  constexpr int kDepth = ONESHOT ? (SLOTS < QUIP_GEMV_DEPTH ? SLOTS : QUIP_GEMV_DEPTH) : SLOTS;
  static_assert(!(L::kRvq3 && kR8), "RVQ3: 12-byte pieces, no line redistribution");
  using Slot = std::conditional_t<L::kRvq3, u32x3, u32x4>;
  Slot qa[SLOTS], qb[SLOTS];
#pragma unroll
  for (int i = 0; i < kDepth; ++i) {
    const int it0 = wave + i * nwaves;
    const bool real = it0 < cnt;
    asm_load16_nt(qa[i], real ? item_ptr(it0, 0) : hot);
    asm_load16_nt(qb[i], real ? item_ptr(it0, 1) : hot + 1);
  }
  QUIP_STAMP(1);

  // (1) zeroed accumulators, then the tables as soon as the table source load has landed (every
  //     later load may still be in flight)
  for (int i = tid; i < kMaxRowsPerBlock * 4; i += nthreads) reinterpret_cast<int*>(smem + L::kAcc)[i] = 0;
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tsrc) : "n"((FUSED ? 2 * (4 + G) : XR) + 2 * kDepth) : "memory");
  if (wave < 8) fill_tables_from_lane<REP>(smem, tsrc, lane, wave);
  if constexpr (L::kRvq3) {
    asm volatile("" : "+v"(tsrc3));
    if (wave < 8) fill_t3_from_lane<REP>(smem, tsrc3, lane, wave);
  }
  int sh[ROWS ? 5 : G];
  if constexpr (ROWS) {
#pragma unroll
    for (int r = 0; r < 5; ++r)
      sh[r] = *reinterpret_cast<const int*>(gp.planes[0] + (size_t)(r < mrows ? r : 0) * pstride + (size_t)3 * Kp);
  } else if constexpr (!FUSED) {
#pragma unroll
    for (int p = 0; p < G; ++p) sh[p] = *reinterpret_cast<const int*>(gp.planes[p] + (size_t)3 * Kp);
  }
  QUIP_STAMP(2);

  if constexpr (FUSED) {
    // (2f) the input side of the layer, computed here (see FusedIn).  All vector loads of the
    //      prologue are older than the weight loads: "at most 2 * SLOTS outstanding" == landed.
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(2 * kDepth) : "memory");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      asm volatile("" : "+v"(pin[h]), "+v"(ppost[h]), "+v"(pres[h]), "+v"(prms[h]));
#pragma unroll
      for (int g = 0; g < G; ++g) asm volatile("" : "+v"(ppre[g][h]));
    }
    QUIP_STAMP2(0);
    auto as4 = [](const u32x4& v) { return make_uint4(v.x, v.y, v.z, v.w); };
    float* buf = reinterpret_cast<float*>(smem + L::kX + G * 3 * Kp);
    float* red = buf + had::buf_floats(Kp);
    const int nfht = fi.n >> 4;
    float xf[16];
    had::unpack8(as4(pin[0]), xf);
    had::unpack8(as4(pin[1]), xf + 8);
    if (fi.z) {
      had::fht16(xf, buf, tid, fi.logn, act);
      float tp[16], tr[16];
      had::unpack8(as4(ppost[0]), tp);
      had::unpack8(as4(ppost[1]), tp + 8);
      had::unpack8(as4(pres[0]), tr);
      had::unpack8(as4(pres[1]), tr + 8);
      f16 o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        o[r] = had::out_elem(xf[r], fi.z_scale, true, tp[r], false, 0.f, fi.residual != nullptr, tr[r]);
      if (act && blockIdx.x == 0) {
        uint4* dst = reinterpret_cast<uint4*>(fi.h_out + tid * 16);
        dst[0] = *reinterpret_cast<uint4*>(&o[0]);
        dst[1] = *reinterpret_cast<uint4*>(&o[8]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) xf[r] = (float)o[r];
    }
    QUIP_STAMP2(1);
    float tot = 0.f;
    if (fi.rms_w) {
      float ss = 0.f;
      had::sumsq8(xf, ss);
      had::sumsq8(xf + 8, ss);
      tot = had::block_reduce(act ? ss : 0.f, false, red, tid, nfht);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) e[r] = xf[r];
      if (fi.rms_w) {
        had::mul8(e, as4(prms[0]));
        had::mul8(e + 8, as4(prms[1]));
      }
      had::mul8(e, as4(ppre[g][0]));
      had::mul8(e + 8, as4(ppre[g][1]));
      const float scale = fi.rms_w ? had::rms_scale(fi.scale[g], tot, fi.n, fi.rms_eps) : fi.scale[g];
      if (g == 0) QUIP_STAMP2(2);
      had::fht16(e, buf, tid, fi.logn, act);
      if (g == 0) QUIP_STAMP2(3);
      const float bound = had::block_reduce(act ? had::absmax16(e, scale) : 0.f, true, red, tid, nfht);
      if (g == 0) QUIP_STAMP2(4);
      sh[g] = had::shift_for(bound);
      uint4 dg[3];
      had::planes16(e, scale, sh[g], dg);
      if (act) {
#pragma unroll
        for (int d = 0; d < 3; ++d)
          *reinterpret_cast<uint4*>(smem + L::kX + (g * 3 + d) * Kp + tid * 16) = dg[d];
      }
      if (g == 0) QUIP_STAMP2(5);
    }
    __syncthreads();
  } else {
    // (2) x digit planes into LDS once the 6 plane loads have landed (the 2 * SLOTS weight
    //     loads behind them may still be in flight)
    asm_wait_vmcnt_x<2 * kDepth>(xr[0], xr[1], xr[2], xr[3], xr[4], xr[5]);
    const int rot = (int)((blockIdx.x * 613u) % (uint32_t)xpieces);
#pragma unroll
    for (int j = 0; j < XR; ++j) {
      const int i = tid + j * nthreads;
      int ic = i + rot;
      ic = ic >= xpieces ? ic - xpieces : ic;
      if (i < xpieces) *reinterpret_cast<u32x4*>(smem + L::kX + ic * 16) = xr[j];
    }
    __syncthreads();
  }
  QUIP_STAMP(3);

  const uint32_t lane_c = L::kD4 ? ((uint32_t)lane << 2)
                          : (L::kRep1 == 32) ? (((uint32_t)(lane & 31) << 3) | 0x00010000u)
                                             : (((uint32_t)(lane & 15) << 3) | (uint32_t)L::kT1);
  const uint32_t lane_c2 = ((uint32_t)(lane & 15) << 3) | (uint32_t)L::kT2;
  const uint32_t lane_c3 = L::kRvq3 ? ((((uint32_t)lane & (uint32_t)(L::kRep3 - 1)) << 3) | (uint32_t)L::kT3) : 0u;
  int* accs = reinterpret_cast<int*>(smem + L::kAcc);
  // A fragment address of this lane: plane (lane & 15) clamped to a valid plane (rows >= 3
  // of A are don't-care), k = slice*512 + (t < 4 ? 0 : 256) + q*64 + (t & 3)*16
  const uint32_t xlane = L::kX + (uint32_t)min(n, ROWS ? 3 * mrows - 1 : 2) * Kp + (uint32_t)q * (kR8 ? 128 : 64);
  // rows mode: this lane's four D rows m = 4 q + i are (activation row m / 3, plane m % 3)
  int roff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = 4 * q + i;
    roff[i] = (ROWS && m < 3 * mrows) ? ((m / 3) * gp.rpb[0]) * 4 + (m % 3) : -1;
  }
  QUIP_STAMP(4);

  // (tried and measured slower on MI355X: slice-major item order with the A fragments of a slice
  //  kept in registers across row blocks -- the stream is bound by the DRAM access pattern, not by
  //  LDS reads, and slice-major makes the waves drift apart)
  auto run_item = [&](int cur, const ItemAddr& ad) {
    const int p = problem_of(cur);
    const int li = cur - pick(cbase, p);
    const int rb = li / J, sl = li - rb * J;
    const uint32_t xa = xlane + (uint32_t)(p * 3 * Kp + sl * 512);
    const i32x4 acc = L::kD4 ? item_mfma_d4(ad, xa) : item_mfma(ad, xa);
    if constexpr (ROWS) {
      int* dst = accs + (rb * 16 + n) * 4;
      const int a4[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (roff[i] >= 0) __hip_atomic_fetch_add(dst + roff[i], a4[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (q == 0) {   // lanes 0..15 (q == 0) hold S_h, S_m, S_l of row rb*16 + n in acc[0..2]
      int* dst = accs + (pick(rbase, p) + rb * 16 + n) * 4;
      __hip_atomic_fetch_add(dst + 0, acc.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst + 1, acc.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst + 2, acc.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };

  // (3) the weight stream.  Every slot is reloaded right after its codes have been turned into
  //     LDS addresses; once a wave has no further item the reload reads a hot L2 line (the
  //     x planes) instead, so the VMEM queue always holds exactly 2 * SLOTS loads in
  //     slot order and "slot i has landed" == vmcnt(2 * (SLOTS - 1)) throughout.
  if constexpr (ONESHOT) {
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      const int cur = wave + i * nwaves;
      // slots 0 .. min(i + kDepth, SLOTS) - 1 have been requested: once slot i has landed at most
      // min(kDepth - 1, SLOTS - 1 - i) newer slots are outstanding (folds after unrolling)
      switch ((kDepth - 1) < (SLOTS - 1 - i) ? (kDepth - 1) : (SLOTS - 1 - i)) {
        case 0: asm_wait_vmcnt<0>(qa[i], qb[i]); break;
        case 1: asm_wait_vmcnt<2>(qa[i], qb[i]); break;
        case 2: asm_wait_vmcnt<4>(qa[i], qb[i]); break;
        case 3: asm_wait_vmcnt<6>(qa[i], qb[i]); break;
        case 4: asm_wait_vmcnt<8>(qa[i], qb[i]); break;
        case 5: asm_wait_vmcnt<10>(qa[i], qb[i]); break;
        case 6: asm_wait_vmcnt<12>(qa[i], qb[i]); break;
        default: asm_wait_vmcnt<14>(qa[i], qb[i]); break;
      }
      if (i + kDepth < SLOTS) {   // request the slot kDepth ahead (its own registers, nothing to wait for)
        const int nxt = wave + (i + kDepth) * nwaves;
        const bool real = nxt < cnt;
        asm_load16_nt(qa[(i + kDepth) < SLOTS ? (i + kDepth) : 0], real ? item_ptr(nxt, 0) : hot);
        asm_load16_nt(qb[(i + kDepth) < SLOTS ? (i + kDepth) : 0], real ? item_ptr(nxt, 1) : hot + 1);
      }
      if (cur < cnt) {   // wave-uniform
        ItemAddr ad;
        if constexpr (kR8) {
          u32x4 da, db;
          redistribute_r8(qa[i], qb[i], lane, da, db);
          item_addresses<REP>(da, db, lane_c, lane_c2, ad, lane_c3);
        } else {
          item_addresses<REP>(qa[i], qb[i], lane_c, lane_c2, ad, lane_c3);
        }
        run_item(cur, ad);
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (SLOTS == 1 && QUIP_GEMV_DYNAMIC && !kR8) {
    // One slot per wave, items handed out by an LDS counter (word 3 of accumulator row 0, zeroed with
    // the accumulators): the SIMD arbiters favour the oldest waves, so with a fixed round-robin split the
    // young waves finish last and the workgroup's tail runs with few loads in flight.
    int cur = wave;
    while (cur < cnt) {   // wave-uniform
      int nxt = 0;
      if (lane == 0) nxt = __hip_atomic_fetch_add(accs + 3, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      nxt = __builtin_amdgcn_readfirstlane(nxt) + nwaves;
      asm_wait_vmcnt<0>(qa[0], qb[0]);
      ItemAddr ad;
      item_addresses<REP>(qa[0], qb[0], lane_c, lane_c2, ad, lane_c3);
#pragma unroll
      for (int t = 0; t < 8; ++t)
        asm volatile("" : "+v"(ad.a1l[t]), "+v"(ad.a2l[t]), "+v"(ad.a1h[t]), "+v"(ad.a2h[t]));
      const bool real = nxt < cnt;
      asm_load16_nt(qa[0], real ? item_ptr(nxt, 0) : hot);
      asm_load16_nt(qb[0], real ? item_ptr(nxt, 1) : hot + 1);
      run_item(cur, ad);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      cur = nxt;
    }
  } else {
    for (int it = wave; it < cnt; it += SLOTS * nwaves) {
#pragma unroll
      for (int i = 0; i < SLOTS; ++i) {
        const int cur = it + i * nwaves;
        asm_wait_vmcnt<2 * (SLOTS - 1)>(qa[i], qb[i]);
        ItemAddr ad;
        if constexpr (kR8) {
          u32x4 da, db;
          redistribute_r8(qa[i], qb[i], lane, da, db);
          item_addresses<REP>(da, db, lane_c, lane_c2, ad, lane_c3);
        } else {
          item_addresses<REP>(qa[i], qb[i], lane_c, lane_c2, ad, lane_c3);
        }
        // the slot's codes are consumed: pin the addresses, then reload the slot in place
#pragma unroll
        for (int t = 0; t < 8; ++t)
          asm volatile("" : "+v"(ad.a1l[t]), "+v"(ad.a2l[t]), "+v"(ad.a1h[t]), "+v"(ad.a2h[t]));
        const int nxt = cur + SLOTS * nwaves;
        const bool real = nxt < cnt;
        asm_load16_nt(qa[i], real ? item_ptr(nxt, 0) : hot);
        asm_load16_nt(qb[i], real ? item_ptr(nxt, 1) : hot + 1);
        if (cur < cnt) run_item(cur, ad);   // wave-uniform
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the trailing hot-line reloads
  QUIP_STAMP(5);
  __syncthreads();
  QUIP_STAMP(6);

  // (5) y = 2^(-sh-2) * (65536 S_h + 256 S_m + S_l), fp16 RN, coalesced
  if constexpr (ROWS) {
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      if (r < mrows) {
        const float unscale = unscale_of(sh[r], L::kD4 ? 1 : 2);
        for (int t = tid; t < rows_here[0]; t += nthreads) {
          const int* a = accs + (r * gp.rpb[0] + t) * 4;
          const float f = __builtin_fmaf((float)a[0], 65536.f, __builtin_fmaf((float)a[1], 256.f, (float)a[2]));
          gp.y[0][(size_t)r * gp.N[0] + row0[0] + t] = (f16)(f * unscale);
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < (ROWS ? 0 : G); ++p) {
    const float unscale = unscale_of(sh[p], L::kD4 ? 1 : 2);   // table entries are 4w (E8P) / 2w (D4)
    for (int t = tid; t < rows_here[p]; t += nthreads) {
      const int* a = accs + (rbase[p] + t) * 4;
      const float f = __builtin_fmaf((float)a[0], 65536.f, __builtin_fmaf((float)a[1], 256.f, (float)a[2]));
      gp.y[p][row0[p] + t] = (f16)(f * unscale);
    }
  }
  QUIP_STAMP(7);
#undef QUIP_STAMP
#undef QUIP_STAMP2
}

template <int REP, int SLOTS, int MAXT, int G, bool ONESHOT = false, bool FUSED = false, bool ROWS = false>
int launch(const GemvGroup<G>& gp, const void* grid, int k, int kp, int nblocks, int threads, uint64_t* dbg,
           hipStream_t stream, const FusedIn* fin = nullptr, int mrows = 1) {
  auto kern = e8p_gemv_mfma_kernel<REP, SLOTS, MAXT, G, ONESHOT, FUSED, ROWS>;
  const int lds = FUSED ? Lds<REP>::bytes_fused(kp, G) : Lds<REP>::bytes(kp, ROWS ? mrows : G);
  const FusedIn fi = fin ? *fin : FusedIn{};
  static DynLdsCache configured;   // per instantiation, per device
  if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(nblocks), dim3(threads), lds, stream, gp, fi,
                     reinterpret_cast<const uint64_t*>(grid), k, kp, dbg, mrows);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

// Streaming probe with the matrix-core kernel's access pattern (item = 16 rows x LINES x 128 B,
// waves <-> slices) and no decode: the read-bandwidth ceiling of this pattern.
template <int LINES>
__global__ __launch_bounds__(1024) void pattern_probe_kernel(const uint4* __restrict__ W,
                                                            uint32_t* __restrict__ out, int N, int K,
                                                            int rows_per_block) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = __builtin_amdgcn_readfirstlane((int)blockDim.x >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int row0 = blockIdx.x * rows_per_block;
  const int rows_here = min(N, row0 + rows_per_block) - row0;
  const int nrb = (rows_here + 15) >> 4;
  const int row_u4 = K >> 6;
  const int J = (K >> 9) / LINES;  // slices of LINES * 512 k
  const int cnt = nrb * J;
  uint32_t acc = 0;
  constexpr int NL = 2 * LINES;
  u32x4 s0[NL], s1[NL];   // two slots, reloaded in place (never copied while in flight)
  auto ptr = [&](int it) -> const uint4* {
    if (it >= cnt) return W + (size_t)lane;   // past the end: a hot line
    const int rb = it / J, s = it - rb * J;
    int row = row0 + rb * 16 + n;
    row = row < N ? row : N - 1;
    // load j of a lane reads bytes [64 j + 16 q, +16) of the row's slice: every instruction
    // covers 64 contiguous bytes per row
    return W + (size_t)row * row_u4 + s * 8 * LINES + q;
  };
  {
    const uint4* p0 = ptr(wave);
    const uint4* p1 = ptr(wave + nwaves);
#pragma unroll
    for (int j = 0; j < NL; ++j) asm_load16_nt(s0[j], p0 + 4 * j);
#pragma unroll
    for (int j = 0; j < NL; ++j) asm_load16_nt(s1[j], p1 + 4 * j);
  }
  for (int it = wave; it < cnt; it += 2 * nwaves) {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NL) : "memory");
#pragma unroll
    for (int j = 0; j < NL; ++j) { asm volatile("" : "+v"(s0[j])); acc ^= s0[j].x ^ s0[j].y ^ s0[j].z ^ s0[j].w; }
    asm volatile("" : "+v"(acc));
    {
      const uint4* p = ptr(it + 2 * nwaves);
#pragma unroll
      for (int j = 0; j < NL; ++j) asm_load16_nt(s0[j], p + 4 * j);
    }
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NL) : "memory");
#pragma unroll
    for (int j = 0; j < NL; ++j) { asm volatile("" : "+v"(s1[j])); acc ^= s1[j].x ^ s1[j].y ^ s1[j].z ^ s1[j].w; }
    asm volatile("" : "+v"(acc));
    {
      const uint4* p = ptr(it + 3 * nwaves);
#pragma unroll
      for (int j = 0; j < NL; ++j) asm_load16_nt(s1[j], p + 4 * j);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) out[0] = acc;
}

// Streaming probe for the shape of a load instruction: every wave keeps 2 slots x 2 loads in flight
// over the workgroup's (contiguous) row range like the GEMV does; one load instruction covers
// R rows x (1024 / R) contiguous bytes (R = 16: the GEMV's pattern, 4, or 1 = fully contiguous).
template <int R>
__global__ __launch_bounds__(1024) void shape_probe_kernel(const uint4* __restrict__ W, uint32_t* __restrict__ out,
                                                          int N, int K, int rows_per_block) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = __builtin_amdgcn_readfirstlane((int)blockDim.x >> 6);
  const int row0 = blockIdx.x * rows_per_block;
  const int rows_here = min(N, row0 + rows_per_block) - row0;
  const int row_u4 = K >> 6;                       // uint4 per row
  constexpr int CH = 64 / R;                        // uint4 per row per instruction
  const int chunks = row_u4 / (2 * CH);             // items along a row (2 instructions each)
  const int rgroups = (rows_here + R - 1) / R;
  const int cnt = rgroups * chunks;
  const int lr = lane / CH, lc = lane % CH;         // lane -> (row in group, uint4 in chunk)
  auto ptr = [&](int it) -> const uint4* {
    if (it >= cnt) return W + (size_t)lane;
    const int rg = it / chunks, c = it - rg * chunks;
    int row = row0 + rg * R + lr;
    row = row < N ? row : N - 1;
    return W + (size_t)row * row_u4 + c * 2 * CH + lc;
  };
  u32x4 a0, a1, b0, b1;
  uint32_t acc = 0;
  {
    const uint4* p0 = ptr(wave);
    const uint4* p1 = ptr(wave + nwaves);
    asm_load16_nt(a0, p0); asm_load16_nt(a1, p0 + CH);
    asm_load16_nt(b0, p1); asm_load16_nt(b1, p1 + CH);
  }
  for (int it = wave; it < cnt; it += 2 * nwaves) {
    asm_wait_vmcnt<2>(a0, a1);
    acc ^= a0.x ^ a0.y ^ a0.z ^ a0.w ^ a1.x ^ a1.y ^ a1.z ^ a1.w;
    asm volatile("" : "+v"(acc));
    { const uint4* p = ptr(it + 2 * nwaves); asm_load16_nt(a0, p); asm_load16_nt(a1, p + CH); }
    asm_wait_vmcnt<2>(b0, b1);
    acc ^= b0.x ^ b0.y ^ b0.z ^ b0.w ^ b1.x ^ b1.y ^ b1.z ^ b1.w;
    asm volatile("" : "+v"(acc));
    { const uint4* p = ptr(it + 3 * nwaves); asm_load16_nt(b0, p); asm_load16_nt(b1, p + CH); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) out[0] = acc;
}

}  // namespace

int shape_probe_launch(const void* qidxs, void* out, int n, int k, const GemvTune& tune, hipStream_t stream) {
  const int ncu = device_cu_count();
  int nblocks = tune.blocks > 0 ? tune.blocks : ncu;
  int rpb = (n + nblocks - 1) / nblocks;
  rpb = (rpb + 15) & ~15;
  nblocks = (n + rpb - 1) / rpb;
  const int waves = tune.max_waves > 0 ? tune.max_waves : 8;
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * waves), 0, stream, reinterpret_cast<const uint4*>(qidxs),
                       reinterpret_cast<uint32_t*>(out), n, k, rpb);
    return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
  };
  if (tune.rows == 1) return go(shape_probe_kernel<1>);
  if (tune.rows == 4) return go(shape_probe_kernel<4>);
  if (tune.rows == 8) return go(shape_probe_kernel<8>);
  if (tune.rows == 2) return go(shape_probe_kernel<2>);
  return go(shape_probe_kernel<16>);
}

namespace {

// x -> plain digit planes [3][Kp] (Kp = K rounded up to 512, zero padded) + shift word.
__global__ __launch_bounds__(1024) void x_to_planes_linear_kernel(const f16* __restrict__ x,
                                                                  uint8_t* __restrict__ planes,
                                                                  int* __restrict__ sh_out, int K, int Kp) {
  __shared__ uint32_t smax[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthreads = blockDim.x;
  // one workgroup per activation row: row r reads x + r K and writes the image at planes + r (3 Kp + 16)
  x += (size_t)blockIdx.x * K;
  planes += (size_t)blockIdx.x * ((size_t)3 * Kp + 16);
  sh_out = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(sh_out) + (size_t)blockIdx.x * ((size_t)3 * Kp + 16));
  const uint4* xg = reinterpret_cast<const uint4*>(x);
  const int pieces = K >> 3;
  uint32_t mx = 0;  // fp16 magnitudes order like their bit patterns
  for (int p = tid; p < pieces; p += nthreads) {
    const uint4 v = xg[p];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) mx = max(mx, max(w[i] & 0x7fffu, (w[i] >> 16) & 0x7fffu));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o, 64));
  if (lane == 0) smax[wave] = mx;
  __syncthreads();
  mx = 0;
  for (int w = 0; w < (nthreads >> 6); ++w) mx = max(mx, smax[w]);
  const int ebits = (int)(mx >> 10);
  // exponent field 31 = an inf or a NaN in the row: no block exponent, the epilogue answers NaN
  const int sh = ebits == 31 ? kShiftNotFinite : 21 - ((ebits ? ebits : 1) - 15);   // |rint(x * 2^sh)| < 2^22
  const float scale = as_f32((uint32_t)(sh + 127) << 23);
  if (tid == 0) *sh_out = sh;
  for (int p = tid; p < (Kp >> 3); p += nthreads) {
    uint32_t dg[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    if (p < pieces) {
      const uint4 v = xg[p];
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const f16x2 h2 = as_f16x2(w[i >> 1]);
        const int X = (int)__builtin_rintf((float)((i & 1) ? h2.y : h2.x) * scale);
        const int l = (X << 24) >> 24;
        const int X1 = (X - l) >> 8;
        const int m = (X1 << 24) >> 24;
        const int h = (X1 - m) >> 8;
        const int sft = 8 * (i & 3);
        dg[0][i >> 2] |= (uint32_t)(h & 0xff) << sft;
        dg[1][i >> 2] |= (uint32_t)(m & 0xff) << sft;
        dg[2][i >> 2] |= (uint32_t)(l & 0xff) << sft;
      }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
      *reinterpret_cast<uint2*>(planes + (size_t)d * Kp + p * 8) = make_uint2(dg[d][0], dg[d][1]);
  }
}

}  // namespace

static inline int kp_of(int k) { return (k + 511) & ~511; }

bool e8p_gemv_mfma_supported(int n, int k) {
  return n >= 1 && k >= 128 && k % 128 == 0 && kp_of(k) <= Lds<16>::kMaxKp;
}

size_t e8p_gemv_mfma_planes_bytes(int k) { return (size_t)3 * kp_of(k) + 16; }

int x_to_planes_linear_launch(const void* x, void* planes, int k, hipStream_t stream, int rows) {
  if (k < 8 || k % 8 != 0 || rows < 1) return QUIP_ERR_BAD_SHAPE;
  const int kp = kp_of(k);
  int* sh = reinterpret_cast<int*>(reinterpret_cast<char*>(planes) + (size_t)3 * kp);
  const int threads = k >= 8192 ? 1024 : (k >= 2048 ? 256 : 64);
  hipLaunchKernelGGL(x_to_planes_linear_kernel, dim3(rows), dim3(threads), 0, stream,
                     reinterpret_cast<const f16*>(x), reinterpret_cast<uint8_t*>(planes), sh, k, kp);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

int pattern_probe_launch(const void* qidxs, void* out, int n, int k, const GemvTune& tune, hipStream_t stream) {
  const int ncu = device_cu_count();
  int nblocks = tune.blocks > 0 ? tune.blocks : ncu;
  int rpb = (n + nblocks - 1) / nblocks;
  rpb = (rpb + 15) & ~15;
  nblocks = (n + rpb - 1) / rpb;
  int waves = tune.max_waves > 0 ? tune.max_waves : 16;
  const int lines = tune.rows ? tune.rows : 1;
  auto go = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(64 * waves), 0, stream, reinterpret_cast<const uint4*>(qidxs),
                       reinterpret_cast<uint32_t*>(out), n, k, rpb);
    return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
  };
  if (lines == 1) return go(pattern_probe_kernel<1>);
  if (lines == 2) return go(pattern_probe_kernel<2>);
  return go(pattern_probe_kernel<4>);
}

// one-shot regime: all items of a wave in flight at once (SLOTS = items per wave, rounded up)
template <int G>
static int launch_oneshot(const GemvGroup<G>& gp, const void* grid, int k, int kp, int nblocks, int threads,
                          int rep, int items_per_wave, uint64_t* dbg, hipStream_t stream) {
  // 9..16 waves: the 128-VGPR budget of a 1024-thread workgroup holds up to 4 slot register sets
#define QUIP_ONE_BIG(R, S)                                                          \
  if (threads > 512 && rep == R && items_per_wave <= S)                             \
    return launch<R, S, 1024, G, true>(gp, grid, k, kp, nblocks, threads, dbg, stream);
  QUIP_ONE_BIG(32, 1) QUIP_ONE_BIG(32, 2) QUIP_ONE_BIG(32, 3) QUIP_ONE_BIG(32, 4)
  QUIP_ONE_BIG(24, 1) QUIP_ONE_BIG(24, 2) QUIP_ONE_BIG(24, 3) QUIP_ONE_BIG(24, 4)
  QUIP_ONE_BIG(16, 1) QUIP_ONE_BIG(16, 2) QUIP_ONE_BIG(16, 3) QUIP_ONE_BIG(16, 4)
  QUIP_ONE_BIG(40, 1) QUIP_ONE_BIG(40, 2) QUIP_ONE_BIG(40, 3) QUIP_ONE_BIG(40, 4)
  QUIP_ONE_BIG(20, 1) QUIP_ONE_BIG(20, 2) QUIP_ONE_BIG(20, 3) QUIP_ONE_BIG(20, 4)
#undef QUIP_ONE_BIG
  if (threads > 512) return QUIP_ERR_UNSUPPORTED;
#define QUIP_ONE(R, S)                                                              \
  if (rep == R && items_per_wave <= S)                                              \
    return launch<R, S, 512, G, true>(gp, grid, k, kp, nblocks, threads, dbg, stream);
  QUIP_ONE(32, 1) QUIP_ONE(32, 2) QUIP_ONE(32, 3) QUIP_ONE(32, 4) QUIP_ONE(32, 6) QUIP_ONE(32, 8)
  QUIP_ONE(24, 1) QUIP_ONE(24, 2) QUIP_ONE(24, 3) QUIP_ONE(24, 4) QUIP_ONE(24, 6) QUIP_ONE(24, 8)
  QUIP_ONE(16, 1) QUIP_ONE(16, 2) QUIP_ONE(16, 3) QUIP_ONE(16, 4) QUIP_ONE(16, 6) QUIP_ONE(16, 8)
  QUIP_ONE(64, 1) QUIP_ONE(64, 2) QUIP_ONE(64, 3) QUIP_ONE(64, 4) QUIP_ONE(64, 6) QUIP_ONE(64, 8)
  QUIP_ONE(40, 1) QUIP_ONE(40, 2) QUIP_ONE(40, 3) QUIP_ONE(40, 4) QUIP_ONE(40, 6) QUIP_ONE(40, 8)
  QUIP_ONE(20, 1) QUIP_ONE(20, 2) QUIP_ONE(20, 3) QUIP_ONE(20, 4) QUIP_ONE(20, 6) QUIP_ONE(20, 8)
#undef QUIP_ONE
  return QUIP_ERR_UNSUPPORTED;
}

// table replication for `g` x vectors of kp digits each: 32 / 32 copies when everything fits, then
// T1 x 32 + T2 x 16, then 16 / 16
static int pick_rep(int g, int kp, int forced) {
  if (forced == 64) return 64;   // D4
  if (forced == 40) return g * kp <= Lds<40>::kMaxKp ? 40 : 20;   // E8P12RVQ3B
  if (forced == 16 || g * kp > Lds<24>::kMaxKp) return 16;
  if (forced == 24 || g * kp > Lds<32>::kMaxKp) return 24;
  return 32;
}

int e8p_gemv_mfma_launch(const void* planes, const void* qidxs, const void* grid, void* y, int n, int k,
                         const GemvTune& tune, hipStream_t stream) {
  if (!e8p_gemv_mfma_supported(n, k)) return QUIP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(grid) & 63u) != 0) return QUIP_ERR_MISALIGNED;
  const int kp = kp_of(k);
  uint64_t* dbg = reinterpret_cast<uint64_t*>(tune.dbg);
  const int ncu = device_cu_count();
  int nblocks = tune.blocks > 0 ? tune.blocks : ncu;
  int rpb = (n + nblocks - 1) / nblocks;
  rpb = (rpb + 15) & ~15;
  if (rpb > kMaxRowsPerBlock) rpb = kMaxRowsPerBlock;
  nblocks = (n + rpb - 1) / rpb;
  // Measured on MI355X (tools/gemv_bench.py, us per launch): up to 8 items per wave -> one-shot mode on
  // 8 waves; longer streams -> ONE slot per wave in flight and more waves (28672 x 8192: 15.4 us with
  // 12 waves x 1 slot vs 16.6 with 8 x 2; 8192 x 28672: 18.4 with 16 x 1 vs 20.2 with 14 x 2).
  const int items = ((rpb + 15) >> 4) * (kp >> 9);
  int waves = tune.max_waves > 0 ? tune.max_waves : (items <= 64 ? 8 : (kp > 16384 ? 16 : 12));
  if (waves > 16) waves = 16;
  if (waves < 8) waves = 8;    // the table build uses waves 0..7
  const int min_waves = (3 * (kp >> 4) + 6 * 64 - 1) / (6 * 64);  // 6 x pieces per thread
  if (waves < min_waves) waves = min_waves;
  const int rep = pick_rep(1, kp, tune.rep);
  const int items_per_wave = (items + waves - 1) / waves;
  int slots = tune.rows ? tune.rows : 1;
  const int threads = waves * 64;
  if (threads > 512 && slots > 2) slots = 2;  // 128-VGPR budget: deeper queues would spill, and
                                              // scratch traffic would corrupt the counted vmcnt waits
  GemvGroup<1> gp{{reinterpret_cast<const uint4*>(qidxs)}, {reinterpret_cast<const uint8_t*>(planes)},
                  {reinterpret_cast<f16*>(y)}, {n}, {rpb}, {0}, tune.grid2};
  if ((rep == 40 || rep == 20) && !tune.grid2) return QUIP_ERR_NULL_POINTER;
  if (rep == 20 && kp > Lds<20>::kMaxKp) return QUIP_ERR_UNSUPPORTED;
  if (!tune.rows && ((threads <= 512 && items_per_wave <= 8) || (threads > 512 && rep != 64 && items_per_wave <= 4)))
    return launch_oneshot<1>(gp, grid, k, kp, nblocks, threads, rep, items_per_wave, dbg, stream);
#define QUIP_CASE(R, S)                                                                        \
  if (rep == R && slots == S && threads <= 512)                                                \
    return launch<R, S, 512, 1>(gp, grid, k, kp, nblocks, threads, dbg, stream);
#define QUIP_CASE_BIG(R, S)                                                                    \
  if (rep == R && slots == S && threads > 512)                                                 \
    return launch<R, S, 1024, 1>(gp, grid, k, kp, nblocks, threads, dbg, stream);
  QUIP_CASE(32, 1) QUIP_CASE(32, 2) QUIP_CASE(32, 3) QUIP_CASE(32, 4) QUIP_CASE(32, 6) QUIP_CASE(32, 8)
  QUIP_CASE(24, 1) QUIP_CASE(24, 2) QUIP_CASE(24, 3) QUIP_CASE(24, 4)
  QUIP_CASE(16, 1) QUIP_CASE(16, 2) QUIP_CASE(16, 3) QUIP_CASE(16, 4) QUIP_CASE(16, 6) QUIP_CASE(16, 8)
  QUIP_CASE_BIG(32, 1) QUIP_CASE_BIG(32, 2)
  QUIP_CASE_BIG(24, 1) QUIP_CASE_BIG(24, 2)
  QUIP_CASE_BIG(16, 1) QUIP_CASE_BIG(16, 2)
  QUIP_CASE(64, 1) QUIP_CASE(64, 2) QUIP_CASE_BIG(64, 1) QUIP_CASE_BIG(64, 2)
  QUIP_CASE(40, 1) QUIP_CASE(40, 2) QUIP_CASE_BIG(40, 1) QUIP_CASE_BIG(40, 2)
  QUIP_CASE(20, 1) QUIP_CASE(20, 2) QUIP_CASE_BIG(20, 1) QUIP_CASE_BIG(20, 2)
#undef QUIP_CASE
#undef QUIP_CASE_BIG
  return QUIP_ERR_UNSUPPORTED;
}

// ---- rows mode: up to 5 activation rows per launch ------------------------------------------
// mode: 0 = E8P12 tables, 64 = D4 table (also HI through its virtual layout), 40 = E8P12RVQ3B tables
int e8p_gemv_mfma_max_rows(int n, int k, int mode) {
  if (!e8p_gemv_mfma_supported(n, k)) return 0;
  const int budget = mode == 64 ? Lds<64>::kMaxKp : (mode == 40 ? Lds<20>::kMaxKp : Lds<16>::kMaxKp);
  const int m = budget / kp_of(k);
  return m > 5 ? 5 : m;
}

int e8p_gemv_mfma_rows_launch(const void* planes, const void* qidxs, const void* grid, void* y, int mrows, int n,
                              int k, const GemvTune& tune, hipStream_t stream) {
  const int mode = tune.rep == 64 ? 64 : (tune.rep == 40 ? 40 : 0);
  if (mrows < 1 || mrows > e8p_gemv_mfma_max_rows(n, k, mode)) return QUIP_ERR_UNSUPPORTED;
  if (mode != 64 && (reinterpret_cast<uintptr_t>(grid) & 63u) != 0) return QUIP_ERR_MISALIGNED;
  if (mode == 40 && !tune.grid2) return QUIP_ERR_NULL_POINTER;
  const int kp = kp_of(k);
  uint64_t* dbg = reinterpret_cast<uint64_t*>(tune.dbg);
  int nblocks = tune.blocks > 0 ? tune.blocks : device_cu_count();
  int rpb;
  for (;;) {   // accumulator rows: mrows per weight row
    rpb = (n + nblocks - 1) / nblocks;
    rpb = (rpb + 15) & ~15;
    if (mrows * rpb <= kMaxRowsPerBlock) break;
    nblocks *= 2;
  }
  nblocks = (n + rpb - 1) / rpb;
  const int items = (rpb >> 4) * (kp >> 9);
  int waves = tune.max_waves > 0 ? tune.max_waves : (items <= 64 ? 8 : 12);
  const int min_waves = (mrows * 3 * (kp >> 4) + 6 * 64 - 1) / (6 * 64);   // 6 plane pieces per thread
  if (waves < min_waves) waves = min_waves;
  if (waves < 8) waves = 8;
  if (waves > 16) return QUIP_ERR_UNSUPPORTED;
  const int threads = waves * 64;
  const int ipw = (items + waves - 1) / waves;
  const int rep = mode == 64 ? 64 : (mode == 40 ? (mrows * kp <= Lds<40>::kMaxKp ? 40 : 20)
                                               : (mrows * kp <= Lds<32>::kMaxKp ? 32 : 16));
  GemvGroup<1> gp{{reinterpret_cast<const uint4*>(qidxs)}, {reinterpret_cast<const uint8_t*>(planes)},
                  {reinterpret_cast<f16*>(y)}, {n}, {rpb}, {0}, tune.grid2};
#define QUIP_ROWS(R, S, T, ONE)                                                                   \
  if (rep == R && (T == 1024) == (threads > 512) && (ONE ? ipw <= S : true))                      \
    return launch<R, S, T, 1, ONE, false, true>(gp, grid, k, kp, nblocks, threads, dbg, stream, nullptr, mrows);
  if ((threads <= 512 && ipw <= 8) || (threads > 512 && ipw <= 4)) {
#define QUIP_ROWS_ONE(R)                                                                                          \
    QUIP_ROWS(R, 1, 512, true) QUIP_ROWS(R, 2, 512, true) QUIP_ROWS(R, 3, 512, true) QUIP_ROWS(R, 4, 512, true)   \
    QUIP_ROWS(R, 6, 512, true) QUIP_ROWS(R, 8, 512, true)                                                         \
    QUIP_ROWS(R, 1, 1024, true) QUIP_ROWS(R, 2, 1024, true) QUIP_ROWS(R, 3, 1024, true) QUIP_ROWS(R, 4, 1024, true)
    QUIP_ROWS_ONE(64) QUIP_ROWS_ONE(40) QUIP_ROWS_ONE(20)
#undef QUIP_ROWS_ONE
    QUIP_ROWS(32, 1, 512, true) QUIP_ROWS(32, 2, 512, true) QUIP_ROWS(32, 3, 512, true) QUIP_ROWS(32, 4, 512, true)
    QUIP_ROWS(32, 6, 512, true) QUIP_ROWS(32, 8, 512, true)
    QUIP_ROWS(16, 1, 512, true) QUIP_ROWS(16, 2, 512, true) QUIP_ROWS(16, 3, 512, true) QUIP_ROWS(16, 4, 512, true)
    QUIP_ROWS(16, 6, 512, true) QUIP_ROWS(16, 8, 512, true)
    QUIP_ROWS(32, 1, 1024, true) QUIP_ROWS(32, 2, 1024, true) QUIP_ROWS(32, 3, 1024, true) QUIP_ROWS(32, 4, 1024, true)
    QUIP_ROWS(16, 1, 1024, true) QUIP_ROWS(16, 2, 1024, true) QUIP_ROWS(16, 3, 1024, true) QUIP_ROWS(16, 4, 1024, true)
  }
  QUIP_ROWS(32, 1, 512, false) QUIP_ROWS(16, 1, 512, false) QUIP_ROWS(32, 1, 1024, false) QUIP_ROWS(16, 1, 1024, false)
  QUIP_ROWS(64, 1, 512, false) QUIP_ROWS(64, 1, 1024, false) QUIP_ROWS(40, 1, 512, false) QUIP_ROWS(40, 1, 1024, false)
  QUIP_ROWS(20, 1, 512, false) QUIP_ROWS(20, 1, 1024, false)
#undef QUIP_ROWS
  return QUIP_ERR_UNSUPPORTED;
}

bool e8p_gemv_mfma_group_supported(const int* ns, int count, int k) {
  if (count < 1 || count > 3) return false;
  for (int i = 0; i < count; ++i)
    if (!e8p_gemv_mfma_supported(ns[i], k)) return false;
  return count * kp_of(k) <= Lds<16>::kMaxKp;
}

template <int G>
static int group_launch(const void* const* planes, const void* const* qidxs, const void* grid, void* const* ys,
                        const int* ns, int k, const GemvTune& tune, hipStream_t stream) {
  const int kp = kp_of(k);
  const int ncu = device_cu_count();
  int nblocks = tune.blocks > 0 ? tune.blocks : ncu;
  GemvGroup<G> gp;
  gp.grid2 = tune.grid2;
  int total_rpb = 0;
  for (;;) {
    total_rpb = 0;
    for (int p = 0; p < G; ++p) {
      int rpb = (ns[p] + nblocks - 1) / nblocks;
      rpb = (rpb + 15) & ~15;
      gp.rpb[p] = rpb;
      total_rpb += rpb;
    }
    if (total_rpb <= kMaxRowsPerBlock) break;
    nblocks *= 2;   // more, smaller workgroups until the accumulator rows fit
  }
  int used = 0, items = 0;
  for (int p = 0; p < G; ++p) {
    gp.W[p] = reinterpret_cast<const uint4*>(qidxs[p]);
    gp.planes[p] = reinterpret_cast<const uint8_t*>(planes[p]);
    gp.y[p] = reinterpret_cast<f16*>(ys[p]);
    gp.N[p] = ns[p];
    used = used > (ns[p] + gp.rpb[p] - 1) / gp.rpb[p] ? used : (ns[p] + gp.rpb[p] - 1) / gp.rpb[p];
    items += (gp.rpb[p] >> 4) * (kp >> 9);
  }
  nblocks = used;
  const bool many_items = items >= 40 && items <= 48;   // one-shot on 12 waves (measured: gate/up group 10.0 vs 10.7 us)
  {   // stagger the problems that do not fill the launch
    int next = 0;
    for (int p = 0; p < G; ++p) {
      const int need = (ns[p] + gp.rpb[p] - 1) / gp.rpb[p];
      gp.boff[p] = 0;
      if (need < nblocks) {
        gp.boff[p] = next % nblocks;
        next += need;
      }
    }
  }
  int waves = tune.max_waves > 0 ? tune.max_waves : (many_items ? 12 : 8);
  if (waves > 16) waves = 16;
  if (waves < 8) waves = 8;
  const int min_waves = (G * 3 * (kp >> 4) + 6 * 64 - 1) / (6 * 64);
  if (waves < min_waves) waves = min_waves;
  if (waves > 16) return QUIP_ERR_UNSUPPORTED;
  const int rep = pick_rep(G, kp, tune.rep);
  if ((rep == 40 || rep == 20) && !tune.grid2) return QUIP_ERR_NULL_POINTER;
  if (rep == 20 && G * kp > Lds<20>::kMaxKp) return QUIP_ERR_UNSUPPORTED;
  const int slots = tune.rows ? (tune.rows >= 2 ? 2 : 1) : ((items + waves - 1) / waves >= 4 ? 2 : 1);
  const int threads = waves * 64;
  uint64_t* dbg = reinterpret_cast<uint64_t*>(tune.dbg);
  if (!tune.rows && ((threads <= 512 && (items + waves - 1) / waves <= 8) ||
                     (threads > 512 && rep != 64 && (items + waves - 1) / waves <= 4)))
    return launch_oneshot<G>(gp, grid, k, kp, nblocks, threads, rep, (items + waves - 1) / waves, dbg, stream);
#define QUIP_CASE(R, S)                                                                        \
  if (rep == R && slots == S)                                                                  \
    return threads > 512 ? launch<R, S, 1024, G>(gp, grid, k, kp, nblocks, threads, dbg, stream) \
                         : launch<R, S, 512, G>(gp, grid, k, kp, nblocks, threads, dbg, stream);
  QUIP_CASE(32, 1) QUIP_CASE(32, 2) QUIP_CASE(24, 1) QUIP_CASE(24, 2) QUIP_CASE(16, 1) QUIP_CASE(16, 2)
  QUIP_CASE(64, 1) QUIP_CASE(64, 2) QUIP_CASE(40, 1) QUIP_CASE(40, 2) QUIP_CASE(20, 1) QUIP_CASE(20, 2)
#undef QUIP_CASE
  return QUIP_ERR_UNSUPPORTED;
}

// ---- fused input side (FusedIn) ------------------------------------------------------------
bool e8p_gemv_mfma_fused_supported(const int* ns, int count, int k) {
  if (count < 1 || count > 3) return false;
  if (k < 1024 || k > 8192 || (k & (k - 1)) != 0) return false;   // K_left == 1, n / 16 FHT threads <= 512
  for (int i = 0; i < count; ++i)
    if (ns[i] < 1) return false;
  return Lds<16>::bytes_fused(k, count) <= 160 * 1024;
}

template <int G>
static int fused_launch(const GemvFusedIn& in, const void* const* qidxs, const void* grid, void* const* ys,
                        const int* ns, int k, const GemvTune& tune, hipStream_t stream) {
  const int kp = k;   // power of two >= 1024: no k padding
  const int ncu = device_cu_count();
  int nblocks = tune.blocks > 0 ? tune.blocks : ncu;
  GemvGroup<G> gp;
  gp.grid2 = nullptr;
  for (;;) {
    int total_rpb = 0;
    for (int p = 0; p < G; ++p) {
      int rpb = (ns[p] + nblocks - 1) / nblocks;
      rpb = (rpb + 15) & ~15;
      gp.rpb[p] = rpb;
      total_rpb += rpb;
    }
    if (total_rpb <= kMaxRowsPerBlock) break;
    nblocks *= 2;
  }
  int used = 0, items = 0;
  FusedIn fi{};
  fi.x = reinterpret_cast<const f16*>(in.x);
  fi.z = reinterpret_cast<const f16*>(in.z);
  fi.post = reinterpret_cast<const f16*>(in.post);
  fi.residual = reinterpret_cast<const f16*>(in.residual);
  fi.h_out = reinterpret_cast<f16*>(in.h_out);
  fi.rms_w = reinterpret_cast<const f16*>(in.rms_w);
  fi.z_scale = in.z_scale; fi.rms_eps = in.rms_eps; fi.n = k;
  fi.logn = 0;
  while ((1 << fi.logn) < k) ++fi.logn;
  for (int p = 0; p < G; ++p) {
    gp.W[p] = reinterpret_cast<const uint4*>(qidxs[p]);
    gp.planes[p] = nullptr;
    gp.y[p] = reinterpret_cast<f16*>(ys[p]);
    gp.N[p] = ns[p];
    fi.pre[p] = reinterpret_cast<const f16*>(in.pre[p]);
    fi.scale[p] = in.scale[p];
    used = used > (ns[p] + gp.rpb[p] - 1) / gp.rpb[p] ? used : (ns[p] + gp.rpb[p] - 1) / gp.rpb[p];
    items += (gp.rpb[p] >> 4) * (kp >> 9);
  }
  nblocks = used;
  for (int p = 0; p < G; ++p) gp.boff[p] = 0;
  const int waves = 8, threads = 512;       // n / 16 <= 512 transform threads
  const int ipw = (items + waves - 1) / waves;
  uint64_t* dbg = reinterpret_cast<uint64_t*>(tune.dbg);
#define QUIP_ONE(S)                                                                                   \
  if (ipw <= S) return launch<16, S, 512, G, true, true>(gp, grid, k, kp, nblocks, threads, dbg, stream, &fi);
  QUIP_ONE(1) QUIP_ONE(2) QUIP_ONE(3) QUIP_ONE(4) QUIP_ONE(6) QUIP_ONE(8)
#undef QUIP_ONE
  return launch<16, 2, 512, G, false, true>(gp, grid, k, kp, nblocks, threads, dbg, stream, &fi);
}

int e8p_gemv_mfma_fused_launch(const GemvFusedIn& in, const void* const* qidxs, const void* grid,
                               void* const* ys, const int* ns, int count, int k, const GemvTune& tune,
                               hipStream_t stream) {
  if (!e8p_gemv_mfma_fused_supported(ns, count, k)) return QUIP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(grid) & 63u) != 0) return QUIP_ERR_MISALIGNED;
  if (count == 1) return fused_launch<1>(in, qidxs, grid, ys, ns, k, tune, stream);
  if (count == 2) return fused_launch<2>(in, qidxs, grid, ys, ns, k, tune, stream);
  return fused_launch<3>(in, qidxs, grid, ys, ns, k, tune, stream);
}

int e8p_gemv_mfma_group_launch(const void* const* planes, const void* const* qidxs, const void* grid,
                               void* const* ys, const int* ns, int count, int k, const GemvTune& tune,
                               hipStream_t stream) {
  if (!e8p_gemv_mfma_group_supported(ns, count, k)) return QUIP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(grid) & 63u) != 0) return QUIP_ERR_MISALIGNED;
  if (count == 1) return e8p_gemv_mfma_launch(planes[0], qidxs[0], grid, ys[0], ns[0], k, tune, stream);
  if (count == 2) return group_launch<2>(planes, qidxs, grid, ys, ns, k, tune, stream);
  return group_launch<3>(planes, qidxs, grid, ys, ns, k, tune, stream);
}

}  // namespace quip
