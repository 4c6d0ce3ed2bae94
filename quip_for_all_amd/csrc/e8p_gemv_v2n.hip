// E8P12 decode GEMV for gfx950: the K-splitting matrix-core kernel (e8p_gemv_v2.hip) in NIBBLE MODE (round 6; the decode
// identity of e8p_gemv_core.hip.h's last section: a code's eight weights are ONE dword T1n[abs] ^ T2n[sign] of two 4-byte
// look-ups, 32 conflict-free copies of both tables in 64 KB, the matrix core takes the nibbles as they are -- twice).  Same
// integers as the byte tables: the accumulators hold exactly 8 x their sums, y is bit identical.
//
// Replaces the M = 1 use of tinygemm_m16n8k16_chunk_kernel<.., BLayout_E8, ..> (origin_order.cu:388-555, 604-648), as
// e8p_gemv_v2.hip does; what differs from that kernel:
//
//  * A dword of B is a whole code, so a lane's 16 landed bytes (8 codes) are TWO B operands, and an A operand needs six rows
//    per k-chunk (planes h, m, l of positions 4..7 of every 8-group -- "hi" -- and of positions 0..3 -- "lo").  The A rows
//    form TWO groups h = 0, 1 of (hi 0, 1, 2, spare, lo 0, 1, 2, spare); B column n = 8 h + r carries weight row r of a row
//    OCTET, chunk 4 h + q.  One load instruction covers 8 weight rows x 128 contiguous bytes (a whole line each).  A segment
//    (the unit of the K walk) is 512 k.
//  * Two accumulators: R1 += A x raw, R2 += A x (raw & 0x0f0f0f0f).  A run's contribution to a row is (R1 - R2) from the lanes
//    that hold the "hi" rows of its column and 16 R2 from the lanes that hold the "lo" rows.
//  * The constant part 8 SX[hi] - 120 SX[lo] (SX = digit sums; e8p_gemv_core.hip.h) depends on the K range only: every
//    workgroup computes it ONCE for its range -- MFMAs of the digit image against a B operand of ones, the segments dealt
//    over the waves -- and adds it to every row in the epilogue.
//  * Tables are 64 KB whatever K is: rows of 28672 (86 KB of digits) keep conflict-free look-ups, where the byte tables fit
//    with 16 copies only (two-way conflicts on every ds_read_b64).
#include <type_traits>

#include "e8p_gemv_core.hip.h"

namespace quip {

namespace {

constexpr int kSegBytesN = 1536;  // LDS bytes of the digit image per 512 k: 8 (q, t) rows x 12 units x 16 B
constexpr int kMaxGN = 3;
constexpr int kTablesN = kNibTableBytes;

struct V2nArgs {
  const uint4* W[kMaxGN];          // (N, K / 8) int16 codes
  const uint8_t* planes[kMaxGN];   // [3][kp_src] digit bytes + int32 shift word at 3 * kp_src
  f16* y[kMaxGN];
  int* ws[kMaxGN];                 // ksplit > 1: zeroed int32 [N][4] accumulators of every problem, back to back
  int* cnt;                        // ... followed by the [row blocks] arrival counters of the launch
  int N[kMaxGN];
  int rpb[kMaxGN];                 // rows per workgroup (multiple of 8)
  const uint64_t* grid;            // grid_packed_abs
  int K;
  int kp_src;                      // digits per plane in `planes` (K rounded up to 512)
  int segs;                        // 512-k segments of a row (ceil)
  int spw;                         // segments per workgroup (K split)
  int ksplit;
  int runlen;                      // segments per run
  int rpr_inv;                     // 2^20 / (runs per octet of a full K part) + 1: run / rpr without a division
  uint64_t* dbg;
};

// grid = (K parts, row blocks)
template <int SLOTS, int G>
__global__ __launch_bounds__(1024) void e8p_gemv_v2n_kernel(V2nArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int rb = (int)blockIdx.y, ks = (int)blockIdx.x, wg = rb * a.ksplit + ks;
#define V2_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[wg * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  V2_STAMP(0);
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = __builtin_amdgcn_readfirstlane(nthreads >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int r = n & 7, h = n >> 3;
  const int seg0 = ks * a.spw;
  const int S = min(a.segs, seg0 + a.spw) - seg0;     // segments of this workgroup
  const int rpr = (S + a.runlen - 1) / a.runlen;      // runs per row octet
  // (measured and dropped: the last quarter of every octet as runs a quarter as long, handed out after the long ones -- the
  //  shorter tail did not pay for the extra flushes: 15.9 against 15.3 us at 28672 x 8192, profiles/r06_gemv_v2_nibble.txt)
  const int row_u4 = a.K >> 6;                        // uint4 per packed row
  int row0[G], rows_here[G], qbase[G + 1], rbase[G];  // first row, rows, first row octet / accumulator row of a problem
  qbase[0] = 0;
#pragma unroll
  for (int p = 0; p < G; ++p) {
    row0[p] = rb * a.rpb[p];
    rows_here[p] = max(0, min(a.N[p], row0[p] + a.rpb[p]) - row0[p]);
    qbase[p + 1] = qbase[p] + ((rows_here[p] + 7) >> 3);
    rbase[p] = p == 0 ? 0 : rbase[p - 1] + a.rpb[p - 1];
  }
  const int nruns = qbase[G] * rpr;
  const uint32_t xbase = (uint32_t)kTablesN;
  const uint32_t accbase = xbase + (uint32_t)(G * S) * kSegBytesN;
  int* accs = reinterpret_cast<int*>(smem + accbase);
  const int accwords = (rbase[G - 1] + a.rpb[G - 1]) * 4;
  int* corr = accs + accwords;          // [G][4]: 8 SX[hi] - 120 SX[lo] of this workgroup's K range, per plane
  int* counter = corr + 4 * G;

  // (0) loads, in the order in which they are needed (VMEM returns in issue order); everything is counted
  int sh[G];
#pragma unroll
  for (int p = 0; p < G; ++p)
    asm volatile("global_load_dword %0, %1, off" : "=v"(sh[p]) : "v"(a.planes[p] + (size_t)3 * a.kp_src) : "memory");
  u32x2 tsrc;
  // table rows: 16 waves build 16 rows each (lane = 16 g + l: row 16 w + l; g & 1: the sign table; g >> 1: which 16 of the 32
  // copies), fewer waves 32 rows each in waves 0..7 (lanes 0..31 / 32..63: T1n / T2n, all 32 copies)
  const bool tw16 = nwaves >= 16;
  const int te = tw16 ? wave * 16 + (lane & 15) : (wave & 7) * 32 + (lane & 31);
  const bool tsecond = tw16 ? ((lane >> 4) & 1) != 0 : (lane & 32) != 0;
  {
    const uint2* t1 = reinterpret_cast<const uint2*>(a.grid) + te;
    const uint2* t2 = reinterpret_cast<const uint2*>(&kT2nImg.v[te & ~1]);
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(tsrc) : "v"(tsecond ? t2 : t1) : "memory");
  }
  // filler for the slots that have nothing to fetch (the load counts are compile-time constants): ONE 16-byte address for the
  // whole wave -- a filler with 64 addresses costs the vector L1 what a real request costs
  const uint4* hot = reinterpret_cast<const uint4*>(a.planes[0]);
  // digit images, requested BEFORE the weights (e8p_gemv_v2.hip).  A thread takes k16 index g (16 digits) of ALL planes of ALL
  // problems: one index decode for 3 G requests, the planes' bases are scalars (round 6: a decode per request -- ~40 VALU x 6 --
  // was 1.7K of the 2.7K clocks between the kernel's start and its first weight request).  Every workgroup starts at a
  // different index so that they do not all queue on the same L2 channels; threads without an index re-read piece 0.
  constexpr int NG = G == 1 ? 2 : 1;        // k16 indices per thread: K parts of up to 32768 k (one problem) / 16384 k
  constexpr int XR = 3 * G * NG;
  const int gper = S * 32;                  // k16 indices in this workgroup's K range
  const int src_pieces = a.kp_src >> 4;     // pieces per plane in the source
  const int rot = (int)(((uint32_t)wg * 5u) & 31u) * (gper >> 5);
  u32x4 xr[XR];
  uint32_t xdst[NG];                        // LDS destination of the "hi" unit's 8 bytes (problem 0, plane 0); 0xffffffff: none; bit 31: zeros
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int i = tid + j * nthreads;
    int g = i + rot;
    g = g >= gper ? g - gper : g;
    g = i < gper ? g : 0;
    // index g of the range: segment g >> 5; inside it chunk (g >> 2) & 7 = 4 h' + q', 32-k step t = (g >> 1) & 1, code pair g & 1
    const int sgi = g >> 5, hh = (g >> 4) & 1, qq = (g >> 2) & 3, t = (g >> 1) & 1, pr = g & 1;
    const int sp = seg0 * 32 + g;
    const bool real = sp < src_pieces;      // beyond the source's zero padding: zeros
    const uint32_t voff = (uint32_t)(real ? sp : 0) << 4;
#pragma unroll
    for (int p = 0; p < G; ++p) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xr[(j * G + p) * 3 + d]) : "v"(voff), "s"(a.planes[p] + (size_t)d * a.kp_src) : "memory");
    }
    const uint32_t dst = xbase + (uint32_t)sgi * kSegBytesN + (uint32_t)(((qq * 2 + t) * 12 + 6 * hh) * 16 + 8 * pr);
    xdst[j] = i < gper ? (dst | (real ? 0u : 0x80000000u)) : 0xffffffffu;
  }

  // run -> (problem, row octet, first segment, length); everything wave uniform
  auto problem_of_octet = [&](int gq) -> int {
    int p = 0;
#pragma unroll
    for (int i = 1; i < G; ++i) p += gq >= qbase[i] ? 1 : 0;
    return p;
  };
  auto pick = [&](const int* arr, int p) -> int {   // arr[p] without dynamic indexing (scratch accesses are VMEM)
    int v = arr[0];
#pragma unroll
    for (int i = 1; i < G; ++i) {
      v = p == i ? arr[i] : v;
      asm volatile("" : "+s"(v));
    }
    return v;
  };
  // load cursor (e8p_gemv_v2.hip: branch free on purpose)
  int l_run = wave;                   // current run (>= nruns: none)
  int l_gq = 0, l_seg = 0, l_left = 0;
  const char* l_ptr = reinterpret_cast<const char*>(hot);   // this lane's 16 bytes of the run's next unit
  int l_mgq = 0, l_xoff = 0;          // accumulator row octet / digit image offset of the run's next unit
  const bool ragged = (a.K & 511) != 0;    // the row's last segment is partial: lanes past the row re-read a valid piece
                                           // (their digits are zero)
  auto open_run = [&](int run) __attribute__((always_inline)) {
    l_run = run;
    const bool ok = run < nruns;
    const int rc = ok ? run : 0;
    // rc / rpr by the host's reciprocal (rc rpr < 2^20: exact); a short last K part has its own rpr -- the plain division there
    l_gq = rpr == a.rpr_inv >> 24 ? (int)(((uint32_t)rc * (uint32_t)(a.rpr_inv & 0xffffff)) >> 20) : __builtin_amdgcn_readfirstlane(rc / rpr);
    const int ri = rc - l_gq * rpr;
    l_seg = ri * a.runlen;
    l_left = ok ? min(a.runlen, S - l_seg) : 0;
    const int p = __builtin_amdgcn_readfirstlane(problem_of_octet(l_gq));
    const int qb = pick(qbase, p);
    int row = pick(row0, p) + 8 * (l_gq - qb) + r;
    const int N = pick(a.N, p);
    row = row < N ? row : N - 1;
    const uint4* W = a.W[0];
#pragma unroll
    for (int i = 1; i < G; ++i) {
      W = p == i ? a.W[i] : W;
      asm volatile("" : "+s"(W));
    }
    l_ptr = reinterpret_cast<const char*>(W) + ((size_t)row * row_u4 + (seg0 + l_seg) * 8 + 4 * h + q) * 16;
    l_mgq = (pick(rbase, p) >> 3) + (l_gq - qb);
    l_xoff = __builtin_amdgcn_readfirstlane((p * S + l_seg) * kSegBytesN);
  };
  open_run(wave);
  int s_gq[SLOTS], s_x[SLOTS], s_flag[SLOTS];   // flag: 0 filler, 1 unit, 3 unit that ends its run
  auto issue = [&](u32x4& dst, int& m_gq, int& m_x, int& m_flag) __attribute__((always_inline)) {
    const bool real = l_left > 0;   // wave uniform
    const char* ptr = real ? l_ptr : reinterpret_cast<const char*>(hot);
    if (ragged && real && seg0 + l_seg == a.segs - 1) {   // wave uniform condition
      const int off = (seg0 + l_seg) * 8 + 4 * h + q;
      ptr = off < row_u4 ? ptr : ptr - (4 * h + q) * 16;
    }
    asm_load16_nt(dst, reinterpret_cast<const uint4*>(ptr));
    m_gq = l_mgq;
    m_x = l_xoff;
    m_flag = real ? (l_left == 1 ? 3 : 1) : 0;
    if (real) {
      ++l_seg;
      --l_left;
      l_ptr += 8 * 16;
      l_xoff += kSegBytesN;
    }
  };
  auto refill = [&]() __attribute__((always_inline)) {
    if (l_left == 0 && l_run < nruns) {
      int nxt = 0;
      if (lane == 0) nxt = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      open_run(__builtin_amdgcn_readfirstlane(nxt) + nwaves);
    }
  };
  u32x4 slot[SLOTS];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) issue(slot[i], s_gq[i], s_x[i], s_flag[i]);
  V2_STAMP(1);

  // (1) accumulators + correction words + run counter, tables
  for (int i = tid; i <= accwords + 4 * G; i += nthreads) accs[i] = 0;
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tsrc) : "n"(XR + SLOTS) : "memory");
  {
    // every lane writes its entry 16 / 32 times, copy (l + c) & 31 at step c: the lanes of a step are on distinct banks (rows
    // are 256 bytes apart -- the same banks --, T1n on banks 0..31, T2n on 32..63, the copy index picks the bank)
    const uint32_t val = tsecond ? ((te & 1) ? tsrc.y : tsrc.x) : t1n_entry(make_uint2(tsrc.x, tsrc.y));
    const uint32_t rowbase = (uint32_t)te * 256u + (tsecond ? 128u : 0u);
    if (tw16) {
      const uint32_t c0 = (uint32_t)(lane & 15) + (uint32_t)((lane >> 5) << 4);
#pragma unroll
      for (int c = 0; c < 16; ++c)
        *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)(rowbase + ((c0 + (uint32_t)c) & 31u) * 4u)) = val;
    } else if (wave < 8) {
#pragma unroll
      for (int c = 0; c < 32; ++c)
        *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)(rowbase + (((uint32_t)(lane + c)) & 31u) * 4u)) = val;
    }
  }
#pragma unroll
  for (int p = 0; p < G; ++p) asm volatile("" : "+v"(sh[p]));   // landed before tsrc
  V2_STAMP(2);

  // (2) digit images into LDS in fragment order: unit ((q * 2 + t) * 12 + 6 h + 3 lo + d) of segment s holds plane d,
  //     positions 4..7 (lo = 0) / 0..3 (lo = 1) of the four 8-groups at k = 512 s + 64 (4 h + q) + 32 t
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(SLOTS) : "memory");
#pragma unroll
  for (int j = 0; j < XR; ++j) asm volatile("" : "+v"(xr[j]));
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    if (xdst[j] != 0xffffffffu) {
      const bool z = (xdst[j] & 0x80000000u) != 0;
      const uint32_t at0 = xdst[j] & 0x7fffffffu;
#pragma unroll
      for (int p = 0; p < G; ++p) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const u32x4 v = xr[(j * G + p) * 3 + d];
          const uint32_t at = at0 + (uint32_t)(p * S) * kSegBytesN + 16u * d;
          const u32x2 hi = z ? u32x2{0u, 0u} : u32x2{v.y, v.w}, lo = z ? u32x2{0u, 0u} : u32x2{v.x, v.z};
          *reinterpret_cast<__attribute__((address_space(3))) u32x2*>((uintptr_t)at) = hi;
          *reinterpret_cast<__attribute__((address_space(3))) u32x2*>((uintptr_t)(at + 48u)) = lo;
        }
      }
    }
  }
  __syncthreads();
  V2_STAMP(3);

  const uint32_t lane_c = nib_lane_const(lane);
  // A fragment of this lane (A row m = lane & 15 = 8 h' + j, k block q): unit (q * 2 + t) * 12 + 6 h' + (j < 4 ? min(j, 2) : 3 + min(j - 4, 2))
  const uint32_t xlane = xbase + (uint32_t)((q * 24 + 6 * h + (r < 4 ? min(r, 2) : 3 + min(r - 4, 2))) * 16);

  // (2b) the constant part of this workgroup's K range: digit sums by the matrix core, segments dealt over the waves
  {
    const i32x4 ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
#pragma unroll
    for (int p = 0; p < G; ++p) {
      i32x4 sx = {0, 0, 0, 0};
      for (int s = wave; s < S; s += nwaves) {
        const uint32_t xa = xlane + (uint32_t)((p * S + s) * kSegBytesN);
        const i32x4 A0 = lds_read16i(xa), A1 = lds_read16i(xa + 192u);
        sx = __builtin_amdgcn_mfma_i32_16x16x64_i8(A0, ones, sx, 0, 0, 0);
        sx = __builtin_amdgcn_mfma_i32_16x16x64_i8(A1, ones, sx, 0, 0, 0);
      }
      // column 0: lane (0, q') holds rows 4 q' + i -- q' even: the "hi" rows of group q' / 2, odd: its "lo" rows
      if (n == 0 && wave < S) {
        const int f = (q & 1) ? -120 : 8;
        __hip_atomic_fetch_add(corr + 4 * p + 0, f * sx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(corr + 4 * p + 1, f * sx.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(corr + 4 * p + 2, f * sx.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }

  // this lane's D registers 0..2 of column n = 8 h + r: rows 4 q + i; valid when q >> 1 == h ("hi" rows for even q, "lo" for odd)
  const bool dvalid = (q >> 1) == h;
  const int cm = (q & 1) ? 16 : -1;
  const uint32_t rmask = (q & 1) ? 0u : 0xffffffffu;

  // (3) the stream
  i32x4 acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, prev1 = {0, 0, 0, 0}, prev2 = {0, 0, 0, 0};
  bool more = true;
  while (more) {
    more = false;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(slot[i]) : "n"(SLOTS - 1) : "memory");
      uint32_t a1l[4], a2l[4], a1h[4], a2h[4];
      const uint32_t dw[4] = {slot[i].x, slot[i].y, slot[i].z, slot[i].w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a1l[t] = __builtin_amdgcn_perm(dw[t], lane_c, 0x0c0c0500u);
        a2l[t] = __builtin_amdgcn_perm(dw[t], lane_c, 0x0c0c0402u);
        a1h[t] = __builtin_amdgcn_perm(dw[t], lane_c, 0x0c0c0700u);
        a2h[t] = __builtin_amdgcn_perm(dw[t], lane_c, 0x0c0c0602u);
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(a1l[t]), "+v"(a2l[t]), "+v"(a1h[t]), "+v"(a2h[t]));
      const int gq = s_gq[i], xo = s_x[i], flag = s_flag[i];
      refill();
      issue(slot[i], s_gq[i], s_x[i], s_flag[i]);
      more = more || s_flag[i] != 0;
      if (flag) {   // wave uniform
        const uint32_t xa = xlane + (uint32_t)xo;
        uint32_t o[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          o[t][0] = lds_read4(a1l[t]); o[t][1] = lds_read4(a2l[t]);
          o[t][2] = lds_read4(a1h[t]); o[t][3] = lds_read4(a2h[t]);
        }
        const i32x4 A0 = lds_read16i(xa), A1 = lds_read16i(xa + 192u);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const i32x4 Br = {(int)(o[2 * t][0] ^ o[2 * t][1]), (int)(o[2 * t][2] ^ o[2 * t][3]),
                            (int)(o[2 * t + 1][0] ^ o[2 * t + 1][1]), (int)(o[2 * t + 1][2] ^ o[2 * t + 1][3])};
          const i32x4 Bm = {Br.x & 0x0f0f0f0f, Br.y & 0x0f0f0f0f, Br.z & 0x0f0f0f0f, Br.w & 0x0f0f0f0f};
          acc1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(t == 0 ? A0 : A1, Br, acc1, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(t == 0 ? A0 : A1, Bm, acc2, 0, 0, 0);
        }
        if (flag & 2) {   // run finished: hand its sums to the LDS accumulators
          // (the MFMA accumulators are never reset: a run's sums are the difference to the accumulators at the previous flush --
          //  e8p_gemv_v2.hip; the asm reads VALU results only)
          const int d1x = acc1.x - prev1.x, d1y = acc1.y - prev1.y, d1z = acc1.z - prev1.z;
          const int d2x = acc2.x - prev2.x, d2y = acc2.y - prev2.y, d2z = acc2.z - prev2.z;
          asm volatile("v_add_u32 %0, %0, %3\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %5"
                       : "+v"(prev1.x), "+v"(prev1.y), "+v"(prev1.z)
                       : "v"(d1x), "v"(d1y), "v"(d1z));
          asm volatile("v_add_u32 %0, %0, %3\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %5"
                       : "+v"(prev2.x), "+v"(prev2.y), "+v"(prev2.z)
                       : "v"(d2x), "v"(d2y), "v"(d2z));
          if (dvalid) {
            int* dst = accs + (gq * 8 + r) * 4;
            __hip_atomic_fetch_add(dst + 0, (int)((uint32_t)d1x & rmask) + cm * d2x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 1, (int)((uint32_t)d1y & rmask) + cm * d2y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 2, (int)((uint32_t)d1z & rmask) + cm * d2z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          }
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the trailing filler loads
  V2_STAMP(4);
  if (a.dbg && lane == 0)   // slot 7: the last wave to leave the stream
    atomicMax(reinterpret_cast<unsigned long long*>(a.dbg + wg * 8 + 7), (unsigned long long)__builtin_amdgcn_s_memtime());
  __syncthreads();
  V2_STAMP(5);

  // (4) y = 2^(-sh-5) (65536 S_h + 256 S_m + S_l), one fp16 rounding  (the sums are 8 x the byte tables')
  if (a.ksplit == 1) {
#pragma unroll
    for (int p = 0; p < G; ++p) {
      const float unscale = unscale_of(sh[p], 5);
      const int c0 = corr[4 * p], c1 = corr[4 * p + 1], c2 = corr[4 * p + 2];
      for (int t = tid; t < rows_here[p]; t += nthreads) {
        const int* s3 = accs + (rbase[p] + t) * 4;
        const float f = __builtin_fmaf((float)(s3[0] + c0), 65536.f, __builtin_fmaf((float)(s3[1] + c1), 256.f, (float)(s3[2] + c2)));
        a.y[p][row0[p] + t] = (f16)(f * unscale);
      }
    }
  } else {
    // partial sums of this K range -> workspace (agent-scope integer atomics: exact, order independent)
#pragma unroll
    for (int p = 0; p < G; ++p) {
      const int c0 = corr[4 * p], c1 = corr[4 * p + 1], c2 = corr[4 * p + 2];
      for (int t = tid; t < rows_here[p]; t += nthreads) {
        const int* s3 = accs + (rbase[p] + t) * 4;
        int* g = a.ws[p] + (size_t)(row0[p] + t) * 4;
        __hip_atomic_fetch_add(g + 0, s3[0] + c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(g + 1, s3[1] + c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(g + 2, s3[2] + c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's atomics have been performed
    __syncthreads();                                    // ... and everybody's
    int* cnt = a.cnt + rb;
    int* flag = accs;                                   // LDS word, free after the barrier
    if (tid == 0) *flag = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*flag == a.ksplit - 1) {                        // last to arrive: every partial sum is in
#pragma unroll
      for (int p = 0; p < G; ++p) {
        const float unscale = unscale_of(sh[p], 5);
        for (int t = tid; t < rows_here[p]; t += nthreads) {
          int* g = a.ws[p] + (size_t)(row0[p] + t) * 4;
          const int s0 = __hip_atomic_exchange(g + 0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int s1 = __hip_atomic_exchange(g + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int s2 = __hip_atomic_exchange(g + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const float f = __builtin_fmaf((float)s0, 65536.f, __builtin_fmaf((float)s1, 256.f, (float)s2));
          a.y[p][row0[p] + t] = (f16)(f * unscale);
        }
      }
      if (tid == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  V2_STAMP(6);
#undef V2_STAMP
}

template <int SLOTS, int G>
int v2n_launch(const V2nArgs& a, int nrb, int threads, int lds, hipStream_t stream) {
  auto kern = e8p_gemv_v2n_kernel<SLOTS, G>;
  static DynLdsCache configured;   // per instantiation, per device
  if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.ksplit, nrb), dim3(threads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

template <int G>
int v2n_group_launch(const void* const* planes, const void* const* qidxs, const void* grid, void* const* ys,
                     void* ws, const int* ns, int k, const GemvTune& tune, hipStream_t stream) {
  const int ncu = device_cu_count();
  const int segs = (k + 511) >> 9;
  const int slots = tune.rows > 0 ? tune.rows : 2;      // (measured, 58.7 MB shapes: 2 < 3 < 4 < 6 < 8 slots)
  // the smallest K split whose digit images fit beside the tables (combining partial sums across workgroups costs two more
  // memory round trips at the end of the launch)
  int ksplit = 0, nrb = 0, spw = 0, rpb[kMaxGN] = {0, 0, 0}, rows = 0, octets = 0;
  for (int ks = tune.waves_g > 0 ? tune.waves_g : 1; ks <= segs && !ksplit; ++ks) {
    int nrb_c = (tune.blocks > 0 ? tune.blocks : ncu) / ks;
    if (nrb_c < 1) nrb_c = 1;
    int rp_c[kMaxGN] = {0, 0, 0}, need = 1;
    for (;;) {   // accumulator rows must fit: more row blocks until they do
      rows = 0; octets = 0; need = 1;
      for (int p = 0; p < G; ++p) {
        int v = (ns[p] + nrb_c - 1) / nrb_c;
        v = (v + 7) & ~7;
        rp_c[p] = v;
        rows += v;
        octets += v >> 3;
        const int nb = (ns[p] + v - 1) / v;
        need = nb > need ? nb : need;
      }
      if (rows <= 1024) break;
      nrb_c *= 2;
    }
    const int spw_c = (segs + ks - 1) / ks;
    const int room = (160 * 1024 - kTablesN - rows * 16 - 16 * kMaxGN - 16) / kSegBytesN;
    if (G * spw_c > room || spw_c * 32 > (G == 1 ? 2 : 1) * 1024) continue;      // (1 or 2 k16 indices per thread)
    ksplit = ks; nrb = need; spw = spw_c;
    for (int p = 0; p < G; ++p) rpb[p] = rp_c[p];
  }
  if (!ksplit) return QUIP_ERR_UNSUPPORTED;
  ksplit = (segs + spw - 1) / spw;
  if (ksplit > 1 && !ws) return QUIP_ERR_NULL_POINTER;
  V2nArgs a;
  size_t ws_off = 0;
  for (int p = 0; p < kMaxGN; ++p) {
    const int pp = p < G ? p : 0;
    a.W[p] = reinterpret_cast<const uint4*>(qidxs[pp]);
    a.planes[p] = reinterpret_cast<const uint8_t*>(planes[pp]);
    a.y[p] = reinterpret_cast<f16*>(ys[pp]);
    a.N[p] = ns[pp];
    a.rpb[p] = rpb[pp];
    a.ws[p] = ws ? reinterpret_cast<int*>(ws) + ws_off : nullptr;
    if (p < G) ws_off += (size_t)ns[p] * 4;        // accumulators back to back; the counters follow the last one
  }
  a.cnt = ws ? reinterpret_cast<int*>(ws) + ws_off : nullptr;
  a.grid = reinterpret_cast<const uint64_t*>(grid);
  a.K = k;
  a.kp_src = (k + 511) & ~511;
  a.segs = segs; a.spw = spw; a.ksplit = ksplit;
  a.dbg = reinterpret_cast<uint64_t*>(tune.dbg);
  int waves = tune.max_waves > 0 ? tune.max_waves : (octets * spw >= 128 ? 16 : 12);
  if (waves < 8) waves = 8;     // the table build uses waves 0..7
  if (waves > 16) waves = 16;
  while (waves < 16 && spw * 32 > (G == 1 ? 2 : 1) * waves * 64) ++waves;   // k16 indices per thread
  // run length: the longest that still leaves about two runs per wave (measured: 28672 x 8192 4 = 8 segments, 8192 x 28672 7-8 > 4 > 2)
  int runlen = tune.digits > 0 ? tune.digits : spw;
  if (tune.digits <= 0)
    while (runlen > slots && octets * ((spw + runlen - 1) / runlen) < 2 * waves) runlen = (runlen + 1) / 2;
  if (runlen > spw) runlen = spw;
  if (runlen < 1) runlen = 1;
  a.runlen = runlen;
  {
    const int rpr = (spw + runlen - 1) / runlen;
    a.rpr_inv = (rpr << 24) | (((1 << 20) / rpr + 1) & 0xffffff);
  }
  const int threads = waves * 64;
  const int lds = kTablesN + G * spw * kSegBytesN + rows * 16 + 16 * G + 16;
#define QUIP_V2N(S) if (slots == S) return v2n_launch<S, G>(a, nrb, threads, lds, stream);
  QUIP_V2N(2) QUIP_V2N(3) QUIP_V2N(4)
#undef QUIP_V2N
  return QUIP_ERR_UNSUPPORTED;
}

}  // namespace

int e8p_gemv_v2n_group_launch(const void* const* planes, const void* const* qidxs, const void* grid, void* const* ys,
                              void* ws, const int* ns, int count, int k, const GemvTune& tune, hipStream_t stream) {
  if (count < 1 || count > kMaxGN) return QUIP_ERR_UNSUPPORTED;
  for (int i = 0; i < count; ++i)
    if (!e8p_gemv_v2_supported(ns[i], k)) return QUIP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(grid) & 63u) != 0) return QUIP_ERR_MISALIGNED;
  if (count == 1) return v2n_group_launch<1>(planes, qidxs, grid, ys, ws, ns, k, tune, stream);
  if (count == 2) return v2n_group_launch<2>(planes, qidxs, grid, ys, ws, ns, k, tune, stream);
  return v2n_group_launch<3>(planes, qidxs, grid, ys, ws, ns, k, tune, stream);
}

}  // namespace quip
