// E8P12 decode GEMV for gfx950, second generation of the matrix-core kernel (e8p_gemv_mfma.hip):
// same arithmetic (4w = T1[abs] ^ T2[sign] as int8, x as three balanced int8 digit planes, exact int32
// sums from v_mfma_i32_16x16x64_i8, one fp16 rounding of y -- every output bit identical to the first
// kernel's), different mapping.
//
// Replaces the M = 1 use of tinygemm_m16n8k16_chunk_kernel<.., BLayout_E8, ..> (origin_order.cu:388-555,
// 604-648).
//
// What changed, and why (profiles/r01_*: the first kernel streamed at the rate of its access pattern,
// 16 rows x 64 B per load instruction, and spent 3-4 us per launch outside the stream):
//
//  * Whole-line loads.  The MFMA has 16 A rows and x needs three (its digit planes).  Here the A rows form
//    FOUR groups h = 0..3 of (plane 0, 1, 2, spare); group h holds the digits of k-chunk 4h + q where the
//    plain layout holds chunk q.  B column n = 4h + r then carries weight row r, chunk 4h + q, and only
//    D rows 4h..4h+2 of column 4h + r mean anything: S_d of row r over the chunks of group h.  So ONE load
//    instruction covers 4 weight rows x 256 contiguous bytes (lane (n, q): row n & 3, 16-byte chunk
//    4 (n >> 2) + q) and feeds four MFMAs without any lane exchange.
//  * A wave walks along K for the same four rows ("run": up to `runlen` consecutive 1024-k segments of a row
//    quad) and keeps the sums in the MFMA accumulator; it touches the LDS accumulators once per run.
//    Runs are handed out by an LDS counter, so that the waves the SIMD arbiters favour take more of them
//    and the workgroup's tail stays short.
//  * x digits in LDS are stored in fragment order ([q][t][3h + d] 16-byte units per 1024 k), which makes the
//    ds_read_b128 of the A fragments bank-conflict free (plane-major storage put planes 0 / 1 / 2 of one
//    k on the same banks).
//  * K split across workgroups (ksplit > 1) whenever the digit image of the whole row does not fit beside
//    the tables: partial sums are integers, so they are combined with agent-scope integer atomics in a
//    caller-provided zeroed workspace and the last workgroup to arrive converts, stores y and leaves the
//    workspace zeroed again.  Exact, order independent, bit identical to the unsplit launch.
//  * Up to three problems of the same K per launch (q/k/v, gate/up): tables, launch and drain paid once.
//  * SLOTS load instructions (1 KiB each) per wave are in flight from the first instructions of the kernel,
//    so the table build runs in the shadow of the first HBM burst.
#include <cstdlib>
#include <type_traits>

#include "quip_device.hip.h"
#include "quip_internal.h"

namespace quip {

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2* lds_u2_ptr;
typedef const __attribute__((address_space(3))) i32x4* lds_i4_ptr;

constexpr int kSegBytes = 3072;   // LDS bytes of the digit image per 1024 k: 16 (q, t) rows x 12 units x 16 B
constexpr int kMaxG = 3;

// LDS map: T1 (REP1 copies) at 0, T2 (REP2 copies) behind it, then the digit images of this workgroup's K range
// (one per problem) and the int32 accumulators [rows][4] (+ one word: the run counter).
// (REP1, REP2) = (32, 32): every table lookup conflict free, 128 KiB; (32, 16): two-way conflicts on the sign
// lookups, 96 KiB; (16, 16): two-way on both, 64 KiB (short launches: half the table build).
// RVQ3 (E8P12RVQ3B, see e8p_gemv_mfma.hip table modes 40 / 20): a third table T3 = the 256 x 8-byte E81B residual
// entries (4r as int8), 16 copies, behind T2; the weight stream is the checkpoint's 3-byte codes (12-byte slots).
template <int REP1, int REP2, bool RVQ3 = false>
struct V2Lds {
  static constexpr int kT2 = 256 * REP1 * 8;
  static constexpr int kT3 = kT2 + 256 * REP2 * 8;
  static constexpr int kRep3 = 16;
  static constexpr int kX = kT3 + (RVQ3 ? 256 * kRep3 * 8 : 0);
  static constexpr int kTotal = 160 * 1024;
};

// a lane's 12 landed bytes = four 3-byte codes [resid8, e8p_lo, e8p_hi] -> the dwords (main16 << 16 | resid8 << 8)
__device__ __forceinline__ u32x4 v2_rvq3_dwords(const u32x3& w) {
  return u32x4{w.x << 8, __builtin_amdgcn_perm(w.y, w.x, 0x0504030cu), __builtin_amdgcn_perm(w.z, w.y, 0x0403020cu),
               w.z & 0xffffff00u};
}
__device__ __forceinline__ u32x4 v2_rvq3_dwords(const u32x4& w) { return w; }

__device__ __forceinline__ uint2 v2_lds_read8(uint32_t addr) {
  const u32x2 v = *reinterpret_cast<lds_u2_ptr>((uintptr_t)addr);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ i32x4 v2_lds_read16i(uint32_t addr) {
  return *reinterpret_cast<lds_i4_ptr>((uintptr_t)addr);
}

// compile-time image of the sign table: entry s = XOR mask that turns the bytes 4a | 1 of an abs entry
// into 4w (negation of a byte whose low bits are 11 is ^0xFC; the odd-parity shift -2 is ^0x02);
// restates decode8weights, origin_order.cu:211-253
struct V2T2Image {
  uint2 v[256];
  constexpr V2T2Image() : v{} {
    for (int s = 0; s < 256; ++s) {
      int par = 0;
      for (int b = 0; b < 8; ++b) par ^= (s >> b) & 1;
      const int sv = s ^ par;
      uint32_t lo = 0, hi = 0;
      for (int p = 0; p < 4; ++p) {
        lo |= (((sv >> (7 - e8p_byte_of_pos(p))) & 1) ? 0xfcu : 0u) << (8 * p);
        hi |= (((sv >> (7 - e8p_byte_of_pos(p + 4))) & 1) ? 0xfcu : 0u) << (8 * p);
      }
      const uint32_t sh = par ? 0x02020202u : 0u;
      v[s].x = lo ^ sh;
      v[s].y = hi ^ sh;
    }
  }
};
__device__ const V2T2Image kV2T2Img{};

struct V2Args {
  const uint4* W[kMaxG];          // (N, K / 8) int16 codes
  const uint8_t* planes[kMaxG];   // [3][kp_src] digit bytes + int32 shift word at 3 * kp_src
  f16* y[kMaxG];
  int* ws[kMaxG];                 // ksplit > 1: zeroed int32 [N][4] accumulators of every problem, back to back
  int* cnt;                       // ... followed by the [row blocks] arrival counters of the launch (after ALL accumulators:
                                  // a launch whose first problem is small has more row blocks than that problem has words)
  int N[kMaxG];
  int rpb[kMaxG];                 // rows per workgroup (multiple of 4)
  const uint64_t* grid;           // grid_packed_abs
  const uint64_t* grid2;          // RVQ3: the E81B table, int8 [256][8] (4r); else unused
  int K;
  int kp_src;                     // digits per plane in `planes` (K rounded up to 512)
  int segs;                       // 1024-k segments of a row (ceil)
  int spw;                        // segments per workgroup (K split)
  int ksplit;
  int runlen;                     // segments per run
  int rpr_inv;                    // (runs per quad of a full K part) << 24 | 2^20 / that + 1: run / rpr without a division
  int d4;                         // D4 table mode (also HI through its virtual layout): `grid` = the fp16 (256, 4) table; a
                                  // 16-bit "code" is two D4 code bytes -- T2[low byte] = (4w of weights 0..3, 0),
                                  // T1[high byte] = (0, 4w of weights 4..7), so T1 ^ T2 is the 8-group as for E8P12
  uint64_t* dbg;
};

template <int REP1, int REP2, int SLOTS, int G, bool RVQ3 = false>
__global__ __launch_bounds__(1024) void e8p_gemv_v2_kernel(V2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using L = V2Lds<REP1, REP2, RVQ3>;
  using Slot = std::conditional_t<RVQ3, u32x3, u32x4>;
  constexpr int kUnit = RVQ3 ? 12 : 16;   // bytes of a lane's piece of the code stream (four dwords' worth of codes)
  const int rb = (int)blockIdx.y, ks = (int)blockIdx.x, wg = rb * a.ksplit + ks;      // grid = (K parts, row blocks)
#define V2_STAMP(i) do { if (a.dbg && threadIdx.x == 0) a.dbg[wg * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
  V2_STAMP(0);
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwaves = __builtin_amdgcn_readfirstlane(nthreads >> 6);
  const int n = lane & 15, q = lane >> 4;
  const int r = n & 3, h = n >> 2;
  const int seg0 = ks * a.spw;
  const int S = min(a.segs, seg0 + a.spw) - seg0;     // segments of this workgroup
  const int rpr = (S + a.runlen - 1) / a.runlen;      // runs per row quad
  const int row_u4 = a.K >> 6;                        // uint4 per packed row
  int row0[G], rows_here[G], qbase[G + 1], rbase[G];  // first row, rows, first row quad / accumulator row of a problem
  qbase[0] = 0;
#pragma unroll
  for (int p = 0; p < G; ++p) {
    row0[p] = rb * a.rpb[p];
    rows_here[p] = max(0, min(a.N[p], row0[p] + a.rpb[p]) - row0[p]);
    qbase[p + 1] = qbase[p] + ((rows_here[p] + 3) >> 2);
    rbase[p] = p == 0 ? 0 : rbase[p - 1] + a.rpb[p - 1];
  }
  const int nruns = qbase[G] * rpr;
  const uint32_t xbase = (uint32_t)L::kX;
  const uint32_t accbase = xbase + (uint32_t)(G * S) * kSegBytes;
  int* accs = reinterpret_cast<int*>(smem + accbase);
  const int accwords = (rbase[G - 1] + a.rpb[G - 1]) * 4;
  int* counter = accs + accwords;

  // (0) loads, in the order in which they are needed (VMEM returns in issue order); everything is counted
  int sh[G];
#pragma unroll
  for (int p = 0; p < G; ++p)
    asm volatile("global_load_dword %0, %1, off" : "=v"(sh[p]) : "v"(a.planes[p] + (size_t)3 * a.kp_src) : "memory");
  u32x2 tsrc;
  {
    const int e = (wave & 7) * 32 + (lane & 31);
    const uint2* t1 = reinterpret_cast<const uint2*>(a.grid) + e;
    const uint2* t2 = a.d4 ? t1 : &kV2T2Img.v[e];
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(tsrc) : "v"((lane & 32) ? t2 : t1) : "memory");
  }
  u32x2 tsrc3 = {0u, 0u};   // RVQ3: this lane's E81B entry (requested right behind tsrc: any later wait covers both)
  if constexpr (RVQ3)
    asm volatile("global_load_dwordx2 %0, %1, off"
                 : "=v"(tsrc3)
                 : "v"(reinterpret_cast<const uint2*>(a.grid2) + ((wave & 7) * 32 + (lane & 31)))
                 : "memory");
  // filler for the slots that have nothing to fetch (the load counts are compile-time constants): ONE 16-byte address for the
  // whole wave -- a filler with 64 addresses costs the vector L1 what a real request costs
  const uint4* hot = reinterpret_cast<const uint4*>(a.planes[0]);
  // digit images, requested BEFORE the weights (loads return in issue order; with the weights first -- HBM requests a
  // few hundred instructions earlier -- every shape measured slower: the digit copy then waits for the first HBM
  // burst).  A thread takes k16 index g (16 digits) of ALL planes of ALL problems: one index decode for 3 G requests, the
  // planes' bases are scalars (round 6: a decode per request -- ~40 VALU x 6 -- was 1.7K of the 2.7K clocks between the
  // kernel's start and its first weight request).  Every workgroup starts at a different index so that they do not all
  // queue on the same L2 channels; threads without an index re-read piece 0.
  constexpr int NG = G == 1 ? 2 : 1;        // k16 indices per thread: K parts of up to 32768 k (one problem) / 16384 k
  constexpr int XR = 3 * G * NG;
  const int gper = S * 64;                  // k16 indices in this workgroup's K range
  const int src_pieces = a.kp_src >> 4;     // pieces per plane in the source
  const int rot = (int)(((uint32_t)wg * 5u) & 31u) * (gper >> 5);
  u32x4 xr[XR];
  uint32_t xdst[NG];                        // LDS destination (problem 0, plane 0); 0xffffffff: none; bit 31: store zeros
#pragma unroll
  for (int j = 0; j < NG; ++j) {
    const int i = tid + j * nthreads;
    int g = i + rot;
    g = g >= gper ? g - gper : g;
    g = i < gper ? g : 0;
    const int sgi = g >> 6, c = (g >> 2) & 15, t = g & 3;
    const int sp = seg0 * 64 + g;
    const bool real = sp < src_pieces;      // beyond the source's zero padding: zeros
    const uint32_t voff = (uint32_t)(real ? sp : 0) << 4;
#pragma unroll
    for (int p = 0; p < G; ++p) {
#pragma unroll
      for (int d = 0; d < 3; ++d)
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xr[(j * G + p) * 3 + d]) : "v"(voff), "s"(a.planes[p] + (size_t)d * a.kp_src) : "memory");
    }
    const uint32_t dst = xbase + (uint32_t)sgi * kSegBytes + (uint32_t)((((c & 3) * 4 + t) * 12 + 3 * (c >> 2)) * 16);
    xdst[j] = i < gper ? (dst | (real ? 0u : 0x80000000u)) : 0xffffffffu;
  }


  // run -> (problem, row quad, first segment, length); everything wave uniform
  auto problem_of_quad = [&](int gq) -> int {
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
  // load cursor
  int l_run = wave;                   // current run (>= nruns: none)
  int l_gq = 0, l_seg = 0, l_left = 0;
  // The run's per-lane source pointer and per-run constants are computed when a run is opened; a unit then costs
  // one 64-bit add (the per-unit address arithmetic was a third of the VALU instructions of the stream).
  const char* l_ptr = reinterpret_cast<const char*>(hot);   // this lane's kUnit bytes of the run's next unit
  int l_mgq = 0, l_xoff = 0;          // accumulator row quad / digit image offset of the run's next unit
  const bool ragged = (a.K & 1023) != 0;   // the row's last segment is partial: lanes past the row re-read a valid
                                           // piece (their digits are zero)
  auto open_run = [&](int run) __attribute__((always_inline)) {
    // (branch free on purpose: with the assignments under `if (run < nruns) .. else ..` LLVM sinks the two branches'
    //  stores into one store through a pointer phi, which keeps the cursor variables in scratch memory -- and scratch
    //  accesses are VMEM operations the counted waits do not know about)
    l_run = run;
    const bool ok = run < nruns;
    const int rc = ok ? run : 0;
    // rc / rpr by the host's reciprocal (rc rpr < 2^20: exact); a short last K part has its own rpr -- the plain division there
    l_gq = rpr == a.rpr_inv >> 24 ? (int)(((uint32_t)rc * (uint32_t)(a.rpr_inv & 0xffffff)) >> 20) : __builtin_amdgcn_readfirstlane(rc / rpr);
    const int ri = rc - l_gq * rpr;
    l_seg = ri * a.runlen;
    l_left = ok ? min(a.runlen, S - l_seg) : 0;
    const int p = __builtin_amdgcn_readfirstlane(problem_of_quad(l_gq));
    const int qb = pick(qbase, p);
    int row = pick(row0, p) + 4 * (l_gq - qb) + r;
    const int N = pick(a.N, p);
    row = row < N ? row : N - 1;
    const uint4* W = a.W[0];
#pragma unroll
    for (int i = 1; i < G; ++i) {
      W = p == i ? a.W[i] : W;
      asm volatile("" : "+s"(W));
    }
    l_ptr = reinterpret_cast<const char*>(W) + ((size_t)row * row_u4 + (seg0 + l_seg) * 16 + 4 * h + q) * kUnit;
    l_mgq = (pick(rbase, p) >> 2) + (l_gq - qb);
    l_xoff = __builtin_amdgcn_readfirstlane((p * S + l_seg) * kSegBytes);
  };
  open_run(wave);
  // per-slot description of the unit in flight: accumulator row quad, digit image offset, flags
  int s_gq[SLOTS], s_x[SLOTS], s_flag[SLOTS];   // flag: 0 filler, 1 unit, 3 unit that ends its run
  auto issue = [&](Slot& dst, int& m_gq, int& m_x, int& m_flag) __attribute__((always_inline)) {
    const bool real = l_left > 0;   // wave uniform
    const char* ptr = real ? l_ptr : reinterpret_cast<const char*>(hot);
    if (ragged && real && seg0 + l_seg == a.segs - 1) {   // wave uniform condition
      const int off = (seg0 + l_seg) * 16 + 4 * h + q;
      ptr = off < row_u4 ? ptr : ptr - (4 * h + q) * kUnit;
    }
    asm_load16_nt(dst, reinterpret_cast<const uint4*>(ptr));
    m_gq = l_mgq;
    m_x = l_xoff;
    m_flag = real ? (l_left == 1 ? 3 : 1) : 0;
    if (real) {
      ++l_seg;
      --l_left;
      l_ptr += 16 * kUnit;
      l_xoff += kSegBytes;
    }
  };
  // claims the next run for the load cursor when the current one is used up (between units, once the LDS
  // counter exists)
  auto refill = [&]() __attribute__((always_inline)) {
    if (l_left == 0 && l_run < nruns) {
      int nxt = 0;
      if (lane == 0) nxt = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      open_run(__builtin_amdgcn_readfirstlane(nxt) + nwaves);
    }
  };
  Slot slot[SLOTS];
  // (no refill here: the run counter does not exist yet; a first run shorter than SLOTS leaves filler slots)
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) issue(slot[i], s_gq[i], s_x[i], s_flag[i]);
  V2_STAMP(1);

  // (1) accumulators + run counter, tables
  for (int i = tid; i <= accwords; i += nthreads) accs[i] = 0;
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tsrc) : "n"(XR + SLOTS) : "memory");
  if (wave < 8) {
    const bool second = (lane & 32) != 0;
    const uint32_t t1x = __builtin_amdgcn_perm(0u, tsrc.x, 0x03010200u) | 0x01010101u;
    const uint32_t t1y = __builtin_amdgcn_perm(0u, tsrc.y, 0x03010200u) | 0x01010101u;
    u32x2 val = {second ? tsrc.x : t1x, second ? tsrc.y : t1y};
    if (a.d4) {
      // four fp16 half-integers -> int8 4w (|w| <= 7.5: exact), in the half of the 8-byte entry its code byte stands for
      const f16x2 lo = as_f16x2(tsrc.x), hi = as_f16x2(tsrc.y);
      const uint32_t pk = ((uint32_t)(int)(4.f * (float)lo.x) & 0xffu) | (((uint32_t)(int)(4.f * (float)lo.y) & 0xffu) << 8) |
                          (((uint32_t)(int)(4.f * (float)hi.x) & 0xffu) << 16) | (((uint32_t)(int)(4.f * (float)hi.y) & 0xffu) << 24);
      val = second ? u32x2{pk, 0u} : u32x2{0u, pk};
    }
    const uint32_t row = (uint32_t)(wave * 32 + (lane & 31));
    const uint32_t rowbase = second ? (uint32_t)L::kT2 + row * (REP2 * 8) : row * (REP1 * 8);
    const uint32_t mask = second ? (uint32_t)(REP2 - 1) : (uint32_t)(REP1 - 1);
    constexpr int kMaxRep = REP1 > REP2 ? REP1 : REP2;
#pragma unroll
    for (int c = 0; c < kMaxRep; ++c) {
      if (c < (second ? REP2 : REP1)) {
        const uint32_t copy = (uint32_t)(lane + c) & mask;
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>((uintptr_t)(rowbase + copy * 8)) = val;
      }
    }
  }
  if constexpr (RVQ3) {
    // T3 row 32 w + (l & 31) from this lane's E81B entry; lanes l and l + 32 share a row and write the even / odd
    // halves of its 16 copies
    asm volatile("" : "+v"(tsrc3));   // landed with tsrc (requested right behind it, before anything waited for)
    if (wave < 8) {
      const uint32_t rowbase = (uint32_t)L::kT3 + (uint32_t)(wave * 32 + (lane & 31)) * (L::kRep3 * 8);
#pragma unroll
      for (int c = 0; c < L::kRep3 / 2; ++c) {
        const uint32_t copy = ((uint32_t)(lane + c) & (uint32_t)(L::kRep3 / 2 - 1)) * 2 + (uint32_t)(lane >> 5);
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>((uintptr_t)(rowbase + copy * 8)) = tsrc3;
      }
    }
  }
#pragma unroll
  for (int p = 0; p < G; ++p) asm volatile("" : "+v"(sh[p]));   // landed before tsrc
  V2_STAMP(2);

  // (2) digit images into LDS in fragment order: unit ((q * 4 + t) * 12 + 3 h + d) of segment s holds plane d,
  //     k = 1024 s + 64 (4 h + q) + 16 t .. +15
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
          const u32x4 v = z ? u32x4{0u, 0u, 0u, 0u} : xr[(j * G + p) * 3 + d];
          *reinterpret_cast<__attribute__((address_space(3))) u32x4*>((uintptr_t)(at0 + (uint32_t)(p * S) * kSegBytes + 16u * d)) = v;
        }
      }
    }
  }
  __syncthreads();
  V2_STAMP(3);

  uint32_t lane_c1, lane_c2;
  if constexpr (REP1 == 32) lane_c1 = ((uint32_t)(lane & 31) << 3) | ((uint32_t)(L::kT2 >> 16) << 16);
  else lane_c1 = (uint32_t)(lane & 15) << 3;
  lane_c2 = ((uint32_t)(lane & 15) << 3) | (uint32_t)L::kT2;
  const uint32_t lane_c3 = ((uint32_t)(lane & 15) << 3) | (uint32_t)L::kT3;
  // A fragment of this lane (A row m = lane & 15 = 4 h' + d', k block q): unit (q * 4 + t) * 12 + 3 h' + min(d', 2)
  const uint32_t xlane = xbase + (uint32_t)((q * 48 + 3 * (n >> 2) + min(n & 3, 2)) * 16);
  const bool dvalid = q == h;      // this lane's D registers 0..2 = S_h, S_m, S_l of row r over chunk group h

  // (3) the stream
  i32x4 acc = {0, 0, 0, 0}, prev = {0, 0, 0, 0};
  bool more = true;
  while (more) {
    more = false;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(slot[i]) : "n"(SLOTS - 1) : "memory");
      uint32_t a1l[4], a2l[4], a1h[4], a2h[4];
      const u32x4 dq = v2_rvq3_dwords(slot[i]);
      const uint32_t dw[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (REP1 == 32) {
          // T1 (32 copies): idx << 8 | (lane & 31) << 3: byte aligned, one v_perm_b32
          a1l[t] = __builtin_amdgcn_perm(dw[t], lane_c1, 0x0c0c0500u);
          a1h[t] = __builtin_amdgcn_perm(dw[t], lane_c1, 0x0c0c0700u);
        } else {
          a1l[t] = ((dw[t] >> 1) & 0x7f80u) | lane_c1;
          a1h[t] = ((dw[t] >> 17) & 0x7f80u) | lane_c1;
        }
        if constexpr (RVQ3) {
          // the low code of the dword is (residual index << 8 | 0) and reads T3 (E81B); its sign byte is 0 and
          // T2[0] == 0, so the common "T1 ^ T2" below leaves the T3 entry unchanged
          a1l[t] = ((dw[t] >> 1) & 0x7f80u) | lane_c3;
        }
        if constexpr (REP1 == 32 && REP2 == 32) {
          // T2 base 0x10000 comes from byte 2 of lane_c1
          a2l[t] = __builtin_amdgcn_perm(dw[t], lane_c1, 0x0c020400u);
          a2h[t] = __builtin_amdgcn_perm(dw[t], lane_c1, 0x0c020600u);
        } else {
          a2l[t] = ((dw[t] << 7) & 0x7f80u) | lane_c2;
          a2h[t] = ((dw[t] >> 9) & 0x7f80u) | lane_c2;
        }
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) asm volatile("" : "+v"(a1l[t]), "+v"(a2l[t]), "+v"(a1h[t]), "+v"(a2h[t]));
      const int gq = s_gq[i], xo = s_x[i], flag = s_flag[i];
      refill();
      issue(slot[i], s_gq[i], s_x[i], s_flag[i]);
      more = more || s_flag[i] != 0;
      if (flag) {   // wave uniform
        const uint32_t xa = xlane + (uint32_t)xo;
        uint2 t1l[4], t2l[4], t1h[4], t2h[4];
        i32x4 A[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          t1l[t] = v2_lds_read8(a1l[t]); t2l[t] = v2_lds_read8(a2l[t]);
          t1h[t] = v2_lds_read8(a1h[t]); t2h[t] = v2_lds_read8(a2h[t]);
          A[t] = v2_lds_read16i(xa + (uint32_t)t * 192u);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const i32x4 B = {(int)(t1l[t].x ^ t2l[t].x), (int)(t1l[t].y ^ t2l[t].y),
                           (int)(t1h[t].x ^ t2h[t].x), (int)(t1h[t].y ^ t2h[t].y)};
          acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], B, acc, 0, 0, 0);
        }
        if (flag & 2) {   // run finished: hand its sums to the LDS accumulators
          // The MFMA accumulator is never reset (a reset in this rare branch made the compiler copy / select the
          // MFMA result in the common path, i.e. drain the matrix pipeline after every unit): a run's sums are
          // the difference to the accumulator at the previous flush (int32 wrap-around arithmetic is exact here).
          const int dx = acc.x - prev.x, dy = acc.y - prev.y, dz = acc.z - prev.z;
          // prev += d, i.e. prev = acc.  Through asm: a plain assignment is if-converted into selects on the MFMA
          // result outside this branch.  The asm reads only VALU results (an asm statement that read the MFMA
          // result itself would need hand-placed wait states).
          asm volatile("v_add_u32 %0, %0, %3\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %5"
                       : "+v"(prev.x), "+v"(prev.y), "+v"(prev.z)
                       : "v"(dx), "v"(dy), "v"(dz));
          if (dvalid) {
            int* dst = accs + (gq * 4 + r) * 4;
            __hip_atomic_fetch_add(dst + 0, dx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 1, dy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(dst + 2, dz, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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

  // (4) y = 2^(-sh-2) (65536 S_h + 256 S_m + S_l), one fp16 rounding
  if (a.ksplit == 1) {
#pragma unroll
    for (int p = 0; p < G; ++p) {
      const float unscale = unscale_of(sh[p], 2);
      for (int t = tid; t < rows_here[p]; t += nthreads) {
        const int* s3 = accs + (rbase[p] + t) * 4;
        const float f = __builtin_fmaf((float)s3[0], 65536.f, __builtin_fmaf((float)s3[1], 256.f, (float)s3[2]));
        a.y[p][row0[p] + t] = (f16)(f * unscale);
      }
    }
  } else {
    // partial sums of this K range -> workspace (agent-scope integer atomics: exact, order independent)
#pragma unroll
    for (int p = 0; p < G; ++p) {
      for (int t = tid; t < rows_here[p]; t += nthreads) {
        const int* s3 = accs + (rbase[p] + t) * 4;
        int* g = a.ws[p] + (size_t)(row0[p] + t) * 4;
        __hip_atomic_fetch_add(g + 0, s3[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(g + 1, s3[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(g + 2, s3[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
        const float unscale = unscale_of(sh[p], 2);
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

template <int REP1, int REP2, int SLOTS, int G, bool RVQ3 = false>
int v2_launch(const V2Args& a, int nrb, int threads, int lds, hipStream_t stream) {
  auto kern = e8p_gemv_v2_kernel<REP1, REP2, SLOTS, G, RVQ3>;
  static DynLdsCache configured;   // per instantiation, per device
  if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.ksplit, nrb), dim3(threads), lds, stream, a);
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

// rep code: 32 = (32, 32) copies, 24 = (32, 16), 16 = (16, 16); 40 = E8P12RVQ3B: (32, 16) + T3 x 16
static int lds_x(int rep) {
  if (rep == 40) return V2Lds<32, 16, true>::kX;
  return rep == 32 ? V2Lds<32, 32>::kX : (rep == 24 ? V2Lds<32, 16>::kX : V2Lds<16, 16>::kX);
}

template <int G>
int v2_group_launch(const void* const* planes, const void* const* qidxs, const void* grid, void* const* ys,
                    void* ws, const int* ns, int k, const GemvTune& tune, hipStream_t stream) {
  const int ncu = device_cu_count();
  const int segs = (k + 1023) >> 10;
  const int slots = tune.rows > 0 ? tune.rows : 2;
  // Candidates (table replication, K split).  K is split only when the digit images of the whole rows do not fit
  // even beside the smallest tables, because combining partial sums across workgroups costs two more memory round
  // trips at the end of the launch; among forced splits (full tables) the one with the fewest units per
  // workgroup wins.
  int rep = 0, ksplit = 0, nrb = 0, spw = 0, rpb[kMaxG] = {0, 0, 0}, best = 0;
  auto consider = [&](int rp, int ks) -> bool {
    int nrb_c = (tune.blocks > 0 ? tune.blocks : ncu) / ks;
    if (nrb_c < 1) nrb_c = 1;
    int rp_c[kMaxG] = {0, 0, 0}, rows = 0, quads = 0, need = 1;
    for (;;) {   // accumulator rows must fit: more row blocks until they do
      rows = 0; quads = 0; need = 1;
      for (int p = 0; p < G; ++p) {
        int v = (ns[p] + nrb_c - 1) / nrb_c;
        v = (v + 3) & ~3;
        rp_c[p] = v;
        rows += v;
        quads += v >> 2;
        const int nb = (ns[p] + v - 1) / v;
        need = nb > need ? nb : need;
      }
      if (rows <= 1024) break;
      nrb_c *= 2;
    }
    const int spw_c = (segs + ks - 1) / ks;
    const int room = (V2Lds<16, 16>::kTotal - lds_x(rp) - rows * 16 - 16) / kSegBytes;
    if (G * spw_c > room || spw_c * 64 > (G == 1 ? 2 : 1) * 1024) return false;      // (1 or 2 k16 indices per thread)
    const int cost = quads * spw_c;
    if (!ksplit || cost < best) {
      ksplit = ks; nrb = need; spw = spw_c; best = cost; rep = rp;
      for (int p = 0; p < G; ++p) rpb[p] = rp_c[p];
    }
    return true;
  };
  if (tune.rep == 40) {            // E8P12RVQ3B: one table configuration
    if (G != 1 || !tune.grid2) return QUIP_ERR_UNSUPPORTED;
    for (int ks = 1; ks <= segs; ++ks)
      if (consider(40, ks)) break;
  } else if ((tune.rep && tune.rep != 64) || tune.waves_g) {
    const int rp = tune.rep && tune.rep != 64 ? tune.rep : 32;
    for (int ks = tune.waves_g > 0 ? tune.waves_g : 1; ks <= segs; ++ks)
      if (consider(rp, ks)) break;
  } else {
    const int reps[3] = {32, 24, 16};
    for (int ri = 0; ri < 3 && !ksplit; ++ri) consider(reps[ri], 1);
    if (!ksplit)
      for (int ks = 2, tried = 0; ks <= segs && tried < 4; ++ks) tried += consider(32, ks) ? 1 : 0;
  }
  if (!ksplit) return QUIP_ERR_UNSUPPORTED;
  ksplit = (segs + spw - 1) / spw;
  if (ksplit > 1 && !ws) return QUIP_ERR_NULL_POINTER;
  V2Args a;
  size_t ws_off = 0;
  int quads = 0, rows = 0;
  for (int p = 0; p < kMaxG; ++p) {
    const int pp = p < G ? p : 0;
    a.W[p] = reinterpret_cast<const uint4*>(qidxs[pp]);
    a.planes[p] = reinterpret_cast<const uint8_t*>(planes[pp]);
    a.y[p] = reinterpret_cast<f16*>(ys[pp]);
    a.N[p] = ns[pp];
    a.rpb[p] = rpb[pp];
    a.ws[p] = ws ? reinterpret_cast<int*>(ws) + ws_off : nullptr;
    if (p < G) {
      ws_off += (size_t)ns[p] * 4;                 // accumulators back to back; the counters follow the last one
      quads += rpb[p] >> 2;
      rows += rpb[p];
    }
  }
  // row blocks <= max_p ceil(n_p / 4) <= the counter words e8p_gemv_v2_workspace_words() reserves in total
  a.cnt = ws ? reinterpret_cast<int*>(ws) + ws_off : nullptr;
  a.grid = reinterpret_cast<const uint64_t*>(grid);
  a.grid2 = reinterpret_cast<const uint64_t*>(tune.grid2);
  a.K = k;
  a.kp_src = (k + 511) & ~511;
  a.segs = segs; a.spw = spw; a.ksplit = ksplit;
  a.dbg = reinterpret_cast<uint64_t*>(tune.dbg);
  a.d4 = tune.rep == 64 ? 1 : 0;      // (the first kernel's mode number for its D4 table)
  // 16 waves for long streams; 12 when a workgroup has few units (8192^2: 64 units, 7.2 vs 7.9 us with 16)
  int waves = tune.max_waves > 0 ? tune.max_waves : (quads * spw >= 128 ? 16 : 12);
  if (waves < 8) waves = 8;     // the table build uses waves 0..7
  if (waves > 16) waves = 16;
  while (waves < 16 && spw * 64 > (G == 1 ? 2 : 1) * waves * 64) ++waves;   // k16 indices per thread
  // run length: the longest (fewest LDS flushes, longest contiguous reads) that still leaves about three runs per wave (measured)
  int runlen = tune.digits > 0 ? tune.digits : spw;
  if (tune.digits <= 0)
    while (runlen > slots && quads * ((spw + runlen - 1) / runlen) < 3 * waves) runlen = (runlen + 1) / 2;
  if (runlen > spw) runlen = spw;
  if (runlen < 1) runlen = 1;
  a.runlen = runlen;
  {
    const int rpr = (spw + runlen - 1) / runlen;
    a.rpr_inv = (rpr << 24) | (((1 << 20) / rpr + 1) & 0xffffff);
  }
  const int threads = waves * 64;
  const int lds = lds_x(rep) + G * spw * kSegBytes + rows * 16 + 16;
  if constexpr (G == 1)
    if (rep == 40) return v2_launch<32, 16, 2, 1, true>(a, nrb, threads, lds, stream);
#define QUIP_V2(R1, R2, RR, S) \
  if (rep == RR && slots == S) return v2_launch<R1, R2, S, G>(a, nrb, threads, lds, stream);
  QUIP_V2(32, 32, 32, 1) QUIP_V2(32, 32, 32, 2) QUIP_V2(32, 32, 32, 3) QUIP_V2(32, 32, 32, 4)
  QUIP_V2(32, 16, 24, 1) QUIP_V2(32, 16, 24, 2) QUIP_V2(32, 16, 24, 3) QUIP_V2(32, 16, 24, 4)
  QUIP_V2(16, 16, 16, 1) QUIP_V2(16, 16, 16, 2) QUIP_V2(16, 16, 16, 3) QUIP_V2(16, 16, 16, 4)
#undef QUIP_V2
  return QUIP_ERR_UNSUPPORTED;
}

}  // namespace

bool e8p_gemv_v2_supported(int n, int k) { return n >= 1 && k >= 128 && k % 128 == 0; }

// int32 words of zeroed workspace a problem may need (accumulators + arrival counters)
size_t e8p_gemv_v2_workspace_words(int n) { return (size_t)n * 4 + (size_t)((n + 3) / 4) + 64; }

int e8p_gemv_v2_group_launch(const void* const* planes, const void* const* qidxs, const void* grid, void* const* ys,
                             void* ws, const int* ns, int count, int k, const GemvTune& tune, hipStream_t stream) {
  if (count < 1 || count > kMaxG) return QUIP_ERR_UNSUPPORTED;
  // nibble mode (e8p_gemv_v2n.hip): asked for, or -- automatic choice -- for rows whose digit image leaves the byte tables 16
  // copies only (k > 10240: two-way conflicts on every look-up; the nibble tables are 64 KB and conflict free whatever k is:
  // 8192 x 28672 -5..9 % on two boxes, level on a third; at k = 8192 both are conflict free and the two kernels trade places from
  // box to box: profiles/r06_gemv_v2_nibble.txt), and for grouped launches (q / k / v of 8192: 11.5-11.9 against 13.3 us and the
  // first kernel's 12.4; gate / up 24.6 against 27.5-28.2; at k = 4096 the byte tables win: 8.7 against 9.4 us for 2 x 11008);
  // QUIP_GEMV_NIB=0 keeps the byte tables
  {
    static int nib_mode = -1;
    if (nib_mode < 0) {
      const char* e = getenv("QUIP_GEMV_NIB");
      nib_mode = e ? atoi(e) : 1;
    }
    if (tune.rep == 4 || (tune.rep == 0 && !tune.waves_g && nib_mode != 0 && (k > 10240 || (count >= 2 && k >= 8192)))) {
      GemvTune t = tune;
      t.rep = 4;
      const int rc = e8p_gemv_v2n_group_launch(planes, qidxs, grid, ys, ws, ns, count, k, t, stream);
      if (tune.rep == 4 || (rc != QUIP_ERR_UNSUPPORTED && rc != QUIP_ERR_NULL_POINTER)) return rc;   // (else: the byte tables' candidates)
    }
  }
  for (int i = 0; i < count; ++i)
    if (!e8p_gemv_v2_supported(ns[i], k)) return QUIP_ERR_UNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(grid) & 63u) != 0) return QUIP_ERR_MISALIGNED;
  if (count == 1) return v2_group_launch<1>(planes, qidxs, grid, ys, ws, ns, k, tune, stream);
  if (count == 2) return v2_group_launch<2>(planes, qidxs, grid, ys, ws, ns, k, tune, stream);
  return v2_group_launch<3>(planes, qidxs, grid, ys, ws, ns, k, tune, stream);
}

int e8p_gemv_v2_launch(const void* planes, const void* qidxs, const void* grid, void* y, void* ws, int n, int k,
                       const GemvTune& tune, hipStream_t stream) {
  return e8p_gemv_v2_group_launch(&planes, &qidxs, grid, &y, ws, &n, 1, k, tune, stream);
}

}  // namespace quip
