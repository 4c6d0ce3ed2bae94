// Decode core shared by the matrix-core GEMV (e8p_gemv_mfma.hip) and the persistent decode engine
// (decode_engine.hip): LDS table layout, table build, code -> LDS address arithmetic and the eight
// v_mfma_i32_16x16x64_i8 steps of one item (16 weight rows x 512 k).  See e8p_gemv_mfma.hip for the
// arithmetic and the mapping; everything here is internal linkage (each translation unit gets its own copy).
#pragma once
#include "had_device.hip.h"
#include "quip_internal.h"
#include <type_traits>

#ifndef QUIP_GEMV_R8
#define QUIP_GEMV_R8 0
#endif
#ifndef QUIP_GEMV_PIPE
#define QUIP_GEMV_PIPE 4
#endif

namespace quip {
namespace {

constexpr bool kR8 = QUIP_GEMV_R8 != 0;

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) u32x2* lds_u2_ptr;
typedef const __attribute__((address_space(3))) i32x4* lds_i4_ptr;

constexpr int kMaxRowsPerBlock = 256;  // int32 accumulator rows per workgroup

// LDS map.  REP = 32 copies per table entry (ds_read_b64 conflict free) when the x image
// is small enough (K <= 8192), REP = 16 (two-way conflicts on average, half the
// footprint) for longer rows so that the digit planes of the WHOLE row stay resident.
template <int REP>
struct Lds {
  // REP = 32 / 16: both tables with that many copies; REP = 24: T1 x 32 (conflict free), T2 x 16
  // REP = 64: the D4 codebook -- ONE table of 256 x 4-byte entries (2w of the code's 4 weights as int8),
  //           64 copies (a private copy per lane: no conflicts), no sign table
  // REP = 40 / 20: E8P12RVQ3B -- the E8P tables (32 / 16 copies of T1, 16 of T2) plus T3 = the 256 x 8-byte
  //           E81B residual table (4r as int8; 16 / 8 copies), looked up by the LOW code of every dword; the
  //           weight stream is the checkpoint's own 3-byte codes (12-byte loads, 3 k / 8 bytes per row)
  static constexpr bool kD4 = REP == 64;
  // REP = 4: NIBBLE MODE (see the end of this file) -- both E8P tables as 4-byte entries, 32 copies each, interleaved in
  //           256-byte rows: 64 KB, every look-up a conflict-free ds_read_b32
  static constexpr bool kNib = REP == 4;
  // REP = 12: E8P12RVQ3B inside the persistent block launch -- 16 / 16 copies and FOUR of T3 (72 KB of tables: the virtual
  //           rows' digit planes need the rest)
  static constexpr bool kRvq3 = REP == 40 || REP == 20 || REP == 12;
  static constexpr int kRep1 = kD4 ? 64 : ((REP == 16 || REP == 20 || REP == 12) ? 16 : 32);
  static constexpr int kRep2 = REP == 32 ? 32 : 16;
  static constexpr int kRep3 = REP == 40 ? 16 : (REP == 20 ? 8 : (REP == 12 ? 4 : 0));
  static constexpr int kRow1 = kRep1 * (kD4 ? 4 : 8);   // bytes per T1 entry row
  static constexpr int kRow2 = kD4 ? 0 : kRep2 * 8;     // bytes per T2 entry row
  static constexpr int kRow3 = kRep3 * 8;               // bytes per T3 entry row
  static constexpr int kT1 = 0;
  static constexpr int kT2 = kNib ? 128 : 256 * kRow1;
  static constexpr int kT3 = kNib ? 256 * 256 : kT2 + 256 * kRow2;
  static constexpr int kAcc = kT3 + 256 * kRow3;     // int32 [kMaxRowsPerBlock][4]
  static constexpr int kX = kAcc + kMaxRowsPerBlock * 16;   // 3 planes x Kp bytes
  static constexpr int kMaxKp = (160 * 1024 - kX) / 3 / 512 * 512;
  static int bytes(int kp, int g = 1) { return kX + 3 * kp * g; }
  // fused prologue: + the Hadamard shuffle buffer (fp32, padded) and 16 reduction words
  static int bytes_fused(int kp, int g) { return kX + 3 * kp * g + (had::buf_floats(kp) + 16) * 4; }
};
static_assert(Lds<32>::kMaxKp >= 8192 && Lds<16>::kMaxKp >= 28672, "LDS budget");
static_assert(Lds<40>::kMaxKp >= 8192 && Lds<20>::kMaxKp >= 22528, "LDS budget (RVQ3: 2 x 4096, 2 x 11008)");

__device__ __forceinline__ uint2 lds_read8(uint32_t addr) {
  const u32x2 v = *reinterpret_cast<lds_u2_ptr>((uintptr_t)addr);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ i32x4 lds_read16i(uint32_t addr) {
  return *reinterpret_cast<lds_i4_ptr>((uintptr_t)addr);
}

// compile-time image of the sign table (see e8p_gemv_i8.hip)
struct T2Image {
  uint2 v[256];
  constexpr T2Image() : v{} {
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
__device__ const T2Image kT2Img{};

// T1 entry from grid_packed_abs[e]: natural position order (bytes 0,2,1,3 / 4,6,5,7), OR 1
__device__ __forceinline__ uint2 t1_entry(uint2 packed) {
  return make_uint2(__builtin_amdgcn_perm(0u, packed.x, 0x03010200u) | 0x01010101u,
                    __builtin_amdgcn_perm(0u, packed.y, 0x03010200u) | 0x01010101u);
}

// LDS tables: wave w < 8 owns table rows [32 w, 32 w + 32); lane l holds
// the 8-byte source of row 32 w + (l & 31) -- T1 source (grid_packed_abs) in lanes 0..31, T2 image in
// lanes 32..63 -- fetched with ONE vector load issued as the very first load of the kernel, and
// writes it REP times into its own row, copy (l + c) mod REP at step c: the lanes of a half-wave
// hit REP distinct bank pairs per step (no broadcast through SGPRs, no scalar-load latency chain).
__device__ __forceinline__ const uint2* table_source_ptr(const uint64_t* grid, int lane, int wave) {
  const int e = (wave & 7) * 32 + (lane & 31);
  const uint2* t1 = reinterpret_cast<const uint2*>(grid) + e;
  const uint2* t2 = &kT2Img.v[e];
  return (lane & 32) ? t2 : t1;
}
// D4: `grid` is the fp16 (256, 4) table (d4.py:26-96); every lane reads entry 32 w + (l & 31) (8 bytes)
__device__ __forceinline__ const uint2* table_source_ptr_d4(const uint64_t* grid, int lane, int wave) {
  return reinterpret_cast<const uint2*>(grid) + ((wave & 7) * 32 + (lane & 31));
}
// RVQ3: T3 row 32 w + (l & 31) from this lane's 8-byte E81B entry; lanes l and l + 32 share a row and
// write the two halves of its copies
template <int REP>
__device__ __forceinline__ void fill_t3_from_lane(char* smem, const u32x2& src, int lane, int wave) {
  using L = Lds<REP>;
  if constexpr (L::kRvq3) {
    const uint32_t rowbase = (uint32_t)L::kT3 + (uint32_t)(wave * 32 + (lane & 31)) * L::kRow3;
    constexpr int half = L::kRep3 / 2;
#pragma unroll
    for (int c = 0; c < half; ++c) {
      const uint32_t copy = (((uint32_t)(lane + c)) & (uint32_t)(half - 1)) + ((lane & 32) ? half : 0);
      *reinterpret_cast<__attribute__((address_space(3))) u32x2*>((uintptr_t)(rowbase + copy * 8)) = src;
    }
  }
}

template <int REP>
__device__ __forceinline__ void fill_tables_from_lane(char* smem, const u32x2& src, int lane, int wave) {
  using L = Lds<REP>;
  if constexpr (L::kD4) {
    // 4 fp16 half-integers -> int8 2w; lanes l and l + 32 hold the same entry and write copies
    // [0, 32) resp. [32, 64) of its row, rotating so that a step touches 32 distinct banks
    const f16x2 lo = as_f16x2(src.x), hi = as_f16x2(src.y);
    const int b0 = (int)(2.f * (float)lo.x), b1 = (int)(2.f * (float)lo.y);
    const int b2 = (int)(2.f * (float)hi.x), b3 = (int)(2.f * (float)hi.y);
    const uint32_t val = (uint32_t)(b0 & 0xff) | ((uint32_t)(b1 & 0xff) << 8) | ((uint32_t)(b2 & 0xff) << 16) |
                         ((uint32_t)(b3 & 0xff) << 24);
    const uint32_t rowbase = (uint32_t)L::kT1 + (uint32_t)(wave * 32 + (lane & 31)) * L::kRow1 + (uint32_t)(lane & 32) * 4;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const uint32_t copy = (uint32_t)(lane + c) & 31u;
      *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)(rowbase + copy * 4)) = val;
    }
    return;
  }
  const bool second = (lane & 32) != 0;
  const uint2 raw = make_uint2(src.x, src.y);
  const uint2 t1 = t1_entry(raw);
  const u32x2 val = {second ? raw.x : t1.x, second ? raw.y : t1.y};
  const uint32_t row = (uint32_t)(wave * 32 + (lane & 31));
  const uint32_t rowbase = second ? (uint32_t)L::kT2 + row * L::kRow2 : (uint32_t)L::kT1 + row * L::kRow1;
  const uint32_t mask = second ? (L::kRep2 - 1) : (L::kRep1 - 1);
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    if (c < L::kRep1 || c < L::kRep2) {
      const uint32_t copy = (uint32_t)(lane + c) & mask;
      if (c < (second ? L::kRep2 : L::kRep1))
        *reinterpret_cast<__attribute__((address_space(3))) u32x2*>((uintptr_t)(rowbase + copy * 8)) = val;
    }
  }
}

// 16 codes of this lane -> eight MFMAs, in two steps so that the caller can reload the
// slot registers between them: item_addresses() consumes the codes completely (32 LDS
// addresses), item_mfma() runs the table / x reads PIPE steps ahead of their MFMA.
struct ItemAddr { uint32_t a1l[8], a2l[8], a1h[8], a2h[8]; };

// E8P12RVQ3B: a checkpoint code is 3 bytes [resid8, e8p_lo, e8p_hi] (e8p12_rvq3.py:81-107); the decode below works
// on dwords (main16 << 16 | resid8 << 8), i.e. the same three bytes behind a zero byte.  A lane's 12 landed bytes =
// four codes: one shift, two v_perm_b32 and one mask.
__device__ __forceinline__ u32x4 rvq3_dwords(const u32x3& w) {
  return u32x4{w.x << 8, __builtin_amdgcn_perm(w.y, w.x, 0x0504030cu), __builtin_amdgcn_perm(w.z, w.y, 0x0403020cu),
               w.z & 0xffffff00u};
}

template <int REP>
__device__ __forceinline__ void item_addresses(const u32x4& q0, const u32x4& q1, uint32_t lane_c,
                                               uint32_t lane_c2, ItemAddr& ad, uint32_t lane_c3 = 0);
template <int REP>
__device__ __forceinline__ void item_addresses(const u32x3& q0, const u32x3& q1, uint32_t lane_c,
                                               uint32_t lane_c2, ItemAddr& ad, uint32_t lane_c3 = 0) {
  item_addresses<REP>(rvq3_dwords(q0), rvq3_dwords(q1), lane_c, lane_c2, ad, lane_c3);
}
template <int REP>
__device__ __forceinline__ void item_addresses(const u32x4& q0, const u32x4& q1, uint32_t lane_c,
                                               uint32_t lane_c2, ItemAddr& ad, uint32_t lane_c3) {
  const uint32_t d[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
  if constexpr (REP == 4) {
    // nibble mode (the end of this file): lane_c = nib_lane_const(lane)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      ad.a1l[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0500u);
      ad.a2l[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0402u);
      ad.a1h[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0700u);
      ad.a2h[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0602u);
    }
    return;
  }
  if constexpr (REP == 64) {
    // D4: dword t = 4 one-byte codes = the 16 weights of MFMA step t; entry address =
    // code << 8 | lane << 2 (lane_c), one v_perm_b32 per code
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      ad.a1l[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0400u);
      ad.a2l[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0500u);
      ad.a1h[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0600u);
      ad.a2h[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0700u);
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if constexpr (Lds<REP>::kRep1 == 32) {
      // T1 (32 copies): table_base | idx << 8 | (lane & 31) << 3: byte aligned, one v_perm_b32
      ad.a1l[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0500u);
      ad.a1h[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0700u);
    } else {
      // 16 copies: table_base | idx << 7 | (lane & 15) << 3: shift + v_and_or_b32
      ad.a1l[t] = ((d[t] >> 1) & 0x7f80u) | lane_c;
      ad.a1h[t] = ((d[t] >> 17) & 0x7f80u) | lane_c;
    }
    if constexpr (Lds<REP>::kRvq3) {
      // RVQ3: the low code of the dword is (residual index << 8 | 0) and reads T3 (E81B); its sign byte is 0
      // and T2[0] == 0, so the generic "T1 ^ T2" below leaves the T3 entry unchanged
      ad.a1l[t] = Lds<REP>::kRep3 == 16  ? (((d[t] >> 1) & 0x7f80u) | lane_c3)
                  : Lds<REP>::kRep3 == 8 ? (((d[t] >> 2) & 0x3fc0u) | lane_c3)
                                         : (((d[t] >> 3) & 0x1fe0u) | lane_c3);
    }
    if constexpr (REP == 32) {
      // T2 base 0x10000 comes from byte 2 of lane_c
      ad.a2l[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c020400u);
      ad.a2h[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c020600u);
    } else {
      ad.a2l[t] = ((d[t] << 7) & 0x7f80u) | lane_c2;
      ad.a2h[t] = ((d[t] >> 9) & 0x7f80u) | lane_c2;
    }
  }
}

// kR8: the two landed loads hold, in lane (r = l >> 3, c = l & 7), chunk c (codes 8c .. 8c+7, k = 64c ..)
// of rows r (qa) and 8 + r (qb).  MFMA step t of lane (n, q) takes dword j = t & 3 of chunk
// c = 2q + (t >> 2) of row n, i.e. k = 128q + 64(t >> 2) + 16(t & 3) of the slice (the A fragments are
// read with the same mapping): two ds_bpermute per step (rows < 8 / >= 8) and a select.
__device__ __forceinline__ void redistribute_r8(const u32x4& qa, const u32x4& qb, int lane, u32x4& da, u32x4& db) {
  const int n = lane & 15, q = lane >> 4;
  const int src0 = (((n & 7) << 3) + 2 * q) << 2, src1 = src0 + 4;
  const bool lo = n < 8;
  const uint32_t a[4] = {qa.x, qa.y, qa.z, qa.w}, b[4] = {qb.x, qb.y, qb.z, qb.w};
  uint32_t d[8];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int a0 = __builtin_amdgcn_ds_bpermute(src0, (int)a[j]), b0 = __builtin_amdgcn_ds_bpermute(src0, (int)b[j]);
    const int a1 = __builtin_amdgcn_ds_bpermute(src1, (int)a[j]), b1 = __builtin_amdgcn_ds_bpermute(src1, (int)b[j]);
    d[j] = (uint32_t)(lo ? a0 : b0);
    d[4 + j] = (uint32_t)(lo ? a1 : b1);
  }
  da = u32x4{d[0], d[1], d[2], d[3]};
  db = u32x4{d[4], d[5], d[6], d[7]};
}

struct StepOperands { uint2 t1l, t2l, t1h, t2h; i32x4 A; };

__device__ __forceinline__ uint32_t lds_read4(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) uint32_t*>((uintptr_t)addr);
}

// D4: the B fragment of a step is four 4-byte table entries, no sign fix-up
template <int HALF = 256>
__device__ __forceinline__ i32x4 item_mfma_d4(const ItemAddr& ad, uint32_t xaddr) {
  constexpr int PIPE = QUIP_GEMV_PIPE;
  i32x4 B[8], A[8];
  auto issue = [&](int t) {
    B[t] = i32x4{(int)lds_read4(ad.a1l[t]), (int)lds_read4(ad.a2l[t]), (int)lds_read4(ad.a1h[t]),
                 (int)lds_read4(ad.a2h[t])};
    A[t] = lds_read16i(xaddr + (t < 4 ? 16 * t : HALF + 16 * (t - 4)));
  };
#pragma unroll
  for (int t = 0; t < PIPE; ++t) issue(t);
  i32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t + PIPE < 8) issue(t + PIPE);
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], B[t], acc, 0, 0, 0);
  }
  return acc;
}

// HALF: bytes between the digits of k and k + 256 of a plane (256 in the plain [3][Kp] image; the engine pads
// every 256 digits by 16 bytes)
// R3 (E8P12RVQ3B, callers that opt in): the low code of a dword is a residual index whose sign byte is 0 and T2[0] == 0 --
// its sign look-up is skipped (a quarter of the item's LDS reads), same integers
template <int HALF = 256, bool R3 = false>
__device__ __forceinline__ i32x4 item_mfma(const ItemAddr& ad, uint32_t xaddr) {
  constexpr int PIPE = QUIP_GEMV_PIPE;
  StepOperands op[8];
  auto issue = [&](int t) {
    op[t].t1l = lds_read8(ad.a1l[t]); op[t].t2l = R3 ? make_uint2(0u, 0u) : lds_read8(ad.a2l[t]);
    op[t].t1h = lds_read8(ad.a1h[t]); op[t].t2h = lds_read8(ad.a2h[t]);
    op[t].A = lds_read16i(xaddr + (kR8 ? 64 * (t >> 2) + 16 * (t & 3) : (t < 4 ? 16 * t : HALF + 16 * (t - 4))));
  };
#pragma unroll
  for (int t = 0; t < PIPE; ++t) issue(t);
  i32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t + PIPE < 8) issue(t + PIPE);
    const i32x4 B = {(int)(op[t].t1l.x ^ op[t].t2l.x), (int)(op[t].t1l.y ^ op[t].t2l.y),
                     (int)(op[t].t1h.x ^ op[t].t2h.x), (int)(op[t].t1h.y ^ op[t].t2h.y)};
    acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(op[t].A, B, acc, 0, 0, 0);
  }
  return acc;
}

// The two halves of item_mfma() as separate calls, for callers that have the codes long before the digit planes (the
// persistent decode engine decodes while it waits for a hand-off): item_decode() turns the 32 addresses into the eight B
// fragments (all the table lookups), item_multiply() reads the eight A fragments and runs the MFMAs.  Same operations, same
// integer sums as item_mfma().
template <bool R3 = false>
__device__ __forceinline__ void item_decode(const ItemAddr& ad, i32x4 (&B)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const uint2 t1l = lds_read8(ad.a1l[t]), t2l = R3 ? make_uint2(0u, 0u) : lds_read8(ad.a2l[t]);
    const uint2 t1h = lds_read8(ad.a1h[t]), t2h = lds_read8(ad.a2h[t]);
    B[t] = i32x4{(int)(t1l.x ^ t2l.x), (int)(t1l.y ^ t2l.y), (int)(t1h.x ^ t2h.x), (int)(t1h.y ^ t2h.y)};
  }
}
__device__ __forceinline__ void item_decode_d4(const ItemAddr& ad, i32x4 (&B)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t)
    B[t] = i32x4{(int)lds_read4(ad.a1l[t]), (int)lds_read4(ad.a2l[t]), (int)lds_read4(ad.a1h[t]), (int)lds_read4(ad.a2h[t])};
}
template <int HALF = 256>
__device__ __forceinline__ i32x4 item_multiply(const i32x4 (&B)[8], uint32_t xaddr) {
  i32x4 A[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) A[t] = lds_read16i(xaddr + (t < 4 ? 16 * t : HALF + 16 * (t - 4)));
  i32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], B[t], acc, 0, 0, 0);
  return acc;
}

// Items of one K slice that multiply the same digit planes (the row blocks of a matrix) share their A fragments: read once
// (item_fragments), then item_mfma_shared() per item -- the lookups of item_mfma() without its eight 16-byte plane reads.
template <int HALF = 256>
__device__ __forceinline__ void item_fragments(uint32_t xaddr, i32x4 (&A)[8]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) A[t] = lds_read16i(xaddr + (t < 4 ? 16 * t : HALF + 16 * (t - 4)));
}
template <bool D4, bool R3 = false>
__device__ __forceinline__ i32x4 item_mfma_shared(const ItemAddr& ad, const i32x4 (&A)[8]) {
  constexpr int PIPE = QUIP_GEMV_PIPE;
  i32x4 acc = {0, 0, 0, 0};
  if constexpr (D4) {
    i32x4 B[8];
    auto issue = [&](int t) {
      B[t] = i32x4{(int)lds_read4(ad.a1l[t]), (int)lds_read4(ad.a2l[t]), (int)lds_read4(ad.a1h[t]), (int)lds_read4(ad.a2h[t])};
    };
#pragma unroll
    for (int t = 0; t < PIPE; ++t) issue(t);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t + PIPE < 8) issue(t + PIPE);
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], B[t], acc, 0, 0, 0);
    }
  } else {
    uint2 o[8][4];
    auto issue = [&](int t) {
      o[t][0] = lds_read8(ad.a1l[t]); o[t][1] = R3 ? make_uint2(0u, 0u) : lds_read8(ad.a2l[t]);
      o[t][2] = lds_read8(ad.a1h[t]); o[t][3] = lds_read8(ad.a2h[t]);
    };
#pragma unroll
    for (int t = 0; t < PIPE; ++t) issue(t);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      if (t + PIPE < 8) issue(t + PIPE);
      const i32x4 B = {(int)(o[t][0].x ^ o[t][1].x), (int)(o[t][0].y ^ o[t][1].y), (int)(o[t][2].x ^ o[t][3].x),
                       (int)(o[t][2].y ^ o[t][3].y)};
      acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], B, acc, 0, 0, 0);
    }
  }
  return acc;
}


// =====================================================================================================================
// Nibble mode (round 6).  The byte tables above cost two 8-byte look-ups per code: with 16 copies (the only size that fits
// beside the 84 KB of down's digit planes) every ds_read_b64 is a two-way bank conflict -- 4 LDS cycles instead of 2
// (MI355X_MICROARCH.md, LDS: lanes l and l + 16 of a 32-lane group share a bank pair whenever their indices have equal
// parity) -- and the product of the 70B launch ran at the LDS rate, not at the stream's.  4 w of an E8P12 weight is one of
// the SIXTEEN odd numbers in [-15, 15] (e8p12.py:82-103: +-{2, 6, 10, 14} + 1 - 2 par), so v = (4 w - 1) / 2 is a 4-bit
// two's-complement number, and the decode identity holds nibble by nibble:
//     v = c ^ (neg ? 0xE : 0) ^ (par ? 1 : 0),   c = grid byte / 2 (odd: -c = ~c + 1 = c ^ 0xE, c - 1 = c ^ 1, -c - 1 = c ^ 0xF).
// A code's eight weights are therefore ONE dword  T1n[abs] ^ T2n[sign]  of two 4-byte look-ups: 32 conflict-free copies of
// each table (ds_read_b32: two groups of 32 lanes on 32 banks, 2 LDS cycles) in the same 64 KB.  Entry byte b holds position
// b in its LOW nibble, biased by 8 (unsigned 0..15), and position 4 + b in its HIGH nibble (two's complement), so that the
// dword read as four int8 is  16 v_hi + lo_u.  No widening: the matrix core takes the bytes as they are, twice --
//     R1 = mfma(A, raw)               rows "hi" of A:  sum x_hi (16 v_hi + lo_u)
//     R2 = mfma(A, raw & 0x0f0f0f0f)  rows "hi":       sum x_hi lo_u             rows "lo":  sum x_lo (v_lo + 8)
// with the digits of positions 4..7 of every 8-group in A rows 0..2 (planes h, m, l) and those of positions 0..3 in rows
// 4..6.  Per plane  8 sum x 4w = (R1 - R2)[hi] + 16 R2[lo] + 8 SX[hi] - 120 SX[lo],  SX = plain digit sums (one more MFMA
// per A fragment against a B operand of ones, shared by every item of the K slice).  Same integers as the byte tables: the
// accumulator rows hold exactly 8 x what they held.  Per item: 32 four-byte look-ups (conflict free) + 4 A fragments instead
// of 32 eight-byte ones (two-way conflicts) + 8 A fragments; 32 v_perm + 16 v_xor + 16 v_and instead of 64 address
// operations + 32 v_xor; the same 8 MFMAs (+ 4 per K slice).
//
// LDS image of the tables: row idx (256 bytes) = T1n[idx] x 32 copies | T2n[idx] x 32 copies; address of a look-up =
// idx << 8 | (lane & 31) << 2 (| 128): one v_perm_b32 (lane_c: byte 0 = (lane & 31) << 2, byte 2 = 128 | (lane & 31) << 2).
// Digit planes for this mode ("half planes"): the natural dwords of a plane de-interleaved -- even dwords (positions 0..3
// of the 8-groups) and odd dwords (positions 4..7) each contiguous -- so that an A fragment is one 16-byte read.
struct T2nImage {
  uint32_t v[256];
  constexpr T2nImage() : v{} {
    for (int s = 0; s < 256; ++s) {
      int par = 0;
      for (int b = 0; b < 8; ++b) par ^= (s >> b) & 1;
      const int sv = s ^ par;
      uint32_t e = 0;
      for (int p = 0; p < 8; ++p) {
        const uint32_t nib = (((sv >> (7 - e8p_byte_of_pos(p))) & 1) ? 0xeu : 0u) ^ (uint32_t)par;
        e |= nib << (8 * (p & 3) + 4 * (p >> 2));
      }
      v[s] = e;
    }
  }
};
__device__ const T2nImage kT2nImg{};

// T1n entry from grid_packed_abs[e] (bytes 4a, byte 7 possibly negative: e8p12.py:63-79)
__device__ __forceinline__ uint32_t t1n_entry(uint2 packed) {
  const uint32_t lo = __builtin_amdgcn_perm(0u, packed.x, 0x03010200u), hi = __builtin_amdgcn_perm(0u, packed.y, 0x03010200u);
  return (((lo >> 1) & 0x0f0f0f0fu) ^ 0x08080808u) | (((hi >> 1) & 0x0f0f0f0fu) << 4);
}
constexpr int kNibTableBytes = 256 * 256;
// wave w < 8 owns table rows [32 w, +32): lanes 0..31 hold grid_packed_abs[32 w + l] (8 bytes, src), lanes 32..63 take
// the constant sign image; every lane writes its entry 32 times, copy (l + c) & 31 at step c (32 distinct banks per step)
__device__ __forceinline__ const uint2* table_source_ptr_nib(const uint64_t* grid, int lane, int wave) {
  return reinterpret_cast<const uint2*>(grid) + ((wave & 7) * 32 + (lane & 31));
}
__device__ __forceinline__ void fill_tables_nib(const u32x2& src, int lane, int wave) {
  const bool second = (lane & 32) != 0;
  const uint32_t row = (uint32_t)((wave & 7) * 32 + (lane & 31));
  const uint32_t val = second ? kT2nImg.v[row] : t1n_entry(make_uint2(src.x, src.y));
  const uint32_t rowbase = row * 256u + (second ? 128u : 0u);
#pragma unroll
  for (int c = 0; c < 32; ++c)
    *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)(rowbase + (((uint32_t)(lane + c)) & 31u) * 4u)) = val;
}
__device__ __forceinline__ uint32_t nib_lane_const(int lane) {
  const uint32_t l4 = ((uint32_t)lane & 31u) << 2;
  return l4 | ((128u | l4) << 16);
}
// 16 codes of this lane (dwords d[t] = code 2t | code 2t + 1 << 16; a code = abs << 8 | sign) -> 32 look-up addresses
__device__ __forceinline__ void item_addresses_nib(const u32x4& q0, const u32x4& q1, uint32_t lane_c, ItemAddr& ad) {
  const uint32_t d[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    ad.a1l[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0500u);
    ad.a2l[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0402u);
    ad.a1h[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0700u);
    ad.a2h[t] = __builtin_amdgcn_perm(d[t], lane_c, 0x0c0c0602u);
  }
}
// A fragments of a 512-k slice of half planes: step s covers k = 256 (s >> 1) + 64 q + 32 (s & 1) + [0, 32) of the slice;
// xaddr = this lane's row (plane / half) + slice * 256 + q * 32
// (HB: bytes between the two 128-byte halves of a slice in a half plane: 128, or more where the writer pads)
template <int HB = 128>
__device__ __forceinline__ void item_fragments_nib(uint32_t xaddr, i32x4 (&A)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) A[s] = lds_read16i(xaddr + 16 * (s & 1) + HB * (s >> 1));
}
// digit sums of the slice (every column of the result holds them: rows 0..2 the "hi" planes, rows 4..6 the "lo" ones)
__device__ __forceinline__ void item_digit_sums_nib(const i32x4 (&A)[4], i32x4& sx) {
  const i32x4 ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
#pragma unroll
  for (int s = 0; s < 4; ++s) sx = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s], ones, sx, 0, 0, 0);
}
// the item's look-ups and MFMAs: raw += A x dwords, msk += A x (dwords & 0x0f0f0f0f)
__device__ __forceinline__ void item_mfma_nib(const ItemAddr& ad, const i32x4 (&A)[4], i32x4& raw, i32x4& msk) {
  uint32_t o[4][8];
  auto lk = [&](int s) {
    o[s][0] = lds_read4(ad.a1l[2 * s]); o[s][1] = lds_read4(ad.a2l[2 * s]);
    o[s][2] = lds_read4(ad.a1h[2 * s]); o[s][3] = lds_read4(ad.a2h[2 * s]);
    o[s][4] = lds_read4(ad.a1l[2 * s + 1]); o[s][5] = lds_read4(ad.a2l[2 * s + 1]);
    o[s][6] = lds_read4(ad.a1h[2 * s + 1]); o[s][7] = lds_read4(ad.a2h[2 * s + 1]);
  };
  lk(0); lk(1);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s + 2 < 4) lk(s + 2);
    const i32x4 Br = {(int)(o[s][0] ^ o[s][1]), (int)(o[s][2] ^ o[s][3]), (int)(o[s][4] ^ o[s][5]), (int)(o[s][6] ^ o[s][7])};
    const i32x4 Bm = {Br.x & 0x0f0f0f0f, Br.y & 0x0f0f0f0f, Br.z & 0x0f0f0f0f, Br.w & 0x0f0f0f0f};
    raw = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s], Br, raw, 0, 0, 0);
    msk = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s], Bm, msk, 0, 0, 0);
  }
}
// what the lanes that own accumulator rows add for an item: q == 0 (D rows 0..2) and q == 1 (D rows 4..6) of column n;
// the sum over the K slices of a row is 8 x (sum of digit x 4w) per plane.  Branch free, per-lane factors:
//   v = (raw & rmask) + cm msk + cs sx,   (rmask, cm, cs) = (~0, -1, 8) for q == 0, (0, 16, -120) for q == 1
// (v_mad_i32_i24: |msk| <= 7 slices x 256 bytes x 128 x 15 < 2^23, |sx| smaller still; the compiler's own version of the
//  two-sided expression was three v_mul_lo_u32 and a nest of exec-mask branches per item)
struct NibLane { int cm, cs; uint32_t rmask; };
__device__ __forceinline__ NibLane nib_lane_factors(int q) {
  return NibLane{q == 0 ? -1 : 16, q == 0 ? 8 : -120, q == 0 ? 0xffffffffu : 0u};
}
// (NOT inline asm: raw / msk / sx come straight out of MFMAs, and the wait states between an MFMA and a VALU read of its
//  result are the COMPILER's to insert -- it does not look inside an asm statement; a hand-written v_mad_i32_i24 here read
//  accumulators that were still being written)
__device__ __forceinline__ int mad_i24(int a, int b, int c) { return __mul24(a, b) + c; }
__device__ __forceinline__ void item_rows_nib(const i32x4& raw, const i32x4& msk, const i32x4& sx, const NibLane& f, int (&v)[3]) {
  const int r[3] = {raw.x, raw.y, raw.z}, m[3] = {msk.x, msk.y, msk.z}, s[3] = {sx.x, sx.y, sx.z};
#pragma unroll
  for (int d = 0; d < 3; ++d) v[d] = mad_i24(m[d], f.cm, mad_i24(s[d], f.cs, (int)((uint32_t)r[d] & f.rmask)));
}
// the three atomic adds of an item's rows into the int32 accumulators at LDS byte address `dst` (this lane's row), from
// the low 32 lanes only (q < 2); the wave's exec mask is all ones before and after
__device__ __forceinline__ void lds_add3_low32(uint32_t dst, const int (&v)[3]) {
  // (exec_hi alone: how a 32-bit literal extends into a 64-bit scalar move is not something to depend on)
  asm volatile("s_mov_b32 exec_hi, 0\n\tds_add_u32 %0, %1\n\tds_add_u32 %0, %2 offset:4\n\tds_add_u32 %0, %3 offset:8\n\ts_mov_b32 exec_hi, -1"
               : : "v"(dst), "v"(v[0]), "v"(v[1]), "v"(v[2]) : "memory");
}
// ---- the nibble mode behind the byte tables' item interface (decode_block.hip: items that return ONE i32x4) ------------------------
// fragments of a K slice + their digit sums; an item's result = what this lane adds to its accumulator row (item_rows_nib)
struct NibFrags { i32x4 A[4]; i32x4 sx; };
template <int HB = 128>
__device__ __forceinline__ void nib_fragments(uint32_t xaddr, NibFrags& F) {
  item_fragments_nib<HB>(xaddr, F.A);
  F.sx = i32x4{0, 0, 0, 0};
  item_digit_sums_nib(F.A, F.sx);
}
__device__ __forceinline__ i32x4 nib_rows(const i32x4& raw, const i32x4& msk, const i32x4& sx, const NibLane& f) {
  int v[3];
  item_rows_nib(raw, msk, sx, f, v);
  return i32x4{v[0], v[1], v[2], 0};
}
__device__ __forceinline__ i32x4 nib_item(const ItemAddr& ad, const NibFrags& F, const NibLane& f) {
  i32x4 raw = {0, 0, 0, 0}, msk = {0, 0, 0, 0};
  item_mfma_nib(ad, F.A, raw, msk);
  return nib_rows(raw, msk, F.sx, f);
}
// an item's 32 look-ups now, its MFMAs later (the persistent launches decode inside hand-off waits): 16 dwords, dword 4 s + c =
// code c of MFMA step s
__device__ __forceinline__ void nib_decode(const ItemAddr& ad, uint32_t (&raw)[16]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    raw[4 * s + 0] = lds_read4(ad.a1l[2 * s]) ^ lds_read4(ad.a2l[2 * s]);
    raw[4 * s + 1] = lds_read4(ad.a1h[2 * s]) ^ lds_read4(ad.a2h[2 * s]);
    raw[4 * s + 2] = lds_read4(ad.a1l[2 * s + 1]) ^ lds_read4(ad.a2l[2 * s + 1]);
    raw[4 * s + 3] = lds_read4(ad.a1h[2 * s + 1]) ^ lds_read4(ad.a2h[2 * s + 1]);
  }
}
__device__ __forceinline__ i32x4 nib_multiply(const uint32_t* raw, const NibFrags& F, const NibLane& f) {
  i32x4 r = {0, 0, 0, 0}, m = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const i32x4 Br = {(int)raw[4 * s], (int)raw[4 * s + 1], (int)raw[4 * s + 2], (int)raw[4 * s + 3]};
    const i32x4 Bm = {Br.x & 0x0f0f0f0f, Br.y & 0x0f0f0f0f, Br.z & 0x0f0f0f0f, Br.w & 0x0f0f0f0f};
    r = __builtin_amdgcn_mfma_i32_16x16x64_i8(F.A[s], Br, r, 0, 0, 0);
    m = __builtin_amdgcn_mfma_i32_16x16x64_i8(F.A[s], Bm, m, 0, 0, 0);
  }
  return nib_rows(r, m, F.sx, f);
}
// this lane's A-fragment row of half planes (plane stride ps, "hi" half at + ho): A rows 0..2 = the planes' "hi" halves, rows
// 4..6 their "lo" halves, the other rows a copy of a neighbour (their results are never read)
__device__ __forceinline__ uint32_t nib_row_offset(int n, int ps, int ho) {
  return (uint32_t)(n < 4 ? min(n, 2) * ps + ho : min(n - 4, 2) * ps);
}

// byte offset of digit k of a plane inside its half plane, and which half (0: positions 0..3, 1: positions 4..7)
__device__ __forceinline__ constexpr int nib_half_of(int k) { return (k >> 2) & 1; }
__device__ __forceinline__ constexpr int nib_byte_of(int k) { return ((k >> 3) << 2) | (k & 3); }

}  // namespace
}  // namespace quip
