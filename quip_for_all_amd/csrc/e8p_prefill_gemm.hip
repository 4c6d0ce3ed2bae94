// Fused E8P12 dequant + GEMM for batches (M >= 32 rows: prompt prefill), gfx950.
//
//   Y[m, n] = sum_k X[m, k] * W[n, k],   W = decode(Qidxs (N, K/8) int16),  fp16 x fp16 -> fp32 -> fp16
//
// Replaces, for M >= 32, the reference's decompress + dense GEMM pair (codebook/e8p12.py:152-155:
// `decompress_e8p_origorder` materialises the dense fp16 W, quip_cuda/origin_order.cu:837-885, then `x @ W.T` runs
// in cuBLAS): here W never exists in memory.  Arithmetic is the reference's: exact fp16 weights, fp16 activations,
// fp32 accumulation (v_mfma_f32_32x32x16_f16), one rounding of the result to fp16.
//
// Why this maps well on CDNA4: the B operand of v_mfma_f32_32x32x16_f16 is, per lane, 8 consecutive-k fp16 values
// of one output column -- exactly the 8 weights of ONE E8P code.  So a lane decodes one 16-bit code (two 8-byte
// LDS table lookups, one XOR -> eight int8 = 4w; v_perm + v_pk_add_f16 turn them into fp16 through the
// 0x5c00 | (4w + 128) = 288 + w identity, cf. origin_order.cu:275-282) and holds a complete B fragment: no dense
// tile, no LDS round trip, no transposition for W.
//
// Workgroup = 256 x 256 output tile, 8 waves; wave w owns columns [32 w, 32 w + 32) for all 256 rows (8 row blocks
// of 32 -> 8 accumulator tiles = 128 registers).  With this split every code of the tile is decoded exactly once
// per workgroup (a 2 x 4 wave grid would decode each twice), the codes go straight from global memory to registers
// (nobody shares them), and only X goes through LDS: 256 x 64 fp16 per K step, double buffered, filled by
// global_load_lds_dwordx4 (no staging registers).  The LDS image is lane-linear, so the bank swizzle
// (16-byte chunk c of row m stored at c ^ ((m >> 1) & 7): conflict-free ds_read_b128 of the A fragments) is
// applied to the per-lane SOURCE address.  The MFMA's k index is relabelled so that a lane's four codes of a K
// step (8 bytes, one load) feed four consecutive MFMAs: k block l >> 5 of step j <-> k = 32 (l >> 5) + 8 j.
//
// Workgroup -> tile: blocks are dealt to the XCDs round robin (b % 8); an XCD walks the column tiles of one row tile
// before moving to the next row tile, so the 32 workgroups resident on an XCD share two X slabs in its L2.
//
// Bound: MFMA (2 M N K flops at the dense fp16 peak).  LDS: 2 x 16 KiB tables (8 copies) + 4 x 32 KiB X tiles
// (one multiplied, one landed, two in flight: counted vmcnt + raw s_barrier).  Measured: three tiles with 16 table
// copies (tile t + 1 awaited one tile time after its request) and this layout (two tile times) run at the same
// rate, i.e. the K loop does not wait on memory; halving the A-fragment LDS reads (experiment) changed 2-3 %.
//
// The other codebooks on the same tile machinery (template parameter MODE; the decode of a B fragment is the only
// thing that differs -- the role of the reference's BLayout_* classes, origin_order.cu:143-385, whose M >= 32 use is
// decompress_* + `x @ W.T`: e8p12_rvq4.py:50-67, e8p12_rvq3.py:109-129, d4.py:128-139, hi.py:52-63):
//   MODE 1 = E8P12RVQ4B: 32-bit codes main << 16 | residual, four lookups, w = fma(s, w_residual, w_main) in packed
//            fp16 -- one rounding per weight, the dense W of decompress_e8prvq4_origorder (origin_order.cu:337-385);
//   MODE 4 = E8P12RVQ3B: the checkpoint's 3-byte codes (12 bytes per lane and tile), main through T1 / T2, the
//            residual through a third table (the packed E81B entries, 8 copies of 4 bytes) as in e8p_skinny_gemm.hip;
//   MODE 2 = D4: two code bytes per B fragment, the table holds the fp16 entries themselves (two 8-byte lookups, no
//            arithmetic);
//   MODE 3 = HI: eight nibbles per B fragment, w = nibble - 7.5 (0x4c00 | n << 6 = 16 + n, one packed add), no table.
// These run the eight-wave layout only (every code decoded once per workgroup).
#include <cstdlib>
#include <type_traits>

#include "quip_device.hip.h"
#include "quip_internal.h"

namespace quip {

namespace {

typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t pu32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t pu32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 256, kBN = 256, kBK = 64;
constexpr int kRep = 8;
constexpr int kT1 = 0;
constexpr int kT2 = 256 * kRep * 8;          // 16 KiB
constexpr int kA = 2 * kT2;                  // 32 KiB; behind it the X tiles (128 bytes per row)
#ifndef QUIP_PREFILL_INTERLEAVE
#define QUIP_PREFILL_INTERLEAVE 1
#endif
constexpr bool kInterleave = QUIP_PREFILL_INTERLEAVE != 0;
constexpr int kStages = 4;                   // X tiles in LDS: one multiplied, one landed, two in flight
constexpr int kRep3 = 8;                     // MODE 4: copies of the residual table (4-byte entries) behind T2
constexpr int kT3Bytes = 256 * kRep3 * 4;    // 8 KiB
constexpr int lds_bytes(int bm, int mode = 0) { return kA + (mode == 4 ? kT3Bytes : 0) + kStages * bm * kBK * 2; }   // 160 KiB at 256 rows
// bytes of codes per 8 weights / dwords a lane loads per tile and column block (32 k) / table lookups per B fragment
constexpr int mode_code_bytes(int mode) { return mode == 1 || mode == 3 ? 4 : mode == 4 ? 3 : 2; }
constexpr int mode_lookups(int mode) { return mode == 1 ? 4 : mode == 4 ? 3 : mode == 3 ? 0 : 2; }
constexpr int mode_convert_valu(int mode) { return mode == 1 ? 30 : mode == 4 ? 28 : mode == 3 ? 12 : mode == 2 ? 0 : 12; }

// sign table image (same statement as the GEMV's: 4w = T1[abs] ^ T2[sign] byte-wise, origin_order.cu:211-253)
struct PT2Image {
  uint2 v[256];
  constexpr PT2Image() : v{} {
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
__device__ const PT2Image kPT2Img{};

__device__ __forceinline__ uint2 p_lds_read8(uint32_t addr) {
  const pu32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) pu32x2*>((uintptr_t)addr);
  return make_uint2(v.x, v.y);
}
__device__ __forceinline__ f16x8v p_lds_read_frag(uint32_t addr) {
  return *reinterpret_cast<const __attribute__((address_space(3))) f16x8v*>((uintptr_t)addr);
}

// four int8 (as 4w + 128, i.e. bytes u) -> two packed fp16 pairs: 0x5c00 | u = 256 + u / 4 = 288 + w
__device__ __forceinline__ void bytes_to_f16x4(uint32_t u4, uint32_t& lo, uint32_t& hi) {
  const uint32_t k5c = 0x5c5c5c5cu;
  const uint32_t a = __builtin_amdgcn_perm(u4, k5c, 0x00050004u);   // [u0, 5c, u1, 5c]
  const uint32_t b = __builtin_amdgcn_perm(u4, k5c, 0x00070006u);   // [u2, 5c, u3, 5c]
  const f16x2 m288 = {(f16)-288.f, (f16)-288.f};
  lo = as_u32(as_f16x2(a) + m288);
  hi = as_u32(as_f16x2(b) + m288);
}

// Wave layout: WM x WN waves; a wave owns NB row blocks x NC column blocks of 32 (tile = 32 NB WM rows x 32 NC WN = 256
// columns).  <8, 1, 1, 8>: every code decoded once per workgroup, every A fragment read from LDS by all eight waves
// (one ds_read_b128 per MFMA: the LDS pipe is as busy as the matrix cores); <4, 2, 2, 4>: an A fragment feeds two
// MFMAs (half the LDS traffic), a code is decoded by the two waves that share its columns; <8, 2, 1, 4>: four waves,
// both.  128-row tiles (<4, 1, 1, 8>, <2, 2, 2, 4>) for launches whose 256-row tiles would not fill the GPU.
template <int NB, int NC, int WM, int WN, int MODE = 0>
__global__ __launch_bounds__(64 * WM * WN) void e8p_prefill_gemm_kernel(const f16* __restrict__ X,
                                                                        const uint8_t* __restrict__ Wc,
                                                                        const uint64_t* __restrict__ grid,
                                                                        f16* __restrict__ Y, int M, int N, int K, int MT,
                                                                        int NT, float resid_scale,
                                                                        const uint32_t* __restrict__ grid2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  static_assert(32 * NC * WN == kBN, "256 columns per tile");
  static_assert(MODE == 0 || NC == 1, "the other codebooks: one column block per wave");
  constexpr int NW = WM * WN, BM = 32 * NB * WM;
  constexpr int kTileBytes = BM * kBK * 2;
  constexpr int CB = mode_code_bytes(MODE);     // code bytes per 8 weights
  constexpr int CW = CB == 3 ? 3 : CB;          // dwords per lane, tile and column block (4 fragments of 8 weights)
  constexpr int NL = mode_lookups(MODE);
  constexpr int kT3 = kA;                       // MODE 4: the residual table, the X tiles behind it
  constexpr int kX = kA + (MODE == 4 ? kT3Bytes : 0);
  using CodeT = std::conditional_t<CW == 2, pu32x2, std::conditional_t<CW == 3, pu32x3, pu32x4>>;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  int mt, nt;
  if (MT >= 8) {
    const int xcd = (int)blockIdx.x & 7, idx = (int)blockIdx.x >> 3;
    mt = (idx / NT) * 8 + xcd;
    nt = idx - (idx / NT) * NT;
    if (mt >= MT) return;
  } else {   // fewer row tiles than XCDs: plain order (the XCD-aware one would leave 8 - MT XCDs idle)
    mt = (int)blockIdx.x % MT;
    nt = (int)blockIdx.x / MT;
  }
  const int m0 = mt * BM, n0 = nt * kBN;
  const int KT = K / kBK;

  // ---- X tile loader (global_load_lds): instruction i of this wave fills LDS slots [(NW i + wave) * 64, +64) of the
  // tile; slot s = (row s >> 3, stored chunk s & 7) holds source chunk (s & 7) ^ ((row >> 1) & 7) of that row
  constexpr int XL = BM / 8 / NW;       // loader instructions per wave and tile (8 rows x 128 bytes each)
  const f16* xsrc[XL];
#pragma unroll
  for (int i = 0; i < XL; ++i) {
    const int row = (NW * i + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    const int gr = min(m0 + row, M - 1);
    xsrc[i] = X + (size_t)gr * K + c * 8;
  }
  auto issue_x = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < XL; ++i)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(xsrc[i] + (size_t)t * kBK),
          (__attribute__((address_space(3))) void*)(smem + kX + buf * kTileBytes + (NW * i + wave) * 1024), 16, 0, 0);
  };
  // ---- codes: lane (n = lane & 31, kb = lane >> 5) holds the 4 codes k = 64 t + 32 kb + 8 j .. (j = 0..3) of columns
  // n0 + 32 (NC wn + c) + n, c < NC
  int ncol[NC];
  const uint8_t* wsrc[NC];
  const int kb = lane >> 5;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    ncol[c] = n0 + 32 * (NC * wn + c) + (lane & 31);
    wsrc[c] = Wc + ((size_t)min(ncol[c], N - 1) * (K >> 3) + kb * 4) * CB;
  }
  // (asm: beside LDS-DMA loads in flight hipcc waits vmcnt(0) for any ordinary register load, which would drain the
  //  prefetch queue every tile; all VMEM traffic of the K loop is counted by hand instead)
  auto load_codes = [&](CodeT (&dst)[NC], int t) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const uint8_t* src = wsrc[c] + (size_t)t * (kBK / 8 * CB);
      if constexpr (CW == 2)
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(dst[c]) : "v"(src) : "memory");
      else if constexpr (CW == 3)
        asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(dst[c]) : "v"(src) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[c]) : "v"(src) : "memory");
    }
  };

  // tiles 0, 1 and 2 are requested here, tile t + 3 in the middle of tile t.  Per tile and wave: NC code loads + XL
  // LDS-DMA instructions.
  // cq[]: registers the code loads write (in flight); cv[]: the codes of the current / next tile, taken over by an
  // asm that waits and then moves them (a register in flight is never a tied operand: the compiler may copy a tied
  // operand ahead of the asm, i.e. ahead of the wait -- see e8p_skinny_gemm.hip)
  CodeT cq[kStages][NC], cv[2][NC];
  load_codes(cq[0], 0);
  issue_x(0, 0);
  load_codes(cq[1], min(1, KT - 1));      // (past the end: tile KT - 1 again -- the queue depth stays constant, so
  issue_x(min(1, KT - 1), 1);             //  every wait below is the same counted wait, with no branch around it)
  load_codes(cq[2], min(2, KT - 1));
  issue_x(min(2, KT - 1), 2);
  auto take = [](CodeT (&dst)[NC], const CodeT (&src)[NC], auto nw) {
    if constexpr (CW == 3)
      asm volatile("s_waitcnt vmcnt(%6)\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5"
                   : "=&v"(dst[0].x), "=&v"(dst[0].y), "=&v"(dst[0].z)
                   : "v"(src[0].x), "v"(src[0].y), "v"(src[0].z), "n"(decltype(nw)::value)
                   : "memory");
    else if constexpr (CW == 4)
      asm volatile("s_waitcnt vmcnt(%8)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                   : "=&v"(dst[0].x), "=&v"(dst[0].y), "=&v"(dst[0].z), "=&v"(dst[0].w)
                   : "v"(src[0].x), "v"(src[0].y), "v"(src[0].z), "v"(src[0].w), "n"(decltype(nw)::value)
                   : "memory");
    else if constexpr (NC == 1)
      asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
                   : "=&v"(dst[0].x), "=&v"(dst[0].y)
                   : "v"(src[0].x), "v"(src[0].y), "n"(decltype(nw)::value)
                   : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%8)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                   : "=&v"(dst[0].x), "=&v"(dst[0].y), "=&v"(dst[1].x), "=&v"(dst[1].y)
                   : "v"(src[0].x), "v"(src[0].y), "v"(src[1].x), "v"(src[1].y), "n"(decltype(nw)::value)
                   : "memory");
  };

  // ---- tables: T1' = (4a | 1) ^ 0x80.. (the ^0x80 turns 4w into the unsigned byte 4w + 128 the fp16 conversion
  // wants), T2 = sign masks; kRep copies each, copy (lane + c) & (kRep - 1) at step c
#pragma unroll
  for (int r = 0; r < 8 / NW; ++r) {
    const int e = (wave + NW * r) * 32 + (lane & 31);
    const bool second = (lane & 32) != 0;
    if constexpr (MODE == 3) break;   // HI: no table
    // (D4: both regions hold the fp16 entries as they are -- the fragment's first code byte looks up T1, its second T2)
    const uint2 raw = (second && MODE != 2) ? kPT2Img.v[e] : reinterpret_cast<const uint2*>(grid)[e];
    const uint32_t t1x = (__builtin_amdgcn_perm(0u, raw.x, 0x03010200u) | 0x01010101u) ^ 0x80808080u;
    const uint32_t t1y = (__builtin_amdgcn_perm(0u, raw.y, 0x03010200u) | 0x01010101u) ^ 0x80808080u;
    const pu32x2 val = {(second || MODE == 2) ? raw.x : t1x, (second || MODE == 2) ? raw.y : t1y};
    const uint32_t rowbase = (second ? (uint32_t)kT2 : (uint32_t)kT1) + (uint32_t)e * (kRep * 8);
#pragma unroll
    for (int c = 0; c < kRep; ++c) {
      const uint32_t copy = (uint32_t)(lane + c) & (kRep - 1);
      *reinterpret_cast<__attribute__((address_space(3))) pu32x2*>((uintptr_t)(rowbase + copy * 8)) = val;
    }
  }
  if constexpr (MODE == 4) {
    // residual table: entry e of the packed E81B table (eight int4 = 2 x value), kRep3 copies of 4 bytes
    if (tid < 256) {
      const uint32_t val = grid2[tid];
#pragma unroll
      for (int c = 0; c < kRep3; ++c)
        *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>(
            (uintptr_t)((uint32_t)kT3 + (uint32_t)tid * (kRep3 * 4) + (((uint32_t)(lane + c) & (kRep3 - 1)) << 2))) = val;
    }
  }
  const uint32_t lane_c1 = (uint32_t)(lane & (kRep - 1)) << 3;
  const uint32_t lane_c2 = lane_c1 | (uint32_t)kT2;
  const uint32_t lane_c3 = ((uint32_t)(lane & (kRep3 - 1)) << 2) | (uint32_t)kT3;
  const f16 rs16 = (f16)resid_scale;
  const f16x2 rs2 = {rs16, rs16};
  // A fragment address of this lane for k step j (without tile base / row block): row m = lane & 31
  const int m = lane & 31;
  uint32_t aoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) aoff[j] = (uint32_t)(wm * NB * 4096 + m * 128 + (((kb * 4 + j) ^ ((m >> 1) & 7)) << 4));

  f32x16 acc[NC][NB];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][b][r] = 0.f;

  // ---- the K loop.  k steps of 16 (8 MFMAs per wave each) flow across the 64-wide tiles without a bubble: during
  // step g the two table lookups and the eight A fragments of step g + 1 are requested (asm LDS reads, pinned by
  // sched_barrier: left to itself the compiler issues two fragment reads, waits, issues two MFMAs); after four of the
  // step's MFMAs the lookups have landed (lgkmcnt(8): the oldest two of ten reads) and the next B fragment is
  // decoded under the other four.  The only workgroup synchronisation is in the MIDDLE of a tile: "tile t + 1 has
  // landed" (vmcnt(0) + barrier), which also says that everybody is done with tile t - 1, whose buffer the loads
  // of tile t + 2 then take.  So step 3 of tile t can already request step 0 of tile t + 1.
  constexpr int NT_ = NL > 2 ? NL : 2;
  pu32x2 tl[2][NC][NT_];  // [parity of the step][column block][table entries: T1 / T2 (main), T1 / T2 or T3 (residual); HI: .x = the code]
  pu32x4 Af[2][NB];
  // the 32 code bits (MODE 0 / 2: the 16) of fragment jj of a lane's tile piece
  auto code_dword = [](const CodeT& cd, int jj) -> uint32_t {
    if constexpr (MODE == 0 || MODE == 2) {
      return jj < 2 ? cd.x : cd.y;
    } else if constexpr (MODE == 4) {
      // 12 landed bytes = four 3-byte codes [residual index, e8p lo, e8p hi] -> main << 16 | residual << 8
      return jj == 0   ? cd.x << 8
             : jj == 1 ? __builtin_amdgcn_perm(cd.y, cd.x, 0x0504030cu)
             : jj == 2 ? __builtin_amdgcn_perm(cd.z, cd.y, 0x0403020cu)
                       : cd.z & 0xffffff00u;
    } else {
      return jj == 0 ? cd.x : jj == 1 ? cd.y : jj == 2 ? cd.z : cd.w;
    }
  };
  auto request = [&](const CodeT (&cvt)[NC], int jj, uint32_t aj, pu32x2 (&tt)[NC][NT_], pu32x4 (&A8)[NB]) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const uint32_t d = code_dword(cvt[c], jj);
      if constexpr (MODE == 0) {
        uint32_t a1, a2;
        if (jj & 1) {
          a1 = ((d >> 18) & 0x3fc0u) | lane_c1;      // entry row = 8 copies x 8 bytes
          a2 = ((d >> 10) & 0x3fc0u) | lane_c2;
        } else {
          a1 = ((d >> 2) & 0x3fc0u) | lane_c1;
          a2 = ((d << 6) & 0x3fc0u) | lane_c2;
        }
        asm volatile("ds_read_b64 %0, %1" : "=v"(tt[c][0]) : "v"(a1));
        asm volatile("ds_read_b64 %0, %1" : "=v"(tt[c][1]) : "v"(a2));
      } else if constexpr (MODE == 2) {
        // D4: the fragment's first code byte (weights 0..3) / second (weights 4..7)
        uint32_t a1, a2;
        if (jj & 1) {
          a1 = ((d >> 10) & 0x3fc0u) | lane_c1;
          a2 = ((d >> 18) & 0x3fc0u) | lane_c2;
        } else {
          a1 = ((d << 6) & 0x3fc0u) | lane_c1;
          a2 = ((d >> 2) & 0x3fc0u) | lane_c2;
        }
        asm volatile("ds_read_b64 %0, %1" : "=v"(tt[c][0]) : "v"(a1));
        asm volatile("ds_read_b64 %0, %1" : "=v"(tt[c][1]) : "v"(a2));
      } else if constexpr (MODE == 3) {
        tt[c][0].x = d;
      } else {
        // main code in the high half: abs index (bits 24..31) through T1, sign byte (16..23) through T2
        const uint32_t a1 = ((d >> 18) & 0x3fc0u) | lane_c1, a2 = ((d >> 10) & 0x3fc0u) | lane_c2;
        asm volatile("ds_read_b64 %0, %1" : "=v"(tt[c][0]) : "v"(a1));
        asm volatile("ds_read_b64 %0, %1" : "=v"(tt[c][1]) : "v"(a2));
        if constexpr (MODE == 1) {
          const uint32_t b1 = ((d >> 2) & 0x3fc0u) | lane_c1, b2 = ((d << 6) & 0x3fc0u) | lane_c2;
          asm volatile("ds_read_b64 %0, %1" : "=v"(tt[c][2]) : "v"(b1));
          asm volatile("ds_read_b64 %0, %1" : "=v"(tt[c][3]) : "v"(b2));
        } else {
          const uint32_t b1 = ((d >> 3) & 0x1fe0u) | lane_c3;      // residual index (bits 8..15): 8 copies x 4 bytes
          asm volatile("ds_read_b32 %0, %1" : "=v"(tt[c][2].x) : "v"(b1));
        }
      }
    }
    asm volatile("ds_read_b128 %0, %1" : "=v"(A8[0]) : "v"(aj));
    asm volatile("ds_read_b128 %0, %1 offset:4096" : "=v"(A8[1]) : "v"(aj));
    if constexpr (NB >= 4) {
      asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(A8[2]) : "v"(aj));
      asm volatile("ds_read_b128 %0, %1 offset:12288" : "=v"(A8[3]) : "v"(aj));
    }
    if constexpr (NB == 8) {
      asm volatile("ds_read_b128 %0, %1 offset:16384" : "=v"(A8[4]) : "v"(aj));
      asm volatile("ds_read_b128 %0, %1 offset:20480" : "=v"(A8[5]) : "v"(aj));
      asm volatile("ds_read_b128 %0, %1 offset:24576" : "=v"(A8[6]) : "v"(aj));
      asm volatile("ds_read_b128 %0, %1 offset:28672" : "=v"(A8[7]) : "v"(aj));
    }
  };
  // all fragment reads of a request have landed
  auto landed = [&](pu32x4 (&A8)[NB]) {
    if constexpr (NB == 8)
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(A8[0]), "+v"(A8[1]), "+v"(A8[2]), "+v"(A8[3]), "+v"(A8[4]), "+v"(A8[5]), "+v"(A8[6]), "+v"(A8[7]));
    else if constexpr (NB == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A8[0]), "+v"(A8[1]), "+v"(A8[2]), "+v"(A8[3]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A8[0]), "+v"(A8[1]));
  };
  // the lookups of a request have landed (the NB fragment reads behind them may still be in flight)
  auto looked_up = [&](pu32x2 (&tt)[NC][NT_]) {
    if constexpr (MODE == 3)
      ;
    else if constexpr (MODE == 1)
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(tt[0][0]), "+v"(tt[0][1]), "+v"(tt[0][2]), "+v"(tt[0][3]) : "n"(NB));
    else if constexpr (MODE == 4)
      asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(tt[0][0]), "+v"(tt[0][1]), "+v"(tt[0][2].x) : "n"(NB));
    else if constexpr (NC == 1)
      asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(tt[0][0]), "+v"(tt[0][1]) : "n"(NB));
    else
      asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(tt[0][0]), "+v"(tt[0][1]), "+v"(tt[1][0]), "+v"(tt[1][1]) : "n"(NB));
  };
  auto frag_b = [&](const pu32x2 (&tt)[NT_]) -> f16x8v {
    if constexpr (MODE == 2) {
      return __builtin_bit_cast(f16x8v, pu32x4{tt[0].x, tt[0].y, tt[1].x, tt[1].y});
    } else if constexpr (MODE == 3) {
      // nibble i of the code is column [0, 2, 4, 6, 1, 3, 5, 7][i] of its 8-group: the fragment's fp16 pair d is
      // (nibble d, nibble d + 4); 0x4c00 | n << 6 is the fp16 number 16 + n, 23.5 is one too: the subtraction is exact
      const uint32_t c = tt[0].x;
      const f16x2 off = {(f16)-23.5f, (f16)-23.5f};
      const uint32_t sh[4] = {c << 6, c << 2, c >> 2, c >> 6};
      uint32_t w[4];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) w[dd] = as_u32(as_f16x2((sh[dd] & 0x03c003c0u) | 0x4c004c00u) + off);
      return __builtin_bit_cast(f16x8v, pu32x4{w[0], w[1], w[2], w[3]});
    } else {
      uint32_t m[4];
      bytes_to_f16x4(tt[0].x ^ tt[1].x, m[0], m[1]);
      bytes_to_f16x4(tt[0].y ^ tt[1].y, m[2], m[3]);
      if constexpr (MODE == 1) {
        uint32_t r[4];
        bytes_to_f16x4(tt[2].x ^ tt[3].x, r[0], r[1]);
        bytes_to_f16x4(tt[2].y ^ tt[3].y, r[2], r[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) m[i] = as_u32(__builtin_elementwise_fma(rs2, as_f16x2(r[i]), as_f16x2(m[i])));
      } else if constexpr (MODE == 4) {
        // residual nibble n = int4 of 2 x value: 0x4c00 | (n ^ 8) << 6 is 16 + (n ^ 8); x 0.5 - 12 = 0.5 ((n ^ 8) - 8), exact
        const uint32_t c = tt[2].x;
        const f16x2 half2 = {(f16)0.5f, (f16)0.5f}, m12 = {(f16)-12.f, (f16)-12.f};
        const uint32_t sh[4] = {c << 6, c << 2, c >> 2, c >> 6};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f16x2 res = __builtin_elementwise_fma(as_f16x2((sh[i] & 0x03c003c0u) ^ 0x4e004e00u), half2, m12);
          m[i] = as_u32(__builtin_elementwise_fma(rs2, res, as_f16x2(m[i])));
        }
      }
      return __builtin_bit_cast(f16x8v, pu32x4{m[0], m[1], m[2], m[3]});
    }
  };
  // tiles 0 and 1 (and the tables) are in LDS; first step's operands
  take(cv[0], cq[0], std::integral_constant<int, 0>{});   // (everything requested so far has landed)
  __syncthreads();
  request(cv[0], 0, (uint32_t)kX + aoff[0], tl[0], Af[0]);
  looked_up(tl[0]);
  landed(Af[0]);
  f16x8v B[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) B[c] = frag_b(tl[0][c]);

  // one tile: `st` = t % 4 (its buffer / code register), compile-time through the 4x unrolled loop
  auto tile = [&](int t, auto stc) {
    constexpr int st = decltype(stc)::value;
    constexpr int st1 = (st + 1) % kStages, st3 = (st + 3) % kStages, cur = st & 1, nxt = cur ^ 1;
    const uint32_t abase = (uint32_t)(kX + st * kTileBytes);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c2 = j & 1, nx = c2 ^ 1;
      // operands of the next step: step j + 1 of this tile, or step 0 of the next one
      if (j < 3)
        request(cv[cur], j + 1, abase + aoff[j + 1], tl[nx], Af[nx]);
      else
        request(cv[nxt], 0, (uint32_t)(kX + st1 * kTileBytes) + aoff[0], tl[nx], Af[nx]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = 0; b < NB / 2; ++b)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          acc[c][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, Af[c2][b]), B[c], acc[c][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      looked_up(tl[nx]);   // the 2 NC oldest of 2 NC + NB
      f16x8v Bn[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) Bn[c] = frag_b(tl[nx][c]);
      if constexpr (!kInterleave) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b = NB / 2; b < NB; ++b)
#pragma unroll
        for (int c = 0; c < NC; ++c)
          acc[c][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8v, Af[c2][b]), B[c], acc[c][b], 0, 0, 0);
      if constexpr (kInterleave) {
        // the next fragments' conversion (12 VALU instructions per code) between this half's MFMAs: one MFMA, then what
        // fits under its 32 cycles
#pragma unroll
        for (int i = 0; i < NB / 2 * NC; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, (mode_convert_valu(MODE) * NC + NB / 2 * NC - 1) / (NB / 2 * NC) + (MODE == 2), 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      landed(Af[nx]);
#pragma unroll
      for (int c = 0; c < NC; ++c) B[c] = Bn[c];
      if (j == 1) {
        // middle of the tile: tile t + 1 (requested in the middle of tile t - 2: two tile times ago) has landed
        // everywhere -- the operations of tile t + 2 may still be in flight; tile t - 1 is dead, its buffer takes
        // tile t + 3 (past the end: tile KT - 1 again, harmless, keeps the code uniform)
        take(cv[nxt], cq[st1], std::integral_constant<int, NC + XL>{});
        __builtin_amdgcn_s_barrier();
        load_codes(cq[st3], min(t + 3, KT - 1));
        issue_x(min(t + 3, KT - 1), st3);
      }
    }
  };
  for (int t = 0; t < KT; t += kStages) {
    tile(t, std::integral_constant<int, 0>{});
    if (t + 1 < KT) tile(t + 1, std::integral_constant<int, 1>{});
    if (t + 2 < KT) tile(t + 2, std::integral_constant<int, 2>{});
    if (t + 3 < KT) tile(t + 3, std::integral_constant<int, 3>{});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing filler tile
  // ---- epilogue: D row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the 32-row block, column = lane & 31.  Neighbour
  // lanes exchange one value per register pair so that every lane stores two adjacent columns (4 bytes): even lanes
  // the even register's row, odd lanes the odd register's
  const bool odd = (lane & 1) != 0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float mine = odd ? acc[c][b][r + 1] : acc[c][b][r];
        const float give = odd ? acc[c][b][r] : acc[c][b][r + 1];
        const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
        const int rr = odd ? r + 1 : r;
        const int row = m0 + 32 * (NB * wm + b) + (rr & 3) + 8 * (rr >> 2) + 4 * kb;
        const int col = ncol[c] & ~1;
        const uint32_t pk = odd ? pack_f16(got, mine) : pack_f16(mine, got);
        if (row < M && col + 1 < N) {
          *reinterpret_cast<uint32_t*>(Y + (size_t)row * N + col) = pk;
        } else if (row < M && col < N) {
          Y[(size_t)row * N + col] = odd ? (f16)got : (f16)mine;
        }
      }
    }
  }
}

}  // namespace

bool e8p_prefill_gemm_supported(int64_t m, int n, int k) {
  return m >= 1 && n >= 2 && n % 2 == 0 && k >= kBK && k % kBK == 0 && m < ((int64_t)1 << 31);
}

// mode 0: E8P12 (int16 codes (n, k/8)), 1: E8P12RVQ4B (int32 codes, resid_scale), 2: D4 (uint8 codes (n, k/4), grid = the fp16
// (256, 4) table), 3: HI (int32 codes, no table), 4: E8P12RVQ3B (packed 3-byte codes, grid2 = the packed E81B table)
static int prefill_launch_mode(int mode, const void* x, const void* qidxs, const void* grid, const void* grid2, float resid_scale,
                               void* y, int64_t m, int n, int k, hipStream_t stream) {
  if (!e8p_prefill_gemm_supported(m, n, k)) return QUIP_ERR_UNSUPPORTED;
  const int NT = (n + kBN - 1) / kBN;
  // 128-row tiles while 256-row tiles would leave CUs without a workgroup (the K loop of a workgroup takes the same
  // time whatever the tile height: a second half-filled round costs less than idle CUs)
  static const int force = getenv("QUIP_PREFILL_TILE") ? atoi(getenv("QUIP_PREFILL_TILE")) : 0;   // 128 / 256: experiments
  // wave layout (see the kernel): 0 = eight waves of 256 x 32, 1 = 2 x 4 waves of 128 x 64, 2 = four waves of 256 x 64
  static const int layout = getenv("QUIP_PREFILL_LAYOUT") ? atoi(getenv("QUIP_PREFILL_LAYOUT")) : 0;   // measured: 0 is the fastest
  // (E8P12RVQ3B: a third table; E8P12RVQ4B: four table entries per fragment in flight, the 256-row tile's 128
  //  accumulator registers would leave it 9 registers short -- 128-row tiles always)
  const bool half = (mode == 4 || mode == 1) ? true : force ? force == 128 : (m + kBM - 1) / kBM * NT < device_cu_count();
  const int bm = half ? kBM / 2 : kBM;
  const int MT = (int)((m + bm - 1) / bm);
  const int64_t blocks = MT >= 8 ? (int64_t)((MT + 7) / 8) * NT * 8 : (int64_t)MT * NT;
  if (blocks > 0x7fffffff) return QUIP_ERR_UNSUPPORTED;
  auto go = [&](auto kern, DynLdsCache& configured, int threads) -> int {
    const int lds = lds_bytes(bm, mode);
    if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), lds, stream, reinterpret_cast<const f16*>(x),
                       reinterpret_cast<const uint8_t*>(qidxs), reinterpret_cast<const uint64_t*>(grid),
                       reinterpret_cast<f16*>(y), (int)m, n, k, MT, NT, resid_scale, reinterpret_cast<const uint32_t*>(grid2));
    return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
  };
  static DynLdsCache c[13];   // per instantiation, per device
  if (mode == 1) return go(e8p_prefill_gemm_kernel<4, 1, 1, 8, 1>, c[6], 512);
  if (mode == 2) return half ? go(e8p_prefill_gemm_kernel<4, 1, 1, 8, 2>, c[8], 512) : go(e8p_prefill_gemm_kernel<8, 1, 1, 8, 2>, c[9], 512);
  if (mode == 3) return half ? go(e8p_prefill_gemm_kernel<4, 1, 1, 8, 3>, c[10], 512) : go(e8p_prefill_gemm_kernel<8, 1, 1, 8, 3>, c[11], 512);
  if (mode == 4) return go(e8p_prefill_gemm_kernel<4, 1, 1, 8, 4>, c[12], 512);
  if (layout == 2) return half ? go(e8p_prefill_gemm_kernel<4, 2, 1, 4>, c[0], 256) : go(e8p_prefill_gemm_kernel<8, 2, 1, 4>, c[1], 256);
  if (layout == 1) return half ? go(e8p_prefill_gemm_kernel<2, 2, 2, 4>, c[2], 512) : go(e8p_prefill_gemm_kernel<4, 2, 2, 4>, c[3], 512);
  return half ? go(e8p_prefill_gemm_kernel<4, 1, 1, 8>, c[4], 512) : go(e8p_prefill_gemm_kernel<8, 1, 1, 8>, c[5], 512);
}

int e8p_prefill_gemm_launch(const void* x, const void* qidxs, const void* grid, void* y, int64_t m, int n, int k,
                            hipStream_t stream) {
  return prefill_launch_mode(0, x, qidxs, grid, nullptr, 0.f, y, m, n, k, stream);
}

int e8prvq4_prefill_gemm_launch(const void* x, const void* qidxs, const void* grid, float resid_scale, void* y, int64_t m, int n,
                                int k, hipStream_t stream) {
  return prefill_launch_mode(1, x, qidxs, grid, nullptr, resid_scale, y, m, n, k, stream);
}

int e8prvq3_prefill_gemm_launch(const void* x, const void* qidxs, const void* grid, const void* e81b_packed, float resid_scale,
                                void* y, int64_t m, int n, int k, hipStream_t stream) {
  return prefill_launch_mode(4, x, qidxs, grid, e81b_packed, resid_scale, y, m, n, k, stream);
}

int d4_prefill_gemm_launch(const void* x, const void* qidxs, const void* grid_f16, void* y, int64_t m, int n, int k,
                           hipStream_t stream) {
  return prefill_launch_mode(2, x, qidxs, grid_f16, nullptr, 0.f, y, m, n, k, stream);
}

int hi_prefill_gemm_launch(const void* x, const void* qidxs, void* y, int64_t m, int n, int k, hipStream_t stream) {
  return prefill_launch_mode(3, x, qidxs, nullptr, nullptr, 0.f, y, m, n, k, stream);
}

}  // namespace quip
