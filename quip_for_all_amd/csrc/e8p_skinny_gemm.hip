// Skinny E8P12 product for 2 <= M <= 32 activation rows in ONE pass over the codes, gfx950.
//
//   Y[m, n] = sum_k X[m, k] * W[n, k],   W = decode(Qidxs (N, K/8) int16),  fp16 x fp16 -> fp32 -> fp16
//
// Replaces the 1 < M < 32 use of tinygemm_m16n8k16_chunk_kernel<.., BLayout_E8, ..> (origin_order.cu:388-555;
// e8p12.py:147-150) with the reference's own arithmetic: fp16 activations, exact fp16 weights, fp32 accumulation on
// the matrix cores, one fp16 rounding of the result.
//
// Why a second skinny path next to rows mode (e8p_gemv_mfma.hip): rows mode keeps the GEMV's exact integer
// arithmetic (every row bit identical to its bs = 1 result) but carries at most 5 activation rows per pass -- a
// workgroup has to hold the digit planes of every row it multiplies, 3 K bytes each -- so M = 16 costs 4 passes and
// M = 31 seven.  Here the activations stay fp16, as in the reference: v_mfma_f32_32x32x16_f16 takes 32 activation rows
// at once, one E8P code per lane is a complete B fragment (8 consecutive-k weights of one column, decoded by two LDS
// table lookups + XOR + the 0x5c00 | (4w + 128) fp16 identity, as in e8p_prefill_gemm.hip), and the codes are
// streamed exactly once.  What is given up is bit identity with the bs = 1 path (fp32 accumulation instead of exact
// integers); a row's result is still independent of the batch it is in, and the exact rows-mode path stays selectable
// (QuantLinear.skinny_exact / QUIP_SKINNY_EXACT=1).
//
// Mapping: workgroup = 16 waves = CB column blocks of 32 x (16 / CB) slices of K (CB = 1 when the matrix has few
// columns: more workgroups and shorter dependent chains per wave).  A wave walks its K slice in units of 128 k =
// four granules of 32 k = two MFMAs each.
//   codes: one 16-byte load per lane and unit (lane (n, kb): the unit's 8-k blocks 8 kb .. 8 kb + 7 of column n),
//     then two v_permlane32_swap exchange code pairs between the kb halves so that in granule i MFMA t multiplies
//     block 4 i + 2 kb + t -- the MFMA's k index is relabelled, activations follow the same labelling;
//   activations: an MFMA A fragment is 32 rows x 32 bytes, i.e. 64 lanes reading 64 different rows/segments.  Read
//     straight from L2 that costs one vector-memory lane slot each (measured: time grew by 0.25 us per activation row
//     at 4096 x 4096, 7.4 us from M = 1 to M = 31, whatever the row stride).  So a granule is brought in ROW
//     CONTIGUOUS -- one global_load_lds_dwordx4 = 16 rows x 64 bytes, four lanes per row, no staging registers --
//     into a ring of three granules per wave, and the fragments are read back from LDS (ds_read_b128; the 16-byte
//     pieces of a row are stored in the order p ^ ((row >> 1) & 3), applied on the SOURCE side because the LDS image
//     of such a load is lane-linear: 8 consecutive rows then fall into 8 different bank groups).
//   Rows >= M are not fetched (the lanes re-read row M - 1); a launch for M <= 16 stages 16 rows.
// All vector-memory traffic of the loop is counted by hand (vmcnt retires in order): per unit the issue order is
// s(4u+3)' = [codes of unit u + 2, granule 4u + 3], granule 4u + 4, 4u + 5, 4u + 6, each right after the granule
// three earlier has been consumed; past the end of the slice the last unit / granule is requested again, which keeps
// the counts constant.  The K slices are added in a fixed order through LDS (every wave reduces a share of the
// tile), so the result does not depend on scheduling.
// Bound: LDS / L2 issue (codes N K / 4 bytes once from HBM; activations M K 2 bytes per 32 columns from L2).
//
// MODE 1 = E8P12RVQ4B (e8p12_rvq4.py:37-45; origin_order.cu:337-385): a code is 32 bits, main << 16 | residual, both E8P12
// codes; the weight is fma(s, w_resid, w_main) in packed fp16 -- ONE fp16 rounding per weight, exactly what
// decompress_e8prvq4_origorder writes (quip_device.hip.h: rvq_combine), so the product is x . W of the reference's dense W.
// Per lane and unit two 16-byte loads (eight 4-byte codes), four table lookups and four v_pk_fma_f16 per MFMA.
// MODE 4 = E8P12RVQ3B (e8p12_rvq3.py:81-129; origin_order.cu:287-335): 3-byte codes [residual index, E8P code]; two 12-byte
// loads per lane and unit, a shift and two v_perm_b32 turn them into the dwords (main << 16 | residual << 8) of MODE 1's
// data movement; the residual is one 4-byte lookup (eight int4 = 2 x value, the reference's packed E81B table) turned into
// fp16 by 0x4c00 | (n ^ 8) << 6 = 16 + (n ^ 8) and one exact fma (x 0.5, - 12), then fma(s, w_resid, w_main) as in MODE 1.
// (Table space: the sign table keeps 8 copies instead of 16, the 16 KB hold 16 copies of the residual table.)
// MODE 2 = D4 (d4.py:26-96; origin_order.cu BLayout_D4): two one-byte codes per 8 weights -- the data movement of MODE 0 --
// and the table holds the fp16 entries themselves: a B fragment is two 8-byte lookups, no arithmetic.
// MODE 3 = HI (hi.py:41-50; origin_order.cu:1028-1051): eight nibbles per 8 weights -- the data movement of MODE 1 --,
// w = nibble - 7.5 through the 0x4c00 | n << 6 = 16 + n identity and one packed subtraction; no table.
#include "quip_device.hip.h"
#include "quip_internal.h"
#include <type_traits>

namespace quip {

namespace {

typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));
typedef float sf32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t su32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t su32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t su32x3 __attribute__((ext_vector_type(3)));

constexpr int kSRep = 16;
constexpr int kST1 = 0;
constexpr int kST2 = 256 * kSRep * 8;        // 32 KiB
constexpr int kSRed = 2 * kST2;              // 64 KiB; behind the tables: the granule rings (16 waves x 3 x 2 KiB), later
                                             // the reduction (16 waves x 16 registers x 64 lanes fp32 = 64 KiB)
constexpr int kSLds = kSRed + 16 * 3 * 2048;      // 160 KiB

struct ST2Image {
  uint2 v[256];
  constexpr ST2Image() : v{} {
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
__device__ const ST2Image kST2Img{};

__device__ __forceinline__ uint2 s_lds_read8(uint32_t addr) {
  const su32x2 v = *reinterpret_cast<const __attribute__((address_space(3))) su32x2*>((uintptr_t)addr);
  return make_uint2(v.x, v.y);
}

__device__ __forceinline__ void s_bytes_to_f16x4(uint32_t u4, uint32_t& lo, uint32_t& hi) {
  const uint32_t k5c = 0x5c5c5c5cu;
  const uint32_t a = __builtin_amdgcn_perm(u4, k5c, 0x00050004u);   // [u0, 5c, u1, 5c]
  const uint32_t b = __builtin_amdgcn_perm(u4, k5c, 0x00070006u);   // [u2, 5c, u3, 5c]
  const f16x2 m288 = {(f16)-288.f, (f16)-288.f};
  lo = as_u32(as_f16x2(a) + m288);
  hi = as_u32(as_f16x2(b) + m288);
}

constexpr int kSDepth = 3;   // granules in flight per wave

// CB column blocks of 32 per workgroup, 16 / CB slices of K (16 waves); MP = activation rows staged (16 or 32)
template <int CB, int MP, int MODE = 0>
__global__ __launch_bounds__(1024) void e8p_skinny_gemm_kernel(const f16* __restrict__ X,
                                                               const uint16_t* __restrict__ Wc,
                                                               const uint64_t* __restrict__ grid,
                                                               f16* __restrict__ Y, int M, int N, int K, float resid_scale,
                                                               const uint32_t* __restrict__ grid2) {
  constexpr int NCL = (MODE == 1 || MODE == 3 || MODE == 4) ? 2 : 1;     // code loads per lane and unit (16 bytes; MODE 4: 12)
  constexpr int kRep2 = MODE == 4 ? 8 : kSRep;                           // copies of the sign table
  constexpr int kST3 = kST2 + 256 * 8 * 8;                               // MODE 4: the residual table (256 x 4 bytes x 16 copies)
  using CodeT = std::conditional_t<MODE == 4, su32x3, su32x4>;           // what one code load writes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // more than 32 rows: grid.y walks over chunks of 32 rows (each workgroup streams its columns' codes again; the
  // chunks of one column block run side by side and share them in L2)
  X += (size_t)blockIdx.y * 32 * K;
  Y += (size_t)blockIdx.y * 32 * N;
  M = min(32, M - (int)blockIdx.y * 32);
  constexpr int NKS = 16 / CB;
  constexpr int L = MP / 16;                 // load instructions per granule
  constexpr int kGran = MP * 64;             // bytes of a granule: MP rows x 32 k
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave % CB, ks = wave / CB;
  const int n = lane & 31, kb = lane >> 5;
  const int ncol = (int)blockIdx.x * (32 * CB) + cb * 32 + n;
  // K slice of this wave in units of 128 k: the first (units % NKS) slices get one unit more
  const int units = K >> 7;
  const int ubase = units / NKS, uextra = units - ubase * NKS;
  const int u0 = ks * ubase + min(ks, uextra), un = ubase + (ks < uextra ? 1 : 0);
  const int ulast = max(un - 1, 0), qlast = max(4 * un - 1, 0);

  // (a wave without units -- K < 2048 -- still issues the loads of the prologue: inside its row)
  // a unit's 16 codes of a column: 32 bytes (MODE 0), 64 bytes (MODE 1); lane kb takes the half with blocks 8 kb .. 8 kb + 7
  // (MODE 4: a unit's 16 codes are 48 bytes; lane kb takes bytes [24 kb, 24 kb + 24) as two 12-byte loads)
  const char* wsrc_b = reinterpret_cast<const char*>(Wc) +
                       (MODE == 4 ? (size_t)min(ncol, N - 1) * (size_t)(K >> 3) * 3 + (size_t)48 * min(u0, units - 1) + 24 * kb
                                  : ((size_t)min(ncol, N - 1) * (K >> 3) * 2 * NCL + (size_t)32 * NCL * min(u0, units - 1) + 16 * NCL * kb));
  auto load_codes = [&](CodeT (&dst)[NCL], int u) {
#pragma unroll
    for (int c = 0; c < NCL; ++c) {
      if constexpr (MODE == 4)
        asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(dst[c]) : "v"(wsrc_b + 48 * u + 12 * c) : "memory");
      else
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[c]) : "v"(wsrc_b + 32 * NCL * u + 16 * c) : "memory");
    }
  };
  // granule loader: instruction h fills LDS slots [64 h, 64 h + 64) of the granule; slot s = (row s >> 2, stored
  // piece s & 3) holds source piece (s & 3) ^ ((row >> 1) & 3) of that row's 64 bytes
  const f16* xsrc[L];
#pragma unroll
  for (int h = 0; h < L; ++h) {
    const int row = 16 * h + (lane >> 2);
    const int p = (lane & 3) ^ ((row >> 1) & 3);
    xsrc[h] = X + (size_t)min(row, M - 1) * K + (size_t)min(u0, units - 1) * 128 + p * 8;
  }
  const uint32_t ring = (uint32_t)kSRed + (uint32_t)wave * (kSDepth * kGran);
  auto issue_x = [&](int q, int buf) {
#pragma unroll
    for (int h = 0; h < L; ++h)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*)(xsrc[h] + (size_t)q * 32),
          (__attribute__((address_space(3))) void*)(smem + ring + buf * kGran + h * 1024), 16, 0, 0);
  };

  // request order: the table source first (every wave asks, waves 0..7 use it: a load and its wait on the same
  // unconditional path keep the counting simple), then the codes of the first two units -- the table build below
  // waits for its source only, the codes stay in flight across it
  const bool second = (lane & 32) != 0;
  const int e = (wave & 7) * 32 + (lane & 31);
  su32x2 rawv;
  {
    const uint2* tsrc = (second || MODE == 3) ? &kST2Img.v[e] : reinterpret_cast<const uint2*>(grid) + e;   // (HI: no table, `grid` is not read)
    if (MODE == 4 && wave >= 8) tsrc = reinterpret_cast<const uint2*>(grid2 + (e & ~1));            // residual table: entries e & ~1, e | 1
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rawv) : "v"(tsrc) : "memory");
  }
  // A register that a load is still going to write must never be a TIED asm operand ("+v") or cross a loop edge:
  // for either the compiler may emit a copy of it -- before the wait.  Loaded values are taken over by an asm that
  // waits and then moves them into fresh registers; only those are used afterwards.  (tools/check_inflight.py walks
  // the ISA for exactly this; tests/test_build_invariants.py runs it.)
  CodeT f0[NCL], f1[NCL], c2[NCL];
  load_codes(f0, 0);
  load_codes(f1, min(1, ulast));
  auto take = [](CodeT (&dst)[NCL], const CodeT (&src)[NCL], auto nw) {
#pragma unroll
    for (int c = 0; c < NCL; ++c) {
      if constexpr (MODE == 4)
        asm volatile("s_waitcnt vmcnt(%6)\n\tv_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %5"
                     : "=&v"(dst[c].x), "=&v"(dst[c].y), "=&v"(dst[c].z)
                     : "v"(src[c].x), "v"(src[c].y), "v"(src[c].z), "n"(decltype(nw)::value)
                     : "memory");
      else
        asm volatile("s_waitcnt vmcnt(%8)\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                     : "=&v"(dst[c].x), "=&v"(dst[c].y), "=&v"(dst[c].z), "=&v"(dst[c].w)
                     : "v"(src[c].x), "v"(src[c].y), "v"(src[c].z), "v"(src[c].w), "n"(decltype(nw)::value)
                     : "memory");
    }
  };
  // tables: T1' = (4a | 1) ^ 0x80.., T2 = sign masks; 16 copies each (waves 0..7)
  uint2 raw;
  asm volatile("s_waitcnt vmcnt(%4)\n\tv_mov_b32 %0, %2\n\tv_mov_b32 %1, %3"
               : "=&v"(raw.x), "=&v"(raw.y)
               : "v"(rawv.x), "v"(rawv.y), "n"(2 * NCL)
               : "memory");
  if (wave < 8) {
    const uint32_t t1x = (__builtin_amdgcn_perm(0u, raw.x, 0x03010200u) | 0x01010101u) ^ 0x80808080u;
    const uint32_t t1y = (__builtin_amdgcn_perm(0u, raw.y, 0x03010200u) | 0x01010101u) ^ 0x80808080u;
    const su32x2 val = {(second || MODE == 2) ? raw.x : t1x, (second || MODE == 2) ? raw.y : t1y};   // (D4: the fp16 entries as they are)
    const uint32_t rowbase = (second ? (uint32_t)kST2 + (uint32_t)e * (kRep2 * 8) : (uint32_t)kST1 + (uint32_t)e * (kSRep * 8));
#pragma unroll
    for (int c = 0; c < kSRep; ++c) {
      const uint32_t copy = (uint32_t)(lane + c) & (uint32_t)((second ? kRep2 : kSRep) - 1);
      if (!second || c < kRep2)
        *reinterpret_cast<__attribute__((address_space(3))) su32x2*>((uintptr_t)(rowbase + copy * 8)) = val;
    }
  } else if (MODE == 4 && !second) {
    // waves 8..15, lanes 0..31: residual entry e = 32 (wave & 7) + lane, 16 copies of 4 bytes
    const uint32_t val = (e & 1) ? raw.y : raw.x;
    const uint32_t rowbase = (uint32_t)kST3 + (uint32_t)e * (kSRep * 4);
#pragma unroll
    for (int c = 0; c < kSRep; ++c)
      *reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)(rowbase + (((uint32_t)(lane + c) & (kSRep - 1)) << 2))) = val;
  }
  __builtin_amdgcn_sched_barrier(0);
  // (a wave without units requests its three granules like the others: clamped addresses, never read)
#pragma unroll
  for (int q = 0; q < kSDepth; ++q) issue_x(min(q, qlast), q);
  __builtin_amdgcn_sched_barrier(0);
  using std::integral_constant;
  // the codes of units 0 and 1 have arrived (the granules were requested after them); tables written
  CodeT c0[NCL], c1[NCL];
  take(c0, f0, integral_constant<int, kSDepth * L>{});
  take(c1, f1, integral_constant<int, kSDepth * L>{});
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const uint32_t lane_c1 = (uint32_t)(lane & 15) << 3;
  const uint32_t lane_c2 = (MODE == 4 ? (uint32_t)(lane & 7) << 3 : lane_c1) | (uint32_t)kST2;
  const uint32_t lane_c3 = ((uint32_t)(lane & 15) << 2) | (uint32_t)kST3;
  // A fragment of MFMA t of a granule: row lane & 31, piece 2 kb + t
  const int arow = n & (MP - 1);
  const uint32_t rd0 = ring + (uint32_t)(arow * 64 + (((2 * kb + 0) ^ ((arow >> 1) & 3)) << 4));
  const uint32_t rd1 = ring + (uint32_t)(arow * 64 + (((2 * kb + 1) ^ ((arow >> 1) & 3)) << 4));

  sf32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // one granule: wait for it (NW = vector-memory operations issued after it), two MFMAs from code pair d
  auto granule = [&](uint32_t d, uint32_t boff, auto nw) {
    constexpr int NW = decltype(nw)::value;
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NW) : "memory");
    su32x4 A0, A1;
    su32x2 t1a, t2a, t1b, t2b;
    const uint32_t a1 = ((d >> 1) & 0x7f80u) | lane_c1, a2 = ((d << 7) & 0x7f80u) | lane_c2;
    const uint32_t b1 = ((d >> 17) & 0x7f80u) | lane_c1, b2 = ((d >> 9) & 0x7f80u) | lane_c2;
    asm volatile("ds_read_b64 %0, %1" : "=v"(t1a) : "v"(a1));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t2a) : "v"(a2));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A0) : "v"(rd0 + boff));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t1b) : "v"(b1));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t2b) : "v"(b2));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A1) : "v"(rd1 + boff));
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(t1a), "+v"(t2a), "+v"(A0));
    uint32_t w0, w1, w2, w3;
    s_bytes_to_f16x4(t1a.x ^ t2a.x, w0, w1);
    s_bytes_to_f16x4(t1a.y ^ t2a.y, w2, w3);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A0),
                                                 __builtin_bit_cast(sf16x8, su32x4{w0, w1, w2, w3}), acc, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t1b), "+v"(t2b), "+v"(A1));
    s_bytes_to_f16x4(t1b.x ^ t2b.x, w0, w1);
    s_bytes_to_f16x4(t1b.y ^ t2b.y, w2, w3);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A1),
                                                 __builtin_bit_cast(sf16x8, su32x4{w0, w1, w2, w3}), acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // MODE 1: two 32-bit codes (main << 16 | residual) of this lane's two MFMAs
  const f16 rs16 = (f16)resid_scale;
  const f16x2 rs2 = {rs16, rs16};
  auto granule_rvq = [&](uint32_t dA, uint32_t dB, uint32_t boff, auto nw) {
    constexpr int NW = decltype(nw)::value;
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NW) : "memory");
    su32x4 A0, A1;
    su32x2 t[8];
    auto addr = [&](uint32_t d, uint32_t (&a)[4]) {
      a[0] = ((d >> 17) & 0x7f80u) | lane_c1;     // main (high half): abs index, sign byte
      a[1] = ((d >> 9) & 0x7f80u) | lane_c2;
      a[2] = ((d >> 1) & 0x7f80u) | lane_c1;      // residual (low half)
      a[3] = ((d << 7) & 0x7f80u) | lane_c2;
    };
    uint32_t aa[4], ab[4];
    addr(dA, aa);
    addr(dB, ab);
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(t[i]) : "v"(aa[i]));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A0) : "v"(rd0 + boff));
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("ds_read_b64 %0, %1" : "=v"(t[4 + i]) : "v"(ab[i]));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A1) : "v"(rd1 + boff));
    auto weights = [&](const su32x2& m1, const su32x2& m2, const su32x2& r1, const su32x2& r2) -> su32x4 {
      uint32_t m[4], r[4], w[4];
      s_bytes_to_f16x4(m1.x ^ m2.x, m[0], m[1]);
      s_bytes_to_f16x4(m1.y ^ m2.y, m[2], m[3]);
      s_bytes_to_f16x4(r1.x ^ r2.x, r[0], r[1]);
      s_bytes_to_f16x4(r1.y ^ r2.y, r[2], r[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) w[i] = as_u32(__builtin_elementwise_fma(rs2, as_f16x2(r[i]), as_f16x2(m[i])));
      return su32x4{w[0], w[1], w[2], w[3]};
    };
    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(A0));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A0),
                                                 __builtin_bit_cast(sf16x8, weights(t[0], t[1], t[2], t[3])), acc, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(A1));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A1),
                                                 __builtin_bit_cast(sf16x8, weights(t[4], t[5], t[6], t[7])), acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // MODE 4: dwords (main << 16 | residual index << 8) of this lane's two MFMAs
  auto granule_rvq3 = [&](uint32_t dA, uint32_t dB, uint32_t boff, auto nw) {
    constexpr int NW = decltype(nw)::value;
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NW) : "memory");
    su32x4 A0, A1;
    su32x2 t[4];
    uint32_t r[2];
    auto addr = [&](uint32_t d, uint32_t (&a)[3]) {
      a[0] = ((d >> 17) & 0x7f80u) | lane_c1;     // main: abs index (16 copies x 8 bytes), sign byte (8 copies)
      a[1] = ((d >> 10) & 0x3fc0u) | lane_c2;
      a[2] = ((d >> 2) & 0x3fc0u) | lane_c3;      // residual index (16 copies x 4 bytes)
    };
    uint32_t aa[3], ab[3];
    addr(dA, aa);
    addr(dB, ab);
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[0]) : "v"(aa[0]));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[1]) : "v"(aa[1]));
    asm volatile("ds_read_b32 %0, %1" : "=v"(r[0]) : "v"(aa[2]));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A0) : "v"(rd0 + boff));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[2]) : "v"(ab[0]));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[3]) : "v"(ab[1]));
    asm volatile("ds_read_b32 %0, %1" : "=v"(r[1]) : "v"(ab[2]));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A1) : "v"(rd1 + boff));
    auto weights = [&](const su32x2& m1, const su32x2& m2, uint32_t c) -> su32x4 {
      uint32_t m[4], w[4];
      s_bytes_to_f16x4(m1.x ^ m2.x, m[0], m[1]);
      s_bytes_to_f16x4(m1.y ^ m2.y, m[2], m[3]);
      const f16x2 half2 = {(f16)0.5f, (f16)0.5f}, m12 = {(f16)-12.f, (f16)-12.f};
      const uint32_t sh[4] = {c << 6, c << 2, c >> 2, c >> 6};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // nibble n = int4 of 2 x value: 0x4c00 | (n ^ 8) << 6 is 16 + (n ^ 8); x 0.5 - 12 = 0.5 ((n ^ 8) - 8), exact
        const f16x2 res = __builtin_elementwise_fma(as_f16x2((sh[i] & 0x03c003c0u) ^ 0x4e004e00u), half2, m12);
        w[i] = as_u32(__builtin_elementwise_fma(rs2, res, as_f16x2(m[i])));
      }
      return su32x4{w[0], w[1], w[2], w[3]};
    };
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(t[0]), "+v"(t[1]), "+v"(r[0]), "+v"(A0));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A0),
                                                 __builtin_bit_cast(sf16x8, weights(t[0], t[1], r[0])), acc, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[2]), "+v"(t[3]), "+v"(r[1]), "+v"(A1));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A1),
                                                 __builtin_bit_cast(sf16x8, weights(t[2], t[3], r[1])), acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // MODE 2: dword d = the four D4 code bytes of this lane's two MFMAs (two per MFMA: weights 0..3, 4..7 of the block)
  auto granule_d4 = [&](uint32_t d, uint32_t boff, auto nw) {
    constexpr int NW = decltype(nw)::value;
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NW) : "memory");
    su32x4 A0, A1;
    su32x2 t[4];
    const uint32_t a0 = ((d << 7) & 0x7f80u) | lane_c1, a1 = ((d >> 1) & 0x7f80u) | lane_c1;
    const uint32_t a2 = ((d >> 9) & 0x7f80u) | lane_c1, a3 = ((d >> 17) & 0x7f80u) | lane_c1;
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[0]) : "v"(a0));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[1]) : "v"(a1));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A0) : "v"(rd0 + boff));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[2]) : "v"(a2));
    asm volatile("ds_read_b64 %0, %1" : "=v"(t[3]) : "v"(a3));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A1) : "v"(rd1 + boff));
    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(t[0]), "+v"(t[1]), "+v"(A0));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A0),
                                                 __builtin_bit_cast(sf16x8, su32x4{t[0].x, t[0].y, t[1].x, t[1].y}), acc, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[2]), "+v"(t[3]), "+v"(A1));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A1),
                                                 __builtin_bit_cast(sf16x8, su32x4{t[2].x, t[2].y, t[3].x, t[3].y}), acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // MODE 3: two 32-bit HI codes; nibble i of a code is column [0,2,4,6,1,3,5,7][i] of its 8-group, i.e. the fp16 pair d of
  // the fragment is (nibble d, nibble d + 4) (quip_device.hip.h: hi_decode_f16)
  auto granule_hi = [&](uint32_t dA, uint32_t dB, uint32_t boff, auto nw) {
    constexpr int NW = decltype(nw)::value;
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NW) : "memory");
    su32x4 A0, A1;
    asm volatile("ds_read_b128 %0, %1" : "=v"(A0) : "v"(rd0 + boff));
    asm volatile("ds_read_b128 %0, %1" : "=v"(A1) : "v"(rd1 + boff));
    auto weights = [&](uint32_t c) -> su32x4 {
      // 0x4c00 | n << 6 is the fp16 number 16 + n (the mantissa bit of weight 1.0 at exponent 4); 23.5 is a fp16 number too,
      // so the subtraction is exact (1024 + n would not do: 1031.5 is not representable)
      const f16x2 off = {(f16)-23.5f, (f16)-23.5f};
      const uint32_t sh[4] = {c << 6, c << 2, c >> 2, c >> 6};
      uint32_t w[4];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) w[dd] = as_u32(as_f16x2((sh[dd] & 0x03c003c0u) | 0x4c004c00u) + off);
      return su32x4{w[0], w[1], w[2], w[3]};
    };
    const su32x4 B0 = weights(dA), B1 = weights(dB);
    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(A0));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A0), __builtin_bit_cast(sf16x8, B0), acc, 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A1));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, A1), __builtin_bit_cast(sf16x8, B1), acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  int rb = 0;                                   // ring buffer of the current granule
  auto next_rb = [&]() { rb = rb == kSDepth - 1 ? 0 : rb + 1; };
  for (int u = 0; u < un; ++u) {
    // the kb halves trade code pairs: afterwards c0.x / .z / .y / .w serve granules 0 / 1 / 2 / 3 (MODE 1: the dword
    // pairs (c0[0].x, .y) / (c0[1].x, .y) / (c0[0].z, .w) / (c0[1].z, .w))
    // (builtin, not asm: v_permlane32_swap has wait-state requirements against neighbouring VALU instructions that
    //  only the compiler's hazard recogniser keeps track of)
    uint32_t gA[4], gB[4];      // per granule: the code dword(s) of this lane
    if constexpr (MODE == 0 || MODE == 2) {
      const auto s01 = __builtin_amdgcn_permlane32_swap(c0[0].x, c0[0].y, false, false);
      const auto s23 = __builtin_amdgcn_permlane32_swap(c0[0].z, c0[0].w, false, false);
      gA[0] = s01[0]; gA[2] = s01[1]; gA[1] = s23[0]; gA[3] = s23[1];
      gB[0] = gB[1] = gB[2] = gB[3] = 0u;
    } else {
      su32x4 cd[2];
      if constexpr (MODE == 4) {
        // 12 landed bytes = four 3-byte codes [residual, e8p lo, e8p hi] -> dwords (main << 16 | residual << 8)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          cd[h] = su32x4{c0[h].x << 8, __builtin_amdgcn_perm(c0[h].y, c0[h].x, 0x0504030cu),
                         __builtin_amdgcn_perm(c0[h].z, c0[h].y, 0x0403020cu), c0[h].z & 0xffffff00u};
      } else {
        cd[0] = su32x4{c0[0].x, c0[0].y, c0[0].z, c0[0][3]};
        cd[1] = su32x4{c0[1].x, c0[1].y, c0[1].z, c0[1][3]};
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const auto sx = __builtin_amdgcn_permlane32_swap(cd[h].x, cd[h].z, false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(cd[h].y, cd[h].w, false, false);
        gA[h] = sx[0]; gB[h] = sy[0];           // granule h: blocks 4 h + 2 kb, + 1
        gA[2 + h] = sx[1]; gB[2 + h] = sy[1];   // granule 2 + h
      }
    }
    auto gran = [&](int i, auto nw) {
      if constexpr (MODE == 0) granule(gA[i], (uint32_t)(rb * kGran), nw);
      else if constexpr (MODE == 1) granule_rvq(gA[i], gB[i], (uint32_t)(rb * kGran), nw);
      else if constexpr (MODE == 2) granule_d4(gA[i], (uint32_t)(rb * kGran), nw);
      else if constexpr (MODE == 4) granule_rvq3(gA[i], gB[i], (uint32_t)(rb * kGran), nw);
      else granule_hi(gA[i], gB[i], (uint32_t)(rb * kGran), nw);
    };
    gran(0, integral_constant<int, 2 * L>{});
    load_codes(c2, min(u + 2, ulast));
    issue_x(min(4 * u + 3, qlast), rb);
    __builtin_amdgcn_sched_barrier(0);
    next_rb();
    gran(1, integral_constant<int, 2 * L + NCL>{});
    issue_x(min(4 * u + 4, qlast), rb);
    __builtin_amdgcn_sched_barrier(0);
    next_rb();
    gran(2, integral_constant<int, 2 * L + NCL>{});
    issue_x(min(4 * u + 5, qlast), rb);
    __builtin_amdgcn_sched_barrier(0);
    next_rb();
    gran(3, integral_constant<int, 2 * L>{});
    issue_x(min(4 * u + 6, qlast), rb);
    __builtin_amdgcn_sched_barrier(0);
    next_rb();
    // the codes of unit u + 2 were requested before granule 4u + 3, which has just been waited for
#pragma unroll
    for (int c = 0; c < NCL; ++c) c0[c] = c1[c];
    take(c1, c2, integral_constant<int, kSDepth * L>{});
  }
  // the trailing filler granules must have landed before the ring is reused for the reduction
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // the K slices of a column block, added in the order 0, 1, 2, ...: every wave reduces CB of the 16 accumulator
  // registers of one column block (deterministic, independent of scheduling)
  float* red = reinterpret_cast<float*>(smem + kSRed);
#pragma unroll
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  {
    const int ocb = wave % CB, r0 = (wave / CB) * CB;      // output column block, first register of this wave
    const int ocol = (int)blockIdx.x * (32 * CB) + ocb * 32 + n;
#pragma unroll
    for (int i = 0; i < CB; ++i) {
      const int r = r0 + i;
      float v = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < NKS; ++s2) v += red[((s2 * CB + ocb) * 16 + r) * 64 + lane];
      const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
      if (row < M && ocol < N) Y[(size_t)row * N + ocol] = (f16)v;
    }
  }
}

}  // namespace

bool e8p_skinny_gemm_supported(int m, int n, int k) {
  return m >= 1 && m <= 32 * 65535 && n >= 2 && n % 2 == 0 && k >= 128 && k % 128 == 0;
}

// mode 0: E8P12 (16-bit codes), 1: E8P12RVQ4B (32-bit codes, resid_scale = the fp16 residual scale), 2: D4 (grid = the fp16
// (256, 4) table), 3: HI (no table)
static int skinny_launch_mode(int mode, const void* x, const void* qidxs, const void* grid, float resid_scale, void* y, int m,
                              int n, int k, hipStream_t stream, const void* grid2 = nullptr) {
  if (!e8p_skinny_gemm_supported(m, n, k)) return QUIP_ERR_UNSUPPORTED;
  // few columns: one column block of 32 per workgroup and 16 slices of K (more workgroups, shorter chains per wave);
  // many columns: two column blocks x 8 slices
  // (a function of n alone: the K slicing fixes the order of the fp32 sums, and a row's result must not depend on
  //  how many rows the launch has)
  const bool one = (n + 63) / 64 < 2 * device_cu_count() / 3;
  auto go = [&](auto kern, int cols, int slot) -> int {
    static DynLdsCache configured[20];   // per instantiation, per device
    if (ensure_dyn_lds(configured[slot], reinterpret_cast<const void*>(kern), kSLds) != QUIP_OK) return QUIP_ERR_LAUNCH;
    hipLaunchKernelGGL(kern, dim3((n + cols - 1) / cols, (m + 31) / 32), dim3(1024), kSLds, stream,
                       reinterpret_cast<const f16*>(x),
                       reinterpret_cast<const uint16_t*>(qidxs), reinterpret_cast<const uint64_t*>(grid),
                       reinterpret_cast<f16*>(y), m, n, k, resid_scale, reinterpret_cast<const uint32_t*>(grid2));
    return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
  };
  if (mode == 1) {
    if (m <= 16) return one ? go(e8p_skinny_gemm_kernel<1, 16, 1>, 32, 4) : go(e8p_skinny_gemm_kernel<2, 16, 1>, 64, 5);
    return one ? go(e8p_skinny_gemm_kernel<1, 32, 1>, 32, 6) : go(e8p_skinny_gemm_kernel<2, 32, 1>, 64, 7);
  }
  if (mode == 2) {
    if (m <= 16) return one ? go(e8p_skinny_gemm_kernel<1, 16, 2>, 32, 8) : go(e8p_skinny_gemm_kernel<2, 16, 2>, 64, 9);
    return one ? go(e8p_skinny_gemm_kernel<1, 32, 2>, 32, 10) : go(e8p_skinny_gemm_kernel<2, 32, 2>, 64, 11);
  }
  if (mode == 4) {
    if (m <= 16) return one ? go(e8p_skinny_gemm_kernel<1, 16, 4>, 32, 16) : go(e8p_skinny_gemm_kernel<2, 16, 4>, 64, 17);
    return one ? go(e8p_skinny_gemm_kernel<1, 32, 4>, 32, 18) : go(e8p_skinny_gemm_kernel<2, 32, 4>, 64, 19);
  }
  if (mode == 3) {
    if (m <= 16) return one ? go(e8p_skinny_gemm_kernel<1, 16, 3>, 32, 12) : go(e8p_skinny_gemm_kernel<2, 16, 3>, 64, 13);
    return one ? go(e8p_skinny_gemm_kernel<1, 32, 3>, 32, 14) : go(e8p_skinny_gemm_kernel<2, 32, 3>, 64, 15);
  }
  if (m <= 16) return one ? go(e8p_skinny_gemm_kernel<1, 16>, 32, 0) : go(e8p_skinny_gemm_kernel<2, 16>, 64, 1);
  return one ? go(e8p_skinny_gemm_kernel<1, 32>, 32, 2) : go(e8p_skinny_gemm_kernel<2, 32>, 64, 3);
}

int e8p_skinny_gemm_launch(const void* x, const void* qidxs, const void* grid, void* y, int m, int n, int k,
                           hipStream_t stream) {
  return skinny_launch_mode(0, x, qidxs, grid, 0.f, y, m, n, k, stream);
}

int e8prvq4_skinny_gemm_launch(const void* x, const void* qidxs, const void* grid, float resid_scale, void* y, int m, int n,
                               int k, hipStream_t stream) {
  return skinny_launch_mode(1, x, qidxs, grid, resid_scale, y, m, n, k, stream);
}

int e8prvq3_skinny_gemm_launch(const void* x, const void* qidxs, const void* grid, const void* e81b_packed, float resid_scale,
                               void* y, int m, int n, int k, hipStream_t stream) {
  if (k % 32 != 0) return QUIP_ERR_UNSUPPORTED;
  return skinny_launch_mode(4, x, qidxs, grid, resid_scale, y, m, n, k, stream, e81b_packed);
}

int d4_skinny_gemm_launch(const void* x, const void* qidxs, const void* grid_f16, void* y, int m, int n, int k, hipStream_t stream) {
  return skinny_launch_mode(2, x, qidxs, grid_f16, 0.f, y, m, n, k, stream);
}

int hi_skinny_gemm_launch(const void* x, const void* qidxs, void* y, int m, int n, int k, hipStream_t stream) {
  return skinny_launch_mode(3, x, qidxs, nullptr, 0.f, y, m, n, k, stream);
}

}  // namespace quip
