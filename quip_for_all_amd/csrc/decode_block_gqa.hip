// Persistent decode engine for gfx950, stage 2 for the grouped-query 8192-wide shape (Llama-2-70B: hidden 8192, 64 heads of
// 128 on 8 KV heads, n_ffn = 7 x 4096): all decoder blocks of a token (bs = 1) in ONE launch.  Same contract as
// decode_block.hip (the HF LlamaDecoderLayer of the reference's metric driver, example_generate.py:28-33, every projection a
// QuantLinear, qlinear.py:87-115; reference kernel for the products: origin_order.cu:388-555 at m = 1), a different
// machine inside, because a block of this shape streams 214 MB of codes (836 KB per CU) instead of 51 MB:
//
//   * ONE weight stream per wave for the whole launch: the 54 items (16 rows x 512 k = 2 KB of codes) a wave multiplies per
//     block form a fixed sequence -- o (4), gate / up (28), down (14), 2 fillers, the NEXT block's q (4) and k | v (2) --
//     that runs through a ring of 9 register slots.  Consuming item s requests item s + 9 into the same slot, so 9 items
//     (18 KB per wave, 147 KB per CU) are always requested ahead, across products, hand-offs and blocks, and "slot s has
//     landed" is the constant `s_waitcnt vmcnt(16)`.  The queue is empty after every hand-off's gather (its polls drain
//     it), in particular at the block loop's back edge, where the compiler is free to copy registers.
//   * Row ownership: o / down rows [32 w, +32) (two row blocks); k | v: row block w >> 1 of the stacked [k; v] rows on the odd
//     workgroups; q: rows [64 p, +64) per PAIR of workgroups, three row blocks on the even one, one on the odd one; gate / up: ONE of the two matrices per workgroup (w >> 7), rows k * 4096 + 32 (w & 127) + i, k < 7, i < 32 (fourteen row blocks: COLUMNS [32 (w & 127), +32) of the
//     (7, 4096) view, so the 7 x 7 mix of the output transform is local to the workgroup).
//   * 8192-point transforms on 512 threads x 16 elements with one LDS exchange, in two directions (fht_wg512x.hip.h):
//     gather (natural order) -> fwd -> residual / RMSNorm / SU in the strided layout (the static vectors are stored
//     pre-permuted by the host) -> rev -> digit planes as 16-byte pieces.
//   * The MLP edge (28672 = 7 x 4096): column-local 7 x 7 mix of z_gate / z_up -> hand-off to 7 chunk owners, which run
//     the 4096-point transforms of their chunk of gate and up, SV, SiLU product, SU and down's 4096-point transform ->
//     hand-off of the 7 x 4096 rows to everybody, where the 7 x 7 mix of down's input side and the digit planes happen
//     on the fly (block exponent from the owners' maxima: a bound, known before the rows are swept).
//   * Grouped-query attention: head h on workgroup 4 h (the owner of its first rows); from 128 positions on the four
//     workgroups of a head take every fourth position each and merge their softmax states through one more hand-off.
//
// Hand-off protocol, liveness, error word: engine_sync.hip.h / decode_block.hip.  E8P12 (16 copies of both tables: 64 KB
// of tables + 84 KB of planes of down's input fill the LDS).
#include "e8p_gemv_core.hip.h"
#include "engine_sync.hip.h"
#include "fht_wg512x.hip.h"
#include <utility>

// table mode (e8p_gemv_core.hip.h: Lds<REP>): 4 = NIBBLE MODE (round 6, the shipped one): 4-byte entries, 32 conflict-free copies
// of both tables in 64 KB, half planes, the accumulator rows hold 8 x the digit sums.  16 = 16 / 16 copies of the 8-byte
// tables (rounds 4-5; kept as the A/B reference: same integers, bit-identical launch).  24 (T1 x 32, T2 x 16) and 32 exist for
// the MEASUREMENT MODE only (tools/dbg: the planes do not fit then; the mode does not use them)
#ifndef QUIP_GQA_NODECODE
#define QUIP_GQA_NODECODE 0
#endif
#ifndef QUIP_GQA_REP
#define QUIP_GQA_REP 4
#endif
// tools/dbg builds only (tools/gqa_waitstat.py): every wave adds up the clocks it spends in the ring's `s_waitcnt vmcnt` and
// writes {wait ticks, total ticks, s_memrealtime span} of the launch to dbg[(workgroup * 8 + wave) * 4 ..]
#ifndef QUIP_GQA_WAITSTAT
#define QUIP_GQA_WAITSTAT 0
#endif
// 1 = the products as a software pipeline over half items (the look-ups of an item's first half issued before the second half of
// the item in front of it is multiplied).  Measured and NOT shipped (profiles/r06_gqa_stream.txt): 1251 instead of 1270 clocks per
// item and wave, the same 0.70-0.72 of 8 TB/s in wall time (the launch runs at the clock its power allows: 1.74-1.94 GHz against
// 1.95-2.06 on the same box), 256 VGPRs instead of 238.
#ifndef QUIP_GQA_PIPE
#define QUIP_GQA_PIPE 0
#endif
// nibble mode: items whose table look-ups run INSIDE a wait (their codes landed long ago; a decoded item waits as 16 scalar
// registers) and are multiplied from registers once the digit planes exist: the first ... items of gate / up (wait for z_o), of down
// (wait for the chunk owners; the owners: for their inbox) and of the next block's q / k | v (wait for z_d), as decode_block.hip
// does.  Measured (profiles/r06_gqa_predecode.txt, same box, tools/dbg/tok70b.py): 2 / 1 / 1 181.2-181.6, 2 / 2 / 2 180.4-180.7,
// 3 / 2 / 2 179.7-179.9, none 180.5-180.6 tok/s -- the short hand-offs (z_o, z_d: ~3.5K clocks = the all-gather's own latency)
// are not idle time for the workgroups that close them: look-ups in front of their poll make the poll late by what they cost
// (top -> z_d gathered 4.2K -> 5.7K clocks with two items).  Only the LONG wait -- 17K clocks for the chunk owners -- takes them
// for free: down's first two items (products down 15.2K -> 14.6K clocks).
#ifndef QUIP_GQA_PRE_GU
#define QUIP_GQA_PRE_GU 0
#endif
#ifndef QUIP_GQA_PRE_D
#define QUIP_GQA_PRE_D 2
#endif
#ifndef QUIP_GQA_PRE_Q
#define QUIP_GQA_PRE_Q 0
#endif
// hop 1 of the MLP edge (columns -> chunk owners): 1 = two values per 8-byte granule as 20-bit mantissas against the pair's larger
// binary exponent (2^-19 of the larger one: far below the fp16 rounding of the modules' outputs that went into them) + a 16-bit tag;
// 0 = one fp32 value per granule (rounds 4-5).  An owner's inbox is 64 KB at one value per granule -- ONE sweep of it is ~6K clocks at
// the ~11 bytes per clock a CU loads (stamps 12 -> 28: 6.1K) -- and 32 KB packed.
#ifndef QUIP_GQA_INBOX_PACK
#define QUIP_GQA_INBOX_PACK 1
#endif
// o_proj's input transform split over its producers (1; 0 = rounds 4-6a: every workgroup transforms all 8192 points of a (.) SU).
// H_8192 = H_64 (x) H_128 with the head index on top: a head's workgroup multiplies its 128 attention outputs by SU_o and runs their
// 128-point transform in ONE wave before it publishes them (two fp32 per granule, esync::pack20x2: the same 64 granules per head);
// everybody then gathers four heads x four columns per thread and finishes with H_64 ACROSS heads -- two register stages and four
// lane stages, no LDS exchange, no barrier inside -- and writes a dword of digits per head and plane.  in(o): 5.8K -> ... clocks.
#ifndef QUIP_GQA_OHEAD
#define QUIP_GQA_OHEAD 1
#endif
#ifndef QUIP_GQA_ZROWS         /* tools/dbg A/B: 0 = the MFMA's unused A rows read digit plane 2 (as rounds 1-5) instead of zeros */
#define QUIP_GQA_ZROWS 1
#endif
#ifndef QUIP_GQA_ASMADD        /* tools/dbg A/B: 0 = the rows' LDS adds as compiler-generated atomics under `if (q < 2)` */
#define QUIP_GQA_ASMADD 1
#endif

namespace quip {

namespace {

using esync::u32x2_t;
using esync::u32x4_t;

// one decoder block as the kernel reads it (256 bytes; same field order as decode_block.hip's descriptor)
struct GLayer {
  const uint4* W[7];       // Qidxs of q, k, v, o, gate, up, down
  const f16* ln[2];        // RMSNorm weights, PERMUTED: p[16 t + k] = w[t + 512 k]
  const f16* su[7];        // SU: q, k, v, gate, up permuted; o, down natural
  const f16* sv[7];        // SV: o, down permuted; q, k, v, gate, up natural
  const float* mix;        // fp32 [3][7][8]: gate.had_right, up.had_right (rows), down.had_left TRANSPOSED (rows)
  f16* kcache;             // [kv_heads, max_len, 128]
  f16* vcache;
  float sc[7];             // wscale_float / sqrt(L_in): L_in = 8192 (q k v o gate up), 4096 (down)
  float pad_[5];
};
static_assert(sizeof(GLayer) == 256, "layer descriptor layout");

struct GArgs {
  const GLayer* layers;
  const f16* h_in;         // [8192] natural order
  f16* h_out;
  const int64_t* pos;
  const float* cos;        // [max_len, 128]
  const float* sin;
  const uint64_t* grid;
  char* ws;
  uint64_t* dbg;
  int n_layers, max_len, dbg_layer;
  float rms_eps, attn_scale;
};

constexpr int kWaves = 8, kThreads = 512, NWG = 256, HD = 128;
constexpr int HID = 8192, NH = 64, NKV = 8, GQ = NH / NKV;
constexpr int FK = 7, FL = 4096, NFFN = FK * FL;
constexpr int EPT = HID / kThreads;                 // 16 elements per thread
constexpr int kRowH = HID / 4, kRowF = NFFN / 4;    // bytes of a weight row (hidden- / ffn-wide input)
constexpr int NS = 9, NSEQ = 54;                    // ring slots, items per wave and block
// item sequence of an iteration: [0, 4) o | [4, 32) gate / up | [32, 46) down | 46, 47 fillers | [48, 54) k | v and q
constexpr int SQ_O = 0, SQ_GU = 4, SQ_D = 32, SQ_F = 46, SQ_A = 48, SQ_B = 50;
// [48, 50): slot A = the k | v row block (odd workgroups) or a third q row block (even ones), slices i = 0, 1;
// [50, 54): slot B = q row blocks 0 / 1 x slices 0, 1 (odd workgroups: their one q row block; row block 1 is a filler)
static_assert(NSEQ % NS == 0, "the ring position of an item is the same in every block");
constexpr float kOutScaleH = 0.011048543456039806f;  // 1 / sqrt(8192)
constexpr float kSqrtH = 90.50966799187809f;         // sqrt(8192): |H_8192 x|_inf <= sqrt(8192) |x|_2
constexpr int kParts = 4, kPartGran = 132, kSplitPos = 128;

// workspace (bytes)
constexpr size_t kWsCtl = 0;
constexpr size_t kWsZq = 64, kWsZk = kWsZq + (HID / 2) * 8, kWsZv = kWsZk + (NKV * HD / 2) * 8;
constexpr size_t kWsA = kWsZv + (NKV * HD / 2) * 8, kWsZo = kWsA + (HID / 2) * 8, kWsZd = kWsZo + (HID / 2) * 8;
constexpr size_t kWsInbox = kWsZd + (HID / 2) * 8;                    // [FK][2][FL] granules (fp32 payload)
constexpr size_t kWsRows = kWsInbox + (size_t)FK * 2 * FL * 8;        // [FK][FL / 2] granules (two 24-bit values each)
constexpr size_t kWsRowMax = kWsRows + (size_t)FK * FL * 8;           // [8 copies][16] granules: a 128-byte line per copy, workgroup w polls copy w & 7
constexpr size_t kWsPart = kWsRowMax + 8 * 128;                       // [NH][kParts][132]
constexpr size_t kWsBytes = kWsPart + (size_t)NH * kParts * kPartGran * 8;

struct GLds {
  using T = Lds<QUIP_GQA_REP>;
  static constexpr int kAcc = T::kAcc;                       // int32 [336][4]: q 0 | kv 32 | o 48 | gate, up 80 | down 304
  static constexpr int kAccRows = 336;
  static constexpr int AQ = 0, AKV = 32, AO = 48, AGU = 80, AD = 304;
  static constexpr int kZcol = kAcc + kAccRows * 16;         // float [224]
  static constexpr int kRed = kZcol + 224 * 4;               // float [64], int [8]
  static constexpr int kDesc = kRed + 256 + 32;              // two descriptors
  static constexpr int kQkv = kDesc + 512;                   // fp16 [4][128]: q, k, v of this head; attention output
  static constexpr int kCs = kQkv + 4 * HD * 2;              // float [2][128]
  static constexpr int kMix = kCs + 2 * HD * 4;              // float [3][7][8]
  static constexpr int kZero = kMix + 3 * 7 * 8 * 4;         // 160 zero bytes: what the A rows that carry no digit plane read
  static constexpr int kArea = kZero + 160;
  static constexpr int kAreaBytes = (160 * 1024 - kArea) & ~15;
  static constexpr bool kNib = T::kNib;
  // plane strides.  Byte tables: 16 bytes off a multiple of 256 (the three planes of an A fragment on different banks).  Nibble mode:
  // a plane = its "lo" half (positions 0..3 of the 8-groups) then, 16 bytes off a multiple of 256 later, its "hi" half; planes 64
  // bytes off a multiple of 256 apart: the seven distinct 16-byte pieces a ds_read_b128 lane group touches lie on different banks
  static constexpr int PSH = HID + (kNib ? 64 : 16), PSD = NFFN + (kNib ? 64 : 16);
  static constexpr int HOH = HID / 2 + 16, HOD = NFFN / 2 + 16;      // nibble mode: offset of a plane's "hi" half
  static constexpr int kBufBytes = hadw::Geo<13>::kBufFloats * 4;
  static constexpr int kBytes = kArea + kAreaBytes;
  static_assert((QUIP_GQA_REP != 16 && QUIP_GQA_REP != 4) || (kAreaBytes >= 3 * PSD && kAreaBytes >= 2 * kBufBytes && kAreaBytes >= 2 * 3 * PSH), "transient area");
  static_assert(kArea % 16 == 0, "alignment");
};
static_assert(GLds::kBytes <= 160 * 1024, "LDS budget");

using hadw::bytes4;
using hadw::digit_words;

template <int I> using IC = std::integral_constant<int, I>;
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

__global__ __launch_bounds__(kThreads) void decode_block_gqa_kernel(GArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using B = GLds;
  using T = Lds<QUIP_GQA_REP>;
  int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = blockIdx.x;
  int n = lane & 15, q = lane >> 4;
  uint32_t* ctl = reinterpret_cast<uint32_t*>(a.ws + kWsCtl);
  uint64_t* zq = reinterpret_cast<uint64_t*>(a.ws + kWsZq);
  uint64_t* zk = reinterpret_cast<uint64_t*>(a.ws + kWsZk);
  uint64_t* zv = reinterpret_cast<uint64_t*>(a.ws + kWsZv);
  uint64_t* za = reinterpret_cast<uint64_t*>(a.ws + kWsA);
  uint64_t* zo = reinterpret_cast<uint64_t*>(a.ws + kWsZo);
  uint64_t* zd = reinterpret_cast<uint64_t*>(a.ws + kWsZd);
  uint64_t* inbox = reinterpret_cast<uint64_t*>(a.ws + kWsInbox);
  uint64_t* frow = reinterpret_cast<uint64_t*>(a.ws + kWsRows);
  uint64_t* rowmax = reinterpret_cast<uint64_t*>(a.ws + kWsRowMax);
  uint64_t* pbuf = reinterpret_cast<uint64_t*>(a.ws + kWsPart);
  int dbg_on = 0;
#define BSTAMP(i) do { if (dbg_on && tid == 0) a.dbg[w * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

  // this workgroup's k | v row block (odd workgroups): block kvb of the stacked [k; v] row blocks
  const bool has_kv = (w & 1) != 0;
  const int kvb = w >> 1, kvm = kvb >> 6;                  // kvm: 0 = k, 1 = v
  // gate / up (end of round 5, as decode_block.hip since its restructuring): workgroup w multiplies ONE of the two matrices --
  // mgu = w >> 7: 0 gate, 1 up -- for TWO groups of 16 columns of the (7, 4096) view, [32 (w & 127), +32), instead of both matrices
  // for one group: the same 28 items per wave, but ONE input transform and one set of digit planes on the edge in front of them
  const int mgu = w >> 7;

  // ---- the weight ring ----------------------------------------------------------------------------------------------
  u32x4 qa[NS], qb[NS];
  uint32_t vo_h, vo_qa, vo_qb, vo_qb1, vo_gu, vo_d, vo_hot;
  uint32_t lane_c, lane_c2;
  auto rederive = [&]() __attribute__((always_inline)) {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    tid = t; lane = t & 63; n = lane & 15; q = lane >> 4;
    // Round 5: the launch streams a RE-TILED copy of the codes (the host makes it once per model: decode.py, quip_tile_codes):
    // inside every aligned block of 16 rows the 64-byte pieces of the 16 rows lie side by side, piece after piece --
    //   tiled[rb][c][q][n] = Qidxs[16 rb + n][64 c + 16 q, +16)      (1 KB per (rb, c): exactly one load instruction)
    // so a wave's two instructions of an item are 2 KB of CONSECUTIVE bytes at slice * 2 KB instead of 16 rows x 64 bytes each.
    // tools/ubench/hbm_read.hip: a pure read stream in the old pattern tops at 0.72 (K = 8192) / 0.68 (K = 28672) of 8 TB/s, in
    // full lines at 0.85-0.88.  Row-block offsets are what they were (a block of 16 rows is 16 x row bytes either way).
    const uint32_t lt = (uint32_t)(wave * 2048 + lane * 16);
    vo_h = (uint32_t)(32 * w * kRowH) + lt;
    // q rows: the pair of workgroups (2 p, 2 p + 1) owns rows [64 p, +64): three row blocks on the even one (it has no k | v
    // rows), one on the odd one -- the odd workgroups' extra transform and k | v items no longer make them the last to publish
    vo_qb = (uint32_t)((64 * (w >> 1) + (has_kv ? 48 : 0)) * kRowH) + lt;
    vo_qa = has_kv ? (uint32_t)(16 * (kvb & 63) * kRowH) + lt : vo_qb + (uint32_t)(32 * kRowH);
    vo_qb1 = has_kv ? (uint32_t)((lane & 31) * 16) : vo_qb + (uint32_t)(16 * kRowH);
    vo_gu = (uint32_t)(32 * (w & 127) * kRowH) + lt;
    vo_d = (uint32_t)(32 * w * kRowF) + lt;
    vo_hot = (uint32_t)((lane & 31) * 16);     // (the hot 2 KB: offsets < 496 + 1024 + 80)
    if constexpr (T::kNib) {
      lane_c = nib_lane_const(lane);
      lane_c2 = 0u;
    } else {
      lane_c = (T::kRep1 == 32) ? ((((uint32_t)lane & 31u) << 3) | 0x00010000u) : ((((uint32_t)lane & 15u) << 3) | (uint32_t)T::kT1);
      lane_c2 = (((uint32_t)lane & 15u) << 3) | (uint32_t)T::kT2;
    }
  };
  rederive();
  // uniform matrix bases of the stream: this block's o, gate, up, down; the next block's q, k | v, o; a hot 2 KB
  const uint4 *pw_o, *pw_g, *pw_d, *pw_q, *pw_qa, *pw_qb1, *pw_o2, *pw_hot;
  auto uni = [](const void* p) -> const uint4* {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<const uint4*>(((uint64_t)hi << 32) | lo);
  };
  pw_hot = uni(a.grid);
  // request item T of the sequence (T >= NSEQ: item T - NSEQ of the next iteration) into ring slot T % NS
  auto issue = [&](auto t_c) __attribute__((always_inline)) {
    constexpr int TT = decltype(t_c)::value, t = TT % NSEQ, slot = TT % NS;
    constexpr bool wrap = TT >= NSEQ;
    const uint4* base;
    uint32_t vo;
    constexpr int u_gu = t - SQ_GU, u_d = t - SQ_D, u_q = t - SQ_B;
    // row block (as before) + the 1 KB column span of the old layout = 16 KB of the tiled one (8 waves x 2 KB)
    constexpr int kSpan = 16 * 1024;
    constexpr int off = t < SQ_GU ? (t & 1) * 16 * kRowH + (t >> 1) * kSpan
                        : t < SQ_D ? (u_gu % 7) * FL * kRowH + ((u_gu / 7) & 1) * 16 * kRowH + ((u_gu / 7) >> 1) * kSpan
                        : t < SQ_F ? (u_d & 1) * 16 * kRowF + (u_d >> 1) * kSpan
                        : t < SQ_A ? 0
                        : t < SQ_B ? (t - SQ_A) * kSpan : 0;
    constexpr int imm = 0;
    // (a wrapped request comes from the tail of an iteration: items 0..2 -- requested by down's last item and the fillers --
    //  belong to the NEXT block, items 3..8 -- requested by q and k | v at the top of the iteration -- to this one)
    if constexpr (t < SQ_GU) { base = (wrap && t < 3) ? pw_o2 : pw_o; vo = vo_h; }
    else if constexpr (t < SQ_D) { base = pw_g; vo = vo_gu; }      // (this workgroup's matrix: gate | up)
    else if constexpr (t < SQ_F) { base = pw_d; vo = vo_d; }
    else if constexpr (t < SQ_A) { base = pw_hot; vo = vo_hot; }
    else if constexpr (t < SQ_B) { base = pw_qa; vo = vo_qa; }
    else if constexpr ((u_q & 1) == 0) { base = pw_q; vo = vo_qb; }
    else { base = pw_qb1; vo = vo_qb1; }
    static_assert(!wrap || t < SQ_GU + 7, "a wrapped request stays inside o / gate");
    // q's items (t >= SQ_B): their span -- not for the odd workgroups' fillers, which read the hot 2 KB
    const int offq = t >= SQ_B ? (((u_q & 1) != 0 && has_kv) ? 0 : (u_q >> 1) * kSpan) : 0;
    const uint4* bo = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + off + offq);
    // s_nop: a scalar base fresh from v_readfirstlane / v_readlane needs 5 wait states before a VMEM instruction reads it,
    // and the compiler pads nothing inside an asm statement
    u32x4& da = qa[slot];                              // (named here: an asm operand alone does not capture in a generic lambda)
    u32x4& db = qb[slot];
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(da) : "v"(vo), "s"(bo), "n"(imm) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(db) : "v"(vo), "s"(bo), "n"(imm + 1024) : "memory");
  };
  // after a drain every slot is a plain register again
  auto own_ring = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < NS; ++s) { esync::own(qa[s]); esync::own(qb[s]); }
  };
  int* accs = reinterpret_cast<int*>(smem + B::kAcc);
#if QUIP_GQA_WAITSTAT
  uint64_t ws_wait = 0;
  const uint64_t ws_start = __builtin_amdgcn_s_memtime(), ws_rstart = __builtin_amdgcn_s_memrealtime();
#endif
  // what an item (or a chain of items over K slices) leaves in registers: byte tables: the three digit sums in r; nibble mode:
  // r = A x dwords, m = A x low nibbles (e8p_gemv_core.hip.h), to be combined with the digit sums sx of the same K slices
  struct Acc { i32x4 r, m; };
  constexpr bool NIB = T::kNib;
  constexpr int NA = NIB ? 4 : 8;                      // A fragments of a K slice
  constexpr int kUnsc = NIB ? 5 : 2;                   // the accumulator rows hold 2^kUnsc x sum of digit x w
  auto add_rows = [&](const Acc& c, const i32x4& sx, int accrow) __attribute__((always_inline)) {
    if constexpr (NIB) {
      int v[3];
      item_rows_nib(c.r, c.m, sx, nib_lane_factors(q), v);
#if QUIP_GQA_ASMADD
      lds_add3_low32((uint32_t)B::kAcc + (uint32_t)(accrow + n) * 16u, v);
#else
      if (q < 2) {
        int* dst = accs + (accrow + n) * 4;
        __hip_atomic_fetch_add(dst + 0, v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 1, v[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 2, v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
#endif
    } else if (q == 0) {
      int* dst = accs + (accrow + n) * 4;
      __hip_atomic_fetch_add(dst + 0, c.r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst + 1, c.r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst + 2, c.r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  const Acc kAcc0 = {i32x4{0, 0, 0, 0}, i32x4{0, 0, 0, 0}};
  // the A fragments of a K slice (+ in nibble mode their digit sums, added to sx)
  auto fragments = [&](uint32_t xa, auto& A, i32x4& sx) __attribute__((always_inline)) {
    if constexpr (NIB) {
      item_fragments_nib(xa, A);
      item_digit_sums_nib(A, sx);
    } else {
      item_fragments(xa, A);
    }
  };
  // item S of the sequence: wait for its slot, turn the codes into table addresses, refill the slot with item S + NS,
  // multiply (A: the digit fragments of the item's K slice, shared by the items of a group); live = false: a filler
  auto consume = [&](auto s_c, const auto& A, Acc& acc_, bool live) {
    i32x4& acc = acc_.r;
    constexpr int S = decltype(s_c)::value, slot = S % NS;
    u32x4& da = qa[slot];
    u32x4& db = qb[slot];
#if QUIP_GQA_WAITSTAT
    const uint64_t ws_t0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(da), "+v"(db) : "n"(2 * (NS - 1)) : "memory");
#if QUIP_GQA_WAITSTAT
    const uint64_t ws_t1 = __builtin_amdgcn_s_memtime();
    ws_wait += ws_t1 - ws_t0;
#endif
#if QUIP_GQA_NODECODE      /* tools/dbg A/B of the measurement mode: the ring alone -- slots waited for and refilled, the codes folded into the accumulator */
    {
      acc.x ^= (int)(da.x ^ da.y ^ da.z ^ da.w);
      acc.y ^= (int)(db.x ^ db.y ^ db.z ^ db.w);
      issue(IC<S + NS>{});
      return;
    }
#endif
    ItemAddr ad;
    if (live) {
      if constexpr (NIB) item_addresses_nib(da, db, lane_c, ad);
      else item_addresses<QUIP_GQA_REP>(da, db, lane_c, lane_c2, ad, 0u);
#pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(ad.a1l[t]), "+v"(ad.a2l[t]), "+v"(ad.a1h[t]), "+v"(ad.a2h[t]));
    }
    issue(IC<S + NS>{});
    if constexpr (NIB) {
      if (live) item_mfma_nib(ad, A, acc_.r, acc_.m);
    } else if (live) {
      constexpr int PIPE = QUIP_GEMV_PIPE;
      uint2 o[8][4];
      auto lk = [&](int t) {
        o[t][0] = lds_read8(ad.a1l[t]); o[t][1] = lds_read8(ad.a2l[t]);
        o[t][2] = lds_read8(ad.a1h[t]); o[t][3] = lds_read8(ad.a2h[t]);
      };
#pragma unroll
      for (int t = 0; t < PIPE; ++t) lk(t);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t + PIPE < 8) lk(t + PIPE);
        const i32x4 Bf = {(int)(o[t][0].x ^ o[t][1].x), (int)(o[t][0].y ^ o[t][1].y), (int)(o[t][2].x ^ o[t][3].x),
                          (int)(o[t][2].y ^ o[t][3].y)};
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[t], Bf, acc, 0, 0, 0);
      }
    }
  };
  // ---- nibble mode: a product's items as a software pipeline over HALF items (round 6) -------------------------------------
  // rocprofv3 on the measurement mode said where an item's ~1400 clocks went: 45 % issuing its ~140 instructions (4 clocks each:
  // ONE wave issues at most one instruction per 4 clocks, and there are two per SIMD), 23 % issue stalls, 32 % in s_waitcnt --
  // most of it lgkmcnt: every item began with a burst of look-ups whose first result its first MFMA had to wait for, because no
  // look-up may move above the asm statements (wait, request) of its own item.  So the look-ups of an item's first half now go out
  // BEFORE the second half of the item in front of it is multiplied:
  //     look(S, half 1) | multiply(S, half 0) | wait(S + 1), codes -> addresses, request(S + 1 + NS) | look(S + 1, half 0) |
  //     multiply(S, half 1) | rows of S -> accumulators
  // -- a look-up has half an item (~60 instructions) between its issue and its use; LDS results return in order, so consuming the
  // older half while 16 newer look-ups are in flight is `lgkmcnt(15)`.  Same integers, same order of the MFMAs per accumulator.
  ItemAddr adn;
  uint32_t L0[16], L1[16];
  // wait for item S's slot, its codes -> the 32 look-up addresses (adn), refill the slot with item S + NS
  auto pre = [&](auto s_c, bool live) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value, slot = S % NS;
    u32x4& da = qa[slot];
    u32x4& db = qb[slot];
#if QUIP_GQA_WAITSTAT
    const uint64_t ws_t0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(da), "+v"(db) : "n"(2 * (NS - 1)) : "memory");
#if QUIP_GQA_WAITSTAT
    ws_wait += __builtin_amdgcn_s_memtime() - ws_t0;
#endif
    if (live) {
      item_addresses_nib(da, db, lane_c, adn);
#pragma unroll
      for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(adn.a1l[t]), "+v"(adn.a2l[t]), "+v"(adn.a1h[t]), "+v"(adn.a2h[t]));
    }
    issue(IC<S + NS>{});
  };
  // the 16 look-ups of half H (MFMA steps 2 H, 2 H + 1: dwords 4 H .. 4 H + 3 of the lane's 8) of the item whose addresses adn holds
  auto look = [&](auto h_c, uint32_t (&L)[16]) __attribute__((always_inline)) {
    constexpr int H = decltype(h_c)::value;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int t = 4 * H + u;
      L[4 * u + 0] = lds_read4(adn.a1l[t]); L[4 * u + 1] = lds_read4(adn.a2l[t]);
      L[4 * u + 2] = lds_read4(adn.a1h[t]); L[4 * u + 3] = lds_read4(adn.a2h[t]);
    }
  };
  auto mulh = [&](auto h_c, const uint32_t (&L)[16], const i32x4 (&A)[4], Acc& acc) __attribute__((always_inline)) {
    constexpr int H = decltype(h_c)::value;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const uint32_t* o = L + 8 * s2;
      const i32x4 Br = {(int)(o[0] ^ o[1]), (int)(o[2] ^ o[3]), (int)(o[4] ^ o[5]), (int)(o[6] ^ o[7])};
      const i32x4 Bm = {Br.x & 0x0f0f0f0f, Br.y & 0x0f0f0f0f, Br.z & 0x0f0f0f0f, Br.w & 0x0f0f0f0f};
      acc.r = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * H + s2], Br, acc.r, 0, 0, 0);
      acc.m = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2 * H + s2], Bm, acc.m, 0, 0, 0);
    }
  };
  // the first item of a stream: nothing in front of it to hide its first look-ups behind
  constexpr bool PIPE = NIB && QUIP_GQA_PIPE != 0;
  auto first = [&](auto s_c, bool live) __attribute__((always_inline)) {
    if constexpr (PIPE) {
      pre(s_c, live);
      if (live) look(IC<0>{}, L0);
    }
  };
  // item S of a stream (its first half's look-ups are in flight: first() or the item in front of it); next_c: another item
  // follows directly (live_next: and it is not a filler)
  auto item = [&](auto s_c, auto next_c, const auto& A, Acc& acc, bool live, bool live_next) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value;
    constexpr bool NEXT = decltype(next_c)::value;
    if constexpr (PIPE) {
      if (live) {
        look(IC<1>{}, L1);
        mulh(IC<0>{}, L0, A, acc);
      }
      if constexpr (NEXT) {
        pre(IC<S + 1>{}, live_next);
        if (live_next) look(IC<0>{}, L0);
      }
      if (live) mulh(IC<1>{}, L1, A, acc);
    } else {
      consume(s_c, A, acc, live);
    }
  };
  using TrueC = std::integral_constant<bool, true>;
  using FalseC = std::integral_constant<bool, false>;
  // CNT consecutive items that multiply the same K slice (A fragments at xa) into accumulator rows accrow0 + 16 c; last_c: the
  // stream ends behind them
  auto group = [&](auto s0_c, auto cnt_c, auto last_c, uint32_t xa, int accrow0) __attribute__((always_inline)) {
    constexpr int S0 = decltype(s0_c)::value, CNT = decltype(cnt_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    i32x4 A[NA];
    i32x4 sx = {0, 0, 0, 0};
    fragments(xa, A, sx);
    static_for<CNT>([&](auto c) {
      constexpr int C = decltype(c)::value;
      Acc acc = kAcc0;
      item(IC<S0 + C>{}, std::integral_constant<bool, !(LAST && C == CNT - 1)>{}, A, acc, true, true);
      add_rows(acc, sx, accrow0 + 16 * C);
    });
  };

  // ---- items decoded ahead (nibble mode) ------------------------------------------------------------------------------
  constexpr bool kPre = NIB && !PIPE && !QUIP_GQA_NODECODE;
  constexpr int kPreGU = kPre ? QUIP_GQA_PRE_GU : 0, kPreD = kPre ? QUIP_GQA_PRE_D : 0, kPreQ = kPre ? QUIP_GQA_PRE_Q : 0;
  static_assert(kPreGU <= 14 && kPreD <= 14 && kPreQ <= 2, "items decoded ahead: inside the first group of their product");
  // item S of the sequence, first half: wait for its slot, codes -> addresses -> the 32 look-ups (16 dwords).  I: how many items
  // in front of it were decoded ahead in the same wait.  The slot is NOT refilled here: a request in front of a hand-off's poll
  // delays the poll's first check (first version: z_o gathered 3.6K -> 4.7K clocks, the whole gain); refill() goes out at the top of
  // the product, behind the gather -- which drained the queue, so the ring's constant wait holds again from there.
  auto predec = [&](auto s_c, auto i_c, uint32_t (&raw)[16]) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value, slot = S % NS, I = decltype(i_c)::value;
    u32x4& da = qa[slot];
    u32x4& db = qb[slot];
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(da), "+v"(db) : "n"(2 * (NS - 1 - I)) : "memory");
    ItemAddr ad;
    item_addresses_nib(da, db, lane_c, ad);
#pragma unroll
    for (int t = 0; t < 8; ++t) asm volatile("" : "+v"(ad.a1l[t]), "+v"(ad.a2l[t]), "+v"(ad.a1h[t]), "+v"(ad.a2h[t]));
    uint32_t r[16];
    nib_decode(ad, r);
#pragma unroll
    for (int t = 0; t < 16; ++t) { asm volatile("" : "+v"(r[t])); raw[t] = r[t]; }
  };
  auto refill = [&](auto s_c) __attribute__((always_inline)) { issue(IC<decltype(s_c)::value + NS>{}); };
  // ... second half: its eight MFMAs
  auto mul_raw = [&](const uint32_t (&raw)[16], const auto& A, Acc& acc) __attribute__((always_inline)) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const i32x4 Br = {(int)raw[4 * s4], (int)raw[4 * s4 + 1], (int)raw[4 * s4 + 2], (int)raw[4 * s4 + 3]};
      const i32x4 Bm = {Br.x & 0x0f0f0f0f, Br.y & 0x0f0f0f0f, Br.z & 0x0f0f0f0f, Br.w & 0x0f0f0f0f};
      acc.r = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s4], Br, acc.r, 0, 0, 0);
      acc.m = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[s4], Bm, acc.m, 0, 0, 0);
    }
  };
  // group() whose first NPRE items were decoded ahead
  auto group_pre = [&](auto s0_c, auto cnt_c, auto last_c, auto npre_c, const auto& pre, uint32_t xa, int accrow0) __attribute__((always_inline)) {
    constexpr int S0 = decltype(s0_c)::value, CNT = decltype(cnt_c)::value, NPRE = decltype(npre_c)::value;
    constexpr bool LAST = decltype(last_c)::value;
    i32x4 A[NA];
    i32x4 sx = {0, 0, 0, 0};
    fragments(xa, A, sx);
    static_for<CNT>([&](auto c) {
      constexpr int C = decltype(c)::value;
      Acc acc = kAcc0;
      if constexpr (C < NPRE) mul_raw(pre[C < NPRE ? C : 0], A, acc);
      else item(IC<S0 + C>{}, std::integral_constant<bool, !(LAST && C == CNT - 1)>{}, A, acc, true, true);
      add_rows(acc, sx, accrow0 + 16 * C);
    });
  };

  // ---- prologue -----------------------------------------------------------------------------------------------------
  GLayer* desc = reinterpret_cast<GLayer*>(smem + B::kDesc);
  u32x2 tsrc;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(tsrc) : "v"(T::kNib ? table_source_ptr_nib(a.grid, lane, wave) : table_source_ptr(a.grid, lane, wave)) : "memory");
  uint32_t gen;
  esync::ld4(gen, ctl);
  asm volatile("s_waitcnt vmcnt(1)" : "+v"(tsrc) : : "memory");
  if constexpr (T::kNib) fill_tables_nib(tsrc, lane, wave);
  else fill_tables_from_lane<QUIP_GQA_REP>(smem, tsrc, lane, wave);
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(gen) : : "memory");
  const uint32_t ebase = ((uint32_t)__builtin_amdgcn_readfirstlane((int)gen) + 1u) << 10;
  for (int i = tid; i < B::kAccRows * 4; i += kThreads) accs[i] = 0;
  if (tid < 40) reinterpret_cast<uint32_t*>(smem + B::kZero)[tid] = 0u;
  // (measurement mode: no edge ever writes digit planes -- fill the area with digits that look like real ones, so that the
  //  matrix cores switch as in a real launch: the rate of this mode is a power figure too)
  if (a.dbg_layer == -2)
    for (int i = tid; i < B::kAreaBytes / 4; i += kThreads) reinterpret_cast<uint32_t*>(smem + B::kArea)[i] = (uint32_t)(i + 1) * 2654435761u ^ (uint32_t)w * 0x9e3779b9u;
  if (tid < 64) reinterpret_cast<uint32_t*>(desc)[tid] = reinterpret_cast<const uint32_t*>(a.layers)[tid];
  const long long pos64 = *a.pos;
  const bool pos_ok = pos64 >= 0 && pos64 < (long long)a.max_len;
  const int pos = pos_ok ? (int)pos64 : 0;
  if (tid < 2 * HD)
    reinterpret_cast<float*>(smem + B::kCs)[tid] = (tid < HD ? a.cos : a.sin - HD)[(size_t)pos * HD + tid];
  // the residual stream, strided layout: this thread's h[t + 512 k], k < 16, as 8 fp16 pairs (registers for the whole launch:
  // the LDS holds tables and planes)
  uint32_t hreg[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const f16 lo = a.h_in[tid + 512 * (2 * j)], hi = a.h_in[tid + 512 * (2 * j + 1)];
    hreg[j] = (uint32_t)__builtin_bit_cast(uint16_t, lo) | ((uint32_t)__builtin_bit_cast(uint16_t, hi) << 16);
  }
  had::wg_barrier<false>();                          // (a full fence: nothing counted is in flight yet)
  uint32_t hop = 0;

  float* red = reinterpret_cast<float*>(smem + B::kRed);
  int* shs = reinterpret_cast<int*>(smem + B::kRed + 256);
  float* xbuf = reinterpret_cast<float*>(smem + B::kArea);

  // bases of the stream for the block whose descriptor sits in slot `cur` (o, gate, up, down) and the one in slot `nxt`
  auto set_bases = [&](const GLayer& C, const GLayer& N) __attribute__((always_inline)) {
    pw_o = uni(C.W[3]); pw_g = uni(C.W[4 + mgu]); pw_d = uni(C.W[6]);
    pw_q = uni(N.W[0]); pw_qa = has_kv ? uni(N.W[1 + kvm]) : pw_q; pw_qb1 = has_kv ? pw_hot : pw_q; pw_o2 = uni(N.W[3]);
  };
  set_bases(desc[0], desc[0]);
  // the ring's first nine items: q and k | v of block 0, its first three o items
  issue(IC<SQ_A>{}); issue(IC<SQ_A + 1>{});
  issue(IC<SQ_B>{}); issue(IC<SQ_B + 1>{}); issue(IC<SQ_B + 2>{}); issue(IC<SQ_B + 3>{});
  issue(IC<NSEQ>{}); issue(IC<NSEQ + 1>{}); issue(IC<NSEQ + 2>{});

  // ---- all-gather of an 8192-vector (4096 granules {2 x fp16, tag}): thread t takes elements [16 t, +16) ----------------
  auto gather16 = [&](const uint64_t* vec, uint32_t tag, uint32_t code, float (&out)[16]) {
    u32x4_t p[4];
    uint32_t spins = 0;
    const uint64_t* src = vec + 8 * tid;
#pragma unroll
    for (int j = 0; j < 4; ++j) p[j] = u32x4_t{0u, 0u, 0u, 0u};
    bool ok = false;
    for (;;) {
      if (!ok) {                                       // (a lane whose pieces are all there sits the retries out)
#pragma unroll
        for (int j = 0; j < 4; ++j) esync::ld16_keep(p[j], src + 2 * j);
      }
      esync::drain();
      bool now = true;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        esync::own(p[j]);
        now = now && p[j].y == tag && p[j].w == tag;
      }
      ok = now;
      if (esync::spin_step(ok, spins, ctl + 1, code + (uint32_t)w)) break;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f16x2 h0 = as_f16x2(p[j].x), h1 = as_f16x2(p[j].z);
      out[4 * j] = (float)h0.x; out[4 * j + 1] = (float)h0.y; out[4 * j + 2] = (float)h1.x; out[4 * j + 3] = (float)h1.y;
    }
    own_ring();
  };
  // a long wait is spent on ONE granule per wave (a line of its own choice), not on re-reading a whole vector per retry next to
  // the stores that are awaited; the gather behind it re-checks every piece
  auto prepoll = [&](const uint64_t* g1, uint32_t tag, uint32_t code) {
    uint32_t sp = 0;
    u32x2_t f;
    for (;;) {
      esync::ld8(f, g1);
      esync::drain();
      esync::own(f);
      if (esync::spin_step(f.y == tag, sp, ctl + 1, code + (uint32_t)w)) break;
    }
  };
  // 16 fp16 of a permuted / natural vector at p + 16 t -> floats
  auto load16 = [&](const f16* p, u32x4 (&v)[2]) {
    v[0] = *reinterpret_cast<const u32x4*>(p + 16 * tid);
    v[1] = *reinterpret_cast<const u32x4*>(p + 16 * tid + 8);
  };
  auto unpack16 = [](const u32x4 (&v)[2], float (&o)[16]) {
    had::unpack8(make_uint4(v[0].x, v[0].y, v[0].z, v[0].w), o);
    had::unpack8(make_uint4(v[1].x, v[1].y, v[1].z, v[1].w), o + 8);
  };
  // rows [row0, row0 + 2 cnt) of the accumulators (block exponent sh) -> cnt granules starting at gran
  auto publish = [&](uint64_t* vec, int gran, int row0, int cnt, int sh, uint32_t tag) __attribute__((always_inline)) {
    if (tid < cnt) {
      const int* s3 = accs + (row0 + 2 * tid) * 4;
      const float us = unscale_of(sh, kUnsc);
      const float f0 = __builtin_fmaf((float)s3[0], 65536.f, __builtin_fmaf((float)s3[1], 256.f, (float)s3[2]));
      const float f1 = __builtin_fmaf((float)s3[4], 65536.f, __builtin_fmaf((float)s3[5], 256.f, (float)s3[6]));
      esync::st_granule(vec + gran + tid, pack_f16(f0 * us, f1 * us), tag);
    }
  };
  auto zero_acc = [&](int row0, int rows) __attribute__((always_inline)) {
    for (int i = tid; i < rows * 4; i += kThreads) accs[row0 * 4 + i] = 0;
  };
  // digit planes of a transformed vector held in natural order (16 consecutive values per thread): three 16-byte pieces
  auto planes_nat = [&](const float (&v)[16], float scale, int sh, uint32_t base) __attribute__((always_inline)) {
#pragma clang fp contract(off)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float s2 = had::fmul(scale, as_f32((uint32_t)(sh + 127) << 23));
    uint32_t dg[3][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {      // (round 5: digits straight from the fp32 magic number, hadw::digit_words_magic)
      const float vv[4] = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
      hadw::digit_words_magic(vv, s2, dg[0][g], dg[1][g], dg[2][g]);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if constexpr (NIB) {       // dwords 0, 2 (positions 0..3 of the thread's two 8-groups) -> "lo" half, dwords 1, 3 -> "hi" half
        *reinterpret_cast<uint2*>(smem + base + d * B::PSH + 8 * tid) = make_uint2(dg[d][0], dg[d][2]);
        *reinterpret_cast<uint2*>(smem + base + d * B::PSH + B::HOH + 8 * tid) = make_uint2(dg[d][1], dg[d][3]);
      } else {
        *reinterpret_cast<uint4*>(smem + base + d * B::PSH + 16 * tid) = make_uint4(dg[d][0], dg[d][1], dg[d][2], dg[d][3]);
      }
    }
  };
  // the same from the strided layout (values X[t + 512 k]): bytes
  auto planes_str = [&](const float (&v)[16], float scale, int sh, uint32_t base) {
#pragma clang fp contract(off)
    const float s2 = had::fmul(scale, as_f32((uint32_t)(sh + 127) << 23));
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int X = (int)__builtin_rintf(v[k] * s2);
      // (nibble mode: digit tid + 512 k of a plane sits at byte nib_byte_of() of its half plane)
      uint8_t* p = reinterpret_cast<uint8_t*>(smem + base) + (NIB ? nib_half_of(tid) * B::HOH + nib_byte_of(tid) + 256 * k : tid + 512 * k);
      p[0] = (uint8_t)((X + 0x8080) >> 16);
      p[B::PSH] = (uint8_t)((X + 0x80) >> 8);
      p[2 * B::PSH] = (uint8_t)X;
    }
  };
  auto wg_max = [&](float mx) __attribute__((always_inline)) {
    mx = had::wave_reduce_to_lane63<true>(mx);
    had::wg_barrier<true>();
    if (lane == 63) red[16 + wave] = mx;
    had::wg_barrier<true>();
    float r = red[16];
#pragma unroll
    for (int i = 1; i < 8; ++i) r = fmaxf(r, red[16 + i]);
    return r;
  };

  // ---- an 8192-wide edge: output side of the producer (+ residual) and the input transforms of up to two consumers ------
  //   zvec: gather z (hand-off `tag`), h += SV_prev (.) H z / sqrt(8192)                  [qlinear.py:106-114 of the producer]
  //   NC consumers (0 | 2; `two` false: only consumer 0): planes_i = digits( sc_i rms(h) H (h (.) ln (.) su_i) )
  //   [RMSNorm + qlinear.py:90-100]; planes of consumer i at area + i * 3 * PSH, block exponent in shs[sh0 + i]
  // sv_prev, ln, su0, su1: permuted vectors.
  auto edge = [&](auto nc_tag, const uint64_t* zvec, uint32_t tag, uint32_t code, const f16* sv_prev, const f16* ln, const f16* su0,
                  const f16* su1, float sc0, float sc1, bool two, int sh0, int sb) __attribute__((always_inline)) {
#define ESTAMP(i) do { if (sb >= 0) BSTAMP(sb + (i)); } while (0)
    constexpr int NC = decltype(nc_tag)::value;
    u32x4 psv[2], pln[2], ps0[2], ps1[2];
    if (zvec) load16(sv_prev, psv);
    if (NC > 0) {
      load16(ln, pln);
      load16(su0, ps0);
      load16(two ? su1 : su0, ps1);
    }
    if (zvec) {
      float v[1][16];
      gather16(zvec, tag, code, v[0]);
      asm volatile("" : "+v"(psv[0]), "+v"(psv[1]));
      if (NC > 0) asm volatile("" : "+v"(pln[0]), "+v"(pln[1]), "+v"(ps0[0]), "+v"(ps0[1]), "+v"(ps1[0]), "+v"(ps1[1]));
      ESTAMP(0);
      hadw::fwd<13, 1, true>(v, xbuf, tid);
      ESTAMP(1);
      float svf[16];
      unpack16(psv, svf);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f16x2 hh = as_f16x2(hreg[j]);
        const f16 n0 = had::out_elem(v[0][2 * j], kOutScaleH, true, svf[2 * j], false, 0.f, true, (float)hh.x);
        const f16 n1 = had::out_elem(v[0][2 * j + 1], kOutScaleH, true, svf[2 * j + 1], false, 0.f, true, (float)hh.y);
        hreg[j] = (uint32_t)__builtin_bit_cast(uint16_t, n0) | ((uint32_t)__builtin_bit_cast(uint16_t, n1) << 16);
      }
    }
    if constexpr (NC > 0) {
      float e[16];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f16x2 hh = as_f16x2(hreg[j]);
        e[2 * j] = (float)hh.x;
        e[2 * j + 1] = (float)hh.y;
      }
      // (the sum of squares is needed for the scale only: it is reduced together with the maxima behind the transform)
      float ssw = 0.f;
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int r = 0; r < 16; ++r) ssw = __builtin_fmaf(e[r], e[r], ssw);
        ssw = had::wave_reduce_to_lane63<false>(ssw);
      }
      ESTAMP(2);
      {
        float lnf[16];
        unpack16(pln, lnf);
#pragma unroll
        for (int k = 0; k < 16; ++k) e[k] = had::fmul(e[k], lnf[k]);
      }
      auto total = [&]() __attribute__((always_inline)) {          // after the barrier pair: the eight waves' sums in order
        float r = red[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) r = had::fadd(r, red[i]);
        return r;
      };
      if (two) {
        float v[2][16], s0f[16], s1f[16];
        unpack16(ps0, s0f);
        unpack16(ps1, s1f);
#pragma unroll
        for (int k = 0; k < 16; ++k) { v[0][k] = had::fmul(e[k], s0f[k]); v[1][k] = had::fmul(e[k], s1f[k]); }
        // (round 5: block exponents from the norm bound |H x|_inf <= sqrt(8192) |x|_2 -- decode_block.hip's edge -- whose sums
        //  are known BEFORE the transform and ride on its barriers: no maxima behind it, one barrier pair less)
        float n0 = 0.f, n1 = 0.f;
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int r = 0; r < 16; ++r) { n0 = __builtin_fmaf(v[0][r], v[0][r], n0); n1 = __builtin_fmaf(v[1][r], v[1][r], n1); }
        }
        n0 = had::wave_reduce_to_lane63<false>(n0);
        n1 = had::wave_reduce_to_lane63<false>(n1);
        if (lane == 63) { red[wave] = ssw; red[16 + wave] = n0; red[24 + wave] = n1; }
        hadw::rev<13, 2, true, true>(v, xbuf, tid);
        ESTAMP(3);
        const float tot = total();
        float q0 = red[16], q1 = red[24];
#pragma unroll
        for (int i = 1; i < 8; ++i) { q0 = had::fadd(q0, red[16 + i]); q1 = had::fadd(q1, red[24 + i]); }
        const float s0 = had::rms_scale(sc0, tot, HID, a.rms_eps), s1 = had::rms_scale(sc1, tot, HID, a.rms_eps);
        const int h0 = had::shift_for(sqrtf(q0) * kSqrtH * fabsf(s0) * 1.0625f), h1 = had::shift_for(sqrtf(q1) * kSqrtH * fabsf(s1) * 1.0625f);
        planes_nat(v[0], s0, h0, (uint32_t)B::kArea);
        planes_nat(v[1], s1, h1, (uint32_t)(B::kArea + 3 * B::PSH));
        if (tid == 0) { shs[sh0] = h0; shs[sh0 + 1] = h1; }
      } else {
        float v[1][16], s0f[16];
        unpack16(ps0, s0f);
#pragma unroll
        for (int k = 0; k < 16; ++k) v[0][k] = had::fmul(e[k], s0f[k]);
        float n0 = 0.f;
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int r = 0; r < 16; ++r) n0 = __builtin_fmaf(v[0][r], v[0][r], n0);
        }
        n0 = had::wave_reduce_to_lane63<false>(n0);
        if (lane == 63) { red[wave] = ssw; red[16 + wave] = n0; }
        hadw::rev<13, 1, true, true>(v, xbuf, tid);
        ESTAMP(3);
        const float tot = total();
        float q0 = red[16];
#pragma unroll
        for (int i = 1; i < 8; ++i) q0 = had::fadd(q0, red[16 + i]);
        const float s0 = had::rms_scale(sc0, tot, HID, a.rms_eps);
        const int h0 = had::shift_for(sqrtf(q0) * kSqrtH * fabsf(s0) * 1.0625f);
        planes_nat(v[0], s0, h0, (uint32_t)B::kArea);
        if (tid == 0) shs[sh0] = h0;
      }
      had::wg_barrier<true>();
      ESTAMP(4);
    }
    // (opaque from here on: otherwise the fp32 images of h made above are kept alive until the next edge's update -- 16 registers
    //  through the attention and o phases)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(hreg[j]));
#undef ESTAMP
  };
  // A fragment address of this lane for K slice (wave + 8 i) of the planes at `base` (plane stride ps)
  auto xaddr = [&](uint32_t base, int ps, int i) __attribute__((always_inline)) -> uint32_t {
    if constexpr (NIB) {         // A rows 0..2: "hi" halves of the planes, rows 4..6: "lo" halves
      const uint32_t half_off = (uint32_t)(ps / 2 - 16);          // = HOH | HOD: (K + 64) / 2 - 16 = K / 2 + 16
      const uint32_t xa = base + (uint32_t)(n < 4 ? min(n, 2) * ps + (int)half_off : min(n - 4, 2) * ps) + (uint32_t)q * 32u + (uint32_t)(wave + 8 * i) * 256u;
      // the ten rows that carry nothing read ZEROS (one broadcast address): the launch runs at the clock its power allows, and
      // ten sixteenths of every MFMA's multipliers switching on a copy of plane 2 is power without a result
      return (QUIP_GQA_ZROWS && (n == 3 || n >= 7)) ? (uint32_t)B::kZero : xa;
    } else {
      return base + (uint32_t)min(n, 2) * (uint32_t)ps + (uint32_t)q * 64u + (uint32_t)(wave + 8 * i) * 512u;
    }
  };

  had::wg_barrier<true>();
  const f16* sv_prev = nullptr;                      // SV of the previous block's down_proj (permuted)
  // MEASUREMENT MODE (dbg_layer == -2, tools/gqa_stream.py, bench.py: gemv_stream_in_launch): the products of every block --
  // the same 54 items per wave through the same ring, the same decode and MFMAs -- WITHOUT the edges, the attention and the
  // hand-offs between them, so that the weight stream never waits for an input: what the E8P12 decode GEMV reaches on the 70B
  // layer shapes when nothing but HBM holds it back (bytes landed per time of a launch that runs only the
  // product phases; FETCH_SIZE of the same launch in profiles/).  The digit planes are whatever the LDS holds: no result.
  const bool so = a.dbg_layer == -2;
  for (int l = 0; l < a.n_layers; ++l) {
    dbg_on = a.dbg != nullptr && l == a.dbg_layer;
    rederive();
    BSTAMP(0);
    const GLayer& Ld = desc[l & 1];
    const GLayer& Ln = desc[(l + 1) & 1];
    {
      // the next block's descriptor (the last block: its own once more -- requests beyond the end re-read its rows)
      const int ln_ = l + 1 < a.n_layers ? l + 1 : l;
      if (tid < 64) reinterpret_cast<uint32_t*>(&desc[(l + 1) & 1])[tid] = reinterpret_cast<const uint32_t*>(a.layers + ln_)[tid];
      if (tid >= 64 && tid < 64 + 3 * 7 * 8) reinterpret_cast<float*>(smem + B::kMix)[tid - 64] = Ld.mix[tid - 64];
      had::wg_barrier<false>();
      set_bases(Ld, Ln);
    }
    // ================= P1: output side of the previous block's down_proj + residual (block 0: h = the embedding row), RMSNorm,
    // input transforms of q and k | v; their products; hand-off ==============================================================
    uint32_t Pq[kPreQ > 0 ? kPreQ : 1][16];            // (slot A's items: decoded inside the wait for z_d)
    static_for<kPreQ>([&](auto ic) { predec(IC<SQ_A + decltype(ic)::value>{}, ic, Pq[decltype(ic)::value]); });
    if (!so) edge(IC<2>{}, l > 0 ? zd : nullptr, ebase | hop, 0x4000u, sv_prev, Ld.ln[0], Ld.su[0], Ld.su[1 + kvm], Ld.sc[0], Ld.sc[1 + kvm], has_kv, 0, 18);
    BSTAMP(2);
    rederive();
    {
      const uint32_t p0 = (uint32_t)B::kArea, p1 = (uint32_t)(B::kArea + 3 * B::PSH);
      // slot A first: the k | v items (odd workgroups), so that z_k / z_v are on their way while q is multiplied
      const uint32_t pa = has_kv ? p1 : p0;
      i32x4 A[NA];
      static_for<kPreQ>([&](auto ic) { refill(IC<SQ_A + decltype(ic)::value>{}); });
      first(IC<SQ_A>{}, true);
      {
        Acc acc = kAcc0;
        i32x4 sx = {0, 0, 0, 0};
        fragments(xaddr(pa, B::PSH, 0), A, sx);
        if constexpr (kPreQ > 0) mul_raw(Pq[0], A, acc); else item(IC<SQ_A>{}, TrueC{}, A, acc, true, true);
        fragments(xaddr(pa, B::PSH, 1), A, sx);
        if constexpr (kPreQ > 1) mul_raw(Pq[kPreQ > 1 ? 1 : 0], A, acc); else item(IC<SQ_A + 1>{}, FalseC{}, A, acc, true, true);
        add_rows(acc, sx, B::AKV);
      }
      had::wg_barrier<true>();
      ++hop;                                           // hand-off: z_k / z_v, then z_q (the same index: they are different vectors)
      if (has_kv && !so) publish(kvm ? zv : zk, 8 * (kvb & 63), B::AKV, 8, shs[1], ebase | hop);
      first(IC<SQ_B>{}, true);
      static_for<2>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        Acc acc0 = kAcc0, acc1 = kAcc0;
        i32x4 sx = {0, 0, 0, 0};
        fragments(xaddr(p0, B::PSH, I), A, sx);
        item(IC<SQ_B + 2 * I>{}, TrueC{}, A, acc0, true, !has_kv);
        item(IC<SQ_B + 2 * I + 1>{}, std::integral_constant<bool, I == 0>{}, A, acc1, !has_kv, true);
        add_rows(acc0, sx, B::AQ);
        if (!has_kv) add_rows(acc1, sx, B::AQ + 16);
      });
    }
    had::wg_barrier<true>();
    // rows [64 p, +48) (even workgroup: accumulator rows 0 .. 47) | [64 p + 48, +16) (odd one: rows 0 .. 15)
    if (!so) {
      if (has_kv) publish(zq, 32 * (w >> 1) + 24, B::AQ, 8, shs[0], ebase | hop);
      else publish(zq, 32 * (w >> 1), B::AQ, 24, shs[0], ebase | hop);
    }
    had::wg_barrier<true>();
    zero_acc(B::AQ, 48);
    BSTAMP(3);

    // ================= P2: attention =====================================================================================
    // head hd on workgroup 4 hd; from kSplitPos positions on the head's four workgroups take every fourth position each
    const bool split = pos >= kSplitPos;
    const int part = w & 3, nparts = split ? kParts : 1;
    const bool head_wg = part == 0;
    const int hd = w >> 2, kvh = hd / GQ;
    if (!so && (head_wg || split)) {
      // wave 0 / 1: the k / v vector (1024 values, 16 per lane), SV of this KV head's values
      const bool kvw = wave < 2;
      const f16* svkv = Ld.sv[1 + (wave & 1)];
      u32x4 psvkv[2];
      psvkv[0] = *reinterpret_cast<const u32x4*>(svkv + 16 * lane);
      psvkv[1] = *reinterpret_cast<const u32x4*>(svkv + 16 * lane + 8);
      float c8[8], s8[8];
      const int d0 = (tid & 15) * 8;
      constexpr int LPK = HD / 8, NG = 256 / LPK, U = 2;
      const int g = (tid & 255) / LPK;
      const f16* kc = Ld.kcache + (size_t)kvh * a.max_len * HD;
      const f16* vc = Ld.vcache + (size_t)kvh * a.max_len * HD;
      uint4 kr0[U], vr0[U], kr1[U], vr1[U];
      auto load_round = [&](uint4 (&kr)[U], uint4 (&vr)[U], int i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int t = part + nparts * (i0 + u * NG);
          const int tc = t < pos ? t : 0;
          kr[u] = *reinterpret_cast<const uint4*>(kc + (size_t)tc * HD + d0);
          vr[u] = *reinterpret_cast<const uint4*>(vc + (size_t)tc * HD + d0);
        }
      };
      if (tid < 256) {
        load_round(kr0, vr0, g);
        if (part + nparts * NG * U < pos) load_round(kr1, vr1, g + NG * U);
      }
      // SV of this head's q values (wave 2 finishes them below): requested in front of the hand-off's wait, not at its use
      uint32_t svq_raw = 0u;
      if (wave == 2) svq_raw = *reinterpret_cast<const uint32_t*>(Ld.sv[0] + HD * hd + 2 * lane);
#if QUIP_GQA_OHEAD
      uint32_t suo_raw = 0u;                           // SU_o of this head's 128 attention outputs (natural order)
      if (wave == 0) suo_raw = *reinterpret_cast<const uint32_t*>(Ld.su[3] + HD * hd + 2 * lane);
#endif
      f16* s_qkv = reinterpret_cast<f16*>(smem + B::kQkv);
      {
        // gather z_q (everybody) and z_k / z_v (waves 0 / 1: they went out first, ahead of the q items) in one poll
        float v[1][16];
        u32x4_t pk[4];
        {
          u32x4_t p[4];
          uint32_t spins = 0;
          const uint64_t* src = zq + 8 * tid;
          const uint64_t* srck = ((wave & 1) ? zv : zk) + 8 * lane;
          const uint32_t tag = ebase | hop;
          for (;;) {
#pragma unroll
            for (int j = 0; j < 4; ++j) esync::ld16(p[j], src + 2 * j);
            if (kvw) {
#pragma unroll
              for (int j = 0; j < 4; ++j) esync::ld16(pk[j], srck + 2 * j);
            }
            esync::drain();
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              esync::own(p[j]);
              ok = ok && p[j].y == tag && p[j].w == tag;
            }
            if (kvw) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                esync::own(pk[j]);
                ok = ok && pk[j].y == tag && pk[j].w == tag;
              }
            }
            if (esync::spin_step(ok, spins, ctl + 1, 0x5000u + (uint32_t)w)) break;
          }
          own_ring();
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f16x2 h0 = as_f16x2(p[j].x), h1 = as_f16x2(p[j].z);
            v[0][4 * j] = (float)h0.x; v[0][4 * j + 1] = (float)h0.y; v[0][4 * j + 2] = (float)h1.x; v[0][4 * j + 3] = (float)h1.y;
          }
        }
        BSTAMP(4);
        // Only this head's 128 values of H_8192 z are needed: H_8192 = H_64 (x) H_128, so u[j2] = sum_j1 H_64[hd][j1] z[128 j1 + j2]
        // (signed sums of the 64 segments), then a 128-point transform of u in one wave (wave 2; waves 0 / 1 transform k / v
        // meanwhile) -- instead of the whole 8192-point transform in every head's workgroup.
        // Thread t holds z[16 t + r]: j1 = t >> 3, j2 = 16 (t & 7) + r.
        {
          const float sgn = (__builtin_popcount((uint32_t)hd & (uint32_t)(tid >> 3)) & 1) ? -1.f : 1.f;
          float u[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) u[r] = had::fadd(v[0][r] * sgn, had::lane_xor<3>(v[0][r] * sgn));      // lane bit 3 = j1 bit 0
          had::wg_barrier<true>();
          if ((lane & 8) == 0) {
            float* dstp = xbuf + ((wave * 4 + (lane >> 4)) * 128 + (lane & 7) * 16);
#pragma unroll
            for (int r = 0; r < 16; r += 4) *reinterpret_cast<float4*>(dstp + r) = float4{u[r], u[r + 1], u[r + 2], u[r + 3]};
          }
        }
        had::wg_barrier<true>();
        // (k | v on waves 0 / 1 and q on wave 2 -- three SIMDs -- side by side behind ONE barrier: the k / v transform used to sit in
        //  front of it, with wave 2 waiting)
        if (kvw) {
          float kv[16], svf[16];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f16x2 h0 = as_f16x2(pk[j].x), h1 = as_f16x2(pk[j].z);
            kv[4 * j] = (float)h0.x; kv[4 * j + 1] = (float)h0.y; kv[4 * j + 2] = (float)h1.x; kv[4 * j + 3] = (float)h1.y;
          }
          hadw::wave_fht1024(kv, lane);
          unpack16(psvkv, svf);
          // this KV head's 128 values: lanes [8 kvh, +8), all 16 registers
          if ((lane >> 3) == kvh) {
            f16* dst = s_qkv + HD * (1 + (wave & 1)) + 16 * (lane & 7);
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[r] = had::out_elem(kv[r], 1.f / 32.f, true, svf[r], false, 0.f, false, 0.f);
          }
        }
        if (wave == 2) {
          float y[2] = {0.f, 0.f};
#pragma unroll
          for (int gg = 0; gg < 32; ++gg) {
            const float2 pr = *reinterpret_cast<const float2*>(xbuf + gg * 128 + 2 * lane);
            y[0] = had::fadd(y[0], pr.x);
            y[1] = had::fadd(y[1], pr.y);
          }
          hadw::reg_stage<2, 1>(y);
          hadw::lane_stages<2, 0, 6>(y, lane);
          const f16x2 svq = as_f16x2(svq_raw);
          s_qkv[2 * lane] = had::out_elem(y[0], kOutScaleH, true, (float)svq.x, false, 0.f, false, 0.f);
          s_qkv[2 * lane + 1] = had::out_elem(y[1], kOutScaleH, true, (float)svq.y, false, 0.f, false, 0.f);
        }
      }
      had::wg_barrier<true>();
      BSTAMP(5);
      ++hop;                                           // hand-off inside the head's group: partial states
      const uint32_t tagg = ebase | hop;
      auto unpack8h = [](const uint4& u, float o[8]) {
        const uint32_t ww[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f16x2 hh = as_f16x2(ww[i]);
          o[2 * i] = (float)hh.x;
          o[2 * i + 1] = (float)hh.y;
        }
      };
      auto rope8 = [&](const f16* vec, float o[8]) {
        float x[8], y[8];
        unpack8h(*reinterpret_cast<const uint4*>(vec + d0), x);
        const int dp = d0 < HD / 2 ? d0 + HD / 2 : d0 - HD / 2;
        unpack8h(*reinterpret_cast<const uint4*>(vec + dp), y);
        const float sgn = d0 < HD / 2 ? -1.f : 1.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (float)(f16)had::fadd(had::fmul(x[i], c8[i]), had::fmul(sgn * y[i], s8[i]));
      };
      {
        const float* cs = reinterpret_cast<const float*>(smem + B::kCs);
#pragma unroll
        for (int i = 0; i < 8; ++i) { c8[i] = cs[d0 + i]; s8[i] = cs[HD + d0 + i]; }
      }
      // single-query attention of head hd over positions [0, pos] (decode_glue.hip's arithmetic): 16 lanes per key,
      // 16 key groups with their own online-softmax state, merged through LDS
      float* s_m = reinterpret_cast<float*>(smem + B::kArea);
      constexpr int NST = 4;                           // online-softmax states that meet in LDS: one per wave (waves 0..3)
      float* s_l = s_m + NST;
      float* s_acc = s_l + NST;                        // [NST][HD + 4]
      float q8[8], kn[8], vn[8];
      float m = -INFINITY, lsum = 0.f, acc8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc8[i] = 0.f; kn[i] = 0.f; vn[i] = 0.f; }
      // one key: score against q (16 lanes: the group's sum on DPP moves, had::sum16_xor), online-softmax update of this
      // group's state.  A round scores all of its keys first (independent chains), then updates the state in key order:
      // the same operations on the same operands as key after key.
      auto score = [&](const float (&k8)[8]) -> float {
        float sc = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) sc = __builtin_fmaf(q8[i], k8[i], sc);
        return had::sum16_xor(sc);
      };
      auto update = [&](float sc, const float (&v8)[8]) {
        const float mn = fmaxf(m, sc);
        const float cc = __expf(m - mn), pp = __expf(sc - mn);
        lsum = __builtin_fmaf(lsum, cc, pp);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc8[i] = __builtin_fmaf(acc8[i], cc, had::fmul(pp, v8[i]));
        m = mn;
      };
      if (tid < 256) {
        rope8(s_qkv, q8);
#pragma unroll
        for (int i = 0; i < 8; ++i) q8[i] *= a.attn_scale;
        rope8(s_qkv + HD, kn);
        const uint4 vraw = *reinterpret_cast<const uint4*>(s_qkv + 2 * HD + d0);
        unpack8h(vraw, vn);
        // append the new row (StaticCache.update): once per KV head -- the first of its query heads, the workgroup whose turn the position is
        if (g == 0 && pos_ok && (hd % GQ) == 0 && part == (split ? (pos & (kParts - 1)) : 0)) {
          uint4 kr;
          kr.x = pack_f16(kn[0], kn[1]); kr.y = pack_f16(kn[2], kn[3]);
          kr.z = pack_f16(kn[4], kn[5]); kr.w = pack_f16(kn[6], kn[7]);
          *const_cast<uint4*>(reinterpret_cast<const uint4*>(kc + (size_t)pos * HD + d0)) = kr;
          *const_cast<uint4*>(reinterpret_cast<const uint4*>(vc + (size_t)pos * HD + d0)) = vraw;
        }
        // (round 5, as in decode_block.hip: the rounds visit cached rows only -- the new row, still in registers, is the LAST key
        //  of its group and scored behind them; as a case inside the rounds it cost sixteen register moves per key and kept the
        //  scores off v_fma_mix)
        auto round = [&](const uint4 (&kr)[U], const uint4 (&vr)[U], int i0) {
          float k8[U][8], v8[U][8], sc[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            unpack8h(kr[u], k8[u]);
            unpack8h(vr[u], v8[u]);
            sc[u] = score(k8[u]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (part + nparts * (i0 + u * NG) < pos) update(sc[u], v8[u]);
        };
        const int n_loc = pos > part ? (pos - part + nparts - 1) / nparts : 0;       // local indices of this workgroup's cached rows
        for (int ib = 0; ib < n_loc; ib += 2 * NG * U) {
          round(kr0, vr0, ib + g);
          if (ib + 2 * NG * U < n_loc) load_round(kr0, vr0, ib + g + 2 * NG * U);
          if (ib + NG * U < n_loc) {
            round(kr1, vr1, ib + g + NG * U);
            if (ib + 3 * NG * U < n_loc) load_round(kr1, vr1, ib + g + 3 * NG * U);
          }
        }
        if (part == (split ? (pos & (kParts - 1)) : 0) && g == (((pos - part) / nparts) & (NG - 1))) update(score(kn), vn);
        // the four key groups of a wave (its rows of 16 lanes) merge in registers: v_permlane16_swap / 32_swap of a value with
        // ITSELF hands every lane both partners' copies, so both sides compute the same state; four states meet in LDS
        auto merge2 = [&](auto swap) {
          const auto tm = swap(as_u32(m)), tl = swap(as_u32(lsum));
          const float mA = as_f32((uint32_t)tm[0]), mB = as_f32((uint32_t)tm[1]);
          const float M = fmaxf(mA, mB);
          const float wA = mA == -INFINITY ? 0.f : __expf(mA - M), wB = mB == -INFINITY ? 0.f : __expf(mB - M);
          lsum = __builtin_fmaf(as_f32((uint32_t)tl[1]), wB, had::fmul(as_f32((uint32_t)tl[0]), wA));
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const auto ta = swap(as_u32(acc8[i]));
            acc8[i] = __builtin_fmaf(as_f32((uint32_t)ta[1]), wB, had::fmul(as_f32((uint32_t)ta[0]), wA));
          }
          m = M;
        };
        merge2([](uint32_t x) { return __builtin_amdgcn_permlane16_swap(x, x, false, false); });
        merge2([](uint32_t x) { return __builtin_amdgcn_permlane32_swap(x, x, false, false); });
        if (lane < LPK) {
          if (lane == 0) { s_m[wave] = m; s_l[wave] = lsum; }
#pragma unroll
          for (int i = 0; i < 8; ++i) s_acc[wave * (HD + 4) + d0 + i] = acc8[i];
        }
      }
      had::wg_barrier<false>();                        // (full fence: the cache rows' loads are the compiler's own)
      own_ring();
      f16* s_a = s_qkv + 3 * HD;
      float pM = -INFINITY, pL = 0.f, pO = 0.f;
      if (tid < HD) {
        for (int g2 = 0; g2 < NST; ++g2) pM = fmaxf(pM, s_m[g2]);
        for (int g2 = 0; g2 < NST; ++g2) {
          const float ww = s_m[g2] == -INFINITY ? 0.f : __expf(s_m[g2] - pM);
          pL = __builtin_fmaf(s_l[g2], ww, pL);
          pO = __builtin_fmaf(s_acc[g2 * (HD + 4) + tid], ww, pO);
        }
      }
      if (split) {
        uint64_t* mine_p = pbuf + ((size_t)hd * kParts + part) * kPartGran;
        if (tid < HD) esync::st_granule(mine_p + tid, as_u32(pO), tagg);
        if (tid == 0) { esync::st_granule(mine_p + HD, as_u32(pM), tagg); esync::st_granule(mine_p + HD + 1, as_u32(pL), tagg); }
        if (head_wg) {
          constexpr int PIECES = kParts * kPartGran / 2;
          float* s_p = reinterpret_cast<float*>(smem + B::kArea);
          const uint64_t* srcp = pbuf + (size_t)hd * kParts * kPartGran;
          u32x4_t pp;
          uint32_t spins = 0;
          const int i = tid < PIECES ? tid : 0;
          const int within = (2 * i) % kPartGran;
          for (;;) {
            esync::ld16(pp, srcp + 2 * i);
            esync::drain();
            esync::own(pp);
            const bool ok = within >= HD + 2 || (pp.y == tagg && (within + 1 >= HD + 2 || pp.w == tagg));
            if (esync::spin_step(ok, spins, ctl + 1, 0x8000u + (uint32_t)w)) break;
          }
          had::wg_barrier<true>();                     // s_m / s_l / s_acc have been read by everybody
          if (tid < PIECES) *reinterpret_cast<uint2*>(s_p + 2 * tid) = make_uint2(pp.x, pp.z);
          own_ring();
          had::wg_barrier<true>();
          if (tid < HD) {
            float M = -INFINITY, Lsum = 0.f, o = 0.f;
            for (int q2 = 0; q2 < kParts; ++q2) M = fmaxf(M, s_p[q2 * kPartGran + HD]);
            for (int q2 = 0; q2 < kParts; ++q2) {
              const float mq = s_p[q2 * kPartGran + HD];
              const float ww = mq == -INFINITY ? 0.f : __expf(mq - M);
              Lsum = __builtin_fmaf(s_p[q2 * kPartGran + HD + 1], ww, Lsum);
              o = __builtin_fmaf(s_p[q2 * kPartGran + tid], ww, o);
            }
            s_a[tid] = pos_ok ? (f16)(o / Lsum) : __builtin_bit_cast(f16, (unsigned short)0x7e00);
          }
        }
      } else if (tid < HD) {
        s_a[tid] = pos_ok ? (f16)(pO / pL) : __builtin_bit_cast(f16, (unsigned short)0x7e00);
      }
      had::wg_barrier<true>();
#if QUIP_GQA_OHEAD
      if (head_wg && tid < 64) {
        // x = a (.) SU_o, its 128-point transform inside this wave (lane l: elements 2 l, 2 l + 1 before and after), two per granule
        const f16x2 av = as_f16x2(*reinterpret_cast<const uint32_t*>(s_a + 2 * tid)), su2 = as_f16x2(suo_raw);
        float y[2] = {had::fmul((float)av.x, (float)su2.x), had::fmul((float)av.y, (float)su2.y)};
        hadw::reg_stage<2, 1>(y);
        hadw::lane_stages<2, 0, 6>(y, lane);
        uint32_t w0, w1;
        esync::pack20x2(y[0], y[1], ebase | (hop + 1u), w0, w1);
        esync::st_granule(za + hd * 64 + tid, w0, w1);
      }
#else
      if (head_wg && tid < 64) {
        const uint32_t pr = *reinterpret_cast<const uint32_t*>(s_a + 2 * tid);
        esync::st_granule(za + hd * 64 + tid, pr, ebase | (hop + 1u));
      }
#endif
      ++hop;                                           // hand-off: attention output
    } else {
      hop += 2;
    }
    BSTAMP(6);

    // ================= o: input side (no norm), product, hand-off =========================================================
    rederive();
#if QUIP_GQA_OHEAD
    if (!so) {
      // thread (wave, lane): columns [4 jq, +4) of heads 4 (lane >> 2) + r, r < 4 -- one 16-byte piece (two granules) per head
      const int jq = 4 * wave + (lane & 3), hg = lane >> 2;
      const uint32_t t16 = (ebase | hop) & 0xffffu;
      {
        // (the attention output is ~10 us away: the wait is spent on ONE granule per wave, the gather behind it checks every piece)
        uint32_t sp = 0;
        u32x2_t f;
        const uint64_t* g1 = za + (size_t)((8 * w + wave) & (NH - 1)) * 64;
        for (;;) {
          esync::ld8(f, g1);
          esync::drain();
          esync::own(f);
          if (esync::spin_step((f.y >> 16) == t16, sp, ctl + 1, 0x6100u + (uint32_t)w)) break;
        }
      }
      u32x4_t p[4];
      {
        uint32_t spins = 0;
        const uint64_t* src = za + (size_t)(4 * hg) * 64 + 2 * jq;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = u32x4_t{0u, 0u, 0u, 0u};
        bool ok = false;
        for (;;) {
          if (!ok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) esync::ld16_keep(p[r], src + 64 * r);
          }
          esync::drain();
          bool now = true;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            esync::own(p[r]);
            now = now && (p[r].y >> 16) == t16 && (p[r].w >> 16) == t16;
          }
          ok = now;
          if (esync::spin_step(ok, spins, ctl + 1, 0x6000u + (uint32_t)w)) break;
        }
        own_ring();
      }
      BSTAMP(7);
      float v[16];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        esync::unpack20x2(p[r].x, p[r].y, v[4 * r], v[4 * r + 1]);
        esync::unpack20x2(p[r].z, p[r].w, v[4 * r + 2], v[4 * r + 3]);
      }
      {
        // the norm bound: |H_8192 x|_inf <= sqrt(8192) |x|_2, and sum y^2 = 128 sum x^2 (the heads' transforms are orthogonal x sqrt 128)
#pragma clang fp contract(off)
        float n0 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) n0 = __builtin_fmaf(v[r], v[r], n0);
        n0 = had::wave_reduce_to_lane63<false>(n0);
        if (lane == 63) red[wave] = n0;
      }
      hadw::reg_stages<16, 4>(v);                        // head index bits 0, 1 (registers 4 r + e)
      hadw::lane_stages<16, 2, 6>(v, lane);              // head index bits 2 .. 5 (lane bits 2 .. 5)
      const float sco = Ld.sc[3];
      had::wg_barrier<true>();                         // the sums; and the heads' attention scratch in the area has been read
      float q0 = red[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) q0 = had::fadd(q0, red[i]);
      const int sh = had::shift_for(sqrtf(q0 * (1.f / 128.f)) * kSqrtH * fabsf(sco) * 1.0625f);
      {
#pragma clang fp contract(off)
        const float s2 = had::fmul(sco, as_f32((uint32_t)(sh + 127) << 23));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // X[head 4 hg + r][4 jq .. + 3]: natural dword 32 head + jq of a plane
          const float vv[4] = {v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]};
          uint32_t dh, dm, dl;
          hadw::digit_words_magic(vv, s2, dh, dm, dl);
          const int kd = 32 * (4 * hg + r) + jq;
          const uint32_t off = NIB ? (uint32_t)((kd & 1) * B::HOH + 4 * (kd >> 1)) : (uint32_t)(4 * kd);
          *reinterpret_cast<uint32_t*>(smem + B::kArea + off) = dh;
          *reinterpret_cast<uint32_t*>(smem + B::kArea + B::PSH + off) = dm;
          *reinterpret_cast<uint32_t*>(smem + B::kArea + 2 * B::PSH + off) = dl;
        }
      }
      if (tid == 0) shs[2] = sh;
      had::wg_barrier<true>();
    }
#else
    if (!so) {
      u32x4 psu[2];
      load16(Ld.su[3], psu);                           // natural order
      float v[1][16], suf[16];
      prepoll(za + (size_t)((8 * w + wave) & (NH - 1)) * 64, ebase | hop, 0x6100u);      // (the attention output is ~10 us away)
      gather16(za, ebase | hop, 0x6000u, v[0]);
      asm volatile("" : "+v"(psu[0]), "+v"(psu[1]));
      BSTAMP(7);
      unpack16(psu, suf);
#pragma unroll
      for (int k = 0; k < 16; ++k) v[0][k] = had::fmul(v[0][k], suf[k]);
      {
        // (round 5: the norm bound, as on the other edges -- the sum of squares of the transform's input, read behind its barriers)
#pragma clang fp contract(off)
        float n0 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) n0 = __builtin_fmaf(v[0][r], v[0][r], n0);
        n0 = had::wave_reduce_to_lane63<false>(n0);
        if (lane == 63) red[wave] = n0;
      }
      hadw::fwd<13, 1, true>(v, xbuf, tid);
      const float sco = Ld.sc[3];
      had::wg_barrier<true>();                         // the transform's last reads of the exchange buffer: the planes land on it
      float q0 = red[0];
#pragma unroll
      for (int i = 1; i < 8; ++i) q0 = had::fadd(q0, red[i]);
      const int sh = had::shift_for(sqrtf(q0) * kSqrtH * fabsf(sco) * 1.0625f);
      planes_str(v[0], sco, sh, (uint32_t)B::kArea);
      if (tid == 0) shs[2] = sh;
      had::wg_barrier<true>();
    }
#endif
    BSTAMP(8);
    rederive();
    first(IC<SQ_O>{}, true);
    group(IC<SQ_O>{}, IC<2>{}, FalseC{}, xaddr((uint32_t)B::kArea, B::PSH, 0), B::AO);
    group(IC<SQ_O + 2>{}, IC<2>{}, TrueC{}, xaddr((uint32_t)B::kArea, B::PSH, 1), B::AO);
    had::wg_barrier<true>();
    ++hop;                                             // hand-off: z_o
    if (!so) publish(zo, 16 * w, B::AO, 16, shs[2], ebase | hop);
    had::wg_barrier<true>();
    zero_acc(B::AO, 32);
    BSTAMP(9);

    // ================= o's output side + residual, RMSNorm, input transforms of gate / up; their products ===================
    rederive();
    uint32_t Pg[kPreGU > 0 ? kPreGU : 1][16];          // (gate / up's first items: decoded inside the wait for z_o)
    static_for<kPreGU>([&](auto ic) { predec(IC<SQ_GU + decltype(ic)::value>{}, ic, Pg[decltype(ic)::value]); });
    if (!so) edge(IC<2>{}, zo, ebase | hop, 0x7000u, Ld.sv[3], Ld.ln[1], Ld.su[4 + mgu], Ld.su[4 + mgu], Ld.sc[4 + mgu], Ld.sc[4 + mgu], false, 3, 23);
    BSTAMP(10);
    rederive();
    {
      // 14 items per K span: chunks k = 0..6 of the first 16 columns (accumulator rows AGU + 16 k + i), then of the second 16
      // (AGU + 112 + 16 k + i) -- one set of A fragments per span
      const uint32_t pg = (uint32_t)B::kArea;
      static_for<kPreGU>([&](auto ic) { refill(IC<SQ_GU + decltype(ic)::value>{}); });
      first(IC<SQ_GU>{}, true);
      group_pre(IC<SQ_GU>{}, IC<14>{}, FalseC{}, IC<kPreGU>{}, Pg, xaddr(pg, B::PSH, 0), B::AGU);
      group(IC<SQ_GU + 14>{}, IC<14>{}, TrueC{}, xaddr(pg, B::PSH, 1), B::AGU);
    }
    had::wg_barrier<true>();
    BSTAMP(11);

    // ================= the MLP edge =========================================================================================
    int sh_d = 0;
    uint32_t Pd[kPreD > 0 ? kPreD : 1][16];            // (down's first items: decoded inside the wait for the chunk owners / the owners' inbox)
    rederive();
    float* zcol = reinterpret_cast<float*>(smem + B::kZcol);
    const float* mixf = reinterpret_cast<const float*>(smem + B::kMix);
    uint32_t tag1 = 0u, tag2 = 0u;
    if (!so) {
    if (tid < 224) {
      const int* s3 = accs + (B::AGU + tid) * 4;
      const float f = __builtin_fmaf((float)s3[0], 65536.f, __builtin_fmaf((float)s3[1], 256.f, (float)s3[2]));
      zcol[tid] = (float)(f16)(f * unscale_of(shs[3], kUnsc));      // [column group][k][i]: one matrix, one exponent
    }
    had::wg_barrier<true>();
    zero_acc(B::AGU, 224);
    ++hop;                                             // hand-off: columns -> chunk owners
    tag1 = ebase | hop;
    if (tid < 224) {
      // output (this workgroup's matrix mgu, chunk k', column 32 (w & 127) + 16 cg + i): t = sum_k had[k'][k] z[k][column]
      const int cg = tid / 112, rem = tid - 112 * cg, kq = rem >> 4, i = rem & 15;
      const float* hs = mixf + (mgu * 7 + kq) * 8;
      const float* zz = zcol + cg * 112 + i;
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < FK; ++k) t = __builtin_fmaf(hs[k], zz[16 * k], t);
      const float tn = __shfl_down(t, 1, 64);
#if QUIP_GQA_INBOX_PACK
      if ((i & 1) == 0) {
        // {m_a (20) | m_b low 12, m_b high 8 | exponent 8 | tag 16}: m = rint(value 2^(145 - e)), e = the larger biased exponent
        uint32_t w0, w1;
        esync::pack20x2(t, tn, tag1, w0, w1);
        esync::st_granule(inbox + ((size_t)(kq * 2 + mgu) * (FL / 2) + 16 * (w & 127) + 8 * cg + (i >> 1)), w0, w1);
      }
#else
      if ((i & 1) == 0)
        esync::st_granule2(inbox + ((size_t)(kq * 2 + mgu) * FL + 32 * (w & 127) + 16 * cg + i), as_u32(t), as_u32(tn), tag1);
#endif
    }
    ++hop;                                             // hand-off: rows -> everybody
    tag2 = ebase | hop;
    BSTAMP(12);
    }
    // (behind the publication of the columns: everybody waits from here on -- the owners for their inbox)
    static_for<kPreD>([&](auto ic) { predec(IC<SQ_D + decltype(ic)::value>{}, ic, Pd[decltype(ic)::value]); });
    if (!so) {
    if (w < FK) {
      // chunk owner w: 4096 values of gate and of up (fp32 payloads) -> transforms, SV, SiLU product, SU, down's transform
      const f16* svg = Ld.sv[4] + (size_t)w * FL;
      const f16* svu = Ld.sv[5] + (size_t)w * FL;
      const f16* sud = Ld.su[6] + (size_t)w * FL;
      uint16_t psg[8], psu_[8], psd[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        psg[j] = reinterpret_cast<const uint16_t*>(svg)[tid + 512 * j];
        psu_[j] = reinterpret_cast<const uint16_t*>(svu)[tid + 512 * j];
        psd[j] = reinterpret_cast<const uint16_t*>(sud)[tid + 512 * j];
      }
      float v[2][8];
#if QUIP_GQA_INBOX_PACK
      {
        // this thread's 8 values of each vector = 4 granules = two 16-byte pieces
        u32x4_t pc[4];
        uint32_t spins = 0;
        const uint32_t t16 = tag1 & 0xffffu;
        const uint64_t* src = inbox + (size_t)(w * 2) * (FL / 2) + 4 * tid;
        for (;;) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) esync::ld16(pc[jj], src + (jj >> 1) * (FL / 2) + 2 * (jj & 1));
          esync::drain();
          bool ok = true;
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            esync::own(pc[jj]);
            ok = ok && (pc[jj].y >> 16) == t16 && (pc[jj].w >> 16) == t16;
          }
          if (esync::spin_step(ok, spins, ctl + 1, 0x1000u + (uint32_t)w)) break;
        }
        own_ring();
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(psg[j]), "+v"(psu_[j]), "+v"(psd[j]));
        auto unpack2 = [](uint32_t w0, uint32_t w1, float& a_, float& b_) { esync::unpack20x2(w0, w1, a_, b_); };
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          unpack2(pc[jj].x, pc[jj].y, v[jj >> 1][4 * (jj & 1)], v[jj >> 1][4 * (jj & 1) + 1]);
          unpack2(pc[jj].z, pc[jj].w, v[jj >> 1][4 * (jj & 1) + 2], v[jj >> 1][4 * (jj & 1) + 3]);
        }
      }
#else
      {
        u32x4_t pc[8];
        uint32_t spins = 0;
        const uint64_t* src = inbox + (size_t)(w * 2) * FL + 8 * tid;
        for (;;) {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) esync::ld16(pc[jj], src + (jj >> 2) * FL + 2 * (jj & 3));
          esync::drain();
          bool ok = true;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            esync::own(pc[jj]);
            ok = ok && pc[jj].y == tag1 && pc[jj].w == tag1;
          }
          if (esync::spin_step(ok, spins, ctl + 1, 0x1000u + (uint32_t)w)) break;
        }
        own_ring();
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(psg[j]), "+v"(psu_[j]), "+v"(psd[j]));
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          v[jj >> 2][2 * (jj & 3)] = as_f32(pc[jj].x);
          v[jj >> 2][2 * (jj & 3) + 1] = as_f32(pc[jj].z);
        }
      }
#endif
      BSTAMP(28);
      hadw::fwd<12, 2, true>(v, xbuf, tid);
      BSTAMP(29);
      float e[1][8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float og = (float)had::out_elem(v[0][j], 1.f / 64.f, true, (float)__builtin_bit_cast(f16, psg[j]), false, 0.f, false, 0.f);
        const float ou = (float)had::out_elem(v[1][j], 1.f / 64.f, true, (float)__builtin_bit_cast(f16, psu_[j]), false, 0.f, false, 0.f);
        e[0][j] = had::fmul(had::fmul(ou, had::silu(og)), (float)__builtin_bit_cast(f16, psd[j]));
      }
      BSTAMP(30);
      hadw::rev<12, 1, true>(e, xbuf, tid);
      BSTAMP(31);
      // the chunk goes out as 24-bit fixed point against its own maximum, TWO values per granule {a | b << 24, b >> 8 | tag16 << 16}
      // (half the bytes of the sweep every workgroup makes; 2^-23 of the maximum: finer than the digits made of it).  The tag's
      // low 16 bits suffice here: every granule of the rows is rewritten in every block, so a stale one carries the tag of the
      // block before, which differs in the hand-off index.
      constexpr float kPre = 1.f / 64.f;
      float mxr = 0.f;
      float vv[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        vv[r] = e[0][r] * kPre;
        const float av = fabsf(vv[r]);
        mxr = fmaxf(mxr, av == av ? av : __builtin_inff());
      }
      mxr = wg_max(mxr);
      const float inv = (mxr > 0.f && mxr < __builtin_inff()) ? 8388607.f / mxr : 0.f;
      uint32_t pk[8];
#pragma unroll
      for (int r = 0; r < 8; ++r)       // (clamped: maximum * fl(8388607 / maximum) may round to 2^23, which would wrap to -2^23)
        pk[r] = (uint32_t)min(max((int)__builtin_rintf(vv[r] * inv), -8388607), 8388607);
      const uint32_t t16 = (tag2 & 0xffffu) << 16;
      uint64_t* dst = frow + (size_t)w * (FL / 2) + 4 * tid;
#pragma unroll
      for (int r = 0; r < 8; r += 4) {
        const u32x4_t piece = {(pk[r] & 0xffffffu) | (pk[r + 1] << 24), ((pk[r + 1] >> 8) & 0xffffu) | t16,
                               (pk[r + 2] & 0xffffffu) | (pk[r + 3] << 24), ((pk[r + 3] >> 8) & 0xffffu) | t16};
        esync::st_payload16(dst + r / 2, piece);
      }
      // (eight copies in eight lines: 256 workgroups polling ONE line delay the owners' stores to it)
      if (tid < 8) esync::st_granule(rowmax + 16 * tid + w, as_u32(mxr), tag2);
      had::wg_barrier<true>();
    }
    BSTAMP(13);
    // everybody: the owners' maxima -> block exponent of down's input; then the rows, mixed and turned into digits on the fly
    const float in_scale = Ld.sc[6] * 64.f;
    if (wave == 0) {
      uint32_t spins0 = 0;
      u32x2_t f;
      for (;;) {
        esync::ld8(f, rowmax + 16 * (w & 7) + (lane < FK ? lane : 0));
        esync::drain();
        esync::own(f);
        if (esync::spin_step(f.y == tag2, spins0, ctl + 1, 0x3000u + (uint32_t)w)) break;
      }
      if (lane < FK) { red[32 + lane] = as_f32(f.x); red[40 + lane] = as_f32(f.x) * (1.f / 8388607.f); }
    }
    had::wg_barrier<false>();
    own_ring();
    {
      float bound = 0.f;
#pragma unroll
      for (int kp = 0; kp < FK; ++kp) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < FK; ++k) s = __builtin_fmaf(fabsf(mixf[(14 + kp) * 8 + k]), red[32 + k], s);
        bound = fmaxf(bound, s == s ? s : __builtin_inff());
      }
      sh_d = had::shift_for(had::fmul(bound, fabsf(in_scale)));
    }
    BSTAMP(14);
    {
      const float s2 = had::fmul(in_scale, as_f32((uint32_t)(sh_d + 127) << 23));
      // the 7 x 7 factors of the mix with their scales folded in -- row kp of down.had_left^T times (2^sh wscale) times the
      // chunk's step M_k / (2^23 - 1) -- once per block into LDS (zcol: free since the columns went out); the sweep reads a row
      // as two broadcast 16-byte pieces.  (They used to be rebuilt per row and chunk through v_readfirstlane: 3 VALU per factor,
      // 294 per thread and block, a third of the sweep's arithmetic.)
      if (tid < FK * 8) zcol[tid] = (tid & 7) < FK ? had::fmul(had::fmul(mixf[(14 + (tid >> 3)) * 8 + (tid & 7)], s2), red[40 + (tid & 7)]) : 0.f;
      had::wg_barrier<true>();
      uint8_t* pl = reinterpret_cast<uint8_t*>(smem + B::kArea);
      const uint32_t t16 = tag2 & 0xffffu;
      // columns [4 (t + 512 c), +4) of the seven rows = one 16-byte piece (two granules) per row; chunk 1 is in flight while
      // chunk 0 is mixed
      u32x4_t p0[FK], p1[FK];
      auto request = [&](u32x4_t (&p)[FK], int c) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < FK; ++k) esync::ld16(p[k], frow + (size_t)k * (FL / 2) + 2 * (tid + 512 * c));
      };
      auto landed = [&](u32x4_t (&p)[FK]) __attribute__((always_inline)) -> bool {          // after a drain
        bool ok = true;
#pragma unroll
        for (int k = 0; k < FK; ++k) {
          esync::own(p[k]);
          ok = ok && (p[k].y >> 16) == t16 && (p[k].w >> 16) == t16;
        }
        return ok;
      };
      auto poll = [&](u32x4_t (&p)[FK], int c) __attribute__((always_inline)) {            // request, wait, check -- until every tag is there
        uint32_t spins = 0;
        for (;;) {
          request(p, c);
          esync::drain();
          if (esync::spin_step(landed(p), spins, ctl + 1, 0x2000u + (uint32_t)w)) break;
        }
      };
      auto mix = [&](const u32x4_t (&p)[FK], int c) __attribute__((always_inline)) {
        float e[FK][4];
#pragma unroll
        for (int k = 0; k < FK; ++k) {
          e[k][0] = (float)((int)(p[k].x << 8) >> 8);
          e[k][1] = (float)((int)(__builtin_amdgcn_alignbit(p[k].y, p[k].x, 24) << 8) >> 8);
          e[k][2] = (float)((int)(p[k].z << 8) >> 8);
          e[k][3] = (float)((int)(__builtin_amdgcn_alignbit(p[k].w, p[k].z, 24) << 8) >> 8);
        }
        const int col = 4 * (tid + 512 * c);
#pragma unroll
        for (int kp = 0; kp < FK; ++kp) {
          const float4 ra = *reinterpret_cast<const float4*>(zcol + kp * 8), rb = *reinterpret_cast<const float4*>(zcol + kp * 8 + 4);
          const float rt[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          f32x2 x01 = {0.f, 0.f}, x23 = {0.f, 0.f};
#pragma unroll
          for (int k = 0; k < FK; ++k) {
            x01 = __builtin_elementwise_fma(f32x2{rt[k], rt[k]}, f32x2{e[k][0], e[k][1]}, x01);
            x23 = __builtin_elementwise_fma(f32x2{rt[k], rt[k]}, f32x2{e[k][2], e[k][3]}, x23);
          }
          const int X[4] = {(int)__builtin_rintf(x01.x), (int)__builtin_rintf(x01.y), (int)__builtin_rintf(x23.x), (int)__builtin_rintf(x23.y)};
          uint32_t dh, dm, dl;
          digit_words(X, dh, dm, dl);
          // (nibble mode: natural dword kp * 1024 + tid + 512 c of a plane = dword (that >> 1) of its "lo" (even) / "hi" (odd) half)
          const int off = NIB ? (tid & 1) * B::HOD + kp * (FL / 2) + (tid >> 1) * 4 + 1024 * c : kp * FL + col;
          *reinterpret_cast<uint32_t*>(pl + off) = dh;
          *reinterpret_cast<uint32_t*>(pl + B::PSD + off) = dm;
          *reinterpret_cast<uint32_t*>(pl + 2 * B::PSD + off) = dl;
        }
      };
      // (measured and dropped, round 6: chunk 0 requested in front of the maxima's barrier -- waves 1..7 waiting on one granule of a
      //  row meanwhile -- takes 1.5K clocks off this sweep, but its 28 registers are those of down's two pre-decoded items:
      //  187.6-187.8 against 188.4 tok/s on the same box; and a first try through the caches instead of at device scope: no change)
      poll(p0, 0);
      own_ring();
      request(p1, 1);                                  // (no request may cross a loop's back edge: the first try of chunk 1 is straight-line code)
      mix(p0, 0);
      esync::drain();
      if (!__all(landed(p1))) poll(p1, 1);
      mix(p1, 1);
    }
    had::wg_barrier<true>();
    BSTAMP(15);
    }
    // ================= down's product ====================================================================================
    rederive();
    {
      Acc acc0 = kAcc0, acc1 = kAcc0;
      i32x4 sx = {0, 0, 0, 0};
      static_for<kPreD>([&](auto ic) { refill(IC<SQ_D + decltype(ic)::value>{}); });
      first(IC<SQ_D>{}, true);
      static_for<7>([&](auto ic) {
        constexpr int I = decltype(ic)::value;
        i32x4 A[NA];
        fragments(xaddr((uint32_t)B::kArea, B::PSD, I), A, sx);
        if constexpr (2 * I < kPreD) mul_raw(Pd[2 * I < kPreD ? 2 * I : 0], A, acc0);
        else item(IC<SQ_D + 2 * I>{}, TrueC{}, A, acc0, true, true);
        if constexpr (2 * I + 1 < kPreD) mul_raw(Pd[2 * I + 1 < kPreD ? 2 * I + 1 : 0], A, acc1);
        else item(IC<SQ_D + 2 * I + 1>{}, TrueC{}, A, acc1, true, I < 6);          // (behind the last one: the first filler)
      });
      add_rows(acc0, sx, B::AD);
      add_rows(acc1, sx, B::AD + 16);
      i32x4 A0[NA];
#pragma unroll
      for (int t = 0; t < NA; ++t) A0[t] = i32x4{0, 0, 0, 0};
      Acc accf = kAcc0;
      // the two fillers keep the ring's period at 54 (nibble mode: SQ_F was waited for and refilled behind down's last item)
      if constexpr (PIPE) {
        pre(IC<SQ_F + 1>{}, false);
      } else {
        consume(IC<SQ_F>{}, A0, accf, false);
        consume(IC<SQ_F + 1>{}, A0, accf, false);
      }
    }
    had::wg_barrier<true>();
    ++hop;                                             // hand-off: z_d
    if (!so) publish(zd, 16 * w, B::AD, 16, sh_d, ebase | hop);
    had::wg_barrier<true>();
    zero_acc(B::AD, 32);
    BSTAMP(16);

    sv_prev = Ld.sv[6];
    // the loop's back edge: nothing in flight (the requests of the last items would be waited for by the next gather anyway)
    esync::drain();
    own_ring();
    BSTAMP(17);
  }
  // output side of the last block's down_proj + residual -> h
  rederive();
  if (!so) edge(IC<0>{}, zd, ebase | hop, 0x4000u, sv_prev, nullptr, nullptr, nullptr, 0.f, 0.f, false, 0, -1);
#if QUIP_GQA_WAITSTAT
  if (a.dbg != nullptr && lane == 0) {
    uint64_t* o = a.dbg + (size_t)(w * 8 + wave) * 4;
    o[0] = ws_wait; o[1] = __builtin_amdgcn_s_memtime() - ws_start; o[2] = __builtin_amdgcn_s_memrealtime() - ws_rstart; o[3] = 0;
  }
#endif
  // ---- h_out (natural order) -------------------------------------------------------------------------------------------------
  if (w == 0) {
    // A launch in which a wait gave up (ctl[1] != 0) has no result: h_out is all NaN then, and ctl[2] keeps position + 1 of
    // the FIRST such launch (decode_block.hip)
    uint32_t e, fp;
    esync::ld4(e, ctl + 1);
    esync::ld4(fp, ctl + 2);
    esync::drain();
    esync::own(e);
    esync::own(fp);
    const bool failed = __builtin_amdgcn_readfirstlane((int)e) != 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      reinterpret_cast<uint16_t*>(a.h_out)[tid + 512 * (2 * j)] = failed ? (uint16_t)0x7e00 : (uint16_t)(hreg[j] & 0xffffu);
      reinterpret_cast<uint16_t*>(a.h_out)[tid + 512 * (2 * j + 1)] = failed ? (uint16_t)0x7e00 : (uint16_t)(hreg[j] >> 16);
    }
    if (tid == 0) {
      if (failed && fp == 0u) esync::st_word(ctl + 2, (uint32_t)pos + 1u);
      esync::st_word(ctl, ebase >> 10);
      // the tag's generation field has 22 bits: ask for a fresh workspace before it wraps (the next launch answers NaN at once)
      if (!failed && (ebase >> 10) >= (1u << 22) - 2u) esync::st_word(ctl + 1, 0xE000u);
    }
  }
#undef BSTAMP
}

}  // namespace

size_t block_engine_gqa_workspace_bytes() { return kWsBytes; }
size_t block_engine_gqa_layer_bytes() { return sizeof(GLayer); }

bool block_engine_gqa_supported(int hidden, int heads, int kv_heads, int head_dim, int n_ffn, int K) {
  return hidden == HID && heads == NH && kv_heads == NKV && head_dim == HD && n_ffn == NFFN && K == FK && device_cu_count_strict() >= NWG;
}

int block_engine_gqa_launch(const BlockEngineArgs& in, hipStream_t stream) {
  if (in.n_layers < 1 || in.n_layers > 146) return QUIP_ERR_BAD_SHAPE;     // 7 hand-offs per block, 10-bit counter
  if (in.codebook != 0) return QUIP_ERR_UNSUPPORTED;
  GArgs a;
  a.layers = reinterpret_cast<const GLayer*>(in.layers);
  a.h_in = reinterpret_cast<const f16*>(in.h_in);
  a.h_out = reinterpret_cast<f16*>(in.h_out);
  a.pos = reinterpret_cast<const int64_t*>(in.pos);
  a.cos = in.cos; a.sin = in.sin;
  a.grid = reinterpret_cast<const uint64_t*>(in.grid);
  a.ws = reinterpret_cast<char*>(in.workspace);
  a.dbg = reinterpret_cast<uint64_t*>(in.dbg);
  a.n_layers = in.n_layers; a.max_len = in.max_len; a.dbg_layer = in.dbg_layer;
  a.rms_eps = in.rms_eps; a.attn_scale = in.attn_scale;
  static DynLdsCache cache;
  if (ensure_dyn_lds(cache, reinterpret_cast<const void*>(decode_block_gqa_kernel), GLds::kBytes) != QUIP_OK) return QUIP_ERR_LAUNCH;
  static ResidencyCache resident;
  if (!persistent_grid_fits(resident, reinterpret_cast<const void*>(decode_block_gqa_kernel), kThreads, GLds::kBytes, NWG))
    return QUIP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(decode_block_gqa_kernel, dim3(NWG), dim3(kThreads), GLds::kBytes, stream, a);
  if (hipGetLastError() != hipSuccess) return QUIP_ERR_LAUNCH;
  return in.dbg_layer == -2 ? QUIP_NO_RESULT : QUIP_OK;      // (measurement mode: h_out holds no hidden state)
}

}  // namespace quip
