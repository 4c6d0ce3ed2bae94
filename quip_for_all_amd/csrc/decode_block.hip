// Persistent decode engine for gfx950, stage 2: whole decoder blocks -- any number of consecutive ones -- in ONE launch.
//
// One token (bs = 1) through blocks [0, n_layers) of a Llama-architecture decoder whose seven projections are E8P12
// QuantLinear modules (qlinear.py:87-115 each; the block itself is the reference metric driver's HF LlamaDecoderLayer,
// example_generate.py:28-33: RMSNorm, q / k / v, rotary embedding, static KV cache, attention, o, residual, RMSNorm,
// gate / up, SiLU product, down, residual).  The stage-wise step issues 9 launches per block; each one rebuilds its
// decode tables, starts its weight stream only once its input exists and pays a drain and a cold prologue, while the
// arithmetic between two matrix-vector products is a few thousand flops.  Here:
//   * 256 workgroups (one per CU) stay resident for all blocks; the E8P decode tables are built once per launch;
//   * the codes of every projection are requested one dependency edge ahead of their use and wait in registers (nine
//     static 2 x 16-byte slots per wave: q k v / gate x 3 | o / up x 3 | down x 3), in bursts placed right AFTER a
//     hand-off has completed and sized to land before the next one polls, so that no poll queues behind them;
//   * every product's 4096-vector goes to ALL workgroups as 8-byte {2 x fp16, tag} granules (engine_sync.hip.h) and
//     every workgroup repeats the small transforms between two products on its own copy (RMSNorm, SU / SV, 4096-point
//     Walsh-Hadamard: the functions of had_device.hip.h, same operations in the same order as the stand-alone
//     kernels, so the same bits); the 11008-wide MLP edge is the distributed two-hop computation of decode_engine.hip;
//   * attention runs on the workgroup that owns the first 16 rows of its head (the other seven wait for the result); from 128
//     positions on all eight take every eighth position each and merge their partial softmax states through one more hand-off.
// Codebooks: E8P12 (32 copies of the abs table, 16 of the sign table), D4 (the one-table mode of e8p_gemv_core.hip.h) and
// E8P12RVQ4B (virtual rows of twice the width against x' = [s x_g | x_g]: twice the digits and items, eight of the nine slots).
// Row ownership: q / k / v / o / down rows [16 w, 16 w + 16); gate / up rows k * 256 + w, k = 0..42 (column w of the
// (43, 256) view, decode_engine.hip).
//
// Shape: hidden = 4096 = 256 x 16, heads x 128 = 4096 (multi-head attention), n_ffn = 43 x 256 -- Llama-2-7B.  Other
// shapes stay on the stage-wise step.
// Liveness: all 256 workgroups must be resident (one per CU by LDS footprint; nothing else may run on the device);
// every wait is bounded and a launch that gives up leaves a code in ctl[1] (engine_sync.hip.h).
#include "e8p_gemv_core.hip.h"
#include "engine_sync.hip.h"
#include "fht_wg512x.hip.h"

#ifndef QUIP_INO_EXACT
#define QUIP_INO_EXACT 0
#endif
#ifndef QUIP_PREDECODE_DOWN
#define QUIP_PREDECODE_DOWN 2      // items of down decoded inside the wait for the MLP rows (3: the register allocator spills 36 bytes)
#endif
#ifndef QUIP_PREDECODE_QKV
#define QUIP_PREDECODE_QKV 2       // items of the NEXT block's q / k / v decoded inside the wait for z_d (0 | 2)
#endif
#ifndef QUIP_QKV_ISSUE_EARLY
#define QUIP_QKV_ISSUE_EARLY 0     // A/B: the next block's q / k / v requested between the two halves of the gate / up products (1) or behind the rows' sweep (0)
#endif
#ifndef QUIP_OWNER_PREDECODE
#define QUIP_OWNER_PREDECODE 1     // A/B: the MLP row owners decode down's items ahead like everybody (1) or at the product (0)
#endif
#ifndef QUIP_ATT_STAMPS
#define QUIP_ATT_STAMPS 0          // tools/dbg: the stamps 18..22 inside the attention instead of inside the gate / up edge
#endif
#if QUIP_ATT_STAMPS
#define ASTAMP(i) BSTAMP(18 + (i))
#else
#define ASTAMP(i) do { } while (0)
#endif
#ifndef QUIP_ATT_THREADS
#define QUIP_ATT_THREADS 256       // threads of a workgroup that visit cached positions (A/B: 512 -- the rounds are VALU-issue bound, the
                                   // second wave of a SIMD only makes the first one wait at the barrier: 6.9K -> 7.1K clocks at 100 positions)
#endif
#ifndef QUIP_POLL2
#define QUIP_POLL2 0               // A/B: the edges' gathers poll with TWO staggered read sets (1) or one (0).  Measured: two sets make
                                   // the hand-offs LONGER (z_o 3.6K -> 4.0K clocks, z_d 3.8K -> 4.4K; stagger 5 / 10 / 16 alike) -- the
                                   // polls' traffic delays the stores they wait for by more than the finer grid gains
#endif
#ifndef QUIP_POLL2_STAGGER
#define QUIP_POLL2_STAGGER 10      // s_sleep units (64 clocks) between the first requests of the two sets
#endif
// nibble mode: a decoded item waits as 16 scalar registers, not 32 -- more of them fit
#ifndef QUIP_NIB_PREDECODE_QKV
#define QUIP_NIB_PREDECODE_QKV 3
#endif
#ifndef QUIP_NIB_PREDECODE_GATE
#define QUIP_NIB_PREDECODE_GATE 3
#endif
#ifndef QUIP_NIB_PREDECODE_DOWN
#define QUIP_NIB_PREDECODE_DOWN 3
#endif
// o_proj's input transform split over its producers (round 6; 1 | 0 = every workgroup transforms all 4096 points of a (.) SU, rounds
// 3-5), for the codebooks whose planes are the plain ones (E8P12, D4): H_4096 = H_32 (x) H_128 with the head index on top -- the head's
// workgroup multiplies its 128 attention outputs by SU_o and runs their 128-point transform in ONE wave before it publishes them (two
// fp32 per granule, esync::pack20x2: the same 64 granules per head); everybody gathers two heads x four columns per thread and finishes
// with H_32 ACROSS heads -- one register stage and four lane stages, no LDS exchange -- and writes a dword of digits per head and
// plane.  (decode_block_gqa.hip: QUIP_GQA_OHEAD.)
#ifndef QUIP_OHEAD
#define QUIP_OHEAD 1
#endif
#ifndef QUIP_PREDECODE_GATE
#define QUIP_PREDECODE_GATE 2      // items of gate / up decoded inside the wait for z_o (3: spills 60 bytes)
#endif

namespace quip {

namespace {

using esync::u32x2_t;
using esync::u32x4_t;

// one decoder block as the kernel reads it (256 bytes; built by the host once per model)
struct BlockLayer {
  const uint4* W[7];       // Qidxs of q, k, v, o, gate, up, down
  const f16* ln[2];        // input_layernorm, post_attention_layernorm weights [4096]
  const f16* su[7];        // SU of q, k, v, o, gate, up, down
  const f16* sv[7];        // SV of q, k, v, o, gate, up, down
  const f16* had3;         // the packed K x K factors of the MLP (qlinear._engine_had3)
  f16* kcache;             // [heads, max_len, 128]
  f16* vcache;
  float sc[7];             // wscale_float / sqrt(L_in) of q, k, v, o, gate, up, down (L_in = 4096, down: 256)
  float pad_[5];
};
static_assert(sizeof(BlockLayer) == 256, "layer descriptor layout");

struct BlockArgs {
  const BlockLayer* layers;
  const f16* h_in;         // [4096]: the embedding row of the current token
  f16* h_out;              // [4096]: hidden state after the last block
  const int64_t* pos;      // device scalar
  const float* cos;        // [max_len, 128]
  const float* sin;
  const uint64_t* grid;    // grid_packed_abs
  char* ws;                // workspace (block_engine_workspace_bytes)
  uint64_t* dbg;           // optional: 32 clock stamps per workgroup of block `dbg_layer`
  int n_layers, max_len, dbg_layer;
  float rms_eps, attn_scale;
  float resid_scale;       // E8P12RVQ4B / RVQ3B: the fp16 residual scale (as float)
  const void* grid2;       // E8P12RVQ3B: the E81B residual table, 256 x 8 int8 (4 r); else unused
};

constexpr int kWaves = 8, kThreads = 512;
constexpr int HID = 4096, NWG = 256, RPW = 16, HD = 128, NH = 32;
// Round 5: the SAME kernel for the grouped-query 4096-wide shape of Llama-3-8B / Mistral-7B (32 heads on 8 KV heads, n_ffn =
// 14336 = 7 x 2048), compiled from this file a second time with QUIP_BLOCK_G8 = 1 (decode_block_g8.hip).  Its MLP is the
// (43, 256) machinery with K = 56: the reference's transform of a 14336-vector, (R_7 (x) H_2048) / sqrt(2048) on the (7, 2048)
// view (quant.py:26-39, 72-88), IS ((R_7 (x) H_8) (x) H_256) / (sqrt 8 sqrt 256) on the (56, 256) view -- H_2048 = H_8 (x) H_256
// in Sylvester order, index k 2048 + j8 256 + j = (8 k + j8) 256 + j -- so the host hands over the 56 x 56 factors
// R_7 (x) H_8 (entries +-R_7: exact in fp16) and the kernel folds the 1 / sqrt 8 into its scales (kMixScale, sc[6]).
// G8 differs in: the q / k / v row blocks (384 instead of 768: one or two per workgroup), the KV head of a query head, seven
// gate / up items and four down items per wave on the nine slots, E8P12 only.
#ifndef QUIP_BLOCK_G8
#define QUIP_BLOCK_G8 0
#endif
constexpr bool G8 = QUIP_BLOCK_G8 != 0;
constexpr int NKVH = G8 ? 8 : NH, GQH = NH / NKVH;              // KV heads; query heads per KV head
constexpr int FK = G8 ? 56 : 43, FLOGL = 8, FL = 256, NFFN = FK * FL, FRB = 3;
constexpr int NGU = G8 ? 7 : 6;                                 // gate / up items per wave (G8: 2 columns x 56 rows = 7 row blocks)
constexpr int KPD = G8 ? 14336 : 11264, JD = G8 ? 28 : 22;      // digits of down's input, slices
constexpr int ND = G8 ? 4 : 3;                                  // down items per wave
constexpr float kMixScale = G8 ? 0.35355339059327373f : 1.f;    // 1 / sqrt 8 of H_8 inside the 56 x 56 factors
constexpr int QKB = G8 ? 384 : 768;                             // 16-row blocks of the stacked [q; k; v] rows
constexpr int kRowU4 = HID / 64, kRowU4D = NFFN / 64;
constexpr int NSLOT = 9;                             // X0-2: q k v, then gate's row blocks | X3-5: o, then up's | X6-8: down

// workspace: ctl | z_q z_k z_v | a | z_o | z_d (2048 granules each) | inbox [11 row owners][256 columns][2][4] | rows [256][48] |
// attention partials [32 heads][8][132]
constexpr size_t kWsCtl = 0, kWsZ = 64, kWsVec = 2048 * 8;
// row-owner workgroups of the MLP edge: RPO rows k' each, a PAIR of waves per row (gate half, up half) on waves 0..2 RPO-1 (RPO = 2: four waves = one per SIMD, 22 owners; with RPO = 4 the eight waves shared SIMDs: 1390 -> 1354 us per 32-block launch, same box: the
// row work is DPP-serial VALU code, two such waves on a SIMD take twice as long)
constexpr int RPO = 2, NRO = (FK + RPO - 1) / RPO;
constexpr size_t kWsInbox = kWsZ + 6 * kWsVec, kWsRows = kWsInbox + (size_t)NRO * FL * 2 * RPO * 8;
// long contexts: the eight workgroups of a head each take every eighth position; their partial softmax states (128 sums +
// maximum + denominator, padded to 132 granules) meet at the head's first workgroup
// rows hand-off: wave 0 polls the last column before everybody sweeps the 90 KB.  Without it (every workgroup sweeping
// until its pieces are all there) the hand-off takes 29.5K clocks instead of 21.4K: 256 x 90 KB of polls per retry crowd out
// the owners' stores
constexpr bool kCheapPoll = true;
constexpr int kParts = 8, kPartGran = 132, kSplitPos = 128;      // (measured: 1.83 ms per token at 200 positions on one workgroup per head, 1.74 at 256 split)
constexpr size_t kWsPart = kWsRows + (size_t)FL * 48 * 8;
// short contexts: the head's second / third workgroup transform k / v of the new position and hand its 128 values over
constexpr size_t kWsKvNew = kWsPart + (size_t)NH * kParts * kPartGran * 8;
constexpr size_t kWsBytes = kWsKvNew + (size_t)NH * HD * 8;

template <int REP, bool RVQ = false>
struct BLds {
  using T = Lds<REP>;
  // RVQ (E8P12RVQ4B): read as 16-bit E8P codes a row has twice as many (virtual) weights, 8-groups alternating residual / main,
  // and multiplies x' = [s x_g | x_g]_g: twice the digits per vector, twice the items per product (hadamard.hip, rvq_scale)
  static constexpr int VM = RVQ ? 2 : 1, KV = VM * HID;
  // NIBBLE MODE (REP = 4, round 6; e8p_gemv_core.hip.h): both E8P tables as 4-byte entries, 32 conflict-free copies each, in 64 KB
  // (the byte tables' 32 / 16 copies: 96 KB); digit planes as HALF planes -- a plane = its "lo" half (positions 0..3 of the
  // 8-groups), then, 16 bytes off a multiple of 256 later, its "hi" half; planes 64 bytes off a multiple of 256 apart (the pieces
  // a ds_read_b128 lane group touches on different banks); the accumulator rows hold 8 x the digit sums of the byte tables
  static constexpr bool kNib = T::kNib;
  static_assert(!kNib || !RVQ, "nibble mode: E8P12 (the virtual rows of the RVQ codebooks stay on the byte tables)");
  static constexpr int PSH = kNib ? KV + 64 : KV, HOH = KV / 2 + 16;       // plane stride of a hidden-wide vector; its "hi" half
  static constexpr int KPDV = RVQ ? 22528 : KPD, JDV = RVQ ? 43 : JD;   // digits of down's input (virtual, padded), its slices
  static constexpr int KKP = (FK * FK + 7) & ~7, KP16 = G8 ? 64 : 48;
  // row stride of down's transposed factor in the image: 64 fp16 = 128 bytes put the 16 rows of a B fragment on ONE bank group
  // (16-way conflicts: 11.6K clocks for the K-mix + planes of the G8 launch instead of ~5K); 72 spreads them
  static constexpr int KPS = G8 ? 72 : KP16;
  static constexpr int kAcc = T::kAcc;                       // int32 [176][4]: q k v (48) | o (16) | gate up (96; G8: 112) | down (16)
  static constexpr int AGU = 64, AD = AGU + 16 * NGU;        // accumulator rows of gate / up and of down
  static constexpr int kAccRows = AD + 16;
  static constexpr int kZcol = kAcc + kAccRows * 16;         // float [2][KP16]: z of this workgroup's two columns (MLP)
  static constexpr int kRed = kZcol + 2 * KP16 * 4;          // float [64] reduction scratch, int [8] shift words
  static constexpr int kDesc = kRed + 256 + 32;              // the current block's descriptor (256 bytes): pointers are read
                                                             // from here, not from memory (a vector load of a pointer ahead of
                                                             // every request would wait for the requests before it)
  static constexpr int kDescN = kDesc + 256;                 // the next block's descriptor (256 bytes).  (The residual stream
                                                             // of rounds 3-4, 8 KB here, lives in registers now.)
  static constexpr int kQkv = kDescN + 256;                  // fp16 [3][128]: this head's q, k, v; [128] attention output
  static constexpr int kCs = kQkv + 4 * HD * 2;              // float [2][128]: the rotary row of this token (cos | sin), the same for every block
  // ONE transient area for everything that lives between two products (E8P12: T1 x 32 + T2 x 16 = 96 KB of tables leave 50.3 KB).
  // In time: the transforms' exchange buffers (16.5 KB per transform from the base: 1, 2 or -- attention -- 3) -> digit planes
  // of the consumers (12 / 24 KB from the base, written after the transforms' last reads) | attention partials; for the MLP:
  // planes [0, 24 K) + the K x K factor image [24 K, 36 K) + the row owners' vectors [36 K, 48 K) -> row owners: inbox rows
  // [0, 8 K), rows on the way out [8 K, 16 K) -> the gathered rows [0, 48 K) (the factor image is in registers by then) ->
  // down's planes [0, 35.1 K).
  static constexpr int kArea = kCs + 2 * HD * 4;
  static constexpr int kBufBytes = ((had::buf_floats(HID) * 4) + 15) & ~15;
  static constexpr int kAreaBytes = (160 * 1024 - kArea) & ~15;
  static constexpr int kBuf0 = kArea;
  static constexpr int kHadElems = 2 * KKP + KP16 * KPS;
  static constexpr int kStage = kArea + 8 * 1024;            // MLP row owners: their four rows, transposed, on the way out (4 KB)
  // fp16 image of the three K x K factors (12 KB; G8: 20.3 KB), behind the planes of gate / up and behind the rows' LDS image
  // ([KP16 / 2][256] fp16 pairs = 24 KB; G8: 32 KB)
  static constexpr int kHad = kArea + (G8 ? 32 * 1024 : 2 * 3 * PSH);
  // MLP row owners: SV_gate / SV_up / SU_down of their rows (6 KB): behind the factor image; G8: at 16 K (free while they are alive:
  // the planes of the ONE consumer end at 12 K, the owners' rows and staging at 12 K)
  static constexpr int kStash = G8 ? kArea + 16 * 1024 : kHad + 12 * 1024;
  // down's planes.  Byte tables: every 256 digits padded by 16 bytes (the K-mix writes a dword per lane, 256 digits apart).  Nibble
  // mode: a half plane = (KPDV / 256) blocks of 128 + 16 bytes (the same writers: 144-byte steps land on 16 different banks; without
  // the pad their dwords, 128 bytes apart, met on two: 6.5K instead of 3.9K clocks for the K-mix + planes), "hi" half at HOD = 16
  // (mod 256) behind the "lo" half, planes kPlaneD = 64 (mod 256) apart
  static constexpr int kHalfD = (KPDV / 256) * 144;
  static constexpr int up256(int x, int r) { return x + ((r - x % 256) + 256) % 256; }
  static constexpr int HOD = up256(kHalfD, 16);
  static constexpr int kPlaneD = kNib ? up256(HOD + kHalfD, 64) : (KPDV / 256) * 272;
  static constexpr int kBytes = kArea + kAreaBytes;
  static_assert(kAreaBytes >= 3 * kBufBytes && kAreaBytes >= FL * KP16 * 2 && kAreaBytes >= 3 * kPlaneD &&
                kHad + kHadElems * 2 <= kArea + kAreaBytes && kStash + 6 * 1024 <= (G8 ? kHad : kArea + kAreaBytes),
                "transient area");
};
#if QUIP_BLOCK_G8
static_assert(BLds<24>::kBytes <= 160 * 1024 && BLds<4>::kBytes <= 160 * 1024, "LDS budget");
#else
static_assert(BLds<4>::kBytes <= 160 * 1024 && BLds<24>::kBytes <= 160 * 1024 && BLds<16>::kBytes <= 160 * 1024 && BLds<64>::kBytes <= 160 * 1024 &&
              BLds<16, true>::kBytes <= 160 * 1024 && BLds<64, true>::kBytes <= 160 * 1024 && BLds<12, true>::kBytes <= 160 * 1024,
              "LDS budget");
#endif

// RVQ: rows of twice the (virtual) width.  HI (with RVQ and the D4 table mode): the HI codebook -- a code byte holds two
// nibbles and reads as a D4 code of the virtual row with the table entry [lo - 7.5, hi - 7.5, 0, 0], against
// x' = [x0 x2 0 0 | x4 x6 0 0 | x1 x3 0 0 | x5 x7 0 0]_g (hadamard.hip, HI layout)
// E8P12RVQ3B (RVQ on the third-table mode REP = 12): the checkpoint's 3-byte codes [resid8, e8p_lo, e8p_hi] behind a zero byte
// are the dwords (main16 << 16 | resid8 << 8) of an RVQ4-style virtual row whose low codes index the E81B table
// (e8p_gemv_core.hip.h, rvq3_dwords).  Same items, digits and slots as E8P12RVQ4B; a lane's half item is 12 bytes (four codes)
// instead of 16, so every byte offset of the weight stream is 3/4 of RVQ4B's (rows of 3 k / 8 bytes).
template <int REP, bool RVQ = false, bool HI = false>
__global__ __launch_bounds__(kThreads) void decode_block_kernel(BlockArgs a) {
  static_assert(!HI || (RVQ && Lds<REP>::kD4), "HI = virtual rows of twice the width on the D4 table mode");
  constexpr bool R3 = Lds<REP>::kRvq3;
  static_assert(!R3 || (RVQ && !HI), "RVQ3B: the virtual rows of RVQ");
  using slot_t = std::conditional_t<R3, u32x3, u32x4>;
  constexpr uint32_t PB = R3 ? 12u : 16u;              // bytes of a lane's piece of the weight stream
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using B = BLds<REP, RVQ>;
  constexpr int VM = B::VM, KV = B::KV;
  constexpr int kRowU4V = VM * kRowU4, kRowU4DV = VM * kRowU4D;     // 16-byte pieces of a weight row (hidden- / ffn-wide input)
  // gate / up (round 5): workgroup w multiplies ONE of the two matrices -- mgu = w >> 7: 0 gate, 1 up -- for TWO columns of
  // the (43, 256) view of its output, 2 (w & 127) and + 1 (rows k * 256 + column: the second column is the next weight row,
  // kRowB bytes on, in the units ld_item_o() takes), so that it needs ONE input transform on the edge in front of them instead of two
  constexpr int kRowB = kRowU4V * 16;
  using T = Lds<REP>;
  // Everything derived from the thread index is RE-derived from an opaque copy at the top of every stage (rederive()):
  // left to itself the compiler hoists ~70 lane-dependent addresses out of the block loop and spills them.
  int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w = blockIdx.x;
  const int mgu = w >> 7;
  int n = lane & 15, q = lane >> 4;
  uint32_t* ctl = reinterpret_cast<uint32_t*>(a.ws + kWsCtl);
  uint64_t* zbufs = reinterpret_cast<uint64_t*>(a.ws + kWsZ);       // [6][2048]: q k v a o d
  uint64_t* inbox = reinterpret_cast<uint64_t*>(a.ws + kWsInbox);
  uint64_t* frow = reinterpret_cast<uint64_t*>(a.ws + kWsRows);
  uint64_t* pbuf = reinterpret_cast<uint64_t*>(a.ws + kWsPart);
  // (kWsKvNew: the k / v hand-off inside a head's group of rounds 3-4; unused since the heads transform q, k, v themselves)
  int dbg_on = 0;
  // (dbg_layer bit 16: the stamps are s_memrealtime -- 100 MHz, ONE counter for the whole device -- instead of s_memtime, whose
  //  counters differ between XCDs: which workgroup gets to a stamp last can only be read off the former)
  const bool dbg_rt = (a.dbg_layer & 0x10000) != 0;
#define BSTAMP(i) do { if (dbg_on && tid == 0) a.dbg[w * 32 + (i)] = dbg_rt ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)

  // ---- weight slots ---------------------------------------------------------------------------------------------
  slot_t qa[NSLOT], qb[NSLOT];
  // item kinds: 0..2 = row blocks 3 w + kind of the stacked [q; k; v] rows (256 row blocks of 16 rows per matrix: a
  // workgroup's three blocks touch at most two of the matrices, so ONE round of input transforms serves it), 3 = o rows
  // [16 w, +16) (slice = wave); 4..9 = gate / up row block (kind - 4) % 3 of matrix (kind - 4) / 3 (rows k * 256 + w);
  // 10..12 = down, slice (kind - 10) * 8 + wave.  A load = scalar base of the matrix + a 32-bit byte offset of this
  // lane (+ 64 for the second half of the item).
  uint32_t vo_row, vo_q[3], vo_gu[G8 ? 7 : FRB], vo_d[G8 ? 4 : 3], vo_d2b, vo_dr[3];
  // G8: the stacked [q; k; v] rows are 384 blocks of 16 (q 0..255, k 256..319, v 320..383): q's two by two on workgroups 0..127,
  // k's and v's one each on 128..255 -- blocks qb0 .. qb0 + qcnt - 1, always inside ONE matrix (two by two on the even workgroups
  // and one on the odd ones, workgroup 170 held q 255 | k 0: two input transforms, 1.1 us for which everybody waited)
  const int qb0 = w < 128 ? 2 * w : 128 + w, qcnt = w < 128 ? 2 : 1;
  auto qmat = [](int b) { return b < 256 ? 0 : (b < 320 ? 1 : 2); };
  auto qblk = [](int b) { return b < 256 ? b : (b < 320 ? b - 256 : b - 320); };
  uint32_t lane_c, lane_c2, lane_c3 = 0u, xlane;
  constexpr bool NIB = T::kNib;
  constexpr bool kOHead = QUIP_OHEAD != 0 && !RVQ;     // (the RVQ codebooks' and HI's virtual rows scatter their planes differently)
  constexpr int kPreW = NIB ? 16 : 32;                  // dwords of a decoded item that waits in registers
  constexpr int kUnsc = T::kD4 ? 1 : (NIB ? 5 : 2);     // the accumulator rows hold 2^kUnsc x sum of digit x w (table entries 4 w / 2 w; nibble mode 8 x 4 w)
  NibLane nlf = {0, 0, 0u};                             // nibble mode: this lane's factors of an item's rows (item_rows_nib)
  auto rederive = [&]() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    tid = t; lane = t & 63; n = lane & 15; q = lane >> 4;
    vo_row = (uint32_t)((w * RPW + n) * kRowU4V + wave * 8 + q) * PB;
    if constexpr (G8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {      // (an odd workgroup's second item: its one block once more, never multiplied)
        const int bq = qb0 + (i < qcnt ? i : 0);
        vo_q[i] = (uint32_t)((qblk(bq) * 16 + n) * kRowU4V + wave * 8 + q) * PB;
      }
      vo_q[2] = vo_q[0];
#pragma unroll
      for (int i = 0; i < 7; ++i) {      // rows k * 256 + column of this workgroup's two columns, as 112 consecutive (column, k)
        const int r = 16 * i + n, c = r >= FK ? 1 : 0, k = r - FK * c;
        vo_gu[i] = (uint32_t)((k * FL + 2 * (w & 127) + c) * kRowU4V + wave * 8 + q) * PB;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {      // down: slice 8 i + wave of the 28 (the fourth item exists on waves 0..3)
        const int sl0 = i * kWaves + wave;
        const int sl = sl0 < JD ? sl0 : 0;
        vo_d[i] = (uint32_t)(((w * RPW + n) * kRowU4D + sl * 8 + q) * 16);
      }
      vo_d2b = 0u;
    } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) vo_q[i] = (uint32_t)((((3 * w + i) & 255) * 16 + n) * kRowU4V + wave * 8 + q) * PB;
#pragma unroll
    for (int rb = 0; rb < FRB; ++rb) {
      int kr = rb * 16 + n;
      kr = kr < FK ? kr : FK - 1;
      vo_gu[rb] = (uint32_t)((kr * FL + 2 * (w & 127)) * kRowU4V + wave * 8 + q) * PB;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int sl0 = i * kWaves + wave;
      const int sl = sl0 < JD ? sl0 : 0;
      vo_d[i] = (uint32_t)(((w * RPW + n) * kRowU4D + sl * 8 + q) * 16);
      if (i == 2) {
        int off = sl * 8 + q + 4;
        off = off < kRowU4D ? off : kRowU4D - 1;       // the row's last slice is a half one: its digits beyond are zero
        vo_d2b = (uint32_t)(((w * RPW + n) * kRowU4D + off) * 16);
      }
    }
    }
    // RVQ: down's 43 virtual slices: slice wave + 8 i, i < 6, at vo_dr[i >> 2] + (i & 3) KB (the sixth exists for waves 0..2)
    vo_dr[0] = (uint32_t)((w * RPW + n) * kRowU4DV + wave * 8 + q) * PB;
    vo_dr[1] = vo_dr[0] + 256u * PB;
    vo_dr[2] = vo_dr[1] + (wave < 3 ? 64u * PB : 0u);
    lane_c = NIB ? nib_lane_const(lane)
             : Lds<REP>::kD4 ? ((uint32_t)lane << 2)
             : (Lds<REP>::kRep1 == 32) ? ((((uint32_t)lane & 31u) << 3) | 0x00010000u)
                                       : ((((uint32_t)lane & 15u) << 3) | (uint32_t)Lds<REP>::kT1);
    lane_c2 = (((uint32_t)lane & 15u) << 3) | (uint32_t)Lds<REP>::kT2;
    if constexpr (R3) lane_c3 = (((uint32_t)lane & (uint32_t)(Lds<REP>::kRep3 - 1)) << 3) | (uint32_t)Lds<REP>::kT3;
    if constexpr (NIB) {
      nlf = nib_lane_factors(q);
      xlane = (uint32_t)B::kArea + nib_row_offset(n, B::PSH, B::HOH) + (uint32_t)q * 32u + (uint32_t)wave * 256u;
    } else {
      xlane = (uint32_t)B::kArea + (uint32_t)min(n, 2) * (uint32_t)KV + (uint32_t)q * 64u + (uint32_t)wave * 512u;
    }
  };
  rederive();
  // a pointer the descriptor holds, as a scalar register pair (the same value in every lane)
  auto uni = [](const uint4* p) -> const uint4* {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return reinterpret_cast<const uint4*>(((uint64_t)hi << 32) | lo);
  };
  auto ld_item = [&](slot_t& da, slot_t& db, const uint4* base0, uint32_t vo) {
    const uint4* base = uni(base0);
    // s_nop: the base was just written by v_readfirstlane (VALU write of an SGPR -> VMEM read needs 5 wait states, and
    // the compiler pads no hazard for an instruction inside an asm statement)
    asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=v"(da) : "v"(vo), "s"(base) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:64 nt" : "=v"(db) : "v"(vo), "s"(base) : "memory");
  };
  // slot of an item kind: q k v -> X0-2, o -> X3, gate -> X0-2, up -> X3-5, down -> X6-8
#define SLOT_OF(kind) ((kind) < 4 ? (kind) : (kind) - 4)
#define ISSUE(Ld, kind) do {                                                                                          \
    if ((kind) < 3) ld_item(qa[SLOT_OF(kind)], qb[SLOT_OF(kind)], Ld.W[(3 * w + ((kind) < 3 ? (kind) : 0)) >> 8], vo_q[(kind) < 3 ? (kind) : 0]); \
    else if ((kind) == 3) ld_item(qa[3], qb[3], Ld.W[3], vo_row);                                                      \
    else if ((kind) < 7) ld_item(qa[SLOT_OF(kind)], qb[SLOT_OF(kind)], Ld.W[4 + mgu], vo_gu[((kind) - 4) % FRB]); \
    else if ((kind) < 10) ld_item_o(OFFC(kRowB), qa[SLOT_OF(kind)], qb[SLOT_OF(kind)], Ld.W[4 + mgu], vo_gu[((kind) - 4) % FRB]); \
    else if ((kind) < 12) ld_item(qa[SLOT_OF(kind)], qb[SLOT_OF(kind)], Ld.W[6], vo_d[(kind) >= 10 ? ((kind) - 10) % 3 : 0]);     \
    else {                                                                                                             \
      const uint4* bd_ = uni(Ld.W[6]);                                                                                 \
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt" : "=v"(qa[8]) : "v"(vo_d[2]), "s"(bd_) : "memory");    \
      asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(qb[8]) : "v"(vo_d2b), "s"(bd_) : "memory");              \
    }                                                                                                                  \
  } while (0)
  // RVQ: the same with a compile-time byte offset (the second virtual slice of a row block is 1 KB further in the row)
  auto ld_item_o = [&](auto off_c, slot_t& da, slot_t& db, const uint4* base0, uint32_t vo) {
    constexpr int OFF = decltype(off_c)::value / 16 * (int)PB;      // (written in RVQ4B's bytes)
    const uint4* base = uni(base0);
    if constexpr (R3) {
      asm volatile("s_nop 4\n\tglobal_load_dwordx3 %0, %1, %2 offset:%3 nt" : "=v"(da) : "v"(vo), "s"(base), "n"(OFF) : "memory");
      asm volatile("global_load_dwordx3 %0, %1, %2 offset:%3 nt" : "=v"(db) : "v"(vo), "s"(base), "n"(OFF + 48) : "memory");
    } else {
      asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(da) : "v"(vo), "s"(base), "n"(OFF) : "memory");
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 nt" : "=v"(db) : "v"(vo), "s"(base), "n"(OFF + 64) : "memory");
    }
  };
#define OFFC(n) std::integral_constant<int, (n)>{}
  // RVQ slot plan (8 of the 9 slots): items (row block rb, virtual half h: slice wave + 8 h) of q k v / gate -> slot 3 h + rb;
  // o -> 6 + h; up: half 0 -> 6 + rb (requested while gate is still in flight), half 1 -> rb (requested when gate's half 0
  // has been multiplied); down: slice wave + 8 i -> slot i
#define ISSUE_RVQ_QKV(Ld) do {                                                                          \
    ld_item_o(OFFC(0), qa[0], qb[0], Ld.W[(3 * w) >> 8], vo_q[0]);    ld_item_o(OFFC(1024), qa[3], qb[3], Ld.W[(3 * w) >> 8], vo_q[0]);     \
    ld_item_o(OFFC(0), qa[1], qb[1], Ld.W[(3 * w + 1) >> 8], vo_q[1]); ld_item_o(OFFC(1024), qa[4], qb[4], Ld.W[(3 * w + 1) >> 8], vo_q[1]); \
    ld_item_o(OFFC(0), qa[2], qb[2], Ld.W[(3 * w + 2) >> 8], vo_q[2]); ld_item_o(OFFC(1024), qa[5], qb[5], Ld.W[(3 * w + 2) >> 8], vo_q[2]); \
  } while (0)
#define ISSUE_RVQ_O(Ld) do { ld_item_o(OFFC(0), qa[6], qb[6], Ld.W[3], vo_row); ld_item_o(OFFC(1024), qa[7], qb[7], Ld.W[3], vo_row); } while (0)
#define ISSUE_RVQ_GATE(Ld) do {                                                                         \
    ld_item_o(OFFC(0), qa[0], qb[0], Ld.W[4 + mgu], vo_gu[0]); ld_item_o(OFFC(0), qa[1], qb[1], Ld.W[4 + mgu], vo_gu[1]); ld_item_o(OFFC(0), qa[2], qb[2], Ld.W[4 + mgu], vo_gu[2]); \
    ld_item_o(OFFC(1024), qa[3], qb[3], Ld.W[4 + mgu], vo_gu[0]); ld_item_o(OFFC(1024), qa[4], qb[4], Ld.W[4 + mgu], vo_gu[1]); ld_item_o(OFFC(1024), qa[5], qb[5], Ld.W[4 + mgu], vo_gu[2]); \
  } while (0)
// the same bursts a third at a time (the CU's address unit takes ~25 clocks per 1 KB request: 48 requests in a row stall
// the issuing waves for ~1.2K clocks, 16 at a time between other work do not)
#define ISSUE_RVQ_GATE_G(Ld, i) do { ld_item_o(OFFC(0), qa[i], qb[i], Ld.W[4 + mgu], vo_gu[i]); ld_item_o(OFFC(1024), qa[3 + (i)], qb[3 + (i)], Ld.W[4 + mgu], vo_gu[i]); } while (0)
#define ISSUE_RVQ_QKV_G(Ld, i) do { ld_item_o(OFFC(0), qa[i], qb[i], Ld.W[(3 * w + (i)) >> 8], vo_q[i]); ld_item_o(OFFC(1024), qa[3 + (i)], qb[3 + (i)], Ld.W[(3 * w + (i)) >> 8], vo_q[i]); } while (0)
#define ISSUE_RVQ_DOWN_G0(Ld) do { ld_item_o(OFFC(0), qa[0], qb[0], Ld.W[6], vo_dr[0]); ld_item_o(OFFC(1024), qa[1], qb[1], Ld.W[6], vo_dr[0]); } while (0)
#define ISSUE_RVQ_DOWN_G1(Ld) do { ld_item_o(OFFC(2048), qa[2], qb[2], Ld.W[6], vo_dr[0]); ld_item_o(OFFC(3072), qa[3], qb[3], Ld.W[6], vo_dr[0]); } while (0)
#define ISSUE_RVQ_DOWN_G2(Ld) do { ld_item_o(OFFC(0), qa[4], qb[4], Ld.W[6], vo_dr[1]); ld_item_o(OFFC(0), qa[5], qb[5], Ld.W[6], vo_dr[2]); } while (0)
#define ISSUE_RVQ_UP_A_G(Ld, i) do { ld_item_o(OFFC(kRowB), qa[6 + (i)], qb[6 + (i)], Ld.W[4 + mgu], vo_gu[i]); } while (0)
#define ISSUE_RVQ_UP_A(Ld) do { ld_item_o(OFFC(kRowB), qa[6], qb[6], Ld.W[4 + mgu], vo_gu[0]); ld_item_o(OFFC(kRowB), qa[7], qb[7], Ld.W[4 + mgu], vo_gu[1]); ld_item_o(OFFC(kRowB), qa[8], qb[8], Ld.W[4 + mgu], vo_gu[2]); } while (0)
#define ISSUE_RVQ_UP_B(Ld) do { ld_item_o(OFFC(kRowB + 1024), qa[0], qb[0], Ld.W[4 + mgu], vo_gu[0]); ld_item_o(OFFC(kRowB + 1024), qa[1], qb[1], Ld.W[4 + mgu], vo_gu[1]); ld_item_o(OFFC(kRowB + 1024), qa[2], qb[2], Ld.W[4 + mgu], vo_gu[2]); } while (0)
#define ISSUE_RVQ_DOWN(Ld) do {                                                                         \
    ld_item_o(OFFC(0), qa[0], qb[0], Ld.W[6], vo_dr[0]); ld_item_o(OFFC(1024), qa[1], qb[1], Ld.W[6], vo_dr[0]); \
    ld_item_o(OFFC(2048), qa[2], qb[2], Ld.W[6], vo_dr[0]); ld_item_o(OFFC(3072), qa[3], qb[3], Ld.W[6], vo_dr[0]); \
    ld_item_o(OFFC(0), qa[4], qb[4], Ld.W[6], vo_dr[1]);                                                        \
    ld_item_o(OFFC(0), qa[5], qb[5], Ld.W[6], vo_dr[2]);      /* (waves 3..7: slice wave + 32 again, not multiplied) */ \
  } while (0)
  // G8 slot plan: the next block's q / k / v items -> X2, X3 | o -> X3 | gate / up items 0..6 -> X0..X6 | down items 0..3 -> X7, X8,
  // X0, X1 (requested once gate / up's first four items have been multiplied)
#define ISSUE8_QKV(Ld, i) ld_item(qa[2 + (i)], qb[2 + (i)], Ld.W[qmat(qb0 + ((i) < qcnt ? (i) : 0))], vo_q[i])
#define ISSUE8_GU(Ld, i) ld_item(qa[i], qb[i], Ld.W[4 + mgu], vo_gu[i])
#define ISSUE8_DOWN(Ld, i) ld_item(qa[(i) < 2 ? 7 + (i) : (i) - 2], qb[(i) < 2 ? 7 + (i) : (i) - 2], Ld.W[6], vo_d[i])
  // after a drain: every slot is a plain register again
  // (`mask`: the slots that can be in flight at that point.  The others are dead there, and saying so frees their
  //  registers for the phase: the attention prologue keeps 8 of the 72 slot registers)
  auto own_slots = [&](auto mask) {
    constexpr unsigned M = decltype(mask)::value;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
      if ((M >> s) & 1u) { esync::own(qa[s]); esync::own(qb[s]); }
  };
#define SLOTS(m) std::integral_constant<unsigned, (m)>{}
  // slots of the items of a product (see ISSUE / ISSUE_RVQ_*)
  constexpr unsigned M_QKV = G8 ? 0x00cu : (RVQ ? 0x03fu : 0x007u), M_O = RVQ ? 0x0c0u : 0x008u, M_GATE = RVQ ? 0x03fu : 0x007u;
  constexpr unsigned M_UP = G8 ? 0x078u : (RVQ ? 0x1c0u : 0x038u) /* RVQ: up's first half */, M_DOWN = G8 ? 0x183u : (RVQ ? 0x03fu : 0x1c0u);
  static_assert(!G8 || (!RVQ && !HI && (REP == 24 || REP == 4)), "the grouped-query 4096-wide shape: E8P12 only");

  // ---- prologue ---------------------------------------------------------------------------------------------------
  u32x2 tsrc;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(tsrc) : "v"(NIB ? table_source_ptr_nib(a.grid, lane, wave) : T::kD4 ? table_source_ptr_d4(a.grid, lane, wave) : table_source_ptr(a.grid, lane, wave)) : "memory");
  u32x2 tsrc3 = {0u, 0u};                            // RVQ3B: this lane's E81B entry (row 32 wave + (lane & 31) of T3)
  if constexpr (R3)
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(tsrc3) : "v"(reinterpret_cast<const uint2*>(a.grid2) + (wave * 32 + (lane & 31))) : "memory");
  uint32_t gen;
  esync::ld4(gen, ctl);
  // the residual stream lives in registers for the whole launch, in the strided layout of the 512-thread transforms: this
  // thread's h[tid + 512 k], k < 8, as four fp16 pairs (k = 2 j | 2 j + 1)
  uint32_t hreg[4];
  uint32_t hraw[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    asm volatile("global_load_ushort %0, %1, off" : "=v"(hraw[k]) : "v"(reinterpret_cast<const uint16_t*>(a.h_in) + tid + 512 * k) : "memory");
  {
    const BlockLayer& L0 = a.layers[0];
    if constexpr (RVQ) ISSUE_RVQ_QKV(L0); else if constexpr (G8) { ISSUE8_QKV(L0, 0); ISSUE8_QKV(L0, 1); } else { ISSUE(L0, 0); ISSUE(L0, 1); ISSUE(L0, 2); }
  }
  constexpr int NQ = RVQ ? 12 : (G8 ? 4 : 6);        // loads of the first q, k, v items in flight across the prologue
  int* accs = reinterpret_cast<int*>(smem + B::kAcc);
  for (int i = tid; i < B::kAccRows * 4; i += kThreads) accs[i] = 0;
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(tsrc), "+v"(tsrc3) : "n"(9 + NQ) : "memory");
  if constexpr (NIB) fill_tables_nib(tsrc, lane, wave);
  else fill_tables_from_lane<REP>(smem, tsrc, lane, wave);
  fill_t3_from_lane<REP>(smem, tsrc3, lane, wave);
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(gen) : "n"(8 + NQ) : "memory");
  const uint32_t ebase = ((uint32_t)__builtin_amdgcn_readfirstlane((int)gen) + 1u) << 10;
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(hraw[0]), "+v"(hraw[1]), "+v"(hraw[2]), "+v"(hraw[3]), "+v"(hraw[4]), "+v"(hraw[5]),
               "+v"(hraw[6]), "+v"(hraw[7]) : "n"(NQ) : "memory");
#pragma unroll
  for (int j = 0; j < 4; ++j) hreg[j] = (hraw[2 * j] & 0xffffu) | (hraw[2 * j + 1] << 16);
  const long long pos64 = *a.pos;
  const bool pos_ok = pos64 >= 0 && pos64 < (long long)a.max_len;
  const int pos = pos_ok ? (int)pos64 : 0;
  if (tid < 2 * HD)
    reinterpret_cast<float*>(smem + B::kCs)[tid] = (tid < HD ? a.cos : a.sin - HD)[(size_t)pos * HD + tid];
  had::wg_barrier<true>();
  uint32_t hop = 0;                                 // hand-offs so far in this launch (tag = ebase | hop)

  float* red = reinterpret_cast<float*>(smem + B::kRed);
  int* shs = reinterpret_cast<int*>(smem + B::kRed + 256);

  // ---- all-gather of one or more 4096-vectors (2048 granules each, {2 x fp16, tag}) into LDS as fp16 -------------------
  // NV vectors starting at zbufs[first]; every thread sweeps 2 NV 16-byte pieces; returns with the data in smem + kZs
  auto gather = [&](auto nv_tag, auto slots, int first, uint32_t tag, uint32_t code, float (&out)[decltype(nv_tag)::value][8]) {
    constexpr int NV = decltype(nv_tag)::value;
    u32x4_t p[2 * NV];
    uint32_t spins = 0;
    const uint64_t* src = zbufs + (size_t)first * 2048 + 4 * tid;      // this thread's 4 granules of vector `first`
    for (;;) {
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        esync::ld16(p[2 * c], src + c * 2048);
        esync::ld16(p[2 * c + 1], src + c * 2048 + 2);
      }
      esync::drain();
      bool ok = true;
#pragma unroll
      for (int j = 0; j < 2 * NV; ++j) {
        esync::own(p[j]);
        ok = ok && p[j].y == tag && p[j].w == tag;
      }
      if (esync::spin_step(ok, spins, ctl + 1, code + (uint32_t)w)) break;
    }
    // straight into the transform's input registers: no staging, no barrier (every thread waited for its own elements)
#pragma unroll
    for (int c = 0; c < NV; ++c)
      had::unpack8(make_uint4(p[2 * c].x, p[2 * c].z, p[2 * c + 1].x, p[2 * c + 1].z), out[c]);
    own_slots(slots);
  };
  // The same with TWO read sets in flight, half a round trip apart.  A hand-off ends when the LAST of 2048 waves has seen its
  // pieces: with one set a wave notices them up to a full round trip (~0.7 us) after they became visible, with two up to half
  // of one.  The loop is ONE asm statement on sixteen NAMED registers (v[236:251]): written in C++ the sets are in flight
  // across the loop's back edge, where the allocator is free to split a live range -- it copied the registers BEFORE the
  // wait (tools/check_inflight.py).  On success the other set is still in flight: its registers stay bound to `pa`, `pb`
  // until the caller has said poll2_settle with N = the VMEM requests it has issued since (vmcnt retires in order).
  auto gather2 = [&](auto slots, int first, uint32_t tag, uint32_t code, float (&out)[1][8], u32x4_t (&pa)[2], u32x4_t (&pb)[2]) {
    const uint64_t* src = zbufs + (size_t)first * 2048 + 4 * tid;
    uint32_t o0, o1, o2, o3, st, cnt, rounds = 0;
    uint64_t tmp;
#pragma nounroll
    for (;;) {
      asm volatile(
          "global_load_dwordx4 v[236:239], %[src], off sc1\n\t"
          "global_load_dwordx4 v[240:243], %[src], off offset:16 sc1\n\t"
          "s_sleep %[stag]\n\t"
          "global_load_dwordx4 v[244:247], %[src], off sc1\n\t"
          "global_load_dwordx4 v[248:251], %[src], off offset:16 sc1\n\t"
          "s_movk_i32 %[cnt], 64\n"
          ".Lp2_top_%=:\n\t"
          "s_waitcnt vmcnt(2)\n\t"
          "v_cmp_eq_u32_e32 vcc, %[tag], v237\n\t"
          "v_cmp_eq_u32_e64 %[tmp], %[tag], v239\n\t"
          "s_and_b64 vcc, vcc, %[tmp]\n\t"
          "v_cmp_eq_u32_e64 %[tmp], %[tag], v241\n\t"
          "s_and_b64 vcc, vcc, %[tmp]\n\t"
          "v_cmp_eq_u32_e64 %[tmp], %[tag], v243\n\t"
          "s_and_b64 vcc, vcc, %[tmp]\n\t"
          "s_cmp_eq_u64 vcc, exec\n\t"
          "s_cbranch_scc1 .Lp2_a_%=\n\t"
          "global_load_dwordx4 v[236:239], %[src], off sc1\n\t"
          "global_load_dwordx4 v[240:243], %[src], off offset:16 sc1\n\t"
          "s_waitcnt vmcnt(2)\n\t"
          "v_cmp_eq_u32_e32 vcc, %[tag], v245\n\t"
          "v_cmp_eq_u32_e64 %[tmp], %[tag], v247\n\t"
          "s_and_b64 vcc, vcc, %[tmp]\n\t"
          "v_cmp_eq_u32_e64 %[tmp], %[tag], v249\n\t"
          "s_and_b64 vcc, vcc, %[tmp]\n\t"
          "v_cmp_eq_u32_e64 %[tmp], %[tag], v251\n\t"
          "s_and_b64 vcc, vcc, %[tmp]\n\t"
          "s_cmp_eq_u64 vcc, exec\n\t"
          "s_cbranch_scc1 .Lp2_b_%=\n\t"
          "global_load_dwordx4 v[244:247], %[src], off sc1\n\t"
          "global_load_dwordx4 v[248:251], %[src], off offset:16 sc1\n\t"
          "s_sleep 1\n\t"
          "s_sub_u32 %[cnt], %[cnt], 1\n\t"
          "s_cmp_lg_u32 %[cnt], 0\n\t"
          "s_cbranch_scc1 .Lp2_top_%=\n\t"
          "s_waitcnt vmcnt(0)\n\t"                    // nothing yet: everything lands, the caller looks at the error word
          "s_mov_b32 %[st], 0\n\t"
          "s_branch .Lp2_end_%=\n"
          ".Lp2_a_%=:\n\t"
          "v_mov_b32 %[o0], v236\n\t"
          "v_mov_b32 %[o1], v238\n\t"
          "v_mov_b32 %[o2], v240\n\t"
          "v_mov_b32 %[o3], v242\n\t"
          "s_mov_b32 %[st], 1\n\t"
          "s_branch .Lp2_end_%=\n"
          ".Lp2_b_%=:\n\t"
          "v_mov_b32 %[o0], v244\n\t"
          "v_mov_b32 %[o1], v246\n\t"
          "v_mov_b32 %[o2], v248\n\t"
          "v_mov_b32 %[o3], v250\n\t"
          "s_mov_b32 %[st], 1\n"
          ".Lp2_end_%=:"
          : "={v[236:239]}"(pa[0]), "={v[240:243]}"(pa[1]), "={v[244:247]}"(pb[0]), "={v[248:251]}"(pb[1]), [o0] "=&v"(o0), [o1] "=&v"(o1),
            [o2] "=&v"(o2), [o3] "=&v"(o3), [st] "=&s"(st), [cnt] "=&s"(cnt), [tmp] "=&s"(tmp)
          : [src] "v"(src), [tag] "s"(tag), [stag] "n"(QUIP_POLL2_STAGGER)
          : "vcc", "scc", "memory");
      if (st != 0u) break;
      // 128 polls without the pieces: the launch-wide error word, the bound (spin_step's arithmetic)
      uint32_t e;
      esync::ld4(e, ctl + 1);
      esync::drain();
      esync::own(e);
      if (__builtin_amdgcn_readfirstlane(e) != 0u) break;
      rounds += 128u;
      if (rounds >= esync::kSpinLimit) {
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) esync::st_word(ctl + 1, code + (uint32_t)w);
        break;
      }
    }
    had::unpack8(make_uint4(o0, o1, o2, o3), out[0]);
    own_slots(slots);
  };
  // this workgroup's 16 values of a product (accumulator rows [row0, row0 + 16), block exponent sh) -> 8 granules
  auto publish16 = [&](int vec, int gr0, int row0, int sh, uint32_t tag) {      // gr0: first granule of the block in its vector
    if (tid < 8) {
      const int* s3 = accs + (row0 + 2 * tid) * 4;
      const float us = unscale_of(sh, kUnsc);              // table entries are 4 w (E8P12) / 2 w (D4); nibble mode: 8 x 4 w
      const float f0 = __builtin_fmaf((float)s3[0], 65536.f, __builtin_fmaf((float)s3[1], 256.f, (float)s3[2]));
      const float f1 = __builtin_fmaf((float)s3[4], 65536.f, __builtin_fmaf((float)s3[5], 256.f, (float)s3[6]));
      esync::st_granule(zbufs + (size_t)vec * 2048 + gr0 + tid, pack_f16(f0 * us, f1 * us), tag);
    }
  };
  auto zero_acc = [&](int row0, int rows) {
    for (int i = tid; i < rows * 4; i += kThreads) accs[row0 * 4 + i] = 0;
  };

  // ---- output side of the producer (+ residual) and the input transforms of up to two consumers --------------------------
  //   HAVE_Z: gather z (hand-off `tag`), then h += SV_prev (.) H z / 64                    [qlinear.py:106-114 of the producer]
  //   NC consumers i (NC = 0 | 2; `two` false: only consumer 0): planes_i = digits( sc_i * rms(h) * H (h (.) ln (.) su_i) )
  //   [RMSNorm + qlinear.py:90-100]; planes of consumer i at area + i * 3 * KV, block exponent in shs[i]
  // Round 5: the edge in the shape of the 8192-wide launch (decode_block_gqa.hip, fht_wg512x.hip.h) --
  //   gather (natural order) -> fwd -> strided layout: residual (registers), sum of squares, ln, SU -> rev -> natural order:
  //   digit planes as 8 / 16-byte pieces.  No layout change through LDS between the two transforms, the residual stream never
  //   leaves its registers, and the three reductions of the old edge (sum of squares, two maxima: three barrier pairs) ride
  //   on rev's own barriers: the block exponent comes from the NORM bound |H x|_inf <= sqrt(4096) |x|_2 (hadamard.hip's
  //   bound for the same planes; up to 4 of the 22 bits of X idle, 2^-18 of the maximum per digit: far below the fp16 output)
  //   whose sum is known BEFORE the transform.  10 barriers -> 6; 10.6K -> ~6K clocks per edge (profiles/r05_block_stamps.txt).
  // sv_prev, ln, su0, su1: vectors PERMUTED by the host to the strided layout, p[8 t + k] = v[t + 512 k].
  // `rev` adds in another order than the stand-alone kernels: the launch is no longer bit identical to the stage-wise step
  // (tests/test_gpu_block_engine.py states the bound, tests/test_gpu_decode.py the distance to the float64 model).
  float* xbuf = reinterpret_cast<float*>(smem + B::kBuf0);
  auto planes_nat = [&](const float (&v)[8], float scale, int sh, uint32_t base) {
#pragma clang fp contract(off)
    const float p2 = as_f32((uint32_t)(sh + 127) << 23);
    const float s2 = had::fmul(scale, p2);
    // (digits straight from the fp32 magic number: hadw::digit_words_magic)
    const float va[4] = {v[0], v[1], v[2], v[3]}, vb[4] = {v[4], v[5], v[6], v[7]};
    if constexpr (HI) {
      // [x0 x2 0 0 | x4 x6 0 0 | x1 x3 0 0 | x5 x7 0 0] (planes_scatter_hi's positions), plane stride 2 * 4096
      const float e0[4] = {v[0], v[2], 0.f, 0.f}, e1[4] = {v[4], v[6], 0.f, 0.f}, e2[4] = {v[1], v[3], 0.f, 0.f}, e3[4] = {v[5], v[7], 0.f, 0.f};
      uint32_t dg[3][4];
      hadw::digit_words_magic(e0, s2, dg[0][0], dg[1][0], dg[2][0]);
      hadw::digit_words_magic(e1, s2, dg[0][1], dg[1][1], dg[2][1]);
      hadw::digit_words_magic(e2, s2, dg[0][2], dg[1][2], dg[2][2]);
      hadw::digit_words_magic(e3, s2, dg[0][3], dg[1][3], dg[2][3]);
#pragma unroll
      for (int d = 0; d < 3; ++d)      // (a zero value's three digits are zero bytes: nothing to mask)
        *reinterpret_cast<uint4*>(smem + base + d * 2 * HID + 16 * tid) = make_uint4(dg[d][0], dg[d][1], dg[d][2], dg[d][3]);
    } else if constexpr (RVQ) {
      // x' = [s x_g | x_g]: the 8-group's residual-side digits, then its main-side digits (planes_scatter_rvq's positions)
      const float s2r = had::fmul(had::fmul(scale, a.resid_scale), p2);
      uint32_t dg[3][4];
      hadw::digit_words_magic(va, s2r, dg[0][0], dg[1][0], dg[2][0]);
      hadw::digit_words_magic(vb, s2r, dg[0][1], dg[1][1], dg[2][1]);
      hadw::digit_words_magic(va, s2, dg[0][2], dg[1][2], dg[2][2]);
      hadw::digit_words_magic(vb, s2, dg[0][3], dg[1][3], dg[2][3]);
#pragma unroll
      for (int d = 0; d < 3; ++d)
        *reinterpret_cast<uint4*>(smem + base + d * 2 * HID + 16 * tid) = make_uint4(dg[d][0], dg[d][1], dg[d][2], dg[d][3]);
    } else {
      uint32_t dg[3][2];
      hadw::digit_words_magic(va, s2, dg[0][0], dg[1][0], dg[2][0]);
      hadw::digit_words_magic(vb, s2, dg[0][1], dg[1][1], dg[2][1]);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if constexpr (NIB) {       // the thread's 8-group: positions 0..3 -> the "lo" half, 4..7 -> the "hi" half
          *reinterpret_cast<uint32_t*>(smem + base + d * B::PSH + 4 * tid) = dg[d][0];
          *reinterpret_cast<uint32_t*>(smem + base + d * B::PSH + B::HOH + 4 * tid) = dg[d][1];
        } else {
          *reinterpret_cast<uint2*>(smem + base + d * HID + 8 * tid) = make_uint2(dg[d][0], dg[d][1]);
        }
      }
    }
  };
  // block exponent of planes made of H x * scale, from the sum of squares of x (4096 values): |H x|_inf <= 64 |x|_2
  auto norm_shift = [&](float sumsq_x, float scale) -> int {
    const float bound = sqrtf(sumsq_x) * 64.f * fabsf(scale) * 1.0625f;
    return had::shift_for(bound * ((RVQ && !HI) ? fmaxf(1.f, fabsf(a.resid_scale)) : 1.f));
  };
  auto red_sum8 = [&](int at) -> float {               // the eight waves' partial sums in order
    float r = red[at];
#pragma unroll
    for (int i = 1; i < 8; ++i) r = had::fadd(r, red[at + i]);
    return r;
  };
  auto edge = [&](auto z_tag, auto nc_tag, auto slots, int zvec, uint32_t tag, uint32_t code, const f16* sv_prev, const f16* ln,
                  const f16* su0, const f16* su1, float sc0, float sc1, bool two, auto after_gather, auto drip1, auto drip2, int sb, auto ag_tag) {
    // ag_tag: VMEM requests `after_gather` issues (a hand count: the wait for the gather's second read set is said in them)
    // stamps of the edge's stages: 18..22 (sb = 18: the gate / up edge) or 23, 24, 28, 29, 30 (sb = 23: the q / k / v edge)
#if QUIP_ATT_STAMPS
#define ESTAMP(i) do { } while (0)
#else
#define ESTAMP(i) do { if (sb >= 0) BSTAMP(sb + (i) + ((sb == 23 && (i) >= 2) ? 3 : 0)); } while (0)
#endif
    constexpr int NC = decltype(nc_tag)::value;
    constexpr bool HAVE_Z = decltype(z_tag)::value;
    u32x4 psv, pln, psu0, psu1;
    if constexpr (HAVE_Z) psv = *reinterpret_cast<const u32x4*>(sv_prev + 8 * tid);
    if constexpr (NC > 0) {
      pln = *reinterpret_cast<const u32x4*>(ln + 8 * tid);
      psu0 = *reinterpret_cast<const u32x4*>(su0 + 8 * tid);
      psu1 = *reinterpret_cast<const u32x4*>(su1 + 8 * tid);
    }
    auto u4 = [](const u32x4& v) { return make_uint4(v.x, v.y, v.z, v.w); };
    if constexpr (HAVE_Z) {
      float v[1][8];
#if QUIP_POLL2
      u32x4_t pa[2], pb[2];
      gather2(slots, zvec, tag, code, v, pa, pb);
#else
      gather(std::integral_constant<int, 1>{}, slots, zvec, tag, code, v);
#endif
      // The vectors have landed (the gather drained the queue).  Take them over HERE: the compiler counts only its own
      // loads, so the wait it would place at their first use would also wait for the burst requested below.
      asm volatile("" : "+v"(psv));
      if constexpr (NC > 0) asm volatile("" : "+v"(pln), "+v"(psu0), "+v"(psu1));
      after_gather();
      ESTAMP(0);
      hadw::fwd<12, 1, true>(v, xbuf, tid);
#if QUIP_POLL2
      // the read set that was still in flight when the other one succeeded has had the transform's time to land
      if (dbg_on) esync::drain();                      // (a stamp is a store: one more request than counted)
      asm volatile("s_waitcnt vmcnt(%0)" : : "n"(decltype(ag_tag)::value), "{v[236:239]}"(pa[0]), "{v[240:243]}"(pa[1]), "{v[244:247]}"(pb[0]),
                   "{v[248:251]}"(pb[1]) : "memory");
#endif
      ESTAMP(1);
      float svf[8];
      had::unpack8(u4(psv), svf);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16x2 hh = as_f16x2(hreg[j]);
        const f16 n0 = had::out_elem(v[0][2 * j], 1.f / 64.f, true, svf[2 * j], false, 0.f, true, (float)hh.x);
        const f16 n1 = had::out_elem(v[0][2 * j + 1], 1.f / 64.f, true, svf[2 * j + 1], false, 0.f, true, (float)hh.y);
        hreg[j] = (uint32_t)__builtin_bit_cast(uint16_t, n0) | ((uint32_t)__builtin_bit_cast(uint16_t, n1) << 16);
      }
      drip1();
    }
    if constexpr (NC > 0) {
      float e[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f16x2 hh = as_f16x2(hreg[j]);
        e[2 * j] = (float)hh.x;
        e[2 * j + 1] = (float)hh.y;
      }
      // the sum of squares of h (the RMSNorm statistic) and of each consumer's input (its planes' bound): per-wave partial
      // sums into LDS HERE, totals read behind rev's barriers -- no barrier of their own
      float ssw = 0.f;
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int r = 0; r < 8; ++r) ssw = __builtin_fmaf(e[r], e[r], ssw);
      }
      had::mul8(e, u4(pln));
      const int lane_ = tid & 63;
      if (two) {
        float v[2][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[0][r] = v[1][r] = e[r];
        had::mul8(v[0], u4(psu0));
        had::mul8(v[1], u4(psu1));
        float n0 = 0.f, n1 = 0.f;
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int r = 0; r < 8; ++r) { n0 = __builtin_fmaf(v[0][r], v[0][r], n0); n1 = __builtin_fmaf(v[1][r], v[1][r], n1); }
        }
        ssw = had::wave_reduce_to_lane63<false>(ssw);
        n0 = had::wave_reduce_to_lane63<false>(n0);
        n1 = had::wave_reduce_to_lane63<false>(n1);
        if (lane_ == 63) { red[wave] = ssw; red[8 + wave] = n0; red[16 + wave] = n1; }
        ESTAMP(2);
        hadw::rev<12, 2, true, true>(v, xbuf, tid);
        ESTAMP(3);
        drip2();
        const float tot = red_sum8(0);
        const float s0 = had::rms_scale(sc0, tot, HID, a.rms_eps), s1 = had::rms_scale(sc1, tot, HID, a.rms_eps);
        const int sh0 = norm_shift(red_sum8(8), s0), sh1 = norm_shift(red_sum8(16), s1);
        planes_nat(v[0], s0, sh0, (uint32_t)B::kArea);
        planes_nat(v[1], s1, sh1, (uint32_t)(B::kArea + 3 * B::PSH));
        if (tid == 0) { shs[0] = sh0; shs[1] = sh1; }
      } else {
        float v[1][8];
#pragma unroll
        for (int r = 0; r < 8; ++r) v[0][r] = e[r];
        had::mul8(v[0], u4(psu0));
        float n0 = 0.f;
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int r = 0; r < 8; ++r) n0 = __builtin_fmaf(v[0][r], v[0][r], n0);
        }
        ssw = had::wave_reduce_to_lane63<false>(ssw);
        n0 = had::wave_reduce_to_lane63<false>(n0);
        if (lane_ == 63) { red[wave] = ssw; red[8 + wave] = n0; }
        ESTAMP(2);
        hadw::rev<12, 1, true, true>(v, xbuf, tid);
        ESTAMP(3);
        drip2();
        const float tot = red_sum8(0);
        const float s0 = had::rms_scale(sc0, tot, HID, a.rms_eps);
        const int sh0 = norm_shift(red_sum8(8), s0);
        planes_nat(v[0], s0, sh0, (uint32_t)B::kArea);
        if (tid == 0) shs[0] = sh0;
      }
      had::wg_barrier<true>();
      ESTAMP(4);
    }
    // (opaque from here on: otherwise the fp32 images of h made above stay alive until the next edge's update)
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(hreg[j]));
#undef ESTAMP
  };

  // ---- one item of a product: slot s, digit planes at xa ------------------------------------------------------------
  // the A fragments of a K slice (byte tables: eight; nibble mode: four + their digit sums)
  struct Frag { i32x4 A[8]; NibFrags N; };
  auto frags = [&](uint32_t xa, Frag& F) {
    if constexpr (NIB) nib_fragments(xa, F.N); else item_fragments(xa, F.A);
  };
  auto frags_d = [&](uint32_t xa, Frag& F) {            // down's planes (nibble mode: 144-byte blocks, see BLds::kHalfD)
    if constexpr (NIB) nib_fragments<144>(xa, F.N); else item_fragments(xa, F.A);
  };
  // the item in slot registers (da, db) against the fragments F
  auto mul_slot = [&](const slot_t& da, const slot_t& db, const Frag& F) -> i32x4 {
    ItemAddr ad;
    item_addresses<REP>(da, db, lane_c, lane_c2, ad, lane_c3);
    if constexpr (NIB) return nib_item(ad, F.N, nlf);
    else return item_mfma_shared<T::kD4, R3>(ad, F.A);
  };
  // an item decoded earlier (pre: its B fragments as scalars -- byte tables: 32 dwords; nibble mode: 16) against the fragments F
  auto mul_pre = [&](const uint32_t* pre, const Frag& F) -> i32x4 {
    if constexpr (NIB) {
      return nib_multiply(pre, F.N, nlf);
    } else {
      i32x4 r = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const i32x4 Bt = {(int)pre[4 * t], (int)pre[4 * t + 1], (int)pre[4 * t + 2], (int)pre[4 * t + 3]};
        r = __builtin_amdgcn_mfma_i32_16x16x64_i8(F.A[t], Bt, r, 0, 0, 0);
      }
      return r;
    }
  };
  // the table look-ups of the item in slot s now (inside a hand-off's wait), its MFMAs later: B fragments as scalar registers
  auto decode_pre = [&](int s, uint32_t (&pre)[kPreW]) {
    ItemAddr ad;
    item_addresses<REP>(qa[s], qb[s], lane_c, lane_c2, ad, lane_c3);
    if constexpr (NIB) {
      uint32_t raw[16];
      nib_decode(ad, raw);
#pragma unroll
      for (int t = 0; t < 16; ++t) { asm volatile("" : "+v"(raw[t])); pre[t] = raw[t]; }
    } else {
      i32x4 Bt[8];
      if constexpr (T::kD4) item_decode_d4(ad, Bt); else item_decode<R3>(ad, Bt);
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        pre[4 * t] = (uint32_t)Bt[t].x; pre[4 * t + 1] = (uint32_t)Bt[t].y; pre[4 * t + 2] = (uint32_t)Bt[t].z; pre[4 * t + 3] = (uint32_t)Bt[t].w;
        asm volatile("" : "+v"(pre[4 * t]), "+v"(pre[4 * t + 1]), "+v"(pre[4 * t + 2]), "+v"(pre[4 * t + 3]));
      }
    }
  };
  // which lanes own accumulator rows: byte tables q == 0 (D rows 0..2); nibble mode q < 2 (+ D rows 4..6 of the same column)
  auto owns_rows = [&]() -> bool { return NIB ? q < 2 : q == 0; };
  auto run_item = [&](int s, uint32_t xa, int accrow) {
    static_assert(!NIB || true, "");
    ItemAddr ad;
    item_addresses<REP>(qa[s], qb[s], lane_c, lane_c2, ad, lane_c3);
    i32x4 r;
    if constexpr (NIB) { Frag F; frags(xa, F); r = nib_item(ad, F.N, nlf); }
    else r = T::kD4 ? item_mfma_d4(ad, xa) : item_mfma<256, R3>(ad, xa);
    if (owns_rows()) {
      int* dst = accs + (accrow + n) * 4;
      __hip_atomic_fetch_add(dst + 0, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst + 1, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst + 2, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };

  // three items of one K slice (the row blocks of a matrix share the slice's A fragments; xa differs between items only where
  // a workgroup's row blocks straddle two matrices: re-read then -- a uniform branch)
  auto run_items3 = [&](int s0, uint32_t xa0, uint32_t xa1, uint32_t xa2, int accrow0) {
    Frag A;
    frags(xa0, A);
    const uint32_t xas[3] = {xa0, xa1, xa2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i > 0 && xas[i] != xas[i - 1]) frags(xas[i], A);
      const i32x4 r = mul_slot(qa[s0 + i], qb[s0 + i], A);
      if (owns_rows()) {
        int* dst = accs + (accrow0 + 16 * i + n) * 4;
        __hip_atomic_fetch_add(dst + 0, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 1, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 2, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  // three items of one K slice whose first NPRE were decoded earlier (pre: their eight B fragments each, as scalars)
  auto run_items3_pre = [&](auto npre_tag, const uint32_t (&pre)[decltype(npre_tag)::value][kPreW], int s0, uint32_t xa0, uint32_t xa1, uint32_t xa2, int accrow0) {
    constexpr int NPRE = decltype(npre_tag)::value;
    Frag A;
    frags(xa0, A);
    const uint32_t xas[3] = {xa0, xa1, xa2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i > 0 && xas[i] != xas[i - 1]) frags(xas[i], A);
      i32x4 r = {0, 0, 0, 0};
      if (i < NPRE) r = mul_pre(pre[i < NPRE ? i : 0], A);
      else r = mul_slot(qa[s0 + i], qb[s0 + i], A);
      if (owns_rows()) {
        int* dst = accs + (accrow0 + 16 * i + n) * 4;
        __hip_atomic_fetch_add(dst + 0, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 1, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(dst + 2, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  // the same item in two halves: the table lookups while a hand-off is awaited (the codes are there long before the digits),
  // the eight MFMAs once the digit planes exist
  auto decode_item = [&](int s, i32x4 (&Bf)[8]) {       // (byte tables: o's item waits as eight 128-bit tuples)
    ItemAddr ad;
    item_addresses<REP>(qa[s], qb[s], lane_c, lane_c2, ad, lane_c3);
    if constexpr (T::kD4) item_decode_d4(ad, Bf); else if constexpr (!NIB) item_decode<R3>(ad, Bf);
  };
  auto add_rows = [&](const i32x4& r, int accrow) {
    if (owns_rows()) {
      int* dst = accs + (accrow + n) * 4;
      __hip_atomic_fetch_add(dst + 0, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst + 1, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      __hip_atomic_fetch_add(dst + 2, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };

  const f16* sv_d_prev = nullptr;
  const BlockLayer& Ld = *reinterpret_cast<const BlockLayer*>(smem + B::kDesc);
  const BlockLayer& Ln = *reinterpret_cast<const BlockLayer*>(smem + B::kDescN);
  constexpr bool kPreQkv = QUIP_PREDECODE_QKV > 0 && !RVQ;
  // slot of down's item i | of the (next block's) q / k / v item i
  constexpr auto SDf = [](int i) { return G8 ? (i < 2 ? 7 + i : i - 2) : 6 + i; };
  constexpr auto SQf = [](int i) { return G8 ? 2 + i : i; };
  if (tid < 64) reinterpret_cast<uint32_t*>(smem + B::kDesc)[tid] = reinterpret_cast<const uint32_t*>(a.layers)[tid];
  had::wg_barrier<true>();
  // ================= P1: [output side of the previous block's down_proj + residual,] RMSNorm, input transforms of q, k, v of
  // the block whose descriptor is in LDS; their products; hand-off ========================================================
  // (called for block 0 here and, for block l + 1, at the bottom of iteration l: the requests of q, k, v go out BEHIND the
  //  hand-off of z_d and are consumed before the loop's back edge, where no request may be in flight -- the compiler is free to
  //  copy registers there)
  constexpr int NPQ = NIB ? (G8 ? 2 : QUIP_NIB_PREDECODE_QKV) : (QUIP_PREDECODE_QKV > 0 ? QUIP_PREDECODE_QKV : 1);
  constexpr int kPreG = NIB ? (G8 ? 2 : QUIP_NIB_PREDECODE_GATE) : QUIP_PREDECODE_GATE, kPreD = (NIB && !G8) ? QUIP_NIB_PREDECODE_DOWN : QUIP_PREDECODE_DOWN;
  // the first / last of the one or two matrices (q 0, k 1, v 2) this workgroup's q / k / v row blocks are in
  const int qc_lo = G8 ? qmat(qb0) : ((3 * w) >> 8), qc_hi = G8 ? qmat(qb0 + qcnt - 1) : ((3 * w + 2) >> 8);
  // one item against the A fragments of its K slice: decoded earlier (pre, as scalars) or from slot `slot_c`
  auto item_vs = [&](auto slot_c, const uint32_t* pre, const Frag& A) -> i32x4 {
    constexpr int S = decltype(slot_c)::value;
    if (pre) return mul_pre(pre, A);
    return mul_slot(qa[S], qb[S], A);
  };
  auto P1_products = [&](auto pre_tag, const uint32_t (&preq)[NPQ][kPreW]) {
    const int c_lo = qc_lo;
    esync::drain();                                    // q, k, v have landed
    own_slots(SLOTS(M_QKV));
    if constexpr (G8) {
      // one or two items (slots X2, X3); the second one's planes are consumer 1's where it lies in the next matrix (workgroup 170)
      constexpr bool PRE = decltype(pre_tag)::value;
      Frag A;
      frags(xlane, A);
      add_rows(item_vs(std::integral_constant<int, 2>{}, PRE ? preq[0] : nullptr, A), 0);
      if (qcnt == 2) {
        if (qc_hi != qc_lo) frags(xlane + (uint32_t)(3 * B::PSH), A);
        add_rows(item_vs(std::integral_constant<int, 3>{}, (PRE && NPQ > 1) ? preq[NPQ > 1 ? 1 : 0] : nullptr, A), 16);
      }
      had::wg_barrier<true>();
      ++hop;                                           // hand-off: z_q, z_k, z_v
      const int sh_lo = shs[0], sh_hi = shs[1];
      publish16(qc_lo, qblk(qb0) * 8, 0, sh_lo, ebase | hop);
      if (qcnt == 2) publish16(qc_hi, qblk(qb0 + 1) * 8, 16, qc_hi == qc_lo ? sh_lo : sh_hi, ebase | hop);
      had::wg_barrier<true>();
      zero_acc(0, 48);
      BSTAMP(3);
      return;
    }
    {
      const uint32_t x0 = xlane + (uint32_t)((((3 * w) >> 8) == c_lo ? 0 : 1) * 3 * B::PSH);
      const uint32_t x1 = xlane + (uint32_t)((((3 * w + 1) >> 8) == c_lo ? 0 : 1) * 3 * B::PSH);
      const uint32_t x2 = xlane + (uint32_t)((((3 * w + 2) >> 8) == c_lo ? 0 : 1) * 3 * B::PSH);
      if constexpr (decltype(pre_tag)::value) run_items3_pre(std::integral_constant<int, NPQ>{}, preq, 0, x0, x1, x2, 0);
      else run_items3(0, x0, x1, x2, 0);
      if constexpr (RVQ) run_items3(3, x0 + 4096u, x1 + 4096u, x2 + 4096u, 0);      // the second virtual slice: wave + 8
    }
    had::wg_barrier<true>();
    ++hop;                                             // hand-off: z_q, z_k, z_v
    const int sh_lo = shs[0], sh_hi = shs[1];          // (both, then a select: a selected ADDRESS was hoisted out of the block loop into a register of its own)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int rbq = 3 * w + i, ci = rbq >> 8;
      publish16(ci, (rbq & 255) * 8, i * 16, ci == c_lo ? sh_lo : sh_hi, ebase | hop);
    }
    had::wg_barrier<true>();
    zero_acc(0, 48);
    BSTAMP(3);
  };
  {
    rederive();
    const int c_lo = qc_lo, c_hi = qc_hi;
    edge(std::false_type{}, std::integral_constant<int, 2>{}, SLOTS(M_QKV), -1, 0u, 0u, nullptr, Ld.ln[0], Ld.su[c_lo], Ld.su[c_hi],
         Ld.sc[c_lo], Ld.sc[c_hi], c_hi != c_lo, [&]() {}, [&]() {}, [&]() {}, -1, std::integral_constant<int, 0>{});
    uint32_t none[NPQ][kPreW];                          // (block 0: nothing decoded ahead; never read)
    P1_products(std::false_type{}, none);
  }
  for (int l = 0; l < a.n_layers; ++l) {
    dbg_on = a.dbg != nullptr && l == (a.dbg_layer & 0xffff);
    rederive();
    BSTAMP(0);
    // the NEXT block's descriptor (the last block: its own once more), for the early requests of its q, k, v rows; read long
    // after the barriers that follow
    if (tid < 64)
      reinterpret_cast<uint32_t*>(smem + B::kDescN)[tid] = reinterpret_cast<const uint32_t*>(a.layers + (l + 1 < a.n_layers ? l + 1 : l))[tid];
    // o of this block, behind the publication.  (A burst in front of a gather makes the gather's first check wait for the
    // burst -- vmcnt retires in order: ~2.3 us of HBM latency against the ~1.7 us a hand-off takes -- and a burst anywhere
    // else stalls the issuing waves for ~1.2K clocks, the CU's address unit taking ~25 clocks per 1 KB request.  Measured per
    // burst: gate, up and down are cheaper behind their hand-off; o, which only the head's workgroups would gain from, here.)
    if constexpr (RVQ) ISSUE_RVQ_O(Ld); else ISSUE(Ld, 3);

    // ================= P2: attention ====================================================================================
    rederive();
    // Short contexts: the workgroup that owns the first rows of a head (w % 8 == 0) runs its attention alone, the other seven
    // wait for the result.  From kSplitPos positions on all eight take part: workgroup `part` every eighth position, and
    // their partial states (maximum, denominator, 128 unnormalised sums) meet at the first one through one more hand-off
    // (2 us, against 21 ns per position and block on one workgroup: 43 us per block at 2048 positions).
    const bool split = pos >= kSplitPos;
    const int part = w & 7, nparts = split ? kParts : 1;
    const bool head_wg = part == 0;
    const int hd = w >> 3;
    if (head_wg || split) {
      float c8[8], s8[8];
      const int d0 = (tid & 15) * 8;
      // the cached rows of this head depend on nothing: the first two rounds (2 x 64 positions) are requested HERE / right after the gather and land
      // while the hand-off is awaited / under the transforms (they come from HBM -- a token's weight stream has flushed every cache since they were
      // written: 2 us per dependent round otherwise); later rounds are requested two rounds ahead of their use
      // (U: cached rows per key group and round; E8P12RVQ4B, at 256 registers, prefetches half as many -- the keys are
      //  visited in the same order either way)
      constexpr int ATH = QUIP_ATT_THREADS;
      constexpr int LPK = HD / 8, NG = ATH / LPK, U = (RVQ && !HI && !R3) ? 2 : 4;
      constexpr int NST = ATH / 64;                    // online-softmax states that meet in LDS: one per wave
      const int g = (tid & (ATH - 1)) / LPK;
      const int kvh = hd / GQH;                        // the KV head of this query head (G8: four query heads per KV head)
      const f16* kc = Ld.kcache + (size_t)kvh * a.max_len * HD;
      const f16* vc = Ld.vcache + (size_t)kvh * a.max_len * HD;
      uint4 kr0[U], vr0[U], kr1[U], vr1[U];
      // local index i of this workgroup <-> position part + nparts i; a round = local indices i0 + u NG, u < U
      auto load_round = [&](uint4 (&kr)[U], uint4 (&vr)[U], int i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int t = part + nparts * (i0 + u * NG);
          const int tc = t < pos ? t : 0;
          kr[u] = *reinterpret_cast<const uint4*>(kc + (size_t)tc * HD + d0);
          vr[u] = *reinterpret_cast<const uint4*>(vc + (size_t)tc * HD + d0);
        }
      };
      if (tid < ATH) {
        load_round(kr0, vr0, g);
        if (part + nparts * NG * U < pos) load_round(kr1, vr1, g + NG * U);
      }
      // SV of this head's q / k / v values (waves 0..2 finish one vector each below): requested HERE, in front of the hand-off's
      // wait -- read at its use it was a memory latency on the heads' critical path
      uint32_t svp_raw = 0u;
      if (wave < 3) svp_raw = *reinterpret_cast<const uint32_t*>(Ld.sv[wave] + HD * ((G8 && wave > 0) ? kvh : hd) + 2 * lane);
      uint32_t suo_raw = 0u;                           // (kOHead: SU_o of this head's 128 attention outputs, natural order)
      if constexpr (kOHead) { if (wave == 0) suo_raw = *reinterpret_cast<const uint32_t*>(Ld.su[3] + HD * hd + 2 * lane); }
      f16* s_qkv = reinterpret_cast<f16*>(smem + B::kQkv);
      {
        // Round 5: only this head's 128 values of H_4096 z are needed, for z = z_q, z_k, z_v.  H_4096 = H_32 (x) H_128 with the
        // index split 128 j1 + j2, so they are H_128 u with u[j2] = sum_j1 H_32[hd][j1] z[128 j1 + j2]: signed sums of the 32
        // segments, then a 128-point transform in ONE wave per vector -- instead of a 4096-point transform of q here and of k, v
        // on two more workgroups with one more hand-off inside the head's group (rounds 3-4: 1.8K clocks + that hop).
        // Thread t holds z[8 t + r]: j1 = t >> 4 = (lane >> 4) + 4 wave, j2 = 8 (t & 15) + r.
        // G8: z_k, z_v are 1024-vectors (512 granules): thread t takes granule t of each -- z[2 t], z[2 t + 1], segment j1 = t >> 6 =
        // the wave, so their signed segment sums (H_1024 = H_8 (x) H_128) are the sum over the waves' values alone.
        constexpr int NVV = G8 ? 1 : 3;                  // 4096-vectors that go through the lane swaps
        float v[NVV][8];
        u32x2_t gk = {0u, 0u}, gv = {0u, 0u};
        if constexpr (G8) {
          u32x4_t p[2];
          uint32_t spins = 0;
          const uint64_t* src = zbufs + 4 * tid;
          const uint32_t tagq = ebase | hop;
          for (;;) {
            esync::ld16(p[0], src);
            esync::ld16(p[1], src + 2);
            esync::ld8(gk, zbufs + 2048 + tid);
            esync::ld8(gv, zbufs + 2 * 2048 + tid);
            esync::drain();
            esync::own(p[0]); esync::own(p[1]); esync::own(gk); esync::own(gv);
            const bool ok = p[0].y == tagq && p[0].w == tagq && p[1].y == tagq && p[1].w == tagq && gk.y == tagq && gv.y == tagq;
            if (esync::spin_step(ok, spins, ctl + 1, 0x5000u + (uint32_t)w)) break;
          }
          had::unpack8(make_uint4(p[0].x, p[0].z, p[1].x, p[1].z), v[0]);
          own_slots(SLOTS(M_O));
        } else {
          gather(std::integral_constant<int, 3>{}, SLOTS(M_O), 0, ebase | hop, 0x5000u, v);
        }
        BSTAMP(4);
        // H_32[hd][j1] = (-1)^popcount(hd & j1): lane bit 5 <-> hd bit 1, lane bit 4 <-> hd bit 0 (folded into the two swap
        // steps), the wave <-> hd bits 2..4 (folded into the sum over the waves' partial results)
        // (scalar registers: as vector constants they were hoisted out of the block loop and held two registers for the whole launch)
        const float sg1 = as_f32((uint32_t)__builtin_amdgcn_readfirstlane((int)((hd & 2) ? 0xbf800000u : 0x3f800000u)));
        const float sg0 = as_f32((uint32_t)__builtin_amdgcn_readfirstlane((int)((hd & 1) ? 0xbf800000u : 0x3f800000u)));
        float P[4 * NVV], Q[2 * NVV];
#pragma unroll
        for (int pi = 0; pi < 4 * NVV; ++pi) {
          // lanes l and l ^ 32 of the values 2 pi and 2 pi + 1: lower half <- value 2 pi, upper half <- value 2 pi + 1
          const int va = 2 * pi, vb = 2 * pi + 1;
          const auto t = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[va >> 3][va & 7]),
                                                         __builtin_bit_cast(unsigned, v[vb >> 3][vb & 7]), false, false);
          P[pi] = __builtin_fmaf(__builtin_bit_cast(float, (unsigned)t[1]), sg1, __builtin_bit_cast(float, (unsigned)t[0]));
        }
#pragma unroll
        for (int si = 0; si < 2 * NVV; ++si) {
          // rows (16 lanes) r and r ^ 1: row 0 <- value 4 si, row 1 <- value 4 si + 2, row 2 <- 4 si + 1, row 3 <- 4 si + 3
          const auto t = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, P[2 * si]), __builtin_bit_cast(unsigned, P[2 * si + 1]), false, false);
          Q[si] = __builtin_fmaf(__builtin_bit_cast(float, (unsigned)t[1]), sg0, __builtin_bit_cast(float, (unsigned)t[0]));
        }
        // this wave's partial sums: [vector][wave][128]
        {
          const int row = lane >> 4, cc = lane & 15;
          const int vsel = ((row & 1) << 1) | (row >> 1);                  // {0, 2, 1, 3}[row]
#pragma unroll
          for (int si = 0; si < 2 * NVV; ++si) {
            const int vi = 4 * si + vsel;                                   // value index: vector vi >> 3, register vi & 7
            xbuf[((vi >> 3) * 8 + wave) * HD + 8 * cc + (vi & 7)] = Q[si];
          }
          if constexpr (G8) {      // k, v: this wave's segment as it is (the signs H_8[kv head][wave] ride on the sum below)
            const f16x2 hk = as_f16x2(gk.x), hv = as_f16x2(gv.x);
            *reinterpret_cast<float2*>(xbuf + (1 * 8 + wave) * HD + 2 * lane) = make_float2((float)hk.x, (float)hk.y);
            *reinterpret_cast<float2*>(xbuf + (2 * 8 + wave) * HD + 2 * lane) = make_float2((float)hv.x, (float)hv.y);
          }
        }
        had::wg_barrier<true>();
        if (wave < 3) {
          float y[2] = {0.f, 0.f};
#pragma unroll
          for (int gg = 0; gg < 8; ++gg) {
            // (the sign as a scalar xor mask: eight +-1.0 constants in vector registers were hoisted out of the block loop and
            //  lived -- and spilled -- across the whole launch)
            // (G8: hd >> 2 is also the KV head, and the waves are z_k's / z_v's eight segments: the same mask for all three)
            const uint32_t sm = ((uint32_t)__builtin_popcount((uint32_t)(hd >> 2) & (uint32_t)gg) & 1u) << 31;
            const float2 pr = *reinterpret_cast<const float2*>(xbuf + (wave * 8 + gg) * HD + 2 * lane);
            y[0] = had::fadd(y[0], as_f32(as_u32(pr.x) ^ sm));
            y[1] = had::fadd(y[1], as_f32(as_u32(pr.y) ^ sm));
          }
          hadw::reg_stage<2, 1>(y);
          hadw::lane_stages<2, 0, 6>(y, lane);
          const f16x2 svp = as_f16x2(svp_raw);
          const float osc = (G8 && wave > 0) ? 1.f / 32.f : 1.f / 64.f;      // 1 / sqrt(1024) | 1 / sqrt(4096)
          s_qkv[wave * HD + 2 * lane] = had::out_elem(y[0], osc, true, (float)svp.x, false, 0.f, false, 0.f);
          s_qkv[wave * HD + 2 * lane + 1] = had::out_elem(y[1], osc, true, (float)svp.y, false, 0.f, false, 0.f);
        }
      }
      // (the gather has drained the queue; said once more for tools/check_inflight.py)
      esync::drain();
      own_slots(SLOTS(M_O));
      had::wg_barrier<true>();
      BSTAMP(5);
      ++hop;                                           // hand-off inside the head's group (long contexts): partial states
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
      {
      // single-query attention of head hd over positions [0, pos] (decode_glue.hip's arithmetic): 16 lanes per key,
      // 16 key groups with their own online-softmax state, merged through LDS
      float* s_m = reinterpret_cast<float*>(smem + B::kArea);
      float* s_l = s_m + NST;
      float* s_acc = s_l + NST;                        // [NST][HD + 4]
      float q8[8], kn[8], vn[8];
      float m = -INFINITY, lsum = 0.f, acc8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc8[i] = 0.f; kn[i] = 0.f; vn[i] = 0.f; }
      // one key: score against q (16 lanes), online-softmax update of this group's state
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
        lsum = __builtin_fmaf(lsum, cc, pp);          // (decode_glue.hip's arithmetic, spelled out there and here)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc8[i] = __builtin_fmaf(acc8[i], cc, had::fmul(pp, v8[i]));
        m = mn;
      };
      if (tid < ATH) {
        rope8(s_qkv, q8);
#pragma unroll
        for (int i = 0; i < 8; ++i) q8[i] *= a.attn_scale;
        {
          rope8(s_qkv + HD, kn);
          const uint4 vraw = *reinterpret_cast<const uint4*>(s_qkv + 2 * HD + d0);
          unpack8h(vraw, vn);
          if (g == 0 && pos_ok && (hd % GQH) == 0 && part == (split ? (pos & (kParts - 1)) : 0)) {   // append the new row (StaticCache.update): once per KV head
            uint4 kr;
            kr.x = pack_f16(kn[0], kn[1]); kr.y = pack_f16(kn[2], kn[3]);
            kr.z = pack_f16(kn[4], kn[5]); kr.w = pack_f16(kn[6], kn[7]);
            *const_cast<uint4*>(reinterpret_cast<const uint4*>(kc + (size_t)pos * HD + d0)) = kr;
            *const_cast<uint4*>(reinterpret_cast<const uint4*>(vc + (size_t)pos * HD + d0)) = vraw;
          }
        }
        ASTAMP(0);
        // one round: positions t0 + u NG of this key group, rows in (kr, vr)
        auto round = [&](const uint4 (&kr)[U], const uint4 (&vr)[U], int i0) {
          float k8[U][8], v8[U][8], sc[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int t = part + nparts * (i0 + u * NG);
            (void)t;
            unpack8h(kr[u], k8[u]);
            unpack8h(vr[u], v8[u]);
            sc[u] = score(k8[u]);
          }
#pragma unroll
          for (int u = 0; u < U; ++u)
            if (part + nparts * (i0 + u * NG) < pos) update(sc[u], v8[u]);
        };
        // (uniform trip count: the lanes of a wave differ in g < NG only)
        const int n_loc = pos > part ? (pos - part + nparts - 1) / nparts : 0;       // local indices of this workgroup's cached rows
        for (int ib = 0; ib < n_loc; ib += 2 * NG * U) {
          round(kr0, vr0, ib + g);
          if (ib + 2 * NG * U < n_loc) load_round(kr0, vr0, ib + g + 2 * NG * U);
          if (ib + NG * U < n_loc) {
            round(kr1, vr1, ib + g + NG * U);
            if (ib + 3 * NG * U < n_loc) load_round(kr1, vr1, ib + g + 3 * NG * U);
          }
        }
        // the new row (position pos, still in registers) is the LAST key of its group: the rounds visit cached rows only -- as
        // a case inside them it cost sixteen register moves per key
        if (part == (split ? (pos & (kParts - 1)) : 0) && g == (((pos - part) / nparts) & (NG - 1))) update(score(kn), vn);
      }
      ASTAMP(1);
      if (tid < ATH) {
        // The four key groups of a wave (its four rows of 16 lanes) merge in registers first: v_permlane16_swap / 32_swap of a
        // value with ITSELF hands every lane both partners' copies -- (even row, odd row) / (lower half, upper half) -- so both
        // sides compute the same merged state and nobody selects.  Eight states meet in LDS instead of sixteen.
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
      had::wg_barrier<true>();
      ASTAMP(2);
      f16* s_a = s_qkv + 3 * HD;
      float pM = -INFINITY, pL = 0.f, pO = 0.f;        // this workgroup's state for dimension tid: maximum, denominator, sum
      if (tid < HD) {
        for (int g2 = 0; g2 < NST; ++g2) pM = fmaxf(pM, s_m[g2]);
        for (int g2 = 0; g2 < NST; ++g2) {
          const float ww = s_m[g2] == -INFINITY ? 0.f : __expf(s_m[g2] - pM);
          pL = __builtin_fmaf(s_l[g2], ww, pL);
          pO = __builtin_fmaf(s_acc[g2 * (HD + 4) + tid], ww, pO);
        }
      }
      if (split) {
        // hand-off: the partial states of the head's eight workgroups -> its first one
        const uint32_t tagp = tagg;
        uint64_t* mine_p = pbuf + ((size_t)hd * kParts + part) * kPartGran;
        if (tid < HD) esync::st_granule(mine_p + tid, as_u32(pO), tagp);
        if (tid == 0) { esync::st_granule(mine_p + HD, as_u32(pM), tagp); esync::st_granule(mine_p + HD + 1, as_u32(pL), tagp); }
        if (head_wg) {
          constexpr int PIECES = kParts * kPartGran / 2;        // 16-byte pieces of the head's partials (the padding is never written)
          float* s_p = reinterpret_cast<float*>(smem + B::kArea);                  // [parts][132]
          const uint64_t* srcp = pbuf + (size_t)hd * kParts * kPartGran;
          u32x4_t pp[2];
          uint32_t spins = 0;
          for (;;) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int i = tid + kThreads * j;
              esync::ld16(pp[j], srcp + 2 * (i < PIECES ? i : 0));
            }
            esync::drain();
            bool ok = true;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              esync::own(pp[j]);
              const int i = tid + kThreads * j;
              const int gi = 2 * (i < PIECES ? i : 0), within = gi % kPartGran;
              ok = ok && (within >= HD + 2 || (pp[j].y == tagp && (within + 1 >= HD + 2 || pp[j].w == tagp)));
            }
            if (esync::spin_step(ok, spins, ctl + 1, 0x8000u + (uint32_t)w)) break;
          }
          had::wg_barrier<true>();                       // s_m / s_l / s_acc have been read by everybody
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int i = tid + kThreads * j;
            if (i < PIECES) *reinterpret_cast<uint2*>(s_p + 2 * i) = make_uint2(pp[j].x, pp[j].z);
          }
          own_slots(SLOTS(M_O));
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
      ASTAMP(3);
      had::wg_barrier<true>();
      ASTAMP(4);
      if constexpr (kOHead) {
        if (head_wg && tid < 64) {
          // x = a (.) SU_o, its 128-point transform inside this wave (lane l: elements 2 l, 2 l + 1 before and after), two per granule
          const f16x2 av = as_f16x2(*reinterpret_cast<const uint32_t*>(s_a + 2 * tid)), su2 = as_f16x2(suo_raw);
          float y[2] = {had::fmul((float)av.x, (float)su2.x), had::fmul((float)av.y, (float)su2.y)};
          hadw::reg_stage<2, 1>(y);
          hadw::lane_stages<2, 0, 6>(y, tid);
          uint32_t w0, w1;
          esync::pack20x2(y[0], y[1], ebase | (hop + 1u), w0, w1);
          esync::st_granule(zbufs + (size_t)3 * 2048 + hd * 64 + tid, w0, w1);
        }
      } else if (head_wg && tid < 64) {
        const uint32_t pr = *reinterpret_cast<const uint32_t*>(s_a + 2 * tid);
        esync::st_granule(zbufs + (size_t)3 * 2048 + hd * 64 + tid, pr, ebase | (hop + 1u));
      }
      }
      ++hop;                                           // hand-off: attention output
    } else {
      hop += 2;                                        // (short context: five workgroups of a head wait for the result)
    }
    BSTAMP(6);
    rederive();
    i32x4 Bo[8];
    uint32_t Bon[kPreW];                                  // (nibble mode: o's item as 16 scalars)
    {
      // o's codes were requested a hand-off ago: their table lookups run HERE, inside the wait for the attention output
      // (RVQ: two items -- their fragments would be 64 registers: decoded with the product instead)
      esync::drain();
      own_slots(SLOTS(M_O));
      if constexpr (NIB) decode_pre(3, Bon);
      else if constexpr (!RVQ) decode_item(3, Bo);
      if constexpr (kOHead) {
        // o_proj's input side, the heads' half done by the heads: thread (wave, lane) takes columns [4 jq, +4) of heads
        // 2 (lane >> 2) + r, r < 2 -- one 16-byte piece (two granules) per head -- and finishes H_32 across the head index
        const int jq = 4 * wave + (lane & 3), hp = lane >> 2;
        const uint32_t t16 = (ebase | hop) & 0xffffu;
        {
          // (the attention output is ~8 us away: the wait is spent on ONE granule per wave, the gather behind it checks every piece)
          uint32_t sp = 0;
          u32x2_t f;
          const uint64_t* g1 = zbufs + (size_t)3 * 2048 + (size_t)((8 * w + wave) & (NH - 1)) * 64;
          for (;;) {
            esync::ld8(f, g1);
            esync::drain();
            esync::own(f);
            if (esync::spin_step((f.y >> 16) == t16, sp, ctl + 1, 0x6100u + (uint32_t)w)) break;
          }
        }
        u32x4_t p[2];
        {
          uint32_t spins = 0;
          const uint64_t* src = zbufs + (size_t)3 * 2048 + (size_t)(2 * hp) * 64 + 2 * jq;
          // (plain loads, as gather(): registers kept across the loop's back edge are the allocator's to copy while in flight)
          for (;;) {
            esync::ld16(p[0], src);
            esync::ld16(p[1], src + 64);
            esync::drain();
            esync::own(p[0]);
            esync::own(p[1]);
            const bool ok = (p[0].y >> 16) == t16 && (p[0].w >> 16) == t16 && (p[1].y >> 16) == t16 && (p[1].w >> 16) == t16;
            if (esync::spin_step(ok, spins, ctl + 1, 0x6000u + (uint32_t)w)) break;
          }
        }
        if constexpr (G8) ISSUE8_GU(Ld, 0); else ISSUE(Ld, 4);      // gate's row blocks (the slots of q, k, v: consumed)
        BSTAMP(7);
        float v[8];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          esync::unpack20x2(p[r].x, p[r].y, v[4 * r], v[4 * r + 1]);
          esync::unpack20x2(p[r].z, p[r].w, v[4 * r + 2], v[4 * r + 3]);
        }
        {
          // the norm bound: |H_4096 x|_inf <= 64 |x|_2, and sum y^2 = 128 sum x^2 (the heads' transforms are orthogonal x sqrt 128)
#pragma clang fp contract(off)
          float n0 = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) n0 = __builtin_fmaf(v[r], v[r], n0);
          n0 = had::wave_reduce_to_lane63<false>(n0);
          if (lane == 63) red[wave] = n0;
        }
        hadw::reg_stages<8, 4>(v);                       // head index bit 0 (registers 4 r + e)
        hadw::lane_stages<8, 2, 6>(v, lane);             // head index bits 1 .. 4 (lane bits 2 .. 5)
        if constexpr (G8) ISSUE8_GU(Ld, 1); else ISSUE(Ld, 5);
        const float sco = Ld.sc[3];
        had::wg_barrier<true>();                         // the sums; and the heads' attention scratch in the area has been read
        const int sh = norm_shift(red_sum8(0) * (1.f / 128.f), sco);
        if constexpr (G8) ISSUE8_GU(Ld, 2); else ISSUE(Ld, 6);
        {
#pragma clang fp contract(off)
          const float s2 = had::fmul(sco, as_f32((uint32_t)(sh + 127) << 23));
          constexpr int kPS = NIB ? B::PSH : 4096;
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            // X[head 2 hp + r][4 jq .. + 3]: natural dword 32 head + jq of a plane
            const float vv[4] = {v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]};
            uint32_t dh, dm, dl;
            hadw::digit_words_magic(vv, s2, dh, dm, dl);
            const int kd = 32 * (2 * hp + r) + jq;
            const uint32_t off = NIB ? (uint32_t)((kd & 1) * B::HOH + 4 * (kd >> 1)) : (uint32_t)(4 * kd);
            *reinterpret_cast<uint32_t*>(smem + B::kArea + off) = dh;
            *reinterpret_cast<uint32_t*>(smem + B::kArea + kPS + off) = dm;
            *reinterpret_cast<uint32_t*>(smem + B::kArea + 2 * kPS + off) = dl;
          }
        }
        if (tid == 0) shs[3] = sh;
        had::wg_barrier<true>();
      } else {
        // o_proj's input side: x = H (a (.) SU_o) * sc  (no norm), every workgroup
        u32x4 psu = *reinterpret_cast<const u32x4*>(Ld.su[3] + 8 * tid);
        float v[1][8];
        {
          // the attention output is ~8 us away for the workgroups that do not compute it: they wait for it on ONE granule (of a
          // head of their own choice per wave) instead of re-reading the whole 16 KB vector per retry next to the heads' stores
          uint32_t sp = 0;
          u32x2_t f;
          const uint64_t* g1 = zbufs + (size_t)3 * 2048 + (size_t)((8 * w + wave) & (NH - 1)) * 64;
          for (;;) {
            esync::ld8(f, g1);
            esync::drain();
            esync::own(f);
            if (esync::spin_step(f.y == (ebase | hop), sp, ctl + 1, 0x6100u + (uint32_t)w)) break;
          }
        }
        gather(std::integral_constant<int, 1>{}, SLOTS(RVQ ? M_O : 0u), 3, ebase | hop, 0x6000u, v);
        // (SU_o has landed -- the gather drained the queue: taken over HERE, or the compiler's wait for it, placed at its first
        //  use, also waits for the request just below: a memory latency on the critical path)
        asm volatile("" : "+v"(psu));
        // gate's row blocks (the slots of q, k, v: consumed) once the hand-off is through, one at a time between the stages
        // of o's input side: they have o's product, a hand-off and an edge to land
        if constexpr (RVQ) ISSUE_RVQ_GATE_G(Ld, 0); else if constexpr (G8) ISSUE8_GU(Ld, 0); else ISSUE(Ld, 4);
        BSTAMP(7);
#if QUIP_INO_EXACT      /* A/B (tools/dbg): rounds 3-4's o input side -- the exact maximum behind the transform */
        had::mul8(v[0], make_uint4(psu.x, psu.y, psu.z, psu.w));
        had8::fht4096<1, true>(v, xbuf, tid);
        if constexpr (RVQ) ISSUE_RVQ_GATE_G(Ld, 1); else if constexpr (G8) ISSUE8_GU(Ld, 1); else ISSUE(Ld, 5);
        const float sco = Ld.sc[3];
        const float mx = had8::max4096<true>(had8::absmax8(v[0], sco), red, tid);
        if constexpr (RVQ) ISSUE_RVQ_GATE_G(Ld, 2); else if constexpr (G8) ISSUE8_GU(Ld, 2); else ISSUE(Ld, 6);
        const int sh = had::shift_for(mx * ((RVQ && !HI) ? fmaxf(1.f, fabsf(a.resid_scale)) : 1.f));
#else
        had::mul8(v[0], make_uint4(psu.x, psu.y, psu.z, psu.w));
        {
          // the planes' block exponent from the norm bound (see edge()): the sum of squares of the transform's INPUT, per-wave
          // partial sums into LDS here, read behind the transform's barriers
#pragma clang fp contract(off)
          float n0 = 0.f;
#pragma unroll
          for (int r = 0; r < 8; ++r) n0 = __builtin_fmaf(v[0][r], v[0][r], n0);
          n0 = had::wave_reduce_to_lane63<false>(n0);
          if ((tid & 63) == 63) red[wave] = n0;
        }
        had8::fht4096<1, true>(v, xbuf, tid);
        if constexpr (RVQ) ISSUE_RVQ_GATE_G(Ld, 1); else if constexpr (G8) ISSUE8_GU(Ld, 1); else ISSUE(Ld, 5);
        const float sco = Ld.sc[3];
        had::wg_barrier<true>();                         // the transform's last reads of the exchange buffer: the planes land on it
        const int sh = norm_shift(red_sum8(0), sco);
        if constexpr (RVQ) ISSUE_RVQ_GATE_G(Ld, 2); else if constexpr (G8) ISSUE8_GU(Ld, 2); else ISSUE(Ld, 6);
#endif
        if constexpr (HI) had8::planes_scatter_hi(v[0], sco, sh, reinterpret_cast<uint8_t*>(smem + B::kArea), tid);
        else if constexpr (RVQ) had8::planes_scatter_rvq(v[0], sco, a.resid_scale, sh, reinterpret_cast<uint8_t*>(smem + B::kArea), tid);
        else if constexpr (NIB) {
          // planes_scatter's digits (element tid + 512 k of the strided layout) at their half-plane bytes
#pragma clang fp contract(off)
          const float s2 = had::fmul(sco, as_f32((uint32_t)(sh + 127) << 23));
          uint8_t* pb = reinterpret_cast<uint8_t*>(smem + B::kArea) + nib_half_of(tid) * B::HOH + nib_byte_of(tid);
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int X = (int)__builtin_rintf(v[0][k] * s2);
            const int X1 = (X + 128) >> 8;
            const int H = (X1 + 128) >> 8;
            uint8_t* p = pb + 256 * k;
            p[0] = (uint8_t)H;
            p[B::PSH] = (uint8_t)X1;
            p[2 * B::PSH] = (uint8_t)X;
          }
        }
        else had8::planes_scatter(v[0], sco, sh, reinterpret_cast<uint8_t*>(smem + B::kArea), tid);
        if (tid == 0) shs[3] = sh;
        had::wg_barrier<true>();
      }
    }
    BSTAMP(8);
    if constexpr (RVQ) {
      run_item(6, xlane, 48);
      run_item(7, xlane + 4096u, 48);
    } else {
      if constexpr (NIB) { Frag Fo; frags(xlane, Fo); add_rows(mul_pre(Bon, Fo), 48); }
      else add_rows(item_multiply(Bo, xlane), 48);
    }
    had::wg_barrier<true>();
    ++hop;                                             // hand-off: z_o
    publish16(4, w * 8, 48, shs[3], ebase | hop);
    had::wg_barrier<true>();
    zero_acc(48, 16);
    BSTAMP(9);

    // ================= P3: o's output side + residual, RMSNorm, input transforms of gate / up; their products =========
    rederive();
#if QUIP_PREDECODE_GATE
    // the first gate / up item's table look-ups inside the wait for z_o (its codes were requested under o's input side)
    uint32_t Bg[kPreG][kPreW];                          // (scalar registers while they wait: see down's)
    if constexpr (!RVQ) {
      esync::drain();
      own_slots(SLOTS(M_GATE));
#pragma unroll
      for (int i = 0; i < kPreG; ++i) decode_pre(i, Bg[i]);
    }
#endif
    // (ONE consumer per workgroup: its half of the machine multiplies gate, the other half up)
    edge(std::true_type{}, std::integral_constant<int, 2>{}, SLOTS(M_GATE), 4, ebase | hop, 0x7000u, Ld.sv[3], Ld.ln[1], Ld.su[4 + mgu], Ld.su[4 + mgu], Ld.sc[4 + mgu], Ld.sc[4 + mgu], false,
         // up's row blocks (RVQ: their first virtual slice) behind the hand-off, one at a time between the edge's stages
         [&]() { BSTAMP(10); if constexpr (RVQ) ISSUE_RVQ_UP_A_G(Ld, 0); else if constexpr (G8) { ISSUE8_GU(Ld, 3); ISSUE8_GU(Ld, 4); } else ISSUE(Ld, 7); },
         [&]() { if constexpr (RVQ) ISSUE_RVQ_UP_A_G(Ld, 1); else if constexpr (G8) ISSUE8_GU(Ld, 5); else ISSUE(Ld, 8); },
         [&]() { if constexpr (RVQ) ISSUE_RVQ_UP_A_G(Ld, 2); else if constexpr (G8) ISSUE8_GU(Ld, 6); else ISSUE(Ld, 9); }, 18,
         std::integral_constant<int, G8 ? 4 : 2>{});
    BSTAMP(11);
    // row owners (w < NRO: rows k' = RPO w .. RPO w + RPO - 1, one per wave): SV_gate / SV_up / SU_down of their rows into the free
    // tail of the area; everybody: the image of the K x K factors, for the MLP edge
    {
      if (w < NRO) {
        for (int p = tid; p < RPO * 3 * (FL / 8); p += kThreads) {
          const int row = p / (3 * (FL / 8)), rem = p - row * (3 * (FL / 8));
          const int vsel = rem / (FL / 8), piece = rem - vsel * (FL / 8);
          const int kr = RPO * w + row < FK ? RPO * w + row : FK - 1;
          const f16* vsrc = (vsel == 0 ? Ld.sv[4] : (vsel == 1 ? Ld.sv[5] : Ld.su[6])) + (size_t)kr * FL + piece * 8;
          *reinterpret_cast<uint4*>(smem + B::kStash + ((row * 3 + vsel) * FL + piece * 8) * 2) = *reinterpret_cast<const uint4*>(vsrc);
        }
      }
      constexpr int HPIECES = B::kHadElems / 8;
      for (int i = tid; i < HPIECES; i += kThreads)
        *reinterpret_cast<uint4*>(smem + B::kHad + i * 16) = reinterpret_cast<const uint4*>(Ld.had3)[i];
    }
    esync::drain();
    own_slots(SLOTS(M_GATE | M_UP));
    static_assert(FRB == 3, "three row blocks per matrix");
    if constexpr (RVQ) {
      const uint32_t xu = xlane;                                                  // (the second column: the same planes)
      run_items3(0, xlane, xlane, xlane, 64);                                     // first column, slice wave
      ISSUE_RVQ_UP_B(Ld);                                                           // up's second half into the slots just freed:
      run_items3(3, xlane + 4096u, xlane + 4096u, xlane + 4096u, 64);             // six items of time to land
      run_items3(6, xu, xu, xu, 64 + 16 * FRB);                                     // up, slice wave
      esync::drain();
      own_slots(SLOTS(0x007u));
      run_items3(0, xu + 4096u, xu + 4096u, xu + 4096u, 64 + 16 * FRB);           // up, slice wave + 8
    } else {
      // down's items: its slots have been free since the previous block.  Requested HERE they land under these products, far
      // from any poll (a burst in front of a poll delays its first check by a memory latency: in front of the row owners'
      // rows poll it made them 2.9K clocks late for down's product, profiles/r05_block_stamps.txt).  Rounds 3-4 requested
      // them behind the rows' sweep, a third at a time: the last third was still on its way when the planes were done.
      if constexpr (G8) {
        // seven items (slots X0..X6) against the same planes; down's four (X7, X8, X0, X1) requested behind the fourth
        Frag A;
        frags(xlane, A);
#if QUIP_PREDECODE_GATE
#define PRE_G(i) ((i) < kPreG ? Bg[(i) < kPreG ? (i) : 0] : nullptr)
#else
#define PRE_G(i) nullptr
#endif
        add_rows(item_vs(std::integral_constant<int, 0>{}, PRE_G(0), A), B::AGU);
        add_rows(item_vs(std::integral_constant<int, 1>{}, PRE_G(1), A), B::AGU + 16);
        add_rows(item_vs(std::integral_constant<int, 2>{}, nullptr, A), B::AGU + 32);
        add_rows(item_vs(std::integral_constant<int, 3>{}, nullptr, A), B::AGU + 48);
        ISSUE8_DOWN(Ld, 0); ISSUE8_DOWN(Ld, 1); ISSUE8_DOWN(Ld, 2); ISSUE8_DOWN(Ld, 3);
        add_rows(item_vs(std::integral_constant<int, 4>{}, nullptr, A), B::AGU + 64);
        add_rows(item_vs(std::integral_constant<int, 5>{}, nullptr, A), B::AGU + 80);
        add_rows(item_vs(std::integral_constant<int, 6>{}, nullptr, A), B::AGU + 96);
#undef PRE_G
      } else {
      ISSUE(Ld, 10); ISSUE(Ld, 11); ISSUE(Ld, 12);
#if QUIP_PREDECODE_GATE
      run_items3_pre(std::integral_constant<int, kPreG>{}, Bg, 0, xlane, xlane, xlane, 64);
      if constexpr (kPreQkv && QUIP_QKV_ISSUE_EARLY) { ISSUE(Ln, 0); ISSUE(Ln, 1); ISSUE(Ln, 2); }
#else
      run_items3(0, xlane, xlane, xlane, 64);                                     // first column's three row blocks
#endif
      run_items3(FRB, xlane, xlane, xlane, 64 + 16 * FRB);                         // second column's
      }
    }
    had::wg_barrier<true>();
    BSTAMP(12);

    // ================= P4: the MLP edge (decode_engine.hip) and down's product ==========================================
    rederive();
    {
      float* zcol = reinterpret_cast<float*>(smem + B::kZcol);
      if (tid < 16 * NGU) {
        // accumulator row -> (column m of this workgroup's two, k): three padded row blocks per column | G8: 112 consecutive (m, k)
        const int m = G8 ? (tid >= FK ? 1 : 0) : tid / 48, k = G8 ? tid - FK * m : tid - 48 * m;
        const int* s3 = accs + (B::AGU + tid) * 4;
        const float f = __builtin_fmaf((float)s3[0], 65536.f, __builtin_fmaf((float)s3[1], 256.f, (float)s3[2]));
        zcol[m * B::KP16 + k] = (float)(f16)(f * unscale_of(shs[0], kUnsc));      // [column 2 (w & 127) + m][k]: one matrix, one exponent
      }
      had::wg_barrier<true>();
      zero_acc(B::AGU, 16 * NGU);
      ++hop;                                           // hand-off: column -> row owners
      const uint32_t tag1 = ebase | hop;
      {
        const int o = tid >> 2, part = tid & 3;
        const int m = o >> 6, kq = o & 63;
        const bool live = kq < FK;
        // (m: which of this workgroup's two columns; the factor is its matrix's: mgu)
        const f16* hs = reinterpret_cast<const f16*>(smem + B::kHad) + mgu * B::KKP + (live ? kq : 0) * FK;
        const float* zz = zcol + m * B::KP16;
        float t = 0.f;
#pragma unroll
        for (int k4 = 0; k4 < (FK + 3) / 4; ++k4) {
          const int k = 4 * k4 + part;
          if (k < FK) t = __builtin_fmaf((float)hs[k], zz[k], t);
        }
        t += __shfl_xor(t, 1, 64);
        t += __shfl_xor(t, 2, 64);
        t *= kMixScale;                                  // (G8: the 1 / sqrt 8 of H_8 inside the 56 x 56 factor; else 1)
        // row k' goes to row owner k' / RPO.  This workgroup's 2 RPO granules per owner are a cache line of its own, written
        // as 16-byte stores of the rows (k', k' + 1), k' even (the odd row's value comes over from the next quad)
        const float tn = __shfl_down(t, 4, 64);
        if (live && part == 0 && (kq & 1) == 0) {
          uint64_t* dst = inbox + (((size_t)(kq / RPO) * FL + (2 * (w & 127) + m)) * (2 * RPO) + mgu * RPO + (kq % RPO));
          if (kq + 1 < FK) esync::st_granule2(dst, as_u32(t), as_u32(tn), tag1);
          else esync::st_granule(dst, as_u32(t), tag1);
        }
      }
      ++hop;                                           // hand-off: rows -> everybody
      const uint32_t tag2 = ebase | hop;
      BSTAMP(13);
#if QUIP_PREDECODE_DOWN
      // down's table look-ups right behind the publication of the columns, i.e. inside a wait -- the row owners' for their inbox,
      // everybody else's for the rows (its codes landed under the gate / up products): the product behind the planes is then eight
      // MFMAs per item.  (Behind the owners' row work instead, the owners were 1K clocks late for down's product.)
      // (held as scalar registers: a waiting B fragment as a 128-bit tuple needs four consecutive, even-aligned registers,
      //  and the allocator spills long-lived tuples once the file is fragmented)
      uint32_t Bd[kPreD][kPreW];                                           // (how many of the three items)
      if constexpr (!RVQ) {
        esync::drain();
        own_slots(SLOTS(M_DOWN));
#pragma unroll
        for (int i = 0; i < kPreD; ++i) decode_pre(SDf(i), Bd[i]);      // (slice 16 + wave >= 22: decoded, never multiplied)
      }
#endif
      if (w < NRO) {
        // the owner's inbox = 256 columns x (2 matrices x RPO rows) granules, swept by all 512 threads (coalesced 16-byte
        // pieces); piece p = column p / RPO, matrix (p / (RPO / 2)) & 1, rows 2 (p % (RPO / 2)) and + 1
        {
          constexpr int NPC = FL * RPO / kThreads;       // pieces per thread
          u32x4_t pc[NPC];
          uint32_t spins = 0;
          for (;;) {
#pragma unroll
            for (int jj = 0; jj < NPC; ++jj) esync::ld16(pc[jj], inbox + ((size_t)w * FL * 2 * RPO + 2 * (tid + kThreads * jj)));
            esync::drain();
            bool ok = true;
#pragma unroll
            for (int jj = 0; jj < NPC; ++jj) {
              esync::own(pc[jj]);
              const int kl = 2 * ((tid + kThreads * jj) % (RPO / 2));
              ok = ok && (RPO * w + kl >= FK || pc[jj].y == tag1) && (RPO * w + kl + 1 >= FK || pc[jj].w == tag1);
            }
            if (esync::spin_step(ok, spins, ctl + 1, 0x1000u + (uint32_t)w)) break;
          }
          if (dbg_on && tid == 0) a.dbg[w * 32 + 25] = __builtin_amdgcn_s_memtime();
          uint32_t* rowbuf = reinterpret_cast<uint32_t*>(smem + B::kArea);        // [RPO rows][2 matrices][256 columns]
#pragma unroll
          for (int jj = 0; jj < NPC; ++jj) {
            const int pi = tid + kThreads * jj, col = pi / RPO, mm = (pi / (RPO / 2)) & 1, kl = 2 * (pi % (RPO / 2));
            rowbuf[(kl * 2 + mm) * FL + col] = pc[jj].x;
            rowbuf[((kl + 1) * 2 + mm) * FL + col] = pc[jj].z;
          }
          had::wg_barrier<true>();
        }
        // a row on a PAIR of waves: wave rw its gate half, wave 4 + rw its up half, 4 consecutive elements per lane (index bits
        // 0..1 in registers, 2..7 = lane bits 0..5; ascending bit order: the same additions as fht16_lanes, fht_wg512.hip.h);
        // the up half crosses to the gate wave through LDS
        const int rw = wave & (RPO - 1), mh = (wave / RPO) & 1;
        const bool act = wave < 2 * RPO;                 // (RPO = 2: waves 0..3, one per SIMD)
        const bool row_ok = RPO * w + rw < FK;
        auto fht256 = [&](float (&x)[4]) {
#pragma clang fp contract(off)
          const float a0 = x[0] + x[1], a1 = x[0] - x[1], a2 = x[2] + x[3], a3 = x[2] - x[3];
          x[0] = a0 + a2; x[2] = a0 - a2; x[1] = a1 + a3; x[3] = a1 - a3;
          auto lst = [&](auto sc) {
            constexpr int S = decltype(sc)::value;
            const float sg = ((lane >> S) & 1) ? -1.f : 1.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = __builtin_fmaf(x[r], sg, had8::lane_partner<S>(x[r], lane));
          };
          lst(std::integral_constant<int, 0>{}); lst(std::integral_constant<int, 1>{}); lst(std::integral_constant<int, 2>{});
          lst(std::integral_constant<int, 3>{}); lst(std::integral_constant<int, 4>{}); lst(std::integral_constant<int, 5>{});
        };
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const f16* vecs = reinterpret_cast<const f16*>(smem + B::kStash) + (row_ok ? rw : 0) * 3 * FL;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (act) {
          {
            const float4 f0 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(smem + B::kArea) +
                                                              ((row_ok ? rw : 0) * 2 + mh) * FL + lane * 4);
            v[0] = f0.x; v[1] = f0.y; v[2] = f0.z; v[3] = f0.w;
          }
          fht256(v);
          const uint2 svp = *reinterpret_cast<const uint2*>(vecs + mh * FL + lane * 4);
          const f16x2 s01 = as_f16x2(svp.x), s23 = as_f16x2(svp.y);
          const float svf[4] = {(float)s01.x, (float)s01.y, (float)s23.x, (float)s23.y};
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (float)had::out_elem(v[r], 1.f / 16.f, true, svf[r], false, 0.f, false, 0.f);
        }
        float* xch = reinterpret_cast<float*>(smem + B::kStage + RPO * FL * 4);     // [RPO][256]: the up halves
        if (act && mh == 1) *reinterpret_cast<float4*>(xch + rw * FL + lane * 4) = float4{o[0], o[1], o[2], o[3]};
        had::wg_barrier<true>();
        uint32_t* stage = reinterpret_cast<uint32_t*>(smem + B::kStage);
        if (act && mh == 0) {
          const float4 u4 = *reinterpret_cast<const float4*>(xch + rw * FL + lane * 4);
          const float u[4] = {u4.x, u4.y, u4.z, u4.w};
          const uint2 sup = *reinterpret_cast<const uint2*>(vecs + 2 * FL + lane * 4);
          const f16x2 s01 = as_f16x2(sup.x), s23 = as_f16x2(sup.y);
          const float suf[4] = {(float)s01.x, (float)s01.y, (float)s23.x, (float)s23.y};
          float e[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) e[r] = had::fmul(had::fmul(u[r], had::silu(o[r])), suf[r]);
          fht256(e);
          // the row, ROUNDED TO fp16 (round 5; the reference materialises fp16 here too: the transform's output feeds
          // `hadK @ x` as an fp16 tensor, quant.py:72-88), prescaled, next to the owner's other row in LDS ([j][RPO] fp16).
          // Rounds 3-4 sent fp16 hi + lo (22 bits): twice the bytes on the hand-off every workgroup sweeps.
          constexpr float kPre = 1.f / 16.f;
          uint16_t* stage16 = reinterpret_cast<uint16_t*>(stage);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            stage16[(4 * lane + r) * RPO + rw] = __builtin_bit_cast(uint16_t, (f16)(e[r] * kPre));
        }
        had::wg_barrier<true>();
        // ... and out: granule (owner w, column j) = {rows 2 w | 2 w + 1 of column j, tag}: [owner][256], two columns per store
        if (tid < FL / 2) {
          static_assert(RPO == 2, "one fp16 pair per column and owner");
          const uint2 d = *reinterpret_cast<const uint2*>(stage + 2 * tid);
          esync::st_granule2(frow + ((size_t)w * FL + 2 * tid), d.x, d.y, tag2);
        }
        had::wg_barrier<true>();                         // the staging area is free again (the gather below zeroes over it)
      }
      BSTAMP(14);
      typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
      // gather the rows: granule (owner o, column j) = {fp16 rows 2 o | 2 o + 1 of column j, tag}; a 16-byte piece = two columns
      {
        constexpr int PIECES = NRO * (FL / 2), NP = (PIECES + kThreads - 1) / kThreads;
        // LDS image [KP16 / 2 row pairs][256 columns] of fp16 pairs (rows 2 o, 2 o + 1): the sweep's pieces (one owner, two
        // columns) land as 8-byte stores on consecutive addresses (a [column][pair] image put 32 lanes on two banks)
        uint32_t* ft = reinterpret_cast<uint32_t*>(smem + B::kArea);
        for (int j = tid; j < (B::KP16 / 2 - NRO) * FL; j += kThreads) ft[NRO * FL + j] = 0u;
        if (kCheapPoll && wave == 0) {
          uint32_t spins0 = 0;
          // (eight of the last columns: 32 workgroups per polled line instead of 256 -- 256 pollers of ONE line delay the owners' stores to it)
          const uint64_t* last = frow + ((size_t)(lane < NRO ? lane : 0) * FL + (FL - 1 - (w & 7)));
          for (;;) {
            u32x2_t f;
            esync::ld8(f, last);
            esync::drain();
            esync::own(f);
            if (esync::spin_step(f.y == tag2, spins0, ctl + 1, 0x3000u + (uint32_t)w)) break;
          }
        }
        had::wg_barrier<true>();
        BSTAMP(26);
        u32x4_t p[NP];
        uint32_t spins = 0;
        for (;;) {
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            const int i = tid + kThreads * j;
            esync::ld16(p[j], frow + 2 * (size_t)(i < PIECES ? i : 0));
          }
          esync::drain();
          bool ok = true;
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            esync::own(p[j]);
            ok = ok && p[j].y == tag2 && p[j].w == tag2;
          }
          if (esync::spin_step(ok, spins, ctl + 1, 0x2000u + (uint32_t)w)) break;
        }
        if constexpr (!RVQ) own_slots(SLOTS(M_DOWN));
        BSTAMP(27);
        if constexpr (RVQ) ISSUE_RVQ_DOWN_G0(Ld);
        // the NEXT block's q, k, v rows (gate's slots: consumed): they land under the K-mix and down's product and are
        // decoded inside the wait for z_d (rounds 3-4 requested them behind that hand-off)
        if constexpr (kPreQkv && !QUIP_QKV_ISSUE_EARLY) {
          if constexpr (G8) { ISSUE8_QKV(Ln, 0); ISSUE8_QKV(Ln, 1); } else { ISSUE(Ln, 0); ISSUE(Ln, 1); ISSUE(Ln, 2); }
        }
        // (the sum of squares of the rows on the way: down's block exponent comes from the norm bound |(H^T (x) I) r|_inf <=
        //  |r|_2 -- the rows of the orthogonal factor are unit vectors (G8: of norm sqrt 8) -- known BEFORE the K-mix: no maximum over its results,
        //  no reduction behind it; 4-5 of the 22 bits idle, as on the 4096-wide edges)
        float ssr = 0.f;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          const int i = tid + kThreads * j;
          if (i < PIECES) {
            const int o = i / (FL / 2), cp = i - o * (FL / 2);
            const uint32_t msk = (2 * o + 1 < FK) ? 0xffffffffu : 0x0000ffffu;       // (row 43 does not exist)
            const uint32_t r0 = p[j].x & msk, r1 = p[j].z & msk;
            *reinterpret_cast<uint2*>(ft + o * FL + 2 * cp) = make_uint2(r0, r1);
            const f16x2 a0 = as_f16x2(r0), a1 = as_f16x2(r1);
            ssr = __builtin_fmaf((float)a0.x, (float)a0.x, ssr); ssr = __builtin_fmaf((float)a0.y, (float)a0.y, ssr);
            ssr = __builtin_fmaf((float)a1.x, (float)a1.x, ssr); ssr = __builtin_fmaf((float)a1.y, (float)a1.y, ssr);
          }
        }
        ssr = had::wave_reduce_to_lane63<false>(ssr);
        if (lane == 63) red[wave] = ssr;
      }
      had::wg_barrier<true>();
      if constexpr (RVQ) ISSUE_RVQ_DOWN_G1(Ld);
      BSTAMP(15);
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      constexpr int KCT = B::KP16 / 16;                 // 16 x 16 tiles of the K x K factor per side (3; G8: 4)
      f32x4 acc[2][KCT];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ct = 0; ct < KCT; ++ct) acc[jt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        const uint32_t* ft = reinterpret_cast<const uint32_t*>(smem + B::kArea);
        // B fragments of the K-mix: had_d^T (the fp16 rows take [0, 24 K) of the area, the factor image behind them stays), a
        // k step at a time
        const f16* hdT = reinterpret_cast<const f16*>(smem + B::kHad) + 2 * B::KKP;
#pragma unroll
        for (int s = 0; s < KCT; ++s) {
          f16x4 bfs[KCT];
#pragma unroll
          for (int ct = 0; ct < KCT; ++ct) bfs[ct] = *reinterpret_cast<const f16x4*>(hdT + (16 * ct + n) * B::KPS + 16 * s + 4 * q);
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
            const int tile = wave + jt * kWaves;
            // A[column 16 tile + n][rows 16 s + 4 q .. + 3] = the row pairs 8 s + 2 q and + 1 of that column
            const uint2 apr = make_uint2(ft[(8 * s + 2 * q) * FL + 16 * tile + n], ft[(8 * s + 2 * q + 1) * FL + 16 * tile + n]);
            const f16x4 ah = __builtin_bit_cast(f16x4, apr);
#pragma unroll
            for (int ct = 0; ct < KCT; ++ct)
              acc[jt][ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bfs[ct], acc[jt][ct], 0, 0, 0);
          }
        }
      }
      if constexpr (RVQ) ISSUE_RVQ_DOWN_G2(Ld);
      const float in_scale = Ld.sc[6] * 16.f;
      // (in_scale carries the prescale 1 / 16 of the rows back: the bound is for acc * in_scale, |acc|_inf <= |rows|_2)
      // (G8: the rows of the 56 x 56 factor R_7 (x) H_8 have norm sqrt 8, not 1 -- in_scale carries the 1 / sqrt 8 that undoes it)
      const float bound = sqrtf(red_sum8(0)) * fabsf(in_scale) * (1.0625f / kMixScale);
      const int sh_d = had::shift_for(bound * ((RVQ && !HI) ? fmaxf(1.f, fabsf(a.resid_scale)) : 1.f));
      had::wg_barrier<true>();                         // every wave's K-mix has read the rows: the planes land on them
      {
        uint8_t* pl = reinterpret_cast<uint8_t*>(smem + B::kArea);
        const float s2 = had::fmul(in_scale, as_f32((uint32_t)(sh_d + 127) << 23));
        const float s2r = had::fmul(had::fmul(in_scale, a.resid_scale), as_f32((uint32_t)(sh_d + 127) << 23));   // RVQ: the residual side
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const int tile = wave + jt * kWaves;
#pragma unroll
          for (int ct = 0; ct < KCT; ++ct) {
            const int kc = 16 * ct + n;
            // (digits straight from the fp32 magic number, four values per word: hadw::digit_words_magic)
            const float av[4] = {acc[jt][ct][0], acc[jt][ct][1], acc[jt][ct][2], acc[jt][ct][3]};
            if (kc < FK) {
              const int kk = kc * FL + 16 * tile + 4 * q;
              if constexpr (HI) {
                // four elements i0 .. i0 + 3 (i0 = 0 | 4) of one 8-group: virtual positions i0 + [0 1 . .] hold elements i0, i0 + 2,
                // positions 8 + i0 + [0 1 . .] elements i0 + 1, i0 + 3; the other two of each word are zero digits
                const int vv = 2 * (kk & ~7) + (kk & 4);
                const int offa = (vv >> 8) * 272 + (vv & 255), offb = offa + 8;
                const float ea[4] = {av[0], av[2], 0.f, 0.f}, eb[4] = {av[1], av[3], 0.f, 0.f};
                uint32_t h, m, l;
                hadw::digit_words_magic(ea, s2, h, m, l);
                *reinterpret_cast<uint32_t*>(pl + offa) = h;
                *reinterpret_cast<uint32_t*>(pl + B::kPlaneD + offa) = m;
                *reinterpret_cast<uint32_t*>(pl + 2 * B::kPlaneD + offa) = l;
                hadw::digit_words_magic(eb, s2, h, m, l);
                *reinterpret_cast<uint32_t*>(pl + offb) = h;
                *reinterpret_cast<uint32_t*>(pl + B::kPlaneD + offb) = m;
                *reinterpret_cast<uint32_t*>(pl + 2 * B::kPlaneD + offb) = l;
              } else if constexpr (RVQ) {
                // four elements of one 8-group: residual-side digits at 2 (kk & ~7) + (kk & 7), main-side 8 further
                const int vv = 2 * (kk & ~7) + (kk & 7);
                const int offr = (vv >> 8) * 272 + (vv & 255), offm = ((vv + 8) >> 8) * 272 + ((vv + 8) & 255);
                uint32_t h, m, l;
                hadw::digit_words_magic(av, s2r, h, m, l);
                *reinterpret_cast<uint32_t*>(pl + offr) = h;
                *reinterpret_cast<uint32_t*>(pl + B::kPlaneD + offr) = m;
                *reinterpret_cast<uint32_t*>(pl + 2 * B::kPlaneD + offr) = l;
                hadw::digit_words_magic(av, s2, h, m, l);
                *reinterpret_cast<uint32_t*>(pl + offm) = h;
                *reinterpret_cast<uint32_t*>(pl + B::kPlaneD + offm) = m;
                *reinterpret_cast<uint32_t*>(pl + 2 * B::kPlaneD + offm) = l;
              } else {
                // (nibble mode: natural dword kk / 4 of a plane = dword (that >> 1) of its "lo" (even) / "hi" (odd) half)
                const int off = NIB ? ((kk >> 2) & 1) * B::HOD + (kk >> 8) * 144 + (((kk & 255) >> 3) << 2) : (kk >> 8) * 272 + (kk & 255);
                uint32_t h, m, l;
                hadw::digit_words_magic(av, s2, h, m, l);
                *reinterpret_cast<uint32_t*>(pl + off) = h;
                *reinterpret_cast<uint32_t*>(pl + B::kPlaneD + off) = m;
                *reinterpret_cast<uint32_t*>(pl + 2 * B::kPlaneD + off) = l;
              }
            }
          }
        }
        for (int i = VM * NFFN + 4 * tid; i < B::KPDV; i += 4 * kThreads) {
          const int off = NIB ? ((i >> 2) & 1) * B::HOD + (i >> 8) * 144 + (((i & 255) >> 3) << 2) : (i >> 8) * 272 + (i & 255);
#pragma unroll
          for (int d = 0; d < 3; ++d) *reinterpret_cast<uint32_t*>(pl + d * B::kPlaneD + off) = 0u;
        }
      }
      had::wg_barrier<true>();
      BSTAMP(16);
      esync::drain();                                  // down's codes (and, behind them, the next block's q, k, v)
      own_slots(SLOTS(M_DOWN | (kPreQkv ? M_QKV : 0u)));
      const uint32_t xlane_d = NIB ? (uint32_t)B::kArea + nib_row_offset(n, B::kPlaneD, B::HOD) + (uint32_t)q * 32u
                                   : (uint32_t)B::kArea + (uint32_t)min(n, 2) * (uint32_t)B::kPlaneD + (uint32_t)q * 64u;
      constexpr uint32_t kSliceD = NIB ? 288u : 544u;          // bytes between the A fragments of consecutive K slices of down's planes
      if constexpr (RVQ) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int sl = i * kWaves + wave;
          if (sl < B::JDV) {
            ItemAddr ad;
            item_addresses<REP>(qa[i], qb[i], lane_c, lane_c2, ad, lane_c3);
            add_rows(T::kD4 ? item_mfma_d4<272>(ad, xlane_d + (uint32_t)(sl * 544)) : item_mfma<272, R3>(ad, xlane_d + (uint32_t)(sl * 544)), B::AD);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < ND; ++i) {
          const int sl = i * kWaves + wave;
          if (sl < JD) {
#if QUIP_PREDECODE_DOWN
            if (i < kPreD) {
              const uint32_t* bs = Bd[i < kPreD ? i : 0];
              if constexpr (NIB) {
                Frag Fd;
                frags_d(xlane_d + (uint32_t)sl * kSliceD, Fd);
                add_rows(mul_pre(bs, Fd), B::AD);
              } else {
                i32x4 Bt[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) Bt[t] = i32x4{(int)bs[4 * t], (int)bs[4 * t + 1], (int)bs[4 * t + 2], (int)bs[4 * t + 3]};
                add_rows(item_multiply<272>(Bt, xlane_d + (uint32_t)(sl * 544)), B::AD);
              }
              continue;
            }
#endif
            if constexpr (NIB) {
              Frag Fd;
              frags_d(xlane_d + (uint32_t)sl * kSliceD, Fd);
              add_rows(mul_slot(qa[SDf(i)], qb[SDf(i)], Fd), B::AD);
            } else {
              ItemAddr ad;
              item_addresses<REP>(qa[SDf(i)], qb[SDf(i)], lane_c, lane_c2, ad, lane_c3);
              add_rows(T::kD4 ? item_mfma_d4<272>(ad, xlane_d + (uint32_t)(sl * 544)) : item_mfma<272, R3>(ad, xlane_d + (uint32_t)(sl * 544)), B::AD);
            }
          }
        }
      }
      had::wg_barrier<true>();
      ++hop;                                           // hand-off: z_d
      publish16(5, w * 8, B::AD, sh_d, ebase | hop);
      had::wg_barrier<true>();
      zero_acc(B::AD, 16);
      BSTAMP(17);
    }
    sv_d_prev = Ld.sv[6];
    const bool more = l + 1 < a.n_layers;
    if (more) {
      had::wg_barrier<true>();                         // everybody has read this block's descriptor for the last time
      if (tid < 64) reinterpret_cast<uint32_t*>(smem + B::kDesc)[tid] = reinterpret_cast<const uint32_t*>(a.layers + l + 1)[tid];
      had::wg_barrier<true>();
    }
    // output side of this block's down_proj + residual -> h, RMSNorm and the input transforms of the NEXT block's q, k, v in ONE
    // edge; their row blocks go out behind the hand-off.
    // (requested unconditionally -- behind the last block: its own q, k, v rows once more, never multiplied, and the input
    //  side of its own q, k, v once more, never used -- so that no request depends on a branch: the compiler makes several
    //  conditional regions of one `if`, and a register a load is still going to write must not meet a copy at their joins)
    rederive();
    uint32_t Bq[NPQ][kPreW];
    if constexpr (kPreQkv) {
      // the first items' table look-ups inside the wait for z_d (their codes landed under down's product); held as scalars
#pragma unroll
      for (int i = 0; i < NPQ; ++i) decode_pre(SQf(i), Bq[i]);
    }
    {
      const int c_lo = qc_lo, c_hi = qc_hi;
      edge(std::true_type{}, std::integral_constant<int, 2>{}, SLOTS(0u), 5, ebase | hop, 0x4000u, sv_d_prev, Ld.ln[0], Ld.su[c_lo], Ld.su[c_hi],
           Ld.sc[c_lo], Ld.sc[c_hi], c_hi != c_lo,
           [&]() { BSTAMP(1); if constexpr (RVQ) { ISSUE_RVQ_QKV_G(Ld, 0); ISSUE_RVQ_QKV_G(Ld, 1); } else if constexpr (!kPreQkv) { ISSUE(Ld, 0); ISSUE(Ld, 1); } },
           [&]() { if constexpr (RVQ) ISSUE_RVQ_QKV_G(Ld, 2); else if constexpr (!kPreQkv) ISSUE(Ld, 2); }, [&]() {}, 23,
           std::integral_constant<int, RVQ ? 8 : (kPreQkv ? 0 : 4)>{});
      BSTAMP(2);
    }
    if (more) P1_products(std::integral_constant<bool, kPreQkv>{}, Bq);
    esync::drain();
    own_slots(SLOTS(M_QKV));
  }
  // ---- h_out -----------------------------------------------------------------------------------------------------------
  if (w == 0) {
    // A launch in which a wait gave up (ctl[1] != 0) has no result: h_out is all NaN then -- whoever consumes it sees that
    // without reading the workspace -- and ctl[2] keeps position + 1 of the FIRST such launch (the host replays from there)
    uint32_t e, fp;
    esync::ld4(e, ctl + 1);
    esync::ld4(fp, ctl + 2);
    esync::drain();
    esync::own(e);
    esync::own(fp);
    const bool failed = __builtin_amdgcn_readfirstlane((int)e) != 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
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
#undef ISSUE
#undef SLOTS
#undef OFFC
#undef ISSUE_RVQ_QKV
#undef ISSUE_RVQ_O
#undef ISSUE_RVQ_GATE
#undef ISSUE_RVQ_UP_A
#undef ISSUE_RVQ_UP_B
#undef ISSUE_RVQ_GATE_G
#undef ISSUE_RVQ_DOWN_G0
#undef ISSUE_RVQ_QKV_G
#undef ISSUE_RVQ_DOWN_G1
#undef ISSUE_RVQ_DOWN_G2
#undef ISSUE_RVQ_UP_A_G
#undef ISSUE_RVQ_DOWN
}

}  // namespace

#if QUIP_BLOCK_G8
size_t block_engine_g8_workspace_bytes() { return kWsBytes; }

// Llama-3-8B / Mistral-7B: hidden 4096, 32 heads of 128 on 8 KV heads, n_ffn = 14336 with the reference's K = 7 factor (quant.py:26-39)
bool block_engine_g8_supported(int hidden, int heads, int kv_heads, int head_dim, int n_ffn, int K) {
  return hidden == HID && heads == NH && kv_heads == NKVH && head_dim == HD && n_ffn == NFFN && K == 7 &&
         device_cu_count_strict() >= NWG;
}

int block_engine_g8_launch(const BlockEngineArgs& in, hipStream_t stream) {
  if (in.n_layers < 1 || in.n_layers > 146) return QUIP_ERR_BAD_SHAPE;     // up to 7 hand-offs per block, 10-bit counter
  if (in.codebook != 0) return QUIP_ERR_UNSUPPORTED;
  BlockArgs a;
  a.layers = reinterpret_cast<const BlockLayer*>(in.layers);
  a.h_in = reinterpret_cast<const f16*>(in.h_in);
  a.h_out = reinterpret_cast<f16*>(in.h_out);
  a.pos = reinterpret_cast<const int64_t*>(in.pos);
  a.cos = in.cos; a.sin = in.sin;
  a.grid = reinterpret_cast<const uint64_t*>(in.grid);
  a.ws = reinterpret_cast<char*>(in.workspace);
  a.dbg = reinterpret_cast<uint64_t*>(in.dbg);
  a.n_layers = in.n_layers; a.max_len = in.max_len; a.dbg_layer = in.dbg_layer;
  a.rms_eps = in.rms_eps; a.attn_scale = in.attn_scale; a.resid_scale = 0.f;
  a.grid2 = nullptr;
  // QUIP_ENG_REP=24: the byte tables of round 5 (32 / 16 copies) instead of the nibble mode, for A/B
  static const bool rep24 = getenv("QUIP_ENG_REP") && atoi(getenv("QUIP_ENG_REP")) == 24;
  auto go = [&](auto kern, int lds, DynLdsCache& configured, ResidencyCache& resident) -> int {
    if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
    if (!persistent_grid_fits(resident, reinterpret_cast<const void*>(kern), kThreads, lds, NWG)) return QUIP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(NWG), dim3(kThreads), lds, stream, a);
    return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
  };
  static DynLdsCache c24, c4;
  static ResidencyCache r24, r4;
  if (rep24) return go(decode_block_kernel<24>, BLds<24>::kBytes, c24, r24);
  return go(decode_block_kernel<4>, BLds<4>::kBytes, c4, r4);
}
#else
size_t block_engine_workspace_bytes() { return kWsBytes; }
size_t block_engine_layer_bytes() { return sizeof(BlockLayer); }

bool block_engine_supported(int hidden, int heads, int kv_heads, int head_dim, int n_ffn, int K) {
  return hidden == HID && heads == NH && kv_heads == NH && head_dim == HD && n_ffn == NFFN && K == FK &&
         device_cu_count_strict() >= NWG;
}

int block_engine_launch(const BlockEngineArgs& in, hipStream_t stream) {
  if (in.n_layers < 1 || in.n_layers > 146) return QUIP_ERR_BAD_SHAPE;     // up to 7 hand-offs per block, 10-bit counter
  BlockArgs a;
  a.layers = reinterpret_cast<const BlockLayer*>(in.layers);
  a.h_in = reinterpret_cast<const f16*>(in.h_in);
  a.h_out = reinterpret_cast<f16*>(in.h_out);
  a.pos = reinterpret_cast<const int64_t*>(in.pos);
  a.cos = in.cos; a.sin = in.sin;
  a.grid = reinterpret_cast<const uint64_t*>(in.grid);
  a.ws = reinterpret_cast<char*>(in.workspace);
  a.dbg = reinterpret_cast<uint64_t*>(in.dbg);
  a.n_layers = in.n_layers; a.max_len = in.max_len; a.dbg_layer = in.dbg_layer;
  a.rms_eps = in.rms_eps; a.attn_scale = in.attn_scale; a.resid_scale = 0.f;
  a.grid2 = in.grid2;
  // codebook 0: E8P12 (32 copies of the abs table, 16 of the sign table), 1: D4 (one table of 256 x 4 bytes, a private copy per lane)
  auto go = [&](auto kern, int lds, DynLdsCache& configured, ResidencyCache& resident) -> int {
    if (ensure_dyn_lds(configured, reinterpret_cast<const void*>(kern), lds) != QUIP_OK) return QUIP_ERR_LAUNCH;
    if (!persistent_grid_fits(resident, reinterpret_cast<const void*>(kern), kThreads, lds, NWG)) return QUIP_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(kern, dim3(NWG), dim3(kThreads), lds, stream, a);
    return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
  };
  static DynLdsCache c16, c64;
  static DynLdsCache crvq, chi, crvq3;
  static ResidencyCache r16, r24, r64, rrvq, rhi, rrvq3;
  if (in.codebook == 4) {
    if (!in.grid2) return QUIP_ERR_NULL_POINTER;
    a.resid_scale = in.resid_scale;
    return go(decode_block_kernel<12, true>, BLds<12, true>::kBytes, crvq3, rrvq3);
  }
  if (in.codebook == 3) return go(decode_block_kernel<64, true, true>, BLds<64, true>::kBytes, chi, rhi);
  if (in.codebook == 2) { a.resid_scale = in.resid_scale; return go(decode_block_kernel<16, true>, BLds<16, true>::kBytes, crvq, rrvq); }
  if (in.codebook == 1) return go(decode_block_kernel<64>, BLds<64>::kBytes, c64, r64);
  if (in.codebook != 0) return QUIP_ERR_UNSUPPORTED;
  // A/B: QUIP_ENG_REP=16: byte tables, two-way conflicts on both; 24: byte tables, 32 / 16 copies (round 5); default: nibble mode
  static const int eng_rep = getenv("QUIP_ENG_REP") ? atoi(getenv("QUIP_ENG_REP")) : 4;
  static DynLdsCache c24, c4;
  static ResidencyCache r4;
  if (eng_rep == 16) return go(decode_block_kernel<16>, BLds<16>::kBytes, c16, r16);
  if (eng_rep == 24) return go(decode_block_kernel<24>, BLds<24>::kBytes, c24, r24);
  return go(decode_block_kernel<4>, BLds<4>::kBytes, c4, r4);
}

#endif

}  // namespace quip
