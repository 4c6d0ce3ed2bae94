// bs=1 decode glue between the QuantLinear calls of one transformer block: rotary embedding of
// q / k, KV-cache append and single-query attention over the static cache, in one launch.
// This is the part of the reference's metric driver (example_generate.py:9-59: HF StaticCache +
// LlamaAttention under torch.compile) that sits between q/k/v_proj and o_proj; it is not part of
// quip_cuda.  Fused here because on MI355X every dependent launch of a decode step costs
// microseconds while the math is nothing.
//
// One workgroup per query head, 256 threads.  A key is handled by HD/8 lanes (16 bytes of the
// fp16 head vector per lane); the 256/(HD/8) lane groups stride over the cached positions with
// their own online-softmax state and are merged through LDS at the end (flash-decoding inside
// a workgroup).  The new k / v row is used from registers (and appended to the cache by the
// first query head of its KV group), so no other workgroup has to see the cache write.
// Contexts of a few thousand tokens are fine; beyond that a split over workgroups would pay.
//
// ZIN variant (round 2): the launch takes the RAW GEMV outputs of q / k / v_proj and runs their output-side
// Hadamard transform (SV (.) H z / sqrt(n), qlinear.py:106-114 with K = 1) in its own prologue -- three groups of 256
// threads transform q, k and v side by side with the functions of had_device.hip.h (the same bits as the separate
// transform launch it replaces), the 8 threads that own this head's 128 values leave them in LDS, and the
// attention proper continues on the first 256 threads.  Every head repeats the three 4096-point transforms
// (~0.5 us on otherwise idle CUs) instead of one more dependent launch per block (~5 us).
#include "had_device.hip.h"
#include "quip_device.hip.h"
#include "quip_internal.h"

namespace quip {
namespace {

struct AttnZ {
  const f16* z[3];      // raw GEMV outputs of q / k / v_proj, [n] each
  const f16* post[3];   // SV of the three modules, [n]
  float scale[3];       // 1 / sqrt(width)
  int n, logL;          // multi-head attention: common width (heads * HD == kv_heads * HD == n), a power of two <= 4096
                        // grouped queries (LQ / LKV instantiations): the width of q; k and v are 2^LKV wide
};

struct AttnArgs {
  const f16* q;        // [heads, HD]
  const f16* k;        // [kv_heads, HD]  (pre-rope)
  const f16* v;        // [kv_heads, HD]
  const float* cos;    // [max_len, HD]
  const float* sin;    // [max_len, HD]
  const int64_t* pos;  // device scalar: index of the current token
  f16* kcache;         // [kv_heads, max_len, HD]
  f16* vcache;
  f16* out;            // [heads, HD]
  int heads, kv_heads, max_len;
  float scale;
  float* ws;           // split mode: partial (acc[HD], m, l) per (head, split); null: one workgroup per head
  unsigned* counters;  // split mode: arrivals per head (zero between launches)
  int window;          // sliding-window attention (Mistral: config.sliding_window): positions (pos - window, pos]; <= 0: [0, pos]
};

constexpr int kSplits = 8;  // workgroups per head for long contexts (grid.y)
constexpr int kSplitFromPos = 256;    // shorter contexts: split 0 does everything, no workspace traffic

__device__ __forceinline__ void unpack8h(const uint4& u, float o[8]) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2 h = as_f16x2(w[i]);
    o[2 * i] = (float)h.x;
    o[2 * i + 1] = (float)h.y;
  }
}

// rotary embedding (HF half-rotation) of the 8 dims [d0, d0 + 8) of one head vector; result
// rounded to fp16 like the eager graph does
template <int HD>
__device__ __forceinline__ void rope8(const f16* vec, const float c8[8], const float s8[8], int d0, float o[8]) {
  float a[8], b[8];
  unpack8h(*reinterpret_cast<const uint4*>(vec + d0), a);
  const int dp = d0 < HD / 2 ? d0 + HD / 2 : d0 - HD / 2;
  unpack8h(*reinterpret_cast<const uint4*>(vec + dp), b);
  const float sgn = d0 < HD / 2 ? -1.f : 1.f;
#pragma unroll
  // x * cos + rot * sin with every operation rounded on its own, as the eager graph does (and so that the two
  // instantiations of the kernel cannot contract the expression differently: an fma in one of them moved single
  // cache elements by one fp16 ulp)
  // (had::fmul / fadd: compiled with contraction switched off; __fmul_rn and friends are plain operators to hipcc)
  for (int i = 0; i < 8; ++i) o[i] = (float)(f16)had::fadd(had::fmul(a[i], c8[i]), had::fmul(sgn * b[i], s8[i]));
}

// barriers inside fht16_fixed<LOGL, false> (had_device.hip.h): two around the lane stages, two per pass through LDS
__host__ __device__ constexpr int fht16_barriers(int logl) { return logl <= 8 ? 2 : 2 + 2 * ((logl + 3) / 4 - 2); }

// ZIN with LQ == 0: q, k, v of one common width on 3 x 256 threads (run-time length).  LQ > 0 (grouped queries): q of
// width 2^LQ on the first 2^LQ / 16 threads, k and v of width 2^LKV on the next 2 x 2^LKV / 16 (whole waves each).
template <int HD, bool ZIN, int LQ = 0, int LKV = 0>
__global__ __launch_bounds__(!ZIN ? 256 : (LQ == 0 ? 768 : (1 << LQ) / 16 + 2 * ((1 << LKV) / 16)))
void rope_attn_decode_kernel(AttnArgs a, AttnZ zz) {
  constexpr int LPK = HD / 8;        // lanes per key
  constexpr int NG = 256 / LPK;      // key groups per workgroup
  constexpr int U = 4;               // keys in flight per group
  __shared__ float s_m[NG], s_l[NG];
  __shared__ float s_acc[NG][HD + 4];
  const int tid = threadIdx.x, h = blockIdx.x;
  const int gl = tid % LPK, grp = tid / LPK, d0 = gl * 8;
  const int group = a.heads / a.kv_heads, kvh = h / group;
  // ZIN: the transform inputs do not depend on the position: requested before it is read
  constexpr bool GQ = LQ > 0;
  constexpr int TQ = GQ ? (1 << LQ) / 16 : 256, TKV = GQ ? (1 << LKV) / 16 : 256;
  static_assert(!GQ || (TQ >= 256 && TKV % 64 == 0), "the attention runs on the first 256 threads; k / v on whole waves");
  const int g3 = GQ ? (tid < TQ ? 0 : (tid < TQ + TKV ? 1 : 2)) : tid >> 8;
  const int tt = GQ ? tid - (g3 == 0 ? 0 : (g3 == 1 ? TQ : TQ + TKV)) : tid & 255, e0 = tt * 16;
  const bool zact = ZIN && (GQ || e0 < zz.n);
  uint4 zraw[2] = {}, praw[2] = {};
  if constexpr (ZIN) {
    if (zact) {
      zraw[0] = *reinterpret_cast<const uint4*>(zz.z[g3] + e0);
      zraw[1] = *reinterpret_cast<const uint4*>(zz.z[g3] + e0 + 8);
      praw[0] = *reinterpret_cast<const uint4*>(zz.post[g3] + e0);
      praw[1] = *reinterpret_cast<const uint4*>(zz.post[g3] + e0 + 8);
    }
  }
  const long long pos64 = *a.pos;
  // A position outside the cache (one step past max_len, a corrupted counter) must not index cos / sin or the
  // cache: nothing is appended, the head's output becomes NaN (visible downstream) and every workgroup of the
  // launch leaves before touching the split workspace.
  if (pos64 < 0 || pos64 >= (long long)a.max_len) {
    if (blockIdx.y == 0 && tid < HD) a.out[(size_t)h * HD + tid] = __builtin_bit_cast(f16, (unsigned short)0x7e00);
    return;
  }
  const int pos = (int)pos64;
  // split mode (long context, workspace given): workgroup (h, s) takes positions [t_lo, t_hi); the last
  // workgroup of a head to arrive merges the partial softmax states (flash-decoding across CUs)
  const bool split = a.ws != nullptr && pos >= kSplitFromPos;
  const int sidx = blockIdx.y;
  if (!split && sidx > 0) return;
  // sliding window: the walk starts at `first` instead of 0 (the cache stays linear: rows [0, pos] all exist)
  const int first = a.window > 0 ? max(0, pos + 1 - a.window) : 0;
  const int chunk = split ? (pos - first + kSplits) / kSplits : pos + 1 - first;      // ceil((pos + 1 - first) / kSplits)
  const int t_lo = first + (split ? sidx * chunk : 0);
  const int t_hi = split ? min(pos + 1, t_lo + chunk) : pos + 1;       // exclusive (a split past the end walks nothing: m = -inf)
  const float* cs = a.cos + (size_t)pos * HD;
  const float* sn = a.sin + (size_t)pos * HD;

  const f16* qh = a.q + (size_t)h * HD;
  const f16* kh = a.k + (size_t)kvh * HD;
  const f16* vh = a.v + (size_t)kvh * HD;
  // rotary factors of this thread's 8 dims (requested before the transforms: one memory round trip less)
  float c8[8], s8[8];
  if (tid < 256) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { c8[i] = cs[d0 + i]; s8[i] = sn[d0 + i]; }
  }
  if constexpr (ZIN) {
    extern __shared__ __attribute__((aligned(16))) float zbuf[];   // 3 x had::buf_floats(n)
    __shared__ __attribute__((aligned(16))) f16 s_qkv[3][HD];
    const bool act = zact;
    float v[16], tp[16];
    had::unpack8(zraw[0], v);
    had::unpack8(zraw[1], v + 8);
    had::unpack8(praw[0], tp);
    had::unpack8(praw[1], tp + 8);
    if constexpr (GQ) {
      // different lengths side by side: the shorter transform passes the barriers the longer one still has to meet
      float* zb = zbuf + (g3 == 0 ? 0 : had::buf_floats(1 << LQ) + (g3 - 1) * had::buf_floats(1 << LKV));
      if (g3 == 0) {
        had::fht16_fixed<LQ, false>(v, zb, 0, tt, true);
      } else {
        had::fht16_fixed<LKV, false>(v, zb, 0, tt, true);
#pragma unroll
        for (int i = 0; i < fht16_barriers(LQ) - fht16_barriers(LKV); ++i) __syncthreads();
      }
    } else {
      had::fht16(v, zbuf + g3 * had::buf_floats(zz.n), tt, zz.logL, act, 0);
    }
    const int base = (g3 == 0 ? h : kvh) * HD;
    if (act && e0 >= base && e0 < base + HD) {
      f16 o[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) o[r] = had::out_elem(v[r], zz.scale[g3], true, tp[r], false, 0.f, false, 0.f);
      uint4* dst = reinterpret_cast<uint4*>(&s_qkv[g3][e0 - base]);
      dst[0] = *reinterpret_cast<uint4*>(&o[0]);
      dst[1] = *reinterpret_cast<uint4*>(&o[8]);
    }
    __syncthreads();
    if (tid >= 256) return;
    qh = s_qkv[0];
    kh = s_qkv[1];
    vh = s_qkv[2];
  }
  float q8[8], kn[8], vn[8];
  rope8<HD>(qh, c8, s8, d0, q8);
  rope8<HD>(kh, c8, s8, d0, kn);
  const uint4 vraw = *reinterpret_cast<const uint4*>(vh + d0);
  unpack8h(vraw, vn);
#pragma unroll
  for (int i = 0; i < 8; ++i) q8[i] *= a.scale;
  f16* kc = a.kcache + (size_t)kvh * a.max_len * HD;
  f16* vc = a.vcache + (size_t)kvh * a.max_len * HD;
  if (h % group == 0 && grp == 0 && sidx == 0) {   // append the new row (StaticCache.update)
    uint4 kr;
    kr.x = pack_f16(kn[0], kn[1]); kr.y = pack_f16(kn[2], kn[3]);
    kr.z = pack_f16(kn[4], kn[5]); kr.w = pack_f16(kn[6], kn[7]);
    *reinterpret_cast<uint4*>(kc + (size_t)pos * HD + d0) = kr;
    *reinterpret_cast<uint4*>(vc + (size_t)pos * HD + d0) = vraw;
  }

  float m = -INFINITY, l = 0.f, acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (int t0 = t_lo + grp; t0 < t_hi; t0 += NG * U) {
    uint4 kr[U], vr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * NG;
      const int tc = t < pos ? t : 0;   // t == pos comes from registers, t > pos is masked
      kr[u] = *reinterpret_cast<const uint4*>(kc + (size_t)tc * HD + d0);
      vr[u] = *reinterpret_cast<const uint4*>(vc + (size_t)tc * HD + d0);
    }
    // the round's scores first (independent chains; the 16-lane sums on DPP moves, had::sum16_xor: the additions of
    // `s += __shfl_xor(s, o)`, o = 1, 2, 4, 8), then the online-softmax updates in key order
    float k8a[U][8], v8a[U][8], sa[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * NG;
      unpack8h(kr[u], k8a[u]);
      unpack8h(vr[u], v8a[u]);
      if (t == pos) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { k8a[u][i] = kn[i]; v8a[u][i] = vn[i]; }
      }
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) s = __builtin_fmaf(q8[i], k8a[u][i], s);
      sa[u] = had::sum16_xor<LPK>(s);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + u * NG;
      const float s = sa[u];
      const float (&v8)[8] = v8a[u];
      if (t < t_hi) {
        const float mn = fmaxf(m, s);
        const float c = __expf(m - mn), p = __expf(s - mn);
        // (spelled out: left to the compiler, the instantiations of this kernel contract a * c + p * v differently -- one
        //  element in 8192 moved by an ulp between the grouped-query prologue and the plain kernel)
        l = __builtin_fmaf(l, c, p);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fmaf(acc[i], c, had::fmul(p, v8[i]));
        m = mn;
      }
    }
  }
  if (gl == 0) { s_m[grp] = m; s_l[grp] = l; }
#pragma unroll
  for (int i = 0; i < 8; ++i) s_acc[grp][d0 + i] = acc[i];
  __syncthreads();
  float M = -INFINITY, Lsum = 0.f, o = 0.f;
  if (tid < HD) {
    for (int g = 0; g < NG; ++g) M = fmaxf(M, s_m[g]);
    for (int g = 0; g < NG; ++g) {
      const float w = s_m[g] == -INFINITY ? 0.f : __expf(s_m[g] - M);
      Lsum = __builtin_fmaf(s_l[g], w, Lsum);
      o = __builtin_fmaf(s_acc[g][tid], w, o);
    }
  }
  if (!split) {
    if (tid < HD) a.out[(size_t)h * HD + tid] = (f16)(o / Lsum);
    return;
  }
  // Publish this split's state, then count arrivals; the last one merges.  The exchange uses
  // agent-scope atomic stores / loads (coherent across the XCDs' L2s by themselves) ordered by
  // "all my stores have completed" (s_waitcnt vmcnt(0)) -> barrier -> counter increment, instead of
  // __threadfence(): a release fence writes back the whole L2, measured ~20 us per launch here.
  float* mine = a.ws + ((size_t)h * kSplits + sidx) * (HD + 4);
  if (tid < HD) __hip_atomic_store(mine + tid, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (tid == 0) {
    __hip_atomic_store(mine + HD, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + HD + 1, Lsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __shared__ unsigned s_old;
  __syncthreads();
  if (tid == 0) s_old = __hip_atomic_fetch_add(a.counters + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  if (s_old != kSplits - 1) return;
  asm volatile("" ::: "memory");
  if (tid < HD) {
    float* base = a.ws + (size_t)h * kSplits * (HD + 4);
    float ms[kSplits], ls[kSplits], os[kSplits];
#pragma unroll
    for (int s2 = 0; s2 < kSplits; ++s2) {
      ms[s2] = __hip_atomic_load(base + s2 * (HD + 4) + HD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ls[s2] = __hip_atomic_load(base + s2 * (HD + 4) + HD + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      os[s2] = __hip_atomic_load(base + s2 * (HD + 4) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float Mx = -INFINITY;
#pragma unroll
    for (int s2 = 0; s2 < kSplits; ++s2) Mx = fmaxf(Mx, ms[s2]);
    float L2 = 0.f, o2 = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < kSplits; ++s2) {
      const float w = ms[s2] == -INFINITY ? 0.f : __expf(ms[s2] - Mx);
      L2 = __builtin_fmaf(ls[s2], w, L2);
      o2 = __builtin_fmaf(os[s2], w, o2);
    }
    a.out[(size_t)h * HD + tid] = (f16)(o2 / L2);
  }
  if (tid == 0) __hip_atomic_store(a.counters + h, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch
}

}  // namespace

size_t rope_attn_workspace_bytes(int heads, int head_dim) {
  return (size_t)heads * kSplits * (head_dim + 4) * sizeof(float) + (size_t)heads * sizeof(unsigned);
}

static int rope_attn_launch_common(AttnArgs a, const AttnZ* zz, int head_dim, int max_len, hipStream_t stream,
                                   void* workspace) {
  const int heads = a.heads;
  const bool split = workspace != nullptr && max_len > kSplitFromPos;
  if (split) {
    a.ws = reinterpret_cast<float*>(workspace);
    a.counters = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(workspace) +
                                             (size_t)heads * kSplits * (head_dim + 4) * sizeof(float));
  }
  const dim3 grid(heads, split ? kSplits : 1);
  if (zz && a.heads != a.kv_heads) {
    const int nq = a.heads * head_dim, nkv = a.kv_heads * head_dim;
    const size_t lds = ((size_t)had::buf_floats(nq) + 2 * (size_t)had::buf_floats(nkv)) * sizeof(float);
    if (head_dim == 128 && nq == 8192 && nkv == 1024)
      hipLaunchKernelGGL((rope_attn_decode_kernel<128, true, 13, 10>), grid, dim3(512 + 128), lds, stream, a, *zz);
    else if (head_dim == 128 && nq == 4096 && nkv == 1024)
      hipLaunchKernelGGL((rope_attn_decode_kernel<128, true, 12, 10>), grid, dim3(256 + 128), lds, stream, a, *zz);
    else
      return QUIP_ERR_UNSUPPORTED;
  } else if (zz) {
    const size_t lds = 3 * (size_t)had::buf_floats(zz->n) * sizeof(float);
    if (head_dim == 128)
      hipLaunchKernelGGL((rope_attn_decode_kernel<128, true>), grid, dim3(768), lds, stream, a, *zz);
    else if (head_dim == 64)
      hipLaunchKernelGGL((rope_attn_decode_kernel<64, true>), grid, dim3(768), lds, stream, a, *zz);
    else
      return QUIP_ERR_UNSUPPORTED;
  } else {
    const AttnZ none{};
    if (head_dim == 128)
      hipLaunchKernelGGL((rope_attn_decode_kernel<128, false>), grid, dim3(256), 0, stream, a, none);
    else if (head_dim == 64)
      hipLaunchKernelGGL((rope_attn_decode_kernel<64, false>), grid, dim3(256), 0, stream, a, none);
    else
      return QUIP_ERR_UNSUPPORTED;
  }
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

int rope_attn_decode_launch(const void* q, const void* k, const void* v, const float* cos, const float* sin,
                            const int64_t* pos, void* kcache, void* vcache, void* out, int heads, int kv_heads,
                            int head_dim, int max_len, float scale, hipStream_t stream, void* workspace, int window) {
  if (heads < 1 || kv_heads < 1 || heads % kv_heads != 0 || max_len < 1) return QUIP_ERR_BAD_SHAPE;
  AttnArgs a{reinterpret_cast<const f16*>(q), reinterpret_cast<const f16*>(k), reinterpret_cast<const f16*>(v),
             cos, sin, pos, reinterpret_cast<f16*>(kcache), reinterpret_cast<f16*>(vcache),
             reinterpret_cast<f16*>(out), heads, kv_heads, max_len, scale, nullptr, nullptr, window};
  return rope_attn_launch_common(a, nullptr, head_dim, max_len, stream, workspace);
}

bool rope_attn_decode_z_supported(int heads, int kv_heads, int head_dim) {
  const int n = heads * head_dim;
  // grouped queries: Llama-2-70B / Llama-3-70B (64 heads, 8 KV heads) and Llama-3-8B / Mistral-7B (32 / 8)
  if (heads != kv_heads) return head_dim == 128 && kv_heads == 8 && (heads == 64 || heads == 32);
  return heads >= 1 && (head_dim == 64 || head_dim == 128) && n >= 256 && n <= 4096 && (n & (n - 1)) == 0;
}

int rope_attn_decode_z_launch(const void* const* z, const void* const* post, const float* scales, const float* cos,
                              const float* sin, const int64_t* pos, void* kcache, void* vcache, void* out, int heads,
                              int kv_heads, int head_dim, int max_len, float scale, hipStream_t stream,
                              void* workspace, int window) {
  if (heads < 1 || kv_heads < 1 || max_len < 1) return QUIP_ERR_BAD_SHAPE;
  if (!rope_attn_decode_z_supported(heads, kv_heads, head_dim)) return QUIP_ERR_UNSUPPORTED;
  AttnArgs a{nullptr, nullptr, nullptr, cos, sin, pos, reinterpret_cast<f16*>(kcache),
             reinterpret_cast<f16*>(vcache), reinterpret_cast<f16*>(out), heads, kv_heads, max_len, scale, nullptr,
             nullptr, window};
  AttnZ zz{};
  const int n = heads * head_dim;
  for (int i = 0; i < 3; ++i) {
    zz.z[i] = reinterpret_cast<const f16*>(z[i]);
    zz.post[i] = reinterpret_cast<const f16*>(post[i]);
    zz.scale[i] = scales[i];
  }
  zz.n = n;
  zz.logL = 31 - __builtin_clz((unsigned)n);
  return rope_attn_launch_common(a, &zz, head_dim, max_len, stream, workspace);
}

// Greedy tail of the decode step (example_generate.py's argmax sampling with temperature 0): next token =
// first index of the largest logit (torch.argmax's tie rule), stored to tok; pos += 1.  One workgroup: the three
// framework launches it replaces (reduce, copy, add) cost ~20 us of a 2.5 ms token.
namespace {
__global__ __launch_bounds__(1024) void argmax_step_kernel(const f16* __restrict__ logits, int n,
                                                           int64_t* __restrict__ tok, int64_t* __restrict__ pos) {
  __shared__ float sv[16];
  __shared__ int si[16];
  const int tid = threadIdx.x;
  float best = -3.0e38f;
  int bi = 0x7fffffff;
  for (int i = tid * 8; i < n; i += 1024 * 8) {
    if (i + 8 <= n) {
      const uint4 q = *reinterpret_cast<const uint4*>(logits + i);
      const f16* h = reinterpret_cast<const f16*>(&q);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = (float)h[j];
        if (v > best || (v == best && i + j < bi)) { best = v; bi = i + j; }
      }
    } else {
      for (int j = i; j < n; ++j) {
        const float v = (float)logits[j];
        if (v > best || (v == best && j < bi)) { best = v; bi = j; }
      }
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    // (all-NaN / all -inf logits -- e.g. the answer of a persistent launch that gave up -- leave no index: token 0, a
    //  valid row of the embedding table, instead of INT_MAX into the next step's lookup)
    tok[0] = bi < n ? bi : 0;
    pos[0] += 1;
  }
}
}  // namespace

int argmax_step_launch(const void* logits, int n, void* tok, void* pos, hipStream_t stream) {
  if (n < 1) return QUIP_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(argmax_step_kernel, dim3(1), dim3(1024), 0, stream, reinterpret_cast<const f16*>(logits), n,
                     reinterpret_cast<int64_t*>(tok), reinterpret_cast<int64_t*>(pos));
  return hipGetLastError() == hipSuccess ? QUIP_OK : QUIP_ERR_LAUNCH;
}

}  // namespace quip
