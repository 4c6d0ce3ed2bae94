// Cross-workgroup hand-off primitives of the persistent decode engine (gfx950).
//
// Per-XCD L2s are not coherent with each other and a CU's vector L1 is never refreshed by another
// CU's stores, so every word that crosses workgroups inside a launch is written and read with
// device-scope (sc1) accesses:
//   granule  = one naturally aligned 8-byte {value, tag} written by ONE sc1 store; the data is its
//              own flag (a reader re-reads until tag == epoch), no fence on either side;
//   payload  = 16-byte sc1 stores, drained (vmcnt(0)) by every storing wave, then ONE sc1 flag
//              store; the reader polls the flag with sc1 loads and reads the payload with sc1 loads.
// Every spin is bounded: a reader that gives up raises the launch-wide error word and every other
// spin loop leaves as soon as it sees that word, so a stuck launch ends instead of hanging the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace quip {
namespace esync {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));

constexpr uint32_t kSpinLimit = 1u << 21;   // polls before a reader gives up (a poll is ~0.5-1 us)

// ---- stores ---------------------------------------------------------------------------------
__device__ __forceinline__ void st_granule(void* p, uint32_t value, uint32_t tag) {
  const u32x2_t v = {value, tag};
  asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_granule2(void* p, uint32_t v0, uint32_t v1, uint32_t tag) {   // two granules, 16 B
  const u32x4_t v = {v0, tag, v1, tag};
  // (s_nop 1: a VMEM store of more than 8 bytes reads its data registers over two more cycles, and the compiler pads no
  //  hazard for an instruction inside an asm statement: the VALU instruction behind it could overwrite them)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_payload16(void* p, const u32x4_t& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_word(void* p, uint32_t v) {
  asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- loads (issued, not waited: the caller waits with drain() / a counted vmcnt) ------------
__device__ __forceinline__ void ld16(u32x4_t& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
// the same into a register that keeps its value in the lanes that do not execute the load (a poll loop's lanes whose
// pieces have arrived sit out the retries: 256 workgroups re-reading a whole vector per retry delay the stores they wait for)
__device__ __forceinline__ void ld16_keep(u32x4_t& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld8(u32x2_t& dst, const void* p) {
  asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void ld4(uint32_t& dst, const void* p) {
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
template <typename T>
__device__ __forceinline__ void own(T& v) { asm volatile("" : "+v"(v)); }   // after a wait: the value is the register's

// ---- two fp32 values in one granule: 20-bit mantissas against the pair's larger binary exponent, 8-bit exponent, 16-bit tag ------
// {m_a (20) | m_b low 12, m_b high 8 | e << 8 | tag16 << 16}, m = rint(value 2^(145 - e)): 2^-19 of the larger one of the pair.
// e = 255: an inf / NaN in the pair (both read back as NaN).  A 16-bit tag suffices where every granule of the vector is
// rewritten in every block (a stale one carries the tag of the block before, which differs in the hand-off index).
__device__ __forceinline__ void pack20x2(float a, float b, uint32_t tag, uint32_t& w0, uint32_t& w1) {
  const uint32_t ea = (__builtin_bit_cast(uint32_t, a) >> 23) & 0xffu, eb = (__builtin_bit_cast(uint32_t, b) >> 23) & 0xffu;
  uint32_t e = ea > eb ? ea : eb;
  e = e < 19u ? 19u : e;
  const float sc = __builtin_bit_cast(float, (272u - (e > 254u ? 254u : e)) << 23);
  int ma = (int)__builtin_rintf(a * sc), mb = (int)__builtin_rintf(b * sc);
  ma = ma < -524287 ? -524287 : (ma > 524287 ? 524287 : ma);
  mb = mb < -524287 ? -524287 : (mb > 524287 ? 524287 : mb);
  w0 = ((uint32_t)ma & 0xfffffu) | ((uint32_t)mb << 20);
  w1 = (((uint32_t)mb >> 12) & 0xffu) | (e << 8) | ((tag & 0xffffu) << 16);
}
__device__ __forceinline__ void unpack20x2(uint32_t w0, uint32_t w1, float& a, float& b) {
  const uint32_t e = (w1 >> 8) & 0xffu;
  const float sc = e == 255u ? __builtin_bit_cast(float, 0x7fc00000u) : __builtin_bit_cast(float, (e - 18u) << 23);      // 2^(e - 145)
  a = (float)((int)(w0 << 12) >> 12) * sc;
  b = (float)((int)(((w0 >> 20) | (w1 << 12)) << 12) >> 12) * sc;
}

// ---- bounded spinning -----------------------------------------------------------------------
// One step of a spin loop of a whole wave: `done` is this lane's condition.  Returns true when the
// wave may leave (everything arrived, or the launch is already failing).  err: the launch-wide error
// word (0 = fine); code: who gives up.
__device__ __forceinline__ bool spin_step(bool done, uint32_t& spins, uint32_t* err, uint32_t code) {
  if (__all(done)) return true;
  ++spins;
  if ((spins & 63u) == 0u) {
    uint32_t e;
    ld4(e, err);
    drain();
    own(e);
    if (__builtin_amdgcn_readfirstlane(e) != 0u) return true;
    if (spins >= kSpinLimit) {
      // (lane id from the exec mask, not from threadIdx: nothing to keep live across a whole persistent launch)
      if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) st_word(err, code);
      return true;
    }
  }
  __builtin_amdgcn_s_sleep(1);
  return false;
}

}  // namespace esync
}  // namespace quip
