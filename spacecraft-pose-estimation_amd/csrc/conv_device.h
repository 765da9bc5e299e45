// Device-side helpers shared by the convolution kernels (MFMA fragments, 16-bit casts, XCD remap).
#pragma once
#include "common.h"

namespace scpose {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

template <typename T> struct FragOf;
template <> struct FragOf<__bf16> { typedef bf16x8 type; };
template <> struct FragOf<_Float16> { typedef f16x8 type; };

template <typename T>
__device__ __forceinline__ f32x4 mfma16(typename FragOf<T>::type a, typename FragOf<T>::type b,
                                        f32x4 c) {
  if constexpr (sizeof(T) == 2 && __is_same(T, __bf16))
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <typename T> __device__ __forceinline__ uint16_t to_bits(float f) {
  T t = (T)f;
  return __builtin_bit_cast(uint16_t, t);
}
template <typename T> __device__ __forceinline__ float from_bits(uint16_t v) {
  return (float)__builtin_bit_cast(T, v);
}

// single-instruction asm pieces of the hand-scheduled MFMA loops (the compiler neither reorders them nor
// inserts waits for them; the loops wait explicitly)
template <int OFF, typename F>
__device__ __forceinline__ void lds_read16(F& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <typename F>
__device__ __forceinline__ void lds_landed(F& d) { asm volatile("" : "+v"(d)); }
template <typename T>
__device__ __forceinline__ void mfma16_acc(f32x4& c, const typename FragOf<T>::type& a, const typename FragOf<T>::type& b) {
  if constexpr (__is_same(T, __bf16)) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// After a hand-scheduled MFMA loop: the trailing s_nop covers the MFMA -> VALU read latency, and every accumulator is
// passed through an (empty) asm so that the compiler cannot hoist their consumers above that s_nop.
template <bool AGPR, typename V>
__device__ __forceinline__ void mfma_result_fence(V& c) {
  if constexpr (AGPR) asm volatile("" : "+a"(c));
  else asm volatile("" : "+v"(c));
}

// Before a hand-scheduled MFMA loop: the hazard recogniser does not see inline-asm MFMAs, so a VALU write of an
// accumulator (its zero / bias initialisation, which the scheduler likes to sink right in front of the first
// MFMA) could be followed by the MFMA's SrcC read without the required wait states.  Passing the accumulator
// through this asm orders the initialisation before two wait states.
template <bool AGPR, typename V>
__device__ __forceinline__ void mfma_input_fence(V& c) {
  if constexpr (AGPR) asm volatile("s_nop 1" : "+a"(c));
  else asm volatile("s_nop 1" : "+v"(c));
}

// MFMA result -> v_permlane32_swap: hipcc pads that pair with 3-4 wait states, too few behind the 8-pass 16x16x32 MFMA -- lanes 12-15
// of the LAST accumulator written come out stale.  20 explicit wait states IN ONE STATEMENT WITH THE ACCUMULATORS
// AS OPERANDS: the MFMAs (their producers) cannot sink below it and the swaps (their consumers) cannot rise above it.  (A nop
// statement without the operands orders nothing: in round 4 the scheduler put the MFMAs behind such a pad and the fused block's
// product build returned one stale 16-bit value per vector in columns 12-15 -- tools_dev/where_block_differs.py.)
__device__ __forceinline__ void mfma_swap_pad(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
}
__device__ __forceinline__ void mfma_swap_pad(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, f32x4& a4, f32x4& a5) {
  asm volatile("s_nop 15\n\ts_nop 3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5));
}

// two fp32 -> one dword of two 16-bit values (a in the low half), one v_cvt_pk_* instruction, RNE like the scalar cast
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  const f32x2 v = {a, b};
  if constexpr (__is_same(T, __bf16)) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  else return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_t));
}
// ReLU on a packed pair of bf16/f16 (sign-magnitude formats): signed 16-bit max against `floor2`, which is 0 for
// ReLU and 0x80008000 (no-op) otherwise -- one v_pk_max_i16, no branch.  (-0 and negative NaNs become +0.)
__device__ __forceinline__ uint32_t relu2_16(uint32_t x, uint32_t floor2) {
  const i16x2_t r = __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, x), __builtin_bit_cast(i16x2_t, floor2));
  return __builtin_bit_cast(uint32_t, r);
}

// Blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a contiguous range of
// logical ids so that neighbouring tiles / Cout blocks of one tile hit the same L2.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
  const int xcd = b & 7, q8 = nb >> 3, r8 = nb & 7;
  const int base = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
  return base + (b >> 3);
}

// Dynamic tile queue of the persistent kernels that other work may run beside (stem, layer1: bench.py's decode / PnP of
// the previous step overlap them).  A static split of the tiles over the workgroups makes the launch as slow as its
// unluckiest workgroup: a CU that another kernel holds for 0.7 ms delays its share by 0.7 ms.  Here every workgroup
// claims its next tile from one device-wide counter instead (sched[0]; one returning atomic per tile, issued a whole
// tile before its value is needed), so the tiles go to whichever CUs are free.  sched[1] counts finished workgroups;
// the last one zeroes both words, which leaves the pair ready for the next launch (the pair is zero-initialised once,
// when the engine is created; launches that share a pair are ordered on one stream).  Tile t's result does not depend on
// who computes it.
__device__ __forceinline__ int tile_claim(uint32_t* sched, int total) {
  const uint32_t c = __hip_atomic_fetch_add(sched, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return c < (uint32_t)total ? (int)c : -1;
}
__device__ __forceinline__ void tile_retire(uint32_t* sched) {   // one lane per workgroup, after its last claim has returned
  const uint32_t d = __hip_atomic_fetch_add(sched + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (d == gridDim.x - 1) {
    __hip_atomic_store(sched, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sched + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// XCD-local form of the queue (kernels whose neighbouring tiles share cache lines: the 1-pixel halo of a 16-byte-per-pixel
// row starts 16 bytes in front of a line): one counter per XCD (sched[0..7]; sched[8] counts finished workgroups), XCD k
// owns the tiles of images k, k + 8, ... in raster order, so the 32 CUs of an XCD work on ~32 consecutive tiles of the
// same one or two images at the same time and the lines two tiles share are fetched into that XCD's L2 once.  A
// workgroup whose own list is exhausted takes tiles from the next XCD's list (CUs held by another kernel delay nothing).
__device__ __forceinline__ int xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return (int)(v & 7u);
}
__device__ __forceinline__ int tile_claim_xcd(uint32_t* sched, int xcd, int n_img, int tiles_per_img) {
  for (int j = 0; j < 8; ++j) {
    const int k = (xcd + j) & 7;
    const int cnt = ((n_img - k + 7) >> 3) * tiles_per_img;
    if (cnt <= 0) continue;
    const uint32_t c = __hip_atomic_fetch_add(sched + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c < (uint32_t)cnt) {
      const int ii = (int)c / tiles_per_img;
      return (ii * 8 + k) * tiles_per_img + ((int)c - ii * tiles_per_img);
    }
  }
  return -1;
}
__device__ __forceinline__ void tile_retire_xcd(uint32_t* sched) {
  const uint32_t d = __hip_atomic_fetch_add(sched + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (d == gridDim.x - 1)
    for (int k = 0; k < 9; ++k) __hip_atomic_store(sched + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// DT: 0 = bf16, 1 = f16 (an int so that profiler kernel names demangle: conv_igemm_kernel<0,3,1,3,4>)
template <int DT> struct DtOf { typedef __bf16 type; };
template <> struct DtOf<1> { typedef _Float16 type; };


}  // namespace scpose
