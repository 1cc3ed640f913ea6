// Shared host/device helpers for the vnqa HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>

#include "../../include/vnqa_hip.h"

// thread-local last-error message (include/vnqa_hip.h: vnqa_last_error)
void vnqa_set_error(const char* fmt, ...);

#define VNQA_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      vnqa_set_error(__VA_ARGS__);                \
      return VNQA_ERR_INVALID_ARG;                \
    }                                             \
  } while (0)

#define VNQA_CHECK_LAUNCH()                                            \
  do {                                                                 \
    hipError_t e__ = hipGetLastError();                                \
    if (e__ != hipSuccess) {                                           \
      vnqa_set_error("HIP launch failed: %s", hipGetErrorString(e__)); \
      return VNQA_ERR_HIP;                                             \
    }                                                                  \
  } while (0)

#ifdef __HIPCC__
typedef __attribute__((ext_vector_type(8))) short vnqa_bf16x8;
typedef __attribute__((ext_vector_type(4))) short vnqa_bf16x4;
typedef __attribute__((ext_vector_type(4))) float vnqa_f32x4;
typedef __attribute__((ext_vector_type(16))) float vnqa_f32x16;
typedef unsigned short vnqa_bf16;  // raw bits

// ---- the library's 16-bit storage format -----------------------------------------------------------------------------
// Every kernel treats its 16-bit elements through the helpers below (conversion, unpacking of two packed elements, the
// three MFMA shapes), so ONE compile-time switch selects the format of the whole library:
//   default            bf16  (8 exponent bits / 8 significand bits: the fp32 range, 2^-9 relative rounding)
//   -DVNQA_H16_IS_F16  IEEE fp16 (5 / 11 bits: 2^-12 relative rounding — 8x finer — at the same MFMA rate; finite range
//                      +-65504; larger magnitudes become inf, as in any IEEE fp16 pipeline).  Built as libvnqa_hip_f16.so, selected by
//                      precision='fp16' on the Python side; VNQA_BF16 then means "the library's 16-bit format".
// (Names keep the historical `bf16`: f32_to_bf16 = fp32 -> storage format.)
#ifdef VNQA_H16_IS_F16
typedef __attribute__((ext_vector_type(8))) _Float16 vnqa_f16x8_;
typedef __attribute__((ext_vector_type(4))) _Float16 vnqa_f16x4_;
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return (float)__builtin_bit_cast(_Float16, b); }
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  return __builtin_bit_cast(unsigned short, (_Float16)f);          // round-to-nearest-even; |f| > 65504 -> inf (IEEE)
}
// two elements -> one 32-bit word (low half = a): one packed conversion instead of two conversions + a pack
typedef __attribute__((ext_vector_type(2))) _Float16 vnqa_f16x2_;
typedef __attribute__((ext_vector_type(2))) float vnqa_f32x2_;
__device__ __forceinline__ unsigned pack2_h16(float a, float b) {
  const vnqa_f32x2_ v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, vnqa_f16x2_));
}
__device__ __forceinline__ float h16_lo(unsigned w) { return bf16_to_f32((unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float h16_hi(unsigned w) { return bf16_to_f32((unsigned short)(w >> 16)); }
// element-wise max of two packed pairs (a 2x2 max-pool over stored values: the max of stored values IS a stored value, no rounding)
__device__ __forceinline__ unsigned h16x2_max(unsigned a, unsigned b) {
  unsigned r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ vnqa_f32x4 VNQA_MFMA_16x16x32(vnqa_bf16x8 a, vnqa_bf16x8 b, vnqa_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vnqa_f16x8_, a), __builtin_bit_cast(vnqa_f16x8_, b), c, 0, 0, 0);
}
__device__ __forceinline__ vnqa_f32x16 VNQA_MFMA_32x32x16(vnqa_bf16x8 a, vnqa_bf16x8 b, vnqa_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vnqa_f16x8_, a), __builtin_bit_cast(vnqa_f16x8_, b), c, 0, 0, 0);
}
__device__ __forceinline__ vnqa_f32x4 VNQA_MFMA_16x16x16(vnqa_bf16x4 a, vnqa_bf16x4 b, vnqa_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(vnqa_f16x4_, a), __builtin_bit_cast(vnqa_f16x4_, b), c, 0, 0, 0);
}
#else
__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}
// round-to-nearest-even; lowers to v_cvt_pk_bf16_f32 on gfx950 (keeps NaN a NaN)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned pack2_h16(float a, float b) {
  return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);      // (the compiler fuses this into v_cvt_pk_bf16_f32)
}
// the two elements packed in a 32-bit word
__device__ __forceinline__ float h16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float h16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
// element-wise max of two packed pairs (a 2x2 max-pool over stored values: the max of stored values IS a stored value, no rounding);
// bf16 has no packed max: through fp32 and back by bit selection
__device__ __forceinline__ unsigned h16x2_max(unsigned a, unsigned b) {
  float lo, hi;
  asm("v_max_f32 %0, %1, %2" : "=v"(lo) : "v"(__uint_as_float(a << 16)), "v"(__uint_as_float(b << 16)));
  asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(__uint_as_float(a & 0xffff0000u)), "v"(__uint_as_float(b & 0xffff0000u)));
  return (__float_as_uint(hi) & 0xffff0000u) | (__float_as_uint(lo) >> 16);
}
__device__ __forceinline__ vnqa_f32x4 VNQA_MFMA_16x16x32(vnqa_bf16x8 a, vnqa_bf16x8 b, vnqa_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ vnqa_f32x16 VNQA_MFMA_32x32x16(vnqa_bf16x8 a, vnqa_bf16x8 b, vnqa_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ vnqa_f32x4 VNQA_MFMA_16x16x16(vnqa_bf16x4 a, vnqa_bf16x4 b, vnqa_f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
#endif

template <typename T> struct ElemOps;
template <> struct ElemOps<vnqa_bf16> {
  static __device__ __forceinline__ float load(vnqa_bf16 v) { return bf16_to_f32(v); }
  static __device__ __forceinline__ vnqa_bf16 store(float f) { return f32_to_bf16(f); }
};
template <> struct ElemOps<float> {
  static __device__ __forceinline__ float load(float v) { return v; }
  static __device__ __forceinline__ float store(float f) { return f; }
};

__device__ __forceinline__ float wave_reduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_reduce_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif
