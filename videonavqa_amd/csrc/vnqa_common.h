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

__device__ __forceinline__ float bf16_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned)b) << 16);
}
// round-to-nearest-even; lowers to v_cvt_pk_bf16_f32 on gfx950 (keeps NaN a NaN)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(unsigned short, h);
}

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
