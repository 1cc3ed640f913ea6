// split3.hip — fp32 -> SPLIT 16-bit operands: v = hi + lo with hi = h16(v), lo = h16(v - hi) (22 significand bits).  An fp32 contraction
// runs on the 16-bit matrix cores as two or three products of such halves with fp32 accumulation,
//
//      x . w  =  x_hi . w_hi  +  x_lo . w_hi  +  x_hi . w_lo   ( + x_lo . w_lo, dropped: 2^-22 relative )
//
// as ONE implicit GEMM over a longer K: the weight operand is [w_hi | w_lo] for a plain 16-bit activation read twice
// (VNQA_CONV_X_WRAP2 / VNQA_GEMM_X_WRAP2) or [w_hi | w_hi | w_lo] for an activation laid out as [hi | lo | hi] by its producer
// (VNQA_CONV_DUAL_OUT | VNQA_CONV_DUAL_HI2, csrc/conv_ps.hip), so the tuned 16-bit conv / GEMM kernels run unchanged.  Precision
// 'fp16h' (the tolerance mode) splits the operands whose rounding dominates the logits error; this file makes the WEIGHT halves
// (one pass over the fp32 K-major pack).  The round-4 'fp16x' machinery (raw accumulators + fp32 finishing pass, gradient split
// scales) was retired in round 5.
#include "vnqa_common.h"

namespace {

// rows x c fp32 (row stride src_ld) -> three 16-bit destinations with row stride dst_ld: hi, (optional) lo and (optional) a second
// copy of hi; with lo == hi2 == null it is a (scaled) fp32 -> 16-bit cast
__global__ void __launch_bounds__(256) split3_kernel(const float* __restrict__ x, unsigned short* __restrict__ hi,
                                                     unsigned short* __restrict__ lo, unsigned short* __restrict__ hi2,
                                                     long long rows, int c8, long long src_ld, long long dst_ld,
                                                     const float* __restrict__ scale) {
  // scale (optional, DEVICE scalar, a power of two): the values are multiplied by it before they are split — gradients (1e-5 .. 1e-8
  // here) are lifted into fp16's normal range first; the consumer multiplies its fp32 result by 1 / scale
  const float sc = scale != nullptr ? *scale : 1.f;
  const long long total = rows * c8;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c8;
    const int j = (int)(i - r * c8) * 8;
    const float4 a = *(const float4*)(x + r * src_ld + j);
    const float4 b = *(const float4*)(x + r * src_ld + j + 4);
    const float v[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = pack2_h16(v[2 * e], v[2 * e + 1]);
      // an overflowed hi (|v| > 65504 in the fp16 build) would make the residual inf - inf: keep lo = 0 there
      const float r0 = v[2 * e] - h16_lo(h[e]), r1 = v[2 * e + 1] - h16_hi(h[e]);
      l[e] = pack2_h16(r0 == r0 ? r0 : 0.f, r1 == r1 ? r1 : 0.f);
    }
    const uint4 hv = make_uint4(h[0], h[1], h[2], h[3]);
    *(uint4*)(hi + r * dst_ld + j) = hv;
    if (lo != nullptr) *(uint4*)(lo + r * dst_ld + j) = make_uint4(l[0], l[1], l[2], l[3]);
    if (hi2 != nullptr) *(uint4*)(hi2 + r * dst_ld + j) = hv;
  }
}

int grid_for(long long total) {
  long long b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

}  // namespace

extern "C" int vnqa_split3_f32(const float* x, void* hi, void* lo, void* hi2, int64_t rows, int32_t c, int64_t src_ld,
                               int64_t dst_ld, const float* scale, void* stream) {
  VNQA_CHECK_ARG(x && hi, "split3_f32: null pointer");
  VNQA_CHECK_ARG(rows > 0 && c > 0 && c % 8 == 0 && src_ld >= c && src_ld % 4 == 0 && dst_ld >= c && dst_ld % 8 == 0,
                 "split3_f32: rows=%lld c=%d src_ld=%lld dst_ld=%lld (c %% 8, src_ld %% 4, dst_ld %% 8 must be 0)",
                 (long long)rows, c, (long long)src_ld, (long long)dst_ld);
  VNQA_CHECK_ARG((((uintptr_t)x | (uintptr_t)hi | (uintptr_t)lo | (uintptr_t)hi2) & 15) == 0, "split3_f32: 16-byte alignment");
  hipLaunchKernelGGL(split3_kernel, dim3(grid_for(rows * (c / 8))), dim3(256), 0, (hipStream_t)stream, x, (unsigned short*)hi,
                     (unsigned short*)lo, (unsigned short*)hi2, (long long)rows, c / 8, (long long)src_ld, (long long)dst_ld, scale);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
