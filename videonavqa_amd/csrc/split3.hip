// split3.hip — the elementwise halves of the "x3" products: an fp32 contraction evaluated on the 16-bit matrix cores as THREE
// products of fp16 halves with fp32 accumulation,
//
//      x . w  =  x_hi . w_hi  +  x_lo . w_hi  +  x_hi . w_lo   ( + x_lo . w_lo, dropped: 2^-22 relative )
//
// with  v_hi = fp16(v),  v_lo = fp16(v - v_hi)  (v_hi + v_lo carries 22 significand bits of v).  The three products are ONE
// implicit GEMM over a three times longer K: the activation operand is stored channel-concatenated [hi | lo | hi] and the
// weight operand [w_hi ; w_hi ; w_lo], so the library's tuned 16-bit conv / GEMM kernels run unchanged; they hand back the raw
// fp32 accumulators (vnqa_conv2d_igemm_raw) and vnqa_x3_post applies bias / ReLU / 2x2 max-pool / affine to them in fp32.
// This is precision='fp16x': the tolerance-compliant 16-bit-MFMA mode (logits within 1e-3 of the fp32 reference with a wide
// margin: measured ~1e-5) that replaces the 1/16-rate exact-f32 matrix path where only the tolerance, not bit-exactness, is
// asked for.  Replaces nn.Conv2d / nn.Linear forward + backward at the same call sites as vnqa_conv2d_igemm_fwd
// (models/obj_detector.py:72-82, models/film_attn_pt_stem.py:211,219,224,244).
#include "vnqa_common.h"

namespace {

// rows x c fp32 (row stride src_ld) -> three 16-bit destinations with row stride dst_ld: hi, (optional) lo and (optional) a second
// copy of hi; with lo == hi2 == null it is the scaled fp32 -> 16-bit cast of the one-product backward ('x1g')
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

struct PostArgs {
  const float* raw;         // [n][h][w][c_out] fp32 (dense, no halo): the conv's accumulators
  const float* bias;
  const float* post_scale;
  const float* post_shift;
  const float* border_sub;  // [n][2w + 2(h-2)][c_out] fp32 or null
  float* y;                 // padded NHWC [n][ho + 2 yh][wo + 2 yh][c_y] fp32, interior written
  const float* raw_scale;   // optional DEVICE scalar: raw sums are multiplied by it first (1 / the operand's split scale)
  unsigned short* y16;      // != null: the output as the NEXT x3 product's operand instead — 16-bit [..][c_y], channels
                            // [hi | lo | hi] at c, c_out + c, 2 c_out + c (c_y >= 3 c_out): no fp32 round trip between layers
  int n, h, w, c_out, c_y, y_halo, relu, pool, two;
  int zero_halo;            // also write zeros to y's halo ring (a fresh, uninitialised output): no separate halo launch
};

__device__ __forceinline__ float4 post_one(const PostArgs& p, int n, int y, int x, int c, const float4 bias) {
  float4 v = *(const float4*)(p.raw + (((size_t)n * p.h + y) * p.w + x) * p.c_out + c);
  if (p.raw_scale != nullptr) {
    const float rs = *p.raw_scale;
    v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs;
  }
  if (p.border_sub != nullptr) {
    int ring = -1;
    if (y == 0) ring = x;
    else if (y == p.h - 1) ring = p.w + x;
    else if (x == 0) ring = 2 * p.w + (y - 1);
    else if (x == p.w - 1) ring = 2 * p.w + (p.h - 2) + (y - 1);
    if (ring >= 0) {
      const float4 s = *(const float4*)(p.border_sub + ((size_t)n * (2 * p.w + 2 * (p.h - 2)) + ring) * p.c_out + c);
      v.x -= s.x; v.y -= s.y; v.z -= s.z; v.w -= s.w;
    }
  }
  v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
  if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  return v;
}

// y = post( pool2?( relu?( raw - border_sub + bias ) ) ): the epilogue of vnqa_conv2d_igemm_fwd_ex, in fp32, on dense raw sums
__global__ void __launch_bounds__(256) x3_post_kernel(const PostArgs p) {
  const int ho = p.pool ? p.h >> 1 : p.h, wo = p.pool ? p.w >> 1 : p.w;
  const int c4 = p.c_out >> 2;
  const long long total = (long long)p.n * ho * wo * c4;
  const int hyp = ho + 2 * p.y_halo, wyp = wo + 2 * p.y_halo;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4) * 4;
    long long q = i / c4;
    const int xo = (int)(q % wo);
    q /= wo;
    const int yo = (int)(q % ho);
    const int n = (int)(q / ho);
    const float4 b = p.bias != nullptr ? *(const float4*)(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v;
    if (p.pool) {
      const float4 a0 = post_one(p, n, 2 * yo, 2 * xo, c, b), a1 = post_one(p, n, 2 * yo, 2 * xo + 1, c, b);
      const float4 a2 = post_one(p, n, 2 * yo + 1, 2 * xo, c, b), a3 = post_one(p, n, 2 * yo + 1, 2 * xo + 1, c, b);
      v.x = fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x));
      v.y = fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y));
      v.z = fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z));
      v.w = fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w));
    } else {
      v = post_one(p, n, yo, xo, c, b);
    }
    if (p.post_scale != nullptr) {
      const float4 s = *(const float4*)(p.post_scale + c), t = *(const float4*)(p.post_shift + c);
      v.x = v.x * s.x + t.x; v.y = v.y * s.y + t.y; v.z = v.z * s.z + t.z; v.w = v.w * s.w + t.w;
    }
    const size_t pix = (((size_t)n * hyp + yo + p.y_halo) * wyp + xo + p.y_halo) * p.c_y;
    if (p.y16 == nullptr) {
      *(float4*)(p.y + pix + c) = v;
    } else {
      uint2 h, l;
      h.x = pack2_h16(v.x, v.y);
      h.y = pack2_h16(v.z, v.w);
      const float r0 = v.x - h16_lo(h.x), r1 = v.y - h16_hi(h.x), r2 = v.z - h16_lo(h.y), r3 = v.w - h16_hi(h.y);
      l.x = pack2_h16(r0 == r0 ? r0 : 0.f, r1 == r1 ? r1 : 0.f);
      l.y = pack2_h16(r2 == r2 ? r2 : 0.f, r3 == r3 ? r3 : 0.f);
      unsigned short* o = p.y16 + pix + c;
      *(uint2*)o = h;
      if (p.two) {                            // plain fp16 output: the consumer reads it twice along K (two products, VNQA_CONV_X_WRAP2)
      } else {
        *(uint2*)(o + p.c_out) = l;
        *(uint2*)(o + 2 * p.c_out) = h;
      }
    }
  }
  if (p.zero_halo && p.y_halo > 0) {          // the halo ring(s) of every image: c_y elements per halo pixel, 8 bytes at a time
    const int esz = p.y16 == nullptr ? 4 : 2;
    const int per = (int)((size_t)p.c_y * esz / 8);                    // 8-byte pieces per pixel (c_y % 4 == 0)
    const int hal = p.y_halo;
    const long long ring = (long long)hyp * wyp - (long long)ho * wo;   // halo pixels per image
    const long long tot = (long long)p.n * ring * per;
    char* const base = p.y16 == nullptr ? (char*)p.y : (char*)p.y16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
      const int k = (int)(i % per);
      long long q = i / per;
      const long long r = q % ring;
      const int n = (int)(q / ring);
      int py, px;
      const long long top = (long long)hal * wyp;
      if (r < top) { py = (int)(r / wyp); px = (int)(r % wyp); }
      else if (r < 2 * top) { const long long t = r - top; py = hyp - hal + (int)(t / wyp); px = (int)(t % wyp); }
      else { const long long t = r - 2 * top; py = hal + (int)(t / (2 * hal)); const int side = (int)(t % (2 * hal));
             px = side < hal ? side : wyp - 2 * hal + side; }
      *(uint2*)(base + ((((size_t)n * hyp + py) * wyp + px) * p.c_y) * esz + (size_t)k * 8) = make_uint2(0u, 0u);
    }
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

// The split scale of a gradient tensor in ONE launch: state = {bits of max |x| (uint), scale, 1 / scale, blocks done (uint)}.  Every
// block folds its maximum into state[0] (non-negative floats order like their bit patterns); the last block to finish turns it into the
// power of two that lifts max |x| into [2^12, 2^13) (1 for an all-zero or non-finite tensor), writes scale and 1 / scale, and clears the
// two words for the next call on the same state.
__global__ void __launch_bounds__(256) grad_scale_kernel(const float* __restrict__ x, long long n4, long long n, unsigned* __restrict__ state) {
  float m = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = *(const float4*)(x + 4 * i);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));
  unsigned bits = __float_as_uint(m);          // (fmaxf drops NaNs; an inf stays the maximum)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)bits, off, 64);
    bits = o > bits ? o : bits;
  }
  __shared__ unsigned part[4];
  __shared__ int last;
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = bits;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned b = part[0];
    for (int i = 1; i < 4; ++i) b = part[i] > b ? part[i] : b;
    atomicMax(&state[0], b);
    __threadfence();
    last = atomicAdd(&state[3], 1u) == gridDim.x - 1;
    if (last) {
      const float amax = __uint_as_float(atomicMax(&state[0], 0u));
      float scale = 1.f;
      if (amax > 0.f && amax < __builtin_inff()) {
        int e = ilogbf(amax);
        e = e < -100 ? -100 : e;
        scale = ldexpf(1.f, 12 - e);
      }
      ((float*)state)[1] = scale;
      ((float*)state)[2] = 1.f / scale;
      __threadfence();
      state[0] = 0u;
      state[3] = 0u;
    }
  }
}

extern "C" int vnqa_grad_split_scale(const float* x, int64_t n, void* state, void* stream) {
  VNQA_CHECK_ARG(x && state && n > 0, "grad_split_scale: null pointer or n <= 0");
  VNQA_CHECK_ARG((((uintptr_t)x | (uintptr_t)state) & 15) == 0, "grad_split_scale: 16-byte alignment");
  const long long n4 = n / 4;
  long long blocks = (n4 + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
  hipLaunchKernelGGL(grad_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, (long long)n, (unsigned*)state);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_x3_post(const float* raw, const float* bias, const float* post_scale, const float* post_shift,
                            const float* border_sub, void* y, int32_t n_img, int32_t h, int32_t w, int32_t c_out, int32_t c_y,
                            int32_t y_halo, int32_t relu, int32_t pool2, int32_t out_x3, const float* raw_scale, void* stream) {
  VNQA_CHECK_ARG(raw && y, "x3_post: null pointer");
  const int zero_halo = (out_x3 & VNQA_X3_POST_ZERO_HALO) ? 1 : 0;
  out_x3 &= ~VNQA_X3_POST_ZERO_HALO;
  VNQA_CHECK_ARG(out_x3 >= 0 && out_x3 <= 2, "x3_post: out_x3 must be 0 (fp32), 1 (16-bit [hi | lo | hi]) or 2 (plain 16-bit)");
  VNQA_CHECK_ARG(n_img > 0 && h > 0 && w > 0 && c_out > 0 && c_out % 4 == 0 && c_y >= (out_x3 == 1 ? 3 : 1) * c_out && c_y % 4 == 0,
                 "x3_post: bad geometry n=%d h=%d w=%d c_out=%d c_y=%d", n_img, h, w, c_out, c_y);
  VNQA_CHECK_ARG(y_halo >= 0 && y_halo <= 2 && (relu == 0 || relu == 1), "x3_post: y_halo in 0..2, relu in 0..1");
  VNQA_CHECK_ARG(!pool2 || (h % 2 == 0 && w % 2 == 0), "x3_post: pool2 needs even h, w");
  VNQA_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "x3_post: post_scale / post_shift come together");
  VNQA_CHECK_ARG(border_sub == nullptr || (h >= 2 && w >= 2), "x3_post: border_sub needs h, w >= 2");
  PostArgs p;
  p.raw = raw; p.bias = bias; p.post_scale = post_scale; p.post_shift = post_shift; p.border_sub = border_sub;
  p.raw_scale = raw_scale;
  p.y = out_x3 ? nullptr : (float*)y;
  p.y16 = out_x3 ? (unsigned short*)y : nullptr;
  p.two = out_x3 == 2 ? 1 : 0;
  p.zero_halo = zero_halo;
  p.n = n_img; p.h = h; p.w = w; p.c_out = c_out; p.c_y = c_y; p.y_halo = y_halo; p.relu = relu; p.pool = pool2 ? 1 : 0;
  const long long total = (long long)n_img * (pool2 ? h / 2 : h) * (pool2 ? w / 2 : w) * (c_out / 4);
  hipLaunchKernelGGL(x3_post_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, p);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
