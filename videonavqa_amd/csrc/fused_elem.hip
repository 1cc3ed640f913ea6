// fused_elem.hip — the memory-bound glue of the FiLM trunk as fused kernels over padded-NHWC
// activations: per-frame train-mode BatchNorm (statistics + apply, forward and backward, with the
// preceding ReLU's mask folded into the backward), and FiLM affine + ReLU + residual (forward, and a
// backward that also reduces the per-(image,channel) gamma/beta gradients).
// Reference: models/film_attn_pt_stem.py:211 (bn_init(relu(conv_init))) and :229-241.
//
// Layout: [n_img][hp][wp][c], c a multiple of 64, zero halo.  A workgroup of 256 threads owns a
// 64-channel group: thread = (pixel lane 0..31, 8-channel chunk 0..7), 16-byte accesses, partial sums
// reduced through LDS across the 32 pixel lanes.  Every kernel (re)writes the halo as zero, so outputs
// need no memset.  HBM-bound: one read of each input and one write of each output.
#include "vnqa_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void load8(const T* p, float v[8]);
template <>
__device__ __forceinline__ void load8<vnqa_bf16>(const vnqa_bf16* p, float v[8]) {
  const uint4 u = *(const uint4*)p;
  v[0] = h16_lo(u.x); v[1] = h16_hi(u.x);
  v[2] = h16_lo(u.y); v[3] = h16_hi(u.y);
  v[4] = h16_lo(u.z); v[5] = h16_hi(u.z);
  v[6] = h16_lo(u.w); v[7] = h16_hi(u.w);
}
template <>
__device__ __forceinline__ void load8<float>(const float* p, float v[8]) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T>
__device__ __forceinline__ void store8(T* p, const float v[8]);
template <>
__device__ __forceinline__ void store8<vnqa_bf16>(vnqa_bf16* p, const float v[8]) {
  uint4 u;
  u.x = pack2_h16(v[0], v[1]);
  u.y = pack2_h16(v[2], v[3]);
  u.z = pack2_h16(v[4], v[5]);
  u.w = pack2_h16(v[6], v[7]);
  *(uint4*)p = u;
}
template <>
__device__ __forceinline__ void store8<float>(float* p, const float v[8]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// reduce 8 per-thread values over the 32 pixel lanes that share a channel chunk; result valid for prow==0
__device__ __forceinline__ void reduce_rows(float v[8], float* s_red /*[32][64]*/, int prow, int chunk) {
#pragma unroll
  for (int e = 0; e < 8; ++e) s_red[prow * 64 + chunk * 8 + e] = v[e];
  __syncthreads();
  if (prow == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = 0.f;
      for (int r = 0; r < 32; ++r) s += s_red[r * 64 + chunk * 8 + e];
      v[e] = s;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ bool is_interior(int pix, int hp, int wp) {
  const int y = pix / wp, x = pix - y * wp;
  return y >= 1 && y <= hp - 2 && x >= 1 && x <= wp - 2;
}

// ---- per-frame BN statistics: mean and biased variance of every (frame, channel) -------------
template <typename T>
__global__ void __launch_bounds__(256) bn_stats_kernel(const T* __restrict__ x, const T* __restrict__ x_lo, const int* __restrict__ frame_off,
                                                       float* __restrict__ mean, float* __restrict__ var, int hp, int wp,
                                                       int c) {
  __shared__ float s_red[32 * 64];
  __shared__ float s_mean[64];
  const int f = blockIdx.y, cg = blockIdx.x;
  const int prow = threadIdx.x >> 3, chunk = threadIdx.x & 7;
  const int i0 = frame_off[f], i1 = frame_off[f + 1];
  const long long P = (long long)(i1 - i0) * hp * wp;
  const float n = (float)((i1 - i0) * (hp - 2) * (wp - 2));
  const T* base = x + (size_t)i0 * hp * wp * c + cg * 64 + chunk * 8;
  // x_lo (optional): x is the hi half of a SPLIT tensor (a conv's VNQA_EPI_SPLIT_OUT output) and x + x_lo the unrounded value
  const T* base_lo = x_lo != nullptr ? x_lo + (size_t)i0 * hp * wp * c + cg * 64 + chunk * 8 : nullptr;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (long long p = prow; p < P; p += 32) {   // halo is zero: plain sum over every padded position (4 loads in flight)
    float v[8];
    load8<T>(base + (size_t)p * c, v);
    if (base_lo != nullptr) {
      float w[8];
      load8<T>(base_lo + (size_t)p * c, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += w[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += v[e];
  }
  reduce_rows(s, s_red, prow, chunk);
  if (prow == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) s_mean[chunk * 8 + e] = s[e] / n;
  }
  __syncthreads();
  float m[8], q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 8; ++e) m[e] = s_mean[chunk * 8 + e];
  const int hw = hp * wp;
#pragma unroll 4
  for (long long p = prow; p < P; p += 32) {   // second pass (L2-resident): centred sum of squares, interior only
    if (!is_interior((int)(p % hw), hp, wp)) continue;
    float v[8];
    load8<T>(base + (size_t)p * c, v);
    if (base_lo != nullptr) {
      float w[8];
      load8<T>(base_lo + (size_t)p * c, w);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += w[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = v[e] - m[e];
      q[e] += d * d;
    }
  }
  reduce_rows(q, s_red, prow, chunk);
  if (prow == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mean[(size_t)f * c + cg * 64 + chunk * 8 + e] = m[e];
      var[(size_t)f * c + cg * 64 + chunk * 8 + e] = q[e] / n;
    }
  }
}

// y = (x - mean[f]) * rstd[f] * gamma + beta on the interior, 0 on the halo
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ x_lo, const int* __restrict__ frame_of,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ y, int hp, int wp, int c) {
  const int n = blockIdx.y, cg = blockIdx.x;
  const int prow = threadIdx.x >> 3, chunk = threadIdx.x & 7;
  const int f = frame_of[n];
  const int c0 = cg * 64 + chunk * 8;
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = rstd[(size_t)f * c + c0 + e] * gamma[c0 + e];
    sh[e] = beta[c0 + e] - mean[(size_t)f * c + c0 + e] * sc[e];
  }
  const size_t base = (size_t)n * hp * wp * c + c0;
  for (int p = prow; p < hp * wp; p += 32) {
    float v[8];
    if (is_interior(p, hp, wp)) {
      load8<T>(x + base + (size_t)p * c, v);
      if (x_lo != nullptr) {
        float w[8];
        load8<T>(x_lo + base + (size_t)p * c, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += w[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
    }
    store8<T>(y + base + (size_t)p * c, v);
  }
}

// s1[f][c] = sum dy, s2[f][c] = sum dy * xhat   (dy's halo is zero)
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const int* __restrict__ frame_off, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ s1,
                                                           float* __restrict__ s2, int hp, int wp, int c) {
  __shared__ float s_red[32 * 64];
  const int f = blockIdx.y, cg = blockIdx.x;
  const int prow = threadIdx.x >> 3, chunk = threadIdx.x & 7;
  const int i0 = frame_off[f], i1 = frame_off[f + 1];
  const long long P = (long long)(i1 - i0) * hp * wp;
  const int c0 = cg * 64 + chunk * 8;
  const size_t base = (size_t)i0 * hp * wp * c + c0;
  float m[8], r[8], a[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    m[e] = mean[(size_t)f * c + c0 + e];
    r[e] = rstd[(size_t)f * c + c0 + e];
  }
  // four rows per trip: eight 16-byte loads in flight per thread (one workgroup per CU walks 2048 positions of its frame —
  // with one row per trip the loop ran at load latency: 49 us for 112 MB)
  long long p = prow;
  for (; p + 96 < P; p += 128) {
    float g[4][8], v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      load8<T>(dy + base + (size_t)(p + 32 * u) * c, g[u]);
      load8<T>(x + base + (size_t)(p + 32 * u) * c, v[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] += g[u][e];
        b[e] += g[u][e] * (v[u][e] - m[e]) * r[e];
      }
  }
  for (; p < P; p += 32) {
    float g[8], v[8];
    load8<T>(dy + base + (size_t)p * c, g);
    load8<T>(x + base + (size_t)p * c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      a[e] += g[e];
      b[e] += g[e] * (v[e] - m[e]) * r[e];
    }
  }
  reduce_rows(a, s_red, prow, chunk);
  reduce_rows(b, s_red, prow, chunk);
  if (prow == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[(size_t)f * c + c0 + e] = a[e];
      s2[(size_t)f * c + c0 + e] = b[e];
    }
  }
}

// dx = gamma*rstd*(dy - s1/n - xhat*s2/n) [* (x > 0) when the BN input is a ReLU output]
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const int* __restrict__ frame_of, const int* __restrict__ frame_off,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ s1,
                                                           const float* __restrict__ s2, T* __restrict__ dx, int hp, int wp,
                                                           int c, int relu_mask) {
  const int n = blockIdx.y, cg = blockIdx.x;
  const int prow = threadIdx.x >> 3, chunk = threadIdx.x & 7;
  const int f = frame_of[n];
  const float inv_n = 1.f / (float)((frame_off[f + 1] - frame_off[f]) * (hp - 2) * (wp - 2));
  const int c0 = cg * 64 + chunk * 8;
  float m[8], r[8], gr[8], a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    m[e] = mean[(size_t)f * c + c0 + e];
    r[e] = rstd[(size_t)f * c + c0 + e];
    gr[e] = gamma[c0 + e] * r[e];
    a[e] = s1[(size_t)f * c + c0 + e] * inv_n;
    b[e] = s2[(size_t)f * c + c0 + e] * inv_n;
  }
  const size_t base = (size_t)n * hp * wp * c + c0;
  for (int p = prow; p < hp * wp; p += 32) {
    float o[8];
    if (is_interior(p, hp, wp)) {
      float g[8], v[8];
      load8<T>(dy + base + (size_t)p * c, g);
      load8<T>(x + base + (size_t)p * c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] - m[e]) * r[e];
        o[e] = gr[e] * (g[e] - a[e] - xh * b[e]);
        if (relu_mask && !(v[e] > 0.f)) o[e] = 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
    }
    store8<T>(dx + base + (size_t)p * c, o);
  }
}

// out = relu(gamma[n] * z + beta[n]) + res on the interior, 0 on the halo
template <typename T>
__global__ void __launch_bounds__(256) film_fwd_kernel(const T* __restrict__ z, const T* __restrict__ res,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ out, int hp, int wp, int c, int ld, int film_c) {
  const int n = blockIdx.y, cg = blockIdx.x;
  const int prow = threadIdx.x >> 3, chunk = threadIdx.x & 7;
  const int c0 = cg * 64 + chunk * 8;
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool ok = c0 + e < film_c;
    ga[e] = ok ? gamma[(size_t)n * ld + c0 + e] : 0.f;
    be[e] = ok ? beta[(size_t)n * ld + c0 + e] : 0.f;
  }
  const size_t base = (size_t)n * hp * wp * c + c0;
  for (int p = prow; p < hp * wp; p += 32) {
    float o[8];
    if (is_interior(p, hp, wp)) {
      float v[8], r[8];
      load8<T>(z + base + (size_t)p * c, v);
      load8<T>(res + base + (size_t)p * c, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = fmaxf(ga[e] * v[e] + be[e], 0.f) + r[e];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
    }
    store8<T>(out + base + (size_t)p * c, o);
  }
}

// dz = dout * [gamma z + beta > 0] * gamma ; dgamma[n][c] = sum_pix dout*mask*z ; dbeta[n][c] = sum_pix dout*mask
template <typename T>
__global__ void __launch_bounds__(256) film_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ z,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       T* __restrict__ dz, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       int hp, int wp, int c, int ld, int film_c, int ld_out) {
  __shared__ float s_red[32 * 64];
  const int n = blockIdx.y, cg = blockIdx.x;
  const int prow = threadIdx.x >> 3, chunk = threadIdx.x & 7;
  const int c0 = cg * 64 + chunk * 8;
  float ga[8], be[8], sg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bool ok = c0 + e < film_c;
    ga[e] = ok ? gamma[(size_t)n * ld + c0 + e] : 0.f;
    be[e] = ok ? beta[(size_t)n * ld + c0 + e] : 0.f;
  }
  const size_t base = (size_t)n * hp * wp * c + c0;
  for (int p = prow; p < hp * wp; p += 32) {
    float o[8];
    if (is_interior(p, hp, wp)) {
      float g[8], v[8];
      load8<T>(dout + base + (size_t)p * c, g);
      load8<T>(z + base + (size_t)p * c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float du = (ga[e] * v[e] + be[e] > 0.f) ? g[e] : 0.f;
        o[e] = du * ga[e];
        sg[e] += du * v[e];
        sb[e] += du;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = 0.f;
    }
    store8<T>(dz + base + (size_t)p * c, o);
  }
  reduce_rows(sg, s_red, prow, chunk);
  reduce_rows(sb, s_red, prow, chunk);
  if (prow == 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (c0 + e < film_c) {
        dgamma[(size_t)n * ld_out + c0 + e] = sg[e];
        dbeta[(size_t)n * ld_out + c0 + e] = sb[e];
      }
    }
  }
}

// g = (a [+ b]) * [y > 0]   — ReLU backward, optionally summing two gradient streams first
template <typename T>
__global__ void __launch_bounds__(256) relu_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                       const T* __restrict__ y, T* __restrict__ g, size_t n8) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float va[8], vy[8];
    load8<T>(a + i * 8, va);
    load8<T>(y + i * 8, vy);
    if (b != nullptr) {
      float vb[8];
      load8<T>(b + i * 8, vb);
#pragma unroll
      for (int e = 0; e < 8; ++e) va[e] += vb[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) va[e] = vy[e] > 0.f ? va[e] : 0.f;
    store8<T>(g + i * 8, va);
  }
}

}  // namespace

#define VNQA_ELEM_DISPATCH(dtype, CALL_BF16, CALL_F32)                          \
  if ((dtype) == VNQA_BF16) { CALL_BF16; }                                      \
  else if ((dtype) == VNQA_F32) { CALL_F32; }                                   \
  else { vnqa_set_error("bad dtype %d", (int)(dtype)); return VNQA_ERR_INVALID_ARG; }

extern "C" int vnqa_frame_bn_stats(const void* x, const int32_t* frame_off, float* mean, float* var,
                                   int32_t n_frames, int32_t hp, int32_t wp, int32_t c, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(x && frame_off && mean && var && n_frames > 0 && c % 64 == 0, "frame_bn_stats: bad arguments");
  dim3 grid(c / 64, n_frames);
  hipStream_t st = (hipStream_t)stream;
  VNQA_ELEM_DISPATCH(dtype,
      hipLaunchKernelGGL(bn_stats_kernel<vnqa_bf16>, grid, dim3(256), 0, st, (const vnqa_bf16*)x, (const vnqa_bf16*)nullptr, frame_off, mean, var, hp, wp, c),
      hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)nullptr, frame_off, mean, var, hp, wp, c));
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

// ... of a SPLIT tensor (16-bit format): the statistics of x_hi + x_lo, the unrounded output of a conv launched with VNQA_EPI_SPLIT_OUT
extern "C" int vnqa_frame_bn_stats_split(const void* x_hi, const void* x_lo, const int32_t* frame_off, float* mean, float* var,
                                         int32_t n_frames, int32_t hp, int32_t wp, int32_t c, void* stream) {
  VNQA_CHECK_ARG(x_hi && x_lo && frame_off && mean && var && n_frames > 0 && c % 64 == 0, "frame_bn_stats_split: bad arguments");
  hipLaunchKernelGGL(bn_stats_kernel<vnqa_bf16>, dim3(c / 64, n_frames), dim3(256), 0, (hipStream_t)stream, (const vnqa_bf16*)x_hi,
                     (const vnqa_bf16*)x_lo, frame_off, mean, var, hp, wp, c);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_frame_bn_apply_split(const void* x_hi, const void* x_lo, const int32_t* frame_of, const float* mean, const float* rstd,
                                         const float* gamma, const float* beta, void* y, int32_t n_img, int32_t hp, int32_t wp,
                                         int32_t c, void* stream) {
  VNQA_CHECK_ARG(x_hi && x_lo && frame_of && mean && rstd && gamma && beta && y && n_img > 0 && c % 64 == 0, "frame_bn_apply_split: bad arguments");
  hipLaunchKernelGGL(bn_apply_kernel<vnqa_bf16>, dim3(c / 64, n_img), dim3(256), 0, (hipStream_t)stream, (const vnqa_bf16*)x_hi,
                     (const vnqa_bf16*)x_lo, frame_of, mean, rstd, gamma, beta, (vnqa_bf16*)y, hp, wp, c);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_frame_bn_apply(const void* x, const int32_t* frame_of, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, void* y, int32_t n_img, int32_t hp,
                                   int32_t wp, int32_t c, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(x && frame_of && mean && rstd && gamma && beta && y && n_img > 0 && c % 64 == 0, "frame_bn_apply: bad arguments");
  dim3 grid(c / 64, n_img);
  hipStream_t st = (hipStream_t)stream;
  VNQA_ELEM_DISPATCH(dtype,
      hipLaunchKernelGGL(bn_apply_kernel<vnqa_bf16>, grid, dim3(256), 0, st, (const vnqa_bf16*)x, (const vnqa_bf16*)nullptr, frame_of, mean, rstd, gamma, beta, (vnqa_bf16*)y, hp, wp, c),
      hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)nullptr, frame_of, mean, rstd, gamma, beta, (float*)y, hp, wp, c));
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_frame_bn_bwd(const void* dy, const void* x, const int32_t* frame_of, const int32_t* frame_off,
                                 const float* mean, const float* rstd, const float* gamma, float* s1, float* s2,
                                 void* dx, int32_t n_img, int32_t n_frames, int32_t hp, int32_t wp, int32_t c,
                                 int32_t relu_mask, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(dy && x && frame_of && frame_off && mean && rstd && gamma && s1 && s2 && dx && c % 64 == 0, "frame_bn_bwd: bad arguments");
  dim3 g1(c / 64, n_frames), g2(c / 64, n_img);
  hipStream_t st = (hipStream_t)stream;
  VNQA_ELEM_DISPATCH(dtype,
      hipLaunchKernelGGL(bn_bwd_stats_kernel<vnqa_bf16>, g1, dim3(256), 0, st, (const vnqa_bf16*)dy, (const vnqa_bf16*)x, frame_off, mean, rstd, s1, s2, hp, wp, c),
      hipLaunchKernelGGL(bn_bwd_stats_kernel<float>, g1, dim3(256), 0, st, (const float*)dy, (const float*)x, frame_off, mean, rstd, s1, s2, hp, wp, c));
  VNQA_CHECK_LAUNCH();
  VNQA_ELEM_DISPATCH(dtype,
      hipLaunchKernelGGL(bn_bwd_apply_kernel<vnqa_bf16>, g2, dim3(256), 0, st, (const vnqa_bf16*)dy, (const vnqa_bf16*)x, frame_of, frame_off, mean, rstd, gamma, s1, s2, (vnqa_bf16*)dx, hp, wp, c, relu_mask),
      hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, g2, dim3(256), 0, st, (const float*)dy, (const float*)x, frame_of, frame_off, mean, rstd, gamma, s1, s2, (float*)dx, hp, wp, c, relu_mask));
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_film_relu_res_fwd_ld(const void* z, const void* res, const float* gamma, const float* beta, void* out,
                                         int32_t n_img, int32_t hp, int32_t wp, int32_t c, int32_t film_ld, int32_t film_c,
                                         int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(z && res && gamma && beta && out && n_img > 0 && c % 64 == 0 && film_ld > 0 && film_c > 0 && film_c <= c,
                 "film_relu_res_fwd: bad arguments");
  dim3 grid(c / 64, n_img);
  hipStream_t st = (hipStream_t)stream;
  VNQA_ELEM_DISPATCH(dtype,
      hipLaunchKernelGGL(film_fwd_kernel<vnqa_bf16>, grid, dim3(256), 0, st, (const vnqa_bf16*)z, (const vnqa_bf16*)res, gamma, beta, (vnqa_bf16*)out, hp, wp, c, film_ld, film_c),
      hipLaunchKernelGGL(film_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)z, (const float*)res, gamma, beta, (float*)out, hp, wp, c, film_ld, film_c));
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_film_relu_res_fwd(const void* z, const void* res, const float* gamma, const float* beta, void* out,
                                      int32_t n_img, int32_t hp, int32_t wp, int32_t c, int32_t dtype, void* stream) {
  return vnqa_film_relu_res_fwd_ld(z, res, gamma, beta, out, n_img, hp, wp, c, c, c, dtype, stream);
}

extern "C" int vnqa_film_relu_res_bwd_ld(const void* dout, const void* z, const float* gamma, const float* beta, void* dz,
                                         float* dgamma, float* dbeta, int32_t n_img, int32_t hp, int32_t wp, int32_t c,
                                         int32_t film_ld, int32_t film_c, int32_t grad_ld, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(dout && z && gamma && beta && dz && dgamma && dbeta && n_img > 0 && c % 64 == 0 && film_ld > 0 &&
                     grad_ld > 0 && film_c > 0 && film_c <= c, "film_relu_res_bwd: bad arguments");
  dim3 grid(c / 64, n_img);
  hipStream_t st = (hipStream_t)stream;
  VNQA_ELEM_DISPATCH(dtype,
      hipLaunchKernelGGL(film_bwd_kernel<vnqa_bf16>, grid, dim3(256), 0, st, (const vnqa_bf16*)dout, (const vnqa_bf16*)z, gamma, beta, (vnqa_bf16*)dz, dgamma, dbeta, hp, wp, c, film_ld, film_c, grad_ld),
      hipLaunchKernelGGL(film_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, (const float*)z, gamma, beta, (float*)dz, dgamma, dbeta, hp, wp, c, film_ld, film_c, grad_ld));
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_film_relu_res_bwd(const void* dout, const void* z, const float* gamma, const float* beta, void* dz,
                                      float* dgamma, float* dbeta, int32_t n_img, int32_t hp, int32_t wp, int32_t c,
                                      int32_t dtype, void* stream) {
  return vnqa_film_relu_res_bwd_ld(dout, z, gamma, beta, dz, dgamma, dbeta, n_img, hp, wp, c, c, c, c, dtype, stream);
}

extern "C" int vnqa_relu_bwd(const void* a, const void* b, const void* y, void* g, int64_t n, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(a && y && g && n > 0 && n % 8 == 0, "relu_bwd: bad arguments (n must be a multiple of 8)");
  const size_t n8 = (size_t)n / 8;
  size_t blocks = (n8 + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipStream_t st = (hipStream_t)stream;
  VNQA_ELEM_DISPATCH(dtype,
      hipLaunchKernelGGL(relu_bwd_kernel<vnqa_bf16>, dim3((int)blocks), dim3(256), 0, st, (const vnqa_bf16*)a, (const vnqa_bf16*)b, (const vnqa_bf16*)y, (vnqa_bf16*)g, n8),
      hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3((int)blocks), dim3(256), 0, st, (const float*)a, (const float*)b, (const float*)y, (float*)g, n8));
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
