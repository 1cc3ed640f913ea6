// cnn3d.hip — VideoOnlyCNN3D (BASELINE config 2; /root/reference models/v_only_cnn3d.py:13-37,59-81) off the generic-kernel
// bring-up path: the layers the igemm / wgrad kernels serve badly or not at all.
//
//   * conv1 (3 -> 64 channels, 3x3x3) forward: bn_input's normalisation applied while the patch is loaded from the fp32
//     NCDHW clip, implicit GEMM with K = 27 taps x 4 channels (x^_0, x^_1, x^_2, 1) = 108 -> 128 on v_mfma_f32_16x16x32,
//     bias + ReLU + MaxPool3d(1,2,2) + arg-max + bn1's batch statistics in the epilogue.  The constant-1 channel (0 in the
//     zero padding, like the others) carries bn_input's shift, so the padded border is exact: W'[co][tap][c] = W gamma_c for
//     c < 3 and sum_c W beta_c for c = 3.
//   * conv1 backward: ONE split-K GEMM  G[co][tap][c] = sum_px dY[px][co] A[px + tap][c]  over the same 4-channel patch, dY
//     formed on the fly from the pooled gradient and the arg-max.  Everything follows from G: dW = gamma_c G_c + beta_c G_3,
//     db = G_3 at the centre tap, and — because sum_px dX[px][c] X^[px][c] = sum_{co,tap} W[co][c][tap] G[co][tap][c] —
//     bn_input's d gamma / d beta WITHOUT a dgrad pass to the 3-channel input.
//   * BatchNorm (train mode) over channel-last data: deterministic two-stage statistics, apply into the next conv's padded
//     NDHWC input (or the NC-flattened fp32 feature vector), backward reduce / apply; MaxPool3d(4,4,4) forward (+ arg-max +
//     the following BN's statistics) and backward (writes the conv's whole padded dY, ReLU mask included).
// The 64 -> 128 and 128 -> 128 convs stay on the 27-tap igemm (forward, dgrad) and the small-channel form of the wgrad kernel.
#include "vnqa_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short c3_s16x4;

__device__ __forceinline__ unsigned c3_lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// transposed LDS read: the 16 lanes of a group load a 4-row x 16-column block of 16-bit elements (lane (q, pp): the 8-byte
// chunk pp of row q, ANY address) and receive its transpose — lane il gets rows 0..3 of column il
__device__ __forceinline__ c3_s16x4 c3_tr_read(unsigned addr) {
  c3_s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}

// ---- statistics ----------------------------------------------------------------------------------------------------------
// partial[blk][C][2] = (sum, sum of squares) of block blk's share; bn_finalize folds the blocks in order, in double
__global__ void __launch_bounds__(256) stats_ncdhw_kernel(const float* __restrict__ x, float* __restrict__ partial, int C,
                                                          long long S, int chunks) {
  __shared__ float s_red[2][4];
  const int c = blockIdx.y, n = blockIdx.x / chunks, chunk = blockIdx.x - n * chunks;
  const long long len = (S + chunks - 1) / chunks, b = chunk * len;
  long long e = b + len;
  e = e < S ? e : S;
  const float* src = x + ((size_t)n * C + c) * S;
  float s = 0.f, q = 0.f;
  for (long long i = b + threadIdx.x; i < e; i += 256) {
    const float v = src[i];
    s += v;
    q = fmaf(v, v, q);
  }
  s = wave_reduce_sum(s);
  q = wave_reduce_sum(q);
  if ((threadIdx.x & 63) == 0) {
    s_red[0][threadIdx.x >> 6] = s;
    s_red[1][threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* o = partial + ((size_t)blockIdx.x * C + c) * 2;
    o[0] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
    o[1] = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
  }
}

// dense [R][C] rows (fp32 or the 16-bit format): block = 64 channels x 16 row lanes over a row range
template <typename T>
__global__ void __launch_bounds__(1024) stats_rows_kernel(const T* __restrict__ x, float* __restrict__ partial, long long R, int C,
                                                          long long rows_per_block) {
  __shared__ float s_part[2][16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  r1 = r1 < R ? r1 : R;
  float s = 0.f, q = 0.f;
  if (c < C)
    for (long long r = r0 + ry; r < r1; r += 16) {
      const float v = ElemOps<T>::load(x[(size_t)r * C + c]);
      s += v;
      q = fmaf(v, v, q);
    }
  s_part[0][ry][cx] = s;
  s_part[1][ry][cx] = q;
  __syncthreads();
  if (ry == 0 && c < C) {
    float ts = 0.f, tq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ts += s_part[0][r][cx];
      tq += s_part[1][r][cx];
    }
    float* o = partial + ((size_t)blockIdx.y * C + c) * 2;
    o[0] = ts;
    o[1] = tq;
  }
}

// one workgroup per channel: 256 threads fold the blocks' partials in double (fixed order per thread, fixed tree)
__device__ __forceinline__ void fold_partials(const float* __restrict__ partial, int nblk, int C, int c, double& s, double& q) {
  __shared__ double s_d[2][256];
  double ls = 0.0, lq = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) {
    ls += (double)partial[((size_t)b * C + c) * 2];
    lq += (double)partial[((size_t)b * C + c) * 2 + 1];
  }
  s_d[0][threadIdx.x] = ls;
  s_d[1][threadIdx.x] = lq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      s_d[0][threadIdx.x] += s_d[0][threadIdx.x + o];
      s_d[1][threadIdx.x] += s_d[1][threadIdx.x + o];
    }
    __syncthreads();
  }
  s = s_d[0][0];
  q = s_d[1][0];
}

__global__ void __launch_bounds__(256) bn_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count, float eps,
                                                          float momentum, float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ rmean, float* __restrict__ rvar) {
  const int c = blockIdx.x;
  double s, q;
  fold_partials(partial, nblk, C, c, s, q);
  if (threadIdx.x != 0) return;
  const double m = s / count;
  double var = q / count - m * m;
  var = var > 0.0 ? var : 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean != nullptr) {                                  // nn.BatchNorm: running_var takes the UNBIASED batch variance
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    rmean[c] = (float)((1.0 - momentum) * rmean[c] + momentum * m);
    rvar[c] = (float)((1.0 - momentum) * rvar[c] + momentum * unb);
  }
}

// (sum dy, sum dy x^) partials -> d gamma, d beta and the two means the apply kernel needs
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, double count,
                                                              float inv_scale, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ m_dy, float* __restrict__ m_dyx) {
  const int c = blockIdx.x;
  double s, q;
  fold_partials(partial, nblk, C, c, s, q);
  if (threadIdx.x != 0) return;
  dbeta[c] = (float)(s * inv_scale);
  dgamma[c] = (float)(q * inv_scale);
  m_dy[c] = (float)(s / count);
  m_dyx[c] = (float)(q / count);
}

// ---- index helper: element (n, d, h, w, c) of a tensor given by element strides (dense, padded NDHWC, or NC-flattened) ------
struct View5 {
  long long base, sn, sd, sh, sw, sc;
  int D, H, W;        // logical extents of the row index r = ((n D + d) H + h) W + w
};
__device__ __forceinline__ long long view_row(const View5& v, long long r) {
  const int w = (int)(r % v.W);
  long long t = r / v.W;
  const int h = (int)(t % v.H);
  t /= v.H;
  const int d = (int)(t % v.D);
  const long long n = t / v.D;
  return v.base + n * v.sn + d * v.sd + h * v.sh + w * v.sw;
}

// y = gamma (x - mean) rstd + beta: x dense [R][C] -> out through `ov`
template <typename TIN, typename TOUT>
__global__ void __launch_bounds__(256) bn_apply_kernel(const TIN* __restrict__ x, TOUT* __restrict__ out, View5 ov,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       long long R, int C) {
  const int cpr = C >> 2;                                   // 4 channels per thread
  const long long total = R * cpr;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cpr;
    const int c = (int)(i - r * cpr) * 4;
    const long long o = view_row(ov, r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float v = ElemOps<TIN>::load(x[(size_t)r * C + c + e]);
      out[o + (long long)(c + e) * ov.sc] = ElemOps<TOUT>::store(fmaf((v - mean[c + e]) * rstd[c + e], gamma[c + e], beta[c + e]));
    }
  }
}

// partial (sum dy, sum dy x^) over a row range; dy through `dv`, x dense [R][C]
template <typename TX, typename TDY>
__global__ void __launch_bounds__(1024) bn_bwd_reduce_kernel(const TDY* __restrict__ dy, View5 dv, const TX* __restrict__ x,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             float* __restrict__ partial, long long R, int C,
                                                             long long rows_per_block) {
  __shared__ float s_part[2][16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cx;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  r1 = r1 < R ? r1 : R;
  float s = 0.f, q = 0.f;
  if (c < C) {
    const float m = mean[c], rs = rstd[c];
    for (long long r = r0 + ry; r < r1; r += 16) {
      const float g = ElemOps<TDY>::load(dy[view_row(dv, r) + (long long)c * dv.sc]);
      const float xh = (ElemOps<TX>::load(x[(size_t)r * C + c]) - m) * rs;
      s += g;
      q = fmaf(g, xh, q);
    }
  }
  s_part[0][ry][cx] = s;
  s_part[1][ry][cx] = q;
  __syncthreads();
  if (ry == 0 && c < C) {
    float ts = 0.f, tq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ts += s_part[0][r][cx];
      tq += s_part[1][r][cx];
    }
    float* o = partial + ((size_t)blockIdx.y * C + c) * 2;
    o[0] = ts;
    o[1] = tq;
  }
}

// the same reduction for 16-bit dy and x with 16-byte loads: thread = (row lane, 8-channel chunk); C / 8 chunks x 256 / (C / 8)
// row lanes per block (C = 64: 32 lanes, C = 128: 16) — the scalar form above moves 2 bytes per lane and load (2.3 TB/s at
// config 2's 0.4 GB passes)
__global__ void __launch_bounds__(256) bn_bwd_reduce_h16x8_kernel(const vnqa_bf16* __restrict__ dy, View5 dv,
                                                                  const vnqa_bf16* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, float* __restrict__ partial,
                                                                  long long R, int C, long long rows_per_block) {
  extern __shared__ float s_red[];                        // [2][lanes][C]
  const int cpr = C >> 3, lanes = 256 / cpr;
  const int chunk = threadIdx.x % cpr, ry = threadIdx.x / cpr;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  long long r1 = r0 + rows_per_block;
  r1 = r1 < R ? r1 : R;
  float m[8], rs[8], s[8], q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { m[e] = mean[chunk * 8 + e]; rs[e] = rstd[chunk * 8 + e]; s[e] = 0.f; q[e] = 0.f; }
  for (long long r = r0 + ry; r < r1; r += lanes) {
    const uint4 g4 = *(const uint4*)(dy + view_row(dv, r) + chunk * 8);
    const uint4 x4 = *(const uint4*)(x + (size_t)r * C + chunk * 8);
    const unsigned gw[4] = {g4.x, g4.y, g4.z, g4.w}, xw[4] = {x4.x, x4.y, x4.z, x4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float g0 = h16_lo(gw[e]), g1 = h16_hi(gw[e]);
      s[2 * e] += g0;
      s[2 * e + 1] += g1;
      q[2 * e] = fmaf(g0, (h16_lo(xw[e]) - m[2 * e]) * rs[2 * e], q[2 * e]);
      q[2 * e + 1] = fmaf(g1, (h16_hi(xw[e]) - m[2 * e + 1]) * rs[2 * e + 1], q[2 * e + 1]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    s_red[ry * C + chunk * 8 + e] = s[e];
    s_red[(lanes + ry) * C + chunk * 8 + e] = q[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float ts = 0.f, tq = 0.f;
    for (int r = 0; r < lanes; ++r) {
      ts += s_red[r * C + c];
      tq += s_red[(lanes + r) * C + c];
    }
    partial[((size_t)blockIdx.x * C + c) * 2] = ts;
    partial[((size_t)blockIdx.x * C + c) * 2 + 1] = tq;
  }
}

// dx = gamma rstd (dy - mean(dy) - x^ mean(dy x^)) -> dense [R][C]
template <typename TX, typename TDY, typename TOUT>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const TDY* __restrict__ dy, View5 dv, const TX* __restrict__ x,
                                                           TOUT* __restrict__ dx, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ m_dy, const float* __restrict__ m_dyx,
                                                           long long R, int C) {
  const int cpr = C >> 2;
  const long long total = R * cpr;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cpr;
    const int c = (int)(i - r * cpr) * 4;
    const long long o = view_row(dv, r);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ce = c + e;
      const float g = ElemOps<TDY>::load(dy[o + (long long)ce * dv.sc]);
      const float xh = (ElemOps<TX>::load(x[(size_t)r * C + ce]) - mean[ce]) * rstd[ce];
      dx[(size_t)r * C + ce] = ElemOps<TOUT>::store(gamma[ce] * rstd[ce] * (g - m_dy[ce] - xh * m_dyx[ce]));
    }
  }
}

// ---- MaxPool3d(4,4,4) over a padded NDHWC conv output (already ReLU'ed) ------------------------------------------------------
// thread = (window, 8-channel chunk).  p dense [N][Do][Ho][Wo][C]; idx = position of the first maximum in (d, h, w) scan order
// (0..63), 255 where the maximum is not positive (the ReLU passes no gradient there).  Also the partial BN statistics of p.
__global__ void __launch_bounds__(256) pool444_fwd_kernel(const vnqa_bf16* __restrict__ y, vnqa_bf16* __restrict__ p,
                                                          unsigned char* __restrict__ idx, float* __restrict__ partial, int N,
                                                          int D, int H, int W, int C) {
  extern __shared__ float s_stat[];                       // [2][C]
  const int Do = D / 4, Ho = H / 4, Wo = W / 4, cc = C >> 3;
  const long long total = (long long)N * Do * Ho * Wo * cc;
  for (int i = threadIdx.x; i < 2 * C; i += 256) s_stat[i] = 0.f;
  __syncthreads();
  const long long sH = (long long)(W + 2) * C, sD = (long long)(H + 2) * sH, sN = (long long)(D + 2) * sD;
  float ssum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ssq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int my_c = -1;
  for (long long it = blockIdx.x * 256ll + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int c8 = (int)(it % cc);
    long long t = it / cc;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho); t /= Ho;
    const int dz = (int)(t % Do);
    const long long n = t / Do;
    const vnqa_bf16* src = y + n * sN + (long long)(4 * dz + 1) * sD + (long long)(4 * ho + 1) * sH + (long long)(4 * wo + 1) * C + c8 * 8;
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll 4
    for (int k = 0; k < 64; ++k) {
      const int kd = k >> 4, kh = (k >> 2) & 3, kw = k & 3;
      const uint4 u = *(const uint4*)(src + kd * sD + kh * sH + kw * C);
      const float v[8] = {h16_lo(u.x), h16_hi(u.x), h16_lo(u.y), h16_hi(u.y), h16_lo(u.z), h16_hi(u.z), h16_lo(u.w), h16_hi(u.w)};
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (v[e] > best[e]) { best[e] = v[e]; bi[e] = k; }
    }
    const size_t o = (size_t)(it / cc) * C + c8 * 8;
    uint4 pv;
    pv.x = pack2_h16(best[0], best[1]); pv.y = pack2_h16(best[2], best[3]);
    pv.z = pack2_h16(best[4], best[5]); pv.w = pack2_h16(best[6], best[7]);
    *(uint4*)(p + o) = pv;
    unsigned long long iv = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) iv |= (unsigned long long)(best[e] > 0.f ? bi[e] : 255) << (8 * e);
    *(unsigned long long*)(idx + o) = iv;
    if (my_c < 0) my_c = c8;
    if (my_c == c8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { ssum[e] += best[e]; ssq[e] = fmaf(best[e], best[e], ssq[e]); }
    } else {                                               // (grid stride not a multiple of the chunk count: rare path)
#pragma unroll
      for (int e = 0; e < 8; ++e) { atomicAdd(&s_stat[c8 * 8 + e], best[e]); atomicAdd(&s_stat[C + c8 * 8 + e], best[e] * best[e]); }
    }
  }
  if (my_c >= 0) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(&s_stat[my_c * 8 + e], ssum[e]); atomicAdd(&s_stat[C + my_c * 8 + e], ssq[e]); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    partial[((size_t)blockIdx.x * C + i) * 2] = s_stat[i];
    partial[((size_t)blockIdx.x * C + i) * 2 + 1] = s_stat[C + i];
  }
}

// dy[n][d][h][w][c] (padded NDHWC interior) = dp of the window that (d,h,w) is the arg-max of, else 0; interior positions no
// window covers (D, H or W not a multiple of 4) are zeroed too.  The halo is never written (stays zero).
__global__ void __launch_bounds__(256) pool444_bwd_kernel(const vnqa_bf16* __restrict__ dp, const unsigned char* __restrict__ idx,
                                                          vnqa_bf16* __restrict__ dy, int N, int D, int H, int W, int C) {
  const int Do = D / 4, Ho = H / 4, Wo = W / 4, cc = C >> 3;
  const long long total = (long long)N * D * H * W * cc;
  const long long sH = (long long)(W + 2) * C, sD = (long long)(H + 2) * sH, sN = (long long)(D + 2) * sD;
  for (long long it = blockIdx.x * 256ll + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int c8 = (int)(it % cc);
    long long t = it / cc;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int d = (int)(t % D);
    const long long n = t / D;
    uint4 out = {0u, 0u, 0u, 0u};
    const int dz = d >> 2, ho = h >> 2, wo = w >> 2;
    if (dz < Do && ho < Ho && wo < Wo) {
      const size_t o = ((((size_t)n * Do + dz) * Ho + ho) * Wo + wo) * C + c8 * 8;
      const unsigned long long iv = *(const unsigned long long*)(idx + o);
      const uint4 g = *(const uint4*)(dp + o);
      const unsigned me = (unsigned)(((d & 3) << 4) | ((h & 3) << 2) | (w & 3));
      const unsigned gw[4] = {g.x, g.y, g.z, g.w};
      unsigned ow[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned lo = ((unsigned)(iv >> (16 * j)) & 0xffu) == me ? (gw[j] & 0xffffu) : 0u;
        const unsigned hi = ((unsigned)(iv >> (16 * j + 8)) & 0xffu) == me ? (gw[j] & 0xffff0000u) : 0u;
        ow[j] = lo | hi;
      }
      out.x = ow[0]; out.y = ow[1]; out.z = ow[2]; out.w = ow[3];
    }
    *(uint4*)(dy + n * sN + (long long)(d + 1) * sD + (long long)(h + 1) * sH + (long long)(w + 1) * C + c8 * 8) = out;
  }
}

// ---- conv1: 3 -> 64 channels ------------------------------------------------------------------------------------------------
constexpr int C1_CO = 64;
constexpr int C1_K = 128;                                   // 27 taps x 4 channels (x^0, x^1, x^2, 1) = 108, padded
constexpr int C1_T = 16;                                    // output tile: 16 x 16 pixels of one depth plane
constexpr int C1_PW = C1_T + 2;                             // patch width / height
constexpr int C1_PLANE = C1_PW * C1_PW * 8;                 // bytes of one patch plane: 4 x 16-bit channels per pixel

struct C1Args {
  const float* x;            // [N][3][D][H][W]
  const float* w;            // [64][3][27]
  const float* bias;         // [64]
  const float *mean, *rstd, *gamma, *beta;   // bn_input (batch statistics / affine)
  vnqa_bf16* p;              // [N][D][H/2][W/2][64]
  unsigned char* idx;        // same shape: arg-max position 0..3 in (h, w) order, +4 when the maximum is not positive
  float* partial;            // forward: [grid][64][2] statistics of p;  backward: [grid][64][112] partial G
  const vnqa_bf16* dp;       // backward: gradient wrt p
  int N, D, H, W;
};

// one plane of the patch: global fp32 -> normalised 4-channel pixels in registers (thread t: pixels t and t + 256)
struct C1Px { unsigned lo, hi; };
__device__ __forceinline__ C1Px c1_load_px(const C1Args& a, long long n, int d, int gy, int gx, const float (&mu)[3],
                                           const float (&rs)[3]) {
  C1Px r = {0u, 0u};
  if (d >= 0 && d < a.D && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
    const long long S = (long long)a.D * a.H * a.W;
    const float* src = a.x + n * 3 * S + ((long long)d * a.H + gy) * a.W + gx;
    const float v0 = (src[0] - mu[0]) * rs[0], v1 = (src[S] - mu[1]) * rs[1], v2 = (src[2 * S] - mu[2]) * rs[2];
    r.lo = pack2_h16(v0, v1);
    r.hi = pack2_h16(v2, 1.0f);
  }
  return r;
}

// per-lane tap geometry of the im2col fragments: K index k = 32 ks + 8 kb + j  ->  tap (k >> 2), channel k & 3
//   forward A operand: lane kb = lane >> 4 reads taps 8 ks + 2 kb and + 1 (8 bytes each)
__device__ __forceinline__ void c1_tap(int t, int& kd, int& off) {
  if (t < 27) {
    kd = t / 9;
    const int rs = t - 9 * kd, kh = rs / 3, kw = rs - 3 * kh;
    off = (kh * C1_PW + kw) * 8;
  } else {
    kd = -1;
    off = 0;
  }
}

__global__ void __launch_bounds__(256) c3d_conv1_fwd_kernel(const C1Args a) {
  __shared__ __attribute__((aligned(16))) char s_patch[3 * C1_PLANE + 64];       // ring of 3 planes (+ a zero chunk)
  __shared__ __attribute__((aligned(16))) vnqa_bf16 s_w[C1_CO * C1_K];            // W' [co][k], later the store staging
  __shared__ float s_stat[4][C1_CO][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tilesX = a.W / C1_T, tilesY = a.H / C1_T;
  int b = blockIdx.x;
  const int tx = b % tilesX; b /= tilesX;
  const int ty = b % tilesY;
  const long long n = b / tilesY;
  float mu[3], rs[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { mu[c] = a.mean[c]; rs[c] = a.rstd[c]; }

  // W'[co][tap*4 + c]
  for (int i = threadIdx.x; i < C1_CO * C1_K; i += 256) {
    const int co = i / C1_K, k = i - co * C1_K, t = k >> 2, c = k & 3;
    float v = 0.f;
    if (t < 27) {
      if (c < 3) v = a.w[(co * 3 + c) * 27 + t] * a.gamma[c];
      else v = a.w[(co * 3 + 0) * 27 + t] * a.beta[0] + a.w[(co * 3 + 1) * 27 + t] * a.beta[1] + a.w[(co * 3 + 2) * 27 + t] * a.beta[2];
    }
    s_w[i] = f32_to_bf16(v);
  }
  if (threadIdx.x < 16) ((unsigned*)(s_patch + 3 * C1_PLANE))[threadIdx.x] = 0u;
  __syncthreads();
  vnqa_bf16x8 bfr[4][4];                                     // [n tile][k step]: lane (co = 16 nt + lane&15, kb = lane>>4)
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      bfr[nt][ks] = *(const vnqa_bf16x8*)(s_w + (nt * 16 + (lane & 15)) * C1_K + ks * 32 + (lane >> 4) * 8);
  float bias[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) bias[nt] = a.bias[nt * 16 + (lane & 15)];
  __syncthreads();                                           // s_w is reused as the store staging below

  // tap offsets of this lane's two taps per K-step
  int tkd[4][2], toff[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) c1_tap(ks * 8 + (lane >> 4) * 2 + e, tkd[ks][e], toff[ks][e]);
  // pixel of fragment row m = lane & 15 inside an M-tile (2 rows x 8 columns): window q = m >> 2, position sub = m & 3
  const int m = lane & 15;
  const int pix_r = (m & 3) >> 1, pix_c = 2 * (m >> 2) + (m & 1);

  const int y0 = ty * C1_T, x0 = tx * C1_T;
  auto fetch = [&](int d, C1Px (&px)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = threadIdx.x + 256 * h;
      px[h] = C1Px{0u, 0u};
      if (t < C1_PW * C1_PW) px[h] = c1_load_px(a, n, d, y0 - 1 + t / C1_PW, x0 - 1 + t % C1_PW, mu, rs);
    }
  };
  auto put = [&](int slot, const C1Px (&px)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int t = threadIdx.x + 256 * h;
      if (t < C1_PW * C1_PW) *(uint2*)(s_patch + slot * C1_PLANE + t * 8) = uint2{px[h].lo, px[h].hi};
    }
  };
  C1Px px[2];
  fetch(-1, px); put(2, px);
  fetch(0, px);  put(0, px);
  fetch(1, px);
  float st_s[4] = {0, 0, 0, 0}, st_q[4] = {0, 0, 0, 0};
  const int Ho = a.H / 2, Wo = a.W / 2;
  char* stage = (char*)s_w + wave * 3072;                    // per wave: 16 pooled pixels x 64 channels: 2 KiB values + 1 KiB idx
  for (int d = 0; d < a.D; ++d) {
    put((d + 1) % 3, px);
    __syncthreads();
    if (d + 2 <= a.D) fetch(d + 2, px);
    const char* const slot_base[3] = {s_patch + ((d + 2) % 3) * C1_PLANE, s_patch + (d % 3) * C1_PLANE,
                                      s_patch + ((d + 1) % 3) * C1_PLANE};
    const char* const zero_p = s_patch + 3 * C1_PLANE;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {                         // M-tile (rp, mc): rows 4 wave + 2 rp .. + 1, columns 8 mc .. + 7
      const int rp = mt >> 1, mc = mt & 1;
      const unsigned pixoff = (unsigned)(((4 * wave + 2 * rp + pix_r) * C1_PW + 8 * mc + pix_c) * 8);
      vnqa_f32x4 acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = vnqa_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        uint2 v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int kd = tkd[ks][e];
          const char* ptr = kd < 0 ? zero_p : (kd == 0 ? slot_base[0] : (kd == 1 ? slot_base[1] : slot_base[2])) + toff[ks][e] + pixoff;
          v[e] = *(const uint2*)ptr;
        }
        const uint4 u = uint4{v[0].x, v[0].y, v[1].x, v[1].y};
        const vnqa_bf16x8 af = __builtin_bit_cast(vnqa_bf16x8, u);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = VNQA_MFMA_16x16x32(af, bfr[nt][ks], acc[nt]);
      }
      // lane holds the 4 pixels of pooling window q = lane >> 4 for channel 16 nt + (lane & 15)
      const int pp = rp * 8 + mc * 4 + (lane >> 4);          // pooled pixel 0..15 of this wave (row rp, column 4 mc + q)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float best = acc[nt][0];
        int bi = 0;
#pragma unroll
        for (int e = 1; e < 4; ++e)
          if (acc[nt][e] > best) { best = acc[nt][e]; bi = e; }
        best += bias[nt];
        const bool pos = best > 0.f;
        const vnqa_bf16 hv = f32_to_bf16(pos ? best : 0.f);
        const float r = bf16_to_f32(hv);
        st_s[nt] += r;
        st_q[nt] = fmaf(r, r, st_q[nt]);
        const int c = nt * 16 + (lane & 15);
        ((vnqa_bf16*)stage)[pp * 64 + c] = hv;
        ((unsigned char*)stage)[2048 + pp * 64 + c] = (unsigned char)(pos ? bi : bi + 4);
      }
    }
    // this wave's 2 pooled rows x 8 pooled columns x 64 channels -> global, 16 bytes per lane
    {
      const int prow0 = (y0 + 4 * wave) / 2, pcol0 = x0 / 2;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int e = it * 64 + lane, ppx = e >> 3, ch = (e & 7) * 8;             // pooled pixel, first channel
        const size_t o = ((((size_t)n * a.D + d) * Ho + prow0 + (ppx >> 3)) * Wo + pcol0 + (ppx & 7)) * 64 + ch;
        *(uint4*)(a.p + o) = *(const uint4*)(stage + (ppx * 64 + ch) * 2);
      }
      {
        const int ppx = lane >> 2, ch = (lane & 3) * 16;
        const size_t o = ((((size_t)n * a.D + d) * Ho + prow0 + (ppx >> 3)) * Wo + pcol0 + (ppx & 7)) * 64 + ch;
        *(uint4*)(a.idx + o) = *(const uint4*)(stage + 2048 + ppx * 64 + ch);
      }
    }
    __syncthreads();
  }
  // statistics: lanes with equal (lane & 15) hold the same channel
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    float s = st_s[nt], q = st_q[nt];
    s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
    s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
    if (lane < 16) { s_stat[wave][nt * 16 + lane][0] = s; s_stat[wave][nt * 16 + lane][1] = q; }
  }
  __syncthreads();
  if (threadIdx.x < C1_CO) {
    const int c = threadIdx.x;
    a.partial[((size_t)blockIdx.x * C1_CO + c) * 2] = (s_stat[0][c][0] + s_stat[1][c][0]) + (s_stat[2][c][0] + s_stat[3][c][0]);
    a.partial[((size_t)blockIdx.x * C1_CO + c) * 2 + 1] = (s_stat[0][c][1] + s_stat[1][c][1]) + (s_stat[2][c][1] + s_stat[3][c][1]);
  }
}

// conv1 backward: G[co][k] = sum_px dY[px][co] A[px][k] (k = tap*4 + c, 112 columns = 7 N-tiles), persistent workgroups over
// (image, tile) with the depth loop inside; wave w owns output channels 16 w .. + 15.
constexpr int C1B_NT = 7;
__global__ void __launch_bounds__(256) c3d_conv1_bwd_kernel(const C1Args a, int n_work) {
  __shared__ __attribute__((aligned(16))) char s_patch[3 * C1_PLANE + 64];
  __shared__ __attribute__((aligned(16))) vnqa_bf16 s_dy[256 * C1_CO];            // [pixel k = row*16 + col][co]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tilesX = a.W / C1_T, tilesY = a.H / C1_T;
  float mu[3], rs[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { mu[c] = a.mean[c]; rs[c] = a.rstd[c]; }
  if (threadIdx.x < 16) ((unsigned*)(s_patch + 3 * C1_PLANE))[threadIdx.x] = 0u;
  const int g = lane >> 4, il = lane & 15, q4 = il >> 2, pp = il & 3;
  // B operand (im2col) through the transposed read: lane (q4, pp) loads pixel q4 (+4) of tap 4 nt + pp
  int bkd[C1B_NT], boff[C1B_NT];
#pragma unroll
  for (int nt = 0; nt < C1B_NT; ++nt) c1_tap(4 * nt + pp, bkd[nt], boff[nt]);
  // wave = (co half ch: 32 output channels = two M-tiles, pixel half ph: K-steps 4 ph .. 4 ph + 3 of every plane): one im2col
  // fragment feeds two MFMAs (18 transposed reads per 14 MFMAs instead of 16 per 7); the two pixel halves are separate partial
  // sums, folded with the other workgroups' by c3d_fold_kernel
  const int ch = wave & 1, ph = wave >> 1;
  vnqa_f32x4 acc[2][C1B_NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < C1B_NT; ++nt) acc[mt][nt] = vnqa_f32x4{0.f, 0.f, 0.f, 0.f};
  const int Ho = a.H / 2, Wo = a.W / 2;

  for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
    int b = work;
    const int tx = b % tilesX; b /= tilesX;
    const int ty = b % tilesY;
    const long long n = b / tilesY;
    const int y0 = ty * C1_T, x0 = tx * C1_T;
    auto fetch = [&](int d, C1Px (&px)[2]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t = threadIdx.x + 256 * h;
        px[h] = C1Px{0u, 0u};
        if (t < C1_PW * C1_PW) px[h] = c1_load_px(a, n, d, y0 - 1 + t / C1_PW, x0 - 1 + t % C1_PW, mu, rs);
      }
    };
    auto put = [&](int slot, const C1Px (&px)[2]) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t = threadIdx.x + 256 * h;
        if (t < C1_PW * C1_PW) *(uint2*)(s_patch + slot * C1_PLANE + t * 8) = uint2{px[h].lo, px[h].hi};
      }
    };
    C1Px px[2];
    __syncthreads();                                         // the previous tile's last plane is consumed
    fetch(-1, px); put(2, px);
    fetch(0, px);  put(0, px);
    fetch(1, px);
    for (int d = 0; d < a.D; ++d) {
      put((d + 1) % 3, px);
      // dY tile of plane d: 64 windows x 8 channel chunks, two work items per thread
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int item = threadIdx.x + 256 * it, win = item >> 3, cc8 = (item & 7) * 8, wy = win >> 3, wx = win & 7;
        const size_t o = ((((size_t)n * a.D + d) * Ho + y0 / 2 + wy) * Wo + x0 / 2 + wx) * 64 + cc8;
        const uint4 gq = *(const uint4*)(a.dp + o);
        const unsigned long long iv = *(const unsigned long long*)(a.idx + o);
        const unsigned gw[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
          unsigned ow[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned lo = ((unsigned)(iv >> (16 * j)) & 0xffu) == (unsigned)sub ? (gw[j] & 0xffffu) : 0u;
            const unsigned hi = ((unsigned)(iv >> (16 * j + 8)) & 0xffu) == (unsigned)sub ? (gw[j] & 0xffff0000u) : 0u;
            ow[j] = lo | hi;
          }
          const int k = (2 * wy + (sub >> 1)) * 16 + 2 * wx + (sub & 1);
          *(uint4*)(s_dy + k * C1_CO + cc8) = uint4{ow[0], ow[1], ow[2], ow[3]};
        }
      }
      __syncthreads();
      if (d + 2 <= a.D) fetch(d + 2, px);
      const unsigned slot_base[3] = {c3_lds_addr(s_patch) + (unsigned)(((d + 2) % 3) * C1_PLANE),
                                     c3_lds_addr(s_patch) + (unsigned)((d % 3) * C1_PLANE),
                                     c3_lds_addr(s_patch) + (unsigned)(((d + 1) % 3) * C1_PLANE)};
      const unsigned zero_addr = c3_lds_addr(s_patch) + 3 * C1_PLANE;
      const unsigned dy_base = c3_lds_addr(s_dy) + (unsigned)((4 * ch + (pp >> 1)) * 16 + ((pp & 1) << 3));
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {                       // K-step: pixels 32 s .. + 31 = tile rows 2 s, 2 s + 1
        const int s = 4 * ph + s4;
        // A: dY^T, lane group g supplies pixels 32 s + 8 g .. + 7 of channels 32 ch + 16 mt .. + 15
        const unsigned arow = (unsigned)((32 * s + 8 * g + q4) * (C1_CO * 2));
        c3_s16x4 alo[2], ahi[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          alo[mt] = c3_tr_read(dy_base + arow + mt * 32);
          ahi[mt] = c3_tr_read(dy_base + arow + mt * 32 + 4 * C1_CO * 2);
        }
        // B: pixels (row 2 s + (g >> 1), columns 8 (g & 1) + q4 [+ 4]) of 4 taps x 4 channels per N-tile
        const unsigned prow = (unsigned)(((2 * s + (g >> 1)) * C1_PW + 8 * (g & 1) + q4) * 8);
        c3_s16x4 blo[C1B_NT], bhi[C1B_NT];
#pragma unroll
        for (int nt = 0; nt < C1B_NT; ++nt) {
          const int kd = bkd[nt];
          const unsigned base = kd < 0 ? zero_addr : (kd == 0 ? slot_base[0] : (kd == 1 ? slot_base[1] : slot_base[2])) + boff[nt] + prow;
          blo[nt] = c3_tr_read(base);
          bhi[nt] = c3_tr_read(kd < 0 ? zero_addr : base + 4 * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        vnqa_bf16x8 af[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          asm volatile("" : "+v"(alo[mt]), "+v"(ahi[mt]));
          af[mt] = vnqa_bf16x8{alo[mt][0], alo[mt][1], alo[mt][2], alo[mt][3], ahi[mt][0], ahi[mt][1], ahi[mt][2], ahi[mt][3]};
        }
#pragma unroll
        for (int nt = 0; nt < C1B_NT; ++nt) {
          asm volatile("" : "+v"(blo[nt]), "+v"(bhi[nt]));
          const vnqa_bf16x8 bf = vnqa_bf16x8{blo[nt][0], blo[nt][1], blo[nt][2], blo[nt][3], bhi[nt][0], bhi[nt][1], bhi[nt][2], bhi[nt][3]};
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = VNQA_MFMA_16x16x32(af[mt], bf, acc[mt][nt]);
        }
      }
      __syncthreads();
    }
  }
  // D[co = 32 ch + 16 mt + 4 (lane >> 4) + e][k = 16 nt + (lane & 15)] of pixel half ph -> partial set 2 blockIdx + ph
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < C1B_NT; ++nt)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        a.partial[((size_t)(2 * blockIdx.x + ph) * C1_CO + 32 * ch + 16 * mt + 4 * (lane >> 4) + e) * (16 * C1B_NT) + 16 * nt + (lane & 15)] =
            acc[mt][nt][e];
}

// first fold of the split-K partials: out[z][i] = sum of partial[b][i] over the z-th share of the blocks
__global__ void __launch_bounds__(256) c3d_fold_kernel(const float* __restrict__ partial, float* __restrict__ out, int nblk, int n,
                                                       int shares) {
  const int i = blockIdx.x * 256 + threadIdx.x, z = blockIdx.y;
  if (i >= n) return;
  const int per = (nblk + shares - 1) / shares, b0 = z * per;
  int b1 = b0 + per;
  b1 = b1 < nblk ? b1 : nblk;
  float s = 0.f;
  for (int b = b0; b < b1; ++b) s += partial[(size_t)b * n + i];
  out[(size_t)z * n + i] = s;
}

// G = sum of the partials; dW[co][c][tap] = gamma_c G[co][tap][c] + beta_c G[co][tap][3]; db[co] = G[co][13][3];
// d gamma_c = sum W[co][c][tap] G[co][tap][c];  d beta_c = sum W[co][c][tap] G[co][tap][3]      (one workgroup)
__global__ void __launch_bounds__(1024) c3d_conv1_bwd_finalize_kernel(const float* __restrict__ partial, int nblk,
                                                                     const float* __restrict__ w, const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, float inv_scale,
                                                                     float* __restrict__ dw, float* __restrict__ db,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float s_g[C1_CO * 112];
  __shared__ float s_red[6][1024];
  for (int i = threadIdx.x; i < C1_CO * 112; i += 1024) {
    float s = 0.f;                                           // (<= 16 shares, already folded in order by c3d_fold_kernel)
    for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * C1_CO * 112 + i];
    s_g[i] = s * inv_scale;
  }
  __syncthreads();
  float r[6] = {0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < C1_CO * 27; i += 1024) {
    const int co = i / 27, t = i - co * 27;
    const float g3 = s_g[co * 112 + t * 4 + 3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float gc = s_g[co * 112 + t * 4 + c], wv = w[(co * 3 + c) * 27 + t];
      dw[(co * 3 + c) * 27 + t] = gamma[c] * gc + beta[c] * g3;
      r[c] = fmaf(wv, gc, r[c]);
      r[3 + c] = fmaf(wv, g3, r[3 + c]);
    }
    if (t == 13) db[co] = g3;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) s_red[k][threadIdx.x] = r[k];
  __syncthreads();
  if (threadIdx.x < 6) {
    double s = 0.0;
    for (int i = 0; i < 1024; ++i) s += (double)s_red[threadIdx.x][i];
    if (threadIdx.x < 3) dgamma[threadIdx.x] = (float)s;
    else dbeta[threadIdx.x - 3] = (float)s;
  }
}

View5 make_view(const vnqa_view5* v) {
  View5 r;
  r.base = v->base; r.sn = v->sn; r.sd = v->sd; r.sh = v->sh; r.sw = v->sw; r.sc = v->sc;
  r.D = v->d; r.H = v->h; r.W = v->w;
  return r;
}

int blocks_for(long long work, int per_block, int cap) {
  long long g = (work + per_block - 1) / per_block;
  return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int32_t vnqa_c3d_stats_blocks(int64_t rows) {
  long long b = rows / 512;
  return (int32_t)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

extern "C" int vnqa_c3d_stats_ncdhw(const float* x, float* partial, int32_t n, int32_t c, int64_t s, int32_t chunks, void* stream) {
  VNQA_CHECK_ARG(x && partial && n > 0 && c > 0 && s > 0 && chunks > 0, "c3d_stats_ncdhw: bad arguments");
  hipLaunchKernelGGL(stats_ncdhw_kernel, dim3(n * chunks, c), dim3(256), 0, (hipStream_t)stream, x, partial, c, (long long)s, chunks);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_c3d_stats_rows(const void* x, float* partial, int64_t rows, int32_t c, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(x && partial && rows > 0 && c > 0, "c3d_stats_rows: bad arguments");
  const int nb = vnqa_c3d_stats_blocks(rows);
  const long long rpb = (rows + nb - 1) / nb;
  if (dtype == VNQA_F32)
    hipLaunchKernelGGL(stats_rows_kernel<float>, dim3((c + 63) / 64, nb), dim3(1024), 0, (hipStream_t)stream, (const float*)x,
                       partial, (long long)rows, c, rpb);
  else
    hipLaunchKernelGGL(stats_rows_kernel<vnqa_bf16>, dim3((c + 63) / 64, nb), dim3(1024), 0, (hipStream_t)stream,
                       (const vnqa_bf16*)x, partial, (long long)rows, c, rpb);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_bn_finalize(const float* partial, int32_t nblk, int32_t c, double count, float eps, float momentum,
                                float* mean, float* rstd, float* running_mean, float* running_var, void* stream) {
  VNQA_CHECK_ARG(partial && mean && rstd && nblk > 0 && c > 0 && count > 0, "bn_finalize: bad arguments");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(c), dim3(256), 0, (hipStream_t)stream, partial, nblk, c, count, eps,
                     momentum, mean, rstd, running_mean, running_var);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_bn_rows_apply(const void* x, int32_t x_dtype, void* out, int32_t out_dtype, const vnqa_view5* out_view,
                                  const float* mean, const float* rstd, const float* gamma, const float* beta, int64_t rows,
                                  int32_t c, void* stream) {
  VNQA_CHECK_ARG(x && out && out_view && mean && rstd && gamma && beta && rows > 0 && c > 0 && c % 4 == 0, "bn_rows_apply: bad arguments");
  const View5 ov = make_view(out_view);
  const int grid = blocks_for(rows * (c / 4), 256 * 4, 4096);
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == VNQA_BF16 && out_dtype == VNQA_BF16)
    hipLaunchKernelGGL((bn_apply_kernel<vnqa_bf16, vnqa_bf16>), dim3(grid), dim3(256), 0, st, (const vnqa_bf16*)x, (vnqa_bf16*)out,
                       ov, mean, rstd, gamma, beta, (long long)rows, c);
  else if (x_dtype == VNQA_BF16 && out_dtype == VNQA_F32)
    hipLaunchKernelGGL((bn_apply_kernel<vnqa_bf16, float>), dim3(grid), dim3(256), 0, st, (const vnqa_bf16*)x, (float*)out, ov,
                       mean, rstd, gamma, beta, (long long)rows, c);
  else if (x_dtype == VNQA_F32 && out_dtype == VNQA_F32)
    hipLaunchKernelGGL((bn_apply_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float*)x, (float*)out, ov, mean, rstd,
                       gamma, beta, (long long)rows, c);
  else {
    vnqa_set_error("bn_rows_apply: unsupported dtype pair %d -> %d", x_dtype, out_dtype);
    return VNQA_ERR_UNSUPPORTED;
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_bn_rows_bwd(const void* dy, int32_t dy_dtype, const vnqa_view5* dy_view, const void* x, int32_t x_dtype,
                                void* dx, int32_t dx_dtype, const float* mean, const float* rstd, const float* gamma,
                                float* dgamma, float* dbeta, float* workspace, float grad_scale, int64_t rows, int32_t c,
                                void* stream) {
  VNQA_CHECK_ARG(dy && dy_view && x && dx && mean && rstd && gamma && dgamma && dbeta && workspace && rows > 0 && c > 0 && c % 4 == 0,
                 "bn_rows_bwd: bad arguments");
  const View5 dv = make_view(dy_view);
  const int nb = vnqa_c3d_stats_blocks(rows);
  const long long rpb = (rows + nb - 1) / nb;
  float* partial = workspace;                    // [nb][c][2]
  float* m_dy = workspace + (size_t)nb * c * 2;  // [c], [c]
  float* m_dyx = m_dy + c;
  hipStream_t st = (hipStream_t)stream;
  const dim3 rg((c + 63) / 64, nb);
  const int grid = blocks_for(rows * (c / 4), 256 * 4, 4096);
  if (x_dtype == VNQA_BF16 && dy_dtype == VNQA_BF16 && dx_dtype == VNQA_BF16) {
    if (dv.sc == 1 && c % 8 == 0 && 256 % (c / 8) == 0 && c <= 512)      // channel-last dy: 16-byte loads
      hipLaunchKernelGGL(bn_bwd_reduce_h16x8_kernel, dim3(nb), dim3(256), 2 * (256 / (c / 8)) * c * sizeof(float), st,
                         (const vnqa_bf16*)dy, dv, (const vnqa_bf16*)x, mean, rstd, partial, (long long)rows, c, rpb);
    else
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<vnqa_bf16, vnqa_bf16>), rg, dim3(1024), 0, st, (const vnqa_bf16*)dy, dv,
                         (const vnqa_bf16*)x, mean, rstd, partial, (long long)rows, c, rpb);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(256), 0, st, partial, nb, c, (double)rows, 1.0f / grad_scale,
                       dgamma, dbeta, m_dy, m_dyx);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<vnqa_bf16, vnqa_bf16, vnqa_bf16>), dim3(grid), dim3(256), 0, st, (const vnqa_bf16*)dy, dv,
                       (const vnqa_bf16*)x, (vnqa_bf16*)dx, mean, rstd, gamma, m_dy, m_dyx, (long long)rows, c);
  } else if (x_dtype == VNQA_BF16 && dy_dtype == VNQA_F32 && dx_dtype == VNQA_BF16) {
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<vnqa_bf16, float>), rg, dim3(1024), 0, st, (const float*)dy, dv, (const vnqa_bf16*)x,
                       mean, rstd, partial, (long long)rows, c, rpb);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(256), 0, st, partial, nb, c, (double)rows, 1.0f / grad_scale,
                       dgamma, dbeta, m_dy, m_dyx);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<vnqa_bf16, float, vnqa_bf16>), dim3(grid), dim3(256), 0, st, (const float*)dy, dv,
                       (const vnqa_bf16*)x, (vnqa_bf16*)dx, mean, rstd, gamma, m_dy, m_dyx, (long long)rows, c);
  } else if (x_dtype == VNQA_F32 && dy_dtype == VNQA_F32 && dx_dtype == VNQA_F32) {
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<float, float>), rg, dim3(1024), 0, st, (const float*)dy, dv, (const float*)x, mean, rstd,
                       partial, (long long)rows, c, rpb);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(c), dim3(256), 0, st, partial, nb, c, (double)rows, 1.0f / grad_scale,
                       dgamma, dbeta, m_dy, m_dyx);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<float, float, float>), dim3(grid), dim3(256), 0, st, (const float*)dy, dv, (const float*)x,
                       (float*)dx, mean, rstd, gamma, m_dy, m_dyx, (long long)rows, c);
  } else {
    vnqa_set_error("bn_rows_bwd: unsupported dtype combination");
    return VNQA_ERR_UNSUPPORTED;
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int32_t vnqa_pool444_blocks(int32_t n, int32_t d, int32_t h, int32_t w, int32_t c) {
  const long long work = (long long)n * (d / 4) * (h / 4) * (w / 4) * (c / 8);
  return blocks_for(work, 256, 1024);
}

extern "C" int vnqa_pool444_fwd(const void* y, void* p, uint8_t* idx, float* partial, int32_t n, int32_t d, int32_t h, int32_t w,
                                int32_t c, void* stream) {
  VNQA_CHECK_ARG(y && p && idx && partial && n > 0 && d >= 4 && h >= 4 && w >= 4 && c > 0 && c % 8 == 0, "pool444_fwd: bad arguments");
  const int grid = vnqa_pool444_blocks(n, d, h, w, c);
  hipLaunchKernelGGL(pool444_fwd_kernel, dim3(grid), dim3(256), 2 * c * sizeof(float), (hipStream_t)stream, (const vnqa_bf16*)y,
                     (vnqa_bf16*)p, idx, partial, n, d, h, w, c);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_pool444_bwd(const void* dp, const uint8_t* idx, void* dy, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                                void* stream) {
  VNQA_CHECK_ARG(dp && idx && dy && n > 0 && d >= 4 && h >= 4 && w >= 4 && c > 0 && c % 8 == 0, "pool444_bwd: bad arguments");
  const int grid = blocks_for((long long)n * d * h * w * (c / 8), 256 * 4, 8192);
  hipLaunchKernelGGL(pool444_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const vnqa_bf16*)dp, idx, (vnqa_bf16*)dy, n,
                     d, h, w, c);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_c3d_conv1_supported(int32_t n, int32_t d, int32_t h, int32_t w) {
  return (n > 0 && d > 0 && h >= 16 && w >= 16 && h % 16 == 0 && w % 16 == 0) ? 1 : 0;
}

extern "C" int32_t vnqa_c3d_conv1_fwd_blocks(int32_t n, int32_t h, int32_t w) { return n * (h / 16) * (w / 16); }
extern "C" int32_t vnqa_c3d_conv1_bwd_blocks(int32_t n, int32_t h, int32_t w) {
  const int work = n * (h / 16) * (w / 16);
  return work < 256 ? work : 256;
}

extern "C" int vnqa_c3d_conv1_fwd(const float* x, const float* weight, const float* bias, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, void* p, uint8_t* idx, float* partial, int32_t n,
                                  int32_t d, int32_t h, int32_t w, void* stream) {
  VNQA_CHECK_ARG(x && weight && bias && mean && rstd && gamma && beta && p && idx && partial, "c3d_conv1_fwd: null pointer");
  VNQA_CHECK_ARG(vnqa_c3d_conv1_supported(n, d, h, w), "c3d_conv1_fwd: H and W must be multiples of 16 (got %d x %d)", h, w);
  C1Args a;
  a.x = x; a.w = weight; a.bias = bias; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
  a.p = (vnqa_bf16*)p; a.idx = idx; a.partial = partial; a.dp = nullptr;
  a.N = n; a.D = d; a.H = h; a.W = w;
  hipLaunchKernelGGL(c3d_conv1_fwd_kernel, dim3(vnqa_c3d_conv1_fwd_blocks(n, h, w)), dim3(256), 0, (hipStream_t)stream, a);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_c3d_conv1_bwd(const float* x, const float* weight, const float* mean, const float* rstd, const float* gamma,
                                  const float* beta, const void* dp, const uint8_t* idx, float* partial, float grad_scale,
                                  float* dweight, float* dbias, float* dgamma, float* dbeta, int32_t n, int32_t d, int32_t h,
                                  int32_t w, void* stream) {
  VNQA_CHECK_ARG(x && weight && mean && rstd && gamma && beta && dp && idx && partial && dweight && dbias && dgamma && dbeta,
                 "c3d_conv1_bwd: null pointer");
  VNQA_CHECK_ARG(vnqa_c3d_conv1_supported(n, d, h, w), "c3d_conv1_bwd: H and W must be multiples of 16 (got %d x %d)", h, w);
  C1Args a;
  a.x = x; a.w = weight; a.bias = nullptr; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta;
  a.p = nullptr; a.idx = (unsigned char*)idx; a.partial = partial; a.dp = (const vnqa_bf16*)dp;
  a.N = n; a.D = d; a.H = h; a.W = w;
  const int grid = vnqa_c3d_conv1_bwd_blocks(n, h, w);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(c3d_conv1_bwd_kernel, dim3(grid), dim3(256), 0, st, a, n * (h / 16) * (w / 16));
  VNQA_CHECK_LAUNCH();
  // partials [2 grid][64][112] (two pixel halves per workgroup) -> 16 shares (behind them in the workspace) -> the one-workgroup finalize
  const int n_el = C1_CO * 112, n_sets = 2 * grid, shares = n_sets < 16 ? n_sets : 16;
  float* folded = partial + (size_t)n_sets * n_el;
  hipLaunchKernelGGL(c3d_fold_kernel, dim3((n_el + 255) / 256, shares), dim3(256), 0, st, (const float*)partial, folded, n_sets, n_el,
                     shares);
  VNQA_CHECK_LAUNCH();
  hipLaunchKernelGGL(c3d_conv1_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, (const float*)folded, shares, weight, gamma, beta,
                     1.0f / grad_scale, dweight, dbias, dgamma, dbeta);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
