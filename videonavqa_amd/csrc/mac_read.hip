// mac_read.hip — fused ReadUnit attention of MACNetwork (models/mac.py:53-62), forward and backward.
//
// Upstream, every reasoning step and frame computes
//     concat = Linear_{2d->d}([mem * know ; know]);  attn = softmax_s( Linear_{d->1}(concat * control) );
//     read   = sum_s attn[s] * know[:, s]
// The score is linear in `concat`, so with  [W1 | W2] = concat.weight,  v = control * w_attn,
// u = mem * (W1^T v)  and the step-invariant  pre = know W2^T + b  (one GEMM per forward) it is
//     score[n][s] = know[n][s][:] . u[n] + pre[n][s][:] . v[n] + b_attn                      (models/mac.py docstring)
// — three passes over the knowledge base of one image.  With pre = v = NULL the same kernels are a plain
// dot-product attention pool, which is ControlUnit's attention over the question words (:36-42:
// softmax_l(context[l] . (cq * w_attn) + b) weighted sum of context).  One workgroup owns one packed image n:
//   forward : scores (a wave per position, lanes split the channels in 16-byte pieces), softmax over the S
//             positions in LDS, then read[c] = sum_s p[s] know[s][c] (second sweep comes from L2);
//   backward: dp[s] = know[s].dread, dscore = p (dp - <p,dp>), du = sum_s dscore[s] know[s], dv = sum_s dscore[s] pre[s];
//             the OUTER-PRODUCT gradients  dknow += dscore (x) u + p (x) dread,  dpre += dscore (x) v  are not
//             accumulated per step (12 read-modify-write sweeps of two [N][S][C] fp32 tensors): the per-step
//             factors are kept and `mac_read_accum` forms both gradients in ONE pass over the positions after the
//             last step's backward (a [S x 2K] x [2K x C] product per image, factors staged in LDS).
// HBM/L2-bound: per call 3 reads of [S][C] per image; know/pre in the compute dtype (bf16 | f32), all
// vectors, probabilities and accumulators fp32.
#include "vnqa_common.h"

namespace {

// 16 waves per image: a wave has two 1-KiB row loads in flight per position, and one workgroup per image is all the
// parallelism there is (280 images on 256 CUs) — with 4 waves the knowledge-base sweeps ran at 2.2 TB/s, bound by the
// latency of their own loads
constexpr int NTH = 1024, NWV = NTH / 64;
constexpr int MAX_S = 1024;

template <typename T> struct Row8;
template <> struct Row8<vnqa_bf16> {
  static __device__ __forceinline__ void load(const vnqa_bf16* p, float v[8]) {
    const uint4 u = *(const uint4*)p;
    v[0] = h16_lo(u.x); v[1] = h16_hi(u.x);
    v[2] = h16_lo(u.y); v[3] = h16_hi(u.y);
    v[4] = h16_lo(u.z); v[5] = h16_hi(u.z);
    v[6] = h16_lo(u.w); v[7] = h16_hi(u.w);
  }
  static __device__ __forceinline__ void store(vnqa_bf16* p, const float v[8]) {
    uint4 u;
    u.x = pack2_h16(v[0], v[1]);
    u.y = pack2_h16(v[2], v[3]);
    u.z = pack2_h16(v[4], v[5]);
    u.w = pack2_h16(v[6], v[7]);
    *(uint4*)p = u;
  }
};
template <> struct Row8<float> {
  static __device__ __forceinline__ void load(const float* p, float v[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ __forceinline__ void store(float* p, const float v[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};

__device__ __forceinline__ void load8f(const float* p, float v[8]) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// block-wide reduction helpers over NTH threads (values broadcast to every thread)
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_reduce_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = red[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) m = fmaxf(m, red[w]);
  return m;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_reduce_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = red[0];
#pragma unroll
  for (int w = 1; w < NWV; ++w) t += red[w];
  return t;
}

// t[s] = sum_c a[s][c] x[c] (+ b[s][c] y[c]) for every position s of image n -> LDS t[]
template <typename T, bool TWO>
__device__ __forceinline__ void rows_dot(const T* a, const T* b, const float* x, const float* y, int S, int C, int ld,
                                         float* t) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int s = wave; s < S; s += NWV) {
    float acc = 0.f;
    for (int c = lane * 8; c < C; c += 512) {
      float av[8], xv[8];
      Row8<T>::load(a + (size_t)s * ld + c, av);
      load8f(x + c, xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf(av[e], xv[e], acc);
      if (TWO) {
        float bv[8], yv[8];
        Row8<T>::load(b + (size_t)s * ld + c, bv);
        load8f(y + c, yv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(bv[e], yv[e], acc);
      }
    }
    acc = wave_reduce_sum(acc);
    if (lane == 0) t[s] = acc;
  }
}

// out[c] = sum_s w[s] a[s][c]: thread = (8-channel group cg = tid % 64, row phase tid / 64); partials over the NWV row
// phases are summed in order through LDS (red8: [NWV][64][8] floats)
// Optional epilogue: out = sum * omask[c] (omask per image, or null), out2[c] = out[c] * out2_col[c] (out2 null: none).
template <typename T>
__device__ __forceinline__ void weighted_colsum(const T* a, const float* w, int S, int C, int ld, float* red8, float* out,
                                                const float* omask = nullptr, float* out2 = nullptr,
                                                const float* out2_col = nullptr) {
  const int cg = threadIdx.x & 63, ph = threadIdx.x >> 6;
  for (int c0 = 0; c0 < C; c0 += 512) {
    const int c = c0 + cg * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c < C) {
      for (int s = ph; s < S; s += NWV) {
        float av[8];
        Row8<T>::load(a + (size_t)s * ld + c, av);
        const float ws = w[s];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(ws, av[e], acc[e]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) red8[(ph * 64 + cg) * 8 + e] = acc[e];
    __syncthreads();
    // 512 threads fold: thread (cg, e) sums element e of channel group cg over the phases
    if (threadIdx.x < 512) {
      const int g = threadIdx.x >> 3, e = threadIdx.x & 7;
      if (c0 + g * 8 < C) {
        float t = red8[g * 8 + e];
#pragma unroll
        for (int w = 1; w < NWV; ++w) t += red8[(w * 64 + g) * 8 + e];
        const int cc = c0 + g * 8 + e;
        if (omask != nullptr) t *= omask[cc];
        out[cc] = t;
        if (out2 != nullptr) out2[cc] = t * out2_col[cc];
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(NTH) mac_read_fwd_kernel(const T* __restrict__ kn, const T* __restrict__ pre,
                                                           const float* __restrict__ u, const float* __restrict__ v,
                                                           const float* __restrict__ bias, float* __restrict__ p,
                                                           float* __restrict__ read, const float* __restrict__ out_mask,
                                                           float* __restrict__ out2, const float* __restrict__ out2_col,
                                                           int S, int C, int ld) {
  __shared__ float sc[MAX_S];
  __shared__ float red[NWV];
  __shared__ float red8[NWV * 64 * 8];
  const int n = blockIdx.x;
  const T* a = kn + (size_t)n * S * ld;
  if (pre != nullptr)
    rows_dot<T, true>(a, pre + (size_t)n * S * ld, u + (size_t)n * C, v + (size_t)n * C, S, C, ld, sc);
  else
    rows_dot<T, false>(a, nullptr, u + (size_t)n * C, nullptr, S, C, ld, sc);
  __syncthreads();
  const float b0 = bias ? bias[0] : 0.f;
  float m = -INFINITY;
  for (int s = threadIdx.x; s < S; s += NTH) m = fmaxf(m, sc[s] + b0);
  m = block_max(m, red);
  float z = 0.f;
  for (int s = threadIdx.x; s < S; s += NTH) {
    const float e = expf(sc[s] + b0 - m);
    sc[s] = e;
    z += e;
  }
  z = block_sum(z, red);
  const float inv = 1.f / z;
  for (int s = threadIdx.x; s < S; s += NTH) {
    const float pr = sc[s] * inv;
    sc[s] = pr;
    p[(size_t)n * S + s] = pr;
  }
  __syncthreads();
  weighted_colsum<T>(a, sc, S, C, ld, red8, read + (size_t)n * C, out_mask ? out_mask + (size_t)n * C : nullptr,
                     out2 ? out2 + (size_t)n * C : nullptr, out2_col);
}

// Optional prologue (pro_x != null): this image's dread row is first FORMED here and written out,
//   dread[c] = (pro_x[c] * pro_col[c] + pro_add[c]) * pro_mask[c]        (pro_add / pro_mask per image, or null)
// — the elementwise steps between two attention backward passes of a MAC reasoning step.  Optional epilogue: du2 = du * du2_col.
// (dread is deliberately neither const nor restrict: it is written and then read through the same pointer.)
template <typename T>
__global__ void __launch_bounds__(NTH) mac_read_bwd_kernel(const T* __restrict__ kn, const T* __restrict__ pre,
                                                           const float* __restrict__ p, float* dread,
                                                           float* __restrict__ dscore, float* __restrict__ du,
                                                           float* __restrict__ dv, const float* __restrict__ pro_x,
                                                           const float* __restrict__ pro_col, const float* __restrict__ pro_add,
                                                           const float* __restrict__ pro_mask, float* __restrict__ du2,
                                                           const float* __restrict__ du2_col, int S, int C, int ld) {
  __shared__ float ds[MAX_S];
  __shared__ float red[NWV];
  __shared__ float red8[NWV * 64 * 8];
  const int n = blockIdx.x;
  const T* a = kn + (size_t)n * S * ld;
  if (pro_x != nullptr) {
    for (int c = threadIdx.x; c < C; c += NTH) {
      const size_t i = (size_t)n * C + c;
      float t = pro_x[i] * pro_col[c];
      if (pro_add != nullptr) t += pro_add[i];
      if (pro_mask != nullptr) t *= pro_mask[i];
      dread[i] = t;
    }
    __threadfence_block();
    __syncthreads();
  }
  rows_dot<T, false>(a, nullptr, dread + (size_t)n * C, nullptr, S, C, ld, ds);      // dp[s]
  __syncthreads();
  float dot = 0.f;
  for (int s = threadIdx.x; s < S; s += NTH) dot += p[(size_t)n * S + s] * ds[s];
  dot = block_sum(dot, red);
  for (int s = threadIdx.x; s < S; s += NTH) {
    const float g = p[(size_t)n * S + s] * (ds[s] - dot);
    ds[s] = g;
    dscore[(size_t)n * S + s] = g;
  }
  __syncthreads();
  weighted_colsum<T>(a, ds, S, C, ld, red8, du + (size_t)n * C, nullptr, du2 ? du2 + (size_t)n * C : nullptr, du2_col);
  if (pre != nullptr) weighted_colsum<T>(pre + (size_t)n * S * ld, ds, S, C, ld, red8, dv + (size_t)n * C);
}

// dkn[n][s][c] = sum_i dscore_i[n][s] u_i[n][c] + p_i[n][s] dread_i[n][c];  dpre[n][s][c] = sum_i dscore_i[n][s] v_i[n][c]
// factors: [K][N][S] / [K][N][C] fp32.  Block = (image, 64-position chunk); the image's 3 K C factor floats sit in LDS.
template <typename T>
__global__ void __launch_bounds__(NTH) mac_read_accum_kernel(const float* __restrict__ dscore, const float* __restrict__ p,
                                                             const float* __restrict__ u, const float* __restrict__ v,
                                                             const float* __restrict__ dread, T* __restrict__ dkn,
                                                             T* __restrict__ dpre, int K, int N, int S, int C, int ld) {
  extern __shared__ __attribute__((aligned(16))) float fac[];      // [3][K][C] then [2][K][64] row factors
  const int n = blockIdx.x, s0 = blockIdx.y * 64;
  float* fu = fac;
  float* fd = fac + (size_t)K * C;
  float* fv = fac + (size_t)2 * K * C;
  float* rs = fac + (size_t)3 * K * C;        // dscore rows [K][64]
  float* rp = rs + K * 64;                    // p rows      [K][64]
  for (int i = threadIdx.x; i < K * C; i += NTH) {
    const int k = i / C, c = i - k * C;
    const size_t src = ((size_t)k * N + n) * C + c;
    fu[i] = u[src];
    fd[i] = dread[src];
    fv[i] = dpre != nullptr ? v[src] : 0.f;
  }
  for (int i = threadIdx.x; i < K * 64; i += NTH) {
    const int k = i >> 6, r = i & 63;
    const int s = s0 + r;
    const size_t src = ((size_t)k * N + n) * S + s;
    rs[i] = s < S ? dscore[src] : 0.f;
    rp[i] = s < S ? p[src] : 0.f;
  }
  __syncthreads();
  const int cg = threadIdx.x & 63, ph = threadIdx.x >> 6;
  for (int c0 = 0; c0 < ld; c0 += 512) {
    const int c = c0 + cg * 8;
    if (c >= ld) continue;
    for (int r = ph; r < 64; r += NWV) {
      const int s = s0 + r;
      if (s >= S) break;
      float a8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (c < C) {
        for (int k = 0; k < K; ++k) {
          const float gs = rs[k * 64 + r], pp = rp[k * 64 + r];
          float uu[8], dd[8], vv[8];
          load8f(fu + (size_t)k * C + c, uu);
          load8f(fd + (size_t)k * C + c, dd);
          load8f(fv + (size_t)k * C + c, vv);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            a8[e] = fmaf(gs, uu[e], fmaf(pp, dd[e], a8[e]));
            b8[e] = fmaf(gs, vv[e], b8[e]);
          }
        }
      }
      const size_t o = ((size_t)n * S + s) * ld + c;      // channels past C (padding) get zeros
      Row8<T>::store(dkn + o, a8);
      if (dpre != nullptr) Row8<T>::store(dpre + o, b8);
    }
  }
}

}  // namespace

extern "C" int vnqa_mac_read_fwd_scaled(const void* know, const void* pre, const float* u, const float* v, const float* bias,
                                        float* p, float* read, const float* out_mask, float* out2, const float* out2_col,
                                        int32_t n, int32_t s, int32_t c, int32_t ld, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(know && u && p && read && (pre == nullptr) == (v == nullptr), "mac_read_fwd: null pointer (pre and v come together)");
  VNQA_CHECK_ARG((out2 == nullptr) == (out2_col == nullptr), "mac_read_fwd: out2 and out2_col come together");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "mac_read_fwd: bad dtype %d", dtype);
  VNQA_CHECK_ARG(n > 0 && s > 0 && s <= MAX_S, "mac_read_fwd: positions per image must be in 1..%d (got %d)", MAX_S, s);
  VNQA_CHECK_ARG(c > 0 && c % 8 == 0 && ld >= c && ld % 8 == 0, "mac_read_fwd: c=%d ld=%d must be multiples of 8, ld>=c", c, ld);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(mac_read_fwd_kernel<vnqa_bf16>, dim3(n), dim3(NTH), 0, st, (const vnqa_bf16*)know,
                       (const vnqa_bf16*)pre, u, v, bias, p, read, out_mask, out2, out2_col, s, c, ld);
  else
    hipLaunchKernelGGL(mac_read_fwd_kernel<float>, dim3(n), dim3(NTH), 0, st, (const float*)know, (const float*)pre, u, v,
                       bias, p, read, out_mask, out2, out2_col, s, c, ld);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_mac_read_fwd(const void* know, const void* pre, const float* u, const float* v, const float* bias,
                                 float* p, float* read, int32_t n, int32_t s, int32_t c, int32_t ld, int32_t dtype,
                                 void* stream) {
  return vnqa_mac_read_fwd_scaled(know, pre, u, v, bias, p, read, nullptr, nullptr, nullptr, n, s, c, ld, dtype, stream);
}

extern "C" int vnqa_mac_read_bwd_fused(const void* know, const void* pre, const float* p, float* dread, const float* pro_x,
                                       const float* pro_col, const float* pro_add, const float* pro_mask, float* dscore,
                                       float* du, float* dv, float* du2, const float* du2_col, int32_t n, int32_t s, int32_t c,
                                       int32_t ld, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(know && p && dread && dscore && du && (pre == nullptr) == (dv == nullptr), "mac_read_bwd: null pointer (pre and dv come together)");
  VNQA_CHECK_ARG((pro_x == nullptr) == (pro_col == nullptr) && (pro_x != nullptr || (pro_add == nullptr && pro_mask == nullptr)),
                 "mac_read_bwd: prologue needs pro_x and pro_col");
  VNQA_CHECK_ARG((du2 == nullptr) == (du2_col == nullptr), "mac_read_bwd: du2 and du2_col come together");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "mac_read_bwd: bad dtype %d", dtype);
  VNQA_CHECK_ARG(n > 0 && s > 0 && s <= MAX_S, "mac_read_bwd: positions per image must be in 1..%d (got %d)", MAX_S, s);
  VNQA_CHECK_ARG(c > 0 && c % 8 == 0 && ld >= c && ld % 8 == 0, "mac_read_bwd: c=%d ld=%d must be multiples of 8, ld>=c", c, ld);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(mac_read_bwd_kernel<vnqa_bf16>, dim3(n), dim3(NTH), 0, st, (const vnqa_bf16*)know,
                       (const vnqa_bf16*)pre, p, dread, dscore, du, dv, pro_x, pro_col, pro_add, pro_mask, du2, du2_col, s, c, ld);
  else
    hipLaunchKernelGGL(mac_read_bwd_kernel<float>, dim3(n), dim3(NTH), 0, st, (const float*)know, (const float*)pre, p,
                       dread, dscore, du, dv, pro_x, pro_col, pro_add, pro_mask, du2, du2_col, s, c, ld);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_mac_read_bwd(const void* know, const void* pre, const float* p, const float* dread, float* dscore,
                                 float* du, float* dv, int32_t n, int32_t s, int32_t c, int32_t ld, int32_t dtype,
                                 void* stream) {
  return vnqa_mac_read_bwd_fused(know, pre, p, const_cast<float*>(dread), nullptr, nullptr, nullptr, nullptr, dscore, du, dv,
                                 nullptr, nullptr, n, s, c, ld, dtype, stream);
}

extern "C" int vnqa_mac_read_accum(const float* dscore, const float* p, const float* u, const float* v,
                                   const float* dread, void* dknow, void* dpre, int32_t k, int32_t n, int32_t s,
                                   int32_t c, int32_t ld, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(dscore && p && u && dread && dknow && (v == nullptr) == (dpre == nullptr), "mac_read_accum: null pointer (v and dpre come together)");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "mac_read_accum: bad dtype %d", dtype);
  VNQA_CHECK_ARG(k > 0 && n > 0 && s > 0, "mac_read_accum: empty problem");
  VNQA_CHECK_ARG(c > 0 && c % 8 == 0 && ld >= c && ld % 8 == 0, "mac_read_accum: c=%d ld=%d must be multiples of 8, ld>=c", c, ld);
  const size_t lds = ((size_t)3 * k * c + (size_t)2 * k * 64) * sizeof(float);
  VNQA_CHECK_ARG(lds <= 160 * 1024, "mac_read_accum: %d steps x %d channels need %zu B of LDS (> 160 KiB)", k, c, lds);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(n, (s + 63) / 64);
  if (dtype == VNQA_BF16) {
    auto kern = mac_read_accum_kernel<vnqa_bf16>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vnqa_set_error("mac_read_accum: cannot reserve %zu B of LDS", lds);
      return VNQA_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NTH), lds, st, dscore, p, u, v, dread, (vnqa_bf16*)dknow, (vnqa_bf16*)dpre, k, n, s, c, ld);
  } else {
    auto kern = mac_read_accum_kernel<float>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      vnqa_set_error("mac_read_accum: cannot reserve %zu B of LDS", lds);
      return VNQA_ERR_HIP;
    }
    hipLaunchKernelGGL(kern, grid, dim3(NTH), lds, st, dscore, p, u, v, dread, (float*)dknow, (float*)dpre, k, n, s, c, ld);
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
