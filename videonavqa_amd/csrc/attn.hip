// attn.hip — temporal softmax-attention over frames (models/film_attn_pt_stem.py:268-290), fused.
//
//   score[b,t] = valid[b,t] * (w . feat[b,t,:] + bias) + mask[b,t]        (fc_attn_1 on valid entries, :268-281,
//                                                                           -(1<<31) mask of :251)
//   coef[b,:]  = softmax_t(score[b,:])                                     (:288; the h-dependent term is constant
//                                                                           along t and cancels)
//   ctxt[b,:]  = sum_t coef[b,t] * feat[b,t,:]                             (:290)
//
// One workgroup per sample; feat is [B][T][A] so every frame row is a coalesced A-float read; the dot
// products and the softmax use wavefront shuffles, the frame axis lives in LDS.  Backward is the same
// shape: d feat (both paths), d w, d bias.  Tiny and latency-bound: it exists to keep the tail of the
// step in two launches instead of ~25 framework kernels.
#include "vnqa_common.h"

namespace {

__device__ __forceinline__ float block_reduce_sum(float v, float* s_tmp, int nwaves) {
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s_tmp[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < nwaves; ++w) t += s_tmp[w];
  return t;
}

// threads: one per feature channel a (blockDim.x = A rounded up to 64, <= 1024)
__global__ void temporal_attn_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ valid,
                                         const float* __restrict__ mask, const float* __restrict__ w,
                                         const float* __restrict__ bias, float* __restrict__ coef,
                                         float* __restrict__ ctxt, int T, int A) {
  extern __shared__ float s_mem[];            // [T] scores/coefs + [16] scratch
  float* s_sc = s_mem;
  float* s_tmp = s_mem + T;
  const int b = blockIdx.x, a = threadIdx.x;
  const int nw = blockDim.x >> 6;
  const float* fb = feat + (size_t)b * T * A;
  const float wa = a < A ? w[a] : 0.f;
  for (int t = 0; t < T; ++t) {
    const float dot = block_reduce_sum(a < A ? wa * fb[(size_t)t * A + a] : 0.f, s_tmp, nw);
    if (a == 0) s_sc[t] = valid[(size_t)b * T + t] * (dot + bias[0]) + mask[(size_t)b * T + t];
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = 0; t < T; ++t) mx = fmaxf(mx, s_sc[t]);
  float den = 0.f;
  for (int t = 0; t < T; ++t) den += expf(s_sc[t] - mx);
  __syncthreads();
  if (a < T) {
    for (int t = a; t < T; t += blockDim.x) {
      const float c = expf(s_sc[t] - mx) / den;
      s_sc[t] = c;
      coef[(size_t)b * T + t] = c;
    }
  }
  __syncthreads();
  if (a < A) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc = fmaf(s_sc[t], fb[(size_t)t * A + a], acc);
    ctxt[(size_t)b * A + a] = acc;
  }
}

// dfeat[b,t,a] = coef_t * dctxt[a] + dscore_t * valid_t * w[a];  dscore_t = coef_t * (g_t - sum_s coef_s g_s),
// g_t = dctxt . feat[t];  dw_part[b,a] = sum_t dscore_t valid_t feat[t,a];  db_part[b] = sum_t dscore_t valid_t
__global__ void temporal_attn_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ valid,
                                         const float* __restrict__ w, const float* __restrict__ coef,
                                         const float* __restrict__ dctxt, float* __restrict__ dfeat,
                                         float* __restrict__ dw_part, float* __restrict__ db_part, int T, int A) {
  extern __shared__ float s_mem[];            // [T] g / dscore*valid, [T] coef, [16] scratch
  float* s_g = s_mem;
  float* s_c = s_mem + T;
  float* s_tmp = s_mem + 2 * T;
  const int b = blockIdx.x, a = threadIdx.x;
  const int nw = blockDim.x >> 6;
  const float* fb = feat + (size_t)b * T * A;
  const float da = a < A ? dctxt[(size_t)b * A + a] : 0.f;
  const float wa = a < A ? w[a] : 0.f;
  for (int t = 0; t < T; ++t) {
    const float g = block_reduce_sum(a < A ? da * fb[(size_t)t * A + a] : 0.f, s_tmp, nw);
    if (a == 0) {
      s_g[t] = g;
      s_c[t] = coef[(size_t)b * T + t];
    }
  }
  __syncthreads();
  float dotcg = 0.f;
  for (int t = 0; t < T; ++t) dotcg += s_c[t] * s_g[t];
  __syncthreads();
  float dbsum = 0.f;
  if (a == 0) {
    for (int t = 0; t < T; ++t) {
      const float ds = s_c[t] * (s_g[t] - dotcg) * valid[(size_t)b * T + t];
      s_g[t] = ds;
      dbsum += ds;
    }
    db_part[b] = dbsum;
  }
  __syncthreads();
  if (a < A) {
    float dwa = 0.f;
    for (int t = 0; t < T; ++t) {
      const float f = fb[(size_t)t * A + a];
      dfeat[((size_t)b * T + t) * A + a] = s_c[t] * da + s_g[t] * wa;
      dwa = fmaf(s_g[t], f, dwa);
    }
    dw_part[(size_t)b * A + a] = dwa;
  }
}

// ---- the same two kernels on the PACKED image list --------------------------------------------------------------------
// f [n_img][ld] (element type T: the fc_embed_attn GEMM's output) holds the feature row of image n = frame_off[t] + b for
// the valid (sample b, frame t) pairs (frame-major packing, sample b valid in frame t iff b < frame_off[t+1]-frame_off[t]).
// The zero-padded dense [B][T][A] tensor, the validity grid and the -(1<<31) mask of models/film_attn_pt_stem.py:245-256
// are never materialised: score = dot + bias on valid pairs, -(1<<31) on processed frames without this sample, 0 (feature
// 0) on frames past the longest video.
template <typename T>
__global__ void temporal_attn_packed_fwd_kernel(const T* __restrict__ f, int ld, const int* __restrict__ frame_off, int n_frames,
                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ coef, float* __restrict__ ctxt, int Tn, int A) {
  extern __shared__ float s_mem[];            // [T] scores/coefs + [16] scratch
  float* s_sc = s_mem;
  float* s_tmp = s_mem + Tn;
  const int b = blockIdx.x, a = threadIdx.x;
  const int nw = blockDim.x >> 6;
  const float wa = a < A ? w[a] : 0.f;
  for (int t = 0; t < Tn; ++t) {
    const bool valid = t < n_frames && b < frame_off[t + 1] - frame_off[t];
    float v = 0.f;
    if (valid && a < A) v = wa * ElemOps<T>::load(f[(size_t)(frame_off[t] + b) * ld + a]);
    const float dot = block_reduce_sum(v, s_tmp, nw);
    if (a == 0) s_sc[t] = valid ? dot + bias[0] : (t < n_frames ? -2147483648.f : 0.f);
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = 0; t < Tn; ++t) mx = fmaxf(mx, s_sc[t]);
  float den = 0.f;
  for (int t = 0; t < Tn; ++t) den += expf(s_sc[t] - mx);
  __syncthreads();
  for (int t = a; t < Tn; t += blockDim.x) {
    const float c = expf(s_sc[t] - mx) / den;
    s_sc[t] = c;
    coef[(size_t)b * Tn + t] = c;
  }
  __syncthreads();
  if (a < A) {
    float acc = 0.f;
    for (int t = 0; t < n_frames && t < Tn; ++t)
      if (b < frame_off[t + 1] - frame_off[t])
        acc = fmaf(s_sc[t], ElemOps<T>::load(f[(size_t)(frame_off[t] + b) * ld + a]), acc);
    ctxt[(size_t)b * A + a] = acc;
  }
}

template <typename T>
__global__ void temporal_attn_packed_bwd_kernel(const T* __restrict__ f, int ld, const int* __restrict__ frame_off, int n_frames,
                                                const float* __restrict__ w, const float* __restrict__ coef,
                                                const float* __restrict__ dctxt, T* __restrict__ df, float* __restrict__ dw_part,
                                                float* __restrict__ db_part, int Tn, int A, float grad_scale) {
  extern __shared__ float s_mem[];            // [T] g / dscore*valid, [T] coef, [16] scratch
  float* s_g = s_mem;
  float* s_c = s_mem + Tn;
  float* s_tmp = s_mem + 2 * Tn;
  const int b = blockIdx.x, a = threadIdx.x;
  const int nw = blockDim.x >> 6;
  const float da = a < A ? dctxt[(size_t)b * A + a] : 0.f;
  const float wa = a < A ? w[a] : 0.f;
  for (int t = 0; t < Tn; ++t) {
    const bool valid = t < n_frames && b < frame_off[t + 1] - frame_off[t];
    float v = 0.f;
    if (valid && a < A) v = da * ElemOps<T>::load(f[(size_t)(frame_off[t] + b) * ld + a]);
    const float g = block_reduce_sum(v, s_tmp, nw);
    if (a == 0) {
      s_g[t] = g;
      s_c[t] = coef[(size_t)b * Tn + t];
    }
  }
  __syncthreads();
  float dotcg = 0.f;
  for (int t = 0; t < Tn; ++t) dotcg += s_c[t] * s_g[t];
  __syncthreads();
  if (a == 0) {
    float dbsum = 0.f;
    for (int t = 0; t < Tn; ++t) {
      const bool valid = t < n_frames && b < frame_off[t + 1] - frame_off[t];
      const float ds = valid ? s_c[t] * (s_g[t] - dotcg) : 0.f;
      s_g[t] = ds;
      dbsum += ds;
    }
    db_part[b] = dbsum;
  }
  __syncthreads();
  float dwa = 0.f;
  for (int t = 0; t < n_frames && t < Tn; ++t) {
    if (b >= frame_off[t + 1] - frame_off[t]) continue;
    const size_t row = (size_t)(frame_off[t] + b) * ld;
    if (a < A) {
      const float fv = ElemOps<T>::load(f[row + a]);
      df[row + a] = ElemOps<T>::store(grad_scale * (s_c[t] * da + s_g[t] * wa));
      dwa = fmaf(s_g[t], fv, dwa);
    }
    for (int c = A + a; c < ld; c += blockDim.x) df[row + c] = ElemOps<T>::store(0.f);     // padding columns of the GEMM operand
  }
  if (a < A) dw_part[(size_t)b * A + a] = dwa;
}

}  // namespace

extern "C" int vnqa_temporal_attn_packed_fwd(const void* f, int32_t ld, int32_t dtype, const int32_t* frame_off,
                                             int32_t n_frames, const float* w, const float* bias, float* coef, float* ctxt,
                                             int32_t b, int32_t t, int32_t a, void* stream) {
  VNQA_CHECK_ARG(f && frame_off && w && bias && coef && ctxt, "temporal_attn_packed_fwd: null pointer");
  VNQA_CHECK_ARG(b > 0 && t > 0 && a > 0 && a <= 1024 && ld >= a && n_frames > 0 && n_frames <= t,
                 "temporal_attn_packed_fwd: need 0 < a <= 1024, ld >= a, 0 < n_frames <= t");
  const int threads = (a + 63) / 64 * 64;
  const size_t lds = (t + 16) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(temporal_attn_packed_fwd_kernel<vnqa_bf16>, dim3(b), dim3(threads), lds, st, (const vnqa_bf16*)f, ld,
                       frame_off, n_frames, w, bias, coef, ctxt, t, a);
  else if (dtype == VNQA_F32)
    hipLaunchKernelGGL(temporal_attn_packed_fwd_kernel<float>, dim3(b), dim3(threads), lds, st, (const float*)f, ld, frame_off,
                       n_frames, w, bias, coef, ctxt, t, a);
  else {
    vnqa_set_error("temporal_attn_packed_fwd: bad dtype %d", dtype);
    return VNQA_ERR_INVALID_ARG;
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_temporal_attn_packed_bwd(const void* f, int32_t ld, int32_t dtype, const int32_t* frame_off,
                                             int32_t n_frames, const float* w, const float* coef, const float* dctxt, void* df,
                                             float* dw_part, float* db_part, int32_t b, int32_t t, int32_t a, float grad_scale,
                                             void* stream) {
  VNQA_CHECK_ARG(f && frame_off && w && coef && dctxt && df && dw_part && db_part, "temporal_attn_packed_bwd: null pointer");
  VNQA_CHECK_ARG(b > 0 && t > 0 && a > 0 && a <= 1024 && ld >= a && n_frames > 0 && n_frames <= t,
                 "temporal_attn_packed_bwd: need 0 < a <= 1024, ld >= a, 0 < n_frames <= t");
  const int threads = (a + 63) / 64 * 64;
  const size_t lds = (2 * t + 16) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(temporal_attn_packed_bwd_kernel<vnqa_bf16>, dim3(b), dim3(threads), lds, st, (const vnqa_bf16*)f, ld,
                       frame_off, n_frames, w, coef, dctxt, (vnqa_bf16*)df, dw_part, db_part, t, a, grad_scale);
  else if (dtype == VNQA_F32)
    hipLaunchKernelGGL(temporal_attn_packed_bwd_kernel<float>, dim3(b), dim3(threads), lds, st, (const float*)f, ld, frame_off,
                       n_frames, w, coef, dctxt, (float*)df, dw_part, db_part, t, a, grad_scale);
  else {
    vnqa_set_error("temporal_attn_packed_bwd: bad dtype %d", dtype);
    return VNQA_ERR_INVALID_ARG;
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_temporal_attn_fwd(const float* feat, const float* valid, const float* mask, const float* w,
                                      const float* bias, float* coef, float* ctxt, int32_t b, int32_t t, int32_t a,
                                      void* stream) {
  VNQA_CHECK_ARG(feat && valid && mask && w && bias && coef && ctxt, "temporal_attn_fwd: null pointer");
  VNQA_CHECK_ARG(b > 0 && t > 0 && a > 0 && a <= 1024, "temporal_attn_fwd: need 0 < a <= 1024");
  const int threads = (a + 63) / 64 * 64;
  hipLaunchKernelGGL(temporal_attn_fwd_kernel, dim3(b), dim3(threads), (t + 16) * sizeof(float), (hipStream_t)stream,
                     feat, valid, mask, w, bias, coef, ctxt, t, a);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_temporal_attn_bwd(const float* feat, const float* valid, const float* w, const float* coef,
                                      const float* dctxt, float* dfeat, float* dw_part, float* db_part, int32_t b,
                                      int32_t t, int32_t a, void* stream) {
  VNQA_CHECK_ARG(feat && valid && w && coef && dctxt && dfeat && dw_part && db_part, "temporal_attn_bwd: null pointer");
  VNQA_CHECK_ARG(b > 0 && t > 0 && a > 0 && a <= 1024, "temporal_attn_bwd: need 0 < a <= 1024");
  const int threads = (a + 63) / 64 * 64;
  hipLaunchKernelGGL(temporal_attn_bwd_kernel, dim3(b), dim3(threads), (2 * t + 16) * sizeof(float), (hipStream_t)stream,
                     feat, valid, w, coef, dctxt, dfeat, dw_part, db_part, t, a);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
