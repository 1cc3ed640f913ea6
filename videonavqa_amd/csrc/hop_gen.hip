// hop_gen.hip — the multi-hop FiLM generator of TimeMultiHopFiLMPretrainedStem (models/time_multi_hop_pt_stem.py:124-184) on the
// packed image list, fp32:
//   compute_film_encoding  : x = encoder_norm(last-token LSTM state), h = x repeated over the words            (:146-158)
//   decode_to_film_values  : p = h (.) rnn_states ; coefs = softmax_words(fc_hidden_attn(p)) — padding words NOT masked —
//                            h = coefs^T p ; film = decoder_norm(fc_attn_out(h))                               (:165-184)
// The reference evaluates this per frame on [ct_B, Lmax, H] tensors; here every valid (frame, sample) pair is one image and
// the per-frame LSTM states are read straight from the persistent LSTM chain's output rows (no gathered [n_img, Lmax, H] copy):
// image i's words are rows base_row[i] .. base_row[i] + qlen[i] - 1 of hs, words >= qlen[i] are zero states.
//   layernorm_fwd / _bwd  : LayerNorm over the last dimension of (optionally row-gathered) fp32 rows; the backward's
//                           d gamma / d beta are per-column sums over the rows in row order (deterministic)
//   hop_fwd / hop_bwd     : one workgroup per image, one thread per hidden unit, the words' products in registers
//   scatter_add_rows      : dst[rows[r]] += src[r] (unique rows): the adjoint of the row gather
// fc_attn_out stays on vnqa_sgemm.
#include "vnqa_common.h"

namespace {

constexpr int LN_T = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {      // blockDim.x == LN_T (4 waves)
  v = wave_reduce_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(LN_T) layernorm_fwd_kernel(const float* __restrict__ x, const int* __restrict__ rows,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ y, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int n, float eps) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* xr = x + (size_t)(rows ? rows[r] : r) * n;
  float s = 0.f;
  for (int c = threadIdx.x; c < n; c += LN_T) s += xr[c];
  const float m = block_sum(s, red) / n;
  float q = 0.f;
  for (int c = threadIdx.x; c < n; c += LN_T) { const float d = xr[c] - m; q += d * d; }
  const float rs = rsqrtf(block_sum(q, red) / n + eps);          // biased variance, as nn.LayerNorm
  for (int c = threadIdx.x; c < n; c += LN_T) y[(size_t)r * n + c] = (xr[c] - m) * rs * gamma[c] + beta[c];
  if (threadIdx.x == 0) { mean[r] = m; rstd[r] = rs; }
}

// d x[r] = rstd (g - mean(g) - xhat mean(g xhat)),  g = dy gamma
__global__ void __launch_bounds__(LN_T) layernorm_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const int* __restrict__ rows, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                float* __restrict__ dx, int n) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* xr = x + (size_t)(rows ? rows[r] : r) * n;
  const float* dyr = dy + (size_t)r * n;
  const float m = mean[r], rs = rstd[r];
  float s1 = 0.f, s2 = 0.f;
  for (int c = threadIdx.x; c < n; c += LN_T) {
    const float g = dyr[c] * gamma[c];
    s1 += g;
    s2 += g * (xr[c] - m) * rs;
  }
  const float a = block_sum(s1, red) / n;
  const float b = block_sum(s2, red) / n;
  for (int c = threadIdx.x; c < n; c += LN_T) {
    const float xh = (xr[c] - m) * rs;
    dx[(size_t)r * n + c] = rs * (dyr[c] * gamma[c] - a - xh * b);
  }
}

// d gamma[c] (+)= sum_r dy[r][c] xhat[r][c],  d beta[c] (+)= sum_r dy[r][c]  — one thread per column, rows in order
__global__ void __launch_bounds__(256) layernorm_bwd_gb_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const int* __restrict__ rows, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int n_rows, int n, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  float sg = 0.f, sb = 0.f;
  for (int r = 0; r < n_rows; ++r) {
    const float d = dy[(size_t)r * n + c];
    const float xv = x[(size_t)(rows ? rows[r] : r) * n + c];
    sg += d * (xv - mean[r]) * rstd[r];
    sb += d;
  }
  if (accumulate) { dgamma[c] += sg; dbeta[c] += sb; }
  else { dgamma[c] = sg; dbeta[c] = sb; }
}

__global__ void __launch_bounds__(256) scatter_add_rows_kernel(float* __restrict__ dst, const int* __restrict__ rows,
                                                               const float* __restrict__ src, int n_rows, int n) {
  const long long total = (long long)n_rows * n;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i - (long long)r * n);
    dst[(size_t)rows[r] * n + c] += src[i];
  }
}

constexpr int HOP_LMAX = 64;      // words a question can have here (the reference pads questions to 56, eval/utils.py:22)

// one workgroup per image, thread = hidden unit; p[l] = hv * states[l] kept in registers for both passes over the words
// (NT threads = the hidden size rounded up to whole waves; threads h >= H carry zeros through the reductions)
template <int NT>
__global__ void __launch_bounds__(NT) hop_fwd_kernel(const float* __restrict__ hv, const float* __restrict__ hs,
                                                     const int* __restrict__ base_row, const int* __restrict__ qlen,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     float* __restrict__ hv_out, float* __restrict__ coefs, int lmax, int H) {
  constexpr int NWV = NT / 64;
  __shared__ float sc[NWV][HOP_LMAX];
  __shared__ float cf[HOP_LMAX];
  const int img = blockIdx.x, h = threadIdx.x, lane = h & 63, wv = h >> 6;
  const int base = base_row[img], ql = qlen[img];
  const bool act = h < H;
  const float hvh = act ? hv[(size_t)img * H + h] : 0.f, wh = act ? w[h] : 0.f;
  float p[HOP_LMAX];
#pragma unroll
  for (int l = 0; l < HOP_LMAX; ++l) {
    p[l] = 0.f;
    if (l < lmax) {
      p[l] = (l < ql && act) ? hvh * hs[(size_t)(base + l) * H + h] : 0.f;
      const float s = wave_reduce_sum(wh * p[l]);
      if (lane == 0) sc[wv][l] = s;
    }
  }
  __syncthreads();
  if (h < 64) {        // softmax over the lmax words (NOT masked: padding words score `bias`), one wave
    float s = -INFINITY;
    if (h < lmax) {
      s = bias[0];
#pragma unroll
      for (int k = 0; k < NWV; ++k) s += sc[k][h];
    }
    const float mx = wave_reduce_max(s);
    const float e = h < lmax ? __expf(s - mx) : 0.f;
    const float den = wave_reduce_sum(e);
    if (h < lmax) {
      const float c = e / den;
      cf[h] = c;
      coefs[(size_t)img * lmax + h] = c;
    }
  }
  __syncthreads();
  float o = 0.f;
#pragma unroll
  for (int l = 0; l < HOP_LMAX; ++l)
    if (l < lmax) o += cf[l] * p[l];
  if (act) hv_out[(size_t)img * H + h] = o;
}

// d p[l] = c[l] dout + ds[l] w,  ds[l] = c[l] (g[l] - sum_k c[k] g[k]),  g[l] = dout . p[l]
// d hv = sum_l dp[l] st[l] ; d st[l] = dp[l] hv (accumulated into d hs) ; d w (per image) = sum_l ds[l] p[l] ; d bias = sum_l ds[l] = 0
template <int NT>
__global__ void __launch_bounds__(NT) hop_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ hv,
                                                     const float* __restrict__ hs, const int* __restrict__ base_row,
                                                     const int* __restrict__ qlen, const float* __restrict__ w,
                                                     const float* __restrict__ coefs, float* __restrict__ dhv,
                                                     float* __restrict__ dhs, float* __restrict__ dw_img, int lmax, int H) {
  constexpr int NWV = NT / 64;
  __shared__ float gs[NWV][HOP_LMAX];
  __shared__ float dsl[HOP_LMAX];
  const int img = blockIdx.x, h = threadIdx.x, lane = h & 63, wv = h >> 6;
  const int base = base_row[img], ql = qlen[img];
  const bool act = h < H;
  const float hvh = act ? hv[(size_t)img * H + h] : 0.f, wh = act ? w[h] : 0.f, doh = act ? dout[(size_t)img * H + h] : 0.f;
  float st[HOP_LMAX];
#pragma unroll
  for (int l = 0; l < HOP_LMAX; ++l) {
    st[l] = 0.f;
    if (l < lmax) {
      st[l] = (l < ql && act) ? hs[(size_t)(base + l) * H + h] : 0.f;
      const float g = wave_reduce_sum(doh * hvh * st[l]);
      if (lane == 0) gs[wv][l] = g;
    }
  }
  __syncthreads();
  if (h < 64) {
    float g = 0.f, c = 0.f;
    if (h < lmax) {
      c = coefs[(size_t)img * lmax + h];
#pragma unroll
      for (int k = 0; k < NWV; ++k) g += gs[k][h];
    }
    const float dot = wave_reduce_sum(c * g);
    if (h < lmax) dsl[h] = c * (g - dot);
  }
  __syncthreads();
  float a_hv = 0.f, a_w = 0.f;
#pragma unroll
  for (int l = 0; l < HOP_LMAX; ++l) {
    if (l < lmax) {
      const float c = coefs[(size_t)img * lmax + l];
      const float dp = c * doh + dsl[l] * wh;
      a_hv += dp * st[l];
      a_w += dsl[l] * hvh * st[l];
      if (l < ql && act) dhs[(size_t)(base + l) * H + h] += dp * hvh;       // rows of one image belong to it alone
    }
  }
  if (act) {
    dhv[(size_t)img * H + h] = a_hv;
    dw_img[(size_t)img * H + h] = a_w;
  }
}

}  // namespace

extern "C" int vnqa_layernorm_fwd(const float* x, const int32_t* rows, const float* gamma, const float* beta, float* y,
                                  float* mean, float* rstd, int32_t n_rows, int32_t n, float eps, void* stream) {
  VNQA_CHECK_ARG(x && gamma && beta && y && mean && rstd, "layernorm_fwd: null pointer");
  VNQA_CHECK_ARG(n_rows > 0 && n > 0, "layernorm_fwd: empty problem");
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(n_rows), dim3(LN_T), 0, (hipStream_t)stream, x, rows, gamma, beta, y, mean, rstd,
                     n, eps);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_layernorm_bwd(const float* dy, const float* x, const int32_t* rows, const float* mean, const float* rstd,
                                  const float* gamma, float* dx, float* dgamma, float* dbeta, int32_t n_rows, int32_t n,
                                  int32_t accumulate, void* stream) {
  VNQA_CHECK_ARG(dy && x && mean && rstd && gamma, "layernorm_bwd: null pointer");
  VNQA_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma / dbeta must come together");
  VNQA_CHECK_ARG(n_rows > 0 && n > 0, "layernorm_bwd: empty problem");
  hipStream_t st = (hipStream_t)stream;
  if (dx != nullptr)
    hipLaunchKernelGGL(layernorm_bwd_dx_kernel, dim3(n_rows), dim3(LN_T), 0, st, dy, x, rows, mean, rstd, gamma, dx, n);
  if (dgamma != nullptr)
    hipLaunchKernelGGL(layernorm_bwd_gb_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dy, x, rows, mean, rstd, dgamma, dbeta,
                       n_rows, n, accumulate);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_scatter_add_rows(float* dst, const int32_t* rows, const float* src, int32_t n_rows, int32_t n, void* stream) {
  VNQA_CHECK_ARG(dst && rows && src && n_rows > 0 && n > 0, "scatter_add_rows: bad arguments");
  long long g = ((long long)n_rows * n + 255) / 256;
  g = g > 2048 ? 2048 : g;
  hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, dst, rows, src, n_rows, n);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_hop_fwd(const float* hv, const float* hs, const int32_t* base_row, const int32_t* qlen, const float* w,
                            const float* bias, float* hv_out, float* coefs, int32_t n_img, int32_t lmax, int32_t hidden,
                            void* stream) {
  VNQA_CHECK_ARG(hv && hs && base_row && qlen && w && bias && hv_out && coefs, "hop_fwd: null pointer");
  VNQA_CHECK_ARG(n_img > 0 && lmax > 0 && lmax <= HOP_LMAX, "hop_fwd: lmax must be in 1..%d (got %d)", HOP_LMAX, lmax);
  VNQA_CHECK_ARG(hidden > 0 && hidden <= 256, "hop_fwd: hidden size %d not supported (1..256)", hidden);
  hipStream_t st = (hipStream_t)stream;
  if (hidden <= 64) hipLaunchKernelGGL(hop_fwd_kernel<64>, dim3(n_img), dim3(64), 0, st, hv, hs, base_row, qlen, w, bias, hv_out, coefs, lmax, hidden);
  else if (hidden <= 128) hipLaunchKernelGGL(hop_fwd_kernel<128>, dim3(n_img), dim3(128), 0, st, hv, hs, base_row, qlen, w, bias, hv_out, coefs, lmax, hidden);
  else hipLaunchKernelGGL(hop_fwd_kernel<256>, dim3(n_img), dim3(256), 0, st, hv, hs, base_row, qlen, w, bias, hv_out, coefs, lmax, hidden);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_hop_bwd(const float* dout, const float* hv, const float* hs, const int32_t* base_row, const int32_t* qlen,
                            const float* w, const float* coefs, float* dhv, float* dhs, float* dw_img, int32_t n_img,
                            int32_t lmax, int32_t hidden, void* stream) {
  VNQA_CHECK_ARG(dout && hv && hs && base_row && qlen && w && coefs && dhv && dhs && dw_img, "hop_bwd: null pointer");
  VNQA_CHECK_ARG(n_img > 0 && lmax > 0 && lmax <= HOP_LMAX, "hop_bwd: lmax must be in 1..%d (got %d)", HOP_LMAX, lmax);
  VNQA_CHECK_ARG(hidden > 0 && hidden <= 256, "hop_bwd: hidden size %d not supported (1..256)", hidden);
  hipStream_t st = (hipStream_t)stream;
  if (hidden <= 64) hipLaunchKernelGGL(hop_bwd_kernel<64>, dim3(n_img), dim3(64), 0, st, dout, hv, hs, base_row, qlen, w, coefs, dhv, dhs, dw_img, lmax, hidden);
  else if (hidden <= 128) hipLaunchKernelGGL(hop_bwd_kernel<128>, dim3(n_img), dim3(128), 0, st, dout, hv, hs, base_row, qlen, w, coefs, dhv, dhs, dw_img, lmax, hidden);
  else hipLaunchKernelGGL(hop_bwd_kernel<256>, dim3(n_img), dim3(256), 0, st, dout, hv, hs, base_row, qlen, w, coefs, dhv, dhs, dw_img, lmax, hidden);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
