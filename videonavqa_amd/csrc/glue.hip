// glue.hip — the small fp32 pieces between the MFMA kernels of the FiLM models, so that a training step enqueues
// no framework (ATen / rocBLAS) kernels for them:
//
//   sgemm          C = act(A' B + bias) for the sub-GFLOP products of the question path and the classifier
//                  (models/film_attn_pt_stem.py:179 FiLM generator Linear+ReLU, :293 LSTMCell input projection, :301
//                  out_linear, and their backward products), arbitrary strides (NN / NT / TN without copies), optional
//                  row gather on A, row scatter on C, ReLU-mask on A (backward of Linear+ReLU), accumulation into C
//   colsum         bias gradients
//   gather_rows    h_last = LSTM output at the last token of every repeat (:163-171)
//   embed_proj     embedding lookup fused with the LSTM input projection (:146 + the W_ih half of :160), and its
//                  backward through per-token gradient sums (deterministic: no atomics)
//   lstm_fold_dxg / lstm_wgrad_operands   the two data movements of the persistent LSTM's backward
//   ce_loss        CrossEntropyLoss forward + d logits (eval/q_and_v_eval.py:124), class weights, sum / mean
//   bn_running_update   the per-frame running-statistics EMA of bn_init (:211), frame by frame as the reference does
//
// All of it is latency-class work (a few microseconds per launch); plain FMA, exact fp32, fixed summation orders.
#include "vnqa_common.h"

namespace {

struct SgemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;     // [N] or null
  const float* a_mask;   // same indexing as A: A'(m,k) = A(m,k) * [a_mask(m,k) > 0]; or null
  const int* a_rows;     // [M]: physical row of A for logical row m (negative: a zero row); or null
  const int* c_rows;     // [M]: physical row of C for logical row m (negative: not written); or null
  long long a_rs, a_cs, b_rs, b_cs;   // element strides: A'(m,k) = A[row(m)*a_rs + k*a_cs], B(k,n) = B[k*b_rs + n*b_cs]
  int ldc;
  int M, N, K;
  int relu, accumulate;
};

// 64x64 output tile, 16-deep K chunks through LDS, 256 threads x (4x4) outputs
__global__ void __launch_bounds__(256) sgemm_kernel(const SgemmArgs p) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  // loader roles: thread -> (k = tid & 15, 4 rows m = (tid >> 4) + 16 r) for A, (k = tid >> 4 .. , n) for B
  const int lk = threadIdx.x & 15, lr = threadIdx.x >> 4;
  long long a_base[4];
  bool a_ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = m0 + lr + 16 * r;
    int row = m;
    a_ok[r] = m < p.M;
    if (a_ok[r] && p.a_rows != nullptr) {
      row = p.a_rows[m];
      a_ok[r] = row >= 0;
    }
    a_base[r] = (long long)row * p.a_rs;
  }
  for (int k0 = 0; k0 < p.K; k0 += 16) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + lk;
      float v = 0.f;
      if (a_ok[r] && k < p.K) {
        const long long off = a_base[r] + (long long)k * p.a_cs;
        v = p.A[off];
        if (p.a_mask != nullptr && !(p.a_mask[off] > 0.f)) v = 0.f;
      }
      As[lk][lr + 16 * r] = v;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = k0 + lr, n = n0 + lk + 16 * r;
      Bs[lr][lk + 16 * r] = (k < p.K && n < p.N) ? p.B[(long long)k * p.b_rs + (long long)n * p.b_cs] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty * 4 + i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx * 4 + j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= p.M) continue;
    int row = m;
    if (p.c_rows != nullptr) {
      row = p.c_rows[m];
      if (row < 0) continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= p.N) continue;
      float v = acc[i][j];
      if (p.bias != nullptr) v += p.bias[n];
      float* dst = p.C + (long long)row * p.ldc + n;
      if (p.accumulate) v += *dst;
      if (p.relu) v = fmaxf(v, 0.f);
      *dst = v;
    }
  }
}

// out[n] = sum_m x[m*ld + n] * [mask > 0]   (one thread per column, 256 columns per block; rows in a fixed order)
template <typename T>
__global__ void colsum_kernel(const T* __restrict__ x, const float* __restrict__ mask, float* __restrict__ out, int rows,
                              int cols, int ld) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= cols) return;
  float s = 0.f;
  for (int m = 0; m < rows; ++m) {
    const float v = ElemOps<T>::load(x[(size_t)m * ld + n]);
    s += (mask == nullptr || mask[(size_t)m * ld + n] > 0.f) ? v : 0.f;
  }
  out[n] = s;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ rows, float* __restrict__ dst,
                                   int n_rows, int cols) {
  const int r = blockIdx.x;
  const int row = rows[r];
  for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(size_t)r * cols + c] = row >= 0 ? src[(size_t)row * cols + c] : 0.f;
}

// xg[b][pos][j] = b_ih[j] + b_hh[j] + sum_e W_ih[j][e] * embed[token(b,pos)][e]
__global__ void embed_proj_fwd_kernel(const long long* __restrict__ tokens, const int* __restrict__ row_perm,
                                      const float* __restrict__ embed, const float* __restrict__ w_ih,
                                      const float* __restrict__ b_ih, const float* __restrict__ b_hh, float* __restrict__ xg,
                                      int Lq, int E, int G, int V) {
  extern __shared__ float s_e[];
  const int b = blockIdx.y, pos = blockIdx.x;
  const int src_b = row_perm != nullptr ? row_perm[b] : b;
  long long tok = tokens[(size_t)src_b * Lq + pos];
  tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
  for (int e = threadIdx.x; e < E; e += blockDim.x) s_e[e] = embed[(size_t)tok * E + e];
  __syncthreads();
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    const float* w = w_ih + (size_t)j * E;
    float acc = b_ih[j] + b_hh[j];
    for (int e = 0; e < E; ++e) acc = fmaf(w[e], s_e[e], acc);
    xg[((size_t)b * Lq + pos) * G + j] = acc;
  }
}

// dsum[v][j] = sum over the (b, pos) whose token is v of dxg[b][pos][j]   (positions in a fixed order)
__global__ void token_dsum_kernel(const long long* __restrict__ tokens, const int* __restrict__ row_perm,
                                  const float* __restrict__ dxg, float* __restrict__ dsum, int B, int Lq, int G) {
  const int v = blockIdx.x;
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) {
      const int src_b = row_perm != nullptr ? row_perm[b] : b;
      for (int pos = 0; pos < Lq; ++pos)
        if (tokens[(size_t)src_b * Lq + pos] == v) s += dxg[((size_t)b * Lq + pos) * G + j];
    }
    dsum[(size_t)v * G + j] = s;
  }
}

// dxg[b][pos][j] = sum_rep dgates[b][rep*ql + pos][j]  (pos < ql), 0 otherwise
__global__ void lstm_fold_dxg_kernel(const float* __restrict__ dgates, const int* __restrict__ q_lens, float* __restrict__ dxg,
                                     int Lq, int S, int G, int n_rep) {
  const int b = blockIdx.y, pos = blockIdx.x;
  const int ql = q_lens[b];
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    float s = 0.f;
    if (pos < ql)
      for (int r = 0; r < n_rep; ++r) s += dgates[((size_t)b * S + (size_t)r * ql + pos) * G + j];
    dxg[((size_t)b * Lq + pos) * G + j] = s;
  }
}

// operands of dW_hh = sum_{b,t} dgates[b,t]^T h_{t-1}[b] for the MFMA GEMM: a = dgates, hp[b][t] = t ? hs[b][t-1] : h0[b],
// both converted to the GEMM's element type in one pass
template <typename T>
__global__ void lstm_wgrad_operands_kernel(const float* __restrict__ dgates, const float* __restrict__ hs,
                                           const float* __restrict__ h0, T* __restrict__ a, T* __restrict__ hp, int S, int H) {
  const int b = blockIdx.y, t = blockIdx.x;
  const size_t row = (size_t)b * S + t;
  for (int j = threadIdx.x; j < 4 * H; j += blockDim.x) a[row * 4 * H + j] = ElemOps<T>::store(dgates[row * 4 * H + j]);
  for (int j = threadIdx.x; j < H; j += blockDim.x)
    hp[row * H + j] = ElemOps<T>::store(t == 0 ? h0[(size_t)b * H + j] : hs[(row - 1) * H + j]);
}

// CrossEntropyLoss(weight, reduction) over logits [B][K]: loss scalar and d loss / d logits; one workgroup
__global__ void ce_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ ys, const int* __restrict__ row_perm,
                               const float* __restrict__ weight, float* __restrict__ loss, float* __restrict__ dlogits, int B,
                               int K, int mean) {
  __shared__ float s_loss[64], s_w[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (threadIdx.x < 64) s_loss[threadIdx.x] = s_w[threadIdx.x] = 0.f;
  __syncthreads();
  // one wave per sample
  float my_loss = 0.f, my_w = 0.f;
  for (int b = wave; b < B; b += nw) {
    const float* lg = logits + (size_t)b * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, lg[k]);
    mx = wave_reduce_max(mx);
    float den = 0.f;
    for (int k = lane; k < K; k += 64) den += expf(lg[k] - mx);
    den = wave_reduce_sum(den);
    const long long y = ys[row_perm != nullptr ? row_perm[b] : b];
    const float wy = weight != nullptr ? weight[y] : 1.f;
    const float lse = mx + logf(den);
    for (int k = lane; k < K; k += 64) {
      const float pk = expf(lg[k] - lse);
      dlogits[(size_t)b * K + k] = wy * (pk - (k == y ? 1.f : 0.f));      // scaled by 1 / sum(w) below for 'mean'
    }
    if (lane == 0) {
      my_loss += wy * (lse - lg[y]);
      my_w += wy;
    }
  }
  if (lane == 0) {
    s_loss[wave] = my_loss;
    s_w[wave] = my_w;
  }
  __syncthreads();
  float tot = 0.f, totw = 0.f;
  for (int w = 0; w < nw; ++w) {
    tot += s_loss[w];
    totw += s_w[w];
  }
  const float scale = mean ? 1.f / totw : 1.f;
  if (threadIdx.x == 0) loss[0] = tot * scale;
  if (mean)
    for (int i = threadIdx.x; i < B * K; i += blockDim.x) dlogits[i] *= scale;     // (after the barrier: every sample is written)
}

// running statistics advanced frame by frame (momentum m, unbiased variance), as nn.BatchNorm2d does per call
__global__ void bn_running_update_kernel(const float* __restrict__ mean, const float* __restrict__ var,
                                         const int* __restrict__ frame_off, float* __restrict__ rmean, float* __restrict__ rvar,
                                         int n_frames, int S, int C, int ld, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = rmean[c], rv = rvar[c];
  for (int f = 0; f < n_frames; ++f) {
    const float cnt = (float)((frame_off[f + 1] - frame_off[f]) * S);
    const float unbias = cnt / fmaxf(cnt - 1.f, 1.f);
    rm = (1.f - momentum) * rm + momentum * mean[(size_t)f * ld + c];
    rv = (1.f - momentum) * rv + momentum * var[(size_t)f * ld + c] * unbias;
  }
  rmean[c] = rm;
  rvar[c] = rv;
}

}  // namespace

extern "C" int vnqa_sgemm(const float* a, const float* b, float* c, const float* bias, const float* a_mask,
                          const int32_t* a_rows, const int32_t* c_rows, int64_t a_rs, int64_t a_cs, int64_t b_rs,
                          int64_t b_cs, int32_t ldc, int32_t m, int32_t n, int32_t k, int32_t relu, int32_t accumulate,
                          void* stream) {
  VNQA_CHECK_ARG(a && b && c && m > 0 && n > 0 && k > 0 && ldc >= n, "sgemm: bad arguments (m=%d n=%d k=%d ldc=%d)", m, n, k, ldc);
  SgemmArgs p;
  p.A = a; p.B = b; p.C = c; p.bias = bias; p.a_mask = a_mask; p.a_rows = a_rows; p.c_rows = c_rows;
  p.a_rs = a_rs; p.a_cs = a_cs; p.b_rs = b_rs; p.b_cs = b_cs; p.ldc = ldc; p.M = m; p.N = n; p.K = k;
  p.relu = relu; p.accumulate = accumulate;
  hipLaunchKernelGGL(sgemm_kernel, dim3((n + 63) / 64, (m + 63) / 64), dim3(256), 0, (hipStream_t)stream, p);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_colsum(const void* x, const float* mask, float* out, int32_t rows, int32_t cols, int32_t ld, int32_t dtype,
                           void* stream) {
  VNQA_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_F32 || (dtype == VNQA_BF16 && mask == nullptr), "colsum: dtype must be f32, or bf16 without a mask");
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(colsum_kernel<vnqa_bf16>, dim3((cols + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const vnqa_bf16*)x,
                       mask, out, rows, cols, ld);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((cols + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float*)x, mask, out,
                       rows, cols, ld);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_gather_rows(const float* src, const int32_t* rows, float* dst, int32_t n_rows, int32_t cols, void* stream) {
  VNQA_CHECK_ARG(src && rows && dst && n_rows > 0 && cols > 0, "gather_rows: bad arguments");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n_rows), dim3(cols >= 256 ? 256 : 64), 0, (hipStream_t)stream, src, rows, dst,
                     n_rows, cols);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_embed_proj_fwd(const int64_t* tokens, const int32_t* row_perm, const float* embed, const float* w_ih,
                                   const float* b_ih, const float* b_hh, float* xg, int32_t b, int32_t lq, int32_t e,
                                   int32_t g, int32_t vocab, void* stream) {
  VNQA_CHECK_ARG(tokens && embed && w_ih && b_ih && b_hh && xg && b > 0 && lq > 0 && e > 0 && g > 0 && vocab > 0,
                 "embed_proj_fwd: bad arguments");
  const int threads = g >= 512 ? 512 : (g + 63) / 64 * 64;
  hipLaunchKernelGGL(embed_proj_fwd_kernel, dim3(lq, b), dim3(threads), e * sizeof(float), (hipStream_t)stream,
                     (const long long*)tokens, row_perm, embed, w_ih, b_ih, b_hh, xg, lq, e, g, vocab);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_token_dsum(const int64_t* tokens, const int32_t* row_perm, const float* dxg, float* dsum, int32_t b,
                               int32_t lq, int32_t g, int32_t vocab, void* stream) {
  VNQA_CHECK_ARG(tokens && dxg && dsum && b > 0 && lq > 0 && g > 0 && vocab > 0, "token_dsum: bad arguments");
  const int threads = g >= 512 ? 512 : (g + 63) / 64 * 64;
  hipLaunchKernelGGL(token_dsum_kernel, dim3(vocab), dim3(threads), 0, (hipStream_t)stream, (const long long*)tokens, row_perm,
                     dxg, dsum, b, lq, g);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_lstm_fold_dxg(const float* dgates, const int32_t* q_lens, float* dxg, int32_t b, int32_t lq, int32_t s,
                                  int32_t hidden, int32_t n_rep, void* stream) {
  VNQA_CHECK_ARG(dgates && q_lens && dxg && b > 0 && lq > 0 && s > 0 && hidden > 0 && n_rep > 0, "lstm_fold_dxg: bad arguments");
  const int g = 4 * hidden;
  hipLaunchKernelGGL(lstm_fold_dxg_kernel, dim3(lq, b), dim3(g >= 512 ? 512 : (g + 63) / 64 * 64), 0, (hipStream_t)stream,
                     dgates, q_lens, dxg, lq, s, g, n_rep);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_lstm_wgrad_operands(const float* dgates, const float* hs, const float* h0, void* a, void* hp, int32_t b,
                                        int32_t s, int32_t hidden, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(dgates && hs && h0 && a && hp && b > 0 && s > 0 && hidden > 0, "lstm_wgrad_operands: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "lstm_wgrad_operands: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(lstm_wgrad_operands_kernel<vnqa_bf16>, dim3(s, b), dim3(256), 0, st, dgates, hs, h0, (vnqa_bf16*)a,
                       (vnqa_bf16*)hp, s, hidden);
  else
    hipLaunchKernelGGL(lstm_wgrad_operands_kernel<float>, dim3(s, b), dim3(256), 0, st, dgates, hs, h0, (float*)a, (float*)hp,
                       s, hidden);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_ce_loss(const float* logits, const int64_t* ys, const int32_t* row_perm, const float* weight, float* loss,
                            float* dlogits, int32_t b, int32_t k, int32_t mean, void* stream) {
  VNQA_CHECK_ARG(logits && ys && loss && dlogits && b > 0 && k > 0, "ce_loss: bad arguments");
  hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(b >= 8 ? 512 : 256), 0, (hipStream_t)stream, logits, (const long long*)ys,
                     row_perm, weight, loss, dlogits, b, k, mean);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_bn_running_update(const float* mean, const float* var, const int32_t* frame_off, float* running_mean,
                                      float* running_var, int32_t n_frames, int32_t pixels_per_image, int32_t c, int32_t ld,
                                      float momentum, void* stream) {
  VNQA_CHECK_ARG(mean && var && frame_off && running_mean && running_var && n_frames > 0 && c > 0 && ld >= c,
                 "bn_running_update: bad arguments");
  hipLaunchKernelGGL(bn_running_update_kernel, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream, mean, var, frame_off,
                     running_mean, running_var, n_frames, pixels_per_image, c, ld, momentum);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
