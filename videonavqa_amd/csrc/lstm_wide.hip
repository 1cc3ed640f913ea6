// lstm_wide.hip — packed-sequence LSTM for WIDE hidden states (H = 512 / 1536), forward and BPTT.
//
// MACNetwork runs a bidirectional question LSTM (embed_hidden -> dim, models/mac.py:185-186,211) and a
// tail LSTM over the per-frame outputs (3*dim -> 3*dim, :193,249-251).  With dim = 512 the recurrent
// matrix W_hh is 4 MB / 37.7 MB: it cannot live in one workgroup's registers like the H <= 256
// persistent kernel (lstm.hip), so here ONE LAUNCH PER TIME STEP streams W_hh through the whole chip:
// a workgroup owns one hidden unit (its 4 gate rows), the 64 lanes of a wave share one row and split K,
// every lane keeps one accumulator per sample, and the cell update of those units runs in the same launch.  A step is
// weight-bandwidth-bound (4H*H*4 bytes from L2/MALL per step), the sequence is a chain of T such
// launches enqueued back to back on the caller's stream by one C call.  (One wave per matrix row: H = 1536 gives
// 1536 / 384 workgroups per step forward / backward; 16 rows per workgroup left the backward at 96 workgroups
// and 3x slower.)
//
// Packed batches (nn.utils.rnn.pack_padded_sequence semantics, :210,249): samples are sorted by length,
// batch_sizes[t] = number of samples with length > t (non-increasing), every sample starts from
// (h0, c0) (zeros when null) at its own first step.  `reverse` walks t = T-1 .. 0 (the second direction
// of a bidirectional nn.LSTM): sample b then starts at t = len_b - 1.
// Time-major buffers: xg [T][B][4H] (x W_ih^T + b_ih + b_hh, gate order i,f,g,o), hs/cs [T][B][H],
// gates [T][B][4H] (activated), rows of inactive (t, b) are left untouched (callers pass zeroed buffers).
//
// Exact fp32.  BPTT: step t computes dh_t = dhs[t] + dgates_succ W_hh (the same row-split matvec on the
// TRANSPOSED matrix, passed in by the caller) and the gate gradients of its units; weight gradients are
// two GEMMs over all steps afterwards (vnqa_gemm_tn), done by the caller.
#include "vnqa_common.h"

namespace {

constexpr int NB = 8;        // samples per workgroup
constexpr int KL = 64;       // lanes sharing one matrix row (one wave)
constexpr int ROWS = 4;      // matrix rows per workgroup (256 threads)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// acc[j] += sum_k wrow[k] * x[j][k], k split over the KL lanes of a row group; x rows `ldx` apart.
// Rows of samples past `nvalid` alias sample 0 (their sums are ignored by the caller).
__device__ __forceinline__ void row_matvec(const float* __restrict__ wrow, const float* __restrict__ x, size_t ldx,
                                           int K, int kl, int nvalid, float (&acc)[NB]) {
  const float* xr[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) xr[j] = x + (size_t)(j < nvalid ? j : 0) * ldx;
  for (int k = kl * 4; k < K; k += KL * 4) {
    const float4 w = *(const float4*)(wrow + k);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const float4 v = *(const float4*)(xr[j] + k);
      acc[j] = fmaf(w.x, v.x, fmaf(w.y, v.y, fmaf(w.z, v.z, fmaf(w.w, v.w, acc[j]))));
    }
  }
#pragma unroll
  for (int j = 0; j < NB; ++j) {
#pragma unroll
    for (int m = KL / 2; m >= 1; m >>= 1) acc[j] += __shfl_xor(acc[j], m, 64);
  }
}

struct WideFwd {
  const float* xg_t;     // [B][4H] of this step
  const float* w_hh;     // [4H][H]
  const float* h_prev;   // [B][H] of the predecessor step (samples < n_prev), else h0 / zero
  const float* c_prev;
  const float* h0;       // [B][H] or null
  const float* c0;
  float* hs_t;           // [B][H]
  float* cs_t;
  float* gates_t;        // [B][4H]
  int H, n_act, n_prev;
};

__global__ void __launch_bounds__(256) lstm_wide_fwd_step(const WideFwd p) {
  __shared__ float pre[ROWS][NB];
  const int H = p.H;
  const int r = threadIdx.x / KL, kl = threadIdx.x % KL;
  const int gate = r, unit = blockIdx.x;
  const int b0 = blockIdx.y * NB;
  const int nb = min(NB, p.n_act - b0);
  // samples of this chunk with a predecessor state form a prefix (batch sizes are non-increasing)
  const int nprev = max(0, min(nb, p.n_prev - b0));
  float acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = 0.f;
  const float* wrow = p.w_hh + ((size_t)gate * H + unit) * H;
  if (nprev > 0) row_matvec(wrow, p.h_prev + (size_t)b0 * H, H, H, kl, nprev, acc);
  if (nprev < nb && p.h0 != nullptr) {
    float acc0[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc0[j] = 0.f;
    row_matvec(wrow, p.h0 + (size_t)(b0 + nprev) * H, H, H, kl, nb - nprev, acc0);
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j >= nprev && j < nb) acc[j] = acc0[j - nprev];
  } else {
#pragma unroll
    for (int j = 0; j < NB; ++j)
      if (j >= nprev) acc[j] = 0.f;
  }
  if (kl == 0) {
#pragma unroll
    for (int j = 0; j < NB; ++j) pre[r][j] = acc[j];
  }
  __syncthreads();
  if (threadIdx.x < NB) {
    const int j = threadIdx.x;
    if (j < nb) {
      const int b = b0 + j, u = blockIdx.x;
      const float* xg = p.xg_t + (size_t)b * 4 * H;
      const float gi = sigm(pre[0][j] + xg[u]);
      const float gf = sigm(pre[1][j] + xg[H + u]);
      const float gg = tanhf(pre[2][j] + xg[2 * H + u]);
      const float go = sigm(pre[3][j] + xg[3 * H + u]);
      const float cp = j < nprev ? p.c_prev[(size_t)b * H + u] : (p.c0 ? p.c0[(size_t)b * H + u] : 0.f);
      const float c = gf * cp + gi * gg;
      p.cs_t[(size_t)b * H + u] = c;
      p.hs_t[(size_t)b * H + u] = go * tanhf(c);
      float* g = p.gates_t + (size_t)b * 4 * H;
      g[u] = gi;
      g[H + u] = gf;
      g[2 * H + u] = gg;
      g[3 * H + u] = go;
    }
  }
}

struct WideBwd {
  const float* w_hh_t;    // [H][4H]  (W_hh transposed)
  const float* dg_succ;   // [B][4H] gate gradients of the successor step in the chain (samples < n_succ)
  const float* dhs_t;     // [B][H] external gradient on this step's h
  const float* gates_t;   // [B][4H]
  const float* cs_t;      // [B][H]
  const float* c_prev;    // [B][H] of the predecessor step (samples < n_prev), else c0 / zero
  const float* c0;
  float* dc;              // [B][H] running dL/dc, in/out
  float* dgates_t;        // [B][4H] out
  int H, n_act, n_prev, n_succ;
};

__global__ void __launch_bounds__(256) lstm_wide_bwd_step(const WideBwd p) {
  __shared__ float dhr[ROWS][NB];
  const int H = p.H;
  const int r = threadIdx.x / KL, kl = threadIdx.x % KL;
  const int unit = blockIdx.x * ROWS + r;
  const int b0 = blockIdx.y * NB;
  const int nb = min(NB, p.n_act - b0);
  const int nsucc = max(0, min(nb, p.n_succ - b0));
  float acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) acc[j] = 0.f;
  if (nsucc > 0 && unit < H)
    row_matvec(p.w_hh_t + (size_t)unit * 4 * H, p.dg_succ + (size_t)b0 * 4 * H, (size_t)4 * H, 4 * H, kl, nsucc, acc);
  if (kl == 0) {
#pragma unroll
    for (int j = 0; j < NB; ++j) dhr[r][j] = j < nsucc ? acc[j] : 0.f;
  }
  __syncthreads();
  if (threadIdx.x < ROWS * NB) {
    const int rr = threadIdx.x / NB, j = threadIdx.x % NB;
    const int u = blockIdx.x * ROWS + rr;
    if (j < nb && u < H) {
      const int b = b0 + j;
      const int nprev = max(0, min(nb, p.n_prev - b0));
      const size_t bh = (size_t)b * H + u;
      const float* g = p.gates_t + (size_t)b * 4 * H;
      const float gi = g[u], gf = g[H + u], gg = g[2 * H + u], go = g[3 * H + u];
      const float tc = tanhf(p.cs_t[bh]);
      const float cp = j < nprev ? p.c_prev[bh] : (p.c0 ? p.c0[bh] : 0.f);
      const float dh = p.dhs_t[bh] + dhr[rr][j];
      const float dct = p.dc[bh] + dh * go * (1.f - tc * tc);
      p.dc[bh] = dct * gf;
      float* dg = p.dgates_t + (size_t)b * 4 * H;
      dg[u] = dct * gg * gi * (1.f - gi);
      dg[H + u] = dct * cp * gf * (1.f - gf);
      dg[2 * H + u] = dct * gi * (1.f - gg * gg);
      dg[3 * H + u] = dh * tc * go * (1.f - go);
    }
  }
}

int check_batch_sizes(const int32_t* bs, int t, int b) {
  for (int i = 0; i < t; ++i) {
    if (bs[i] <= 0 || bs[i] > b) return 1;
    if (i > 0 && bs[i] > bs[i - 1]) return 1;
  }
  return 0;
}

}  // namespace

extern "C" int vnqa_lstm_wide_fwd(const float* xg, const float* w_hh, const float* h0, const float* c0,
                                  const int32_t* batch_sizes_host, float* hs, float* cs, float* gates, int32_t t,
                                  int32_t b, int32_t h, int32_t reverse, void* stream) {
  VNQA_CHECK_ARG(xg && w_hh && batch_sizes_host && hs && cs && gates, "lstm_wide_fwd: null pointer");
  VNQA_CHECK_ARG(t > 0 && b > 0 && h > 0, "lstm_wide_fwd: empty problem");
  VNQA_CHECK_ARG(h % 4 == 0, "lstm_wide_fwd: hidden size %d must be a multiple of 4", h);
  VNQA_CHECK_ARG(!check_batch_sizes(batch_sizes_host, t, b),
                 "lstm_wide_fwd: batch_sizes must be positive, <= b and non-increasing");
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (size_t)b * h, sg = (size_t)b * 4 * h;
  for (int i = 0; i < t; ++i) {
    const int step = reverse ? t - 1 - i : i;
    const int pred = reverse ? step + 1 : step - 1;            // predecessor in the chain
    WideFwd a;
    a.xg_t = xg + step * sg;
    a.w_hh = w_hh;
    a.h0 = h0;
    a.c0 = c0;
    a.hs_t = hs + step * sh;
    a.cs_t = cs + step * sh;
    a.gates_t = gates + step * sg;
    a.H = h;
    a.n_act = batch_sizes_host[step];
    const bool has_pred = pred >= 0 && pred < t;
    a.n_prev = has_pred ? (batch_sizes_host[pred] < a.n_act ? batch_sizes_host[pred] : a.n_act) : 0;
    a.h_prev = has_pred ? hs + pred * sh : hs;
    a.c_prev = has_pred ? cs + pred * sh : cs;
    dim3 grid(h, (a.n_act + NB - 1) / NB);
    hipLaunchKernelGGL(lstm_wide_fwd_step, grid, dim3(256), 0, st, a);
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_lstm_wide_bwd(const float* w_hh_t, const float* c0, const int32_t* batch_sizes_host,
                                  const float* gates, const float* cs, const float* dhs, float* dgates,
                                  float* dc_work, int32_t t, int32_t b, int32_t h, int32_t reverse, void* stream) {
  VNQA_CHECK_ARG(w_hh_t && batch_sizes_host && gates && cs && dhs && dgates && dc_work, "lstm_wide_bwd: null pointer");
  VNQA_CHECK_ARG(t > 0 && b > 0 && h > 0, "lstm_wide_bwd: empty problem");
  VNQA_CHECK_ARG(h % 4 == 0, "lstm_wide_bwd: hidden size %d must be a multiple of 4", h);
  VNQA_CHECK_ARG(!check_batch_sizes(batch_sizes_host, t, b),
                 "lstm_wide_bwd: batch_sizes must be positive, <= b and non-increasing");
  hipStream_t st = (hipStream_t)stream;
  const size_t sh = (size_t)b * h, sg = (size_t)b * 4 * h;
  for (int i = 0; i < t; ++i) {
    // walk the chain backwards: the forward direction ended at t-1, the reverse direction at 0
    const int step = reverse ? i : t - 1 - i;
    const int succ = reverse ? step - 1 : step + 1;            // processed AFTER `step` in the forward pass
    const int pred = reverse ? step + 1 : step - 1;
    WideBwd a;
    a.w_hh_t = w_hh_t;
    a.c0 = c0;
    a.H = h;
    a.n_act = batch_sizes_host[step];
    const bool has_succ = succ >= 0 && succ < t;
    const bool has_pred = pred >= 0 && pred < t;
    a.n_succ = has_succ ? (batch_sizes_host[succ] < a.n_act ? batch_sizes_host[succ] : a.n_act) : 0;
    a.n_prev = has_pred ? (batch_sizes_host[pred] < a.n_act ? batch_sizes_host[pred] : a.n_act) : 0;
    a.dg_succ = has_succ ? dgates + succ * sg : dgates;
    a.c_prev = has_pred ? cs + pred * sh : cs;
    a.dhs_t = dhs + step * sh;
    a.gates_t = gates + step * sg;
    a.cs_t = cs + step * sh;
    a.dc = dc_work;
    a.dgates_t = dgates + step * sg;
    dim3 grid((h + ROWS - 1) / ROWS, (a.n_act + NB - 1) / NB);
    hipLaunchKernelGGL(lstm_wide_bwd_step, grid, dim3(256), 0, st, a);
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
