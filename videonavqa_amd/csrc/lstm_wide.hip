// lstm_wide.hip — packed-sequence LSTM for WIDE hidden states (H = 512 / 1536), forward and BPTT.
//
// MACNetwork runs a bidirectional question LSTM (embed_hidden -> dim, models/mac.py:185-186,211) and a
// tail LSTM over the per-frame outputs (3*dim -> 3*dim, :193,249-251).  With dim = 512 the recurrent
// matrix W_hh is 4 MB / 37.7 MB: it cannot live in one workgroup's registers like the H <= 256
// persistent kernel (lstm.hip), so here ONE LAUNCH PER TIME STEP streams W_hh through the whole chip:
// a workgroup owns one hidden unit (its 4 gate rows), the 64 lanes of a wave share one row and split K,
// every lane keeps one accumulator per sample, and the cell update of those units runs in the same launch.  A step is
// weight-bandwidth-bound (4H*H*4 bytes from L2/MALL per step), the sequence is a chain of T such
// launches enqueued back to back on the caller's stream by one C call.  Forward: one wave per gate row of the unit.  Backward
// (rows of W_hh^T are 4H long): the four waves of the unit's workgroup split the row's K four ways and fold their partial sums
// in fixed order through LDS — 6 dependent 1-KiB loads per lane instead of 24 (a wave owning a whole 24-KiB row ran at L2
// latency: 21 us per step against the forward's 10).  The two directions of a bidirectional LSTM are independent chains of
// the same length: vnqa_lstm_wide_bidir_{fwd,bwd} put step i of the forward direction and step t-1-i of the reverse one into
// ONE launch (blockIdx.z = direction), halving the launches on MACNetwork's question chain.
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
constexpr int NR = 4;        // matrix rows a wave multiplies at once (they share the x registers)
constexpr int UW = 4;        // hidden units per workgroup (256 threads)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// acc[r * NB + j] += sum over this lane's k of w_r[k] * x[j][k] for NR matrix rows at once, k split over the 64 lanes in
// float4 pieces; rows of W `ldw` apart, rows of x `ldx` apart.  Samples >= nvalid alias sample 0 (unconditional loads; the
// caller ignores their sums).  One row per wave re-read the whole x chunk (NB x K floats) per row — 8x the bytes of the
// matrix itself through the vector-memory path, which is what bounded a step; NR rows per wave share those registers, and
// every load of an iteration is issued before its first FMA.
__device__ __forceinline__ void rows_matvec(const float* __restrict__ w, size_t ldw, const float* __restrict__ x, size_t ldx,
                                            int K, int kl, int nvalid, float (&acc)[NR * NB]) {
  const float* xr[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) xr[j] = x + (size_t)(j < nvalid ? j : 0) * ldx;
  // two register sets: the loads of iteration i + 1 are in flight while iteration i multiplies (a wave's chain is
  // K / 256 dependent round trips to L2 / MALL otherwise); the last prefetch re-reads the current piece (in bounds, unused)
  float4 wv[NR], xv[NB], wn[NR], xn[NB];
  int k = kl * 4;
  if (k >= K) return;
#pragma unroll
  for (int r = 0; r < NR; ++r) wv[r] = *(const float4*)(w + r * ldw + k);
#pragma unroll
  for (int j = 0; j < NB; ++j) xv[j] = *(const float4*)(xr[j] + k);
  for (; k < K; k += KL * 4) {
    const int kn = k + KL * 4 < K ? k + KL * 4 : k;
#pragma unroll
    for (int r = 0; r < NR; ++r) wn[r] = *(const float4*)(w + r * ldw + kn);
#pragma unroll
    for (int j = 0; j < NB; ++j) xn[j] = *(const float4*)(xr[j] + kn);
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[r * NB + j] = fmaf(wv[r].x, xv[j].x, fmaf(wv[r].y, xv[j].y, fmaf(wv[r].z, xv[j].z, fmaf(wv[r].w, xv[j].w, acc[r * NB + j]))));
#pragma unroll
    for (int r = 0; r < NR; ++r) wv[r] = wn[r];
#pragma unroll
    for (int j = 0; j < NB; ++j) xv[j] = xn[j];
  }
}

// Folds the 64 lane partials of all NR*NB = 32 sums with 32 shuffles (a reduce-scatter: at distance M a lane keeps N of its
// 2N values and sends the other N) instead of 32 x 6.  The tree is the xor butterfly's (32, 16, .., 1), so every total has the
// bits a per-value butterfly gives.  Returns the total of value index (lane >> 1) = row * NB + sample.
template <int N, int M>
__device__ __forceinline__ void fold_step(float (&v)[NR * NB], int lane) {
  const bool up = (lane & M) != 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float keep = up ? v[i + N] : v[i];
    const float send = up ? v[i] : v[i + N];
    v[i] = keep + __shfl_xor(send, M, 64);
  }
}
__device__ __forceinline__ float fold_lanes(float (&v)[NR * NB], int lane) {
  static_assert(NR * NB == 32, "fold_lanes: 32 values over 64 lanes");
  fold_step<16, 32>(v, lane);
  fold_step<8, 16>(v, lane);
  fold_step<4, 8>(v, lane);
  fold_step<2, 4>(v, lane);
  fold_step<1, 2>(v, lane);
  return v[0] + __shfl_xor(v[0], 1, 64);
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

struct WideFwd2 { WideFwd d[2]; };      // blockIdx.z selects the direction (single-direction calls launch z = 1)

// a wave owns one hidden unit: its four gate rows in one pass over h_prev, then the cell update by its first NB lanes
__global__ void __launch_bounds__(256) lstm_wide_fwd_step(const WideFwd2 pp) {
  const WideFwd& p = pp.d[blockIdx.z];
  const int H = p.H;
  const int lane = threadIdx.x % KL;
  const int unit = blockIdx.x * UW + threadIdx.x / KL;
  const int b0 = blockIdx.y * NB;
  if (b0 >= p.n_act || unit >= H) return;             // (wave-uniform; the other direction may have more active samples)
  const int nb = min(NB, p.n_act - b0);
  // samples of this chunk with a predecessor state form a prefix (batch sizes are non-increasing)
  const int nprev = max(0, min(nb, p.n_prev - b0));
  float acc[NR * NB];
#pragma unroll
  for (int i = 0; i < NR * NB; ++i) acc[i] = 0.f;
  const float* wrow = p.w_hh + (size_t)unit * H;       // gate g of this unit: row g * H + unit
  if (nprev > 0) rows_matvec(wrow, (size_t)H * H, p.h_prev + (size_t)b0 * H, H, H, lane, nprev, acc);
  float tot = fold_lanes(acc, lane);                   // (gate, sample) = (lane >> 4, (lane >> 1) & 7)
  if (((lane >> 1) & 7) >= nprev) tot = 0.f;           // no predecessor: zero recurrent term (or h0's, below)
  if (nprev < nb && p.h0 != nullptr) {                 // samples starting at this step: from h0
#pragma unroll
    for (int i = 0; i < NR * NB; ++i) acc[i] = 0.f;
    rows_matvec(wrow, (size_t)H * H, p.h0 + (size_t)(b0 + nprev) * H, H, H, lane, nb - nprev, acc);
    const float tot0 = fold_lanes(acc, lane);
    // sample j >= nprev takes column j - nprev of the h0 product
    const int j = (lane >> 1) & 7;
    const float shifted = __shfl(tot0, (lane & 48) | (((j - nprev) & 7) << 1), 64);
    if (j >= nprev) tot = shifted;
  }
  const int j = lane < NB ? lane : 0;
  const float pi = __shfl(tot, 0 | (j << 1), 64), pf = __shfl(tot, 16 | (j << 1), 64);
  const float pg = __shfl(tot, 32 | (j << 1), 64), po = __shfl(tot, 48 | (j << 1), 64);
  if (lane < nb) {
    const int b = b0 + lane, u = unit;
    const float* xg = p.xg_t + (size_t)b * 4 * H;      // (no predecessor and no h0: the recurrent sums are exactly zero)
    const float gi = sigm(pi + xg[u]);
    const float gf = sigm(pf + xg[H + u]);
    const float gg = tanhf(pg + xg[2 * H + u]);
    const float go = sigm(po + xg[3 * H + u]);
    const float cp = lane < nprev ? p.c_prev[(size_t)b * H + u] : (p.c0 ? p.c0[(size_t)b * H + u] : 0.f);
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

struct WideBwd2 { WideBwd d[2]; };

// a workgroup owns UW = NR units (rows of W_hh^T, 4H long); wave q multiplies quarter q of the four rows' K range, the
// quarters are folded in fixed order through LDS
__global__ void __launch_bounds__(256) lstm_wide_bwd_step(const WideBwd2 pp) {
  __shared__ float part[4][NR * NB];
  const WideBwd& p = pp.d[blockIdx.z];
  const int H = p.H;
  const int q = threadIdx.x / KL, lane = threadIdx.x % KL;
  const int unit0 = blockIdx.x * UW;
  const int b0 = blockIdx.y * NB;
  if (b0 >= p.n_act) return;
  const int nb = min(NB, p.n_act - b0);
  const int nsucc = max(0, min(nb, p.n_succ - b0));
  float acc[NR * NB];
#pragma unroll
  for (int i = 0; i < NR * NB; ++i) acc[i] = 0.f;
  // (units past H — H is a multiple of 4 = UW, so there are none — would alias in-range rows)
  if (nsucc > 0)
    rows_matvec(p.w_hh_t + (size_t)unit0 * 4 * H + (size_t)q * H, (size_t)4 * H, p.dg_succ + (size_t)b0 * 4 * H + (size_t)q * H,
                (size_t)4 * H, H, lane, nsucc, acc);
  const float tot = fold_lanes(acc, lane);             // (unit, sample) = (lane >> 4, (lane >> 1) & 7)
  if ((lane & 1) == 0) part[q][lane >> 1] = tot;
  __syncthreads();
  if (threadIdx.x < NR * NB) {
    const int rr = threadIdx.x / NB, j = threadIdx.x % NB;
    const int u = unit0 + rr;
    if (j < nb && u < H) {
      const int b = b0 + j;
      const int nprev = max(0, min(nb, p.n_prev - b0));
      const size_t bh = (size_t)b * H + u;
      const float* g = p.gates_t + (size_t)b * 4 * H;
      const float gi = g[u], gf = g[H + u], gg = g[2 * H + u], go = g[3 * H + u];
      const float tc = tanhf(p.cs_t[bh]);
      const float cp = j < nprev ? p.c_prev[bh] : (p.c0 ? p.c0[bh] : 0.f);
      const int i = rr * NB + j;
      const float dhr = j < nsucc ? ((part[0][i] + part[1][i]) + part[2][i]) + part[3][i] : 0.f;
      const float dh = p.dhs_t[bh] + dhr;
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

// arguments of chain position i (0-based along the walk) of one direction
WideFwd fwd_args(const float* xg, const float* w_hh, const float* h0, const float* c0, const int32_t* bs, float* hs, float* cs,
                 float* gates, int t, int b, int h, int reverse, int i) {
  const size_t sh = (size_t)b * h, sg = (size_t)b * 4 * h;
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
  a.n_act = bs[step];
  const bool has_pred = pred >= 0 && pred < t;
  a.n_prev = has_pred ? (bs[pred] < a.n_act ? bs[pred] : a.n_act) : 0;
  a.h_prev = has_pred ? hs + pred * sh : hs;
  a.c_prev = has_pred ? cs + pred * sh : cs;
  return a;
}

// chain position i of the BACKWARD walk: the forward direction ended at t-1, the reverse direction at 0
WideBwd bwd_args(const float* w_hh_t, const float* c0, const int32_t* bs, const float* gates, const float* cs, const float* dhs,
                 float* dgates, float* dc_work, int t, int b, int h, int reverse, int i) {
  const size_t sh = (size_t)b * h, sg = (size_t)b * 4 * h;
  const int step = reverse ? i : t - 1 - i;
  const int succ = reverse ? step - 1 : step + 1;            // processed AFTER `step` in the forward pass
  const int pred = reverse ? step + 1 : step - 1;
  WideBwd a;
  a.w_hh_t = w_hh_t;
  a.c0 = c0;
  a.H = h;
  a.n_act = bs[step];
  const bool has_succ = succ >= 0 && succ < t;
  const bool has_pred = pred >= 0 && pred < t;
  a.n_succ = has_succ ? (bs[succ] < a.n_act ? bs[succ] : a.n_act) : 0;
  a.n_prev = has_pred ? (bs[pred] < a.n_act ? bs[pred] : a.n_act) : 0;
  a.dg_succ = has_succ ? dgates + succ * sg : dgates;
  a.c_prev = has_pred ? cs + pred * sh : cs;
  a.dhs_t = dhs + step * sh;
  a.gates_t = gates + step * sg;
  a.cs_t = cs + step * sh;
  a.dc = dc_work;
  a.dgates_t = dgates + step * sg;
  return a;
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
  for (int i = 0; i < t; ++i) {
    WideFwd2 a;
    a.d[0] = a.d[1] = fwd_args(xg, w_hh, h0, c0, batch_sizes_host, hs, cs, gates, t, b, h, reverse, i);
    dim3 grid((h + UW - 1) / UW, (a.d[0].n_act + NB - 1) / NB, 1);
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
  for (int i = 0; i < t; ++i) {
    WideBwd2 a;
    a.d[0] = a.d[1] = bwd_args(w_hh_t, c0, batch_sizes_host, gates, cs, dhs, dgates, dc_work, t, b, h, reverse, i);
    dim3 grid((h + UW - 1) / UW, (a.d[0].n_act + NB - 1) / NB, 1);
    hipLaunchKernelGGL(lstm_wide_bwd_step, grid, dim3(256), 0, st, a);
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

// Both directions of a bidirectional packed LSTM (zero initial state), chain position i of each in one launch.
extern "C" int vnqa_lstm_wide_bidir_fwd(const float* xg_f, const float* xg_r, const float* w_hh_f, const float* w_hh_r,
                                        const int32_t* batch_sizes_host, float* hs_f, float* hs_r, float* cs_f, float* cs_r,
                                        float* gates_f, float* gates_r, int32_t t, int32_t b, int32_t h, void* stream) {
  VNQA_CHECK_ARG(xg_f && xg_r && w_hh_f && w_hh_r && batch_sizes_host && hs_f && hs_r && cs_f && cs_r && gates_f && gates_r,
                 "lstm_wide_bidir_fwd: null pointer");
  VNQA_CHECK_ARG(t > 0 && b > 0 && h > 0, "lstm_wide_bidir_fwd: empty problem");
  VNQA_CHECK_ARG(h % 4 == 0, "lstm_wide_bidir_fwd: hidden size %d must be a multiple of 4", h);
  VNQA_CHECK_ARG(!check_batch_sizes(batch_sizes_host, t, b),
                 "lstm_wide_bidir_fwd: batch_sizes must be positive, <= b and non-increasing");
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < t; ++i) {
    WideFwd2 a;
    a.d[0] = fwd_args(xg_f, w_hh_f, nullptr, nullptr, batch_sizes_host, hs_f, cs_f, gates_f, t, b, h, 0, i);
    a.d[1] = fwd_args(xg_r, w_hh_r, nullptr, nullptr, batch_sizes_host, hs_r, cs_r, gates_r, t, b, h, 1, i);
    const int n_act = a.d[0].n_act > a.d[1].n_act ? a.d[0].n_act : a.d[1].n_act;
    dim3 grid((h + UW - 1) / UW, (n_act + NB - 1) / NB, 2);
    hipLaunchKernelGGL(lstm_wide_fwd_step, grid, dim3(256), 0, st, a);
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_lstm_wide_bidir_bwd(const float* w_hh_t_f, const float* w_hh_t_r, const int32_t* batch_sizes_host,
                                        const float* gates_f, const float* gates_r, const float* cs_f, const float* cs_r,
                                        const float* dhs_f, const float* dhs_r, float* dgates_f, float* dgates_r,
                                        float* dc_work_f, float* dc_work_r, int32_t t, int32_t b, int32_t h, void* stream) {
  VNQA_CHECK_ARG(w_hh_t_f && w_hh_t_r && batch_sizes_host && gates_f && gates_r && cs_f && cs_r && dhs_f && dhs_r && dgates_f &&
                     dgates_r && dc_work_f && dc_work_r && dc_work_f != dc_work_r, "lstm_wide_bidir_bwd: null / shared pointer");
  VNQA_CHECK_ARG(t > 0 && b > 0 && h > 0, "lstm_wide_bidir_bwd: empty problem");
  VNQA_CHECK_ARG(h % 4 == 0, "lstm_wide_bidir_bwd: hidden size %d must be a multiple of 4", h);
  VNQA_CHECK_ARG(!check_batch_sizes(batch_sizes_host, t, b),
                 "lstm_wide_bidir_bwd: batch_sizes must be positive, <= b and non-increasing");
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < t; ++i) {
    WideBwd2 a;
    a.d[0] = bwd_args(w_hh_t_f, nullptr, batch_sizes_host, gates_f, cs_f, dhs_f, dgates_f, dc_work_f, t, b, h, 0, i);
    a.d[1] = bwd_args(w_hh_t_r, nullptr, batch_sizes_host, gates_r, cs_r, dhs_r, dgates_r, dc_work_r, t, b, h, 1, i);
    const int n_act = a.d[0].n_act > a.d[1].n_act ? a.d[0].n_act : a.d[1].n_act;
    dim3 grid((h + UW - 1) / UW, (n_act + NB - 1) / NB, 2);
    hipLaunchKernelGGL(lstm_wide_bwd_step, grid, dim3(256), 0, st, a);
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
