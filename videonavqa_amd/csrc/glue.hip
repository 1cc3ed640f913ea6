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
#include <cstdlib>

#include "vnqa_common.h"

namespace {

struct SgemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;     // [N] or null
  const float* bias2;    // second bias vector added like `bias` (b_ih + b_hh of an LSTM input projection); or null
  float* out2;           // optional second output (rows / ldc as C): out2 = result * out2_col[n] * out2_mul(m, n); or null
  const float* out2_col; // [N] or null
  const float* out2_mul; // matrix indexed like `addend` (logical row m, ldc) or null
  const float* addend;   // matrix added to the product (same rows / ldc as C, read at the LOGICAL row m); or null
  const float* a_mask;   // same indexing as A: A'(m,k) = A(m,k) * [a_mask(m,k) > 0]; or null
  const int* a_rows;     // [M]: physical row of A for logical row m (negative: a zero row); or null
  const int* c_rows;     // [M]: physical row of C for logical row m (negative: not written); or null
  long long a_rs, a_cs, b_rs, b_cs;   // element strides: A'(m,k) = A[row(m)*a_rs + k*a_cs], B(k,n) = B[k*b_rs + n*b_cs]
  int ldc;
  int M, N, K;
  int relu, accumulate;
};

// The fp32 GEMM on the exact-f32 matrix core (v_mfma_f32_32x32x2_f32: each product is an fmaf into the fp32 accumulator, no
// reduced-precision step anywhere).  Made for the products this library actually issues — [a few hundred rows] x [512] over
// K = 512 (MACNetwork's reasoning step: 216 of them per training step), out_linear, the FiLM generators — which are latency
// problems, not throughput problems: a workgroup is 8 waves laid out WM x WN x WK; every wave owns one 32 x 32 output tile
// over 1/WK of the K range, loads its operands straight from L2 into the MFMA register layout (float4 along K where the
// operand is K-contiguous — any permutation of K is valid as long as A and B use the same one), and the WK partial tiles
// are summed through LDS in a fixed order (deterministic).  <1,1,8>: 32 x 32 tile, K split 8 ways — [280,512]x[512,512] is
// 144 workgroups of 8 waves with 64 K-values each, i.e. the whole chip for ~32 MFMAs per wave.  <2,2,2>: 64 x 64 tiles for
// larger outputs.  gridDim.z > 1: split-K — slice z covers K range [z * k_per_slice, ...) and writes its raw partial tile to
// `partial` [z][M][N]; sgemm_finish_kernel applies the epilogue.
template <int WM, int WN, int WK>
__device__ __forceinline__ void sgemm_mfma_body(const SgemmArgs& p, float* __restrict__ partial, int k_per_slice, int slice,
                                                int a_vec, int b_vec) {
  static_assert(WM * WN * WK == 8, "8 waves");
  constexpr int TILE = 32 * 33;
  __shared__ float red[WK * WM * WN * TILE];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wk = wave % WK, wn = (wave / WK) % WN, wm = wave / (WK * WN);
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * (32 * WM), n0 = blockIdx.x * (32 * WN);
  const int ks = slice * k_per_slice;
  const int kse = (ks + k_per_slice < p.K) ? ks + k_per_slice : p.K;
  const int kper = (((kse - ks) + WK - 1) / WK + 7) / 8 * 8;
  const int kb = ks + wk * kper;
  const int ke = (kb + kper < kse) ? kb + kper : kse;
  const int am = m0 + wm * 32 + r;
  bool a_ok = am < p.M;
  long long arow = am;
  if (a_ok && p.a_rows != nullptr) {
    arow = p.a_rows[am];
    a_ok = arow >= 0;
  }
  const float* Ap = p.A + arow * p.a_rs;
  const float* Mp = p.a_mask != nullptr ? p.a_mask + arow * p.a_rs : nullptr;
  const int bn = n0 + wn * 32 + r;
  const bool b_ok = bn < p.N;
  const float* Bp = p.B + (long long)bn * p.b_cs;
  auto load_a = [&](int k, float (&a)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = 0.f;
    if (!a_ok || k >= ke) return;
    if (a_vec && k + 3 < ke) {
      const float4 v = *reinterpret_cast<const float4*>(Ap + k);
      a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
      if (Mp != nullptr) {
        const float4 q = *reinterpret_cast<const float4*>(Mp + k);
        if (!(q.x > 0.f)) a[0] = 0.f;
        if (!(q.y > 0.f)) a[1] = 0.f;
        if (!(q.z > 0.f)) a[2] = 0.f;
        if (!(q.w > 0.f)) a[3] = 0.f;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k + j < ke) {
        const long long off = (long long)(k + j) * p.a_cs;
        float v = Ap[off];
        if (Mp != nullptr && !(Mp[off] > 0.f)) v = 0.f;
        a[j] = v;
      }
  };
  auto load_b = [&](int k, float (&b)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = 0.f;
    if (!b_ok || k >= ke) return;
    if (b_vec && k + 3 < ke) {
      const float4 v = *reinterpret_cast<const float4*>(Bp + k);
      b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k + j < ke) b[j] = Bp[(long long)(k + j) * p.b_rs];
  };
  vnqa_f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // chunks of 8 K-values (lane half h takes k0 + 4 h .. + 3 of each, so the two halves of a row read ADJACENT 16 bytes and one load
  // instruction touches 32 lines); the next stage's loads fly under this stage's MFMAs.  (Measured and dropped: giving each
  // half 32 consecutive K-values, so that a lane's eight loads walk one 128-byte line — 64 lines per instruction instead of
  // 32 — made the [280,512]x[512,512] product slower, 11.9 -> 15.4 us: the address coalescer, not L1 reuse, is the limit.)
  constexpr int U = 4;
  float a0[U][4], b0[U][4], a1[U][4], b1[U][4];
#pragma unroll
  for (int c = 0; c < U; ++c) {
    load_a(kb + 8 * c + 4 * h, a0[c]);
    load_b(kb + 8 * c + 4 * h, b0[c]);
  }
  for (int k0 = kb; k0 < ke; k0 += 16 * U) {
#pragma unroll
    for (int c = 0; c < U; ++c) {
      load_a(k0 + 8 * (U + c) + 4 * h, a1[c]);
      load_b(k0 + 8 * (U + c) + 4 * h, b1[c]);
    }
#pragma unroll
    for (int c = 0; c < U; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[c][j], b0[c][j], acc, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < U; ++c) {
      load_a(k0 + 8 * (2 * U + c) + 4 * h, a0[c]);
      load_b(k0 + 8 * (2 * U + c) + 4 * h, b0[c]);
    }
#pragma unroll
    for (int c = 0; c < U; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[c][j], b1[c][j], acc, 0, 0, 0);
  }
  // accumulator element i of a lane: row 8 (i / 4) + 4 h + i % 4, column r
  float* mine = red + (size_t)((wm * WN + wn) * WK + wk) * TILE;
#pragma unroll
  for (int i = 0; i < 16; ++i) mine[(8 * (i >> 2) + 4 * h + (i & 3)) * 33 + r] = acc[i];
  __syncthreads();
  for (int e = threadIdx.x; e < WM * WN * 1024; e += 512) {
    const int t = e >> 10, idx = e & 1023, row = idx >> 5, col = idx & 31;
    const int m = m0 + (t / WN) * 32 + row, n = n0 + (t % WN) * 32 + col;
    if (m >= p.M || n >= p.N) continue;
    const float* src = red + (size_t)t * WK * TILE + row * 33 + col;
    float v = 0.f;
#pragma unroll
    for (int z = 0; z < WK; ++z) v += src[z * TILE];
    if (partial != nullptr) {
      partial[((size_t)slice * p.M + m) * p.N + n] = v;
      continue;
    }
    int orow = m;
    if (p.c_rows != nullptr) {
      orow = p.c_rows[m];
      if (orow < 0) continue;
    }
    if (p.bias != nullptr) v += p.bias[n];
    if (p.bias2 != nullptr) v += p.bias2[n];
    if (p.addend != nullptr) v += p.addend[(long long)m * p.ldc + n];
    float* dst = p.C + (long long)orow * p.ldc + n;
    if (p.accumulate) v += *dst;
    if (p.relu) v = fmaxf(v, 0.f);
    *dst = v;
    if (p.out2 != nullptr) {
      float w = v;
      if (p.out2_col != nullptr) w *= p.out2_col[n];
      if (p.out2_mul != nullptr) w *= p.out2_mul[(long long)m * p.ldc + n];
      p.out2[(long long)orow * p.ldc + n] = w;
    }
  }
}

template <int WM, int WN, int WK>
__global__ void __launch_bounds__(512) sgemm_mfma_kernel(const SgemmArgs p, float* __restrict__ partial, int k_per_slice,
                                                         int a_vec, int b_vec) {
  sgemm_mfma_body<WM, WN, WK>(p, partial, k_per_slice, blockIdx.z, a_vec, b_vec);
}

// Up to SGEMM_BATCH independent products in ONE launch (blockIdx.z = problem; one pass over K each): MACNetwork's reasoning
// step issues small products that do not depend on each other back to back — each launch costs the dependent chain ~5.5 us
// before any work starts, about what the product itself takes.
constexpr int SGEMM_BATCH = 4;
struct SgemmBatchArgs {
  SgemmArgs a[SGEMM_BATCH];
  int a_vec[SGEMM_BATCH], b_vec[SGEMM_BATCH];
};
__global__ void __launch_bounds__(512) sgemm_mfma_batch_kernel(const SgemmBatchArgs pp) {
  const SgemmArgs& p = pp.a[blockIdx.z];
  if ((int)blockIdx.y * 32 >= p.M || (int)blockIdx.x * 32 >= p.N) return;      // grid = the largest problem's (workgroup-uniform)
  sgemm_mfma_body<1, 1, 8>(p, nullptr, (p.K + 15) / 16 * 16, 0, pp.a_vec[blockIdx.z], pp.b_vec[blockIdx.z]);
}

// float4 operand loads along K: contiguous K, 16-byte aligned rows
static int sgemm_a_vec(const SgemmArgs& p, int kps) {
  return (p.a_cs == 1 && p.a_rs % 4 == 0 && ((uintptr_t)p.A & 15) == 0 && kps % 8 == 0 &&
          (p.a_mask == nullptr || ((uintptr_t)p.a_mask & 15) == 0)) ? 1 : 0;
}
static int sgemm_b_vec(const SgemmArgs& p, int kps) {
  return (p.b_rs == 1 && p.b_cs % 4 == 0 && ((uintptr_t)p.B & 15) == 0 && kps % 8 == 0) ? 1 : 0;
}

// one launch of the GEMM proper (z = split-K slices)
static void sgemm_launch(const SgemmArgs& p, float* partial, int kps, int nsl, hipStream_t st) {
  const int a_vec = sgemm_a_vec(p, kps), b_vec = sgemm_b_vec(p, kps);
  const long long tiles32 = (long long)((p.M + 31) / 32) * ((p.N + 31) / 32);
  if (tiles32 <= 1024)
    hipLaunchKernelGGL((sgemm_mfma_kernel<1, 1, 8>), dim3((p.N + 31) / 32, (p.M + 31) / 32, nsl), dim3(512), 0, st, p, partial, kps,
                       a_vec, b_vec);
  else
    hipLaunchKernelGGL((sgemm_mfma_kernel<2, 2, 2>), dim3((p.N + 63) / 64, (p.M + 63) / 64, nsl), dim3(512), 0, st, p, partial, kps,
                       a_vec, b_vec);
}

// split-K second pass: sum the slices in order (deterministic), then bias / accumulate / ReLU / row scatter
__global__ void sgemm_finish_kernel(const SgemmArgs p, const float* __restrict__ partial, int slices) {
  const size_t total = (size_t)p.M * p.N;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i / p.N), n = (int)(i - (size_t)m * p.N);
    int row = m;
    if (p.c_rows != nullptr) {
      row = p.c_rows[m];
      if (row < 0) continue;
    }
    float v = 0.f;
    for (int z = 0; z < slices; ++z) v += partial[(size_t)z * total + i];
    if (p.bias != nullptr) v += p.bias[n];
    if (p.bias2 != nullptr) v += p.bias2[n];
    if (p.addend != nullptr) v += p.addend[(long long)m * p.ldc + n];
    float* dst = p.C + (long long)row * p.ldc + n;
    if (p.accumulate) v += *dst;
    if (p.relu) v = fmaxf(v, 0.f);
    *dst = v;
    if (p.out2 != nullptr) {
      float w = v;
      if (p.out2_col != nullptr) w *= p.out2_col[n];
      if (p.out2_mul != nullptr) w *= p.out2_mul[(long long)m * p.ldc + n];
      p.out2[(long long)row * p.ldc + n] = w;
    }
  }
}

// out[n] = sum_m x[m*ld + n] * [mask > 0]: block = 64 columns x 16 row lanes (row lane r sums rows r, r+16, ...), the 16
// partial sums folded through LDS in a fixed order
template <typename T>
__global__ void __launch_bounds__(1024) colsum_kernel(const T* __restrict__ x, const float* __restrict__ mask,
                                                      float* __restrict__ out, int rows, int cols, int ld) {
  __shared__ float s_part[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  float s = 0.f;
  if (n < cols) {
    for (int m = ry; m < rows; m += 16) {
      const float v = ElemOps<T>::load(x[(size_t)m * ld + n]);
      s += (mask == nullptr || mask[(size_t)m * ld + n] > 0.f) ? v : 0.f;
    }
  }
  s_part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && n < cols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += s_part[r][cx];
    out[n] = t;
  }
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ rows, float* __restrict__ dst,
                                   int n_rows, int cols) {
  const int r = blockIdx.x;
  const int row = rows[r];
  for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(size_t)r * cols + c] = row >= 0 ? src[(size_t)row * cols + c] : 0.f;
}

// rows[b*Lq + pos] = token(row_perm[b], pos) clamped to the vocabulary: the row-gather table of the embedding GEMM
__global__ void token_rows_kernel(const long long* __restrict__ tokens, const int* __restrict__ row_perm, int* __restrict__ rows,
                                  int B, int Lq, int V) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Lq) return;
  const int b = i / Lq, pos = i - b * Lq;
  const int src_b = row_perm != nullptr ? row_perm[b] : b;
  long long tok = tokens[(size_t)src_b * Lq + pos];
  rows[i] = (int)(tok < 0 ? 0 : (tok >= V ? V - 1 : tok));
}

// dsum[v][j] = sum over the (b, pos) whose token is v of dxg[b][pos][j]   (positions in a fixed order)
__global__ void token_dsum_kernel(const int* __restrict__ rows, const float* __restrict__ dxg, float* __restrict__ dsum, int n_pos,
                                  int G) {
  extern __shared__ int s_hit[];              // positions holding token v, in order
  __shared__ int s_n;
  const int v = blockIdx.x;
  if (threadIdx.x < 64) {                     // wave 0: ordered compaction, 64 positions per ballot (summation order fixed)
    const int lane = threadIdx.x;
    int n = 0;
    for (int i0 = 0; i0 < n_pos; i0 += 64) {
      const int i = i0 + lane;
      const bool hit = i < n_pos && rows[i] == v;
      const unsigned long long m = __ballot(hit);
      if (hit) s_hit[n + __popcll(m & ((1ull << lane) - 1ull))] = i;
      n += __popcll(m);
    }
    if (lane == 0) s_n = n;
  }
  __syncthreads();
  const int n = s_n;
  for (int j = threadIdx.x; j < G; j += blockDim.x) {
    float s = 0.f;
    for (int h = 0; h < n; ++h) s += dxg[(size_t)s_hit[h] * G + j];
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
    // a target outside [0, K) is IGNORED (zero weight, zero gradient) — nn.CrossEntropyLoss's ignore_index = -100 lands here;
    // it never indexes weight[] / the logits row out of bounds
    const bool valid = y >= 0 && y < K;
    const float wy = !valid ? 0.f : (weight != nullptr ? weight[y] : 1.f);
    const float lse = mx + logf(den);
    for (int k = lane; k < K; k += 64) {
      const float pk = expf(lg[k] - lse);
      dlogits[(size_t)b * K + k] = wy * (pk - (k == y ? 1.f : 0.f));      // scaled by 1 / sum(w) below for 'mean'
    }
    if (lane == 0 && valid) {
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

// Split-K plan: few output tiles but a long K (out_linear: 8 x 70 outputs over K = 4480; the FiLM generator's d h:
// 280 x 128 over K = 1024) would leave the chip to a handful of workgroups walking K serially at memory latency per chunk.
static int sgemm_slices(int m, int n, int k) {
  // a workgroup already splits its K range over 8 waves; more slices only when the output has too few 32 x 32 tiles to occupy
  // the chip AND every slice still gets >= 128 K-values (16 per wave)
  const long long tiles = (long long)((m + 31) / 32) * ((n + 31) / 32);
  if (tiles >= 96 || k < 512) return 1;
  int s = (int)((192 + tiles - 1) / tiles);
  const int max_s = k / 128;
  s = s > max_s ? max_s : s;
  return s < 1 ? 1 : s;
}

extern "C" int64_t vnqa_sgemm_workspace(int32_t m, int32_t n, int32_t k) {
  const int s = sgemm_slices(m, n, k);
  return s <= 1 ? 0 : (int64_t)s * m * n * 4;
}

static int sgemm_run(const float* a, const float* b, float* c, const float* bias, const float* a_mask, const int32_t* a_rows,
                     const int32_t* c_rows, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int32_t ldc, int32_t m,
                     int32_t n, int32_t k, int32_t relu, int32_t accumulate, const float* addend, void* workspace, void* stream,
                     float* out2, const float* out2_col, const float* out2_mul);

extern "C" int vnqa_sgemm(const float* a, const float* b, float* c, const float* bias, const float* a_mask,
                          const int32_t* a_rows, const int32_t* c_rows, int64_t a_rs, int64_t a_cs, int64_t b_rs,
                          int64_t b_cs, int32_t ldc, int32_t m, int32_t n, int32_t k, int32_t relu, int32_t accumulate,
                          const float* addend, void* workspace, void* stream) {
  return sgemm_run(a, b, c, bias, a_mask, a_rows, c_rows, a_rs, a_cs, b_rs, b_cs, ldc, m, n, k, relu, accumulate, addend, workspace,
                   stream, nullptr, nullptr, nullptr);
}

// vnqa_sgemm with a second output written by the same epilogue: out2 = C_result * out2_col[n] * out2_mul(m, n)
extern "C" int vnqa_sgemm2(const float* a, const float* b, float* c, const float* bias, int64_t a_rs, int64_t a_cs, int64_t b_rs,
                           int64_t b_cs, int32_t ldc, int32_t m, int32_t n, int32_t k, const float* addend, float* out2,
                           const float* out2_col, const float* out2_mul, void* workspace, void* stream) {
  VNQA_CHECK_ARG(out2 != nullptr, "sgemm2: out2 required");
  return sgemm_run(a, b, c, bias, nullptr, nullptr, nullptr, a_rs, a_cs, b_rs, b_cs, ldc, m, n, k, 0, 0, addend, workspace, stream,
                   out2, out2_col, out2_mul);
}

static int sgemm_run(const float* a, const float* b, float* c, const float* bias, const float* a_mask, const int32_t* a_rows,
                     const int32_t* c_rows, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int32_t ldc, int32_t m,
                     int32_t n, int32_t k, int32_t relu, int32_t accumulate, const float* addend, void* workspace, void* stream,
                     float* out2, const float* out2_col, const float* out2_mul) {
  VNQA_CHECK_ARG(a && b && c && m > 0 && n > 0 && k > 0 && ldc >= n, "sgemm: bad arguments (m=%d n=%d k=%d ldc=%d)", m, n, k, ldc);
  SgemmArgs p;
  p.A = a; p.B = b; p.C = c; p.bias = bias; p.bias2 = nullptr; p.out2 = out2; p.out2_col = out2_col; p.out2_mul = out2_mul; p.addend = addend; p.a_mask = a_mask; p.a_rows = a_rows; p.c_rows = c_rows;
  p.a_rs = a_rs; p.a_cs = a_cs; p.b_rs = b_rs; p.b_cs = b_cs; p.ldc = ldc; p.M = m; p.N = n; p.K = k;
  p.relu = relu; p.accumulate = accumulate;
  hipStream_t st = (hipStream_t)stream;
  const int slices = workspace != nullptr ? sgemm_slices(m, n, k) : 1;     // workspace == NULL: one pass over K
  if (slices <= 1) {
    sgemm_launch(p, nullptr, (k + 15) / 16 * 16, 1, st);
    VNQA_CHECK_LAUNCH();
    return VNQA_OK;
  }
  const int kps = ((k + slices - 1) / slices + 15) / 16 * 16;
  const int nsl = (k + kps - 1) / kps;
  sgemm_launch(p, (float*)workspace, kps, nsl, st);
  VNQA_CHECK_LAUNCH();
  const size_t total = (size_t)m * n;
  int g = (int)((total + 255) / 256);
  g = g > 1024 ? 1024 : g;
  hipLaunchKernelGGL(sgemm_finish_kernel, dim3(g), dim3(256), 0, st, p, (const float*)workspace, nsl);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_sgemm_batch(const vnqa_sgemm_problem* problems, int32_t count, void* stream) {
  VNQA_CHECK_ARG(problems != nullptr && count >= 1 && count <= VNQA_SGEMM_BATCH_MAX, "sgemm_batch: 1..%d problems", VNQA_SGEMM_BATCH_MAX);
  static_assert(VNQA_SGEMM_BATCH_MAX == SGEMM_BATCH, "header and kernel batch limits");
  bool one_launch = count > 1;
  int gx = 0, gy = 0;
  for (int i = 0; i < count; ++i) {
    const vnqa_sgemm_problem& q = problems[i];
    VNQA_CHECK_ARG(q.a && q.b && q.c && q.m > 0 && q.n > 0 && q.k > 0 && q.ldc >= q.n,
                   "sgemm_batch: bad problem %d (m=%d n=%d k=%d ldc=%d)", i, q.m, q.n, q.k, q.ldc);
    const int tx = (q.n + 31) / 32, ty = (q.m + 31) / 32;
    if ((long long)tx * ty > 1024) one_launch = false;        // large outputs take the 64 x 64 tile form: run them one by one
    gx = tx > gx ? tx : gx;
    gy = ty > gy ? ty : gy;
  }
  if (!one_launch) {
    for (int i = 0; i < count; ++i) {
      const vnqa_sgemm_problem& q = problems[i];
      const int rc = sgemm_run(q.a, q.b, q.c, q.bias, nullptr, nullptr, nullptr, q.a_rs, q.a_cs, q.b_rs, q.b_cs, q.ldc, q.m, q.n, q.k,
                               q.relu, q.accumulate, q.addend, nullptr, stream, q.out2, q.out2_col, q.out2_mul);
      if (rc != VNQA_OK) return rc;
    }
    return VNQA_OK;
  }
  SgemmBatchArgs pp;
  for (int i = 0; i < SGEMM_BATCH; ++i) {
    const vnqa_sgemm_problem& q = problems[i < count ? i : 0];
    SgemmArgs& p = pp.a[i];
    p.A = q.a; p.B = q.b; p.C = q.c; p.bias = q.bias; p.bias2 = nullptr; p.out2 = q.out2; p.out2_col = q.out2_col;
    p.out2_mul = q.out2_mul; p.addend = q.addend; p.a_mask = nullptr; p.a_rows = nullptr; p.c_rows = nullptr;
    p.a_rs = q.a_rs; p.a_cs = q.a_cs; p.b_rs = q.b_rs; p.b_cs = q.b_cs; p.ldc = q.ldc; p.M = q.m; p.N = q.n; p.K = q.k;
    p.relu = q.relu; p.accumulate = q.accumulate;
    const int kps = (q.k + 15) / 16 * 16;
    pp.a_vec[i] = sgemm_a_vec(p, kps);
    pp.b_vec[i] = sgemm_b_vec(p, kps);
  }
  hipLaunchKernelGGL(sgemm_mfma_batch_kernel, dim3(gx, gy, count), dim3(512), 0, (hipStream_t)stream, pp);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_colsum(const void* x, const float* mask, float* out, int32_t rows, int32_t cols, int32_t ld, int32_t dtype,
                           void* stream) {
  VNQA_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols, "colsum: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_F32 || (dtype == VNQA_BF16 && mask == nullptr), "colsum: dtype must be f32, or bf16 without a mask");
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(colsum_kernel<vnqa_bf16>, dim3((cols + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const vnqa_bf16*)x,
                       mask, out, rows, cols, ld);
  else
    hipLaunchKernelGGL(colsum_kernel<float>, dim3((cols + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const float*)x, mask, out,
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
                                   const float* b_ih, const float* b_hh, float* xg, int32_t* rows, int32_t b, int32_t lq,
                                   int32_t e, int32_t g, int32_t vocab, void* stream) {
  VNQA_CHECK_ARG(tokens && embed && w_ih && b_ih && b_hh && xg && rows && b > 0 && lq > 0 && e > 0 && g > 0 && vocab > 0,
                 "embed_proj_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int n_pos = b * lq;
  hipLaunchKernelGGL(token_rows_kernel, dim3((n_pos + 255) / 256), dim3(256), 0, st, (const long long*)tokens, row_perm, rows, b,
                     lq, vocab);
  VNQA_CHECK_LAUNCH();
  // xg [n_pos][g] = embed[rows] (n_pos x e)  @  w_ih^T (e x g)  + b_ih + b_hh : the embedding lookup IS the GEMM's row gather
  SgemmArgs p;
  p.A = embed; p.B = w_ih; p.C = xg; p.bias = b_ih; p.bias2 = b_hh; p.out2 = nullptr; p.out2_col = nullptr; p.out2_mul = nullptr; p.addend = nullptr; p.a_mask = nullptr; p.a_rows = rows; p.c_rows = nullptr;
  p.a_rs = e; p.a_cs = 1; p.b_rs = 1; p.b_cs = e; p.ldc = g; p.M = n_pos; p.N = g; p.K = e; p.relu = 0; p.accumulate = 0;
  sgemm_launch(p, nullptr, (e + 15) / 16 * 16, 1, st);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_token_dsum(const int32_t* rows, const float* dxg, float* dsum, int32_t n_pos, int32_t g, int32_t vocab,
                               void* stream) {
  VNQA_CHECK_ARG(rows && dxg && dsum && n_pos > 0 && g > 0 && vocab > 0 && n_pos <= 12288, "token_dsum: bad arguments");
  const int threads = g >= 512 ? 512 : (g + 63) / 64 * 64;
  hipLaunchKernelGGL(token_dsum_kernel, dim3(vocab), dim3(threads), n_pos * sizeof(int), (hipStream_t)stream, rows, dxg, dsum,
                     n_pos, g);
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
