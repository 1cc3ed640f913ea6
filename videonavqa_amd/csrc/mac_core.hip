// mac_core.hip — one MAC reasoning step (ControlUnit, ReadUnit, WriteUnit.concat of the reference's models/mac.py:28-42,
// 53-62,82-85, evaluated for all packed images at once) as ONE C-ABI call per direction.
//
// The step is 5 launches forward and 6 backward (plus 9 with per-step weight gradients) (fp32 GEMMs of [n_img, dim] x [dim, dim], the fused attention-pool
// kernels of mac_read.hip, a few elementwise products).  Issued one by one from Python the 12 steps of a training pass cost the
// launch thread more time than the GPU needs to run them (16.7 ms of host time against 13.8 ms of kernels per step of
// `bench.py --model mac`); here the whole sequence is enqueued from C++.
//
//   cq      = control Wc^T + pq                       (pq = position_aware_i(question) Wp^T + b, hoisted by the caller)
//   control'= pool(ctx, cq * w_ca, b_ca) [* mask]     (attention over the question words)
//   mem     = memory Wm^T + bm ;  v = control' * w_ra ;  u = mem * (v W1)
//   read    = pool(know, pre; u, v, b_ra)             (re-associated ReadUnit, see models/mac.py of this repo)
//   concat  = read Wr^T + memory Wmm^T + bw
#include "vnqa_common.h"

namespace {

// out[i] = (acc ? out[i] : 0) + x[i] * y[cols ? i % cols : i] (+ z[i])
__global__ void ew_mul_kernel(float* __restrict__ out, const float* __restrict__ x, const float* __restrict__ y,
                              const float* __restrict__ z, int n, int cols, int acc) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float v = x[i] * y[cols > 0 ? i % cols : i];
    if (z != nullptr) v += z[i];
    if (acc) v += out[i];
    out[i] = v;
  }
}

int ew_mul(float* out, const float* x, const float* y, const float* z, int n, int cols, int acc, hipStream_t st) {
  int g = (n + 255) / 256;
  g = g > 1024 ? 1024 : g;
  hipLaunchKernelGGL(ew_mul_kernel, dim3(g), dim3(256), 0, st, out, x, y, z, n, cols, acc);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

// partial[z][j] = sum over row share z of x[r][j] * (y ? y[r][j] : 1): 64 columns x 16 row lanes per block, grid (cols / 64, shares);
// fold_shares_kernel sums the shares in order (deterministic)
constexpr int MC_SHARES = 32;
__global__ void __launch_bounds__(1024) dot_colsum_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                          float* __restrict__ partial, int rows, int cols) {
  __shared__ float s_part[16][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  const int per = (rows + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per;
  int r1 = r0 + per;
  r1 = r1 < rows ? r1 : rows;
  float s = 0.f;
  if (n < cols) {
    if (y != nullptr)
      for (int m = r0 + ry; m < r1; m += 16) s = fmaf(x[(size_t)m * cols + n], y[(size_t)m * cols + n], s);
    else
      for (int m = r0 + ry; m < r1; m += 16) s += x[(size_t)m * cols + n];
  }
  s_part[ry][cx] = s;
  __syncthreads();
  if (ry == 0 && n < cols) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += s_part[r][cx];
    partial[(size_t)blockIdx.y * cols + n] = t;
  }
}
__global__ void fold_shares_kernel(const float* __restrict__ partial, float* __restrict__ out, int shares, int cols) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= cols) return;
  float t = 0.f;
  for (int z = 0; z < shares; ++z) t += partial[(size_t)z * cols + n];
  out[n] = t;
}

// o1 = x * y1, o2 = x * y2 (the two products of one backward factor in one launch)
__global__ void ew_mul2_kernel(float* __restrict__ o1, float* __restrict__ o2, const float* __restrict__ x,
                               const float* __restrict__ y1, const float* __restrict__ y2, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float v = x[i];
    o1[i] = v * y1[i];
    o2[i] = v * y2[i];
  }
}

#define MC_TRY(call)             \
  do {                           \
    const int rc__ = (call);     \
    if (rc__ != VNQA_OK) return rc__; \
  } while (0)

// C[m,n] = A[m,k] B^T (B [n,k])  + bias / addend
// (ws: split-K scratch of vnqa_mac_core_workspace bytes — the [n_img, dim] x [dim, dim] products are 40 output tiles)
int gemm_nt(const float* a, const float* b, float* c, const float* bias, const float* addend, int m, int n, int k, int acc,
            void* ws, void* st) {
  return vnqa_sgemm(a, b, c, bias, nullptr, nullptr, nullptr, k, 1, 1, k, n, m, n, k, 0, acc, addend, ws, st);
}
// C = A B^T + addend and, from the same epilogue, out2 = C * colscale[n]
int gemm_nt_scaled(const float* a, const float* b, float* c, const float* addend, float* out2, const float* colscale, int m, int n,
                   int k, void* ws, void* st) {
  return vnqa_sgemm2(a, b, c, nullptr, k, 1, 1, k, n, m, n, k, addend, out2, colscale, nullptr, ws, st);
}
// C = A B and out2 = C * mul (elementwise matrix)
int gemm_nn_mul(const float* a, const float* b, float* c, float* out2, const float* mul, int m, int n, int k, void* ws, void* st) {
  return vnqa_sgemm2(a, b, c, nullptr, k, 1, n, 1, n, m, n, k, nullptr, out2, nullptr, mul, ws, st);
}
// C[m,n] = A[m,k] B (B [k,n])
int gemm_nn(const float* a, const float* b, float* c, int m, int n, int k, int acc, void* ws, void* st) {
  return vnqa_sgemm(a, b, c, nullptr, nullptr, nullptr, nullptr, k, 1, n, 1, n, m, n, k, 0, acc, nullptr, ws, st);
}
// C[m,n] (+)= A^T B (A [k,m], B [k,n])
int gemm_tn(const float* a, const float* b, float* c, int m, int n, int k, int acc, void* ws, void* st) {
  return vnqa_sgemm(a, b, c, nullptr, nullptr, nullptr, nullptr, 1, m, n, 1, n, m, n, k, 0, acc, nullptr, ws, st);
}

// problem descriptors of vnqa_sgemm_batch in the forms used here
vnqa_sgemm_problem prob_nt(const float* a, const float* b, float* c, const float* bias, const float* addend, int m, int n, int k,
                           int acc) {                                   // C (+)= A B^T (+ bias / addend), B [n,k]
  vnqa_sgemm_problem q = {a, b, c, bias, addend, nullptr, nullptr, nullptr, k, 1, 1, k, n, m, n, k, 0, acc};
  return q;
}
vnqa_sgemm_problem prob_nn(const float* a, const float* b, float* c, int m, int n, int k, int acc) {     // C (+)= A B, B [k,n]
  vnqa_sgemm_problem q = {a, b, c, nullptr, nullptr, nullptr, nullptr, nullptr, k, 1, n, 1, n, m, n, k, 0, acc};
  return q;
}

}  // namespace

extern "C" int64_t vnqa_mac_core_workspace(int32_t n, int32_t d) {
  int64_t need = vnqa_sgemm_workspace(n, d, d);
  const int64_t b = vnqa_sgemm_workspace(d, d, n), c = vnqa_sgemm_workspace(d, 1, n);
  need = b > need ? b : need;
  return c > need ? c : need;
}

extern "C" int vnqa_mac_core_fwd(const vnqa_mac_core* a, void* stream) {
  VNQA_CHECK_ARG(a != nullptr, "mac_core_fwd: null argument block");
  VNQA_CHECK_ARG(a->control && a->memory && a->pq && a->ctxw && a->know && a->pre && a->wc && a->w_ca && a->b_ca && a->wm &&
                     a->bm && a->w1 && a->w_ra && a->b_ra && a->wr && a->wmm && a->bw,
                 "mac_core_fwd: null input / parameter");
  VNQA_CHECK_ARG(a->cq && a->qv && a->p_c && a->cnew && a->mem && a->v && a->t && a->u && a->p_r && a->read && a->concat,
                 "mac_core_fwd: null output");
  const int N = a->n, d = a->d;
  {
    // the three products that only need the step's inputs, in ONE launch: cq = pq + control Wc^T (and qv = cq * w_ca from
    // the same epilogue), mem = memory Wm^T + bm, concat = memory Wmm^T + bw (read Wr^T is added at the end)
    vnqa_sgemm_problem q[3] = {prob_nt(a->control, a->wc, a->cq, nullptr, a->pq, N, d, d, 0),
                               prob_nt(a->memory, a->wm, a->mem, a->bm, nullptr, N, d, d, 0),
                               prob_nt(a->memory, a->wmm, a->concat, a->bw, nullptr, N, d, d, 0)};
    q[0].out2 = a->qv;
    q[0].out2_col = a->w_ca;
    MC_TRY(vnqa_sgemm_batch(q, 3, stream));
  }
  // control' = pool(ctx, qv) [* mask] and v = control' * w_ra from the pool kernel's epilogue
  MC_TRY(vnqa_mac_read_fwd_scaled(a->ctxw, nullptr, a->qv, nullptr, a->b_ca, a->p_c, a->cnew, a->mask_c, a->v, a->w_ra, N, a->lq,
                                  d, d, VNQA_F32, stream));
  MC_TRY(gemm_nn_mul(a->v, a->w1, a->t, a->u, a->mem, N, d, d, a->workspace, stream));                      // t = v W1; u = mem * t
  MC_TRY(vnqa_mac_read_fwd(a->know, a->pre, a->u, a->v, a->b_ra, a->p_r, a->read, N, a->s, d, a->ld, a->dtype, stream));
  MC_TRY(gemm_nt(a->read, a->wr, a->concat, nullptr, nullptr, N, d, d, 1, a->workspace, stream));          // concat += read Wr^T
  return VNQA_OK;
}

extern "C" int vnqa_mac_core_bwd(const vnqa_mac_core* a, void* stream) {
  VNQA_CHECK_ARG(a != nullptr, "mac_core_bwd: null argument block");
  VNQA_CHECK_ARG(a->d_concat && a->d_control && a->d_memory && a->d_cq && a->ds_r && a->d_read && a->ds_c && a->d_c && a->du &&
                     a->dv && a->dqv && a->d_mem && a->d_t && (a->ones || a->defer_wgrad),
                 "mac_core_bwd: null gradient buffer");
  const bool defer = a->defer_wgrad != 0;      // parameter gradients come from ONE vnqa_mac_core_wgrad call over all steps instead
  VNQA_CHECK_ARG(defer || (a->g_wc && a->g_wca && a->g_wm && a->g_bm && a->g_w1 && a->g_wra && a->g_wr && a->g_wmm && a->g_bw),
                 "mac_core_bwd: null parameter-gradient accumulator");
  const int N = a->n, d = a->d;
  hipStream_t st = (hipStream_t)stream;
  // WriteUnit.concat
  {
    vnqa_sgemm_problem q[2] = {prob_nn(a->d_concat, a->wr, a->d_read, N, d, d, 0), prob_nn(a->d_concat, a->wmm, a->d_memory, N, d, d, 0)};
    MC_TRY(vnqa_sgemm_batch(q, 2, stream));
  }
  if (!defer) {
    MC_TRY(gemm_tn(a->d_concat, a->read, a->g_wr, d, d, N, 1, a->workspace, stream));
    MC_TRY(gemm_tn(a->d_concat, a->memory, a->g_wmm, d, d, N, 1, a->workspace, stream));
    MC_TRY(gemm_tn(a->d_concat, a->ones, a->g_bw, d, 1, N, 1, a->workspace, stream));
  }
  // ReadUnit attention
  MC_TRY(vnqa_mac_read_bwd(a->know, a->pre, a->p_r, a->d_read, a->ds_r, a->du, a->dv, N, a->s, d, a->ld, a->dtype, stream));
  {                                                                                          // d mem = du * t, d t = du * mem
    int g = (N * d + 255) / 256;
    g = g > 1024 ? 1024 : g;
    hipLaunchKernelGGL(ew_mul2_kernel, dim3(g), dim3(256), 0, st, a->d_mem, a->d_t, (const float*)a->du, (const float*)a->t,
                       (const float*)a->mem, N * d);
    VNQA_CHECK_LAUNCH();
  }
  {                                                                  // dv += d t W1^T and d memory += d mem Wm in one launch
    vnqa_sgemm_problem q[2] = {prob_nt(a->d_t, a->w1, a->dv, nullptr, nullptr, N, d, d, 1), prob_nn(a->d_mem, a->wm, a->d_memory, N, d, d, 1)};
    MC_TRY(vnqa_sgemm_batch(q, 2, stream));
  }
  if (!defer) {
    MC_TRY(gemm_tn(a->v, a->d_t, a->g_w1, d, d, N, 1, a->workspace, stream));
    MC_TRY(ew_mul(a->g_wra, a->dv, a->cnew, nullptr, N * d, 0, 1, st));                      // per-image w_ra gradient terms
  }
  if (!defer) {
    MC_TRY(gemm_tn(a->d_mem, a->memory, a->g_wm, d, d, N, 1, a->workspace, stream));
    MC_TRY(gemm_tn(a->d_mem, a->ones, a->g_bm, d, 1, N, 1, a->workspace, stream));
  }
  // ControlUnit attention; its prologue forms d control' = (dv * w_ra + upstream) [* mask] (kept in d_c for the gradient
  // factors), its epilogue d cq = dqv * w_ca
  MC_TRY(vnqa_mac_read_bwd_fused(a->ctxw, nullptr, a->p_c, a->d_c, a->dv, a->w_ra, a->d_cnew, a->mask_c, a->ds_c, a->dqv, nullptr,
                                 a->d_cq, a->w_ca, N, a->lq, d, d, VNQA_F32, stream));
  if (!defer) MC_TRY(ew_mul(a->g_wca, a->dqv, a->cq, nullptr, N * d, 0, 1, st));
  MC_TRY(gemm_nn(a->d_cq, a->wc, a->d_control, N, d, d, 0, a->workspace, stream));
  if (!defer) MC_TRY(gemm_tn(a->d_cq, a->control, a->g_wc, d, d, N, 1, a->workspace, stream));
  return VNQA_OK;
}

// ---- all reasoning steps in one call per direction (no self-attention, no memory gate: the reference's defaults) ----------------
// With the step a single launch sequence, what is left of a training pass's cost is the HOST: 24 autograd-node invocations of
// ~90 us of Python each plus the ATen launches between them (mask multiplies, gradient sums) — the chain ran at the speed of
// the launch thread (the same trunk on a quarter of the frames took as long).  Here the loop over steps is in C++.
namespace {

// o = x * y, or a copy when y == null
__global__ void mul_or_copy_kernel(float* __restrict__ o, const float* __restrict__ x, const float* __restrict__ y, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) o[i] = y != nullptr ? x[i] * y[i] : x[i];
}
void mul_or_copy(float* o, const float* x, const float* y, int n, hipStream_t st) {
  int g = (n + 255) / 256;
  g = g > 1024 ? 1024 : g;
  hipLaunchKernelGGL(mul_or_copy_kernel, dim3(g), dim3(256), 0, st, o, x, y, n);
}

// step i's argument block: every per-step buffer sits i * (rows of step 0) further on in the caller's step-stacked slabs
vnqa_mac_core step_args(const vnqa_mac_core* a0, int i, const float* memories) {
  vnqa_mac_core a = *a0;
  const size_t nd = (size_t)a.n * a.d, nl = (size_t)a.n * a.lq, ns = (size_t)a.n * a.s;
  auto at = [&](float* p, size_t per) { return p ? p + i * per : nullptr; };
  a.control = i == 0 ? a0->control : a0->cnew + (size_t)(i - 1) * nd;
  a.memory = memories + (size_t)i * nd;
  a.pq = a0->pq + i * nd;
  a.cq = at(a0->cq, nd); a.qv = at(a0->qv, nd); a.p_c = at(a0->p_c, nl); a.cnew = at(a0->cnew, nd); a.mem = at(a0->mem, nd);
  a.v = at(a0->v, nd); a.t = at(a0->t, nd); a.u = at(a0->u, nd); a.p_r = at(a0->p_r, ns); a.read = at(a0->read, nd);
  a.concat = at(a0->concat, nd);
  a.d_control = at(a0->d_control, nd); a.d_memory = at(a0->d_memory, nd); a.d_cq = at(a0->d_cq, nd);
  a.ds_r = at(a0->ds_r, ns); a.d_read = at(a0->d_read, nd); a.ds_c = at(a0->ds_c, nl); a.d_c = at(a0->d_c, nd);
  a.du = at(a0->du, nd); a.dv = at(a0->dv, nd); a.dqv = at(a0->dqv, nd); a.d_mem = at(a0->d_mem, nd); a.d_t = at(a0->d_t, nd);
  return a;
}

}  // namespace

extern "C" int vnqa_mac_chain_fwd(const vnqa_mac_core* step0, int32_t n_steps, float* memories, const float* mask_m, void* stream) {
  VNQA_CHECK_ARG(step0 != nullptr && n_steps > 0 && memories != nullptr, "mac_chain_fwd: null argument / no steps");
  VNQA_CHECK_ARG(step0->pq && step0->cnew && step0->concat, "mac_chain_fwd: null slab");
  const size_t nd = (size_t)step0->n * step0->d;
  for (int i = 0; i < n_steps; ++i) {
    const vnqa_mac_core a = step_args(step0, i, memories);
    MC_TRY(vnqa_mac_core_fwd(&a, stream));
    mul_or_copy(memories + (size_t)(i + 1) * nd, a.concat, mask_m, (int)nd, (hipStream_t)stream);      // memory_{i+1} = concat_i [* mask]
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_mac_chain_bwd(const vnqa_mac_core* step0, int32_t n_steps, const float* memories, const float* mask_m,
                                  const float* d_memory_out, float* d_concat, void* stream) {
  VNQA_CHECK_ARG(step0 != nullptr && n_steps > 0 && memories != nullptr && d_memory_out != nullptr && d_concat != nullptr,
                 "mac_chain_bwd: null argument / no steps");
  VNQA_CHECK_ARG(step0->defer_wgrad != 0, "mac_chain_bwd: parameter gradients come from vnqa_mac_core_wgrad (defer_wgrad = 1)");
  VNQA_CHECK_ARG(step0->d_control && step0->d_memory, "mac_chain_bwd: null slab");
  const size_t nd = (size_t)step0->n * step0->d;
  for (int i = n_steps - 1; i >= 0; --i) {
    vnqa_mac_core a = step_args(step0, i, memories);
    const bool last = i == n_steps - 1;
    float* dc = d_concat + (size_t)i * nd;
    // the step's concat fed memory_{i+1} = concat [* mask] only; its control' fed step i+1's control only
    mul_or_copy(dc, last ? d_memory_out : step0->d_memory + (size_t)(i + 1) * nd, mask_m, (int)nd, (hipStream_t)stream);
    a.d_concat = dc;
    a.d_cnew = last ? nullptr : step0->d_control + (size_t)(i + 1) * nd;
    MC_TRY(vnqa_mac_core_bwd(&a, stream));
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

// Parameter gradients of ALL reasoning steps at once: every factor is the per-step [n][d] matrices stacked to [rows = steps * n][d]
// (the caller keeps them step-major in one slab), so each weight gradient is ONE product over K = rows instead of `steps`
// accumulating products on the backward pass's dependent chain (12 steps x 9 launches off that chain).
extern "C" int64_t vnqa_mac_core_wgrad_workspace(int32_t rows, int32_t d) {
  return vnqa_sgemm_workspace(d, d, rows) + (int64_t)MC_SHARES * d * 4;      // split-K scratch, then the column sums' row shares
}

extern "C" int vnqa_mac_core_wgrad(const vnqa_mac_wgrad* w, void* stream) {
  VNQA_CHECK_ARG(w != nullptr && w->rows > 0 && w->d > 0, "mac_wgrad: bad argument block");
  VNQA_CHECK_ARG(w->d_concat && w->read && w->memory && w->v && w->d_t && w->d_mem && w->d_cq && w->control && w->dv && w->cnew &&
                     w->dqv && w->cq, "mac_wgrad: null factor");
  VNQA_CHECK_ARG(w->g_wc && w->g_wca && w->g_wm && w->g_bm && w->g_w1 && w->g_wra && w->g_wr && w->g_wmm && w->g_bw,
                 "mac_wgrad: null output");
  const int R = w->rows, d = w->d;
  hipStream_t st = (hipStream_t)stream;
  MC_TRY(gemm_tn(w->d_concat, w->read, w->g_wr, d, d, R, 0, w->workspace, stream));
  MC_TRY(gemm_tn(w->d_concat, w->memory, w->g_wmm, d, d, R, 0, w->workspace, stream));
  MC_TRY(gemm_tn(w->v, w->d_t, w->g_w1, d, d, R, 0, w->workspace, stream));
  MC_TRY(gemm_tn(w->d_mem, w->memory, w->g_wm, d, d, R, 0, w->workspace, stream));
  MC_TRY(gemm_tn(w->d_cq, w->control, w->g_wc, d, d, R, 0, w->workspace, stream));
  VNQA_CHECK_ARG(w->workspace != nullptr, "mac_wgrad: workspace required (vnqa_mac_core_wgrad_workspace bytes)");
  float* shares = (float*)((char*)w->workspace + vnqa_sgemm_workspace(d, d, R));
  auto colsum2 = [&](const float* x, const float* y, float* out) {
    hipLaunchKernelGGL(dot_colsum_kernel, dim3((d + 63) / 64, MC_SHARES), dim3(1024), 0, st, x, y, shares, R, d);
    hipLaunchKernelGGL(fold_shares_kernel, dim3((d + 255) / 256), dim3(256), 0, st, (const float*)shares, out, MC_SHARES, d);
  };
  colsum2(w->d_concat, nullptr, w->g_bw);
  colsum2(w->d_mem, nullptr, w->g_bm);
  colsum2(w->dv, w->cnew, w->g_wra);
  colsum2(w->dqv, w->cq, w->g_wca);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
