// optim.hip — global-norm gradient clip + Adam + zero_grad fused over flat fp32 buffers.
// HBM-bound: reads g twice (norm, update), p/m/v once, writes p/m/v/g once: 36 B per parameter.
#include "vnqa_common.h"

namespace {

constexpr int NORM_BLOCK = 256;
constexpr int MAX_PARTIAL = 1024;

__global__ void l2norm_partial_kernel(const float* __restrict__ g, long long n, float* __restrict__ partial) {
  __shared__ float red[NORM_BLOCK / 64];
  float s = 0.f;
  const long long n4 = n >> 2;
  const float4* g4 = (const float4*)g;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = g4[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0)
    for (long long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  s = wave_reduce_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < NORM_BLOCK / 64; ++w) t += red[w];
    partial[blockIdx.x] = t;
  }
}

__global__ void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, long long n, const float* __restrict__ partial, int n_partial,
                                 float clip, float lr, float beta1, float beta2, float eps, int step,
                                 int* __restrict__ overflow_count) {
  __shared__ float s_coef, s_step_size, s_inv_sqrt_bc2;
  if (threadIdx.x < 64) {
    float t = 0.f;
    for (int i = threadIdx.x; i < n_partial; i += 64) t += partial[i];
    t = wave_reduce_sum(t);
    if (threadIdx.x == 0) {
      const float c = clip / (sqrtf(t) + 1e-6f);
      // a non-finite gradient norm (fp16 storage: the loss scale overflowed somewhere in the backward pass; any precision: an
      // inf / NaN activation) — every block sees the same partials and takes the same decision: SKIP the update (coef = -1),
      // still zero g.  Unconditional: `c < 1 ? c : 1` would turn a NaN norm into coef 1 and apply NaN gradients.
      s_coef = !(t <= 3.0e38f) ? -1.f : (c < 1.f ? c : 1.f);
      // Bias correction from the number of updates actually APPLIED = launches so far (`step`, counted by the host) minus the
      // skipped ones (counted HERE, on the device): no host read-back enters the update, so data-parallel replicas — which see
      // the same reduced gradient and therefore skip the same launches — stay bit-identical whatever their host timing.
      // (*overflow_count is only written by a launch that skips, which never reads it: all blocks of an applying launch agree.)
      const int skipped = overflow_count != nullptr ? *(volatile int*)overflow_count : 0;
      const int applied = step - skipped > 1 ? step - skipped : 1;
      const double bc1 = 1.0 - pow((double)beta1, (double)applied);
      const double bc2 = 1.0 - pow((double)beta2, (double)applied);
      s_step_size = (float)((double)lr / bc1);
      s_inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    }
  }
  __syncthreads();
  const float coef = s_coef, step_size = s_step_size, inv_sqrt_bc2 = s_inv_sqrt_bc2;
  if (coef < 0.f) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && overflow_count != nullptr) atomicAdd(overflow_count, 1);   // (the host reads it back with a fixed lag)
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) g[i] = 0.f;
    return;
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = beta1 * m[i] + (1.f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
    g[i] = 0.f;
  }
}

int norm_blocks(long long n) {
  long long b = (n / 4 + NORM_BLOCK - 1) / NORM_BLOCK;
  return (int)(b < 1 ? 1 : (b > MAX_PARTIAL ? MAX_PARTIAL : b));
}

}  // namespace

extern "C" int32_t vnqa_l2norm_blocks(int64_t n) { return norm_blocks(n); }

extern "C" int vnqa_l2norm_partial(const float* g, int64_t n, float* partial, void* stream) {
  VNQA_CHECK_ARG(g && partial && n > 0, "l2norm_partial: bad arguments");
  VNQA_CHECK_ARG(((uintptr_t)g & 15) == 0, "l2norm_partial: g must be 16-byte aligned");
  hipLaunchKernelGGL(l2norm_partial_kernel, dim3(norm_blocks(n)), dim3(NORM_BLOCK), 0, (hipStream_t)stream, g,
                     (long long)n, partial);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_clip_adam(float* p, float* g, float* m, float* v, int64_t n, const float* partial,
                              int32_t n_partial, float clip, float lr, float beta1, float beta2, float eps,
                              int32_t step, int32_t* overflow_count, void* stream) {
  VNQA_CHECK_ARG(p && g && m && v && partial && n > 0 && n_partial > 0 && step >= 1, "clip_adam: bad arguments");
  long long blocks = (n + 255) / 256;
  blocks = blocks > 4096 ? 4096 : blocks;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                     partial, n_partial, clip, lr, beta1, beta2, eps, (int)step, (int*)overflow_count);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
