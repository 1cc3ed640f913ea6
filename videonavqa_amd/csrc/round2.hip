// round2.hip — second-order rounding of frozen weights onto the 16-bit grid (stem.second_order_round; GPTQ / OBQ sequential rounding).
//
//   for j = 0 .. K-1:   q_j = h16(w_j);   e = (w_j - q_j) / U_jj;   w_t -= e U_jt   for t > j
//
// U = the upper Cholesky factor of H^-1 (H = the layer's input-patch second moment, damped; columns in order of decreasing H_jj):
// the rounding error of column j is pushed onto the columns not yet rounded, which greedily minimises dw^T H dw — the mean squared
// error the rounding adds to the layer's output.  The recursion is sequential in j but independent per OUTPUT CHANNEL (row of w),
// so one workgroup owns a row: the row lives in LDS as float64, every step is one barrier and K / 256 fused multiply-adds per
// thread against row j of U (read by all workgroups at about the same time: L2).  One launch per layer — 9 ms for K = 4608 —
// instead of the 25 000 elementwise launches of the tensor-library form.  fp64 throughout; the grid is this build's 16-bit format.
#include "vnqa_common.h"

namespace {

__global__ void __launch_bounds__(256) second_order_round_kernel(const double* __restrict__ w, const double* __restrict__ U,
                                                                 float* __restrict__ q, int K) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* s_w = (double*)smem_raw;
  const size_t row = blockIdx.x;
  for (int t = threadIdx.x; t < K; t += 256) s_w[t] = w[row * K + t];
  for (int j = 0; j < K; ++j) {
    __syncthreads();
    const double wj = s_w[j];
    const float qf = bf16_to_f32(f32_to_bf16((float)wj));      // (the build's 16-bit format: fp32 -> storage -> fp32)
    const double* __restrict__ u = U + (size_t)j * K;
    const double e = (wj - (double)qf) / u[j];
    if (threadIdx.x == 0) q[row * K + j] = qf;
    for (int t = j + 1 + threadIdx.x; t < K; t += 256) s_w[t] -= e * u[t];
  }
}

}  // namespace

extern "C" int vnqa_second_order_round(const double* w, const double* u, float* q, int32_t rows, int32_t k, void* stream) {
  VNQA_CHECK_ARG(w && u && q && rows > 0 && k > 0, "second_order_round: bad arguments");
  VNQA_CHECK_ARG(k <= 16384, "second_order_round: K = %d does not fit the row buffer (16384 float64)", k);
  const int lds = k * (int)sizeof(double);
  static std::atomic<bool> attr_done{false};
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)second_order_round_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8) != hipSuccess) {
      vnqa_set_error("second_order_round: cannot reserve %d B of LDS", 16384 * 8);
      return VNQA_ERR_HIP;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(second_order_round_kernel, dim3(rows), dim3(256), lds, (hipStream_t)stream, w, u, q, k);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
