// pool_tail.hip — tail of the global-max-pooling models (FiLMGlobalPoolingPretrainedStem, TimeMultiHopFiLMPretrainedStem):
//   x = relu(c1x1_tail(x))  per frame, zero-padded to batch_size rows, stacked over frames, max over frames, Linear
//   (models/film_global_pooling_pt_stem.py:228-238, models/time_multi_hop_pt_stem.py:240-250).
// The relu'd tail maps of ALL valid (frame, sample) pairs live in ONE packed image list [n_img][h+2][w+2][c_pad] (16-bit
// storage, zero halo); the reference's dense [T, B, ...] stack, its zero padding rows and the index_put that fills it are
// never materialised: a sample's frames are the images frame_off[t] + b for every frame t it reaches.
//   frame_max_fwd : pooled[b][c*h*w + y*w + x] = max(0, max_t map[img(t, b)][y][x][c])  written straight in the reference's
//                   NCHW-flattened order (the operand order of out_linear.weight, no weight re-layout), + the arg-max image
//                   (-1 where every frame is <= 0: relu output and padding rows tie at 0 and carry no gradient either way;
//                   ties between frames resolve to the FIRST frame, as torch.max does on this path)
//   frame_max_bwd : d map[img][y][x][c] = scale * d pooled[b][...] where img is the arg-max, else 0 — the whole padded NHWC
//                   gradient tensor incl. its zero halo and zero padding channels, 16 bytes per lane.
#include "vnqa_common.h"

namespace {

template <typename T>
__global__ void __launch_bounds__(256) frame_max_fwd_kernel(const T* __restrict__ t, const int* __restrict__ frame_off,
                                                            float* __restrict__ pooled, int* __restrict__ argmax,
                                                            int B, int n_frames, int h, int w, int c_pad, int tail) {
  constexpr int EPC = 16 / (int)sizeof(T);                 // elements per 16-byte chunk
  const int chunks = (tail + EPC - 1) / EPC;               // channel chunks that hold real channels
  const int hw = h * w;
  const long long total = (long long)B * hw * chunks;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int ck = (int)(idx % chunks);
    const int pix = (int)((idx / chunks) % hw);
    const int b = (int)(idx / ((long long)chunks * hw));
    const int y = pix / w, x = pix - y * w;
    const size_t in_img = ((size_t)(y + 1) * (w + 2) + (x + 1)) * c_pad + (size_t)ck * EPC;
    const size_t img_stride = (size_t)(h + 2) * (w + 2) * c_pad;
    float best[EPC];
    int arg[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { best[e] = 0.f; arg[e] = -1; }
    for (int f = 0; f < n_frames; ++f) {
      const int o = frame_off[f];
      if (b >= frame_off[f + 1] - o) break;                // v_lens sorted descending: later frames do not hold sample b either
      const int img = o + b;
      const uint4 raw = *(const uint4*)(t + (size_t)img * img_stride + in_img);
      const T* v = (const T*)&raw;
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float val = ElemOps<T>::load(v[e]);
        if (val > best[e]) { best[e] = val; arg[e] = img; }
      }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      const int c = ck * EPC + e;
      if (c < tail) {
        const size_t o = (size_t)b * tail * hw + (size_t)c * hw + pix;
        pooled[o] = best[e];
        argmax[o] = arg[e];
      }
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) frame_max_bwd_kernel(const float* __restrict__ dpooled, const int* __restrict__ argmax,
                                                            const int* __restrict__ sample_of, T* __restrict__ dt,
                                                            int n_img, int h, int w, int c_pad, int tail, float scale) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cpc = c_pad / EPC;                             // chunks per pixel
  const int hw = h * w, wp = w + 2, hp = h + 2;
  const long long total = (long long)n_img * hp * wp * cpc;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int ck = (int)(idx % cpc);
    const long long pp = idx / cpc;
    const int px = (int)(pp % wp), py = (int)((pp / wp) % hp);
    const int img = (int)(pp / ((long long)wp * hp));
    T out[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) out[e] = ElemOps<T>::store(0.f);
    if (py >= 1 && py <= h && px >= 1 && px <= w && ck * EPC < tail) {
      const int b = sample_of[img];
      const int pix = (py - 1) * w + (px - 1);
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const int c = ck * EPC + e;
        if (c < tail) {
          const size_t o = (size_t)b * tail * hw + (size_t)c * hw + pix;
          if (argmax[o] == img) out[e] = ElemOps<T>::store(dpooled[o] * scale);
        }
      }
    }
    *(uint4*)(dt + (size_t)idx * EPC) = *(const uint4*)out;
  }
}

}  // namespace

extern "C" int vnqa_frame_max_fwd(const void* maps, const int32_t* frame_off, float* pooled, int32_t* argmax, int32_t batch,
                                  int32_t n_frames, int32_t h, int32_t w, int32_t c_pad, int32_t tail, int32_t dtype,
                                  void* stream) {
  VNQA_CHECK_ARG(maps && frame_off && pooled && argmax, "frame_max_fwd: null pointer");
  VNQA_CHECK_ARG(batch > 0 && n_frames > 0 && h > 0 && w > 0 && tail > 0 && tail <= c_pad, "frame_max_fwd: bad geometry");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "frame_max_fwd: bad dtype %d", dtype);
  VNQA_CHECK_ARG(c_pad % (dtype == VNQA_BF16 ? 8 : 4) == 0, "frame_max_fwd: c_pad must hold whole 16-byte chunks");
  const int epc = dtype == VNQA_BF16 ? 8 : 4;
  const long long total = (long long)batch * h * w * ((tail + epc - 1) / epc);
  long long g = (total + 255) / 256;
  g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(frame_max_fwd_kernel<vnqa_bf16>, dim3((int)g), dim3(256), 0, st, (const vnqa_bf16*)maps, frame_off, pooled,
                       argmax, batch, n_frames, h, w, c_pad, tail);
  else
    hipLaunchKernelGGL(frame_max_fwd_kernel<float>, dim3((int)g), dim3(256), 0, st, (const float*)maps, frame_off, pooled, argmax,
                       batch, n_frames, h, w, c_pad, tail);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_frame_max_bwd(const float* dpooled, const int32_t* argmax, const int32_t* sample_of, void* dmaps,
                                  int32_t n_img, int32_t h, int32_t w, int32_t c_pad, int32_t tail, float scale, int32_t dtype,
                                  void* stream) {
  VNQA_CHECK_ARG(dpooled && argmax && sample_of && dmaps, "frame_max_bwd: null pointer");
  VNQA_CHECK_ARG(n_img > 0 && h > 0 && w > 0 && tail > 0 && tail <= c_pad, "frame_max_bwd: bad geometry");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "frame_max_bwd: bad dtype %d", dtype);
  const int epc = dtype == VNQA_BF16 ? 8 : 4;
  VNQA_CHECK_ARG(c_pad % epc == 0, "frame_max_bwd: c_pad must hold whole 16-byte chunks");
  const long long total = (long long)n_img * (h + 2) * (w + 2) * (c_pad / epc);
  long long g = (total + 255) / 256;
  g = g > 2048 ? 2048 : (g < 1 ? 1 : g);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(frame_max_bwd_kernel<vnqa_bf16>, dim3((int)g), dim3(256), 0, st, dpooled, argmax, sample_of,
                       (vnqa_bf16*)dmaps, n_img, h, w, c_pad, tail, scale);
  else
    hipLaunchKernelGGL(frame_max_bwd_kernel<float>, dim3((int)g), dim3(256), 0, st, dpooled, argmax, sample_of, (float*)dmaps,
                       n_img, h, w, c_pad, tail, scale);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
