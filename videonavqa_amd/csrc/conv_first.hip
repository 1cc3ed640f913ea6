// conv_first.hip — first VGG conv (3 -> 64, 3x3 pad 1) + ReLU, reading the reference's clip
// layout [B][3][H][W][T] (frames LAST, eval/dataset.py:63,81-91) directly.
//
// The per-frame slice v[:, :, :, :, j] the reference takes (eval/q_and_v_eval.py:106) is a
// stride-T gather; here one workgroup owns a 2x16 pixel tile of one clip for ALL T frames, so
// HBM is read in (TX+2)*T-float contiguous runs, staged once in LDS as [c][py][px][t], and
// every wave then walks frames: 32 lanes = the 2x16 pixels of frame t.
// K = 27 is padded to 32 and run on MFMA with the weights as the A operand, so each lane ends
// up with 4 consecutive output channels of its pixel per register group (8/16-byte NHWC stores).
//
// HBM-bound: 12*H*W*T bytes read, 2*64*H*W*T (bf16) written per clip.
#include "vnqa_common.h"

namespace {

constexpr int TX = 16, TY = 2, CO = 64;

struct FirstArgs {
  const float* clip;
  const float* w;     // [64][27]
  const float* bias;  // [64]
  const int* img_of;  // [B*T]
  char* y;
  int B, T, H, W;
};

template <typename T>
__global__ void __launch_bounds__(256) conv_first_kernel(const FirstArgs p) {
  extern __shared__ __attribute__((aligned(16))) float patch[];  // [3][TY+2][TX+2][T]
  const int T_ = p.T;
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY, b = blockIdx.z;
  const int row_floats = (TX + 2) * T_;

  // ---- stage the input patch (zero outside the image) ----
  // element i of a patch row is (px = i / T, t = i % T); its column validity does not depend on the row,
  // so it is computed once per thread (no per-element integer division in the copy loop)
  constexpr int MAXIT = 8;                         // row_floats <= 8 * 256 (T <= 113)
  const int nit = (row_floats + 255) / 256;
  unsigned colmask = 0;
  for (int k = 0; k < nit && k < MAXIT; ++k) {
    const int i = threadIdx.x + 256 * k;
    const int gx = x0 + i / T_ - 1;
    if (i < row_floats && gx >= 0 && gx < p.W) colmask |= 1u << k;
  }
  for (int rc = 0; rc < 3 * (TY + 2); ++rc) {
    const int c = rc / (TY + 2), py = rc - c * (TY + 2);
    const int gy = y0 + py - 1;
    const bool rowok = gy >= 0 && gy < p.H;
    const float* src = p.clip + (((size_t)b * 3 + c) * p.H + gy) * (size_t)p.W * T_ + (size_t)(x0 - 1) * T_;
    float* dst = patch + rc * row_floats;
    for (int k = 0; k < nit; ++k) {
      const int i = threadIdx.x + 256 * k;
      if (i < row_floats) dst[i] = (rowok && ((colmask >> k) & 1u)) ? src[i] : 0.f;
    }
  }
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 31, h = lane >> 5;
  const int xx = col & (TX - 1), yy = col >> 4;  // TX == 16
  const int gx = x0 + xx, gy = y0 + yy;
  const bool inside = (gx < p.W) && (gy < p.H);
  const int pix_base = (yy * (TX + 2) + xx) * T_;

  if constexpr (sizeof(T) == 2) {
    // A fragments: W[cout = tn*32 + col][k = 16 s + 8 h + e]
    vnqa_bf16x8 wa[2][2];
    int koff[2][8];
    bool kval[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * s + 8 * h + e;
        kval[s][e] = k < 27;
        const int kk = k < 27 ? k : 0;
        const int c = kk / 9, r = (kk - c * 9) / 3, s_ = kk - c * 9 - r * 3;
        koff[s][e] = ((c * (TY + 2) + r) * (TX + 2) + s_) * T_;
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          wa[tn][s][e] = (short)(k < 27 ? f32_to_bf16(p.w[(tn * 32 + col) * 27 + k]) : 0);
      }
    float bias4[2][4][4];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) bias4[tn][g][e] = p.bias[tn * 32 + 8 * g + 4 * h + e];

    for (int t = wave; t < T_; t += 4) {
      const int img = p.img_of[b * T_ + t];
      if (img < 0) continue;  // wave-uniform
      vnqa_bf16x8 xb[2];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = patch[pix_base + koff[s][e] + t];
          xb[s][e] = (short)(kval[s][e] ? f32_to_bf16(v) : 0);
        }
      // both cout tiles -> wave-private LDS tile [32 px][64 co] (144-B rows), then 16-byte row-contiguous stores
      char* ctile = (char*)(patch + 3 * (TY + 2) * row_floats) + wave * (32 * 144);
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        vnqa_f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        acc = VNQA_MFMA_32x32x16(wa[tn][0], xb[0], acc);
        acc = VNQA_MFMA_32x32x16(wa[tn][1], xb[1], acc);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 pk;
          const float v0 = fmaxf(acc[4 * g + 0] + bias4[tn][g][0], 0.f);
          const float v1 = fmaxf(acc[4 * g + 1] + bias4[tn][g][1], 0.f);
          const float v2 = fmaxf(acc[4 * g + 2] + bias4[tn][g][2], 0.f);
          const float v3 = fmaxf(acc[4 * g + 3] + bias4[tn][g][3], 0.f);
          pk.x = pack2_h16(v0, v1);
          pk.y = pack2_h16(v2, v3);
          *(uint2*)(ctile + col * 144 + (tn * 32 + 8 * g + 4 * h) * 2) = pk;
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = lane + 64 * k;                 // 16-byte chunk of the 32 x 128-B tile
        const int prow = c >> 3, chunk = c & 7;
        const uint4 v = *(const uint4*)(ctile + prow * 144 + chunk * 16);
        const int oy = y0 + (prow >> 4), ox = x0 + (prow & 15);
        if (oy < p.H && ox < p.W)
          *(uint4*)((unsigned short*)p.y + ((((size_t)img * (p.H + 2)) + oy + 1) * (p.W + 2) + ox + 1) * CO + chunk * 8) = v;
      }
    }
  } else {
    // exact f32: v_mfma_f32_32x32x2_f32, k = 2 q + h, q = 0..13 (k = 27 is zero padding)
    float wa[2][14];
    int koff[14];
    bool kval[14];
#pragma unroll
    for (int q = 0; q < 14; ++q) {
      const int k = 2 * q + h;
      kval[q] = k < 27;
      const int kk = k < 27 ? k : 0;
      const int c = kk / 9, r = (kk - c * 9) / 3, s_ = kk - c * 9 - r * 3;
      koff[q] = ((c * (TY + 2) + r) * (TX + 2) + s_) * T_;
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) wa[tn][q] = k < 27 ? p.w[(tn * 32 + col) * 27 + k] : 0.f;
    }
    for (int t = wave; t < T_; t += 4) {
      const int img = p.img_of[b * T_ + t];
      if (img < 0) continue;
      float xb[14];
#pragma unroll
      for (int q = 0; q < 14; ++q) {
        const float v = patch[pix_base + koff[q] + t];
        xb[q] = kval[q] ? v : 0.f;
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn) {
        vnqa_f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int q = 0; q < 14; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[tn][q], xb[q], acc, 0, 0, 0);
        if (inside) {
          float* dst = (float*)p.y +
              ((((size_t)img * (p.H + 2)) + gy + 1) * (p.W + 2) + gx + 1) * CO + tn * 32 + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float4 o;
            o.x = fmaxf(acc[4 * g + 0] + p.bias[tn * 32 + 8 * g + 4 * h + 0], 0.f);
            o.y = fmaxf(acc[4 * g + 1] + p.bias[tn * 32 + 8 * g + 4 * h + 1], 0.f);
            o.z = fmaxf(acc[4 * g + 2] + p.bias[tn * 32 + 8 * g + 4 * h + 2], 0.f);
            o.w = fmaxf(acc[4 * g + 3] + p.bias[tn * 32 + 8 * g + 4 * h + 3], 0.f);
            *(float4*)(dst + 8 * g) = o;
          }
        }
      }
    }
  }
}

}  // namespace

extern "C" int vnqa_conv_first_fwd(const float* clip, const float* w, const float* bias,
                                   const int32_t* img_of, void* y, int32_t b, int32_t t, int32_t h,
                                   int32_t wd, int32_t c_out, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(clip && w && bias && img_of && y, "conv_first_fwd: null pointer");
  VNQA_CHECK_ARG(c_out == CO, "conv_first_fwd: c_out must be 64 (got %d)", c_out);
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "conv_first_fwd: bad dtype %d", dtype);
  VNQA_CHECK_ARG(b > 0 && t > 0 && h > 0 && wd > 0, "conv_first_fwd: empty problem");
  const size_t lds = (size_t)3 * (TY + 2) * (TX + 2) * t * sizeof(float) + 4 * 32 * 144;   // patch + 4 wave output tiles
  VNQA_CHECK_ARG(lds <= 160 * 1024 && 18 * t <= 8 * 256, "conv_first_fwd: t=%d frames do not fit the LDS patch", t);
  FirstArgs a{clip, w, bias, img_of, (char*)y, b, t, h, wd};
  dim3 grid((wd + TX - 1) / TX, (h + TY - 1) / TY, b);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16) {
    auto kern = conv_first_kernel<vnqa_bf16>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  } else {
    auto kern = conv_first_kernel<float>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);
  }
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
