// pack.hip — layout conversion kernels between the reference's tensor layouts (OIHW fp32
// weights, [B][C][h][w][T] fp32 features) and the kernel-native ones (K-major weights,
// padded NHWC activations).  All memory-bound, one pass each.
#include "vnqa_common.h"

namespace {

template <typename T>
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, int c_out, int c_in, int taps,
                                        int rows_pad, int k_ch_pad, const float* __restrict__ out_scale,
                                        int transpose_flip, T* __restrict__ wt) {
  // destination index space: [row][tap][ch]
  const size_t total = (size_t)rows_pad * taps * k_ch_pad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ch = (int)(i % k_ch_pad);
    const int tap = (int)((i / k_ch_pad) % taps);
    const int row = (int)(i / ((size_t)k_ch_pad * taps));
    float v = 0.f;
    if (!transpose_flip) {
      if (row < c_out && ch < c_in) {
        v = w[((size_t)row * c_in + ch) * taps + tap];
        if (out_scale) v *= out_scale[row];
      }
    } else {
      // row = input channel, ch = output channel, tap index flipped (180 degree rotation)
      if (row < c_in && ch < c_out) v = w[((size_t)ch * c_in + row) * taps + (taps - 1 - tap)];
    }
    wt[i] = ElemOps<T>::store(v);
  }
}

__global__ void unpack_conv_wgrad_kernel(const float* __restrict__ dwt, int c_out, int c_in, int taps,
                                         int c_out_pad, int c_in_pad, float* __restrict__ dw, float alpha,
                                         const float* __restrict__ alpha_dev) {
  if (alpha_dev != nullptr) alpha *= *alpha_dev;          // (a scale that only exists on the device: the x1g / x3g split scale's inverse)
  const size_t total = (size_t)c_out * c_in * taps;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int tap = (int)(i % taps);
    const int ci = (int)((i / taps) % c_in);
    const int co = (int)(i / ((size_t)taps * c_in));
    dw[i] = alpha * dwt[((size_t)co * taps + tap) * c_in_pad + ci];
  }
}

// [B][C][h][w][T] fp32 (frames LAST) -> padded NHWC [n_img][h+2][w+2][c_pad].  The source is contiguous along
// (x, t) for a fixed (b, c, y) and the destination along c for a fixed (image, y, x): one workgroup per
// (CC-channel chunk, row y, sample b) reads CC runs of w*T floats (coalesced), parks them in LDS [CC][w*T+1] and
// writes every (frame, x) as CC contiguous channels (8 per thread).  The earlier one-thread-per-element gather
// (stride h*w*T floats between neighbouring lanes) ran at 0.33 TB/s.
template <typename T, int CC>
__global__ void __launch_bounds__(256) feat_to_nhwc_kernel(const float* __restrict__ v, const int* __restrict__ img_of,
                                                           T* __restrict__ y, int B, int C, int h, int w, int Tn, int c_pad) {
  extern __shared__ __attribute__((aligned(16))) float stage[];     // [CC][w*Tn + 1]
  const int c0 = blockIdx.x * CC, py = blockIdx.y, b = blockIdx.z;
  const int WT = w * Tn, LD = WT + 1;
  for (int i = threadIdx.x; i < CC * WT; i += 256) {
    const int cc = i / WT, j = i - cc * WT;
    const int c = c0 + cc;
    stage[cc * LD + j] = c < C ? v[(((size_t)b * C + c) * h + py) * (size_t)WT + j] : 0.f;
  }
  __syncthreads();
  constexpr int G = CC / 8;                     // 8-channel groups per pixel
  for (int i = threadIdx.x; i < WT * G; i += 256) {
    const int g = i % G, j = i / G;             // j = t * w + px  (frame-major so that one image's row is contiguous)
    const int t = j / w, px = j - t * w;
    const int img = img_of[b * Tn + t];
    if (img < 0) continue;
    T out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = ElemOps<T>::store(stage[(g * 8 + e) * LD + px * Tn + t]);
    T* dst = y + ((((size_t)img * (h + 2)) + py + 1) * (w + 2) + px + 1) * c_pad + c0 + g * 8;
    if constexpr (sizeof(T) == 2) {
      *(uint4*)dst = *(const uint4*)out;
    } else {
      *(float4*)dst = *(const float4*)out;
      *(float4*)(dst + 4) = *(const float4*)(out + 4);
    }
  }
}

// padded NHWC [n][h+2*halo][w+2*halo][c_pad] -> dense NCHW fp32 [n][C][h][w]
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ out, int n_img, int C, int h, int w,
                                    int c_pad, int halo) {
  const size_t total = (size_t)n_img * C * h * w;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int px = (int)(i % w);
    const int py = (int)((i / w) % h);
    const int c = (int)((i / ((size_t)w * h)) % C);
    const int n = (int)(i / ((size_t)w * h * C));
    out[i] = ElemOps<T>::load(x[((((size_t)n * (h + 2 * halo)) + py + halo) * (w + 2 * halo) + px + halo) * c_pad + c]);
  }
}

// dense NCHW fp32 [n][C][h][w] -> padded NHWC (halo 1), channels zero-padded to c_pad
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int n_img, int C, int h, int w,
                                    int c_pad) {
  const size_t total = (size_t)n_img * h * w * c_pad;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % c_pad);
    const int px = (int)((i / c_pad) % w);
    const int py = (int)((i / ((size_t)c_pad * w)) % h);
    const int n = (int)(i / ((size_t)c_pad * w * h));
    const float v = c < C ? x[(((size_t)n * C + c) * h + py) * w + px] : 0.f;
    y[((((size_t)n * (h + 2)) + py + 1) * (w + 2) + px + 1) * c_pad + c] = ElemOps<T>::store(v);
  }
}

// ---- fc_embed_attn weight: nn.Linear over an NCHW-flattened map <-> the column order of a flattened padded-NHWC image --
// w fp32 [rows][C][h][w]  ->  nat [rows_pad][(h+2)(w+2)][c_pad]   (kernel A: block = (row, 64-channel chunk))
//                         ->  nat_t [(h+2)(w+2)][c_pad][rows_pad]  (kernel B: block = one channel, all rows)
// Both read the source in contiguous h*w-float runs through LDS and write 128..256-byte segments; halo positions,
// padded rows and padded channels are written as zeros, so the destinations need no memset.
template <typename T>
__global__ void __launch_bounds__(256) fc_pack_nat_kernel(const float* __restrict__ w, T* __restrict__ nat, int rows, int C,
                                                          int h, int wd, int c_pad) {
  extern __shared__ float tile[];                      // [64][S + 1]
  const int row = blockIdx.y, c0 = blockIdx.x * 64;
  const int S = h * wd, LD = S + 1, Sp = (h + 2) * (wd + 2);
  const bool live = row < rows;
  if ((S & 3) == 0) {             // runs of S floats start 16-byte aligned: float4 loads
    const int S4 = S >> 2;
    for (int i = threadIdx.x; i < 64 * S4; i += 256) {
      const int cc = i / S4, q = i - cc * S4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && c0 + cc < C) v = *(const float4*)(w + ((size_t)row * C + c0 + cc) * S + 4 * q);
      float* t = tile + cc * LD + 4 * q;
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
  } else {
    for (int i = threadIdx.x; i < 64 * S; i += 256) {
      const int cc = i / S, sidx = i - cc * S;
      tile[cc * LD + sidx] = (live && c0 + cc < C) ? w[((size_t)row * C + c0 + cc) * S + sidx] : 0.f;
    }
  }
  __syncthreads();
  T* dst = nat + (size_t)row * Sp * c_pad + c0;
  for (int i = threadIdx.x; i < Sp * 8; i += 256) {
    const int p = i >> 3, g = i & 7;
    const int py = p / (wd + 2), px = p - py * (wd + 2);
    const bool inside = py >= 1 && py <= h && px >= 1 && px <= wd;
    const int sidx = (py - 1) * wd + px - 1;
    T out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = ElemOps<T>::store(inside ? tile[(g * 8 + e) * LD + sidx] : 0.f);
    T* d = dst + (size_t)p * c_pad + g * 8;
    if constexpr (sizeof(T) == 2) {
      *(uint4*)d = *(const uint4*)out;
    } else {
      *(float4*)d = *(const float4*)out;
      *(float4*)(d + 4) = *(const float4*)(out + 4);
    }
  }
}

template <typename T>
__global__ void __launch_bounds__(256) fc_pack_nat_t_kernel(const float* __restrict__ w, T* __restrict__ nat_t, int rows, int C,
                                                            int h, int wd, int c_pad, int rows_pad) {
  extern __shared__ float tile[];                      // [rows_pad][S + 1]
  const int c = blockIdx.x;
  const int S = h * wd, LD = S + 1, Sp = (h + 2) * (wd + 2);
  if ((S & 3) == 0) {
    const int S4 = S >> 2;
    for (int i = threadIdx.x; i < rows_pad * S4; i += 256) {
      const int r = i / S4, q = i - r * S4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < rows && c < C) v = *(const float4*)(w + ((size_t)r * C + c) * S + 4 * q);
      float* t = tile + r * LD + 4 * q;
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
  } else {
    for (int i = threadIdx.x; i < rows_pad * S; i += 256) {
      const int r = i / S, sidx = i - r * S;
      tile[r * LD + sidx] = (r < rows && c < C) ? w[((size_t)r * C + c) * S + sidx] : 0.f;
    }
  }
  __syncthreads();
  const int groups = rows_pad / 8;
  for (int i = threadIdx.x; i < Sp * groups; i += 256) {
    const int p = i / groups, g = i - p * groups;
    const int py = p / (wd + 2), px = p - py * (wd + 2);
    const bool inside = py >= 1 && py <= h && px >= 1 && px <= wd;
    const int sidx = (py - 1) * wd + px - 1;
    T out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = ElemOps<T>::store(inside ? tile[(g * 8 + e) * LD + sidx] : 0.f);
    T* d = nat_t + ((size_t)p * c_pad + c) * rows_pad + g * 8;
    if constexpr (sizeof(T) == 2) {
      *(uint4*)d = *(const uint4*)out;
    } else {
      *(float4*)d = *(const float4*)out;
      *(float4*)(d + 4) = *(const float4*)(out + 4);
    }
  }
}

// dw_nat fp32 [rows_pad][(h+2)(w+2)][c_pad] -> dw fp32 [rows][C][h][w]   (block = (row, 64-channel chunk))
__global__ void __launch_bounds__(256) fc_unpack_grad_kernel(const float* __restrict__ dnat, float* __restrict__ dw, int rows,
                                                             int C, int h, int wd, int c_pad, float alpha,
                                                             const float* __restrict__ alpha_dev) {
  extern __shared__ float tile[];                      // [S][64 + 1]
  if (alpha_dev != nullptr) alpha *= *alpha_dev;
  const int row = blockIdx.y, c0 = blockIdx.x * 64;
  const int S = h * wd, Sp = (h + 2) * (wd + 2);
  const float* src = dnat + (size_t)row * Sp * c_pad + c0;
  for (int i = threadIdx.x; i < S * 64; i += 256) {
    const int sidx = i >> 6, cc = i & 63;
    const int py = sidx / wd + 1, px = sidx - (py - 1) * wd + 1;
    tile[sidx * 65 + cc] = src[(size_t)(py * (wd + 2) + px) * c_pad + cc];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * S; i += 256) {
    const int cc = i / S, sidx = i - cc * S;
    if (c0 + cc < C) dw[((size_t)row * C + c0 + cc) * S + sidx] = alpha * tile[sidx * 65 + cc];
  }
}

// zero the 1-pixel halo ring of a padded NHWC tensor [n][hp][wp][c] (16-byte pieces): what a fresh conv output needs
// instead of a memset of the whole tensor (the kernels write every interior pixel, never the halo)
__global__ void __launch_bounds__(256) zero_halo_kernel(uint4* __restrict__ y, int hp, int wp, int c16) {
  const int ring = 2 * wp + 2 * (hp - 2);               // halo pixels per image
  uint4* img = y + (size_t)blockIdx.x * hp * wp * c16;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int i = threadIdx.x; i < ring * c16; i += 256) {
    const int r = i / c16, k = i - r * c16;
    int py, px;
    if (r < wp) { py = 0; px = r; }
    else if (r < 2 * wp) { py = hp - 1; px = r - wp; }
    else { const int q = r - 2 * wp; py = 1 + (q >> 1); px = (q & 1) ? wp - 1 : 0; }
    img[((size_t)py * wp + px) * c16 + k] = z;
  }
}

// ---- border correction of the composed conv pair (stem.py: FrozenStem._compose_pair) --------------------------------
// ring positions q of the (H+2)x(W+2) grid in the order top row (W+2), bottom row (W+2), left column (H), right column (H)
__device__ __forceinline__ void ring_pos(int r, int H, int W, int& qy, int& qx) {
  if (r < W + 2) { qy = -1; qx = r - 1; }
  else if (r < 2 * (W + 2)) { qy = H; qx = r - (W + 2) - 1; }
  else if (r < 2 * (W + 2) + H) { qy = r - 2 * (W + 2); qx = -1; }
  else { qy = r - 2 * (W + 2) - H; qx = W; }
}

// out[n][r][tap][c] = x[n][qy + dy + 2][qx + dx + 2][c]   (x: halo-2 padded NHWC, 16-byte pieces)
__global__ void __launch_bounds__(256) ring_im2col_kernel(const uint4* __restrict__ x, uint4* __restrict__ out, int H, int W,
                                                          int c16) {
  const int R = 2 * (W + 2) + 2 * H;
  const int r = blockIdx.x, n = blockIdx.y;
  int qy, qx;
  ring_pos(r, H, W, qy, qx);
  const int Wp = W + 4;
  const uint4* img = x + (size_t)n * (H + 4) * Wp * c16;
  uint4* dst = out + ((size_t)n * R + r) * 9 * c16;
  for (int i = threadIdx.x; i < 9 * c16; i += 256) {
    const int tap = i / c16, k = i - tap * c16;
    const int dy = tap / 3, dx = tap - 3 * dy;                      // 0..2 == -1..1 shifted by the +1 below
    dst[i] = img[((size_t)(qy + dy + 1) * Wp + (qx + dx + 1)) * c16 + k];
  }
}

// edge operands: for border pixel j of `edge` (0 top, 1 bottom: j = x; 2 left, 3 right: j = y) the three outside
// neighbours' y1 rows, zero where the neighbour is a corner owned by the top / bottom group or lies inside the image.
//   y1 [n][R][cm] -> out [n][len][3][cm], len = W (top/bottom) or H (left/right)
// group_rows > 0: all four edges in one launch (edge = blockIdx.z), edge e written to rows [e * group_rows, ...) of out
// — the row-padded operand of vnqa_gemm_nt_grouped.
__global__ void __launch_bounds__(256) ring_edge_gather_kernel(const uint4* __restrict__ y1, uint4* __restrict__ out, int H,
                                                               int W, int c16, int edge, int group_rows) {
  const int R = 2 * (W + 2) + 2 * H;
  const int j = blockIdx.x, n = blockIdx.y;
  if (group_rows > 0) edge = blockIdx.z;
  const int len = edge < 2 ? W : H;
  if (j >= len) return;
  uint4* dst = out + ((size_t)(group_rows > 0 ? edge * group_rows : 0) + (size_t)n * len + j) * 3 * c16;
  for (int i = threadIdx.x; i < 3 * c16; i += 256) {
    const int slot = i / c16, k = i - slot * c16;
    int r = -1;
    if (edge == 0) r = j + slot;                                    // q = (-1, j + slot - 1)
    else if (edge == 1) r = (W + 2) + j + slot;                     // q = (H, j + slot - 1)
    else {
      const int yy = j + slot - 1;                                  // q = (yy, -1 | W); corners (yy = -1, H) excluded
      if (yy >= 0 && yy < H) r = 2 * (W + 2) + (edge == 3 ? H : 0) + yy;
    }
    uint4 v = make_uint4(0u, 0u, 0u, 0u);       // (a ternary between a load and the constant kept `z` in scratch)
    if (r >= 0) v = y1[((size_t)n * R + r) * c16 + k];
    dst[i] = v;
  }
}

// ring[n][2W + 2(H-2)][c] = top | bottom | left[1:-1] | right[1:-1], corners += left/right ends   (float accumulate)
// One thread per 16-byte chunk of a ring row (8 bf16 / 4 f32), 256 threads cover 256 / (C * sizeof(T) / 16) rows.
// corner != NULL (vnqa_ring_assemble_corners): [n][4][C], SUBTRACTED from the four corner pixels (top-left, top-right, bottom-left,
// bottom-right): the term both of a corner's edge convs counted (vnqa_conv2d_border_edge_fwd)
template <typename T>
__global__ void __launch_bounds__(256) ring_assemble_kernel(const T* __restrict__ top, const T* __restrict__ bottom,
                                                            const T* __restrict__ left, const T* __restrict__ right,
                                                            T* __restrict__ ring, int n_img, int H, int W, int C,
                                                            const T* __restrict__ corner) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const int cpr = C / EPC;                                 // chunks per row
  const int RL = 2 * W + 2 * (H - 2);
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = idx / cpr;
  if (row >= (long long)n_img * RL) return;
  const int c = (int)(idx - row * cpr) * EPC;
  const int n = (int)(row / RL), r = (int)(row - (long long)n * RL);
  float v[EPC];
  auto add = [&](const T* src, int j, int len) {
    const uint4 raw = *(const uint4*)(src + ((size_t)n * len + j) * C + c);
    const T* e = (const T*)&raw;
#pragma unroll
    for (int k = 0; k < EPC; ++k) v[k] += ElemOps<T>::load(e[k]);
  };
#pragma unroll
  for (int k = 0; k < EPC; ++k) v[k] = 0.f;
  auto sub_corner = [&](int which) {
    if (corner == nullptr) return;
    const uint4 raw = *(const uint4*)(corner + ((size_t)n * 4 + which) * C + c);
    const T* e = (const T*)&raw;
#pragma unroll
    for (int k = 0; k < EPC; ++k) v[k] -= ElemOps<T>::load(e[k]);
  };
  if (r < W) {
    add(top, r, W);
    if (r == 0) { add(left, 0, H); sub_corner(0); }
    if (r == W - 1) { add(right, 0, H); sub_corner(1); }
  } else if (r < 2 * W) {
    const int x = r - W;
    add(bottom, x, W);
    if (x == 0) { add(left, H - 1, H); sub_corner(2); }
    if (x == W - 1) { add(right, H - 1, H); sub_corner(3); }
  } else if (r < 2 * W + H - 2) {
    add(left, r - 2 * W + 1, H);
  } else {
    add(right, r - 2 * W - (H - 2) + 1, H);
  }
  uint4 outv;
  T* o = (T*)&outv;
#pragma unroll
  for (int k = 0; k < EPC; ++k) o[k] = ElemOps<T>::store(v[k]);
  *(uint4*)(ring + (size_t)row * C + c) = outv;
}

inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int vnqa_pack_conv_weight(const float* w_oihw, int32_t c_out, int32_t c_in, int32_t taps,
                                     int32_t c_out_pad, int32_t c_in_pad, const float* out_scale,
                                     int32_t transpose_flip, int32_t dtype, void* wt, void* stream) {
  VNQA_CHECK_ARG(w_oihw && wt, "pack_conv_weight: null pointer");
  VNQA_CHECK_ARG(taps == 9 || taps == 1 || taps == 25 || taps == 27, "pack_conv_weight: taps must be 1, 9, 25 or 27");
  VNQA_CHECK_ARG(c_out_pad >= c_out && c_in_pad >= c_in, "pack_conv_weight: pads smaller than sizes");
  VNQA_CHECK_ARG(!(transpose_flip && out_scale), "pack_conv_weight: out_scale unsupported with transpose_flip");
  const int rows = transpose_flip ? c_in_pad : c_out_pad;
  const int kch = transpose_flip ? c_out_pad : c_in_pad;
  const size_t total = (size_t)rows * taps * kch;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(pack_conv_weight_kernel<vnqa_bf16>, dim3(grid_for(total, 256)), dim3(256), 0, st, w_oihw, c_out,
                       c_in, taps, rows, kch, out_scale, transpose_flip, (vnqa_bf16*)wt);
  else if (dtype == VNQA_F32)
    hipLaunchKernelGGL(pack_conv_weight_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, w_oihw, c_out, c_in,
                       taps, rows, kch, out_scale, transpose_flip, (float*)wt);
  else
    VNQA_CHECK_ARG(false, "pack_conv_weight: bad dtype %d", dtype);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_unpack_conv_wgrad_dev(const float* dwt, int32_t c_out, int32_t c_in, int32_t taps, int32_t c_out_pad,
                                          int32_t c_in_pad, float* dw_oihw, float alpha, const float* alpha_dev, void* stream) {
  VNQA_CHECK_ARG(dwt && dw_oihw, "unpack_conv_wgrad: null pointer");
  VNQA_CHECK_ARG(c_out > 0 && c_in > 0 && taps > 0 && c_out_pad >= c_out && c_in_pad >= c_in, "unpack_conv_wgrad: bad geometry");
  const size_t total = (size_t)c_out * c_in * taps;
  hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dwt, c_out,
                     c_in, taps, c_out_pad, c_in_pad, dw_oihw, alpha, alpha_dev);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_unpack_conv_wgrad_scaled(const float* dwt, int32_t c_out, int32_t c_in, int32_t taps,
                                             int32_t c_out_pad, int32_t c_in_pad, float* dw_oihw, float alpha, void* stream) {
  return vnqa_unpack_conv_wgrad_dev(dwt, c_out, c_in, taps, c_out_pad, c_in_pad, dw_oihw, alpha, nullptr, stream);
}

extern "C" int vnqa_unpack_conv_wgrad(const float* dwt, int32_t c_out, int32_t c_in, int32_t taps,
                                      int32_t c_out_pad, int32_t c_in_pad, float* dw_oihw, void* stream) {
  return vnqa_unpack_conv_wgrad_scaled(dwt, c_out, c_in, taps, c_out_pad, c_in_pad, dw_oihw, 1.f, stream);
}

namespace {
template <typename T, int CC>
int feat_launch(const float* v, const int32_t* img_of, void* y, int b, int c, int h, int w, int t, int c_pad, hipStream_t st) {
  const size_t lds = (size_t)CC * (w * t + 1) * sizeof(float);
  auto kern = feat_to_nhwc_kernel<T, CC>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    vnqa_set_error("feat_to_nhwc: cannot reserve %zu B of LDS", lds);
    return VNQA_ERR_HIP;
  }
  hipLaunchKernelGGL(kern, dim3(c_pad / CC, h, b), dim3(256), lds, st, v, img_of, (T*)y, b, c, h, w, t, c_pad);
  return VNQA_OK;
}
}  // namespace

extern "C" int vnqa_feat_to_nhwc(const float* v, const int32_t* img_of, void* y, int32_t b, int32_t c,
                                 int32_t h, int32_t w, int32_t t, int32_t c_pad, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(v && img_of && y, "feat_to_nhwc: null pointer");
  VNQA_CHECK_ARG(c_pad >= c && c_pad % 8 == 0, "feat_to_nhwc: c_pad must be >= c and a multiple of 8");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "feat_to_nhwc: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  // widest channel chunk whose [CC][w*t+1] fp32 stage fits 128 KiB of LDS and divides c_pad
  const size_t row = (size_t)(w * t + 1) * sizeof(float);
  int rc;
  if (c_pad % 32 == 0 && 32 * row <= 128 * 1024)
    rc = dtype == VNQA_BF16 ? feat_launch<vnqa_bf16, 32>(v, img_of, y, b, c, h, w, t, c_pad, st)
                            : feat_launch<float, 32>(v, img_of, y, b, c, h, w, t, c_pad, st);
  else {
    VNQA_CHECK_ARG(8 * row <= 128 * 1024, "feat_to_nhwc: a row of %d x %d frames does not fit the LDS stage", w, t);
    rc = dtype == VNQA_BF16 ? feat_launch<vnqa_bf16, 8>(v, img_of, y, b, c, h, w, t, c_pad, st)
                            : feat_launch<float, 8>(v, img_of, y, b, c, h, w, t, c_pad, st);
  }
  if (rc != VNQA_OK) return rc;
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_nhwc_to_nchw(const void* x, float* out, int32_t n_img, int32_t c, int32_t h, int32_t w,
                                 int32_t c_pad, int32_t halo, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(x && out, "nhwc_to_nchw: null pointer");
  const size_t total = (size_t)n_img * c * h * w;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<vnqa_bf16>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const vnqa_bf16*)x, out,
                       n_img, c, h, w, c_pad, halo);
  else if (dtype == VNQA_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const float*)x, out, n_img,
                       c, h, w, c_pad, halo);
  else
    VNQA_CHECK_ARG(false, "nhwc_to_nchw: bad dtype %d", dtype);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_nchw_to_nhwc(const float* x, void* y, int32_t n_img, int32_t c, int32_t h, int32_t w,
                                 int32_t c_pad, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(x && y, "nchw_to_nhwc: null pointer");
  const size_t total = (size_t)n_img * h * w * c_pad;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<vnqa_bf16>, dim3(grid_for(total, 256)), dim3(256), 0, st, x, (vnqa_bf16*)y, n_img, c,
                       h, w, c_pad);
  else if (dtype == VNQA_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, x, (float*)y, n_img, c, h, w,
                       c_pad);
  else
    VNQA_CHECK_ARG(false, "nchw_to_nhwc: bad dtype %d", dtype);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_pack_fc_weight(const float* w, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t rows_pad,
                                   int32_t c_pad, int32_t dtype, void* nat, void* nat_t, void* stream) {
  VNQA_CHECK_ARG(w && nat, "pack_fc_weight: null pointer");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "pack_fc_weight: bad dtype %d", dtype);
  VNQA_CHECK_ARG(rows > 0 && c > 0 && h > 0 && wd > 0 && rows_pad >= rows && rows_pad % 8 == 0 && c_pad >= c && c_pad % 64 == 0,
                 "pack_fc_weight: rows_pad must be a multiple of 8 >= rows, c_pad a multiple of 64 >= c");
  const size_t ldsA = (size_t)64 * (h * wd + 1) * sizeof(float), ldsB = (size_t)rows_pad * (h * wd + 1) * sizeof(float);
  VNQA_CHECK_ARG(ldsA <= 160 * 1024 && (nat_t == nullptr || ldsB <= 160 * 1024),
                 "pack_fc_weight: a %dx%d map with %d rows does not fit the LDS tile", h, wd, rows_pad);
  hipStream_t st = (hipStream_t)stream;
  dim3 gA(c_pad / 64, rows_pad);
#define VNQA_FC_LAUNCH(T)                                                                                           \
  do {                                                                                                             \
    auto ka = fc_pack_nat_kernel<T>;                                                                               \
    if (ldsA > 64 * 1024) (void)hipFuncSetAttribute((const void*)ka, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsA); \
    hipLaunchKernelGGL(ka, gA, dim3(256), ldsA, st, w, (T*)nat, rows, c, h, wd, c_pad);                           \
    if (nat_t != nullptr) {                                                                                        \
      auto kb = fc_pack_nat_t_kernel<T>;                                                                           \
      if (ldsB > 64 * 1024) (void)hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB); \
      hipLaunchKernelGGL(kb, dim3(c_pad), dim3(256), ldsB, st, w, (T*)nat_t, rows, c, h, wd, c_pad, rows_pad);    \
    }                                                                                                              \
  } while (0)
  if (dtype == VNQA_BF16) VNQA_FC_LAUNCH(vnqa_bf16);
  else VNQA_FC_LAUNCH(float);
#undef VNQA_FC_LAUNCH
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_unpack_fc_wgrad_scaled(const float* dw_nat, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t c_pad,
                                           float* dw, float alpha, void* stream);
extern "C" int vnqa_unpack_fc_wgrad(const float* dw_nat, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t c_pad,
                                    float* dw, void* stream) {
  return vnqa_unpack_fc_wgrad_scaled(dw_nat, rows, c, h, wd, c_pad, dw, 1.f, stream);
}

extern "C" int vnqa_unpack_fc_wgrad_dev(const float* dw_nat, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t c_pad,
                                        float* dw, float alpha, const float* alpha_dev, void* stream);
extern "C" int vnqa_unpack_fc_wgrad_scaled(const float* dw_nat, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t c_pad,
                                           float* dw, float alpha, void* stream) {
  return vnqa_unpack_fc_wgrad_dev(dw_nat, rows, c, h, wd, c_pad, dw, alpha, nullptr, stream);
}

extern "C" int vnqa_unpack_fc_wgrad_dev(const float* dw_nat, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t c_pad,
                                        float* dw, float alpha, const float* alpha_dev, void* stream) {
  VNQA_CHECK_ARG(dw_nat && dw, "unpack_fc_wgrad: null pointer");
  VNQA_CHECK_ARG(rows > 0 && c > 0 && h > 0 && wd > 0 && c_pad >= c && c_pad % 64 == 0, "unpack_fc_wgrad: bad geometry");
  const size_t lds = (size_t)h * wd * 65 * sizeof(float);
  VNQA_CHECK_ARG(lds <= 160 * 1024, "unpack_fc_wgrad: a %dx%d map does not fit the LDS tile", h, wd);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)fc_unpack_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(fc_unpack_grad_kernel, dim3((c + 63) / 64, rows), dim3(256), lds, (hipStream_t)stream, dw_nat, dw, rows, c, h,
                     wd, c_pad, alpha, alpha_dev);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

// ---- frame layout tables on the device (models/common.py: FrameLayout) ------------------------------------------------------------
// The packed image list of a minibatch: image n <-> (frame t, sorted sample s), frame-major, cts[t] = #samples with v_len > t
// (film_attn_pt_stem.py:201-208 iterates the frames of the length-sorted batch).  The tables used to be built on the host and copied
// through pinned memory on the stream the stem runs on; with the clips themselves arriving over PCIe (3 ms of DMA per minibatch)
// that 4-KB copy queued behind the clip of the minibatch after next and held the stem's first kernel for 0.1 - 0.9 ms.  Here the
// sorted lengths and the sort permutation travel as KERNEL ARGUMENTS and one workgroup writes the tables: no host-to-device copy.
struct LayoutArgs {
  int B, T, n_frames, n_img;
  int v[VNQA_LAYOUT_MAX_BATCH];       // lengths, sorted descending
  int perm[VNQA_LAYOUT_MAX_BATCH];    // sorted position s holds ORIGINAL sample perm[s]
};

__global__ void __launch_bounds__(256) frame_layout_kernel(const LayoutArgs a, int* __restrict__ img_of, int* __restrict__ frame_of,
                                                           int* __restrict__ sample_of, int* __restrict__ offsets) {
  __shared__ int off[1025];
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < a.n_frames; ++t) {
      off[t] = run;
      int ct = 0;
      for (int s = 0; s < a.B; ++s) ct += a.v[s] > t ? 1 : 0;
      run += ct;
    }
    off[a.n_frames] = run;
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= a.n_frames; i += blockDim.x) offsets[i] = off[i];
  for (int i = threadIdx.x; i < a.B * a.T; i += blockDim.x) {
    const int s = i / a.T, t = i - s * a.T;               // SORTED position s, frame t
    const bool live = t < a.n_frames && a.v[s] > t;       // (sorted descending: the live samples of frame t are s = 0 .. ct - 1)
    const int n = live ? off[t] + s : -1;
    img_of[a.perm[s] * a.T + t] = n;
    if (live) {
      frame_of[n] = t;
      sample_of[n] = s;
    }
  }
}

extern "C" int vnqa_frame_layout(const int32_t* v_sorted_host, const int32_t* perm_host, int32_t batch, int32_t frames, int32_t* img_of,
                                 int32_t* frame_of, int32_t* sample_of, int32_t* offsets, void* stream) {
  VNQA_CHECK_ARG(v_sorted_host && perm_host && img_of && frame_of && sample_of && offsets, "frame_layout: null pointer");
  VNQA_CHECK_ARG(batch > 0 && batch <= VNQA_LAYOUT_MAX_BATCH && frames > 0 && frames <= 1024,
                 "frame_layout: batch=%d (1..%d) frames=%d (1..1024)", batch, VNQA_LAYOUT_MAX_BATCH, frames);
  LayoutArgs a;
  a.B = batch; a.T = frames;
  int nf = 0, n_img = 0;
  for (int s = 0; s < batch; ++s) {
    VNQA_CHECK_ARG(v_sorted_host[s] >= 0 && v_sorted_host[s] <= frames && (s == 0 || v_sorted_host[s] <= v_sorted_host[s - 1]),
                   "frame_layout: lengths must be sorted descending and lie in 0..frames (v[%d]=%d)", s, v_sorted_host[s]);
    VNQA_CHECK_ARG(perm_host[s] >= 0 && perm_host[s] < batch, "frame_layout: perm[%d]=%d out of range", s, perm_host[s]);
    a.v[s] = v_sorted_host[s];
    a.perm[s] = perm_host[s];
    n_img += v_sorted_host[s];
  }
  nf = v_sorted_host[0];
  a.n_frames = nf; a.n_img = n_img;
  hipLaunchKernelGGL(frame_layout_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, img_of, frame_of, sample_of, offsets);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_zero_halo(void* y, int32_t n_img, int32_t hp, int32_t wp, int32_t c, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(y && n_img > 0 && hp >= 2 && wp >= 2, "zero_halo: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "zero_halo: bad dtype %d", dtype);
  const int row_bytes = c * (dtype == VNQA_BF16 ? 2 : 4);
  VNQA_CHECK_ARG(row_bytes % 16 == 0, "zero_halo: %d channels are not a whole number of 16-byte pieces", c);
  hipLaunchKernelGGL(zero_halo_kernel, dim3(n_img), dim3(256), 0, (hipStream_t)stream, (uint4*)y, hp, wp, row_bytes / 16);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_ring_im2col(const void* x, void* out, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t dtype,
                                void* stream) {
  VNQA_CHECK_ARG(x && out && n_img > 0 && h >= 2 && w >= 2, "ring_im2col: bad arguments");
  const int rb = c * (dtype == VNQA_BF16 ? 2 : 4);
  VNQA_CHECK_ARG((dtype == VNQA_BF16 || dtype == VNQA_F32) && rb % 16 == 0, "ring_im2col: bad dtype / channel count");
  hipLaunchKernelGGL(ring_im2col_kernel, dim3(2 * (w + 2) + 2 * h, n_img), dim3(256), 0, (hipStream_t)stream, (const uint4*)x,
                     (uint4*)out, h, w, rb / 16);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_ring_edge_gather(const void* y1, void* out, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t edge,
                                     int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(y1 && out && n_img > 0 && h >= 2 && w >= 2 && edge >= 0 && edge < 4, "ring_edge_gather: bad arguments");
  const int rb = c * (dtype == VNQA_BF16 ? 2 : 4);
  VNQA_CHECK_ARG((dtype == VNQA_BF16 || dtype == VNQA_F32) && rb % 16 == 0, "ring_edge_gather: bad dtype / channel count");
  hipLaunchKernelGGL(ring_edge_gather_kernel, dim3(edge < 2 ? w : h, n_img), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)y1, (uint4*)out, h, w, rb / 16, edge, 0);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_ring_edge_gather_all(const void* y1, void* out, int32_t n_img, int32_t h, int32_t w, int32_t c,
                                         int32_t group_rows, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(y1 && out && n_img > 0 && h >= 2 && w >= 2, "ring_edge_gather_all: bad arguments");
  VNQA_CHECK_ARG(group_rows >= n_img * (h > w ? h : w), "ring_edge_gather_all: group_rows=%d is smaller than an edge's %d rows",
                 group_rows, n_img * (h > w ? h : w));
  const int rb = c * (dtype == VNQA_BF16 ? 2 : 4);
  VNQA_CHECK_ARG((dtype == VNQA_BF16 || dtype == VNQA_F32) && rb % 16 == 0, "ring_edge_gather_all: bad dtype / channel count");
  hipLaunchKernelGGL(ring_edge_gather_kernel, dim3(h > w ? h : w, n_img, 4), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)y1, (uint4*)out, h, w, rb / 16, 0, group_rows);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_ring_assemble_corners(const void* top, const void* bottom, const void* left, const void* right, const void* corner,
                                          void* ring, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(top && bottom && left && right && ring && n_img > 0 && h >= 2 && w >= 2, "ring_assemble: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "ring_assemble: bad dtype %d", dtype);
  const int es = dtype == VNQA_BF16 ? 2 : 4;
  VNQA_CHECK_ARG(c > 0 && (c * es) % 16 == 0, "ring_assemble: c=%d must fill whole 16-byte chunks", c);
  const long long chunks = (long long)n_img * (2 * w + 2 * (h - 2)) * (c * es / 16);
  dim3 grid((unsigned)((chunks + 255) / 256));
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(ring_assemble_kernel<vnqa_bf16>, grid, dim3(256), 0, (hipStream_t)stream, (const vnqa_bf16*)top,
                       (const vnqa_bf16*)bottom, (const vnqa_bf16*)left, (const vnqa_bf16*)right, (vnqa_bf16*)ring, n_img, h,
                       w, c, (const vnqa_bf16*)corner);
  else
    hipLaunchKernelGGL(ring_assemble_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)top,
                       (const float*)bottom, (const float*)left, (const float*)right, (float*)ring, n_img, h, w, c, (const float*)corner);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int vnqa_ring_assemble(const void* top, const void* bottom, const void* left, const void* right, void* ring,
                                  int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t dtype, void* stream) {
  return vnqa_ring_assemble_corners(top, bottom, left, right, nullptr, ring, n_img, h, w, c, dtype, stream);
}
