// Internal: argument block shared by the implicit-GEMM conv kernels (conv_igemm.hip, conv_patch.hip).
#pragma once
#include "vnqa_common.h"

// activation of the plain epilogue (vnqa_conv_desc.relu): 0 none, 1 ReLU, 2 ELU(alpha = 1).  ELU lives in its OWN kernel
// instantiations (conv_igemm_kernel<..., TAG = VNQA_TAG_ELU>): as a runtime branch in the shared epilogue it cost the frozen
// stem's kernels 5.5 % end to end (same-box A/B 888 vs 938 clips/s: the inlined expm1f at every store site changed the
// register allocation and code layout of kernels that never take it).  Internal experiment bits >= 256 in `relu` mean ReLU.
#define VNQA_ACT_ELU 2
#define VNQA_TAG_ELU 3
template <int TAG>
__device__ __forceinline__ float vnqa_conv_act(float v, int mode) {
  if constexpr (TAG == VNQA_TAG_ELU) return v > 0.f ? v : expm1f(v);
  else return mode ? fmaxf(v, 0.f) : v;
}

struct ConvArgs {
  const char* x;
  const char* wt;
  const float* bias;
  const float* post_scale;
  const float* post_shift;
  char* y;
  int n_img, H, W, Hp, Wp;  // Hp/Wp: padded input dims
  int Cin, Cout, Cy;
  int taps, x_halo, y_halo;
  int relu, pool;
  int M;                    // n_img*H*W conv-output pixels
  int tilesN;
  int Hyp, Wyp;             // padded OUTPUT dims (after pooling)
  int wt_tiled;             // weights are pre-tiled LDS images (vnqa_pack_conv_weight_tiled)
  int D;                    // > 0: 3-D conv over [n][D+2][H+2][W+2][C]; "images" are (n, d) depth slices
  int slices, kt_per_slice; // split-K: K-steps [slice*kt_per_slice, ...) -> fp32 slab
  float* partial;           // [slices][M][Cout] fp32 when slices > 1
  int group_tiles;          // > 0: grouped GEMM — pixel tile t uses weight rows [(t / group_tiles) * Cout, ...) (vnqa_gemm_nt_grouped)
  int x_wrap2 = 0;          // 1: x has Cin / 2 physical channels, read twice along K (VNQA_CONV_X_WRAP2 / VNQA_GEMM_X_WRAP2; TAG 4 kernels)
  int tap3_vertical = 0;    // taps == 3: the window runs down a COLUMN (tap stride Wp pixels) instead of along a row (vnqa_conv2d_ring_edge_fwd: left / right edges)
  int dual_out = 0;         // 1: y gets 2 Cout channels per pixel, [h16(v) | h16(v - h16(v))] (VNQA_CONV_DUAL_OUT; conv_ps TAG 2); 2: [hi | lo | hi] (VNQA_CONV_DUAL_HI2); 4: hi in y, lo in y2, two plain tensors (VNQA_EPI_SPLIT_OUT)
  int xcd_split = 0;        // 1: TAG 1 launches with two cout tiles give every XCD ONE cout half (VNQA_CONV_XCD_SPLIT_N)
  int zero_halo = 0;        // 1: the store loop also writes zeros to y's (and y2's) 1-pixel halo ring (vnqa_conv_desc.flags)
  int ring_h, ring_w;       // > 0: the "pixels" are the outside-ring positions of the (ring_h+2) x (ring_w+2) grid of halo-2 images
                            // (vnqa_conv2d_ring_fwd): pixel m = image m / R, ring position m % R, R = 2(ring_w+2) + 2 ring_h
  // fused epilogues of the FiLM trunk (vnqa_conv2d_igemm_fused_fwd; applied to the LDS-staged, storage-rounded tile in the
  // store loop, so the results are bit-identical to the separate elementwise kernels):
  int epi;                  // VNQA_EPI_NONE | VNQA_EPI_BNSTATS | VNQA_EPI_FILM_RES
  const int* frame_of;      // BNSTATS: [n_img] frame of every image (images of a frame are contiguous)
  float* stats_partial;     // BNSTATS: [tilesM][3][2][Cout] fp32: per pixel tile and frame slot (frame - frame of the tile's
                            //          first pixel; a 256-pixel tile overlaps at most 3 frames), sum and sum of squares
  const float* film_gamma;  // FILM_RES: fp32, gamma of image n / channel c at film_gamma[n * film_ld + c]
  const float* film_beta;
  int film_ld, film_c;      // row stride (floats); channels >= film_c (padding) get gamma = beta = 0
  const char* res;          // FILM_RES: residual, same padded-NHWC geometry / element type as y
  char* y2;                 // FILM_RES: second output relu(gamma*z+beta)+res (y receives z = conv + bias, needed by the backward)
  const void* border_sub;   // optional [n_img][2W + 2(H-2)][Cout] (element type of y): value SUBTRACTED from the border
                            // pixels' sums before ReLU/pool (ring order: top row, bottom row, left column, right column)
};

// conv_patch.hip: 224-pixel 2-D tiles, activation patch DMA'd once per 64-channel chunk (bf16, 3x3, 2-D only).
// Returns VNQA_ERR_UNSUPPORTED (with the reason in vnqa_last_error) when the geometry does not fit.
int vnqa_conv_patch_dispatch(const ConvArgs& a, int tag, hipStream_t st);

// conv_ps.hip: patch-stationary 3x3 conv, 4 waves / 512 registers each, hand-placed main loop (bf16, 3x3, 2-D, plain epilogue).
int vnqa_conv_ps_dispatch(const ConvArgs& a, int tag, hipStream_t st);
