/* vnqa_hip.h — C ABI of libvnqa_hip.so, the MI355X (gfx950) kernel library for the
 * VideoNavQA video-question fusion path.
 *
 * The reference (catalina17/VideoNavQA) has no native layer: every op below replaces an
 * implicit ATen/cuDNN dispatch made from the Python files cited per entry point.  The
 * boundary is plain C: raw device pointers + sizes, a caller-owned stream, no torch types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless named host_*;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no hidden syncs,
 *     no allocation: the caller owns every buffer (workspaces are explicit arguments);
 *   - return value: 0 on success, a negative VNQA_ERR_* otherwise; the message is available
 *     from vnqa_last_error() (thread-local); nothing aborts or throws across the ABI;
 *   - dtype: VNQA_BF16 (bf16 storage, fp32 MFMA accumulate) or VNQA_F32 (exact fp32 MFMA);
 *   - activation layout: "padded NHWC" = [n_img][h+2*halo][w+2*halo][c] with a ZERO halo that
 *     kernels never write; channel counts are padded to a multiple of 64 by the caller.
 */
#ifndef VNQA_HIP_H_
#define VNQA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VNQA_BF16 0
#define VNQA_F32 1

#define VNQA_OK 0
#define VNQA_ERR_INVALID_ARG (-1)
#define VNQA_ERR_HIP (-2)
#define VNQA_ERR_UNSUPPORTED (-3)

/* ABI version of this header: bumped whenever an entry point, a struct layout or an enum value changes.  vnqa_version() returns
 * the value the LIBRARY was built with; the Python binding (videonavqa_amd/_lib.py: ABI_VERSION) refuses a library that
 * reports a different one (a stale build supplied through VNQA_LIB / kept with VNQA_NO_REBUILD=1). */
#define VNQA_ABI_VERSION 432
int vnqa_version(void);
const char* vnqa_last_error(void);

/* CU partition for a software pipeline of full-chip kernels (the frozen stem) against a latency-bound chain of small kernels or a
 * collective on another stream.  vnqa_stream_create_reserved: a HIP stream (hipStream_t in *stream; the caller destroys it with
 * hipStreamDestroy) whose kernels never run on `reserve_cus` of the device's CUs (a multiple of n_cu / 8, spread evenly over the
 * XCDs; 0 = an ordinary stream).  The persistent one-workgroup-per-CU conv kernels (vnqa_conv2d_c64_fwd, vnqa_conv_first_c64_fwd,
 * vnqa_conv2d_wreg_fwd) size their grids for n_cu - n CUs when the call's descriptor carries VNQA_CONV_RESERVE_CUS(n) in `flags`:
 * a per-call argument — the library keeps NO mutable process-wide state and reads NO environment variable. */
int vnqa_stream_create_reserved(int32_t reserve_cus, void** stream);
int vnqa_stream_create_masked(const uint32_t* host_mask, int32_t words, void** stream);   /* explicit mask, bit i = CU i usable */

/* ---------------------------------------------------------------------------------------
 * conv2d, stride 1, 'same' (3x3 pad 1 or 1x1), implicit GEMM on MFMA.
 * Replaces nn.Conv2d forward at models/obj_detector.py:72,77,82 (+ the folded eval-mode
 * BatchNorm2d / ReLU / MaxPool2d of :70-86), the VGG-16 front convs reached through
 * eval/q_and_v_eval.py:106, and models/film_attn_pt_stem.py:211,219,224 (conv_init,
 * conv1x1, FiLM 3x3).  The same entry point computes dgrad (conv of dY with the
 * flipped/transposed weights from vnqa_pack_conv_weight(..., transpose_flip=1)).
 *
 *   y = post( pool2?( relu?( conv(x, wt) + bias ) ) ),  post(v) = v*post_scale + post_shift
 *
 * x  : padded NHWC [n_img][h+2*x_halo][w+2*x_halo][c_in]
 * wt : [c_out][taps][c_in]  (K-major; see vnqa_pack_conv_weight)
 * y  : padded NHWC [n_img][ho+2*y_halo][wo+2*y_halo][c_y], ho = h (or h/2 with pool2)
 * bias, post_scale, post_shift: fp32 [c_out] or NULL.
 */
typedef struct vnqa_conv_desc {
  int32_t dtype;   /* VNQA_BF16 | VNQA_F32 : element type of x, wt, y */
  int32_t n_img;
  int32_t h, w;    /* spatial size of the conv output (== input) before pooling */
  int32_t c_in;    /* multiple of 64 (bf16) / 32 (f32) */
  int32_t c_out;   /* multiple of 8; rows of wt */
  int32_t c_y;     /* channel stride of y, >= c_out */
  int32_t taps;    /* 9 or 1 */
  int32_t x_halo;  /* 1 for taps==9; 0 or 1 for taps==1 */
  int32_t y_halo;  /* 0 or 1 */
  int32_t relu;    /* activation after conv + bias: 0 none, 1 ReLU, 2 ELU(alpha 1) (MACNetwork's conv stack, models/mac.py:174-179;
                    * vnqa_conv2d_igemm_fwd[_ex] with tile = VNQA_TILE_AUTO or the plain tile it resolves to, no pooling) */
  int32_t pool2;   /* requires h, w even */
  int32_t tile;    /* 0 = auto, else a VNQA_TILE_* id */
  int32_t wt_tiled; /* 0: wt is [c_out][taps][c_in]; 1: wt comes from vnqa_pack_conv_weight_tiled for THIS tile id */
  int32_t depth;   /* 0: 2-D conv.  > 0: 3-D conv (nn.Conv3d k=3 pad=1, models/v_only_cnn3d.py:13-26) with taps == 27:
                    * x is [n_img][depth+2][h+2][w+2][c_in], y is [n_img][depth+2][ho+2][wo+2][c_y]; pool2 pools (1,2,2) */
  int32_t flags;   /* VNQA_CONV_ZERO_HALO: the kernel also writes ZEROS to the 1-pixel halo ring of y (channels < c_out), so y
                    * may be an uninitialised buffer — 2-D convs with y_halo == 1 on the igemm tiles (not the 224-pixel patch
                    * tiles 11 / 12); round 2: replaces a separate halo-zeroing launch per fresh conv output */
} vnqa_conv_desc;
#define VNQA_CONV_ZERO_HALO 1
#define VNQA_CONV_XCD_SPLIT_N 2   /* stem-tagged implicit-GEMM tiles with exactly two cout tiles (the composed 5x5 conv, c_out 512 on
                                   * 256-cout tiles): every XCD computes ONE cout half, so its 4 MiB L2 holds half of the weight set */
#define VNQA_CONV_X_WRAP2 4       /* x has c_in / 2 PHYSICAL channels and is read twice along K against wt = [w_hi | w_lo] ([c_out][taps][c_in]):
                                   * the two-product form x . w_hi + x . w_lo of a 16-bit activation with split weights (precision 'fp16w');
                                   * plain epilogues, tiles 256x256 / 256x128 / 256x64 / 512x128 / 320x128 or AUTO; c_in % 128 == 0 */
#define VNQA_CONV_DUAL_OUT 0x20000 /* y gets 2 c_out channels per pixel (c_y >= 2 c_out): [hi | lo] with hi = h16(v), lo = h16(v - hi) — the fp32
                                   * result v (bias, ReLU, 2x2 max-pool, affine all applied in fp32) as a PAIR of 16-bit values.  A consumer
                                   * that is a plain conv over 2 c_out input channels against [w | w] contracts the unrounded activation.
                                   * Patch-stationary tiles (VNQA_TILE_PS_224x256 / _STEM_PS_224x256: 3x3, no border_sub) and — round 6, for the
                                   * geometries those do not serve (the 10 x 13 maps of 160 x 208 frames) and for the composed 5x5 conv with its
                                   * border_sub — the 256x256 implicit-GEMM tiles (VNQA_TILE_256x256 / _STEM_256x256: any taps, K-major or
                                   * pre-tiled weights); both write the same bits */
#define VNQA_CONV_DUAL_HI2 0x40000 /* with VNQA_CONV_DUAL_OUT: three segments [hi | lo | hi] (c_y >= 3 c_out) — the operand of a consumer that
                                   * contracts the unrounded activation against SPLIT weights [w_hi | w_hi | w_lo] as a plain conv over
                                   * 3 c_out input channels (x_hi w_hi + x_lo w_hi + x_hi w_lo) */
#define VNQA_CONV_F32_EPILOGUE 0x80000 /* ONE plain 16-bit output whose bias / border correction / ReLU / 2x2 max-pool / affine are all applied in
                                   * fp32 and rounded ONCE (the plain epilogues of the LDS-staged tiles round to storage before the affine and again
                                   * after it): the dual epilogue's hi half alone.  Tiles and restrictions of VNQA_CONV_DUAL_OUT.  Round 6:
                                   * mean-shifted storage (post_shift = -mean) needs the single rounding to pay.  | VNQA_CONV_DUAL_HI2: the value
                                   * is written TWICE, y = [v | v] (c_y >= 2 c_out) — the operand of a consumer that is a plain conv over 2 c_out
                                   * channels against SPLIT weights [w_hi | w_lo]: x w_hi + x w_lo without reading x twice along K */
#define VNQA_CONV_RELU_FLOOR 0x200000 /* stem-tagged 256x256 implicit-GEMM tile with ReLU: post_shift[c] (post_scale == NULL) is the per-channel FLOOR
                                   * of the ReLU instead of 0 — relu(a) - mean = max(a - mean, -mean): with the caller's bias carrying -mean_c and
                                   * floor = -mean_c the tile stores a mean-shifted output with ONE rounding through its ordinary 16-bit epilogue
                                   * (the pooled maximum of rounded values is the rounded maximum), at no cost: the composed 5x5 conv */
#define VNQA_CONV_FIRST_MID_SHIFT 0x100000 /* vnqa_conv_first_c64_fwd[_sched]: b1 has 128 entries [bias (64) | mu (64)] — the first conv's
                                   * output is kept (in LDS) as relu(.) - mu[c], and as -mu[c] where it is the second conv's zero padding; the
                                   * caller adds sum_taps(W2 mu) to the second conv's bias.  Wide (default) kernel only */
/* bits 8..15 of flags: the persistent conv kernels leave n CUs (a multiple of 8, <= 224) to the other streams of the process */
#define VNQA_CONV_RESERVE_CUS(n) ((((n) < 0 ? 0 : ((n) > 224 ? 224 : (n))) / 8) << 8)
#define VNQA_CONV_RESERVE_OF(flags) ((((flags) >> 8) & 0xff) * 8)

#define VNQA_TILE_AUTO 0
#define VNQA_TILE_256x256 1
#define VNQA_TILE_256x128 2
#define VNQA_TILE_256x64 3
#define VNQA_TILE_128x128 4
#define VNQA_TILE_128x64 5
#define VNQA_TILE_P4_256x256 7   /* 4-stage ring, counted vmcnt (bf16; c_in % 32 == 0) */
#define VNQA_TILE_P4_256x128 8
#define VNQA_TILE_P4_256x64 9
#define VNQA_TILE_256x128_W24 10 /* 256x128 with 2x4 waves (wave tile 128x32), staggered like the 256x256 tile */
#define VNQA_TILE_PATCH_224x256 11 /* 2-D 224-pixel tiles (8x28 / 16x14), activation patch resident in LDS over the 9 taps
                                      (bf16 3x3 only; w % 14 == 0; returns VNQA_ERR_UNSUPPORTED otherwise) */
#define VNQA_TILE_STEM_PATCH_224x256 12 /* same kernel, own symbol for the frozen stem */
#define VNQA_TILE_256x256_W16 13 /* 256x256 with 16 waves (4 per SIMD, 64x64 wave tiles): measured variant, see DESIGN.md */
#define VNQA_TILE_256x128_W16 14 /* 256x128 with 16 waves (64x32 wave tiles) */
#define VNQA_TILE_512x128 15 /* 512 pixels x 128 couts, 4x2 waves (128x64 wave tiles as the 256x256 tile; 160 KiB of LDS) */
#define VNQA_TILE_P3_256x128 16 /* 256x128, 4x2 waves, ring of three 32-channel stages: 72 KiB of LDS, two workgroups per CU */
#define VNQA_TILE_STEM_256x256 6 /* 256x256 geometry, own kernel symbol for the frozen stem (bf16) */
#define VNQA_TILE_I5_256x256 18  /* 256x256, 8 waves, hand-pipelined main loop (PIPE 5: mid-K-step barrier, DMA two stages ahead, one filler per MFMA group) */
#define VNQA_TILE_STEM_I5_256x256 19 /* same kernel, own symbol for the frozen stem */
#define VNQA_TILE_PS_224x256 20  /* patch-stationary 3x3 conv (csrc/conv_ps.hip): 224 px x 256 couts, 4 waves x 512 registers, hand-placed loop;
                                    8 x 28 tiles for w % 28 == 0, else 16 x 14 for any even w >= 14 (last column block overlaps its neighbour) */
#define VNQA_TILE_STEM_PS_224x256 21 /* same kernel, own symbol for the frozen stem */
#define VNQA_TILE_320x128 17 /* 320 rows x 128 couts, 4x2 waves (80x64 wave tiles): skinny GEMMs whose 257..320 rows would waste half of
                                  a second 256-row tile (fc_embed_attn forward at 280 packed images) */

int vnqa_conv2d_igemm_fwd(const vnqa_conv_desc* d, const void* x, const void* wt,
                          const float* bias, const float* post_scale, const float* post_shift,
                          void* y, void* stream);

/* Same conv with two extensions used by the frozen stem's COMPOSED layer (ObjDetectCNN applies conv12 directly to
 * conv11's output, models/obj_detector.py:72 — no nonlinearity in between — so for frozen weights the pair equals ONE
 * 5x5 conv with composed weights, 25*128 instead of 9*128 + 9*512 multiply-adds per output):
 *   - taps == 25 (5x5, x_halo == 2; wt [c_out][25][c_in]); y_halo may be 0, 1 or 2;
 *   - border_sub (optional): [n_img][2*w + 2*(h-2)][c_out] in `dtype`, SUBTRACTED from the sums of the image-border
 *     pixels before ReLU / pooling (ring order: top row x = 0..w-1, bottom row, left column y = 1..h-2, right column).
 *     It carries the exact correction for conv12 seeing zero padding rather than conv11 evaluated outside the image.
 */
int vnqa_conv2d_igemm_fwd_ex(const vnqa_conv_desc* d, const void* x, const void* wt, const float* bias,
                             const float* post_scale, const float* post_shift, const void* border_sub,
                             void* y, void* stream);

/* Second-order rounding of FROZEN weights onto this build's 16-bit grid (stem.second_order_round; the GPTQ / OBQ sequential rounding):
 * w [rows][k] float64 (one row per output channel, columns in rounding order), u [k][k] float64 = the upper Cholesky factor of H^-1 of the
 * layer's (damped, equally permuted) input-patch second moment; per row, j = 0 .. k-1: q_j = h16(w_j), e = (w_j - q_j) / u_jj,
 * w_t -= e u_jt for t > j.  q [rows][k] fp32 receives the rounded values (exactly representable in the 16-bit format).  k <= 16384. */
int vnqa_second_order_round(const double* w, const double* u, float* q, int32_t rows, int32_t k, void* stream);

/* ---------------------------------------------------------------------------------------
 * Split operands (precision 'fp16h', csrc/split3.hip): v = hi + lo, hi = h16(v), lo = h16(v - hi) — an fp32 contraction on the 16-bit
 * matrix cores as two or three products of halves with fp32 accumulation, x . w = x_hi w_hi + x_lo w_hi + x_hi w_lo, at the same
 * nn.Conv2d / nn.Linear call sites as vnqa_conv2d_igemm_fwd / vnqa_gemm_nt (models/obj_detector.py:82, film_attn_pt_stem.py:211,219,244).
 *   vnqa_split3_f32 : rows x c fp32 (row stride src_ld) -> hi, lo (may be NULL) and (optional, may be NULL) a second copy of hi, each
 *                     rows x c in the library's 16-bit format with row stride dst_ld.  The weight operands: [w_hi | w_lo] =
 *                     (base, -, base + c) with dst_ld = 2c (VNQA_CONV_X_WRAP2 / VNQA_GEMM_X_WRAP2 consumers), [w_hi | w_hi | w_lo] =
 *                     (base, base + 2c, base + c) with dst_ld = 3c (consumers of a VNQA_CONV_DUAL_OUT | _HI2 activation).
 *                     scale: optional DEVICE scalar (a power of two) multiplied in before the split.
 *   vnqa_gemm_nt with dtype = VNQA_BF16 | VNQA_GEMM_OUT_F32 : 16-bit operands, fp32 `out` (workspace >= m*n*4 bytes required).
 */
int vnqa_split3_f32(const float* x, void* hi, void* lo, void* hi2, int64_t rows, int32_t c, int64_t src_ld, int64_t dst_ld,
                    const float* scale, void* stream);
#define VNQA_GEMM_OUT_F32 0x200
#define VNQA_GEMM_X_WRAP2 0x400   /* vnqa_gemm_nt: a has k / 2 physical columns (row stride k / 2), read twice against b = [b_hi | b_lo];
                                   * also accepted in the dtype of vnqa_conv2d_ring_fwd / vnqa_ring_edge_conv_fwd (c_in / c_mid = the contraction's
                                   * channel count, twice the tensor's) */

/* Fused trunk epilogues (BIAS_RELU_BNSTATS / BIAS_FILM_RELU_RES) — the same conv with the elementwise op
 * that FOLLOWS it in the reference applied while the output tile is still in LDS:
 *
 *   VNQA_EPI_BNSTATS  (conv_init -> ReLU -> train-mode BatchNorm2d per frame, models/film_attn_pt_stem.py:211)
 *     y = relu?(conv + bias) as usual, plus mean[f][c] / biased var[f][c] of y over the pixels of every frame f
 *     (frame_of [n_img], frame_off [n_frames+1]: images of a frame are contiguous).  Each pixel tile writes per-frame
 *     partial sums to `partial` (vnqa_conv2d_bnstats_workspace bytes) and a second tiny launch reduces them in tile
 *     order: no atomics, run-to-run deterministic.  Returns VNQA_ERR_UNSUPPORTED when frames are too small for the tile
 *     (workspace query < 0): the caller then uses vnqa_frame_bn_stats.
 *   VNQA_EPI_FILM_RES (conv3x3 -> FiLM affine -> ReLU -> + residual, models/film_attn_pt_stem.py:224-241)
 *     y  = conv + bias                                   (z: the FiLM backward needs it)
 *     y2 = relu(gamma[n][c] * y + beta[n][c]) + res      (gamma/beta fp32 rows of stride film_ld; channels >= film_c: 0)
 *     computed from the storage-rounded y, i.e. bit-identical to vnqa_film_relu_res_fwd applied to y.
 *     y == NULL (this epilogue only): forward-only form for inference (eval/q_and_v_eval.py:159-224, eval/q_and_v_test.py:64-142)
 *     — z is not stored, y2 is identical to the two-output call's.
 * pool2 / depth / wt_tiled / post_scale are not available with a fused epilogue.
 */
#define VNQA_EPI_NONE 0
#define VNQA_EPI_BNSTATS 1
#define VNQA_EPI_FILM_RES 2
#define VNQA_EPI_ADD_MASK 3     /* y = (conv + res) * [y2 > 0]  (y2 is READ: the tensor whose sign is the mask) — the FiLM block's
                                 * backward: dgrad of the 3x3 conv + the residual branch's gradient, masked by the 1x1 conv's
                                 * ReLU (models/film_attn_pt_stem.py:219-241 differentiated); bit-identical to vnqa_relu_bwd(a, b, y) */
#define VNQA_EPI_SPLIT_OUT 4    /* y = h16(v), y2 = h16(v - y): the conv's fp32 result kept as TWO plain 16-bit tensors of y's geometry (hi + lo =
                                 * v to 2^-22 relative) — conv_init of precision 'fp16h', whose BatchNorm then reads the unrounded value
                                 * (vnqa_frame_bn_stats_split / _apply_split); patch-stationary tile or VNQA_TILE_256x256, 16-bit formats;
                                 * bias / ReLU of the descriptor apply before the split */
typedef struct vnqa_conv_epilogue {
  int32_t kind;              /* VNQA_EPI_* */
  int32_t n_frames;          /* BNSTATS */
  int32_t min_frame_images;  /* BNSTATS: smallest number of images in a frame (host knowledge; sizes the slot count) */
  int32_t film_ld, film_c;   /* FILM_RES */
  const int32_t* frame_of;   /* BNSTATS: [n_img] */
  const int32_t* frame_off;  /* BNSTATS: [n_frames + 1] */
  float* partial;            /* BNSTATS workspace */
  float* mean;               /* BNSTATS out: [n_frames][c_out] */
  float* var;                /* BNSTATS out: [n_frames][c_out], biased */
  const float* gamma;        /* FILM_RES */
  const float* beta;         /* FILM_RES */
  const void* res;           /* FILM_RES: padded NHWC like y */
  void* y2;                  /* FILM_RES, SPLIT_OUT: padded NHWC like y */
} vnqa_conv_epilogue;
int64_t vnqa_conv2d_bnstats_workspace(const vnqa_conv_desc* d, int32_t min_frame_images);
int vnqa_conv2d_igemm_fused_fwd(const vnqa_conv_desc* d, const void* x, const void* wt, const float* bias,
                                const vnqa_conv_epilogue* e, void* y, void* stream);

/* Operand builders of that border correction (all tensors in `dtype`, c * element size a multiple of 16 bytes; the
 * outside ring of the (h+2)x(w+2) grid is enumerated top row (w+2), bottom row (w+2), left column (h), right column (h)):
 *   vnqa_ring_im2col      : x halo-2 padded NHWC [n][h+4][w+4][c] -> [n][2(w+2)+2h][9][c], the 3x3 patches around the ring
 *                           positions (GEMM with conv11's K-major weights gives conv11 evaluated OUTSIDE the image);
 *   vnqa_ring_edge_gather : y1 [n][ring][c] -> [n][w|h][3][c]: per border pixel of edge 0/1/2/3 (top/bottom/left/right)
 *                           its three outside neighbours (zeros where a corner belongs to the top/bottom group);
 *   vnqa_ring_assemble    : the four edge GEMM results [n][w|h][c] -> border_sub [n][2w+2(h-2)][c] (corners summed).
 */
/* vnqa_conv2d_ring_fwd: y1 [n][2(w+2)+2h][c_out] = conv3x3(x, wt) + bias evaluated AT the outside-ring positions, as an
 * implicit GEMM straight from the halo-2 image (replaces vnqa_ring_im2col + vnqa_gemm_nt: no [n*ring, 9*c_in] matrix). */
int vnqa_conv2d_ring_fwd(const void* x, const void* wt, const float* bias, void* y1, int32_t n_img, int32_t h, int32_t w,
                         int32_t c_in, int32_t c_out, int32_t padded, int32_t dtype, void* stream);
/* padded = 1: y1 is [n][R + 4][c_out] with the rows laid out top (w+2) | bottom (w+2) | 0 | left (h) | 0 | 0 | right (h) | 0; the
 * four zero rows are NEVER written (the caller zeroes the buffer once) and stand in for the corner neighbours that belong to
 * the top / bottom group when vnqa_ring_edge_conv_fwd slides its 1x3 window along the left / right columns:
 *   out[n][j][co] = sum_{slot < 3, c} y1p[n][base(edge) + j + slot][c] * wt[co][slot][c]      (edge 0/1/2/3 = top/bottom/left/right)
 * — the edge products of the border correction without vnqa_ring_edge_gather's [n*len, 3*c_mid] operands. */
int vnqa_ring_edge_conv_fwd(const void* y1p, const void* wt, void* out, int32_t n_img, int32_t h, int32_t w, int32_t c_mid,
                            int32_t c_out, int32_t edge, int32_t dtype, void* stream);
/* vnqa_conv2d_ring_fwd for ONE edge of the ring with the three taps that can see the image (a position one pixel outside has six of its nine taps in the zero
 * halo): a 1x3 (edge 0 / 1 = top / bottom) or 3x1 (edge 2 / 3 = left / right) conv over one image row / column, K = 3 c_in instead of 9 c_in,
 * the same sums in the same order.  wt [c_out][3][c_in]: kernel row 2 / row 0 / column 2 / column 0 of the 3x3 weights; y1p: the padded ring
 * layout [n][R + 4][c_out] of vnqa_conv2d_ring_fwd(padded = 1), whose segment of this edge is written (four calls fill it). */
int vnqa_conv2d_ring_edge_fwd(const void* x, const void* wt, const float* bias, void* y1p, int32_t n_img, int32_t h, int32_t w,
                              int32_t c_in, int32_t c_out, int32_t edge, int32_t dtype, void* stream);
int vnqa_ring_im2col(const void* x, void* out, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t dtype,
                     void* stream);
int vnqa_ring_edge_gather(const void* y1, void* out, int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t edge,
                          int32_t dtype, void* stream);
/* all four edges in one launch: edge e -> rows [e * group_rows, e * group_rows + n_img * (w | h)) of out [4 * group_rows][3 * c] */
int vnqa_ring_edge_gather_all(const void* y1, void* out, int32_t n_img, int32_t h, int32_t w, int32_t c,
                              int32_t group_rows, int32_t dtype, void* stream);
int vnqa_ring_assemble(const void* top, const void* bottom, const void* left, const void* right, void* ring,
                       int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t dtype, void* stream);
/* Round 6 — the border correction as FOUR composed edge convs (no ring tensor):
 *   vnqa_conv2d_border_edge_fwd : out[n][j][co] = bias[co] + sum_{t < 5, c} x[n][border row / column, j + t - 2][c] wt[co][t][c] for edge
 *                           0/1/2/3 = top/bottom/left/right; x: halo-2 images [n][h+4][w+4][c_in]; wt [c_out][5][c_in] = conv12's edge
 *                           taps composed with conv11's facing taps (stem._compose_pair); out dense [n][w | h][c_out].  K = 5 c_in
 *                           instead of 3 c_in + 3 c_mid: a quarter of the two-step form's FLOPs at 128 -> 512 -> 512.
 *   vnqa_ring_assemble_corners : vnqa_ring_assemble with corner [n][4][c] (or NULL) SUBTRACTED from the four corner pixels (top-left,
 *                           top-right, bottom-left, bottom-right): the outside-ring CORNER position is adjacent to one pixel only, and both
 *                           of that pixel's edge convs count it. */
int vnqa_conv2d_border_edge_fwd(const void* x, const void* wt, const float* bias, void* out, int32_t n_img, int32_t h, int32_t w,
                                int32_t c_in, int32_t c_out, int32_t edge, int32_t dtype, void* stream);
int vnqa_ring_assemble_corners(const void* top, const void* bottom, const void* left, const void* right, const void* corner, void* ring,
                               int32_t n_img, int32_t h, int32_t w, int32_t c, int32_t dtype, void* stream);

/* Persistent direct 3x3 conv for c_in == 64 (bf16): weights resident in LDS, 16x16 tiles with a DMA'd
 * 18x18 halo patch, no barrier inside the K loop.  Same contract as vnqa_conv2d_igemm_fwd restricted to
 * taps == 9, c_in == 64, c_out % 64 == 0, x_halo == y_halo == 1.  Used for VGG conv1_2 / conv2_1.
 */
int vnqa_conv2d_c64_fwd(const vnqa_conv_desc* d, const void* x, const void* wt, const float* bias,
                        const float* post_scale, const float* post_shift, void* y, void* stream);

/* Tail of the global-max-pooling models (models/film_global_pooling_pt_stem.py:228-238, models/time_multi_hop_pt_stem.py:240-250:
 * relu(c1x1_tail) per frame -> zero-padded stack over frames -> max over frames -> out_linear) on the PACKED image list:
 *   maps      [n_img][h+2][w+2][c_pad] relu'd tail maps (16-bit storage or f32, zero halo), images frame-major:
 *             image of (frame t, sorted sample b) = frame_off[t] + b for b < frame_off[t+1] - frame_off[t]
 *   pooled    fp32 [batch][tail*h*w] in the reference's NCHW-flattened order (= the column order of out_linear.weight)
 *   argmax    int32, same shape: the image that supplied the maximum, -1 where every frame is <= 0 (no gradient either way)
 * vnqa_frame_max_bwd writes the WHOLE gradient tensor d maps (zero halo / padding channels included):
 *   d maps[img][y][x][c] = scale * d pooled[sample_of[img]][c*h*w + y*w + x] where argmax == img, else 0.
 */
int vnqa_frame_max_fwd(const void* maps, const int32_t* frame_off, float* pooled, int32_t* argmax, int32_t batch,
                       int32_t n_frames, int32_t h, int32_t w, int32_t c_pad, int32_t tail, int32_t dtype, void* stream);
int vnqa_frame_max_bwd(const float* dpooled, const int32_t* argmax, const int32_t* sample_of, void* dmaps, int32_t n_img,
                       int32_t h, int32_t w, int32_t c_pad, int32_t tail, float scale, int32_t dtype, void* stream);

/* Multi-hop FiLM generator of TimeMultiHopFiLMPretrainedStem (models/time_multi_hop_pt_stem.py:124-184), fp32, packed image list.
 *   vnqa_layernorm_fwd : y[r] = LayerNorm(x[rows ? rows[r] : r]) * gamma + beta over the last dimension n (nn.LayerNorm, biased
 *                        variance; encoder_norm :148, decoder_norm :184); saves mean / rstd per row
 *   vnqa_layernorm_bwd : dx [n_rows][n] (dense, NOT scattered through `rows`; may be NULL) and d gamma / d beta (per-column sums
 *                        over the rows in row order; `accumulate` != 0 adds to them; may be NULL together)
 *   vnqa_scatter_add_rows : dst[rows[r]] += src[r] for unique rows (adjoint of the row gather)
 *   vnqa_hop_fwd       : decode_to_film_values' attention (:165-176) for every image: p = hv (.) states, coefs =
 *                        softmax_words(p . w + bias) over lmax words (padding words NOT masked, as upstream), hv_out = coefs^T p.
 *                        states of image i = rows base_row[i] .. base_row[i] + qlen[i] - 1 of hs [*, hidden] (the persistent
 *                        LSTM chain's output), zero beyond; coefs [n_img][lmax] is kept for the backward
 *   vnqa_hop_bwd       : d hv, d hs (ACCUMULATED into the caller's zero-initialised / running buffer), per-image d w rows
 *                        [n_img][hidden] (the caller sums them over images); d bias is identically 0 (softmax shift invariance)
 * hidden <= 256, lmax <= 64.
 */
int vnqa_layernorm_fwd(const float* x, const int32_t* rows, const float* gamma, const float* beta, float* y, float* mean,
                       float* rstd, int32_t n_rows, int32_t n, float eps, void* stream);
int vnqa_layernorm_bwd(const float* dy, const float* x, const int32_t* rows, const float* mean, const float* rstd,
                       const float* gamma, float* dx, float* dgamma, float* dbeta, int32_t n_rows, int32_t n,
                       int32_t accumulate, void* stream);
int vnqa_scatter_add_rows(float* dst, const int32_t* rows, const float* src, int32_t n_rows, int32_t n, void* stream);
int vnqa_hop_fwd(const float* hv, const float* hs, const int32_t* base_row, const int32_t* qlen, const float* w,
                 const float* bias, float* hv_out, float* coefs, int32_t n_img, int32_t lmax, int32_t hidden, void* stream);
int vnqa_hop_bwd(const float* dout, const float* hv, const float* hs, const int32_t* base_row, const int32_t* qlen,
                 const float* w, const float* coefs, float* dhv, float* dhs, float* dw_img, int32_t n_img, int32_t lmax,
                 int32_t hidden, void* stream);

/* Weights-stationary-in-REGISTERS persistent direct 3x3 conv (bf16 / the library's 16-bit format) for the short-K layers
 * of the VGG front (get_frcnn_feature_extractor features[2], [5], [7]; call sites eval/q_and_v_eval.py:106): 4 waves per
 * workgroup, one per SIMD, each keeping its slice of the weights in 288 of its SIMD's 512 registers for the whole
 * launch; only the activation patch goes through LDS (csrc/conv_wreg.hip).  Same tensors and epilogue contract as
 * vnqa_conv2d_igemm_fwd (bias -> ReLU -> 2x2 max-pool -> per-channel affine; y_halo 1 or 2) for exactly these geometries:
 *   c_in 128 -> c_out 128 with pool2 (conv2_2), c_in 64 -> c_out 128 without pool2 (conv2_1), c_in 64 -> c_out 64 with
 *   pool2 (conv1_2); w >= 16 and even (a width that is not a multiple of 16 costs one overlapping 16-pixel tile per
 *   row: the 80 x 104 maps of the reference's 160 x 208 frames), h % 8 == 0 (conv1_2: h % 16 == 0).
 * vnqa_conv2d_wreg_supported returns 1 when a descriptor qualifies (callers fall back to the igemm / c64 kernels).
 */
int vnqa_conv2d_wreg_supported(const vnqa_conv_desc* d);
/* 1 when the patch-stationary tiles (VNQA_TILE_PS_224x256 / VNQA_TILE_STEM_PS_224x256 of vnqa_conv2d_igemm_fwd[_ex] and the fused
 * FILM_RES / ADD_MASK calls) serve this descriptor's geometry: 16-bit 3x3 / 5x5 2-D conv, c_in % 64 == 0, even width >= 14, the
 * LDS patch and the 32-bit DMA offsets fit.  The library's own dispatch uses the same test. */
int vnqa_conv_ps_supported(const vnqa_conv_desc* d);
int vnqa_conv2d_wreg_fwd(const vnqa_conv_desc* d, const void* x, const void* wt, const float* bias,
                         const float* post_scale, const float* post_shift, void* y, void* stream);

/* conv1_1 + conv1_2 of the VGG front in ONE launch (get_frcnn_feature_extractor, features[0:4]; call sites
 * eval/q_and_v_eval.py:106): the 3 -> 64 conv + ReLU is evaluated inside the c_in == 64 direct kernel for the
 * 18x18 patch each 16x16 output tile needs, so its 64-channel output never goes to HBM.
 *   vnqa_clip_to_nhwc4 : clip fp32 [b][3][h][w][t] (frames LAST, eval/dataset.py:63,81-91) -> img4 bf16
 *                        [n_img][h+4][w+4][4] (image img_of[b*t+f] or -1 to skip; halo 2 and channel 3 are never
 *                        written: zero the buffer once)
 *   vnqa_conv_first_c64_fwd : d describes the SECOND conv (c_in = 64, h, w, c_out, relu, pool2 ...); w1 fp32
 *                        [64][3][3][3], b1 fp32 [64] are the first conv's parameters; wt/bias/post_* as in
 *                        vnqa_conv2d_c64_fwd; y padded NHWC bf16.
 */
int vnqa_clip_to_nhwc4(const float* clip, const int32_t* img_of, void* img4, int32_t b, int32_t t, int32_t h,
                       int32_t w, void* stream);
/* vnqa_clip_to_nhwc4_shifted (round 6): either source (lut == NULL: clip is fp32 as in vnqa_clip_to_nhwc4; else uint8 as in
 * vnqa_clip_u8_to_nhwc4) with MEAN-SHIFTED storage — the list holds pixel - shift[c] (shift: 3 floats on the device), so the 16-bit
 * rounding error scales with |pixel - shift|; the caller keeps -shift[c] in the list's halo (what a zero pixel becomes) and adds
 * sum_taps(W1 shift) to the first conv's bias.  Replaces the implicit .half() of the clip a 16-bit torch model would apply. */
int vnqa_clip_to_nhwc4_shifted(const void* clip, const float* lut, const float* shift, const int32_t* img_of, void* img4, int32_t b,
                               int32_t t, int32_t h, int32_t w, void* stream);
/* vnqa_clip_u8_to_nhwc4: the same image list from RAW 8-bit pixels [b][3][h][w][t] (what cv2 decodes, eval/dataset.py:66-77)
 * and the caller's table lut[256] = float32(k / 255.0) evaluated in double precision — bit for bit the value
 * `clip / 255.0` (dataset.py:91, float64) takes after `.float()` (eval/q_and_v_eval.py:92) — so the host uploads a quarter of
 * the reference's bytes per clip and the device sees identical inputs. */
int vnqa_clip_u8_to_nhwc4(const uint8_t* clip, const float* lut, const int32_t* img_of, void* img4, int32_t b, int32_t t,
                          int32_t h, int32_t w, void* stream);
int vnqa_conv_first_c64_fwd(const vnqa_conv_desc* d, const void* img4, const float* w1, const float* b1,
                            const void* wt, const float* bias, const float* post_scale,
                            const float* post_shift, void* y, void* stream);
/* ... with a DYNAMIC tile schedule: `sched` = 8 bytes on the device, zero on first use (the launch leaves them zero again), owned by one
 * stream at a time.  The persistent workgroups draw their tiles from a counter instead of owning a fixed stride of them, so a
 * workgroup whose CU was held by another stream's kernel for a while simply draws fewer: beside the trunk's forward pass (the
 * pipelined training step) the launch takes its share of the chip's time instead of waiting for its last-started workgroup's
 * full static share.  Results are identical (the tile -> pixels map does not depend on who computes a tile).  sched == NULL: the
 * static stride of vnqa_conv_first_c64_fwd. */
int vnqa_conv_first_c64_fwd_sched(const vnqa_conv_desc* d, const void* img4, const float* w1, const float* b1,
                                  const void* wt, const float* bias, const float* post_scale,
                                  const float* post_shift, void* y, void* sched, void* stream);

/* First VGG conv (3 -> c_out, 3x3 pad 1) + ReLU straight from the reference's clip layout.
 * Replaces the strided frame slice v_inputs[:, :, :, :, j] + conv1_1 of the external
 * feature extractor (eval/q_and_v_eval.py:104-106; eval/dataset.py:63,81-91 for the layout).
 * clip   : fp32 [b][3][h][w][t]  (frames on the LAST axis)
 * w      : fp32 [c_out][3][3][3] (OIHW), bias fp32 [c_out]; c_out == 64
 * img_of : int32 [b*t] -> destination image index in y, or -1 to skip that frame
 * y      : padded NHWC [n_img][h+2][w+2][c_out], halo 1, dtype as given
 */
int vnqa_conv_first_fwd(const float* clip, const float* w, const float* bias,
                        const int32_t* img_of, void* y, int32_t b, int32_t t, int32_t h,
                        int32_t wd, int32_t c_out, int32_t dtype, void* stream);

/* Repack an OIHW fp32 conv weight into the K-major layout the igemm consumes.
 *   transpose_flip == 0:  wt[o][tap][i]        = w[o][i][r][s] * (out_scale ? out_scale[o] : 1)
 *   transpose_flip == 1:  wt[i][(2-r)*3+(2-s)][o] = w[o][i][r][s]        (dgrad weights)
 * rows/cols beyond the real sizes are zero-filled up to rows_pad / k_pad channels.
 */
int vnqa_pack_conv_weight(const float* w_oihw, int32_t c_out, int32_t c_in, int32_t taps,
                          int32_t c_out_pad, int32_t c_in_pad, const float* out_scale,
                          int32_t transpose_flip, int32_t dtype, void* wt, void* stream);

/* Pre-tiled weights: the exact LDS image of every (cout tile, K stage) weight tile, swizzle included, laid
 * out contiguously in the order the igemm consumes them (channel chunk outer, tap inner), so that each
 * wave-level DMA instruction of the B operand reads 1 KiB of consecutive bytes.  Valid for the 128-byte-row
 * tiles (VNQA_TILE_256x256 / 256x128 / 256x64 / 128x128 / 128x64 / STEM_256x256); `tile` must be the id the
 * conv will be launched with.  Size: vnqa_conv_weight_tiled_bytes().
 */
int64_t vnqa_conv_weight_tiled_bytes(int32_t c_out, int32_t c_in_pad, int32_t taps, int32_t tile, int32_t dtype);
int vnqa_pack_conv_weight_tiled(const float* w_oihw, int32_t c_out, int32_t c_in, int32_t taps,
                                int32_t c_in_pad, const float* out_scale, int32_t tile, int32_t dtype,
                                void* wt_tiled, void* stream);

/* Inverse of the above for gradients: dW (fp32, [c_out_pad][taps][c_in_pad]) -> OIHW fp32. */
int vnqa_unpack_conv_wgrad(const float* dwt, int32_t c_out, int32_t c_in, int32_t taps,
                           int32_t c_out_pad, int32_t c_in_pad, float* dw_oihw, void* stream);
int vnqa_unpack_conv_wgrad_scaled(const float* dwt, int32_t c_out, int32_t c_in, int32_t taps, int32_t c_out_pad,
                                  int32_t c_in_pad, float* dw_oihw, float alpha, void* stream);   /* dw = alpha * un-packed */
/* ... with a second factor that only exists on the DEVICE (alpha_dev, may be NULL): dw = alpha * *alpha_dev * un-packed — e.g.
 * the inverse of a scale chosen on the device, applied here instead of by a pass of its own */
int vnqa_unpack_conv_wgrad_dev(const float* dwt, int32_t c_out, int32_t c_in, int32_t taps, int32_t c_out_pad, int32_t c_in_pad,
                               float* dw_oihw, float alpha, const float* alpha_dev, void* stream);

/* fc_embed_attn = nn.Linear(spatial*C -> at_hidden) applied to the NCHW-flattened feature map
 * (models/film_attn_pt_stem.py:56-57,244).  The kernels keep maps as padded NHWC, so its weight is re-laid out once
 * per step:  w fp32 [rows][c][h][w]  ->  nat [rows_pad][(h+2)(w+2)][c_pad] (forward GEMM operand) and, when nat_t is
 * not NULL, nat_t [(h+2)(w+2)][c_pad][rows_pad] (its transpose, the dX GEMM operand), both in `dtype`, zeros on halo
 * positions / padded rows / padded channels (no memset needed).  vnqa_unpack_fc_wgrad maps the fp32 gradient of
 * `nat` back to the parameter's layout.  rows_pad % 8 == 0, c_pad % 64 == 0.
 */
int vnqa_pack_fc_weight(const float* w, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t rows_pad,
                        int32_t c_pad, int32_t dtype, void* nat, void* nat_t, void* stream);
int vnqa_unpack_fc_wgrad(const float* dw_nat, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t c_pad,
                         float* dw, void* stream);
/* Input gradient of that Linear from the FORWARD operand alone (round 2; replaces torch's `grad_output @ weight` of
 * models/film_attn_pt_stem.py:244 in backward):  dx[m][k] = sum_r dout[m][r] * nat[r][k],  dout [m][r], nat [r][k],
 * dx [m][k], all in the library's 16-bit format (dtype VNQA_BF16), r == 128, k % 128 == 0.  Per 320 rows of dout (kept in LDS)
 * `nat` is streamed once in [128][128] slabs read through the transposed LDS read and dx is written once — no `nat_t` needed.
 */
int vnqa_fc_dx(const void* dout, const void* nat, void* dx, int32_t m, int32_t r, int32_t k, int32_t dtype, void* stream);
int vnqa_unpack_fc_wgrad_scaled(const float* dw_nat, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t c_pad, float* dw,
                                float alpha, void* stream);
int vnqa_unpack_fc_wgrad_dev(const float* dw_nat, int32_t rows, int32_t c, int32_t h, int32_t wd, int32_t c_pad, float* dw,
                             float alpha, const float* alpha_dev, void* stream);      /* dw = alpha * *alpha_dev * un-packed */

/* Layout converters between the reference's tensors and padded NHWC.
 *   vnqa_feat_to_nhwc : v fp32 [b][c][h][w][t] (the model-input layout, eval/q_and_v_eval.py:110)
 *                       -> y padded NHWC [n_img][h+2][w+2][c_pad], frame (b,t) -> image img_of[b*t_n+t]
 *   vnqa_nchw_to_nhwc : dense fp32 [n][c][h][w] -> padded NHWC halo 1
 *   vnqa_nhwc_to_nchw : padded NHWC (halo 0/1) -> dense fp32 [n][c][h][w]
 */
int vnqa_feat_to_nhwc(const float* v, const int32_t* img_of, void* y, int32_t b, int32_t c,
                      int32_t h, int32_t w, int32_t t, int32_t c_pad, int32_t dtype, void* stream);
int vnqa_nchw_to_nhwc(const float* x, void* y, int32_t n_img, int32_t c, int32_t h, int32_t w,
                      int32_t c_pad, int32_t dtype, void* stream);
int vnqa_nhwc_to_nchw(const void* x, float* out, int32_t n_img, int32_t c, int32_t h, int32_t w,
                      int32_t c_pad, int32_t halo, int32_t dtype, void* stream);

/* Zero the 1-pixel halo ring of a padded NHWC tensor [n_img][hp][wp][c] (the invariant every conv input relies on):
 * a fresh output buffer needs only this, not a memset of the whole tensor — the conv kernels write every interior
 * pixel and never the halo.  c * element size must be a multiple of 16 bytes. */
int vnqa_zero_halo(void* y, int32_t n_img, int32_t hp, int32_t wp, int32_t c, int32_t dtype, void* stream);

/* The packed image list of a length-sorted minibatch (film_attn_pt_stem.py:201-208: frame t is processed for the cts[t] samples that
 * have it) as device tables, written by ONE small kernel from HOST arrays that travel as kernel arguments — no host-to-device copy on
 * the stem's stream (a 4-KB copy there queues behind the clips' own PCIe transfer):
 *   v_sorted_host[batch] lengths sorted descending (0..frames), perm_host[batch]: sorted position s holds ORIGINAL sample perm[s];
 *   img_of[batch * frames]  : image index of (original sample b, frame t) at b * frames + t, or -1;
 *   frame_of / sample_of [n_img = sum v] : frame and sorted sample of image n (frame-major order);
 *   offsets[v_sorted[0] + 1] : first image of every frame, then n_img. */
#define VNQA_LAYOUT_MAX_BATCH 256
int vnqa_frame_layout(const int32_t* v_sorted_host, const int32_t* perm_host, int32_t batch, int32_t frames, int32_t* img_of,
                      int32_t* frame_of, int32_t* sample_of, int32_t* offsets, void* stream);

/* conv2d weight gradient (+ optional bias gradient), stride 1 'same'.
 * Replaces the autograd wgrad of nn.Conv2d on the trainable convs
 * (models/film_attn_pt_stem.py:40,98 reached by loss.backward(), eval/q_and_v_eval.py:136).
 *   dwt[o][tap][i] = sum_{n,y,x} dy[n,y,x,o] * x[n,y+r-1,x+s-1,i]      (fp32 output)
 *   dbias[o]       = sum_{n,y,x} dy[n,y,x,o]                            (fp32, nullable)
 * x : padded NHWC [n_img][h+2][w+2][c_in] (halo 1), dy : padded NHWC [n_img][h+2][w+2][c_out]
 * (halo 1, ZERO halo).  workspace: fp32, vnqa_conv2d_wgrad_workspace() bytes.
 * dtype may carry the option bit VNQA_WGRAD_FUSED_REDUCE: the last-arriving workgroup of every output tile folds the split-K
 * slabs itself (no reduce launch; bf16, <= 8192 tiles — measured slower end to end, DESIGN 5: an A/B option, off by default).
 */
#define VNQA_WGRAD_FUSED_REDUCE 0x100
/* VNQA_WGRAD_X_PAIR / _X_TRIPLE (16-bit format): x is a [hi | lo] / [hi | lo | hi] tensor with 2 / 3 c_in physical channels per pixel
 * (the output of a producer launched with VNQA_CONV_DUAL_OUT [| VNQA_CONV_DUAL_HI2]); the gradient contracts its first segment —
 * h16(x), what a plain 16-bit tensor would hold — with no copy (precision 'fp16h': conv_init's weight gradient from the stem's
 * split features). */
#define VNQA_WGRAD_X_PAIR 0x200
#define VNQA_WGRAD_X_TRIPLE 0x400
/* VNQA_WGRAD_EIGHT_WAVES (16-bit format): run the first form of the kernel (8 waves, two 64-pixel stages) instead of the default
 * 4-wave form with its ring of four 32-pixel stages — same tiles, slabs and summation order per slab; the tests' cross-check. */
#define VNQA_WGRAD_EIGHT_WAVES 0x800
int64_t vnqa_conv2d_wgrad_workspace(int32_t n_img, int32_t h, int32_t w, int32_t c_in,
                                    int32_t c_out, int32_t taps);
int vnqa_conv2d_wgrad(const void* x, const void* dy, float* dwt, float* dbias, void* workspace,
                      int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out,
                      int32_t taps, int32_t dtype, void* stream);

/* 3-D counterpart of vnqa_conv2d_wgrad (taps = 27): x [n][d+2][h+2][w+2][c_in], dy [n][d+2][h+2][w+2][c_out]. */
int64_t vnqa_conv3d_wgrad_workspace(int32_t n_img, int32_t d, int32_t h, int32_t w, int32_t c_in, int32_t c_out);
int vnqa_conv3d_wgrad(const void* x, const void* dy, float* dwt, float* dbias, void* workspace, int32_t n_img,
                      int32_t d, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t dtype, void* stream);

/* Dense GEMMs on the same MFMA kernels (nn.Linear fc_embed_attn forward/backward,
 * models/film_attn_pt_stem.py:57,244).
 *   vnqa_gemm_nt : out[m][n] = act( sum_k a[m][k] * b[n][k] + bias[n] )   (a, b, out: dtype; K-major operands)
 *                  K is split over workgroups when m x n alone cannot fill the chip; workspace of
 *                  vnqa_gemm_nt_workspace() bytes (may be 0 -> NULL allowed).  ldo = row stride of out.
 *   vnqa_gemm_tn : out[m][n] = sum_k a[k][m] * b[k][n]   (fp32 out; the wgrad kernel with explicit K)
  * workspace == NULL opts out of split-K (single pass over K, summation order independent of m).
 */
int64_t vnqa_gemm_nt_workspace(int32_t m, int32_t n, int32_t k, int32_t dtype);
int vnqa_gemm_nt(const void* a_mk, const void* b_nk, const float* bias, void* out, void* workspace,
                 int32_t m, int32_t n, int32_t k, int32_t ldo, int32_t relu, int32_t dtype, void* stream);
/* Grouped form: `groups` independent products with one shape in ONE launch — a [groups][m_group][k], b [groups][n][k],
 * out [groups][m_group][ldo]; m_group a multiple of the row tile (256 bf16 / 128 f32; pad rows are computed and ignored).
 * No bias, no split-K.  Used for the four edge products of the composed stem conv's border correction (stem.py). */
int vnqa_gemm_nt_grouped(const void* a_gmk, const void* b_gnk, void* out, int32_t groups, int32_t m_group,
                         int32_t n, int32_t k, int32_t ldo, int32_t dtype, void* stream);
int64_t vnqa_gemm_tn_workspace(int32_t m, int32_t n, int32_t k, int32_t dtype);
int vnqa_gemm_tn(const void* a_km, const void* b_kn, float* out, void* workspace, int32_t m, int32_t n,
                 int32_t k, int32_t dtype, void* stream);

/* Fused memory-bound glue of the FiLM trunk on padded-NHWC activations [n_img][hp][wp][c], c % 64 == 0,
 * zero halo (outputs get their halo written as zero).
 *   vnqa_frame_bn_stats : per-(frame, channel) mean / biased variance over the frame's images, i.e. the
 *                         batch statistics of train-mode bn_init applied frame by frame
 *                         (models/film_attn_pt_stem.py:211); frame f owns images [frame_off[f], frame_off[f+1])
 *   vnqa_frame_bn_apply : y = (x - mean[f]) * rstd[f] * gamma + beta, f = frame_of[image]
 *   vnqa_frame_bn_bwd   : BatchNorm backward per frame (s1 = sum dy, s2 = sum dy*xhat are returned for
 *                         dbeta/dgamma); relu_mask != 0 also applies the mask of the ReLU that produced x
 *   vnqa_film_relu_res_fwd : out = relu(gamma[n] * z + beta[n]) + res            (:229-241)
 *   vnqa_film_relu_res_bwd : dz, dgamma[n][c], dbeta[n][c] (wave/LDS reductions over pixels); dres == dout
 *   vnqa_relu_bwd       : g = (a [+ b]) * [y > 0]
 */
int vnqa_frame_bn_stats(const void* x, const int32_t* frame_off, float* mean, float* var, int32_t n_frames,
                        int32_t hp, int32_t wp, int32_t c, int32_t dtype, void* stream);
int vnqa_frame_bn_apply(const void* x, const int32_t* frame_of, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, void* y, int32_t n_img, int32_t hp,
                        int32_t wp, int32_t c, int32_t dtype, void* stream);
/* ... of a SPLIT tensor (two 16-bit tensors x_hi + x_lo, a conv's VNQA_EPI_SPLIT_OUT output): statistics and normalisation of the
 * unrounded value; y is ONE 16-bit tensor (the single rounding of the layer) */
int vnqa_frame_bn_stats_split(const void* x_hi, const void* x_lo, const int32_t* frame_off, float* mean, float* var,
                              int32_t n_frames, int32_t hp, int32_t wp, int32_t c, void* stream);
int vnqa_frame_bn_apply_split(const void* x_hi, const void* x_lo, const int32_t* frame_of, const float* mean, const float* rstd,
                              const float* gamma, const float* beta, void* y, int32_t n_img, int32_t hp, int32_t wp,
                              int32_t c, void* stream);
int vnqa_frame_bn_bwd(const void* dy, const void* x, const int32_t* frame_of, const int32_t* frame_off,
                      const float* mean, const float* rstd, const float* gamma, float* s1, float* s2,
                      void* dx, int32_t n_img, int32_t n_frames, int32_t hp, int32_t wp, int32_t c,
                      int32_t relu_mask, int32_t dtype, void* stream);
int vnqa_film_relu_res_fwd(const void* z, const void* res, const float* gamma, const float* beta, void* out,
                           int32_t n_img, int32_t hp, int32_t wp, int32_t c, int32_t dtype, void* stream);
int vnqa_film_relu_res_bwd(const void* dout, const void* z, const float* gamma, const float* beta, void* dz,
                           float* dgamma, float* dbeta, int32_t n_img, int32_t hp, int32_t wp, int32_t c,
                           int32_t dtype, void* stream);
/* The same two ops on gamma/beta that are COLUMN SLICES of a wider fp32 matrix (the FiLM generator's output
 * [n_img][2*C*blocks], models/film_attn_pt_stem.py:229-233): row stride film_ld (floats), channels >= film_c are padding
 * (gamma = beta = 0, no gradient written); dgamma/dbeta rows have stride grad_ld. */
int vnqa_film_relu_res_fwd_ld(const void* z, const void* res, const float* gamma, const float* beta, void* out,
                              int32_t n_img, int32_t hp, int32_t wp, int32_t c, int32_t film_ld, int32_t film_c,
                              int32_t dtype, void* stream);
int vnqa_film_relu_res_bwd_ld(const void* dout, const void* z, const float* gamma, const float* beta, void* dz,
                              float* dgamma, float* dbeta, int32_t n_img, int32_t hp, int32_t wp, int32_t c,
                              int32_t film_ld, int32_t film_c, int32_t grad_ld, int32_t dtype, void* stream);
int vnqa_relu_bwd(const void* a, const void* b, const void* y, void* g, int64_t n, int32_t dtype, void* stream);

/* Temporal softmax-attention over frames, fused (models/film_attn_pt_stem.py:268-290):
 *   score[b,t] = valid[b,t]*(w . feat[b,t,:] + bias) + mask[b,t];  coef = softmax_t(score);  ctxt[b,:] = sum_t coef*feat
 * feat fp32 [b][t][a] (frame rows contiguous), valid/mask/coef fp32 [b][t], w fp32 [a], bias fp32 [1], ctxt fp32 [b][a].
 * backward: dfeat [b][t][a], per-sample partials dw_part [b][a] and db_part [b] (summed over b by the caller).
 */
int vnqa_temporal_attn_fwd(const float* feat, const float* valid, const float* mask, const float* w,
                           const float* bias, float* coef, float* ctxt, int32_t b, int32_t t, int32_t a,
                           void* stream);
int vnqa_temporal_attn_bwd(const float* feat, const float* valid, const float* w, const float* coef,
                           const float* dctxt, float* dfeat, float* dw_part, float* db_part, int32_t b,
                           int32_t t, int32_t a, void* stream);
/* The same op on the PACKED image list (what the product path runs): f [n_img][ld] in `dtype` is the output of the
 * fc_embed_attn GEMM, image n = frame_off[t] + b for the valid (sample b, frame t) pairs; the zero-padded [B][T][A] tensor,
 * the validity grid and the -(1<<31) masks of models/film_attn_pt_stem.py:245-256 are formed on the fly.  The backward
 * writes d f [n_img][ld] in `dtype` (padding columns a..ld-1 zero), ready to be the next GEMM's operand. */
int vnqa_temporal_attn_packed_fwd(const void* f, int32_t ld, int32_t dtype, const int32_t* frame_off, int32_t n_frames,
                                  const float* w, const float* bias, float* coef, float* ctxt, int32_t b, int32_t t,
                                  int32_t a, void* stream);
int vnqa_temporal_attn_packed_bwd(const void* f, int32_t ld, int32_t dtype, const int32_t* frame_off, int32_t n_frames,
                                  const float* w, const float* coef, const float* dctxt, void* df, float* dw_part,
                                  float* db_part, int32_t b, int32_t t, int32_t a, float grad_scale, void* stream);
                                  /* grad_scale: d f is multiplied by it BEFORE rounding to `dtype` — the loss scale of the
                                   * fp16 storage build (activation gradients of 1e-5..1e-7 would sit in fp16's subnormal
                                   * range); the consumers divide their fp32 results by it (…_scaled un-pack entry points) */

/* ---------------------------------------------------------------------------------------
 * Small fp32 pieces of the question path / classifier / loss (csrc/glue.hip): with them a training step of the FiLM models
 * enqueues no ATen / rocBLAS kernel for these ops.  All exact fp32, fixed summation order, no atomics.
 *
 * vnqa_sgemm: C[row_c(m)][n] = act( sum_k A'(m,k) * B(k,n) + bias[n] ) [+ C], with
 *     A'(m,k) = A[row_a(m) * a_rs + k * a_cs] * [a_mask(same index) > 0],   B(k,n) = B[k * b_rs + n * b_cs]
 *   (element strides: NN / NT / TN products without copies); a_rows / c_rows optional int32 [m] row maps (negative: zero
 *   row / row not written); a_mask optional (the ReLU mask of a Linear+ReLU backward).  Replaces nn.Linear at
 *   models/film_attn_pt_stem.py:179 (FiLM generator Linear+ReLU), :293 (LSTMCell input projection), :301 (out_linear),
 *   models/time_multi_hop_pt_stem.py:179 (fc_attn_out) and the backward GEMMs of each.
 * vnqa_colsum: out[n] = sum_m x[m][n] * [mask[m][n] > 0]  (bias gradients).
 * vnqa_gather_rows: dst[r] = src[rows[r]] (negative: zeros) — the LSTM output at the last token of every repeat
 *   (film_attn_pt_stem.py:163-171).
 * vnqa_embed_proj_fwd: xg[b][pos] = W_ih embed[tokens[row_perm[b]][pos]] + b_ih + b_hh — nn.Embedding (:146) fused with the
 *   input half of nn.LSTM (:160): the lookup is the row gather of the GEMM's A operand.  vnqa_token_dsum: dsum[v] = sum of d xg over the positions holding token v, from which
 *   d embed = dsum W_ih, d W_ih = dsum^T embed, d b = colsum(dsum) (rows of equal tokens share their embedding).
 * vnqa_lstm_fold_dxg: d xg[b][pos] = sum over the n_rep repeats of d gates (the question is re-run once per frame, :213).
 * vnqa_lstm_wgrad_operands: the two operands of dW_hh = sum d gates^T h_prev in the GEMM's element type, one pass.
 * vnqa_ce_loss: nn.CrossEntropyLoss(weight, reduction sum | mean) forward and d logits (eval/q_and_v_eval.py:124);
 *   ys int64 [b] read through row_perm (the batch sort of :113-116).
 * vnqa_bn_running_update: bn_init's running statistics advanced once per processed frame in frame order
 *   (film_attn_pt_stem.py:211: one BatchNorm2d call per frame; momentum 0.1, unbiased variance).
 */
int64_t vnqa_sgemm_workspace(int32_t m, int32_t n, int32_t k);   /* bytes of split-K scratch (0: none needed) */
int vnqa_sgemm(const float* a, const float* b, float* c, const float* bias, const float* a_mask, const int32_t* a_rows,
               const int32_t* c_rows, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs, int32_t ldc, int32_t m,
               int32_t n, int32_t k, int32_t relu, int32_t accumulate, const float* addend, void* workspace, void* stream);
/* vnqa_sgemm (no gather / scatter / mask / accumulate / ReLU) with a SECOND output written by the same epilogue:
 * out2[m][n] = C[m][n] * out2_col[n] * out2_mul[m][n] (either factor may be NULL; out2 / out2_mul rows of ldc floats) — the
 * elementwise products that follow MACNetwork's reasoning-step projections (mac.py:33,57-58). */
int vnqa_sgemm2(const float* a, const float* b, float* c, const float* bias, int64_t a_rs, int64_t a_cs, int64_t b_rs, int64_t b_cs,
                int32_t ldc, int32_t m, int32_t n, int32_t k, const float* addend, float* out2, const float* out2_col,
                const float* out2_mul, void* workspace, void* stream);

/* Up to VNQA_SGEMM_BATCH_MAX independent vnqa_sgemm / vnqa_sgemm2 products in ONE launch (one pass over K each, no row
 * gather / scatter / operand mask): the small fp32 products of a MACNetwork reasoning step that do not depend on each other
 * (models/mac.py:31-32 with :55 and :84; their gradients) cost the step's dependent chain one launch instead of two or three.
 * Every field as the same-named argument of vnqa_sgemm / vnqa_sgemm2; results are bit-identical to separate calls without a
 * workspace.  Problems whose outputs exceed 1024 tiles of 32 x 32 are run one after the other instead. */
typedef struct vnqa_sgemm_problem {
  const float* a;
  const float* b;
  float* c;
  const float* bias;
  const float* addend;
  float* out2;
  const float* out2_col;
  const float* out2_mul;
  int64_t a_rs, a_cs, b_rs, b_cs;
  int32_t ldc, m, n, k, relu, accumulate;
} vnqa_sgemm_problem;
#define VNQA_SGEMM_BATCH_MAX 4
int vnqa_sgemm_batch(const vnqa_sgemm_problem* problems, int32_t count, void* stream);
               /* addend: optional fp32 matrix [m][ldc] added to the product (torch.addmm's first argument) */
               /* workspace: vnqa_sgemm_workspace bytes (skinny outputs over a long K are split over workgroups and summed in
                * slice order by a second launch); NULL = one pass over K */
int vnqa_colsum(const void* x, const float* mask, float* out, int32_t rows, int32_t cols, int32_t ld, int32_t dtype,
                void* stream);   /* x in `dtype`; a mask needs dtype == VNQA_F32 */
int vnqa_gather_rows(const float* src, const int32_t* rows, float* dst, int32_t n_rows, int32_t cols, void* stream);
int vnqa_embed_proj_fwd(const int64_t* tokens, const int32_t* row_perm, const float* embed, const float* w_ih,
                        const float* b_ih, const float* b_hh, float* xg, int32_t* rows, int32_t b, int32_t lq, int32_t e,
                        int32_t g, int32_t vocab, void* stream);   /* rows: out, int32 [b*lq] token per position (kept for the backward) */
int vnqa_token_dsum(const int32_t* rows, const float* dxg, float* dsum, int32_t n_pos, int32_t g, int32_t vocab,
                    void* stream);
int vnqa_lstm_fold_dxg(const float* dgates, const int32_t* q_lens, float* dxg, int32_t b, int32_t lq, int32_t s,
                       int32_t hidden, int32_t n_rep, void* stream);
int vnqa_lstm_wgrad_operands(const float* dgates, const float* hs, const float* h0, void* a, void* hp, int32_t b, int32_t s,
                             int32_t hidden, int32_t dtype, void* stream);
int vnqa_ce_loss(const float* logits, const int64_t* ys, const int32_t* row_perm, const float* weight, float* loss,
                 float* dlogits, int32_t b, int32_t k, int32_t mean, void* stream);
int vnqa_bn_running_update(const float* mean, const float* var, const int32_t* frame_off, float* running_mean,
                           float* running_var, int32_t n_frames, int32_t pixels_per_image, int32_t c, int32_t ld,
                           float momentum, void* stream);

/* Persistent LSTM over a repeated sequence (one workgroup per sample, W_hh rows in registers).
 * Replaces the per-frame packed nn.LSTM calls with carried state of compute_film_values /
 * compute_film_encoding (models/film_attn_pt_stem.py:146-171 called at :213;
 * models/time_multi_hop_pt_stem.py:124-158) and, with q_lens == 1, the 35-step nn.LSTMCell chain of the
 * attention tail (models/film_attn_pt_stem.py:283-295).
 * Sample b advances q_lens[b] * n_rep cells; at cell t its input gates are xg[b][t % q_lens[b]].
 *   xg    : fp32 [b][lq][4*hidden]  = x W_ih^T + b_ih + b_hh, gate order i,f,g,o
 *   w_hh  : fp32 [4*hidden][hidden]
 *   hs    : fp32 [b][s][hidden]   h after every cell (s >= max q_len*n_rep; rows past a sample's end untouched)
 *   gates : fp32 [b][s][5*hidden] activated i,f,g,o and c per cell (saved for backward)
 * backward: dhs = external gradient on every cell's h (same shape as hs, zero where unused);
 *   dgates fp32 [b][s][4*hidden] = gradient w.r.t. the gate pre-activations of every cell; the caller
 *   forms dW_hh = sum dgates^T h_prev, dxg[b][pos] = sum over repeats, db = sum dgates from it.
 * hidden in {16, 32, 64, 128}.
 */
int vnqa_lstm_seq_fwd(const float* xg, const float* w_hh, const int32_t* q_lens, const float* h0,
                      const float* c0, float* hs, float* gates, float* hN, float* cN, int32_t b,
                      int32_t lq, int32_t hidden, int32_t s, int32_t n_rep, void* stream);
int vnqa_lstm_seq_bwd(const float* w_hh, const int32_t* q_lens, const float* c0, const float* gates,
                      const float* dhs, const float* dhN, const float* dcN, float* dgates, float* dh0,
                      float* dc0, int32_t b, int32_t hidden, int32_t s, int32_t n_rep, void* stream);

/* Packed-sequence LSTM with a WIDE hidden state, one launch per time step (all enqueued by this call).
 * Replaces nn.LSTM(embed_hidden, dim, bidirectional=True) and nn.LSTM(3*dim, 3*dim) of MACNetwork
 * (models/mac.py:185-186,193 run at :210-213 and :249-251 on pack_padded_sequence inputs).
 *   xg     : fp32 [t][b][4*hidden] time-major input projection (x W_ih^T + b_ih + b_hh), gates i,f,g,o
 *   w_hh   : fp32 [4*hidden][hidden];  w_hh_t (backward): its transpose [hidden][4*hidden]
 *   h0, c0 : fp32 [b][hidden] initial state of every sample at ITS first step, or NULL for zeros
 *   batch_sizes_host : HOST int32 [t], batch_sizes[i] = #samples with length > i (samples sorted by
 *                      length descending; positive, <= b, non-increasing) — PackedSequence.batch_sizes
 *   hs, cs : fp32 [t][b][hidden] h / c after each step;  gates : fp32 [t][b][4*hidden] activated gates.
 *            Rows of inactive (step, sample) pairs are not written: pass zero-filled buffers.
 *   reverse: 0 = walk steps 0..t-1; 1 = walk t-1..0 (second direction of a bidirectional LSTM).
 * backward: dhs = gradient on every step's h (zero where unused); dc_work fp32 [b][hidden] ZEROED scratch;
 *   dgates fp32 [t][b][4*hidden] (pass zero-filled) = gradient w.r.t. the gate pre-activations = d xg; the
 *   caller forms dW_hh = sum_t dgates_t^T h_pred(t) with vnqa_gemm_tn.  No gradient is produced for h0/c0.
 * hidden must be a multiple of 4.  Exact fp32.
 */
int vnqa_lstm_wide_fwd(const float* xg, const float* w_hh, const float* h0, const float* c0,
                       const int32_t* batch_sizes_host, float* hs, float* cs, float* gates, int32_t t,
                       int32_t b, int32_t hidden, int32_t reverse, void* stream);
int vnqa_lstm_wide_bwd(const float* w_hh_t, const float* c0, const int32_t* batch_sizes_host,
                       const float* gates, const float* cs, const float* dhs, float* dgates,
                       float* dc_work, int32_t t, int32_t b, int32_t hidden, int32_t reverse, void* stream);

/* Both directions of a bidirectional packed LSTM from zero states (models/mac.py:185,210-213: the question encoder) with
 * chain position i of the forward direction (step i) and of the reverse direction (step t-1-i) in ONE launch: t launches
 * instead of 2 t on the dependent chain.  Buffers as vnqa_lstm_wide_fwd / _bwd, one set per direction (`_f` walks 0..t-1,
 * `_r` walks t-1..0); dc_work_f / dc_work_r are two distinct zeroed [b][hidden] scratch buffers.  Results are bit-identical
 * to two single-direction calls.
 */
int vnqa_lstm_wide_bidir_fwd(const float* xg_f, const float* xg_r, const float* w_hh_f, const float* w_hh_r,
                             const int32_t* batch_sizes_host, float* hs_f, float* hs_r, float* cs_f, float* cs_r,
                             float* gates_f, float* gates_r, int32_t t, int32_t b, int32_t hidden, void* stream);
int vnqa_lstm_wide_bidir_bwd(const float* w_hh_t_f, const float* w_hh_t_r, const int32_t* batch_sizes_host,
                             const float* gates_f, const float* gates_r, const float* cs_f, const float* cs_r,
                             const float* dhs_f, const float* dhs_r, float* dgates_f, float* dgates_r,
                             float* dc_work_f, float* dc_work_r, int32_t t, int32_t b, int32_t hidden, void* stream);

/* Fused ReadUnit attention of MACNetwork (models/mac.py:53-62; replaces, per reasoning step, the Linear(2d->d)
 * over [mem*know ; know] at every position, the control-weighted Linear(d->1), the softmax over positions and the
 * weighted sum — see videonavqa_amd/models/mac.py for the re-association).
 *   know, pre : [n][s][ld] in `dtype` (ld >= c channels per row, both multiples of 8); pre = know W2^T + b
 *   u, v      : fp32 [n][c]   score[n][s] = know[n][s].u[n] + pre[n][s].v[n] + bias[0]
 *   p         : fp32 [n][s]   softmax over s (saved for backward);  read : fp32 [n][c] = sum_s p know
 * backward: dread fp32 [n][c] -> dscore fp32 [n][s], du, dv fp32 [n][c].  The outer-product gradients of know / pre
 * are formed ONCE for all k reasoning steps by vnqa_mac_read_accum from the stacked per-step factors
 * (dscore, p: [k][n][s]; u, v, dread: [k][n][c]):  dknow = sum_i dscore_i (x) u_i + p_i (x) dread_i,
 * dpre = sum_i dscore_i (x) v_i, written in `dtype` with row stride ld (channels >= c zero).
 * pre / v / dv / dpre may all be NULL: the kernels are then a plain dot-product attention pool over `know`, used for
 * ControlUnit's attention over the question words (models/mac.py:36-42).
 * s <= 1024; 3*k*c + 128*k floats must fit 160 KiB of LDS in vnqa_mac_read_accum.
 */
int vnqa_mac_read_fwd(const void* know, const void* pre, const float* u, const float* v, const float* bias,
                      float* p, float* read, int32_t n, int32_t s, int32_t c, int32_t ld, int32_t dtype,
                      void* stream);
int vnqa_mac_read_bwd(const void* know, const void* pre, const float* p, const float* dread, float* dscore,
                      float* du, float* dv, int32_t n, int32_t s, int32_t c, int32_t ld, int32_t dtype,
                      void* stream);
/* The same two kernels with the elementwise steps that sit between them in a reasoning step folded in (models/mac.py:36-42
 * with :57-58: control' = pool(...) [* dropout mask], v = control' * w_attn and their gradients):
 *   _fwd_scaled : read = pool * out_mask (fp32 [n][c], or NULL);  out2 = read * out2_col[c] (fp32 [n][c] / [c]; both or neither)
 *   _bwd_fused  : pro_x != NULL: dread is an OUTPUT first — dread = (pro_x * pro_col[c] + pro_add) * pro_mask, with
 *                 pro_x / pro_add / pro_mask fp32 [n][c] (the latter two may be NULL), pro_col fp32 [c] — and then used as
 *                 in vnqa_mac_read_bwd;  du2 = du * du2_col[c] (both or neither).
 */
int vnqa_mac_read_fwd_scaled(const void* know, const void* pre, const float* u, const float* v, const float* bias, float* p,
                             float* read, const float* out_mask, float* out2, const float* out2_col, int32_t n, int32_t s,
                             int32_t c, int32_t ld, int32_t dtype, void* stream);
int vnqa_mac_read_bwd_fused(const void* know, const void* pre, const float* p, float* dread, const float* pro_x,
                            const float* pro_col, const float* pro_add, const float* pro_mask, float* dscore, float* du,
                            float* dv, float* du2, const float* du2_col, int32_t n, int32_t s, int32_t c, int32_t ld,
                            int32_t dtype, void* stream);
int vnqa_mac_read_accum(const float* dscore, const float* p, const float* u, const float* v,
                        const float* dread, void* dknow, void* dpre, int32_t k, int32_t n, int32_t s,
                        int32_t c, int32_t ld, int32_t dtype, void* stream);

/* One MAC reasoning step for all packed images (ControlUnit, ReadUnit, WriteUnit.concat of the reference's
 * models/mac.py:28-42,53-62,82-85) as ONE call per direction: the ~11 forward / ~25 backward launches (fp32 GEMMs on
 * vnqa_sgemm, the attention pools above, elementwise products) are enqueued from C++ instead of one by one from Python —
 * `--model mac` was bound by the launch thread, not by the GPU.
 *   forward : cq = control Wc^T + pq;  control' = pool(ctxw; cq * w_ca, b_ca) [* mask_c];  mem = memory Wm^T + bm;
 *             v = control' * w_ra;  t = v W1;  u = mem * t;  read = pool(know, pre; u, v, b_ra);
 *             concat = read Wr^T + memory Wmm^T + bw        (every intermediate is an output: the backward reads them)
 *   backward: from d_concat and (optional) d_cnew: d_control, d_memory, d_cq, the per-step attention factors
 *             (ds_r, d_read, ds_c, d_c — the caller stacks them over steps for vnqa_mac_read_accum) and the parameter
 *             gradients ACCUMULATED (+=) into g_*: g_wca / g_wra are per-image [n][d] partial sums of the two attention
 *             weight vectors' gradients (the caller sums them over images once), g_bm / g_bw vectors, the rest [d][d].
 * All matrices fp32 row-major [n][d] unless noted; know / pre [n*s][ld] in `dtype`; ctxw fp32 [n*lq][d]; ones fp32 [n] = 1.
 */
typedef struct vnqa_mac_core {
  int32_t n, d, lq, s, ld, dtype;
  const float *control, *memory, *pq, *ctxw;
  const void *know, *pre;
  const float* mask_c;                                             /* [n][d] or NULL */
  const float *wc, *w_ca, *b_ca, *wm, *bm, *w1, *w_ra, *b_ra, *wr, *wmm, *bw;
  float *cq, *qv, *p_c, *cnew, *mem, *v, *t, *u, *p_r, *read, *concat;   /* forward outputs (p_c [n][lq], p_r [n][s]) */
  const float *d_cnew, *d_concat;                                  /* backward inputs (d_cnew may be NULL) */
  float *d_control, *d_memory, *d_cq;                              /* backward outputs */
  float *ds_r, *d_read, *ds_c, *d_c, *du, *dv, *dqv, *d_mem, *d_t; /* backward factors / scratch (ds_r [n][s], ds_c [n][lq]) */
  float *g_wc, *g_wca, *g_wm, *g_bm, *g_w1, *g_wra, *g_wr, *g_wmm, *g_bw;
  const float* ones;
  void* workspace;                                                 /* vnqa_mac_core_workspace(n, d) bytes, or NULL */
  int32_t defer_wgrad;   /* != 0: _bwd leaves every g_* untouched (they may be NULL); the caller keeps the per-step factors stacked
                          * step-major and forms all parameter gradients with ONE vnqa_mac_core_wgrad call after the last step */
} vnqa_mac_core;
int64_t vnqa_mac_core_workspace(int32_t n, int32_t d);
int vnqa_mac_core_fwd(const vnqa_mac_core* a, void* stream);
int vnqa_mac_core_bwd(const vnqa_mac_core* a, void* stream);

/* All `n_steps` reasoning steps in one call per direction, for MACNetwork without self-attention and memory gate (the reference's
 * defaults, models/mac.py:131-155 with :86-105 skipped):
 *     step i: (control_i, memory_i) -> vnqa_mac_core_fwd -> (cnew_i, concat_i);  control_{i+1} = cnew_i;
 *             memory_{i+1} = concat_i * mask_m       (mask_m fp32 [n][d], the memory dropout mask of :125-129, or NULL)
 * `step0` describes step 0 as for vnqa_mac_core_fwd / _bwd; every per-step buffer of step i lies i * (step 0's rows) further on —
 * [n][d] buffers i*n*d floats, p_c / ds_c i*n*lq, p_r / ds_r i*n*s, pq i*n*d: the caller's STEP-STACKED slabs, which are what
 * vnqa_mac_core_wgrad and vnqa_mac_read_accum take afterwards.  step0->memory is not read: memory_i = memories[i],
 *   memories : fp32 [n_steps + 1][n][d], memories[0] = the initial memory (in), memories[i + 1] written by step i; memories[n_steps]
 *              is the result.  control_0 = step0->control; control_i = cnew of step i - 1.
 * backward (defer_wgrad must be set): d_memory_out fp32 [n][d] = gradient on memories[n_steps];
 *   d_concat : fp32 [n_steps][n][d] out (d_concat_i = d memory_{i+1} * mask_m: a factor of vnqa_mac_core_wgrad);
 *   step0->d_control / d_memory slabs [n_steps][n][d]: entry 0 = the gradients on the initial control / memory.
 * step0->d_concat / d_cnew are ignored.  Results are bit-identical to n_steps vnqa_mac_core calls with the elementwise products between. */
int vnqa_mac_chain_fwd(const vnqa_mac_core* step0, int32_t n_steps, float* memories, const float* mask_m, void* stream);
int vnqa_mac_chain_bwd(const vnqa_mac_core* step0, int32_t n_steps, const float* memories, const float* mask_m,
                       const float* d_memory_out, float* d_concat, void* stream);

/* Parameter gradients of all reasoning steps in one call (the deferred form of vnqa_mac_core_bwd's g_* accumulation; same
 * reference lines, mac.py:28-42,53-62,82-85).  Every factor is the per-step [n][d] fp32 matrix stacked over the steps to
 * [rows = steps * n][d] (any consistent step order).  Outputs are WRITTEN: g_wr = d_concat^T read, g_wmm = d_concat^T memory,
 * g_w1 = v^T d_t, g_wm = d_mem^T memory, g_wc = d_cq^T control ([d][d]); g_bw / g_bm = column sums of d_concat / d_mem;
 * g_wra[j] = sum_r dv*cnew, g_wca[j] = sum_r dqv*cq ([d]). */
typedef struct vnqa_mac_wgrad {
  int32_t rows, d;
  const float *d_concat, *read, *memory, *v, *d_t, *d_mem, *d_cq, *control, *dv, *cnew, *dqv, *cq;
  float *g_wc, *g_wca, *g_wm, *g_bm, *g_w1, *g_wra, *g_wr, *g_wmm, *g_bw;
  void* workspace;                                                 /* vnqa_mac_core_wgrad_workspace(rows, d) bytes (required) */
} vnqa_mac_wgrad;
int64_t vnqa_mac_core_wgrad_workspace(int32_t rows, int32_t d);
int vnqa_mac_core_wgrad(const vnqa_mac_wgrad* w, void* stream);

/* Fused global-norm clip + Adam + zero_grad over flat fp32 buffers.
 * Replaces clip_grad_norm(model.parameters(), clip); optimizer.step(); optimizer.zero_grad()
 * (eval/q_and_v_eval.py:137-139, torch.optim.Adam defaults betas .9/.999 eps 1e-8).
 *   vnqa_l2norm_partial : partial[i] = sum of squares of block i's slice (n_partial = return of
 *                         vnqa_l2norm_blocks(n)); deterministic two-stage reduction
 *   vnqa_clip_adam      : coef = min(1, clip/(sqrt(sum partial)+1e-6)); g*=coef; Adam update of
 *                         p, m, v; g = 0.  When the gradient norm is not finite (fp16 storage: the loss scale overflowed;
 *                         any precision: an inf / NaN activation) the update is SKIPPED: p, m, v untouched, g zeroed, and
 *                         *overflow_count (optional device int32) incremented.
 *                         step = 1-based count of the LAUNCHES that shared this overflow_count; the bias correction uses the
 *                         number of updates actually applied, step - *overflow_count (>= 1), formed ON THE DEVICE — the
 *                         caller never has to subtract skipped steps from a host read-back (which would make replicas with
 *                         different host timing take different bias corrections on the same gradient).  The caller reads
 *                         *overflow_count back only to steer its loss scale.
 */
int32_t vnqa_l2norm_blocks(int64_t n);
int vnqa_l2norm_partial(const float* g, int64_t n, float* partial, void* stream);
int vnqa_clip_adam(float* p, float* g, float* m, float* v, int64_t n, const float* partial,
                   int32_t n_partial, float clip, float lr, float beta1, float beta2, float eps,
                   int32_t step, int32_t* overflow_count, void* stream);

/* ---------------------------------------------------------------------------------------
 * VideoOnlyCNN3D (models/v_only_cnn3d.py:13-37,59-81) — csrc/cnn3d.hip.  16-bit storage format only.
 *
 * BatchNorm in train mode is three steps: partial (sum, sum of squares) per block -> vnqa_bn_finalize (mean, rstd, running
 * statistics with the unbiased variance, momentum as nn.BatchNorm) -> apply.  Partials are [blocks][c][2] fp32 and are produced
 * by vnqa_c3d_stats_ncdhw (fp32 [n][c][s] input, blocks = n * chunks), vnqa_c3d_stats_rows (dense [rows][c],
 * blocks = vnqa_c3d_stats_blocks(rows)), vnqa_c3d_conv1_fwd (blocks = vnqa_c3d_conv1_fwd_blocks) and vnqa_pool444_fwd
 * (blocks = vnqa_pool444_blocks).
 *
 * vnqa_view5: where element (n, d, h, w, c) of a tensor lives, in ELEMENTS: base + n sn + d sd + h sh + w sw + c sc; the row index
 * of the dense operand runs over (n, d, h, w) with extents d, h, w.  Serves padded NDHWC (the next conv's input), dense rows and
 * the NC(DHW)-flattened fp32 feature matrix alike.
 *   vnqa_bn_rows_apply : out(view) = gamma (x - mean) rstd + beta, x dense [rows][c]
 *   vnqa_bn_rows_bwd   : dx dense [rows][c] = gamma rstd (dy - mean(dy) - x^ mean(dy x^)), dgamma = sum dy x^ / grad_scale,
 *                        dbeta = sum dy / grad_scale; workspace (vnqa_c3d_stats_blocks(rows) * c * 2 + 2 c) floats
 *   vnqa_pool444_fwd/bwd : MaxPool3d(4,4,4) of a padded NDHWC conv output y (ReLU already applied) -> p dense
 *                        [n][d/4][h/4][w/4][c] + arg-max bytes (255: maximum not positive) + statistics of p; backward writes the
 *                        whole interior of the padded dy (zero where not an arg-max), never the halo
 *   vnqa_c3d_conv1_fwd : p = MaxPool3d(1,2,2)(relu(conv3d(bn_input(x), weight) + bias)) straight from the fp32 [n][3][d][h][w]
 *                        clip: p dense [n][d][h/2][w/2][64], idx = arg-max 0..3 in (h, w) order (+4: not positive), statistics of p.
 *                        h, w multiples of 16 (vnqa_c3d_conv1_supported)
 *   vnqa_c3d_conv1_bwd : from dp (gradient wrt p): dweight [64][3][27], dbias [64], and bn_input's dgamma / dbeta [3], all
 *                        divided by grad_scale; partial: (2 * vnqa_c3d_conv1_bwd_blocks + 16) * 64 * 112 floats
 */
typedef struct vnqa_view5 {
  int64_t base, sn, sd, sh, sw, sc;
  int32_t d, h, w;
} vnqa_view5;
int32_t vnqa_c3d_stats_blocks(int64_t rows);
int vnqa_c3d_stats_ncdhw(const float* x, float* partial, int32_t n, int32_t c, int64_t s, int32_t chunks, void* stream);
int vnqa_c3d_stats_rows(const void* x, float* partial, int64_t rows, int32_t c, int32_t dtype, void* stream);
int vnqa_bn_finalize(const float* partial, int32_t nblk, int32_t c, double count, float eps, float momentum, float* mean,
                     float* rstd, float* running_mean, float* running_var, void* stream);
int vnqa_bn_rows_apply(const void* x, int32_t x_dtype, void* out, int32_t out_dtype, const vnqa_view5* out_view, const float* mean,
                       const float* rstd, const float* gamma, const float* beta, int64_t rows, int32_t c, void* stream);
int vnqa_bn_rows_bwd(const void* dy, int32_t dy_dtype, const vnqa_view5* dy_view, const void* x, int32_t x_dtype, void* dx,
                     int32_t dx_dtype, const float* mean, const float* rstd, const float* gamma, float* dgamma, float* dbeta,
                     float* workspace, float grad_scale, int64_t rows, int32_t c, void* stream);
int32_t vnqa_pool444_blocks(int32_t n, int32_t d, int32_t h, int32_t w, int32_t c);
int vnqa_pool444_fwd(const void* y, void* p, uint8_t* idx, float* partial, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                     void* stream);
int vnqa_pool444_bwd(const void* dp, const uint8_t* idx, void* dy, int32_t n, int32_t d, int32_t h, int32_t w, int32_t c,
                     void* stream);
int vnqa_c3d_conv1_supported(int32_t n, int32_t d, int32_t h, int32_t w);
int32_t vnqa_c3d_conv1_fwd_blocks(int32_t n, int32_t h, int32_t w);
int32_t vnqa_c3d_conv1_bwd_blocks(int32_t n, int32_t h, int32_t w);
int vnqa_c3d_conv1_fwd(const float* x, const float* weight, const float* bias, const float* mean, const float* rstd,
                       const float* gamma, const float* beta, void* p, uint8_t* idx, float* partial, int32_t n, int32_t d,
                       int32_t h, int32_t w, void* stream);
int vnqa_c3d_conv1_bwd(const float* x, const float* weight, const float* mean, const float* rstd, const float* gamma,
                       const float* beta, const void* dp, const uint8_t* idx, float* partial, float grad_scale, float* dweight,
                       float* dbias, float* dgamma, float* dbeta, int32_t n, int32_t d, int32_t h, int32_t w, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VNQA_HIP_H_ */
