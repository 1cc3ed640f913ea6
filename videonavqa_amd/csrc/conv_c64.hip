// conv_c64.hip — persistent direct 3x3 convolution for C_in = 64 layers (bf16) on MFMA.
//
// The VGG front's conv1_2 (64->64 @HxW) and conv2_1 (64->128 @H/2xW/2) have K = 9*64 = 576: only 9
// stages of the generic implicit GEMM, so its per-workgroup prologue/epilogue and DMA latency dominate.
// Here a workgroup is PERSISTENT: it DMA-loads its 64 output channels' complete weight set once
// (576 rows x 128 B = 72 KiB, resident in LDS for the whole launch) and then walks 16x16-pixel tiles.
// Per tile it needs just the 18x18 halo patch (324 rows x 128 B = 40.5 KiB), fetched by LDS-DMA into
// one of two patch slots while the previous tile is being computed; the 9 taps are plain address shifts
// inside the patch, so the 36 MFMA k-steps of a tile run without a single barrier.
//   D[cout][pixel] += W[tap][cout][:] . patch[pixel + tap][:]     (weights = MFMA A operand)
// Epilogue: bias + ReLU in registers, through the (now free) patch slot for 2x2 max-pool / affine and
// 16-byte NHWC stores.  C_out > 64 is handled by giving each workgroup one 64-channel slice.
#include "vnqa_common.h"

namespace {

constexpr int TS = 16;                 // tile side (pixels)
constexpr int PS = TS + 2;             // patch side
constexpr int PROWS = PS * PS;         // 324 patch rows of 128 B
constexpr int W_BYTES = 576 * 128;     // 73728
constexpr int P_INSTR = (PROWS + 7) / 8;            // 41 wave-level DMA instructions per patch
constexpr int P_BYTES = P_INSTR * 1024;             // 41984
constexpr int LDS_BYTES = W_BYTES + 2 * P_BYTES;    // 157696
constexpr int CROW = 64 * 2 + 16;                   // epilogue row stride

struct C64Args {
  const char* x;
  const char* wt;      // [Cout][9][64] bf16
  const float* bias;
  const float* post_scale;
  const float* post_shift;
  char* y;
  int n_img, H, W, Hp, Wp;
  int Cout, Cy;
  int relu, pool;
  int tilesX, tilesY, nsplit;   // nsplit = Cout / 64
  long long n_work;             // n_img * tilesY * tilesX * nsplit
  int Hyp, Wyp;
};

__device__ __forceinline__ void glds16c(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz128(int row) { return (row >> 1) & 7; }

// NW = 8: wave tile 64 px x 32 couts (2 waves per SIMD); NW = 4: wave tile 64 px x 64 couts (1 wave per SIMD,
// 8 fragment reads per 16 MFMAs instead of 6 per 8: the 8-wave shape runs close to the LDS read bandwidth).
template <int NW>
__global__ void __launch_bounds__(NW * 64) conv_c64_kernel(const C64Args p) {
  constexpr int TN = NW == 8 ? 2 : 4;              // 16-cout fragments per wave
  constexpr int NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsW = smem;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = NW == 8 ? (wave >> 1) : wave, wn = NW == 8 ? (wave & 1) : 0;
  const int fr = lane & 15, fh = lane >> 4;     // v_mfma_f32_16x16x32_bf16: row/col = lane&15, k-chunk = lane>>4

  // work items: this workgroup keeps ONE cout slice for its whole life (weights loaded once)
  const int G = gridDim.x;
  const int nsl = blockIdx.x % p.nsplit;
  const long long tiles_total = p.n_work / p.nsplit;                 // spatial tiles
  const int gslot = blockIdx.x / p.nsplit, gstride = G / p.nsplit;   // host guarantees G % nsplit == 0

  // ---- resident weights: LDS row (tap*64 + c) <- wt[(nsl*64 + c)][tap][0..63] ----
  {
    const size_t wbase = (size_t)nsl * 64 * 9 * 128;
#pragma unroll
    for (int j = 0; j < 72 / NW; ++j) {
      const int q = wave * (72 / NW) + j;          // 72 instructions, 8 rows each
      const int row = q * 8 + (lane >> 3);         // tap*64 + c
      const int tap = row >> 6, c = row & 63;
      const int lc = (lane & 7) ^ swz128(row);
      glds16c(p.wt + wbase + ((size_t)c * 9 + tap) * 128 + lc * 16, ldsW + q * 1024);
    }
  }

  auto tile_coords = [&](long long t, int& n, int& y0, int& x0) {
    const int per_img = p.tilesX * p.tilesY;
    n = (int)(t / per_img);
    const int r = (int)(t - (long long)n * per_img);
    const int ty = r / p.tilesX;
    y0 = ty * TS;
    x0 = (r - ty * p.tilesX) * TS;
  };

  // number of patch DMA instructions this wave issues (41 spread over NW waves: wave 0 has one more)
  constexpr int PI = 40 / NW;
  const int my_pinstr = wave == 0 ? PI + 1 : PI;
  auto issue_patch = [&](long long t, int slot) {
    int n, y0, x0;
    tile_coords(t, n, y0, x0);
    char* lds = smem + W_BYTES + slot * P_BYTES;
    for (int j = 0; j < my_pinstr; ++j) {
      const int q = wave + NW * j;
      int row = q * 8 + (lane >> 3);
      const int lrow = row;
      row = row < PROWS ? row : PROWS - 1;
      const int py = row / PS, px = row - py * PS;
      int gy = y0 + py, gx = x0 + px;              // padded coordinates (patch origin = tile origin - 1)
      gy = gy < p.Hp ? gy : p.Hp - 1;
      gx = gx < p.Wp ? gx : p.Wp - 1;
      const int lc = (lane & 7) ^ swz128(lrow);
      glds16c(p.x + (((size_t)n * p.Hp + gy) * p.Wp + gx) * 128 + lc * 16, lds + q * 1024);
    }
  };

  // ---- fragment geometry ----
  // pixel fragment i of this wave = tile row ty = 4*wm + i, tx = fr; cout fragment j = couts wn*32 + 16j + (0..15)
  int prow0[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) prow0[i] = (4 * wm + i) * PS + fr;
  int b_rd[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) b_rd[j] = (wn * 32 + j * 16 + fr) * 128;
  const int b_sw = swz128(fr);       // (wn*32 + 16j) is a multiple of 16: the swizzle only sees fr

  // bias of this lane's 8 output channels: cout = nsl*64 + wn*32 + 16j + 4fh + e
  float bias_r[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e)
      bias_r[j][e] = p.bias ? p.bias[nsl * 64 + wn * 32 + 16 * j + 4 * fh + e] : 0.f;

  long long t_cur = gslot;
  if (t_cur < tiles_total) issue_patch(t_cur, 0);
  if (t_cur + gstride < tiles_total) issue_patch(t_cur + gstride, 1);

  int it = 0;
  for (; t_cur < tiles_total; t_cur += gstride, ++it) {
    const int slot = it & 1;
    const bool have_next = t_cur + gstride < tiles_total;
    // everything but the youngest patch (tile it+1) must have landed: weights, this tile's patch
    if (have_next) {
      if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PI + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PI) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    const char* ldsP = smem + W_BYTES + slot * P_BYTES;
    vnqa_f32x4 acc[4][TN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

    // 18 k-steps (tap-major, 2 x 32 channels per tap), fragments double-buffered in registers: the
    // LDS reads of step k+1 are issued before the MFMAs of step k
    auto load_step = [&](int k, vnqa_f32x4* wf, vnqa_f32x4* xf) {
      const int tap = k >> 1, s = k & 1;
      const int toff = (tap / 3) * PS + (tap % 3);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        wf[j] = *(const vnqa_f32x4*)(ldsW + tap * 8192 + b_rd[j] + (((4 * s + fh) ^ b_sw) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int pr = prow0[i] + toff;
        xf[i] = *(const vnqa_f32x4*)(ldsP + pr * 128 + (((4 * s + fh) ^ swz128(pr)) << 4));
      }
    };
    vnqa_f32x4 wfb[2][TN], xfb[2][4];
    load_step(0, wfb[0], xfb[0]);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int k = 0; k < 18; ++k) {
      if (k + 1 < 18) load_step(k + 1, wfb[(k + 1) & 1], xfb[(k + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(vnqa_bf16x8, wfb[k & 1][j]),
                                                             __builtin_bit_cast(vnqa_bf16x8, xfb[k & 1][i]), acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_s_barrier();          // every wave is done reading this patch slot

    // ---- epilogue: C tile [m][64 couts] in the patch slot; m is quad-major when pooling ----
    char* ldsC = smem + W_BYTES + slot * P_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ty = 4 * wm + i, tx = fr;
      const int m = p.pool ? ((((ty >> 1) * 8 + (tx >> 1)) << 2) + ((ty & 1) << 1) + (tx & 1)) : (ty * TS + tx);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[i][j][e] + bias_r[j][e];
          if (p.relu) v[e] = fmaxf(v[e], 0.f);
        }
        uint2 pk;
        pk.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
        pk.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
        *(uint2*)(ldsC + m * CROW + (wn * 32 + 16 * j + 4 * fh) * 2) = pk;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int n, y0, x0;
    tile_coords(t_cur, n, y0, x0);
    const bool has_post = p.post_scale != nullptr;
    const int rows_out = p.pool ? 64 : 256;
    const int side = p.pool ? 8 : 16;
    const int Ho = p.pool ? (p.H >> 1) : p.H, Wo = p.pool ? (p.W >> 1) : p.W;
    const int oy0 = p.pool ? (y0 >> 1) : y0, ox0 = p.pool ? (x0 >> 1) : x0;
    for (int idx = threadIdx.x; idx < rows_out * 8; idx += NT) {
      const int orow = idx >> 3, c = idx & 7;
      const int oy = oy0 + orow / side, ox = ox0 + orow % side;
      if (oy >= Ho || ox >= Wo) continue;
      float v[8];
      if (p.pool) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = -INFINITY;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const uint4 u = *(const uint4*)(ldsC + (orow * 4 + d) * CROW + c * 16);
          const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] = fmaxf(v[2 * e], __uint_as_float(w4[e] << 16));
            v[2 * e + 1] = fmaxf(v[2 * e + 1], __uint_as_float(w4[e] & 0xffff0000u));
          }
        }
      } else {
        const uint4 u = *(const uint4*)(ldsC + orow * CROW + c * 16);
        const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = __uint_as_float(w4[e] << 16);
          v[2 * e + 1] = __uint_as_float(w4[e] & 0xffff0000u);
        }
      }
      const int co0 = nsl * 64 + c * 8;
      if (has_post) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * p.post_scale[co0 + e] + p.post_shift[co0 + e];
      }
      uint4 o;
      o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
      o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
      o.z = (unsigned)f32_to_bf16(v[4]) | ((unsigned)f32_to_bf16(v[5]) << 16);
      o.w = (unsigned)f32_to_bf16(v[6]) | ((unsigned)f32_to_bf16(v[7]) << 16);
      unsigned short* dst = (unsigned short*)p.y +
                            (((size_t)n * p.Hyp + oy + 1) * p.Wyp + ox + 1) * (size_t)p.Cy + co0;
      *(uint4*)dst = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // C tile consumed: the slot may be refilled
    if (t_cur + 2 * (long long)gstride < tiles_total) issue_patch(t_cur + 2 * (long long)gstride, slot);
  }
}

}  // namespace

// 3x3 'same' conv for c_in == 64, bf16, fused bias/ReLU/pool2/affine; x and y padded NHWC with halo 1.
extern "C" int vnqa_conv2d_c64_fwd(const vnqa_conv_desc* d, const void* x, const void* wt, const float* bias,
                                   const float* post_scale, const float* post_shift, void* y, void* stream) {
  VNQA_CHECK_ARG(d && x && wt && y, "conv2d_c64_fwd: null pointer");
  VNQA_CHECK_ARG(d->dtype == VNQA_BF16, "conv2d_c64_fwd: bf16 only");
  VNQA_CHECK_ARG(d->c_in == 64 && d->taps == 9 && d->x_halo == 1 && d->y_halo == 1,
                 "conv2d_c64_fwd: needs c_in == 64, taps == 9, halos == 1");
  VNQA_CHECK_ARG(d->c_out % 64 == 0 && d->c_out >= 64 && d->c_y >= d->c_out && d->c_y % 8 == 0,
                 "conv2d_c64_fwd: c_out must be a multiple of 64 (got %d)", d->c_out);
  VNQA_CHECK_ARG(!d->pool2 || (d->h % 2 == 0 && d->w % 2 == 0), "conv2d_c64_fwd: pool2 needs even h,w");
  VNQA_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "conv2d_c64_fwd: post_scale/post_shift must come together");
  C64Args a;
  a.x = (const char*)x;
  a.wt = (const char*)wt;
  a.bias = bias;
  a.post_scale = post_scale;
  a.post_shift = post_shift;
  a.y = (char*)y;
  a.n_img = d->n_img;
  a.H = d->h;
  a.W = d->w;
  a.Hp = d->h + 2;
  a.Wp = d->w + 2;
  a.Cout = d->c_out;
  a.Cy = d->c_y;
  a.relu = d->relu;
  a.pool = d->pool2;
  a.tilesX = (d->w + TS - 1) / TS;
  a.tilesY = (d->h + TS - 1) / TS;
  a.nsplit = d->c_out / 64;
  a.n_work = (long long)d->n_img * a.tilesX * a.tilesY * a.nsplit;
  const int ho = d->pool2 ? d->h / 2 : d->h, wo = d->pool2 ? d->w / 2 : d->w;
  a.Hyp = ho + 2;
  a.Wyp = wo + 2;
  const bool four = d->tile == 2;   // tile == 2 selects the 4-wave shape (A/B runs: 15 % slower); default = 8 waves
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)conv_c64_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void*)conv_c64_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) {
      vnqa_set_error("conv2d_c64_fwd: cannot reserve %d B of LDS", LDS_BYTES);
      return VNQA_ERR_HIP;
    }
    attr_set = true;
  }
  long long grid = 256;                       // one persistent workgroup per CU
  grid = grid / a.nsplit * a.nsplit;
  if (grid > a.n_work) grid = (a.n_work / a.nsplit) * a.nsplit;
  if (grid < a.nsplit) grid = a.nsplit;
  if (four)
    hipLaunchKernelGGL(conv_c64_kernel<4>, dim3((int)grid), dim3(256), LDS_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(conv_c64_kernel<8>, dim3((int)grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}
