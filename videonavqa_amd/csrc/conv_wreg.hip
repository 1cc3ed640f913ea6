// conv_wreg.hip — weights-stationary-in-REGISTERS persistent direct 3x3 convolution (bf16/fp16 storage) on MFMA.
//
// The short-K layers of the VGG front (conv1_2 64->64, conv2_1 64->128, conv2_2 128->128: K = 576 / 1152) spend more
// time around their main loop than in it when they run as implicit GEMMs (9-18 K-steps per tile) or with LDS-resident
// weights (both MFMA operands through LDS, prologue / epilogue phases serialized by workgroup barriers).  MI355X has a
// 512 KiB vector register file per CU (4 SIMDs x 512 registers x 64 lanes x 4 B) — more than three times its LDS — and
// these layers' whole weight tensors are 72-288 KiB.  Here a workgroup is 4 waves, ONE per SIMD, each owning all 512
// registers of its SIMD:
//   * every wave keeps its slice of the weights, already in MFMA A-operand form, in 288 registers for the whole
//     launch (conv2_2: 32 couts x K 1152 per wave; conv2_1: 64 couts x K 576, two cout halves x two pixel halves;
//     conv1_2: all 64 couts x K 576, four pixel quarters) — no weight DMA, no weight LDS reads, ever again;
//   * the only LDS traffic is the activation patch: a (TH+2) x 18 pixel halo patch per TH x 16 pixel tile, fetched by
//     LDS-DMA into one of two slots a whole tile ahead; the nine taps are immediate address offsets into it;
//   * one B-fragment read (ds_read_b128) feeds NJ = 2..4 MFMAs (v_mfma_f32_16x16x32), 0.25-0.5 LDS reads per MFMA;
//   * the epilogue (bias, ReLU, 2x2 max-pool, affine, 16-byte NHWC stores) works from the accumulators in registers:
//     no LDS staging and no barrier, so the compiler interleaves it with the next half tile's MFMAs;
//   * ONE workgroup barrier per tile.
// D[cout][pixel] += W[tap][cout][:] . patch[pixel + tap][:]        (weights = MFMA A operand, pixels = B operand)
//
// Patch swizzle (conflict-free ds_read_b128 for every tap shift under gfx950's non-contiguous 16-lane groups, exhaustive
// check in tools/lds_swizzle_check.py): 16-byte chunk index ^= (px & 6) for 128-byte pixels (C_in = 64) and
// ^= 2 (px & 7) for 256-byte pixels (C_in = 128), px = patch column — a function of the COLUMN only, so that the row
// part of every fragment address is an instruction immediate.
//
// Reference semantics: torch.nn.Conv2d(k=3, padding=1) + ReLU (+ MaxPool2d(2)) of the VGG-16 front
// (get_frcnn_feature_extractor, call sites eval/q_and_v_eval.py:106) with ObjDetectCNN's eval-mode bn_input affine
// folded into the producer's epilogue (models/obj_detector.py:70).
#include <cstdlib>
#include "vnqa_common.h"

namespace {

struct WregArgs {
  const char* x;       // padded NHWC [n][H+2][W+2][CIN], zero halo
  const char* wt;      // [COUT][9][CIN] (K-major pack, vnqa_pack_conv_weight)
  const float* bias;
  const float* post_scale;
  const float* post_shift;
  char* y;             // padded NHWC [n][Ho+2yh][Wo+2yh][Cy]
  int n_img, H, W, Hp, Wp;
  int Cy, relu;
  int tilesX, tilesY;
  int n_tiles;
  int Hyp, Wyp, y_halo;
};

// LDS-DMA issued from inline asm: hipcc does not see it, so it neither drains it with a vmcnt(0) in front of unrelated LDS
// reads (cdna_hip_programming.md §5, "Three .s-level traps" (a)) nor counts it: every wait for it below is hand-placed.
// sbase / lds_addr are wave-uniform; voff is the lane's byte offset.  M0 (the DMA's LDS base) is compiler-reserved: saved
// and restored inside the statement.
__device__ __forceinline__ void glds16_asm(const char* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

template <int CIN> __device__ __forceinline__ int wreg_swz(int px) { return CIN == 64 ? (px & 6) : 2 * (px & 7); }

template <int CIN, int COUT, int NCG, int TH, int HR, bool POOL>
__global__ void __launch_bounds__(256, 1) conv_wreg_kernel(const WregArgs p) {
  constexpr int NPG = 4 / NCG;                 // pixel groups (waves that share a cout slice split the tile rows)
  constexpr int RW = TH / NPG, NP = RW / HR;   // conv rows per wave; a wave works through them in NP parts of HR rows
  constexpr int CPW = COUT / NCG, NJ = CPW / 16, KS = CIN / 32;
  constexpr int PIXB = CIN * 2, CPP = PIXB / 16, PPI = 1024 / PIXB;
  constexpr int PW = 18, PH = TH + 2, NPIX = PH * PW;
  constexpr int NI = (NPIX * PIXB + 1023) / 1024, SLOT = NI * 1024;
  constexpr int NGRP = 3 * KS;                 // fragment groups (column shift s, k-step ks) per part
  constexpr int NR = HR + 2;                   // patch rows a part touches
  static_assert(NJ * 9 * KS == 72, "a wave's weight slice must be 72 fragments (288 registers)");
  static_assert(RW % HR == 0 && NP % 2 == 0 && (!POOL || HR % 2 == 0), "parts must hold whole pooling windows");
  static_assert(POOL ? ((HR / 2) * NJ == 4 || (HR / 2) * NJ == 2) : (NJ % 2 == 0), "epilogue store grouping");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const ldsPar = (float*)(smem + 2 * SLOT);     // bias[COUT], post_scale[COUT], post_shift[COUT]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int fr = lane & 15, fh = lane >> 4;
  const int cg = wave % NCG, pg = wave / NCG;
  const int cout0 = cg * CPW;

  if (threadIdx.x < COUT) {
    ldsPar[threadIdx.x] = p.bias ? p.bias[threadIdx.x] : 0.f;
    ldsPar[COUT + threadIdx.x] = p.post_scale ? p.post_scale[threadIdx.x] : 1.f;
    ldsPar[2 * COUT + threadIdx.x] = p.post_shift ? p.post_shift[threadIdx.x] : 0.f;
  }

  // ---- this wave's weights: A fragments (cout row fr, k-chunk fh) of all 9 taps x KS k-steps x NJ cout blocks ----
  vnqa_bf16x8 Wf[9][KS][NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        Wf[tap][ks][j] = *(const vnqa_bf16x8*)(p.wt + (((size_t)(cout0 + 16 * j + fr) * 9 + tap) * CIN + ks * 32 + fh * 8) * 2);

  // ---- tile walk: the 32 workgroups that share an XCD (blockIdx % 8) take 32 consecutive tiles per round ----
  const int G = gridDim.x;
  const int per_round_base = (G & 7) == 0 ? (blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3) : blockIdx.x;
  const int per_img = p.tilesX * p.tilesY;
  auto tile_origin = [&](int t, int& n, int& y0, int& x0) {
    n = t / per_img;
    const int r = t - n * per_img;
    const int ty = r / p.tilesX;
    y0 = ty * TH;
    x0 = (r - ty * p.tilesX) * 16;
  };

  // patch DMA: instruction q writes LDS bytes [q*1024, q*1024+1024) of the slot = PPI consecutive patch pixels
  auto issue_patch = [&](int t, int slot) {
    int n, y0, x0;
    tile_origin(t, n, y0, x0);
    const char* src0 = p.x + (((size_t)n * p.Hp + y0) * p.Wp + x0) * PIXB;
    asm volatile("s_nop 4" ::: "memory");   // (SGPR operands below may come fresh from VALU lane reads)
#pragma unroll
    for (int k = 0; k < (NI + 3) / 4; ++k) {
      const int q = wave + 4 * k;
      if (q < NI) {      // wave-uniform
        int pix = q * PPI + lane / CPP;
        pix = pix < NPIX ? pix : NPIX - 1;
        const int py = pix / PW, px = pix - py * PW;
        const int sc = (lane % CPP) ^ wreg_swz<CIN>(px);
        glds16_asm(src0, (unsigned)((py * p.Wp + px) * PIXB + sc * 16), __builtin_amdgcn_readfirstlane(lds0 + slot * SLOT + q * 1024));
      }
    }
  };

  // per-lane fragment bases: pixel column fr + s, k-chunk 4 ks + fh (swizzled); the row offset is an immediate
  int fbase[3][KS];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      fbase[s][ks] = (fr + s) * PIXB + (((4 * ks + fh) ^ wreg_swz<CIN>(fr + s)) << 4);

  int t = per_round_base;
  if (t < p.n_tiles) issue_patch(t, 0);

  // 16-byte stores per lane and tile (every lane executes every store: tiles are always whole)
  constexpr int NST = NP * (POOL ? 1 : HR * NJ / 2);
  const int hp = fh & 1;

  for (int it = 0; t < p.n_tiles; t += G, ++it) {
    const int slot = it & 1;
    // This tile's patch was issued a whole tile ago (before that tile's NST stores): all but those stores must be done.
    // First tile: everything (the patch, and the parameter table written above).
    if (it > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // patch visible to all waves; everyone is done reading the other slot
    if (t + G < p.n_tiles) issue_patch(t + G, slot ^ 1);

    int n, y0, x0;
    tile_origin(t, n, y0, x0);
    const char* const ldsP = smem + slot * SLOT + (pg * RW) * PW * PIXB;

    vnqa_f32x4 acc[2][HR][NJ];       // parts alternate between the two sets: the epilogue of part k runs under part k + 1's MFMAs
    // Step k = (half h, group g = (s, ks)).  The NR patch-row fragments of a group are read ONCE and serve all three tap
    // rows r (output row i = R - r): NR reads per 3 HR NJ MFMAs.  The reads of step k + 1 are issued in the middle of step
    // k's MFMAs (register double buffer Xa / Xb), so that the wait in front of a step never covers reads younger than its own.
    auto load_step = [&](int k, vnqa_bf16x8* X) {
      const int h = k / NGRP, g = k - h * NGRP;
      const int s = g / KS, ks = g - s * KS;
#pragma unroll
      for (int R = 0; R < NR; ++R) X[R] = *(const vnqa_bf16x8*)(ldsP + fbase[s][ks] + (h * HR + R) * PW * PIXB);
    };
    auto mma_rows = [&](int k, const vnqa_bf16x8* X, int R0, int R1) {
      const int h = k / NGRP, g = k - h * NGRP;
      const int s = g / KS, ks = g - s * KS;
#pragma unroll
      for (int R = R0; R < R1; ++R)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int i = R - r;
          if (i < 0 || i >= HR) continue;
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            if (g == 0 && r == 0) acc[h & 1][i][j] = vnqa_f32x4{0.f, 0.f, 0.f, 0.f};      // first product of this accumulator
            acc[h & 1][i][j] = VNQA_MFMA_16x16x32(Wf[3 * r + s][ks][j], X[R], acc[h & 1][i][j]);
          }
        }
    };
    // ---- epilogue of half tile h, from registers: lane holds couts cout0 + 16 j + 4 fh + e of pixel column fr ----
    auto epilogue = [&](int h) {
      if constexpr (POOL) {
        constexpr int HRP = HR / 2;
        uint2 P[HRP][NJ];
#pragma unroll
        for (int pr = 0; pr < HRP; ++pr)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int co = cout0 + 16 * j + 4 * fh;
            const float4 bb = *(const float4*)(ldsPar + co);
            const float4 sc = *(const float4*)(ldsPar + COUT + co);
            const float4 sh = *(const float4*)(ldsPar + 2 * COUT + co);
            const float b4[4] = {bb.x, bb.y, bb.z, bb.w}, s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float u = fmaxf(acc[h & 1][2 * pr][j][e], acc[h & 1][2 * pr + 1][j][e]) + b4[e];      // rows 2pr, 2pr+1 (bias commutes with max)
              if (p.relu) u = fmaxf(u, 0.f);
              // columns fr and fr ^ 1: DPP quad_perm [1,0,3,2]
              u = fmaxf(u, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u), 0xB1, 0xF, 0xF, true)));
              // affine on the storage-rounded value, as the LDS-staged kernels do (scale 1 / shift 0 = identity)
              v[e] = p.post_scale ? bf16_to_f32(f32_to_bf16(u)) * s4[e] + h4[e] : u;
            }
            P[pr][j].x = pack2_h16(v[0], v[1]);
            P[pr][j].y = pack2_h16(v[2], v[3]);
          }
        // 4 (pooled row, cout block) combinations per lane quad (fr parity q, fh parity hp): q picks the pair, the fh-pair
        // lanes trade one 8-byte group (v_permlane16_swap) so that each lane owns 8 consecutive couts of ONE combination
        const int q = fr & 1;
        uint2 lo, hi;
        int pr_st, j_lo, j_hi;
        bool st = true;
        if constexpr (NJ == 2 && HRP == 2) {        // q = pooled row, hp = cout block
          lo = q ? P[HRP - 1][0] : P[0][0];
          hi = q ? P[HRP - 1][NJ - 1] : P[0][NJ - 1];
          pr_st = q; j_lo = 0; j_hi = 1;
        } else if constexpr (NJ == 2) {             // HRP == 1: two combinations only — the odd column of a pair does not store
          lo = P[0][0];
          hi = P[0][NJ - 1];
          pr_st = 0; j_lo = 0; j_hi = 1;
          st = q == 0;
        } else {                                    // NJ == 4, HRP == 1: block = 2 hp + q
          lo = q ? P[0][1] : P[0][0];
          hi = q ? P[0][NJ - 1] : P[0][NJ - 2];
          pr_st = 0; j_lo = q; j_hi = 2 + q;
        }
        const auto sx = __builtin_amdgcn_permlane16_swap(lo.x, hi.x, false, false);
        const auto sy = __builtin_amdgcn_permlane16_swap(lo.y, hi.y, false, false);
        const uint4 o = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        const int oy = ((y0 + pg * RW + h * HR) >> 1) + pr_st, ox = (x0 >> 1) + (fr >> 1);
        const int co = cout0 + 16 * (hp ? j_hi : j_lo) + 8 * (fh >> 1);
        unsigned short* dst = (unsigned short*)p.y +
                              (((size_t)n * p.Hyp + oy + p.y_halo) * p.Wyp + ox + p.y_halo) * (size_t)p.Cy + co;
        if (st) *(uint4*)dst = o;
      } else {
        // un-pooled: per conv row and pair of cout blocks (2m, 2m+1) one 16-byte store per lane
#pragma unroll
        for (int i = 0; i < HR; ++i) {
          uint2 P[NJ];
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int co = cout0 + 16 * j + 4 * fh;
            const float4 bb = *(const float4*)(ldsPar + co);
            const float4 sc = *(const float4*)(ldsPar + COUT + co);
            const float4 sh = *(const float4*)(ldsPar + 2 * COUT + co);
            const float b4[4] = {bb.x, bb.y, bb.z, bb.w}, s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float u = acc[h & 1][i][j][e] + b4[e];
              if (p.relu) u = fmaxf(u, 0.f);
              v[e] = p.post_scale ? bf16_to_f32(f32_to_bf16(u)) * s4[e] + h4[e] : u;
            }
            P[j].x = pack2_h16(v[0], v[1]);
            P[j].y = pack2_h16(v[2], v[3]);
          }
          const int oy = y0 + pg * RW + h * HR + i, ox = x0 + fr;
          unsigned short* dst0 = (unsigned short*)p.y +
                                 (((size_t)n * p.Hyp + oy + p.y_halo) * p.Wyp + ox + p.y_halo) * (size_t)p.Cy + cout0 + 8 * (fh >> 1);
#pragma unroll
          for (int m = 0; m < NJ / 2; ++m) {
            const auto sx = __builtin_amdgcn_permlane16_swap(P[2 * m].x, P[2 * m + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane16_swap(P[2 * m].y, P[2 * m + 1].y, false, false);
            *(uint4*)(dst0 + 16 * (2 * m + hp)) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
          }
        }
      }
    };

    constexpr int SPLIT = NR >= 6 ? 2 : 1;     // patch rows whose MFMAs run before the next step's reads are issued
    constexpr int NSTEP = NP * NGRP;
    static_assert(NSTEP % 2 == 0 && NGRP % 2 == 0, "steps come in Xa / Xb pairs");
    vnqa_bf16x8 Xa[NR], Xb[NR];
    load_step(0, Xa);
#pragma unroll
    for (int k = 0; k < NSTEP; k += 2) {
      // even step: fragments in Xa, next step's into Xb
      __builtin_amdgcn_sched_barrier(0);
      mma_rows(k, Xa, 0, SPLIT);
      __builtin_amdgcn_sched_barrier(0);
      load_step(k + 1, Xb);
      __builtin_amdgcn_sched_barrier(0);
      mma_rows(k, Xa, SPLIT, NR);
      if (k >= NGRP && k % NGRP == 0) epilogue(k / NGRP - 1);   // previous part is complete: its VALU-only epilogue shares
                                                                // a region with this part's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      // odd step
      mma_rows(k + 1, Xb, 0, SPLIT);
      __builtin_amdgcn_sched_barrier(0);
      if (k + 2 < NSTEP) load_step(k + 2, Xa);
      __builtin_amdgcn_sched_barrier(0);
      mma_rows(k + 1, Xb, SPLIT, NR);
    }
    __builtin_amdgcn_sched_barrier(0);
    epilogue(NP - 1);
  }
}

template <int CIN, int COUT, int NCG, int TH, int HR, bool POOL>
int wreg_launch(WregArgs a, hipStream_t stream) {
  constexpr int PIXB = CIN * 2;
  constexpr int NI = ((TH + 2) * 18 * PIXB + 1023) / 1024;
  constexpr int LDS = 2 * NI * 1024 + 3 * COUT * 4;
  static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
  a.tilesX = a.W / 16;
  a.tilesY = a.H / TH;
  a.n_tiles = a.n_img * a.tilesX * a.tilesY;
  auto kern = conv_wreg_kernel<CIN, COUT, NCG, TH, HR, POOL>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
      vnqa_set_error("conv2d_wreg_fwd: cannot reserve %d B of LDS", LDS);
      return VNQA_ERR_HIP;
    }
    attr_set = true;
  }
  static const int reserve = [] { const char* e = getenv("VNQA_PERSISTENT_RESERVE_CUS"); const int v = e ? atoi(e) : 0;
                                  return v < 0 ? 0 : (v > 128 ? 128 : v); }();
  int grid = (256 - reserve) & ~7;
  if (grid > a.n_tiles) grid = a.n_tiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, stream, a);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

}  // namespace

// 1 when vnqa_conv2d_wreg_fwd serves this geometry (the caller falls back to the igemm / c64 kernels otherwise)
extern "C" int vnqa_conv2d_wreg_supported(const vnqa_conv_desc* d) {
  if (!d || d->dtype != VNQA_BF16 || d->taps != 9 || d->x_halo != 1 || d->depth != 0) return 0;
  if (d->w % 16 != 0 || d->h % 8 != 0 || d->c_y % 8 != 0) return 0;
  if (d->c_in == 128 && d->c_out == 128 && d->pool2) return 1;
  if (d->c_in == 64 && d->c_out == 128 && !d->pool2) return 1;
  if (d->c_in == 64 && d->c_out == 64 && d->pool2 && d->h % 16 == 0) return 1;
  return 0;
}

// 3x3 'same' conv, weights stationary in registers; same tensors / epilogue contract as vnqa_conv2d_igemm_fwd
// (bias -> ReLU -> 2x2 max-pool -> per-channel affine), restricted to the geometries above.  y_halo may be 1 or 2.
extern "C" int vnqa_conv2d_wreg_fwd(const vnqa_conv_desc* d, const void* x, const void* wt, const float* bias,
                                    const float* post_scale, const float* post_shift, void* y, void* stream) {
  VNQA_CHECK_ARG(d && x && wt && y, "conv2d_wreg_fwd: null pointer");
  VNQA_CHECK_ARG(vnqa_conv2d_wreg_supported(d), "conv2d_wreg_fwd: unsupported geometry (c_in %d, c_out %d, %d x %d, pool %d)",
                 d->c_in, d->c_out, d->h, d->w, d->pool2);
  VNQA_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "conv2d_wreg_fwd: post_scale/post_shift must come together");
  VNQA_CHECK_ARG(d->c_y >= d->c_out && d->y_halo >= 1, "conv2d_wreg_fwd: bad output geometry");
  WregArgs a;
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
  a.Cy = d->c_y;
  a.relu = d->relu;
  const int ho = d->pool2 ? d->h / 2 : d->h, wo = d->pool2 ? d->w / 2 : d->w;
  a.y_halo = d->y_halo;
  a.Hyp = ho + 2 * d->y_halo;
  a.Wyp = wo + 2 * d->y_halo;
  hipStream_t st = (hipStream_t)stream;
  if (d->c_in == 128) return wreg_launch<128, 128, 4, 8, 2, true>(a, st);
  if (d->c_out == 128) return wreg_launch<64, 128, 2, 8, 2, false>(a, st);
  return wreg_launch<64, 64, 1, 16, 2, true>(a, st);
}
