// conv_patch.hip — 3x3 implicit-GEMM conv with the ACTIVATION PATCH resident in LDS across the 9 taps.
//
// conv_igemm.hip stages one (tap, 64-channel) slab of BOTH operands per K-step: 9 x (32 KiB pixels +
// 32 KiB weights) per channel chunk and tile, although the nine pixel slabs are the same ~300 pixel rows
// shifted by one column / one image row.  Measured there (PMC + a no-DMA diagnostic build): the chip is
// neither MFMA- nor bandwidth-bound but pays ~100-180 issue cycles per global_load_lds instruction, 8 per
// wave and K-step.  This kernel halves that count:
//   * a workgroup owns a 2-D tile of TR image rows x TC columns = 224 conv-output pixels (8x28 for
//     widths 28/56/112, 16x14 for the 14-wide trunk maps) and 256 output channels;
//   * per 64-channel chunk the (TR+2) x (TC+2) halo'd pixel patch (<= 360 rows x 128 B) is DMA'd ONCE into
//     one of two LDS patch buffers, a few instructions per K-step while the previous chunk is consumed;
//   * the 9 taps read their pixel fragments from that patch at shifted rows (ds_read_b128, same XOR swizzle,
//     keyed on the patch row), only the 32 KiB weight slab is DMA'd per tap.
// DMA instructions per wave and K-step: 4 (weights) + <=1 (patch)  instead of 8.
// Rows are taken from the global list of (image, y) rows, so tiles may straddle images: the patch is the
// contiguous range of PADDED rows between the first and last row's halos (the zero halo rows in between
// are simply part of it).
//
// MFMA: v_mfma_f32_16x16x32_bf16, weights as the A operand (4 consecutive couts per lane), 2x4 waves,
// wave tile 112 px x 64 cout (7 x 4 MFMA tiles), waves 4-7 staggered half a K-step behind waves 0-3
// as in conv_igemm.hip.  Epilogue identical in effect: bias, ReLU, 2x2 max-pool, per-channel affine,
// 16-byte NHWC stores through LDS.  bf16 only.
#include "conv_args.h"

namespace {

constexpr int BM = 224, BN = 256, WAVES_M = 2, WAVES_N = 4, NW = 8, NT = 512;
constexpr int MT = 16, WTM = BM / WAVES_M, WTN = BN / WAVES_N, TM = WTM / MT, TN = WTN / MT;   // 112, 64, 7, 4
constexpr int PATCH_ROWS = 360, PATCH_BYTES = PATCH_ROWS * 128, B_BYTES = BN * 128;
constexpr int PATCH_INSTR = PATCH_ROWS / 8;                 // 45 wave-level DMA instructions cover a patch
constexpr int PIW = (PATCH_INSTR + NW - 1) / NW;            // <= 6 of them per wave
constexpr int B_PER_WAVE = (BN / 8) / NW;                   // 4
constexpr int LDS_BYTES = 2 * PATCH_BYTES + 2 * B_BYTES;    // 157696
constexpr int CROW = BN * 2 + 16;
static_assert(BM * CROW <= LDS_BYTES, "epilogue tile must fit");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget exceeded");

// XOR swizzle of the PATCH rows (16-byte chunk ^= pswz(row)).  The tap shifts start a pixel fragment at any patch row, and a
// ds_read_b128 is served in non-contiguous 16-lane groups: `row & 6` is conflict-free for every start row (see conv_c64.hip,
// tools/lds_swizzle_check.py); the weight slab, whose fragments start at multiples of 16, keeps (row >> 1) & 7.
#ifdef VNQA_PATCH_OLD_SWIZZLE
__device__ __forceinline__ int pswz(int row) { return (row >> 1) & 7; }
#else
__device__ __forceinline__ int pswz(int row) { return row & 6; }
#endif

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void mma16(const vnqa_f32x4& a, const vnqa_f32x4& b, vnqa_f32x4& c) {
  c = VNQA_MFMA_16x16x32(__builtin_bit_cast(vnqa_bf16x8, a), __builtin_bit_cast(vnqa_bf16x8, b), c);
}

template <int TC, int TAG>
__global__ void __launch_bounds__(NT) conv_patch_kernel(const ConvArgs p) {
  constexpr int TR = BM / TC, PW = TC + 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const patch0 = smem;
  char* const bbuf0 = smem + 2 * PATCH_BYTES;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware bijective remap (as conv_igemm.hip): an XCD gets a contiguous run of tiles, n-tile fastest
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_n = t % p.tilesN;
    tile_m = t / p.tilesN;
  }
  const int CB = p.W / TC;
  const int rt = tile_m / CB, cb = tile_m - rt * CB;
  const int total_rows = p.n_img * p.H;
  const int g0 = rt * TR;
  auto padrow = [&](int g) {
    const int n = g / p.H;
    return n * p.Hp + (g - n * p.H) + 1;
  };
  const int g_last = min(g0 + TR - 1, total_rows - 1);
  const int pr_first = padrow(g0) - 1;
  const int n_lin = (padrow(g_last) + 1 - pr_first + 1) * PW;      // patch pixels (rows of 128 B) actually needed
  const int n_instr = (n_lin + 7) >> 3;

  const int kchunks = p.Cin >> 6;
  const int KT = 9 * kchunks;
  const size_t cin_b = (size_t)p.Cin * 2;

  // ---- per-lane DMA source offsets ----
  size_t a_off[PIW];                      // patch instruction q = wave + 8 j: LDS rows 8q .. 8q+7
#pragma unroll
  for (int j = 0; j < PIW; ++j) {
    int lin = (wave + NW * j) * 8 + (lane >> 3);
    lin = lin < n_lin ? lin : n_lin - 1;  // rows past the patch are never read; keep the address in bounds
    const int i = lin / PW, jj = lin - i * PW;
    const int chunk = (lane & 7) ^ pswz((wave + NW * j) * 8 + (lane >> 3));      // keyed on the LDS row (before the clamp)
    a_off[j] = ((size_t)(pr_first + i) * p.Wp + cb * TC + jj) * cin_b + (size_t)chunk * 16;
  }
  size_t b_off[B_PER_WAVE];
  const size_t w_row_bytes = (size_t)9 * cin_b;
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    const int row = (wave * B_PER_WAVE + j) * 8 + (lane >> 3);
    int co = tile_n * BN + row;
    co = co < p.Cout ? co : p.Cout - 1;
    b_off[j] = (size_t)co * w_row_bytes + (size_t)((lane & 7) ^ ((row >> 1) & 7)) * 16;
  }
  auto load_patch_piece = [&](int kc, int j) {      // j-th instruction of this wave, chunk kc
    if (wave + NW * j < n_instr)
      glds16(p.x + a_off[j] + (size_t)kc * 128, patch0 + (kc & 1) * PATCH_BYTES + (wave + NW * j) * 1024);
  };
  auto load_weights = [&](int kt) {
    const int kc = kt / 9, tap = kt - kc * 9;
    const size_t woff = ((size_t)tap * p.Cin + (size_t)kc * 64) * 2;
    char* dst = bbuf0 + (kt & 1) * B_BYTES;
#pragma unroll
    for (int j = 0; j < B_PER_WAVE; ++j) glds16(p.wt + b_off[j] + woff, dst + (wave * B_PER_WAVE + j) * 1024);
  };
  // DMA issued while K-step kt is consumed: weights of kt+1 and, during taps 0..PIW-1 of a chunk, one piece per
  // wave of the NEXT chunk's patch (its buffer was last read in the previous chunk)
#ifdef VNQA_DIAG_SKIP_DMA   // timing-only diagnostic build: drop one operand's DMA after the prologue
  const bool skipA = (p.relu & 256) != 0, skipB = (p.relu & 512) != 0;
#else
  constexpr bool skipA = false, skipB = false;
#endif
  auto prefetch = [&](int kt) {
    if (kt + 1 < KT && !skipB) load_weights(kt + 1);
    const int kc = kt / 9, tap = kt - kc * 9;
    if (tap < PIW && kc + 1 < kchunks && !skipA) {
#pragma unroll
      for (int j = 0; j < PIW; ++j)
        if (j == tap) load_patch_piece(kc + 1, j);
    }
  };

  // ---- fragment addressing ----
  const int fr = lane & 15, fh = lane >> 4;
  int x_lin[TM];                          // patch row of (pixel, tap (0,0))
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ml = wm * WTM + i * MT + fr;
    const int tr = ml / TC, tc = ml - tr * TC;
    const int g = min(g0 + tr, total_rows - 1);
    x_lin[i] = (padrow(g) - pr_first - 1) * PW + tc;
  }
  int w_rd[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WTN + j * MT + fr;
    w_rd[j] = row * 128 + ((fh ^ ((row >> 1) & 7)) << 4);      // k-substep 0; substep 1 = ^ 64
  }

  vnqa_f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // fragments of K-step kt, substep s (32 channels): pixels from the patch at the tap's shift, weights from the slab
  auto load_frags = [&](int kt, int s, vnqa_f32x4* xf, vnqa_f32x4* wf) {
    const int kc = kt / 9, tap = kt - kc * 9;
    const int r = tap / 3;
    const int tapoff = r * PW + (tap - 3 * r);
    const char* pb = patch0 + (kc & 1) * PATCH_BYTES;
    const char* bb = bbuf0 + (kt & 1) * B_BYTES;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int lin = x_lin[i] + tapoff;
      xf[i] = *(const vnqa_f32x4*)(pb + lin * 128 + (((4 * s + fh) ^ pswz(lin)) << 4));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) wf[j] = *(const vnqa_f32x4*)(bb + (w_rd[j] ^ (s << 6)));
  };
  auto mma_sub = [&](vnqa_f32x4* xf, vnqa_f32x4* wf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) mma16(wf[j], xf[i], acc[i][j]);
  };

  // ---- prologue: whole patch of chunk 0 + weights of K-step 0 ----
#pragma unroll
  for (int j = 0; j < PIW; ++j) load_patch_piece(0, j);
  load_weights(0);
  __syncthreads();

  vnqa_f32x4 xf0[TM], wf0[TN], xf1[TM], wf1[TN];
  if (wave < 4) {
    for (int kt = 0; kt < KT; ++kt) {
      prefetch(kt);
      load_frags(kt, 0, xf0, wf0);
      load_frags(kt, 1, xf1, wf1);
      __builtin_amdgcn_sched_barrier(0);
      mma_sub(xf0, wf0);
      mma_sub(xf1, wf1);
      __syncthreads();
    }
  } else {
    // half a K-step behind: the MFMAs of the previous step's second substep run while waves 0-3 wait on LDS
    for (int kt = 0; kt < KT; ++kt) {
      prefetch(kt);
      if (kt > 0) mma_sub(xf1, wf1);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(kt, 0, xf0, wf0);
      __builtin_amdgcn_sched_barrier(0);
      mma_sub(xf0, wf0);
      __builtin_amdgcn_sched_barrier(0);
      load_frags(kt, 1, xf1, wf1);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
    mma_sub(xf1, wf1);
  }
  __syncthreads();   // (waves 4-7 finished their last register-only MFMAs; LDS is free for the epilogue)

  // ---------------- epilogue ----------------
  // acc[i][j][e]: pixel = wm*WTM + i*16 + fr ; cout = wn*WTN + j*16 + 4*fh + e
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = wn * WTN + j * MT + 4 * fh;
    const int co = tile_n * BN + col;
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) b4[e] = (co + e < p.Cout) ? p.bias[co + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int prow = wm * WTM + i * MT + fr;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = acc[i][j][e] + b4[e];
        if (p.relu) v[e] = fmaxf(v[e], 0.f);
      }
      uint2 pk;
      pk.x = pack2_h16(v[0], v[1]);
      pk.y = pack2_h16(v[2], v[3]);
      *(uint2*)(smem + prow * CROW + col * 2) = pk;
    }
  }
  __syncthreads();

  constexpr int CH = BN * 2 / 16;       // 16-byte chunks per tile row
  const bool has_post = (p.post_scale != nullptr);
  const int rows_out = p.pool ? BM / 4 : BM;
  const int OC = p.pool ? TC / 2 : TC;  // output columns per tile row
  for (int idx = threadIdx.x; idx < rows_out * CH; idx += NT) {
    const int orow = idx / CH, c = idx - orow * CH;
    const int co0 = tile_n * BN + c * 8;
    const int orr = orow / OC, occ = orow - orr * OC;
    const int g = g0 + (p.pool ? 2 * orr : orr);
    if (g >= total_rows || co0 >= p.Cout) continue;
    float v[8];
    if (p.pool) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = -INFINITY;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int ml = (2 * orr + (d >> 1)) * TC + 2 * occ + (d & 1);
        const vnqa_bf16* src = (const vnqa_bf16*)(smem + ml * CROW + c * 16);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], bf16_to_f32(src[e]));
      }
    } else {
      const vnqa_bf16* src = (const vnqa_bf16*)(smem + orow * CROW + c * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = bf16_to_f32(src[e]);
    }
    if (has_post) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * p.post_scale[co0 + e] + p.post_shift[co0 + e];
    }
    const int n = g / p.H;
    const int y = g - n * p.H;
    const int yo = p.pool ? (y >> 1) : y;
    const int xo = (p.pool ? (cb * TC) >> 1 : cb * TC) + occ;
    vnqa_bf16* dst = (vnqa_bf16*)(p.y) + (((size_t)n * p.Hyp + yo + p.y_halo) * p.Wyp + xo + p.y_halo) * (size_t)p.Cy + co0;
    vnqa_bf16 out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = f32_to_bf16(v[e]);
    *(uint4*)dst = *(const uint4*)out;
  }
}

template <int TC, int TAG>
int launch_patch(const ConvArgs& a, hipStream_t stream) {
  constexpr int TR = BM / TC;
  ConvArgs p = a;
  const int rows = p.n_img * p.H;
  const int tilesM = ((rows + TR - 1) / TR) * (p.W / TC);
  p.tilesN = (p.Cout + BN - 1) / BN;
  auto kern = conv_patch_kernel<TC, TAG>;
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: a race only repeats it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) {
      vnqa_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", LDS_BYTES, hipGetErrorString(e));
      return VNQA_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tilesM * p.tilesN), dim3(NT), LDS_BYTES, stream, p);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

}  // namespace

int vnqa_conv_patch_dispatch(const ConvArgs& a, int tag, hipStream_t st) {
  if (a.taps != 9 || a.D != 0 || a.x_halo != 1 || a.wt_tiled || a.partial != nullptr || a.Cin % 64 != 0) {
    vnqa_set_error("conv patch tile: needs bf16 3x3 2-D conv, x_halo 1, c_in %% 64 == 0, K-major weights, no split-K");
    return VNQA_ERR_UNSUPPORTED;
  }
  const int tc = a.W % 28 == 0 ? 28 : (a.W % 14 == 0 ? 14 : 0);
  if (tc == 0) {
    vnqa_set_error("conv patch tile: width %d is not a multiple of 14", a.W);
    return VNQA_ERR_UNSUPPORTED;
  }
  const int tr = BM / tc;
  // patch rows needed at worst: tile rows + 2 halo rows + 2 per image boundary the tile can straddle
  const int max_cross = (tr - 1 + a.H - 1) / a.H;
  if ((tr + 2 + 2 * max_cross) * (tc + 2) > PATCH_ROWS || (a.pool && (a.H % 2 != 0 || a.W % 2 != 0))) {
    vnqa_set_error("conv patch tile: %dx%d images do not fit the %d-row LDS patch", a.H, a.W, PATCH_ROWS);
    return VNQA_ERR_UNSUPPORTED;
  }
  if (tc == 28) return tag ? launch_patch<28, 1>(a, st) : launch_patch<28, 0>(a, st);
  return tag ? launch_patch<14, 1>(a, st) : launch_patch<14, 0>(a, st);
}
