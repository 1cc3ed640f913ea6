// conv_ps.hip — "patch-stationary" 3x3 implicit-GEMM conv for the wide layers (C_in % 64 == 0, C_out tiles of 256): the
// activation patch stays in LDS across the 9 taps, the weights stream through LDS, ONE wave per SIMD owns all 512 registers
// of its SIMD, and every instruction of the main loop is placed by hand.
//
// Why (measured): conv_igemm.hip moves 64 KiB of operands into LDS per 64-channel K-step and CU.  With
// its MFMAs compiled out the same loop still takes 61 % of its time: the L2 -> LDS DMA path delivers ~70 GB/s per CU
// (~33 B/clk), which is what 64 KiB per 2048 MFMA cycles asks for — the kernel is co-limited by the DMA path and the matrix
// pipe, and the two overlap badly because a wave that waits to issue a DMA cannot issue its MFMAs (two waves of a SIMD hit
// their DMA bursts together).  This kernel asks the DMA path for 45 % less: the nine taps of a 64-channel chunk re-read the
// SAME halo'd pixel patch from LDS (fetched once per chunk), only the 32 KiB weight slab is fetched per tap.  The freed
// margin is what lets 4 waves (one per SIMD, 112 px x 128 cout wave tiles, 224 accumulator registers in the AGPR half of the
// register file) keep the matrix pipe fed: 15 fragment reads per 56 MFMAs instead of 12 per 32, one barrier per K-step placed
// BETWEEN its two k-substeps, DMA two K-steps ahead, one filler (fragment read, DMA instruction, address arithmetic) behind
// every group of four MFMAs.
//
// Geometry (as conv_patch.hip): a workgroup owns 224 conv-output pixels = TR image rows x TC columns (8 x 28 or 16 x 14) of
// the global (image, row) list — tiles may straddle images, the patch is the contiguous range of PADDED rows between the
// first and the last row's halos — and 256 output channels.
//   D[cout][pixel] += W[tap][cout][chunk] . patch[pixel + tap][chunk]       (weights = MFMA A operand, pixels = B operand)
// Patch swizzle: 16-byte chunk ^= (col + 4 row) & 6 (TC = 28: conflict-free ds_read_b128 for every fragment, wrap position
// and tap under gfx950's lane groups; TC = 14: (col + 6 row) & 6 — with the two extra halo rows a 16-row tile crosses at every
// image boundary of 14 x 14 maps, multiplier 2 put 31 % of the kernel's LDS cycles into bank conflicts (PMC), 6 leaves 5 %;
// tools/lds_swizzle_check.py enumerates every fragment, tap and boundary phase).
// Epilogue: bias, ReLU, 2x2 max-pool, per-channel affine through LDS, 16-byte NHWC stores (same contract as
// vnqa_conv2d_igemm_fwd).  bf16 / fp16 storage only.
#include "conv_args.h"

namespace {

namespace ps {
constexpr int BM = 224, BN = 256, NW = 4, NT = 256;
constexpr int WTM = 112, WTN = 128, TM = 7, TN = 8;
constexpr int B_BYTES = BN * 128;
// patch pixels (rows of 128 B) an LDS patch buffer holds: 3x3 — up to 360 (tiles may straddle images); 5x5 — (TR + 4) (TC + 4)
// <= 384 (tiles never straddle images: h %% TR == 0 is required), which with two 32 KiB weight slabs is exactly 160 KiB
constexpr int patch_rows(int halo) { return halo == 1 ? 360 : 384; }
constexpr int WPW = (BN / 8) / NW;                          // 8 weight DMA instructions per wave and K-step
constexpr int CROW = BN * 2 + 16;
}  // namespace ps

#ifdef VNQA_H16_IS_F16
#define VNQA_PS_MFMA "v_mfma_f32_16x16x32_f16"
#else
#define VNQA_PS_MFMA "v_mfma_f32_16x16x32_bf16"
#endif

// accumulators live in AGPRs ("a"): with 224 of them per lane the VGPR half stays free for two fragment sets
template <bool FIRST>
__device__ __forceinline__ void ps_mfma(vnqa_f32x4& acc, const vnqa_f32x4& w, const vnqa_f32x4& x) {
  if constexpr (FIRST) asm volatile(VNQA_PS_MFMA " %0, %1, %2, 0" : "=&a"(acc) : "v"(w), "v"(x));
  else asm volatile(VNQA_PS_MFMA " %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(x));
}

// LDS-DMA from inline asm (invisible to hipcc's wait counting; every wait is hand-placed): wave-uniform base + 32-bit lane offset
// (M0 is clobbered, not saved and restored: nothing else in this kernel lives in it — no LDS-DMA builtin, no s_movrel — and with one wave per
// SIMD the two extra scalar moves per transfer were issue slots of the K loop: 8 transfers per K-step)
#ifdef VNQA_PS_GLDS_KEEP_M0      // the round 3-6 form (A/B partner)
__device__ __forceinline__ void ps_glds(const char* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
#else
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // ("reserved register on the clobber list": that IS the statement)
__device__ __forceinline__ void ps_glds(const char* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
// the same transfer as TWO statements for the K loop, each placed behind a different MFMA of a group: an MFMA holds the SIMD's vector
// issue for 8 of its 16 cycles, so ONE short instruction per MFMA gap is free and a clump behind the group's last MFMA is not
// (MI355X_MICROARCH.md, cycle constants); the MFMA between the two also is the wait state the M0 write needs before the transfer reads it
__device__ __forceinline__ void ps_m0(unsigned lds_addr) { asm volatile("s_mov_b32 m0, %0" : : "s"(lds_addr) : "m0"); }
__device__ __forceinline__ void ps_go(const char* sbase, unsigned voff) {
  asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase) : "memory");
}
#pragma clang diagnostic pop
#endif

// row multiplier of the patch swizzle key (col + RM row) & 6: exhaustive search per geometry (tools/lds_swizzle_check.py)
#ifndef VNQA_PS_RM14
#define VNQA_PS_RM14 6
#endif
template <int TC, int HALO> constexpr int ps_rm() { return TC == 14 ? (HALO == 1 ? VNQA_PS_RM14 : 6) : 4; }
template <int TC, int HALO> __device__ __forceinline__ int ps_swz(int row, int col) { return (col + ps_rm<TC, HALO>() * row) & 6; }

#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
// Diagnostic build only (tools/experiments/ps_inkernel_clock.py; MI355X_MICROARCH.md, DVFS item 6): s_memtime (shader clock) and s_memrealtime
// (100 MHz) stamped around the K loop by one lane of every workgroup of the 5x5 instantiation, into a buffer of their own.
__device__ unsigned long long g_ps_stamps[8 * 16384];
#ifndef VNQA_PS_DIAG_TC          // which instantiation stamps (default: the composed pair's <28, 2, 1>)
#define VNQA_PS_DIAG_TC 28
#define VNQA_PS_DIAG_HALO 2
#define VNQA_PS_DIAG_TAG 1
#endif
#define PS_DIAG_THIS (TC == VNQA_PS_DIAG_TC && HALO == VNQA_PS_DIAG_HALO && TAG == VNQA_PS_DIAG_TAG)
#endif

template <int TC, int HALO, int TAG>
__global__ void __launch_bounds__(ps::NT, 1) conv_ps_kernel(const ConvArgs p) {
  using namespace ps;
  constexpr int TR = BM / TC, PW = TC + 2 * HALO, KW = 2 * HALO + 1, NTAPS = KW * KW;
  constexpr int PATCH_ROWS = patch_rows(HALO), PATCH_BYTES = PATCH_ROWS * 128;
  constexpr int PATCH_INSTR = PATCH_ROWS / 8, PIW = (PATCH_INSTR + NW - 1) / NW;      // patch DMA instructions: all / per wave
  constexpr int RM = ps_rm<TC, HALO>();
  static_assert(BM * CROW <= 2 * PATCH_BYTES + 2 * B_BYTES && 2 * PATCH_BYTES + 2 * B_BYTES <= 160 * 1024, "LDS budget");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int WOFF = 2 * PATCH_BYTES;        // weight slabs behind the two patch buffers
#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
  const unsigned long long st_entry = __builtin_amdgcn_s_memrealtime();
#endif

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fh = lane >> 4;

  // XCD-aware bijective remap (as conv_igemm.hip): an XCD gets a contiguous run of tiles, n-tile fastest
  int tile_m, tile_n;
  {
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_n = t % p.tilesN;
    tile_m = t / p.tilesN;
    if constexpr (TAG == 1) {
      // VNQA_CONV_XCD_SPLIT_N (the composed 5x5: two cout tiles): XCD x owns cout half x & 1 only (as conv_igemm.hip)
      if (p.xcd_split && p.tilesN == 2 && (nwg & 1) == 0) {
        const int par = xcd & 1, j = xcd >> 1;
        int off = 0;
        for (int i = 0; i < j; ++i) off += q + ((2 * i + par) < r ? 1 : 0);
        tile_n = par;
        tile_m = off + (bid >> 3);
      }
    }
  }
  // column blocks: widths that are not a multiple of TC (26-wide maps of the reference's 160 x 208 frames) put the last block at
  // W - TC, overlapping its neighbour — the shared columns are computed twice to the same bits, no masking
  const int CB = (p.W + TC - 1) / TC;
  const int rt = tile_m / CB, cb = tile_m - rt * CB;
  const int xb = min(cb * TC, p.W - TC);       // first image column of this tile
  const int total_rows = p.n_img * p.H;
  const int g0 = rt * TR;
  // image of a global row: g / H as ONE multiply-high (the store loops ask once per 16-byte chunk; a 32-bit division by a run-time divisor
  // is ~30 instructions).  Exact for g * H < 2^32: rows < 2^31 / H is checked by the launcher's 32-bit addressing test.
  const unsigned h_magic = 0xFFFFFFFFu / (unsigned)p.H + 1u;
  auto img_of_row = [&](int g) { return p.H == 1 ? g : (int)__umulhi((unsigned)g, h_magic); };
  auto padrow = [&](int g) {
    const int n = img_of_row(g);
    return n * p.Hp + (g - n * p.H) + HALO;
  };
  const int g_last = min(g0 + TR - 1, total_rows - 1);
  const int pr_first = padrow(g0) - HALO;
  const int n_lin = (padrow(g_last) + HALO - pr_first + 1) * PW;   // patch pixels (rows of 128 B) actually needed
  const int n_instr = (n_lin + 7) >> 3;

  const int kchunks = p.Cin >> 6;
  const unsigned cin_b = (unsigned)p.Cin * 2;

  // ---- per-lane DMA source offsets (32-bit, relative to wave-uniform bases) ----
  const char* const x_base = p.x + ((size_t)pr_first * p.Wp + (size_t)xb) * cin_b;
  unsigned a_off[PIW];                    // patch instruction q = wave + 4 j: LDS pixels 8q .. 8q+7
#pragma unroll
  for (int j = 0; j < PIW; ++j) {
    const int lin0 = (wave + NW * j) * 8 + (lane >> 3);
    const int i0 = lin0 / PW, c0 = lin0 - i0 * PW;                 // LDS position (keys the swizzle, before the clamp)
    const int lin = lin0 < n_lin ? lin0 : n_lin - 1;               // pixels past the patch are never read; keep the address in bounds
    const int i = lin / PW, jj = lin - i * PW;
    a_off[j] = (unsigned)(i * p.Wp + jj) * cin_b + (unsigned)(((lane & 7) ^ ps_swz<TC, HALO>(i0, c0)) << 4);
  }
  unsigned b_off[WPW];
  const unsigned w_row_bytes = (unsigned)NTAPS * cin_b;
#pragma unroll
  for (int j = 0; j < WPW; ++j) {
    const int row = (wave * WPW + j) * 8 + (lane >> 3);
    int co = tile_n * BN + row;
    co = co < p.Cout ? co : p.Cout - 1;
    b_off[j] = (unsigned)co * w_row_bytes + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
  }
  auto dma_patch_piece = [&](int kc, int j) {      // j-th instruction of this wave, chunk kc (wave-uniform predicate)
    if (wave + NW * j < n_instr)
      ps_glds(x_base + (size_t)kc * 128, a_off[j],
              __builtin_amdgcn_readfirstlane(lds0 + (kc & 1) * PATCH_BYTES + (wave + NW * j) * 1024));
  };
  auto dma_weight_piece = [&](int kc, int tap, int slab, int j) {
    ps_glds(p.wt + ((size_t)tap * p.Cin + (size_t)kc * 64) * 2, b_off[j],
            __builtin_amdgcn_readfirstlane(lds0 + WOFF + slab * B_BYTES + (wave * WPW + j) * 1024));
  };

  // ---- fragment addressing ----
  // weights: row = wn*128 + 16 j + fr, its swizzle (row >> 1) & 7 = (fr >> 1) & 7 for every j: one base per k-substep + immediates
  const int w_rd0 = WOFF + (wn * WTN + fr) * 128 + ((fh ^ ((fr >> 1) & 7)) << 4);
  // pixels: patch position of (pixel, tap (0,0)); tap (r, s) adds r PW + s pixels and (s + 4 r [2 r]) to the swizzle key
  int x_lin128[TM], x_key16[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int ml = wm * WTM + i * 16 + fr;
    const int tr = ml / TC, tc = ml - tr * TC;
    const int g = min(g0 + tr, total_rows - 1);
    const int R0 = padrow(g) - pr_first - HALO;
    x_lin128[i] = (R0 * PW + tc) * 128 + (fh << 4);
    x_key16[i] = (tc + RM * R0) << 4;
  }

  vnqa_f32x4 acc[TM][TN];          // zeroed here, long before the first MFMA reads them as C (no wait states needed there)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = vnqa_f32x4{0.f, 0.f, 0.f, 0.f};
  vnqa_f32x4 xf0[TM], wf0[TN], xf1[TM], wf1[TN];

#ifdef VNQA_PS_STAGGER      // experiment builds only (tools/experiments/ps_stagger.sh): do two tiles that read the same lines share them through L2 when
  // the second asks ~3 us later instead of at the same time?  1: the odd cout half waits; 2: odd pixel tiles wait
  if ((VNQA_PS_STAGGER == 1 && (tile_n & 1)) || (VNQA_PS_STAGGER == 2 && (tile_m & 1))) {
    __builtin_amdgcn_s_sleep(100);
  }
#endif
  // ---- prologue: the whole patch of chunk 0, weights of K-steps 0 and 1 ----
  asm volatile("s_nop 4" ::: "memory");
#pragma unroll
  for (int j = 0; j < PIW; ++j) dma_patch_piece(0, j);
#pragma unroll
  for (int j = 0; j < WPW; ++j) dma_weight_piece(0, 0, 0, j);
#pragma unroll
  for (int j = 0; j < WPW; ++j) dma_weight_piece(0, 1, 1, j);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // x fragment address of pixel fragment i for the tap whose (pixel offset, swizzle key offset) are compile-time constants
  auto x_addr = [&](int i, int pbuf_off, int tap) {
    const int r = tap / KW, s = tap - KW * r;
    const int key = ((x_key16[i] + ((s + RM * r) << 4)) & 0x60);
    return (x_lin128[i] ^ key) + pbuf_off + (r * PW + s) * 128;      // (the XOR only touches the chunk bits of the pixel's 128 bytes)
  };
  // first fragments: K-step 0, substep 0
#pragma unroll
  for (int i = 0; i < TM; ++i) xf0[i] = *(const vnqa_f32x4*)(smem + x_addr(i, 0, 0));
#pragma unroll
  for (int j = 0; j < TN; ++j) wf0[j] = *(const vnqa_f32x4*)(smem + w_rd0 + j * 2048);

#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
  const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  int slab = 0;                                  // slab of the current K-step's weights = kt & 1
  constexpr int PPT = (PIW + NTAPS - 4) / (NTAPS - 3);     // patch pieces of the next chunk issued per tap (taps 0 .. NTAPS-4)
  for (int kc = 0; kc < kchunks; ++kc) {
    const int pbuf = (kc & 1) * PATCH_BYTES;
    const bool more_chunks = kc + 1 < kchunks;
#pragma unroll
    for (int tap = 0; tap < NTAPS; ++tap) {
      // position of the K-steps to come
      const int tap1 = tap == NTAPS - 1 ? 0 : tap + 1;                          // K-step kt + 1
      const int tap2 = tap >= NTAPS - 2 ? tap - (NTAPS - 2) : tap + 2;          // K-step kt + 2
      const bool has1 = tap < NTAPS - 1 || more_chunks;
      const int kc1 = tap == NTAPS - 1 ? kc + 1 : kc;
      const int kc2 = tap >= NTAPS - 2 ? kc + 1 : kc;
      const bool do2 = kc2 < kchunks;
      const int pbuf1 = (kc1 & 1) * PATCH_BYTES;
      // ---- phase 0: substep 0 MFMAs; behind the first eight groups the 15 fragment reads of substep 1 (the last six groups
      //      cover their latency) ----
#pragma unroll
      for (int slot = 0; slot < 14; ++slot) {
        const int i = slot >> 1, h = slot & 1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 4 * h; j < 4 * h + 4; ++j) ps_mfma<false>(acc[i][j], wf0[j], xf0[i]);
        if (slot < TM) xf1[slot] = *(const vnqa_f32x4*)(smem + (x_addr(slot, pbuf, tap) ^ 64));
        if (slot < TN) wf1[slot] = *(const vnqa_f32x4*)(smem + (w_rd0 ^ 64) + slab * B_BYTES + slot * 2048);
      }
      __builtin_amdgcn_sched_barrier(0);
#if !defined(VNQA_PS_DIAG) || VNQA_PS_DIAG != 2
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // weights of K-step kt+1 and any patch piece issued a K-step ago have landed
#endif
      __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): every fragment of this K-step is in registers
#if !defined(VNQA_PS_DIAG) || VNQA_PS_DIAG != 1          // timing-only builds (tools/experiments/ps_kstep_diag.sh): 1 = no mid-step barrier, 2 = no DMA wait
      __builtin_amdgcn_s_barrier();                        // slab (kt & 1) is free; slab ((kt+1) & 1) / the next patch are visible
#endif
      // ---- phase 1: substep 1 MFMAs; behind them the 15 fragment reads of K-step kt+1 / substep 0 (groups 0..7), the 8
      //      weight DMA instructions of K-step kt+2 and the patch pieces of the next chunk ----
#if defined(VNQA_PS_GLDS_KEEP_M0) || defined(VNQA_PS_DMA_CLUMPED)      // the round 3-6 placement (A/B partner): everything behind the group's last MFMA
#pragma unroll
      for (int slot = 0; slot < 14; ++slot) {
        const int i = slot >> 1, h = slot & 1;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 4 * h; j < 4 * h + 4; ++j) ps_mfma<false>(acc[i][j], wf1[j], xf1[i]);
        if (has1 && slot < TM) xf0[slot] = *(const vnqa_f32x4*)(smem + x_addr(slot, pbuf1, tap1));
        if (has1 && slot < TN) wf0[slot] = *(const vnqa_f32x4*)(smem + w_rd0 + (slab ^ 1) * B_BYTES + slot * 2048);
        // weight pieces behind groups 0,1,3,5,7,8,10,12; patch pieces behind groups 9 and 11
        constexpr int wslot[8] = {0, 1, 3, 5, 7, 8, 10, 12};
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (slot == wslot[q] && do2) dma_weight_piece(kc2, tap2, slab, q);
        if (more_chunks && tap < NTAPS - 3 && (slot == 9 || slot == 11)) {
          const int pj = tap * PPT + (slot == 11 ? 1 : 0);
          if ((slot == 9 || PPT > 1) && pj < PIW) dma_patch_piece(kc + 1, pj);
        }
      }
#else
      // weight pieces in groups 0,1,3,5,7,8,10,12, patch pieces in groups 9 and 11 — a piece's M0 write behind the group's first MFMA, its
      // transfer behind the second, the group's fragment reads behind the third and fourth: one short instruction per MFMA gap
#pragma unroll
      for (int slot = 0; slot < 14; ++slot) {
        const int i = slot >> 1, h = slot & 1;
        constexpr int wslot[8] = {0, 1, 3, 5, 7, 8, 10, 12};
        int q = -1;
#pragma unroll
        for (int qq = 0; qq < 8; ++qq)
          if (slot == wslot[qq]) q = qq;
        const bool wpiece = q >= 0 && do2;
        const int pj = tap * PPT + (slot == 11 ? 1 : 0);
        const bool ppiece = more_chunks && tap < NTAPS - 3 && (slot == 9 || (slot == 11 && PPT > 1)) && pj < PIW &&
                            wave + NW * pj < n_instr;                     // (wave-uniform)
        __builtin_amdgcn_sched_barrier(0);
        ps_mfma<false>(acc[i][4 * h + 0], wf1[4 * h + 0], xf1[i]);
        if (wpiece) ps_m0(__builtin_amdgcn_readfirstlane(lds0 + WOFF + slab * B_BYTES + (wave * WPW + q) * 1024));
        if (ppiece) ps_m0(__builtin_amdgcn_readfirstlane(lds0 + ((kc + 1) & 1) * PATCH_BYTES + (wave + NW * pj) * 1024));
        ps_mfma<false>(acc[i][4 * h + 1], wf1[4 * h + 1], xf1[i]);
        if (wpiece) ps_go(p.wt + ((size_t)tap2 * p.Cin + (size_t)kc2 * 64) * 2, b_off[q < 0 ? 0 : q]);
        if (ppiece) ps_go(x_base + (size_t)(kc + 1) * 128, a_off[pj < PIW ? pj : 0]);
        ps_mfma<false>(acc[i][4 * h + 2], wf1[4 * h + 2], xf1[i]);
        if (has1 && slot < TM) xf0[slot] = *(const vnqa_f32x4*)(smem + x_addr(slot, pbuf1, tap1));
        ps_mfma<false>(acc[i][4 * h + 3], wf1[4 * h + 3], xf1[i]);
        if (has1 && slot < TN) wf0[slot] = *(const vnqa_f32x4*)(smem + w_rd0 + (slab ^ 1) * B_BYTES + slot * 2048);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      slab ^= 1;
    }
  }
#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
  if (PS_DIAG_THIS && threadIdx.x == 0 && blockIdx.x < 16384) {
    const unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
    g_ps_stamps[8 * blockIdx.x + 0] = st_c1 - st_c0;
    g_ps_stamps[8 * blockIdx.x + 1] = st_r1 - st_r0;
    g_ps_stamps[8 * blockIdx.x + 2] = st_r0 - st_entry;
    g_ps_stamps[8 * blockIdx.x + 3] = (unsigned long long)kchunks * NTAPS;
    g_ps_stamps[8 * blockIdx.x + 4] = st_entry;
    g_ps_stamps[8 * blockIdx.x + 5] = st_r1;
  }
#endif
  // wait states between the last MFMAs and the first read of an accumulator (8-pass XDL: 12+), then free the LDS
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
  asm volatile("s_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
  __builtin_amdgcn_s_barrier();      // every wave is done reading fragments: the LDS becomes the epilogue tile

  // ---------------- epilogue ----------------
  // acc[i][j][e]: pixel = wm*112 + i*16 + fr ; cout = wn*128 + j*16 + 4*fh + e
  if constexpr (TAG == 2) {
    // DUAL output (VNQA_CONV_DUAL_OUT; precision 'fp16h'): y has 2 Cout channels per pixel, [hi | lo] with hi = h16(v) and
    // lo = h16(v - hi) — the fp32 result as a PAIR of 16-bit values (22 significand bits), so that the consumer, a plain conv
    // over 2 Cout input channels against [w | w], contracts the UNROUNDED activation: the storage rounding of this tensor, one
    // of the few that dominate the 16-bit logits error (profiles/r05_precision_budget.txt), is gone at the price of the
    // consumer's second product.  Everything is finished in fp32 (bias, ReLU, 2x2 max-pool, affine) — both 16-bit halves of
    // the tile at once would need 2 x 118 KiB of LDS, so the tile goes through LDS as fp32 in two passes of 128 couts (the
    // waves of cout half `pass` stage, all four store): same LDS bytes per pass as the plain 16-bit epilogue.
    constexpr int CROWF = WTN * 4 + 16;
    static_assert(BM * CROWF <= 2 * PATCH_BYTES + 2 * B_BYTES, "LDS budget (dual epilogue)");
    constexpr int CHF = WTN / 8;          // 8-channel chunks (32 B of fp32) per staged row
    const bool has_post_d = (p.post_scale != nullptr);
    const int rows_out_d = p.pool ? BM / 4 : BM;
    const int OCd = p.pool ? TC / 2 : TC;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      if (wn == pass) {
        float b4a[TN][4];              // (all of the pass's bias values requested before the first is used: see the plain epilogue)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int co = tile_n * BN + pass * WTN + j * 16 + 4 * fh;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float bv = p.bias != nullptr ? p.bias[co + e < p.Cout ? co + e : p.Cout - 1] : 0.f;
            b4a[j][e] = co + e < p.Cout ? bv : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = j * 16 + 4 * fh;                 // within this pass's 128 couts
          const float* b4 = b4a[j];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int prow = wm * WTM + i * 16 + fr;
            float4 v;
            v.x = acc[i][j][0] + b4[0]; v.y = acc[i][j][1] + b4[1]; v.z = acc[i][j][2] + b4[2]; v.w = acc[i][j][3] + b4[3];
            if (p.relu) {      // (one v_max each: fmaxf canonicalises its operands first)
              asm("v_max_f32 %0, %0, 0" : "+v"(v.x)); asm("v_max_f32 %0, %0, 0" : "+v"(v.y));
              asm("v_max_f32 %0, %0, 0" : "+v"(v.z)); asm("v_max_f32 %0, %0, 0" : "+v"(v.w));
            }
            *(float4*)(smem + prow * CROWF + col * 4) = v;
          }
        }
      }
      __syncthreads();
#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
      if (PS_DIAG_THIS && pass == 0 && threadIdx.x == 0 && blockIdx.x < 16384) g_ps_stamps[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();
#endif
      // (a thread keeps its channel chunk over the loop — NT % CHF == 0 —: its 16 affine constants are fetched ONCE per pass, not in
      // every iteration behind the previous iteration's store)
      static_assert(NT % CHF == 0, "dual store loop: constant chunk per thread");
      float psc[8], psh[8];
      {
        const int cq = tile_n * BN + pass * WTN + (threadIdx.x % CHF) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int cc = cq + e < p.Cout ? cq + e : p.Cout - 1;
          psc[e] = has_post_d ? p.post_scale[cc] : 1.f;
          psh[e] = has_post_d ? p.post_shift[cc] : 0.f;
        }
      }
      for (int idx = threadIdx.x; idx < rows_out_d * CHF; idx += NT) {
        const int orow = idx / CHF, c = idx - orow * CHF;
        const int co0 = tile_n * BN + pass * WTN + c * 8;
        const int orr = orow / OCd, occ = orow - orr * OCd;
        const int g = g0 + (p.pool ? 2 * orr : orr);
        if (g >= total_rows || co0 >= p.Cout) continue;
        float v[8];
        if (p.pool) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = -INFINITY;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int ml = (2 * orr + (d >> 1)) * TC + 2 * occ + (d & 1);
            const float4 a0 = *(const float4*)(smem + ml * CROWF + c * 32);
            const float4 a1 = *(const float4*)(smem + ml * CROWF + c * 32 + 16);
            v[0] = fmaxf(v[0], a0.x); v[1] = fmaxf(v[1], a0.y); v[2] = fmaxf(v[2], a0.z); v[3] = fmaxf(v[3], a0.w);
            v[4] = fmaxf(v[4], a1.x); v[5] = fmaxf(v[5], a1.y); v[6] = fmaxf(v[6], a1.z); v[7] = fmaxf(v[7], a1.w);
          }
        } else {
          const float4 a0 = *(const float4*)(smem + orow * CROWF + c * 32);
          const float4 a1 = *(const float4*)(smem + orow * CROWF + c * 32 + 16);
          v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
        }
        if (has_post_d) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] * psc[e] + psh[e];
        }
        unsigned hw[4], lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hw[e] = pack2_h16(v[2 * e], v[2 * e + 1]);
        const int n = img_of_row(g);
        const int y = g - n * p.H;
        const int yo = p.pool ? (y >> 1) : y;
        const int xo = (p.pool ? xb >> 1 : xb) + occ;
        const size_t ooff = (((size_t)n * p.Hyp + yo + p.y_halo) * p.Wyp + xo + p.y_halo) * (size_t)p.Cy + co0;
        vnqa_bf16* dst = (vnqa_bf16*)(p.y) + ooff;
        *(uint4*)dst = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        if (p.dual_out >= 8) {              // VNQA_CONV_F32_EPILOGUE: ONE 16-bit value, rounded once after pool / affine in fp32 ...
          if (p.dual_out == 9) *(uint4*)(dst + p.Cout) = make_uint4(hw[0], hw[1], hw[2], hw[3]);      // ... | VNQA_CONV_DUAL_HI2: written TWICE, [v | v]
          continue;                         // (no lo half to form: a wave-uniform exit before its twelve instructions)
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) lw[e] = pack2_h16(v[2 * e] - h16_lo(hw[e]), v[2 * e + 1] - h16_hi(hw[e]));
        if (p.dual_out == 4) {      // VNQA_EPI_SPLIT_OUT: hi and lo as TWO plain tensors of y's geometry (y, y2)
          *(uint4*)((vnqa_bf16*)(p.y2) + ooff) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
          continue;
        }
        *(uint4*)(dst + p.Cout) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        // VNQA_CONV_DUAL_HI2: [hi | lo | hi] — the operand of a THREE-product consumer (a plain conv over 3 Cout channels against
        // [w_hi | w_hi | w_lo]) laid out by the producer, no wrap logic in the consumer's main loop
        if (p.dual_out == 2) *(uint4*)(dst + 2 * p.Cout) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      }
      __syncthreads();
    }
#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
    if (PS_DIAG_THIS && threadIdx.x == 0 && blockIdx.x < 16384) g_ps_stamps[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }
  // composed-conv border correction (vnqa_conv2d_igemm_fwd_ex): row of the correction tensor for each of this lane's pixels
  // (-1: interior pixel or no correction); ring order: top row, bottom row, left column, right column
  int ring_row[TM];
  {
    // (5x5 tiles never straddle images — ps_geometry —: one image / first row per tile, scalar; 3x3 tiles may: per pixel)
    const int n0 = g0 / p.H, y00 = g0 - n0 * p.H;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ring_row[i] = -1;
      if (p.border_sub != nullptr) {
        const int ml = wm * WTM + i * 16 + fr;
        const int tr = ml / TC, tc = ml - tr * TC;
        const int g = g0 + tr;
        if (g < total_rows) {
          int n = n0, y = y00 + tr;
          if constexpr (HALO != 2) {
            n = img_of_row(g);
            y = g - n * p.H;
          }
          const int x = xb + tc;
          int ring = -1;
          if (y == 0) ring = x;
          else if (y == p.H - 1) ring = p.W + x;
          else if (x == 0) ring = 2 * p.W + (y - 1);
          else if (x == p.W - 1) ring = 2 * p.W + (p.H - 2) + (y - 1);
          if (ring >= 0) ring_row[i] = n * (2 * p.W + 2 * (p.H - 2)) + ring;
        }
      }
    }
  }
  // Every global operand of this loop — bias, the ReLU floor, the border correction of this lane's border pixels — is requested in ONE batch
  // before the first is used.  (Round 6: fetched where they were used, each (j, i) fragment waited for its own 8-byte correction load and each j
  // for its bias / floor loads — up to 40 serial L2 round trips per tile of the composed 5x5, whose every tile touches the image border.)
  // Unpredicated loads from clamped addresses, selected ONCE where they land: one wave per SIMD pays an issue slot for every instruction of
  // the 56-fragment loop below, which is down to read accumulator / add / subtract / max / convert / store.
  // VNQA_CONV_RELU_FLOOR (stem launches): post_shift WITHOUT post_scale is the ReLU's per-channel floor — relu(a) - m = max(a - m, -m),
  // a mean-shifted output rounded once through this 16-bit staging (conv_igemm.hip has the same lines); no ReLU: floor = -inf.
  float b4[TN][4], fl4[TN][4];
  const bool has_floor = TAG == 1 && p.post_scale == nullptr && p.post_shift != nullptr;
  const bool vec4 = (p.Cout & 3) == 0;       // (always, for the K-major packs of this library: c_out padded to 64)
  const float* const bias_p = p.bias;
  const float* const floor_p = p.post_shift;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int co = tile_n * BN + wn * WTN + j * 16 + 4 * fh;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), fv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec4) {
      const unsigned cc = co + 3 < p.Cout ? (unsigned)co * 4u : 0u;       // 32-bit byte offset from a uniform base
      if (bias_p != nullptr) bv = *(const float4*)((const char*)bias_p + cc);
      if (has_floor) fv = *(const float4*)((const char*)floor_p + cc);
    } else {
      float bb[4] = {0.f, 0.f, 0.f, 0.f}, ff[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int cc = co + e < p.Cout ? co + e : p.Cout - 1;
        if (bias_p != nullptr) bb[e] = bias_p[cc];
        if (has_floor) ff[e] = floor_p[cc];
      }
      bv = make_float4(bb[0], bb[1], bb[2], bb[3]);
      fv = make_float4(ff[0], ff[1], ff[2], ff[3]);
    }
    b4[j][0] = bv.x; b4[j][1] = bv.y; b4[j][2] = bv.z; b4[j][3] = bv.w;       // (couts past the tensor: finite values, never stored)
    fl4[j][0] = fv.x; fl4[j][1] = fv.y; fl4[j][2] = fv.z; fl4[j][3] = fv.w;
    if (!p.relu) fl4[j][0] = fl4[j][1] = fl4[j][2] = fl4[j][3] = -INFINITY;
  }
  // border correction: only the pixel fragments with a border pixel in this wave ask for it (a wave's 112 pixels are 4 tile rows: the image's
  // left / right column lands in up to four of its seven fragments, the top / bottom row in two or three), only their border lanes load, and all of
  // a fragment's eight loads are in flight together; the subtraction below runs under the same lane mask
  const bool has_sub = p.border_sub != nullptr;
  // (c_out % 16 == 0 with a correction — the launcher checks —: a 16-cout fragment column is inside the tensor or outside it as a whole, a
  // wave-uniform test; couts outside are computed on finite garbage and never stored)
  uint2 subraw[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (has_sub && ring_row[i] >= 0) {
      const unsigned ro = (unsigned)ring_row[i] * (unsigned)p.Cout * 2u + (unsigned)(tile_n * BN + wn * WTN + 4 * fh) * 2u;     // (< 2^32: checked by the launcher)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned cj = tile_n * BN + wn * WTN + j * 16 < p.Cout ? (unsigned)(j * 32) : 0u;
        subraw[i][j] = *(const uint2*)((const char*)p.border_sub + (ro + cj));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int prow = wm * WTM + i * 16 + fr;
    float v[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] = acc[i][j][e] + b4[j][e];
    if (has_sub && ring_row[i] >= 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const uint2 raw = subraw[i][j];
        v[j][0] -= h16_lo(raw.x); v[j][1] -= h16_hi(raw.x); v[j][2] -= h16_lo(raw.y); v[j][3] -= h16_hi(raw.y);
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = wn * WTN + j * 16 + 4 * fh;
#pragma unroll
      for (int e = 0; e < 4; ++e)      // max(v, floor) as ONE instruction (fmaxf: a canonicalising v_max of each operand first)
        asm("v_max_f32 %0, %1, %2" : "=v"(v[j][e]) : "v"(v[j][e]), "v"(fl4[j][e]));
      uint2 pk;
      pk.x = pack2_h16(v[j][0], v[j][1]);
      pk.y = pack2_h16(v[j][2], v[j][3]);
      *(uint2*)(smem + prow * CROW + col * 2) = pk;
    }
  }
  __syncthreads();
#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
  if (PS_DIAG_THIS && threadIdx.x == 0 && blockIdx.x < 16384) g_ps_stamps[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();
#endif

  constexpr int CH = BN * 2 / 16;       // 16-byte chunks per tile row
#ifndef VNQA_PS_EPI_UNBATCHED       // (tools/build_variant.py unbatched -DVNQA_PS_EPI_UNBATCHED: the A/B partner)
  if constexpr (TAG == 0) {
    // Fused trunk epilogues (FILM_RES / ADD_MASK; un-pooled, y_halo = 1): the same arithmetic as the generic loop below, but
    // a thread's 28 chunks go in batches of UN whose global operands (res / add / mask, gamma / beta rows) are ALL requested
    // before the first is used.  One chunk at a time, every iteration waited for its own loads: ~1.7 us x 28 per tile —
    // FILM_RES cost a C = 1024 conv +31 % (3.83 vs 2.92 ms) for 0.9 GB of extra traffic worth 0.2 ms.
    if (p.epi == VNQA_EPI_FILM_RES || p.epi == VNQA_EPI_ADD_MASK) {
#ifndef VNQA_PS_EPI_UN      // batch size of the fused store loop (kernel alone, FILM_RES at C = 1024: 2: 3.24 ms, 4: 3.21, 7: 3.21)
#define VNQA_PS_EPI_UN 4
#endif
      constexpr int UN = VNQA_PS_EPI_UN;
      const bool film = p.epi == VNQA_EPI_FILM_RES;
      const int c = threadIdx.x % CH;                       // NT % CH == 0: a thread keeps its channel chunk
      const int co0 = tile_n * BN + c * 8;
      static_assert(NT % CH == 0, "store loop: constant chunk per thread");
      for (int row0 = threadIdx.x / CH; row0 < BM; row0 += UN * (NT / CH)) {
        size_t ooff[UN];
        bool ok[UN];
        int nn[UN];
        uint4 ra[UN], rb[UN];
        float4 g0v[UN], g1v[UN], b0v[UN], b1v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int orow = row0 + u * (NT / CH);
          const int orr = orow / TC, occ = orow - orr * TC;
          const int g = g0 + orr;
          ok[u] = orow < BM && g < total_rows && co0 < p.Cout;
          const int gg = ok[u] ? g : g0;
          const int n = img_of_row(gg), y = gg - n * p.H;
          nn[u] = n;
          ooff[u] = (((size_t)n * p.Hyp + y + 1) * p.Wyp + xb + occ + 1) * (size_t)p.Cy + co0;
          ra[u] = rb[u] = make_uint4(0u, 0u, 0u, 0u);
          g0v[u] = g1v[u] = b0v[u] = b1v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok[u]) {
            ra[u] = *(const uint4*)((const vnqa_bf16*)p.res + ooff[u]);
            if (!film) rb[u] = *(const uint4*)((const vnqa_bf16*)p.y2 + ooff[u]);
            else if (co0 + 7 < p.film_c) {
              const float* gp = p.film_gamma + (size_t)n * p.film_ld + co0;
              const float* bp = p.film_beta + (size_t)n * p.film_ld + co0;
              g0v[u] = *(const float4*)gp; g1v[u] = *(const float4*)(gp + 4);
              b0v[u] = *(const float4*)bp; b1v[u] = *(const float4*)(bp + 4);
            } else {
              float ga[8], be[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const bool in = co0 + e < p.film_c;
                ga[e] = in ? p.film_gamma[(size_t)n * p.film_ld + co0 + e] : 0.f;
                be[e] = in ? p.film_beta[(size_t)n * p.film_ld + co0 + e] : 0.f;
              }
              g0v[u] = make_float4(ga[0], ga[1], ga[2], ga[3]); g1v[u] = make_float4(ga[4], ga[5], ga[6], ga[7]);
              b0v[u] = make_float4(be[0], be[1], be[2], be[3]); b1v[u] = make_float4(be[4], be[5], be[6], be[7]);
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          if (!ok[u]) continue;
          const int orow = row0 + u * (NT / CH);
          const uint4 uu = *(const uint4*)(smem + orow * CROW + c * 16);
          const unsigned w4[4] = {uu.x, uu.y, uu.z, uu.w};
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] = h16_lo(w4[e]);
            v[2 * e + 1] = h16_hi(w4[e]);
          }
          vnqa_bf16* dst = p.y != nullptr ? (vnqa_bf16*)(p.y) + ooff[u] : nullptr;      // (FILM_RES with y == NULL: z is not kept)
          vnqa_bf16* second = nullptr;
          const unsigned aw[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w};
          if (!film) {
            const unsigned mw[4] = {rb[u].x, rb[u].y, rb[u].z, rb[u].w};
            float w[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              w[2 * e] = h16_lo(mw[e]) > 0.f ? v[2 * e] + h16_lo(aw[e]) : 0.f;
              w[2 * e + 1] = h16_hi(mw[e]) > 0.f ? v[2 * e + 1] + h16_hi(aw[e]) : 0.f;
            }
            uint4 o;
            o.x = pack2_h16(w[0], w[1]); o.y = pack2_h16(w[2], w[3]); o.z = pack2_h16(w[4], w[5]); o.w = pack2_h16(w[6], w[7]);
            *(uint4*)dst = o;
          } else {
            if (dst != nullptr) *(uint4*)dst = uu;           // z, exactly the 16-bit values staged in LDS
            const float ga[8] = {g0v[u].x, g0v[u].y, g0v[u].z, g0v[u].w, g1v[u].x, g1v[u].y, g1v[u].z, g1v[u].w};
            const float be[8] = {b0v[u].x, b0v[u].y, b0v[u].z, b0v[u].w, b1v[u].x, b1v[u].y, b1v[u].z, b1v[u].w};
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float r = (e & 1) ? h16_hi(aw[e >> 1]) : h16_lo(aw[e >> 1]);
              w[e] = fmaxf(ga[e] * v[e] + be[e], 0.f) + r;
            }
            uint4 o2;
            o2.x = pack2_h16(w[0], w[1]); o2.y = pack2_h16(w[2], w[3]); o2.z = pack2_h16(w[4], w[5]); o2.w = pack2_h16(w[6], w[7]);
            second = (vnqa_bf16*)p.y2 + ooff[u];
            *(uint4*)second = o2;
          }
          if (p.zero_halo) {
            const int orr = orow / TC, occ = orow - orr * TC;
            const int gg = g0 + orr;
            const int yo = gg - nn[u] * p.H, xo = xb + occ;
            const uint4 zz = make_uint4(0u, 0u, 0u, 0u);
            const long long rs = (long long)p.Wyp * p.Cy, cs = p.Cy;
            const bool x0 = xo == 0, x1 = xo == p.W - 1, y0 = yo == 0, y1 = yo == p.H - 1;
            if (x0 | x1 | y0 | y1) {
#pragma unroll
              for (int which = 0; which < 2; ++which) {
                vnqa_bf16* b = which == 0 ? dst : second;
                if (b == nullptr) continue;
                if (x0) *(uint4*)(b - cs) = zz;
                if (x1) *(uint4*)(b + cs) = zz;
                if (y0) {
                  *(uint4*)(b - rs) = zz;
                  if (x0) *(uint4*)(b - rs - cs) = zz;
                  if (x1) *(uint4*)(b - rs + cs) = zz;
                }
                if (y1) {
                  *(uint4*)(b + rs) = zz;
                  if (x0) *(uint4*)(b + rs - cs) = zz;
                  if (x1) *(uint4*)(b + rs + cs) = zz;
                }
              }
            }
          }
        }
      }
      return;
    }
  }
#endif
  const bool has_post = (p.post_scale != nullptr);
  const int rows_out = p.pool ? BM / 4 : BM;
  const int OC = p.pool ? TC / 2 : TC;  // output columns per tile row
  static_assert(NT % CH == 0, "store loop: constant chunk per thread");
  float psc[8], psh[8];                 // (fetched once: a thread keeps its channel chunk)
  {
    const int cq = tile_n * BN + (threadIdx.x % CH) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int cc = cq + e < p.Cout ? cq + e : p.Cout - 1;
      psc[e] = has_post ? p.post_scale[cc] : 1.f;
      psh[e] = has_post ? p.post_shift[cc] : 0.f;
    }
  }
  // No affine, no fused trunk epilogue (the stem's plain launches): the staged 16-bit values go out as they are — copied, or their 2x2
  // maximum taken on the packed pairs (the same bits as unpack / fmaxf / repack at a tenth of the instructions) — in batches of UN rows whose
  // LDS reads are ALL requested before the first store: one row at a time every iteration waited for its own read (7.5 us of a 93-us
  // conv21 tile, profiles/r06_ps_tile_phases.txt).
  if (!has_post && (TAG != 0 || p.epi == VNQA_EPI_NONE)) {
    constexpr int UN = 4, RPP = NT / CH;          // rows per pass of the workgroup
    const int c = threadIdx.x % CH, co0 = tile_n * BN + c * 8;
    const int Ho = p.pool ? p.H >> 1 : p.H, Wo = p.pool ? p.W >> 1 : p.W;
    const long long rs = (long long)p.Wyp * p.Cy, cs = p.Cy;
    if (co0 < p.Cout) {
      for (int r0 = threadIdx.x / CH; r0 < rows_out; r0 += UN * RPP) {
        uint4 o[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int orow = min(r0 + u * RPP, rows_out - 1);
          if (p.pool) {
            const int orr = orow / OC, occ = orow - orr * OC;
            const int ml = 2 * orr * TC + 2 * occ;
            const uint4 u0 = *(const uint4*)(smem + ml * CROW + c * 16), u1 = *(const uint4*)(smem + (ml + 1) * CROW + c * 16);
            const uint4 u2 = *(const uint4*)(smem + (ml + TC) * CROW + c * 16), u3 = *(const uint4*)(smem + (ml + TC + 1) * CROW + c * 16);
            o[u].x = h16x2_max(h16x2_max(u0.x, u1.x), h16x2_max(u2.x, u3.x));
            o[u].y = h16x2_max(h16x2_max(u0.y, u1.y), h16x2_max(u2.y, u3.y));
            o[u].z = h16x2_max(h16x2_max(u0.z, u1.z), h16x2_max(u2.z, u3.z));
            o[u].w = h16x2_max(h16x2_max(u0.w, u1.w), h16x2_max(u2.w, u3.w));
          } else {
            o[u] = *(const uint4*)(smem + orow * CROW + c * 16);
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int orow = r0 + u * RPP;
          const int orr = orow / OC, occ = orow - orr * OC;
          const int g = g0 + (p.pool ? 2 * orr : orr);
          if (orow >= rows_out || g >= total_rows) continue;
          const int n = img_of_row(g);
          const int y = g - n * p.H;
          const int yo = p.pool ? (y >> 1) : y;
          const int xo = (p.pool ? xb >> 1 : xb) + occ;
          vnqa_bf16* b = (vnqa_bf16*)(p.y) + (((size_t)n * p.Hyp + yo + p.y_halo) * p.Wyp + xo + p.y_halo) * (size_t)p.Cy + co0;
          *(uint4*)b = o[u];
          if (p.zero_halo) {      // (as in the generic loop below)
            const uint4 zz = make_uint4(0u, 0u, 0u, 0u);
            const bool x0 = xo == 0, x1 = xo == Wo - 1, y0 = yo == 0, y1 = yo == Ho - 1;
            if (x0 | x1 | y0 | y1) {
              if (x0) *(uint4*)(b - cs) = zz;
              if (x1) *(uint4*)(b + cs) = zz;
              if (y0) {
                *(uint4*)(b - rs) = zz;
                if (x0) *(uint4*)(b - rs - cs) = zz;
                if (x1) *(uint4*)(b - rs + cs) = zz;
              }
              if (y1) {
                *(uint4*)(b + rs) = zz;
                if (x0) *(uint4*)(b + rs - cs) = zz;
                if (x1) *(uint4*)(b + rs + cs) = zz;
              }
            }
          }
        }
      }
    }
#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
    __syncthreads();
    if (PS_DIAG_THIS && threadIdx.x == 0 && blockIdx.x < 16384) g_ps_stamps[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }
  for (int idx = threadIdx.x; idx < rows_out * CH; idx += NT) {
    const int orow = idx / CH, c = idx - orow * CH;
    const int co0 = tile_n * BN + c * 8;
    const int orr = orow / OC, occ = orow - orr * OC;
    const int g = g0 + (p.pool ? 2 * orr : orr);
    if (g >= total_rows || co0 >= p.Cout) continue;
    float v[8];
    uint4 o;
    if (p.pool) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = -INFINITY;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int ml = (2 * orr + (d >> 1)) * TC + 2 * occ + (d & 1);
        const uint4 u = *(const uint4*)(smem + ml * CROW + c * 16);
        const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = fmaxf(v[2 * e], h16_lo(w4[e]));
          v[2 * e + 1] = fmaxf(v[2 * e + 1], h16_hi(w4[e]));
        }
      }
    } else {
      const uint4 u = *(const uint4*)(smem + orow * CROW + c * 16);
      const unsigned w4[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = h16_lo(w4[e]);
        v[2 * e + 1] = h16_hi(w4[e]);
      }
    }
    if (has_post) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * psc[e] + psh[e];
    }
    const int n = img_of_row(g);
    const int y = g - n * p.H;
    const int yo = p.pool ? (y >> 1) : y;
    const int xo = (p.pool ? xb >> 1 : xb) + occ;
    const size_t ooff = (((size_t)n * p.Hyp + yo + p.y_halo) * p.Wyp + xo + p.y_halo) * (size_t)p.Cy + co0;
    const bool no_z = TAG == 0 && p.epi == VNQA_EPI_FILM_RES && p.y == nullptr;     // forward-only FiLM block: z is not kept
    vnqa_bf16* dst = no_z ? nullptr : (vnqa_bf16*)(p.y) + ooff;
    if (TAG == 0 && p.epi == VNQA_EPI_ADD_MASK) {
      // y = (conv + add) * [mask > 0] on the storage-rounded conv output (v holds exactly the 16-bit values staged in LDS):
      // the FiLM block's dgrad joined with the residual branch's gradient and masked by the 1x1 conv's ReLU
      const uint4 araw = *(const uint4*)((const vnqa_bf16*)p.res + ooff);
      const uint4 mraw = *(const uint4*)((const vnqa_bf16*)p.y2 + ooff);
      const unsigned aw[4] = {araw.x, araw.y, araw.z, araw.w}, mw[4] = {mraw.x, mraw.y, mraw.z, mraw.w};
      float w[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        w[2 * e] = h16_lo(mw[e]) > 0.f ? v[2 * e] + h16_lo(aw[e]) : 0.f;
        w[2 * e + 1] = h16_hi(mw[e]) > 0.f ? v[2 * e + 1] + h16_hi(aw[e]) : 0.f;
      }
      o.x = pack2_h16(w[0], w[1]); o.y = pack2_h16(w[2], w[3]); o.z = pack2_h16(w[4], w[5]); o.w = pack2_h16(w[6], w[7]);
    } else {
      o.x = pack2_h16(v[0], v[1]); o.y = pack2_h16(v[2], v[3]); o.z = pack2_h16(v[4], v[5]); o.w = pack2_h16(v[6], v[7]);
    }
    if (!no_z) *(uint4*)dst = o;
    vnqa_bf16* second = nullptr;
    if (TAG == 0 && p.epi == VNQA_EPI_FILM_RES) {
      // out2 = relu(gamma[n] * z + beta[n]) + res from the storage-rounded z just written (film_attn_pt_stem.py:229-241)
      const float* gp = p.film_gamma + (size_t)n * p.film_ld + co0;
      const float* bp = p.film_beta + (size_t)n * p.film_ld + co0;
      const uint4 rraw = *(const uint4*)((const vnqa_bf16*)p.res + ooff);
      const unsigned rw[4] = {rraw.x, rraw.y, rraw.z, rraw.w};
      float w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = co0 + e < p.film_c;
        const float ga = ok ? gp[e] : 0.f, be = ok ? bp[e] : 0.f;
        const float r = (e & 1) ? h16_hi(rw[e >> 1]) : h16_lo(rw[e >> 1]);
        w[e] = fmaxf(ga * v[e] + be, 0.f) + r;
      }
      uint4 o2;
      o2.x = pack2_h16(w[0], w[1]); o2.y = pack2_h16(w[2], w[3]); o2.z = pack2_h16(w[4], w[5]); o2.w = pack2_h16(w[6], w[7]);
      second = (vnqa_bf16*)p.y2 + ooff;
      *(uint4*)second = o2;
    }
    if (p.zero_halo) {
      // the halo ring of a fresh output buffer: every border pixel's thread also zeroes the halo positions next to it (corners
      // by the corner pixels) for its 16-byte channel chunk — for y and, with FILM_RES, for the second output
      const int Ho = p.pool ? p.H >> 1 : p.H, Wo = p.pool ? p.W >> 1 : p.W;
      const uint4 zz = make_uint4(0u, 0u, 0u, 0u);
      const long long rs = (long long)p.Wyp * p.Cy, cs = p.Cy;
      const bool x0 = xo == 0, x1 = xo == Wo - 1, y0 = yo == 0, y1 = yo == Ho - 1;
      if (x0 | x1 | y0 | y1) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          vnqa_bf16* b = which == 0 ? dst : second;
          if (b == nullptr) continue;
          if (x0) *(uint4*)(b - cs) = zz;
          if (x1) *(uint4*)(b + cs) = zz;
          if (y0) {
            *(uint4*)(b - rs) = zz;
            if (x0) *(uint4*)(b - rs - cs) = zz;
            if (x1) *(uint4*)(b - rs + cs) = zz;
          }
          if (y1) {
            *(uint4*)(b + rs) = zz;
            if (x0) *(uint4*)(b + rs - cs) = zz;
            if (x1) *(uint4*)(b + rs + cs) = zz;
          }
        }
      }
    }
  }
#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
  __syncthreads();
  if (PS_DIAG_THIS && threadIdx.x == 0 && blockIdx.x < 16384) g_ps_stamps[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int TC, int HALO, int TAG>
int launch_ps(const ConvArgs& a, hipStream_t stream) {
  using namespace ps;
  constexpr int TR = BM / TC;
  constexpr int LDS_BYTES = 2 * patch_rows(HALO) * 128 + 2 * B_BYTES;
  ConvArgs p = a;
  const int rows = p.n_img * p.H;
  const int tilesM = ((rows + TR - 1) / TR) * ((p.W + TC - 1) / TC);
  p.tilesN = (p.Cout + BN - 1) / BN;
  auto kern = conv_ps_kernel<TC, HALO, TAG>;
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

#if defined(VNQA_PS_DIAG) && VNQA_PS_DIAG == 3
extern "C" int vnqa_ps_diag_stamps(unsigned long long* out, int n_words) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ps_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif

// Geometry the patch-stationary tiles serve (the ONE copy of the test: vnqa_conv_ps_dispatch and the exported predicate
// vnqa_conv_ps_supported both call it).  tc: 28-wide tiles for widths that are multiples of 28, else 14-wide ones (any even width
// >= 14: the last column block overlaps its neighbour).
static int ps_geometry(int H, int W, int Cin, int Cout, int taps, int pool, int& tc) {
  using namespace ps;
  const int halo = taps == 25 ? 2 : 1;
  tc = W % 28 == 0 ? 28 : ((W >= 14 && (W % 14 == 0 || W % 2 == 0)) ? 14 : 0);
  if (tc == 0) {
    vnqa_set_error("conv patch-stationary tile: width %d must be even and >= 14 (or a multiple of 14)", W);
    return VNQA_ERR_UNSUPPORTED;
  }
  const int tr = BM / tc;
  if (halo == 2) {
    // 5x5: the (tr + 4) x (tc + 4) patch fills the buffer exactly: tiles must not straddle images
    if (H % tr != 0) {
      vnqa_set_error("conv patch-stationary tile (5x5): height %d is not a multiple of the %d tile rows", H, tr);
      return VNQA_ERR_UNSUPPORTED;
    }
  } else {
    // patch rows needed at worst: tile rows + 2 halo rows + 2 per image boundary the tile can straddle
    const int max_cross = (tr - 1 + H - 1) / H;
    if ((tr + 2 + 2 * max_cross) * (tc + 2) > patch_rows(1)) {
      vnqa_set_error("conv patch-stationary tile: %dx%d images do not fit the %d-pixel LDS patch", H, W, patch_rows(1));
      return VNQA_ERR_UNSUPPORTED;
    }
  }
  if (pool && (H % 2 != 0 || W % 2 != 0)) {
    vnqa_set_error("conv patch-stationary tile: pool2 needs even h, w");
    return VNQA_ERR_UNSUPPORTED;
  }
  const int Hp = H + 2 * halo, Wp = W + 2 * halo;
  if ((size_t)(Hp + 4) * Wp * Cin * 2 * 4 >= (1ull << 32) || (size_t)Cout * taps * Cin * 2 >= (1ull << 32)) {
    vnqa_set_error("conv patch-stationary tile: tensor too large for its 32-bit DMA offsets");
    return VNQA_ERR_UNSUPPORTED;
  }
  return VNQA_OK;
}

// 1 when the patch-stationary tiles (VNQA_TILE_PS_224x256 / VNQA_TILE_STEM_PS_224x256) serve this descriptor's conv — callers
// choose between them and the implicit-GEMM tiles with it instead of restating the geometry test (the reason is left in
// vnqa_last_error when the answer is 0)
extern "C" int vnqa_conv_ps_supported(const vnqa_conv_desc* d) {
  if (!d || d->dtype != VNQA_BF16 || (d->taps != 9 && d->taps != 25) || d->depth != 0 || d->wt_tiled || d->relu == 2) return 0;
  if (d->x_halo != (d->taps == 25 ? 2 : 1) || d->c_in < 64 || d->c_in % 64 != 0 || d->h <= 0 || d->w <= 0 || d->n_img <= 0) return 0;
  int tc = 0;
  return ps_geometry(d->h, d->w, d->c_in, d->c_out, d->taps, d->pool2, tc) == VNQA_OK ? 1 : 0;
}

int vnqa_conv_ps_dispatch(const ConvArgs& a, int tag, hipStream_t st) {
  using namespace ps;
  const int halo = a.taps == 25 ? 2 : 1;
  if (a.dual_out) {        // [hi | lo] pair output (TAG 2 instantiations: fp32 epilogue in two cout passes)
    if (a.taps != 9 || a.epi != VNQA_EPI_NONE || a.border_sub != nullptr || a.zero_halo || a.Cout % 8 != 0 ||
        a.Cy < ((a.dual_out == 4 || a.dual_out == 8) ? 1 : (a.dual_out == 9 ? 2 : a.dual_out + 1)) * a.Cout || (a.dual_out == 4 && a.y2 == nullptr)) {
      vnqa_set_error("conv patch-stationary tile: VNQA_CONV_DUAL_OUT needs a plain 3x3 conv (no fused epilogue / border correction / "
                     "halo zeroing), c_out %% 8 == 0 and c_y >= 2 c_out (3 c_out with VNQA_CONV_DUAL_HI2)");
      return VNQA_ERR_UNSUPPORTED;
    }
  }
  if ((a.taps != 9 && a.taps != 25) || a.D != 0 || a.x_halo != halo || a.wt_tiled || a.partial != nullptr || a.Cin % 64 != 0 ||
      a.Cin < 64 || a.group_tiles != 0 || a.ring_h != 0 ||
      (a.border_sub != nullptr && (a.Cout % 16 != 0 || (long long)a.n_img * (2 * a.W + 2 * (a.H - 2)) * a.Cout * 2 >= (1ll << 32))) ||
      !(a.epi == VNQA_EPI_NONE || ((a.epi == VNQA_EPI_FILM_RES || a.epi == VNQA_EPI_ADD_MASK) && tag == 0 && !a.pool && a.y_halo == 1)) ||
      (a.zero_halo && a.y_halo != 1) || a.relu == 2) {
    vnqa_set_error("conv patch-stationary tile: needs a bf16 3x3 / 5x5 2-D conv, x_halo = 1 / 2, c_in %% 64 == 0, K-major weights; of the "
                   "fused epilogues FILM_RES and ADD_MASK (un-pooled, y_halo = 1, trunk tag)");
    return VNQA_ERR_UNSUPPORTED;
  }
  int tc = 0;
  const int rc = ps_geometry(a.H, a.W, a.Cin, a.Cout, a.taps, a.pool, tc);
  if (rc != VNQA_OK) return rc;
  if (halo == 1) {
    if (a.dual_out) return tc == 28 ? launch_ps<28, 1, 2>(a, st) : launch_ps<14, 1, 2>(a, st);
    if (tc == 28) return tag ? launch_ps<28, 1, 1>(a, st) : launch_ps<28, 1, 0>(a, st);
    return tag ? launch_ps<14, 1, 1>(a, st) : launch_ps<14, 1, 0>(a, st);
  }
  if (tc == 28) return tag ? launch_ps<28, 2, 1>(a, st) : launch_ps<28, 2, 0>(a, st);
  return tag ? launch_ps<14, 2, 1>(a, st) : launch_ps<14, 2, 0>(a, st);
}
