// conv_igemm.hip — implicit-GEMM 3x3 / 1x1 convolution on CDNA4 MFMA (gfx950).
//
// GEMM view:  D[cout][pixel] = sum_k Wt[cout][k] * X[pixel][k],  k = tap*Cin + ci.
// The weight tile is the MFMA "A" operand and the pixel tile the "B" operand, so the 32x32
// accumulator holds 4 CONSECUTIVE output channels of one pixel in 4 consecutive registers:
// the epilogue packs them and goes through LDS once to emit full 16-byte NHWC stores.
//
// Staging: every K-step is one (tap, 64-channel) slab = a 128-byte contiguous run per pixel
// (bf16; 32 channels for f32).  Tiles are copied HBM->LDS by global_load_lds_dwordx4
// (1 KiB = 8 rows x 128 B per wave instruction, LDS destination lane-linear).  The XOR
// swizzle (chunk ^= (row>>1)&7) is applied on the per-lane SOURCE address and again on the
// ds_read_b128 address (same involution), which makes the 32x32x16 fragment reads
// conflict-free (rows r and r+1 sit in the two 128-B halves of a 256-B bank row, the 8 row
// pairs of a 16-lane group land in 8 different 16-B slots).
// Zero padding costs nothing: activations live in "padded NHWC" buffers with a zero halo.
//
// Loop: 2-stage LDS double buffer, one barrier per K-step; the next slab's DMA is in flight
// while the current slab's MFMAs run.
#include <cstdlib>
#include "conv_args.h"

namespace {


// MFMA shape per element type.  bf16 uses v_mfma_f32_16x16x32_bf16 by default: same FLOPs per LDS byte and
// per cycle as 32x32x16, but the chip holds a higher clock on it under load (MI355X_MICROARCH.md, DVFS
// give-back item 7).  -DVNQA_MFMA32 selects 32x32x16 for A/B runs.
#ifdef VNQA_MFMA32
constexpr int kBf16Mt = 32;
#else
constexpr int kBf16Mt = 16;
#endif
template <typename T> struct Mma;
template <> struct Mma<vnqa_bf16> {
  static constexpr int MT = kBf16Mt;
  static __device__ __forceinline__ void run(const vnqa_f32x4& a, const vnqa_f32x4& b, vnqa_f32x16& c) {
    c = VNQA_MFMA_32x32x16(__builtin_bit_cast(vnqa_bf16x8, a),
                                                __builtin_bit_cast(vnqa_bf16x8, b), c);
  }
  static __device__ __forceinline__ void run(const vnqa_f32x4& a, const vnqa_f32x4& b, vnqa_f32x4& c) {
    c = VNQA_MFMA_16x16x32(__builtin_bit_cast(vnqa_bf16x8, a),
                                                __builtin_bit_cast(vnqa_bf16x8, b), c);
  }
};
template <int MT> struct AccOf { typedef vnqa_f32x16 type; };
template <> struct AccOf<16> { typedef vnqa_f32x4 type; };
template <> struct Mma<float> {
  static constexpr int MT = 32;
  // 16 bytes = 4 f32 k-values per lane; the k permutation is the same for A and B.
  static __device__ __forceinline__ void run(const vnqa_f32x4& a, const vnqa_f32x4& b, vnqa_f32x16& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], c, 0, 0, 0);
  }
};

__device__ __forceinline__ void glds16(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// LDS-DMA from inline asm (64-bit per-lane source address, wave-uniform LDS destination): invisible to hipcc's wait-count
// bookkeeping, so it can stay in flight across raw barriers and next to LDS reads; every wait for it is hand-placed
// (PIPE == 5 main loop).  M0 is compiler-reserved: saved and restored inside the statement.
__device__ __forceinline__ void glds16_asm64(const char* src, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}
// Rejected experiment, kept for A/B: non-temporal (aux = 2) weight-tile DMA measured -8 % on conv12/conv22
// (every CU re-reads the weight tiles from L2; nt gives that reuse up).
#ifdef VNQA_NT_WEIGHTS
#ifndef VNQA_W_AUX
#define VNQA_W_AUX 2
#endif
__device__ __forceinline__ void glds16w_(const char* src, char* lds_wave_base) {   // cache-policy experiment
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, VNQA_W_AUX);
}
#else
#define glds16w_ glds16
#endif

// conv-output pixel index m -> (n, y, x); pooled layers enumerate pixels quad-major so that
// the four members of a 2x2 pooling window are consecutive m.
__device__ __forceinline__ void decode_pixel(int m, int H, int W, int pool, int& n, int& y, int& x) {
  if (pool) {
    const int q = m >> 2, d = m & 3;
    const int W2 = W >> 1, H2 = H >> 1;
    n = q / (H2 * W2);
    const int rem = q - n * (H2 * W2);
    const int yo = rem / W2;
    y = 2 * yo + (d >> 1);
    x = 2 * (rem - yo * W2) + (d & 1);
  } else {
    n = m / (H * W);
    const int rem = m - n * (H * W);
    y = rem / W;
    x = rem - y * W;
  }
}

// TAG only gives the frozen-stem launches their own kernel symbol (so that profiles and the
// bench's roofline line can name "the stem igemm" apart from the trunk's uses of the template).
//
// PIPE selects the main-loop structure:
//   PIPE == 2 : two 128-byte-row stages, __syncthreads() (drains the DMA) once per 64-channel K-step;
//   PIPE == 5 : two 128-byte-row stages like PIPE == 2, but software-pipelined by hand (see the branch below): one raw
//               barrier per K-step placed BETWEEN its two k-substeps, DMA two stages ahead issued piece by piece between
//               MFMA groups, fragment reads one substep ahead, one per MFMA gap; no wave stagger.
//   PIPE == 4 : a ring of four 64-byte-row stages (32 channels each); the DMA of three stages stays in
//               flight ACROSS the workgroup barriers: counted s_waitcnt vmcnt(N) + raw s_barrier, one
//               barrier per stage, a slot is refilled right after the barrier that retires its readers.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int TAG = 0, int PIPE = 2>
__global__ void __launch_bounds__(WAVES_M* WAVES_N * 64) conv_igemm_kernel(const ConvArgs p) {
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int NT = NW * 64;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N;
  constexpr int MT = Mma<T>::MT;              // MFMA tile side: 32 (32x32x16 / 32x32x2) or 16 (16x16x32)
  constexpr int TM = WTM / MT, TN = WTN / MT;
  constexpr int NR = MT == 32 ? 16 : 4;       // accumulator registers per MFMA tile
  constexpr int NG = NR / 4;                  // groups of 4 consecutive couts per lane and tile
  constexpr int CPS = MT == 32 ? 2 : 4;       // 16-byte chunks one k-substep spans (lane takes chunk fh of them)
  constexpr int ES = (int)sizeof(T);
  constexpr int ROWB = (PIPE == 3 || PIPE == 4) ? 64 : 128;  // bytes of one tile row per stage
  constexpr int BK = ROWB / ES;               // channels per stage
  constexpr int CPR = ROWB / 16;              // 16-byte chunks per row
  constexpr int RPI = 1024 / ROWB;            // rows covered by one wave-level DMA instruction
  constexpr int NSUB = CPR / CPS;             // k-substeps per stage
  typedef typename AccOf<MT>::type acc_t;
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int A_PER_WAVE = (BM / RPI) / NW, B_PER_WAVE = (BN / RPI) / NW;
  constexpr int LPS = A_PER_WAVE + B_PER_WAVE;  // DMA instructions per wave per stage
  constexpr int EPC = 16 / ES;                // elements per 16-byte chunk
  constexpr int CROW = BN * ES + 16;          // epilogue LDS row stride (bytes)
  static_assert((BM / RPI) % NW == 0 && (BN / RPI) % NW == 0, "tile/wave mismatch");
  static_assert(TM >= 1 && TN >= 1, "wave tile too small");
  static_assert(PIPE == 2 || PIPE == 3 || PIPE == 4 || PIPE == 5, "PIPE must be 2, 3, 4 or 5");
  // swizzle: spread the 16 rows a ds_read_b128 lane group touches over all 16 slots of a 256-B bank row
  auto swz = [](int row) { return ROWB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // XCD-aware, bijective remap: blocks that share an XCD (bid % 8) get a contiguous run of
  // tiles, n-tile fastest, so neighbouring tiles share pixel rows / weight panels in one L2.
  int tile_m, tile_n, slice;
  {
    const int nwg = gridDim.x / p.slices;
    slice = blockIdx.x / nwg;
    const int bid = blockIdx.x - slice * nwg;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    tile_n = swz % p.tilesN;
    tile_m = swz / p.tilesN;
    if constexpr (TAG == 1 || TAG == 5 || TAG == 6) {
      // VNQA_CONV_XCD_SPLIT_N (stem launches with two cout tiles: the composed 5x5 conv): XCD x owns cout half x & 1 ONLY, the pixel
      // tiles of that half dealt to the four XCDs of its parity in contiguous runs.  An XCD's L2 (4 MiB) then holds ONE half of the
      // weight set (1.65 of the composed conv's 3.3 MB) next to its 32 tiles' activation windows instead of thrashing on both
      // halves; the price is that each activation window is fetched by two XCDs.  Bijective: hardware deals block b to XCD b % 8,
      // so the XCDs of one parity hold 4q + r/2 = nwg/2 blocks (nwg = 2 tilesM is even, hence r is).
      if (p.xcd_split && p.tilesN == 2 && (nwg & 1) == 0) {
        const int par = xcd & 1, j = xcd >> 1;
        int off = 0;
        for (int i = 0; i < j; ++i) off += q + ((2 * i + par) < r ? 1 : 0);
        tile_n = par;
        tile_m = off + (bid >> 3);
      }
    }
  }

  const int kchunks = p.Cin / BK;
  const int KT = p.taps * kchunks;
  // TAG 4 (VNQA_CONV_X_WRAP2): the activation tensor has Cin / 2 PHYSICAL channels and is read twice along K — channel chunks
  // [0, Cin/2) and [Cin/2, Cin) of the contraction both come from it, against a weight operand [w_hi | w_lo]: the two-product form
  // x . w_hi + x . w_lo of a 16-bit activation with split weights (precision 'fp16w'), with no duplicated copy of x in memory.
  const int CX = TAG == 4 ? p.Cin / 2 : p.Cin;          // pixel stride of x in elements
  const size_t w_row_bytes = (size_t)p.taps * p.Cin * ES;

  // ---- per-lane source offsets (constant over the K loop) ----
  size_t a_off[A_PER_WAVE];
  size_t b_off[B_PER_WAVE];
#pragma unroll
  for (int j = 0; j < A_PER_WAVE; ++j) {
    const int row = (wave * A_PER_WAVE + j) * RPI + lane / CPR;
    int m = tile_m * BM + row;
    m = m < p.M ? m : p.M - 1;
    int n, y, x;
    if (p.ring_h > 0) {
      // ring position q = (qy, qx), qy in [-1, H], qx in [-1, W] (top row, bottom row, left column, right column); its 3x3
      // window's top-left tap sits at (qy + 1, qx + 1) of the halo-2 padded image
      const int R = 2 * (p.ring_w + 2) + 2 * p.ring_h;
      n = m / R;
      const int r = m - n * R;
      int qy, qx;
      if (r < p.ring_w + 2) { qy = -1; qx = r - 1; }
      else if (r < 2 * (p.ring_w + 2)) { qy = p.ring_h; qx = r - (p.ring_w + 2) - 1; }
      else if (r < 2 * (p.ring_w + 2) + p.ring_h) { qy = r - 2 * (p.ring_w + 2); qx = -1; }
      else { qy = r - 2 * (p.ring_w + 2) - p.ring_h; qx = p.ring_w; }
      y = qy + 1;
      x = qx + 1;
    } else {
      decode_pixel(m, p.H, p.W, p.pool, n, y, x);
    }
    size_t img = n;
    if (p.D > 0) {   // depth slice (nn, d): top-front-left tap sits at padded depth d
      const int nn = n / p.D;
      img = (size_t)nn * (p.D + 2) + (n - nn * p.D);
    }
    const int lc = (lane % CPR) ^ swz(row);
    a_off[j] = ((img * p.Hp + y) * p.Wp + x) * (size_t)CX * ES + (size_t)lc * 16;
  }
#pragma unroll
  for (int j = 0; j < B_PER_WAVE; ++j) {
    const int row = (wave * B_PER_WAVE + j) * RPI + lane / CPR;
    int co = tile_n * BN + row;
    co = co < p.Cout ? co : p.Cout - 1;
    if (p.group_tiles > 0) co += (tile_m / p.group_tiles) * p.Cout;   // grouped GEMM: this pixel tile's own weight matrix
    const int lc = (lane % CPR) ^ swz(row);
    b_off[j] = (size_t)co * w_row_bytes + (size_t)lc * 16;
  }

  // K order: channel chunk OUTER, tap INNER.  Nine consecutive stages then re-read the same
  // 128-byte runs of neighbouring pixels (L1/L2 hits); tap-major order would revisit a pixel row
  // only after Cin/BK stages, by which time a 4 MiB XCD L2 shared by 32 workgroups has evicted it.
  auto stage = [&](int kt, int buf) {
    const int kc = kt / p.taps;
    const int tap = kt - kc * p.taps;
    int r, s, q = 0;
    if (p.taps == 9) {
      r = tap / 3;
      s = tap - 3 * r;
    } else if (p.taps == 25) {       // 5x5, input halo 2
      r = tap / 5;
      s = tap - 5 * r;
    } else if (p.taps == 27) {
      q = tap / 9;
      const int rs = tap - 9 * q;
      r = rs / 3;
      s = rs - 3 * r;
    } else if (p.taps == 3 || p.taps == 5) {   // 1x3 / 1x5 window along a row, or 3x1 / 5x1 down a column (the border-correction edge launches)
      r = p.tap3_vertical ? tap : 0;
      s = p.tap3_vertical ? 0 : tap;
    } else {
      r = p.x_halo;
      s = p.x_halo;
    }
    const int kcx = (TAG == 4 && kc >= (kchunks >> 1)) ? kc - (kchunks >> 1) : kc;
    const size_t tapoff = ((((size_t)q * p.Hp + r) * p.Wp + s) * CX + (size_t)kcx * BK) * ES;
    const size_t woff = ((size_t)tap * p.Cin + (size_t)kc * BK) * ES;
    char* lds = smem + buf * STAGE_BYTES;
#ifdef VNQA_DIAG_SKIP_DMA   // timing-only diagnostic build: drop one operand's DMA after the first stage
    const bool skipA = (p.relu & 256) && kt != 0, skipB = (p.relu & 512) && kt != 0;
#else
    constexpr bool skipA = false, skipB = false;
#endif
    if (!skipA) {
#pragma unroll
      for (int j = 0; j < A_PER_WAVE; ++j)
        glds16(p.x + a_off[j] + tapoff, lds + (wave * A_PER_WAVE + j) * 1024);
    }
    if (!skipB) {
      if (p.wt_tiled) {
        // contiguous image of this (cout tile, stage): lane reads exactly the bytes it deposits
        const char* src = p.wt + ((size_t)tile_n * KT + kt) * B_BYTES + (size_t)lane * 16;
#pragma unroll
        for (int j = 0; j < B_PER_WAVE; ++j)
          glds16w_(src + (wave * B_PER_WAVE + j) * 1024, lds + A_BYTES + (wave * B_PER_WAVE + j) * 1024);
      } else {
#pragma unroll
        for (int j = 0; j < B_PER_WAVE; ++j)
          glds16w_(p.wt + b_off[j] + woff, lds + A_BYTES + (wave * B_PER_WAVE + j) * 1024);
      }
    }
  };

  acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < NR; ++e) acc[i][j][e] = 0.f;

  // fragment read addresses: row = lane % MT within an MT-row sub-tile, k-chunk selector fh = lane / MT
  const int fr = lane & (MT - 1), fh = lane / MT;
  int x_rd[TM], w_rd[TN], x_sw[TM], w_sw[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int row = wm * WTM + i * MT + fr;
    x_rd[i] = row * ROWB;
    x_sw[i] = swz(row);
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * WTN + j * MT + fr;
    w_rd[j] = A_BYTES + row * ROWB;
    w_sw[j] = swz(row);
  }

  const int kt0 = slice * p.kt_per_slice;
#ifdef VNQA_DIAG_SKIP_DMA   // 2048 = main loop cut to 2 K-steps (what is left is the per-tile fixed cost)
  const int kt1 = (p.relu & 2048) ? kt0 + 2 : ((kt0 + p.kt_per_slice < KT) ? kt0 + p.kt_per_slice : KT);
#else
  const int kt1 = (kt0 + p.kt_per_slice < KT) ? kt0 + p.kt_per_slice : KT;
#endif

  // One stage = NSUB k-substeps.  Fragments are double-buffered in registers: the ds_read_b128s of
  // substep s+1 are issued BEFORE the MFMAs of substep s, so LDS latency hides behind the matrix pipe
  // of the same wave instead of relying on the SIMD's other wave.
  auto load_frags = [&](const char* lds, int s, vnqa_f32x4* xf, vnqa_f32x4* wf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
      xf[i] = *(const vnqa_f32x4*)(lds + x_rd[i] + (((CPS * s + fh) ^ x_sw[i]) << 4));
#pragma unroll
    for (int j = 0; j < TN; ++j)
      wf[j] = *(const vnqa_f32x4*)(lds + w_rd[j] + (((CPS * s + fh) ^ w_sw[j]) << 4));
  };
  auto compute = [&](const char* lds) {
    if constexpr (NW == 16) {     // 4 waves per SIMD: 128-VGPR budget, the other waves hide the LDS latency
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        vnqa_f32x4 xf1[TM], wf1[TN];
        load_frags(lds, s, xf1, wf1);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(wf1[j], xf1[i], acc[i][j]);
      }
      return;
    }
#ifdef VNQA_NO_FRAG_PREFETCH
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      vnqa_f32x4 xf1[TM], wf1[TN];
      load_frags(lds, s, xf1, wf1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(wf1[j], xf1[i], acc[i][j]);
    }
    return;
#endif
    vnqa_f32x4 xf[2][TM], wf[2][TN];
    load_frags(lds, 0, xf[0], wf[0]);
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      if (s + 1 < NSUB) load_frags(lds, s + 1, xf[(s + 1) & 1], wf[(s + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this substep's MFMAs
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(wf[s & 1][j], xf[s & 1][i], acc[i][j]);
    }
  };

  auto mma_sub = [&](vnqa_f32x4* xf, vnqa_f32x4* wf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) Mma<T>::run(wf[j], xf[i], acc[i][j]);
  };

#ifndef VNQA_NO_STAGGER
#ifdef VNQA_STAGGER_512
  constexpr bool kStagger = (PIPE == 2 && NSUB == 2 && NW == 8 && (WAVES_M == 2 || BM == 512));
#else
  constexpr bool kStagger = (PIPE == 2 && NSUB == 2 && NW == 8 && WAVES_M == 2);   // 256x256 tile only: measured -4 % on the 4x2-wave 256x128 tile
#endif
#else
  constexpr bool kStagger = false;
#endif
  if constexpr (kStagger) {
    // Two waves share every SIMD (wave w and w+4).  Waves 4-7 run half a stage behind: right after the
    // barrier they issue the MFMAs of substep 1 of the PREVIOUS stage (fragments already in registers)
    // while waves 0-3 wait on their first LDS reads; the SIMD's matrix pipe and its LDS port are then used
    // by different waves at any time instead of both waves stalling together.  Both variants execute the
    // same barrier sequence; late waves never touch a slot after the barrier that allows its refill.
    stage(kt0, 0);
    __syncthreads();
    vnqa_f32x4 xf0[TM], wf0[TN], xf1[TM], wf1[TN];
    if (wave < 4) {
      for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const char* lds = smem + cur * STAGE_BYTES;
        if (kt + 1 < kt1) stage(kt + 1, cur ^ 1);
        load_frags(lds, 0, xf0, wf0);
        load_frags(lds, 1, xf1, wf1);
        __builtin_amdgcn_sched_barrier(0);
        mma_sub(xf0, wf0);
        mma_sub(xf1, wf1);
        __syncthreads();
      }
    } else {
      for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        const char* lds = smem + cur * STAGE_BYTES;
        if (kt + 1 < kt1) stage(kt + 1, cur ^ 1);
        if (kt > kt0) mma_sub(xf1, wf1);                 // substep 1 of stage kt-1 (registers only)
        __builtin_amdgcn_sched_barrier(0);
        load_frags(lds, 0, xf0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        mma_sub(xf0, wf0);
        __builtin_amdgcn_sched_barrier(0);
        load_frags(lds, 1, xf1, wf1);                    // lands in registers before the barrier below
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                 // (its fence waits for lgkmcnt(0) and vmcnt(0))
      }
      mma_sub(xf1, wf1);
    }
  } else if constexpr (PIPE == 5) {
    // Hand-pipelined loop (256x256 tile, 8 waves, 128x64 wave tiles, 16x16x32 MFMA).  Per K-step kt and wave:
    //   phase 0: the 32 MFMAs of k-substep 0 (fragments already in registers), the 12 fragment reads of substep 1 behind them;
    //   --- s_waitcnt vmcnt(0) [stage kt+1, issued a whole K-step ago, has landed] + lgkmcnt(0) [every fragment of stage kt is
    //       in registers], raw s_barrier: slot kt%2 is free, slot (kt+1)%2 is visible ---
    //   phase 1: the 32 MFMAs of substep 1; behind them the 8 DMA instructions of stage kt+2 (into the slot just freed) and
    //            the 12 fragment reads of stage kt+1 / substep 0.
    // A DMA thus has a full K-step to land and its wait never stalls; a barrier never has a DMA burst or a fragment-read
    // burst behind it; the matrix pipe always has queued work while this wave issues an expensive instruction.
    static_assert(NSUB == 2 && NW == 8 && MT == 16 && TM == LPS, "PIPE 5 is laid out for the 256x256 tile");
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // K position of a stage (tap = (q, r, sx) window coordinates, kc = 64-channel chunk), advanced incrementally: the
    // per-stage offsets cost a handful of scalar instructions instead of a division chain behind the barrier
    const bool line = p.taps == 3 || p.taps == 5;      // a 1 x taps (or taps x 1) window
    const int kw = (p.taps == 9 || p.taps == 27) ? 3 : (p.taps == 25 ? 5 : ((line && !p.tap3_vertical) ? p.taps : 1));
    const int kh = (p.taps == 9 || p.taps == 27) ? 3 : (p.taps == 25 ? 5 : ((line && p.tap3_vertical) ? p.taps : 1));
    const int r_base = (p.taps == 9 || p.taps == 25 || p.taps == 27 || line) ? 0 : p.x_halo;   // 1x1: the centre tap
    struct KPos { int tap, kc, q, r, sx; };
    auto kpos_at = [&](int kt) {
      KPos k;
      k.kc = kt / p.taps;
      k.tap = kt - k.kc * p.taps;
      k.q = k.tap / (kw * kh);
      const int rs = k.tap - k.q * (kw * kh);
      k.r = rs / kw;
      k.sx = rs - k.r * kw;
      return k;
    };
    auto kpos_next = [&](KPos& k) {
      ++k.tap;
      if (++k.sx == kw) { k.sx = 0; if (++k.r == kh) { k.r = 0; ++k.q; } }
      if (k.tap == p.taps) { k.tap = 0; k.sx = 0; k.r = 0; k.q = 0; ++k.kc; }
    };
    auto stage_off = [&](const KPos& k, size_t& tapoff, size_t& woff) {
      tapoff = ((((size_t)k.q * p.Hp + (k.r + r_base)) * p.Wp + (k.sx + r_base)) * p.Cin + (size_t)k.kc * BK) * ES;
      woff = ((size_t)k.tap * p.Cin + (size_t)k.kc * BK) * ES;
    };
    auto dma_piece = [&](size_t tapoff, size_t woff, int kt, int buf, int j) {
      if (j < A_PER_WAVE) {
        glds16_asm64(p.x + a_off[j < A_PER_WAVE ? j : 0] + tapoff,
                     __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + (wave * A_PER_WAVE + j) * 1024));
      } else {
        const int jj = j - A_PER_WAVE;
        const char* src = p.wt_tiled ? p.wt + ((size_t)tile_n * KT + kt) * B_BYTES + (size_t)lane * 16 + (wave * B_PER_WAVE + jj) * 1024
                                     : p.wt + b_off[jj >= 0 && jj < B_PER_WAVE ? jj : 0] + woff;
        glds16_asm64(src, __builtin_amdgcn_readfirstlane(lds0 + buf * STAGE_BYTES + A_BYTES + (wave * B_PER_WAVE + jj) * 1024));
      }
    };
    // fragment read idx (0 .. TM+TN-1) of substep s: pixel fragments first, then weight fragments
    auto read_frag = [&](const char* lds, int s, int idx, vnqa_f32x4* xf, vnqa_f32x4* wf) {
      if (idx < TM) xf[idx] = *(const vnqa_f32x4*)(lds + x_rd[idx] + (((CPS * s + fh) ^ x_sw[idx]) << 4));
      else wf[idx - TM] = *(const vnqa_f32x4*)(lds + w_rd[idx - TM] + (((CPS * s + fh) ^ w_sw[idx - TM]) << 4));
    };
    // the 12 reads behind the 8 MFMA groups of a phase: two behind each of the first four groups, one behind the others
    auto reads_behind = [&](const char* lds, int s, int g, vnqa_f32x4* xf, vnqa_f32x4* wf) {
      if (g < 4) { read_frag(lds, s, 2 * g, xf, wf); read_frag(lds, s, 2 * g + 1, xf, wf); }
      else read_frag(lds, s, 4 + g, xf, wf);
    };
    size_t tapoff, woff;
    KPos kp = kpos_at(kt0);
    stage_off(kp, tapoff, woff);
    asm volatile("s_nop 4" ::: "memory");
#pragma unroll
    for (int j = 0; j < LPS; ++j) dma_piece(tapoff, woff, kt0, 0, j);
    kpos_next(kp);
    if (kt0 + 1 < kt1) {
      stage_off(kp, tapoff, woff);
#pragma unroll
      for (int j = 0; j < LPS; ++j) dma_piece(tapoff, woff, kt0 + 1, 1, j);
    }
    kpos_next(kp);                       // kp = position of stage kt + 2 from here on
    stage_off(kp, tapoff, woff);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    vnqa_f32x4 xf0[TM], wf0[TN], xf1[TM], wf1[TN];
#pragma unroll
    for (int idx = 0; idx < TM + TN; ++idx) read_frag(smem, 0, idx, xf0, wf0);
    // timing-only diagnostics of this loop (compile-time, so that the rest of the code is generated as in the product build):
    // -DVNQA_P5_DIAG=<bits>: 1 = no pixel DMA, 2 = no weight DMA, 4 = no fragment reads, 8 = no barrier, 16 = no MFMAs
#ifndef VNQA_P5_DIAG
#define VNQA_P5_DIAG 0
#endif
    constexpr bool dgA = VNQA_P5_DIAG & 1, dgB = VNQA_P5_DIAG & 2, dgR = VNQA_P5_DIAG & 4, dgBar = VNQA_P5_DIAG & 8, dgM = VNQA_P5_DIAG & 16;
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      const char* lds = smem + cur * STAGE_BYTES;
      const char* ldn = smem + (cur ^ 1) * STAGE_BYTES;
      const bool has1 = kt + 1 < kt1 && !dgR, has2 = kt + 2 < kt1;
      // ---- phase 0 ----
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) if (!dgM) Mma<T>::run(wf0[j], xf0[i], acc[i][j]);
        if (!dgR) reads_behind(lds, 1, i, xf1, wf1);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stage kt+1 (issued during phase 1 of K-step kt-1) has landed
      __builtin_amdgcn_s_waitcnt(0xC07F);                  // lgkmcnt(0): all of stage kt's fragments are in registers
      if (!dgBar) __builtin_amdgcn_s_barrier();
      // ---- phase 1 ----
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TN; ++j) if (!dgM) Mma<T>::run(wf1[j], xf1[i], acc[i][j]);
        if (has2 && !(i < A_PER_WAVE ? dgA : dgB)) dma_piece(tapoff, woff, kt + 2, cur, i);
        if (has1) reads_behind(ldn, 0, i, xf0, wf0);
      }
      kpos_next(kp);
      stage_off(kp, tapoff, woff);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();     // every wave is done with its fragment reads before the epilogue reuses the LDS
  } else if constexpr (PIPE == 2) {
    stage(kt0, 0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      if (kt + 1 < kt1) stage(kt + 1, cur ^ 1);
      compute(smem + cur * STAGE_BYTES);
      __syncthreads();
    }
  } else if constexpr (PIPE == 3) {
    // ring of 3 slots (stages kt+1, kt+2 in flight): 3/4 of the 4-slot ring's LDS, so that TWO workgroups of the
    // 256x128 tile share a CU and one's prologue / epilogue hides behind the other's main loop (short-K layers)
#pragma unroll
    for (int d = 0; d < 2; ++d)
      if (kt0 + d < kt1) stage(kt0 + d, d);
    int cur = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      const int nxt = cur == 0 ? 2 : cur - 1;               // slot of stage kt-1 == slot of stage kt+2
      if (kt + 2 < kt1) stage(kt + 2, nxt);
      __builtin_amdgcn_s_setprio(1);
      compute(smem + cur * STAGE_BYTES);
      __builtin_amdgcn_s_setprio(0);
      cur = cur == 2 ? 0 : cur + 1;
    }
    __builtin_amdgcn_s_barrier();
  } else {
    // ring of 4 slots; stages kt+1..kt+3 are in flight while stage kt is consumed
#pragma unroll
    for (int d = 0; d < 3; ++d)
      if (kt0 + d < kt1) stage(kt0 + d, d);
    for (int kt = kt0; kt < kt1; ++kt) {
      // retire stage kt: all but the DMA of the (up to two) younger issued stages must have landed
      if (kt + 2 < kt1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
      } else if (kt + 1 < kt1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();   // stage kt visible to every wave; everyone is done reading stage kt-1
      if (kt + 3 < kt1) stage(kt + 3, (kt + 3 - kt0) & 3);   // refill the slot stage kt-1 occupied
      __builtin_amdgcn_s_setprio(1);
      compute(smem + ((kt - kt0) & 3) * STAGE_BYTES);
      __builtin_amdgcn_s_setprio(0);
    }
    __builtin_amdgcn_s_barrier();     // all fragment reads done before the epilogue reuses the LDS
  }

#ifdef VNQA_DIAG_SKIP_DMA   // timing-only diagnostic: 1024 = no epilogue at all (one lane keeps the accumulators alive)
  if (p.relu & 1024) {
    float sink = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) sink += acc[i][j][0];
    if (sink == 12345.678f) ((float*)p.y)[0] = sink;
    return;
  }
#endif
  // ---------------- epilogue ----------------
  // acc[i][j][4g+e]: pixel = wm*WTM + i*MT + fr ; cout = wn*WTN + j*MT + (MT==32 ? 8g + 4fh : 4fh) + e
  auto col_of = [&](int j, int g) { return wn * WTN + j * MT + (MT == 32 ? 8 * g + 4 * fh : 4 * fh); };
  // (Measured and rejected: an LDS-free epilogue for the stem launches — 8-byte stores straight from the accumulators,
  // 2x2 pooling by quad lane exchange.  The 32-byte store segments cost more than the LDS round trip saves:
  // conv11 1.26 -> 1.56 ms, conv12 3.2 -> 3.7 ms.)
  if (p.partial != nullptr) {
    // split-K: raw fp32 partial sums, 4 consecutive couts per lane -> 16-byte stores
    float* slab = p.partial + (size_t)slice * p.M * p.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = tile_m * BM + wm * WTM + i * MT + fr;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int co = tile_n * BN + col_of(j, g);
          if (m < p.M && co < p.Cout)
            *(float4*)(slab + (size_t)m * p.Cout + co) =
                make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
    }
    return;
  }
  if constexpr (TAG == 5 || TAG == 6) {
    // (TAG 6 = the same epilogue with ONE plain output, VNQA_CONV_F32_EPILOGUE: its own instantiation, no runtime branch)
    // DUAL output on the implicit-GEMM tile (VNQA_CONV_DUAL_OUT [| VNQA_CONV_DUAL_HI2], VNQA_EPI_SPLIT_OUT; precision 'fp16h' on the
    // geometries the patch-stationary tiles do not serve — the 10 x 13 maps of the reference's 160 x 208 frames, eval/utils.py:24-25 —
    // and on the composed 5x5 conv): the fp32 result as a PAIR of 16-bit values hi = h16(v), lo = h16(v - hi), channel segments
    // [hi | lo (| hi)] of y or two plain tensors (y, y2) — the same values, bit for bit, as conv_ps_kernel<.., TAG 2> writes.
    // Everything is finished in fp32 (bias, border correction, ReLU, 2x2 max-pool, affine); the tile goes through LDS as fp32 in
    // passes of 128 couts (the wave columns of cout range `pass` stage, all waves store): the LDS bytes of the 16-bit epilogue.
    static_assert(ES == 2 && MT == 16, "dual epilogue: 16-bit instantiations on the 16x16x32 MFMA");
    constexpr int PASSES = BN > 128 ? BN / 128 : 1;
    constexpr int PN = BN / PASSES;                 // couts per pass
    constexpr int WPP = WAVES_N / PASSES;           // wave columns per pass
    static_assert(WAVES_N % PASSES == 0 && PN % 8 == 0 && WPP * WTN == PN, "dual epilogue: wave columns / passes mismatch");
    constexpr int CROWF = PN * 4 + 16;
    constexpr int CHF = PN / 8;                     // 8-channel chunks (32 B of fp32) per staged row
    const bool has_post_d = (p.post_scale != nullptr);
    const int rows_out_d = p.pool ? BM / 4 : BM;
    const int M_out_d = p.pool ? (p.M >> 2) : p.M;
    const int Hod = p.pool ? (p.H >> 1) : p.H, Wod = p.pool ? (p.W >> 1) : p.W;
    int ring_row_d[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      ring_row_d[i] = -1;
      if (p.border_sub != nullptr) {
        const int m = tile_m * BM + wm * WTM + i * MT + fr;
        if (m < p.M) {
          int n, y, x;
          decode_pixel(m, p.H, p.W, p.pool, n, y, x);
          int ring = -1;
          if (y == 0) ring = x;
          else if (y == p.H - 1) ring = p.W + x;
          else if (x == 0) ring = 2 * p.W + (y - 1);
          else if (x == p.W - 1) ring = 2 * p.W + (p.H - 2) + (y - 1);
          if (ring >= 0) ring_row_d[i] = n * (2 * p.W + 2 * (p.H - 2)) + ring;
        }
      }
    }
#pragma unroll 1
    for (int pass = 0; pass < PASSES; ++pass) {
      if (wn / WPP == pass) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int col = wn * WTN + j * MT + 4 * fh;          // tile-local cout of e = 0
          const int co = tile_n * BN + col;
          float b4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) b4[e] = (p.bias != nullptr && co + e < p.Cout) ? p.bias[co + e] : 0.f;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int prow = wm * WTM + i * MT + fr;
            float s4[4] = {0.f, 0.f, 0.f, 0.f};
            if (ring_row_d[i] >= 0 && co + 3 < p.Cout) {
              const uint2 raw = *(const uint2*)((const char*)p.border_sub + ((size_t)ring_row_d[i] * p.Cout + co) * ES);
              s4[0] = h16_lo(raw.x); s4[1] = h16_hi(raw.x); s4[2] = h16_lo(raw.y); s4[3] = h16_hi(raw.y);
            }
            float4 v;
            v.x = vnqa_conv_act<TAG>(acc[i][j][0] + b4[0] - s4[0], p.relu);
            v.y = vnqa_conv_act<TAG>(acc[i][j][1] + b4[1] - s4[1], p.relu);
            v.z = vnqa_conv_act<TAG>(acc[i][j][2] + b4[2] - s4[2], p.relu);
            v.w = vnqa_conv_act<TAG>(acc[i][j][3] + b4[3] - s4[3], p.relu);
            *(float4*)(smem + prow * CROWF + (col - pass * PN) * 4) = v;
          }
        }
      }
      __syncthreads();
      for (int idx = threadIdx.x; idx < rows_out_d * CHF; idx += NT) {
        const int orow = idx / CHF, c = idx - orow * CHF;
        const int mo = tile_m * rows_out_d + orow;
        const int co0 = tile_n * BN + pass * PN + c * 8;
        if (mo >= M_out_d || co0 >= p.Cout) continue;
        float v[8];
        if (p.pool) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = -INFINITY;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const float4 a0 = *(const float4*)(smem + (orow * 4 + d) * CROWF + c * 32);
            const float4 a1 = *(const float4*)(smem + (orow * 4 + d) * CROWF + c * 32 + 16);
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
          for (int e = 0; e < 8; ++e) v[e] = v[e] * p.post_scale[co0 + e] + p.post_shift[co0 + e];
        }
        const int n = mo / (Hod * Wod);
        const int rem = mo - n * (Hod * Wod);
        const int yo = rem / Wod;
        const int xo = rem - yo * Wod;
        const size_t ooff = (((size_t)n * p.Hyp + yo + p.y_halo) * p.Wyp + xo + p.y_halo) * (size_t)p.Cy + co0;
        vnqa_bf16* dst = (vnqa_bf16*)(p.y) + ooff;
        unsigned hw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) hw[e] = pack2_h16(v[2 * e], v[2 * e + 1]);
        *(uint4*)dst = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        if constexpr (TAG == 6) {              // VNQA_CONV_F32_EPILOGUE: ONE 16-bit value, rounded once after pool / affine in fp32 ...
          if (p.dual_out == 9) *(uint4*)(dst + p.Cout) = make_uint4(hw[0], hw[1], hw[2], hw[3]);      // ... | VNQA_CONV_DUAL_HI2: written TWICE, [v | v]
          continue;
        }
        unsigned lw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) lw[e] = pack2_h16(v[2 * e] - h16_lo(hw[e]), v[2 * e + 1] - h16_hi(hw[e]));
        if (p.dual_out == 4) {      // VNQA_EPI_SPLIT_OUT: hi and lo as TWO plain tensors of y's geometry (y, y2)
          *(uint4*)((vnqa_bf16*)(p.y2) + ooff) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
          continue;
        }
        *(uint4*)(dst + p.Cout) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        if (p.dual_out == 2) *(uint4*)(dst + 2 * p.Cout) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      }
      __syncthreads();
    }
    return;
  }
  // row of the border-correction tensor for each of this lane's pixels (-1: interior pixel or no correction)
  int ring_row[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    ring_row[i] = -1;
    if (p.border_sub != nullptr) {
      const int m = tile_m * BM + wm * WTM + i * MT + fr;
      if (m < p.M) {
        int n, y, x;
        decode_pixel(m, p.H, p.W, p.pool, n, y, x);
        int ring = -1;
        if (y == 0) ring = x;
        else if (y == p.H - 1) ring = p.W + x;
        else if (x == 0) ring = 2 * p.W + (y - 1);
        else if (x == p.W - 1) ring = 2 * p.W + (p.H - 2) + (y - 1);
        if (ring >= 0) ring_row[i] = n * (2 * p.W + 2 * (p.H - 2)) + ring;
      }
    }
  }
  float bias_r[TN][NG][4];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int co = tile_n * BN + col_of(j, g);
#pragma unroll
      for (int e = 0; e < 4; ++e) bias_r[j][g][e] = (p.bias != nullptr && co + e < p.Cout) ? p.bias[co + e] : 0.f;
    }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int prow = wm * WTM + i * MT + fr;
    // composed-conv border correction (vnqa_conv2d_igemm_fwd_ex): all of this pixel's values in one batch of 8-byte loads
    float sub[TN][NG][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) sub[j][g][e] = 0.f;
    if (ring_row[i] >= 0) {
      const char* src = (const char*)p.border_sub + ((size_t)ring_row[i] * p.Cout + tile_n * BN) * ES;
      if constexpr (ES == 2) {
        uint2 raw[TN][NG];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int col = col_of(j, g);
            raw[j][g] = (tile_n * BN + col + 3 < p.Cout) ? *(const uint2*)(src + col * ES) : make_uint2(0u, 0u);
          }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            sub[j][g][0] = h16_lo(raw[j][g].x);
            sub[j][g][1] = h16_hi(raw[j][g].x);
            sub[j][g][2] = h16_lo(raw[j][g].y);
            sub[j][g][3] = h16_hi(raw[j][g].y);
          }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            const int col = col_of(j, g);
            if (tile_n * BN + col + 3 < p.Cout) {
              const float4 raw = *(const float4*)(src + col * ES);
              sub[j][g][0] = raw.x; sub[j][g][1] = raw.y; sub[j][g][2] = raw.z; sub[j][g][3] = raw.w;
            }
          }
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int col = col_of(j, g);  // tile-local cout of e=0
        float v[4];
        if constexpr (TAG == 1) {
          // stem launches: the ReLU floor is PER CHANNEL — 0, or -mean_c for a mean-shifted output (post_shift without post_scale; the
          // caller's bias already carries -mean_c): relu(a) - mean = max(a - mean, -mean), rounded ONCE into the staged tile; the
          // 2x2 max-pool of the rounded values is the rounding of the pooled one (monotone).  No fp32 staging needed (TAG 6: +0.14 ms).
          float fl[4] = {0.f, 0.f, 0.f, 0.f};
          if (p.post_scale == nullptr && p.post_shift != nullptr) {
            const int co = tile_n * BN + col;
            if (co + 3 < p.Cout) {
              const float4 f4 = *(const float4*)(p.post_shift + co);
              fl[0] = f4.x; fl[1] = f4.y; fl[2] = f4.z; fl[3] = f4.w;
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float a = acc[i][j][4 * g + e] + bias_r[j][g][e] - sub[j][g][e];
            v[e] = p.relu ? fmaxf(a, fl[e]) : a;
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = vnqa_conv_act<TAG>(acc[i][j][4 * g + e] + bias_r[j][g][e] - sub[j][g][e], p.relu);
          }
        }
        char* dst = smem + prow * CROW + col * ES;
        if constexpr (ES == 2) {
          uint2 pk;
          pk.x = pack2_h16(v[0], v[1]);
          pk.y = pack2_h16(v[2], v[3]);
          *(uint2*)dst = pk;
        } else {
          *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
        }
      }
    }
  }
  __syncthreads();

  constexpr int CH = BN * ES / 16;  // 16-byte chunks per tile row
  const bool has_post = (p.post_scale != nullptr);
  const int rows_out = p.pool ? BM / 4 : BM;
  const int M_out = p.pool ? (p.M >> 2) : p.M;
  const int Ho = p.pool ? (p.H >> 1) : p.H, Wo = p.pool ? (p.W >> 1) : p.W;
  // BNSTATS: every thread keeps sum / sum of squares of ITS 16-byte channel chunk over the tile rows it stores, per frame
  // slot (NT % CH == 0: a thread's chunk is the same on every pass of the loop)
  constexpr bool kStatsOk = (TAG == 0 || TAG == 4) && (NT % CH == 0) && (CH <= 64) && (64 % CH == 0);   // trunk instantiations (plain and wrap) only
  float st_s[3][EPC], st_q[3][EPC];
  int frame0 = 0;
  if (kStatsOk && p.epi == VNQA_EPI_BNSTATS) {
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
      for (int e = 0; e < EPC; ++e) st_s[sl][e] = st_q[sl][e] = 0.f;
    frame0 = p.frame_of[(tile_m * rows_out) / (Ho * Wo)];
  }
#ifndef VNQA_EPI_UNBATCHED        // (tools/build_variant.py unbatched -DVNQA_EPI_UNBATCHED -DVNQA_PS_EPI_UNBATCHED: the A/B partner)
  // (Only on the two large tiles: the batch's operand registers are the high-water mark of the small-tile instantiations —
  // 256 x 64 went from 116 to 133 VGPRs, i.e. from two workgroups per CU to one, and VideoOnlyCNN3D's conv2 dgrad on that
  // tile from 0.88 to 1.31 ms, although it never takes this path.)
  if constexpr (TAG == 0 && BM * BN >= 256 * 128 && (PIPE == 2 || PIPE == 5 || BN == 256)) {     // (not the 4-stage 256 x 128 ring: 119 -> 147 VGPRs)
    // FILM_RES / ADD_MASK (2-D, un-pooled, y_halo = 1, no ring): the arithmetic of the generic loop below with a thread's chunks
    // taken UN at a time and ALL their global operands (res / add / mask, gamma / beta rows) requested before the first is
    // used — chunk by chunk every iteration waited ~1.7 us for its own loads (conv_ps.hip has the same loop and the numbers).
    if ((p.epi == VNQA_EPI_FILM_RES || p.epi == VNQA_EPI_ADD_MASK) && !p.pool && p.ring_h == 0 && p.D == 0) {
      constexpr int UN = 4;
      const bool film = p.epi == VNQA_EPI_FILM_RES;
      for (int idx0 = threadIdx.x; idx0 < rows_out * CH; idx0 += UN * NT) {
        size_t ooff[UN];
        bool ok[UN];
        int xo_[UN], yo_[UN];
        uint4 ra[UN], rb[UN];
        float ga[UN][EPC], be[UN][EPC];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int idx = idx0 + u * NT;
          const int orow = idx / CH, c = idx - orow * CH;
          const int mo = tile_m * rows_out + orow;
          const int co0 = tile_n * BN + c * EPC;
          ok[u] = idx < rows_out * CH && mo < M_out && co0 < p.Cout;
          const int mm = ok[u] ? mo : 0;
          const int n = mm / (Ho * Wo);
          const int rem = mm - n * (Ho * Wo);
          yo_[u] = rem / Wo;
          xo_[u] = rem - yo_[u] * Wo;
          ooff[u] = (((size_t)n * p.Hyp + yo_[u] + p.y_halo) * p.Wyp + xo_[u] + p.y_halo) * (size_t)p.Cy + co0;
          ra[u] = rb[u] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
          for (int e = 0; e < EPC; ++e) ga[u][e] = be[u][e] = 0.f;
          if (ok[u]) {
            ra[u] = *(const uint4*)((const T*)p.res + ooff[u]);
            if (!film) rb[u] = *(const uint4*)((const T*)p.y2 + ooff[u]);
            else {
              const float* gp = p.film_gamma + (size_t)n * p.film_ld + co0;
              const float* bp = p.film_beta + (size_t)n * p.film_ld + co0;
#pragma unroll
              for (int e = 0; e < EPC; ++e) {
                const bool in = co0 + e < p.film_c;
                ga[u][e] = in ? gp[e] : 0.f;
                be[u][e] = in ? bp[e] : 0.f;
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          if (!ok[u]) continue;
          const int idx = idx0 + u * NT;
          const int orow = idx / CH, c = idx - orow * CH;
          const uint4 staged = *(const uint4*)(smem + orow * CROW + c * 16);
          const T* src = (const T*)&staged;
          T* dst = p.y != nullptr ? (T*)(p.y) + ooff[u] : nullptr;      // (FILM_RES with y == NULL: forward-only, z is not kept)
          T* second = nullptr;
          T o2[EPC];
          if (film) {
            if (dst != nullptr) *(uint4*)dst = staged;           // z: exactly the storage-rounded values staged in LDS
            const T* rt = (const T*)&ra[u];
#pragma unroll
            for (int e = 0; e < EPC; ++e)
              o2[e] = ElemOps<T>::store(fmaxf(ga[u][e] * ElemOps<T>::load(src[e]) + be[u][e], 0.f) + ElemOps<T>::load(rt[e]));
            second = (T*)p.y2 + ooff[u];
            *(uint4*)second = *(const uint4*)o2;
          } else {
            const T* at = (const T*)&ra[u];
            const T* mt = (const T*)&rb[u];
#pragma unroll
            for (int e = 0; e < EPC; ++e)
              o2[e] = ElemOps<T>::store(ElemOps<T>::load(mt[e]) > 0.f ? ElemOps<T>::load(src[e]) + ElemOps<T>::load(at[e]) : 0.f);
            *(uint4*)dst = *(const uint4*)o2;
          }
          if (p.zero_halo) {
            const int xo = xo_[u], yo = yo_[u];
            const uint4 zz = make_uint4(0u, 0u, 0u, 0u);
            const long long rs = (long long)p.Wyp * p.Cy, cs = p.Cy;
            const bool x0 = xo == 0, x1 = xo == Wo - 1, y0 = yo == 0, y1 = yo == Ho - 1;
            if (x0 | x1 | y0 | y1) {
#pragma unroll
              for (int which = 0; which < 2; ++which) {
                T* b = which == 0 ? dst : second;
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
  for (int idx = threadIdx.x; idx < rows_out * CH; idx += NT) {
    const int orow = idx / CH, c = idx - orow * CH;
    const int mo = tile_m * rows_out + orow;
    const int co0 = tile_n * BN + c * EPC;
    if (mo >= M_out || co0 >= p.Cout) continue;
    float v[EPC];
    if (p.pool) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = -INFINITY;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const T* src = (const T*)(smem + (orow * 4 + d) * CROW + c * 16);
#pragma unroll
        for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e], ElemOps<T>::load(src[e]));
      }
    } else {
      const T* src = (const T*)(smem + orow * CROW + c * 16);
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = ElemOps<T>::load(src[e]);
    }
    if (has_post) {
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = v[e] * p.post_scale[co0 + e] + p.post_shift[co0 + e];
    }
    const int n = mo / (Ho * Wo);
    const int rem = mo - n * (Ho * Wo);
    const int yo = rem / Wo;
    int xo = rem - yo * Wo;
    if (p.ring_h > 0 && p.Wyp > Wo) {
      // padded ring layout: top (W+2) | bottom (W+2) | 0 | left (H) | 0 | 0 | right (H) | 0 — the zero rows (never written) are
      // what the 1x3 edge windows read in place of the corner neighbours that belong to the top / bottom group
      xo += (xo >= 2 * (p.ring_w + 2) ? 1 : 0) + (xo >= 2 * (p.ring_w + 2) + p.ring_h ? 2 : 0);
    }
    size_t oimg = n;
    if (p.D > 0) {   // output depth slices carry a depth halo of 1 as well
      const int nn = n / p.D;
      oimg = (size_t)nn * (p.D + 2) + (n - nn * p.D) + 1;
    }
    const size_t ooff = ((oimg * p.Hyp + yo + p.y_halo) * p.Wyp + xo + p.y_halo) * (size_t)p.Cy + co0;
    const bool no_z = TAG == 0 && p.epi == VNQA_EPI_FILM_RES && p.y == nullptr;     // forward-only FiLM block: z is not kept
    T* dst = no_z ? nullptr : (T*)(p.y) + ooff;
    T out[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) out[e] = ElemOps<T>::store(v[e]);
    if (!(TAG == 0 && p.epi == VNQA_EPI_ADD_MASK) && !no_z) *(uint4*)dst = *(const uint4*)out;
    if (p.zero_halo) {
      // the halo ring of a fresh output buffer: every border pixel's thread also zeroes the halo positions next to it
      // (corners by the corner pixels) for its 16-byte channel chunk — for y and, with FILM_RES, for the second output
      const uint4 zz = make_uint4(0u, 0u, 0u, 0u);
      const long long rs = (long long)p.Wyp * p.Cy, cs = p.Cy;
      const bool x0 = xo == 0, x1 = xo == Wo - 1, y0 = yo == 0, y1 = yo == Ho - 1;
      if (x0 | x1 | y0 | y1) {
        T* const second = (TAG == 0 && p.epi == VNQA_EPI_FILM_RES) ? (T*)p.y2 + ooff : nullptr;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          T* b = which == 0 ? dst : second;
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
    if (kStatsOk && p.epi == VNQA_EPI_BNSTATS) {
      const int sl = p.frame_of[n] - frame0;        // 0..2 (checked on the host: every frame holds >= BM/3 pixels)
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const float x = v[e];
        if (sl == 0) { st_s[0][e] += x; st_q[0][e] += x * x; }
        else if (sl == 1) { st_s[1][e] += x; st_q[1][e] += x * x; }
        else { st_s[2][e] += x; st_q[2][e] += x * x; }
      }
    } else if (TAG == 0 && p.epi == VNQA_EPI_FILM_RES) {
      // out2 = relu(gamma[n] * z + beta[n]) + res, from the storage-rounded z just written (film_attn_pt_stem.py:229-241)
      float ga[EPC], be[EPC], r[EPC];
      const float* gp = p.film_gamma + (size_t)n * p.film_ld + co0;
      const float* bp = p.film_beta + (size_t)n * p.film_ld + co0;
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        const bool ok = co0 + e < p.film_c;
        ga[e] = ok ? gp[e] : 0.f;
        be[e] = ok ? bp[e] : 0.f;
      }
      const uint4 rraw = *(const uint4*)((const T*)p.res + ooff);
      const T* rt = (const T*)&rraw;
      T o2[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
        r[e] = ElemOps<T>::load(rt[e]);
        o2[e] = ElemOps<T>::store(fmaxf(ga[e] * v[e] + be[e], 0.f) + r[e]);
      }
      *(uint4*)((T*)p.y2 + ooff) = *(const uint4*)o2;
    } else if (TAG == 0 && p.epi == VNQA_EPI_ADD_MASK) {
      // y = (conv + add) * [mask > 0]: the residual join and the ReLU mask of the FiLM block's backward on the dgrad's
      // storage-rounded output (`out`: exactly what the plain conv would have stored)
      const uint4 araw = *(const uint4*)((const T*)p.res + ooff);
      const uint4 mraw = *(const uint4*)((const T*)p.y2 + ooff);
      const T* at = (const T*)&araw;
      const T* mt = (const T*)&mraw;
      T o2[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e)
        o2[e] = ElemOps<T>::store(ElemOps<T>::load(mt[e]) > 0.f ? ElemOps<T>::load(out[e]) + ElemOps<T>::load(at[e]) : 0.f);
      *(uint4*)dst = *(const uint4*)o2;
    }
  }
  if (kStatsOk && p.epi == VNQA_EPI_BNSTATS) {
    // lanes l and l ^ CH ... share a channel chunk inside a wave (64 / CH row groups): fold them with shuffles, then the
    // waves' partials through LDS (the staged tile is dead after the barrier) in a fixed order: deterministic
#pragma unroll
    for (int sl = 0; sl < 3; ++sl)
#pragma unroll
      for (int e = 0; e < EPC; ++e) {
#pragma unroll
        for (int o = 32; o >= CH; o >>= 1) {
          st_s[sl][e] += __shfl_xor(st_s[sl][e], o, 64);
          st_q[sl][e] += __shfl_xor(st_q[sl][e], o, 64);
        }
      }
    __syncthreads();
    float* red = (float*)smem;                       // [NW][3][2][BN]
    if (lane < CH) {
#pragma unroll
      for (int sl = 0; sl < 3; ++sl)
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          red[((wave * 3 + sl) * 2 + 0) * BN + lane * EPC + e] = st_s[sl][e];
          red[((wave * 3 + sl) * 2 + 1) * BN + lane * EPC + e] = st_q[sl][e];
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 3 * 2 * BN; o += NT) {
      const int ch = o % BN, sq = o / BN;              // sq = slot * 2 + {sum, sumsq}
      float acc_s = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) acc_s += red[(w * 6 + sq) * BN + ch];
      const int co = tile_n * BN + ch;
      if (co < p.Cout) p.stats_partial[((size_t)tile_m * 6 + sq) * p.Cout + co] = acc_s;
    }
  }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int TAG = 0, int PIPE = 2>
int launch(const ConvArgs& a, hipStream_t stream) {
  constexpr int NT = WAVES_M * WAVES_N * 64;
  constexpr int ES = (int)sizeof(T);
  constexpr int STAGE = (BM + BN) * ((PIPE == 3 || PIPE == 4) ? 64 : 128);
  constexpr int CT = BM * (BN * ES + 16);
  constexpr int NSTAGE = PIPE == 5 ? 2 : PIPE;
  constexpr int LDS = (NSTAGE * STAGE > CT) ? NSTAGE * STAGE : CT;
  static_assert(LDS <= 160 * 1024, "LDS budget exceeded");
  // the BNSTATS epilogue is compiled into an instantiation only where the kernel's own predicate (kStatsOk) holds: refuse the
  // request here instead of launching a kernel that would leave the statistics workspace unwritten
  constexpr bool kStatsOk = (TAG == 0 || TAG == 4) && (NT % (BN * ES / 16) == 0) && (BN * ES / 16 <= 64) && (64 % (BN * ES / 16) == 0);
  if (a.epi == VNQA_EPI_BNSTATS && !kStatsOk) {
    vnqa_set_error("conv2d_igemm_fused_fwd: the BNSTATS epilogue is not available on this tile (%dx%d, %d threads, tag %d)",
                   BM, BN, NT, TAG);
    return VNQA_ERR_UNSUPPORTED;
  }
  if ((a.epi == VNQA_EPI_FILM_RES || a.epi == VNQA_EPI_ADD_MASK) && TAG != 0) {
    vnqa_set_error("conv2d_igemm_fused_fwd: fused trunk epilogues are compiled into the TAG 0 instantiations only");
    return VNQA_ERR_UNSUPPORTED;
  }
  ConvArgs p = a;
  const int tilesM = (p.M + BM - 1) / BM;
  p.tilesN = (p.Cout + BN - 1) / BN;
  auto kern = conv_igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, TAG, PIPE>;
  static std::atomic<bool> attr_set{false};   // idempotent attribute call: a race only repeats it
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      vnqa_set_error("hipFuncSetAttribute(%d B LDS) failed: %s", LDS, hipGetErrorString(e));
      return VNQA_ERR_HIP;
    }
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tilesM * p.tilesN * p.slices), dim3(NT), LDS, stream, p);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

int resolve_tile(const ConvArgs& a, int dtype, int tile) {
  if (tile != VNQA_TILE_AUTO) return tile;
  if (dtype == VNQA_BF16) {
    if (a.Cout <= 64) return VNQA_TILE_256x64;
    if (a.Cout <= 128) return VNQA_TILE_256x128;   // (128x128 is faster alone, slower when two streams co-run)
    return VNQA_TILE_256x256;       // (callers A/B other shapes through vnqa_conv_desc.tile: the library reads no environment)
  }
  return a.Cout <= 64 ? VNQA_TILE_128x64 : VNQA_TILE_128x128;
}

// pixel rows of a tile id with TAG == 0 (the instantiations that carry the fused trunk epilogues); 0 = none
int fused_tile_rows(int dtype, int tile) {
  if (dtype == VNQA_BF16) {
    switch (tile) {
      case VNQA_TILE_256x256: case VNQA_TILE_256x128: case VNQA_TILE_256x64: case VNQA_TILE_256x128_W24:
      case VNQA_TILE_I5_256x256: return 256;
      case VNQA_TILE_128x128: case VNQA_TILE_128x64: return 128;
      default: return 0;
    }
  }
  return (tile == VNQA_TILE_128x128 || tile == VNQA_TILE_128x64) ? 128 : 0;
}

int conv_dispatch(const ConvArgs& a, int dtype, int tile, hipStream_t st) {
  tile = resolve_tile(a, dtype, tile);
  if (a.dual_out && tile != VNQA_TILE_PS_224x256 && tile != VNQA_TILE_STEM_PS_224x256) {
    // [hi | lo (| hi)] / (hi, lo) output on the implicit-GEMM tile: the TAG 5 instantiation of the 256 x 256 tile (fp32 epilogue)
    if (dtype != VNQA_BF16 || a.epi != VNQA_EPI_NONE || a.D != 0 || a.ring_h != 0 || a.partial != nullptr || a.group_tiles != 0 ||
        a.zero_halo || a.x_wrap2 || a.relu == VNQA_ACT_ELU || a.Cout % 8 != 0 || a.Cy < ((a.dual_out == 4 || a.dual_out == 8) ? 1 : (a.dual_out == 9 ? 2 : a.dual_out + 1)) * a.Cout ||
        (a.dual_out == 4 && a.y2 == nullptr) || (a.pool && (a.H % 2 != 0 || a.W % 2 != 0))) {
      vnqa_set_error("conv2d_igemm_fwd: VNQA_CONV_DUAL_OUT on the implicit-GEMM tile needs a plain 16-bit 2-D conv (no fused epilogue / "
                     "split-K / halo zeroing / wrap), c_out %% 8 == 0 and c_y >= 2 c_out (3 c_out with VNQA_CONV_DUAL_HI2)");
      return VNQA_ERR_UNSUPPORTED;
    }
    if (tile != VNQA_TILE_256x256 && tile != VNQA_TILE_STEM_256x256) {
      vnqa_set_error("conv2d_igemm_fwd: VNQA_CONV_DUAL_OUT is served by the patch-stationary tiles and the 256x256 tiles (got tile %d)", tile);
      return VNQA_ERR_UNSUPPORTED;
    }
    return a.dual_out >= 8 ? launch<vnqa_bf16, 256, 256, 2, 4, 6>(a, st) : launch<vnqa_bf16, 256, 256, 2, 4, 5>(a, st);
  }
  if (a.relu == VNQA_ACT_ELU) {        // own instantiations of the plain tiles (conv_args.h: why not a runtime branch)
    if (a.epi != VNQA_EPI_NONE || a.partial != nullptr || a.pool) {
      vnqa_set_error("conv2d_igemm_fwd: the ELU epilogue comes without pooling, split-K and fused trunk epilogues");
      return VNQA_ERR_UNSUPPORTED;
    }
    if (dtype == VNQA_BF16) {
      switch (tile) {
        case VNQA_TILE_256x256: return launch<vnqa_bf16, 256, 256, 2, 4, VNQA_TAG_ELU>(a, st);
        case VNQA_TILE_256x128: return launch<vnqa_bf16, 256, 128, 4, 2, VNQA_TAG_ELU>(a, st);
        case VNQA_TILE_256x64: return launch<vnqa_bf16, 256, 64, 8, 1, VNQA_TAG_ELU>(a, st);
        default: break;
      }
    } else {
      switch (tile) {
        case VNQA_TILE_128x128: return launch<float, 128, 128, 2, 2, VNQA_TAG_ELU>(a, st);
        case VNQA_TILE_128x64: return launch<float, 128, 64, 4, 1, VNQA_TAG_ELU>(a, st);
        default: break;
      }
    }
    vnqa_set_error("conv2d_igemm_fwd: the ELU epilogue is built for the automatic tiles only (256x256 / 256x128 / 256x64; f32 128x128 / 128x64), not tile %d", tile);
    return VNQA_ERR_UNSUPPORTED;
  }
  if (a.x_wrap2) {          // two-product form: x read twice along K (TAG 4 instantiations of the plain tiles)
    if (dtype != VNQA_BF16 || !(a.epi == VNQA_EPI_NONE || a.epi == VNQA_EPI_BNSTATS) || a.wt_tiled || a.D != 0 || a.group_tiles != 0 || a.Cin % 128 != 0) {
      vnqa_set_error("conv2d_igemm_fwd: VNQA_CONV_X_WRAP2 needs a 16-bit 2-D conv / GEMM (plain or BNSTATS epilogue) with c_in %% 128 == 0 and K-major weights");
      return VNQA_ERR_UNSUPPORTED;
    }
    switch (tile) {
      case VNQA_TILE_256x256: case VNQA_TILE_STEM_256x256: return launch<vnqa_bf16, 256, 256, 2, 4, 4>(a, st);
      case VNQA_TILE_256x128: return launch<vnqa_bf16, 256, 128, 4, 2, 4>(a, st);
      case VNQA_TILE_256x64: return launch<vnqa_bf16, 256, 64, 8, 1, 4>(a, st);
      case VNQA_TILE_512x128: return launch<vnqa_bf16, 512, 128, 4, 2, 4>(a, st);
      case VNQA_TILE_320x128: return launch<vnqa_bf16, 320, 128, 4, 2, 4>(a, st);
      default:
        vnqa_set_error("conv2d_igemm_fwd: VNQA_CONV_X_WRAP2 is available on tiles 256x256, 256x128, 256x64, 512x128, 320x128 (got %d)", tile);
        return VNQA_ERR_UNSUPPORTED;
    }
  }
  if (dtype == VNQA_BF16) {
    switch (tile) {
      case VNQA_TILE_256x256: return launch<vnqa_bf16, 256, 256, 2, 4>(a, st);
      case VNQA_TILE_256x128: return launch<vnqa_bf16, 256, 128, 4, 2>(a, st);
      case VNQA_TILE_256x64: return launch<vnqa_bf16, 256, 64, 8, 1>(a, st);
      case VNQA_TILE_128x128: return launch<vnqa_bf16, 128, 128, 2, 2>(a, st);
      case VNQA_TILE_128x64: return launch<vnqa_bf16, 128, 64, 4, 1>(a, st);
      case VNQA_TILE_STEM_256x256: return launch<vnqa_bf16, 256, 256, 2, 4, 1>(a, st);
      case VNQA_TILE_256x128_W24: return launch<vnqa_bf16, 256, 128, 2, 4>(a, st);
      case VNQA_TILE_P4_256x256: return launch<vnqa_bf16, 256, 256, 2, 4, 0, 4>(a, st);
      case VNQA_TILE_P4_256x128: return launch<vnqa_bf16, 256, 128, 4, 2, 0, 4>(a, st);
      case VNQA_TILE_P4_256x64: return launch<vnqa_bf16, 256, 64, 4, 1, 0, 4>(a, st);
      case VNQA_TILE_256x256_W16: return launch<vnqa_bf16, 256, 256, 4, 4, 2>(a, st);
      case VNQA_TILE_256x128_W16: return launch<vnqa_bf16, 256, 128, 4, 4, 2>(a, st);
      case VNQA_TILE_512x128: return launch<vnqa_bf16, 512, 128, 4, 2, 2>(a, st);
      case VNQA_TILE_P3_256x128: return launch<vnqa_bf16, 256, 128, 4, 2, 2, 3>(a, st);
      case VNQA_TILE_320x128: return launch<vnqa_bf16, 320, 128, 4, 2, 2>(a, st);
      case VNQA_TILE_I5_256x256: return launch<vnqa_bf16, 256, 256, 2, 4, 0, 5>(a, st);
      case VNQA_TILE_STEM_I5_256x256: return launch<vnqa_bf16, 256, 256, 2, 4, 1, 5>(a, st);
      case VNQA_TILE_PATCH_224x256:
      case VNQA_TILE_STEM_PATCH_224x256:
        if (a.zero_halo) {
          vnqa_set_error("conv2d_igemm_fwd: VNQA_CONV_ZERO_HALO is not available on the 224-pixel patch tiles");
          return VNQA_ERR_UNSUPPORTED;
        }
        return vnqa_conv_patch_dispatch(a, tile == VNQA_TILE_PATCH_224x256 ? 0 : 1, st);
      case VNQA_TILE_PS_224x256:
      case VNQA_TILE_STEM_PS_224x256:
        return vnqa_conv_ps_dispatch(a, tile == VNQA_TILE_PS_224x256 ? 0 : 1, st);
      default: break;
    }
  } else {
    switch (tile) {
      case VNQA_TILE_128x128: return launch<float, 128, 128, 2, 2>(a, st);
      case VNQA_TILE_128x64: return launch<float, 128, 64, 4, 1>(a, st);
      default: break;
    }
  }
  vnqa_set_error("conv2d_igemm_fwd: tile id %d not available for dtype %d", tile, dtype);
  return VNQA_ERR_UNSUPPORTED;
}


// out[m][n] = act( sum_s slab[s][m][n] + bias[n] ), fixed summation order (deterministic).
// A 256-thread block owns 16 groups of 4 consecutive outputs: thread (g = tid & 15, q = tid >> 4) sums the slices
// q, q + 16, q + 32, ... of its group with 16-byte loads (16 independent streams per output instead of one thread walking
// all slices at load latency: 62 -> 15 us for fc_embed_attn's 256 x [280 x 128] partials), then the 16 partial sums are
// added in the order q = 0..15 through LDS.
template <typename T>
__global__ void __launch_bounds__(256) splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                            T* __restrict__ out, int M, int N, int ldo, int slices, int relu) {
  __shared__ float4 part[16][16];
  const size_t total = (size_t)M * N;                 // N % 8 == 0, so groups of 4 never straddle a row
  const int g = threadIdx.x & 15, q = threadIdx.x >> 4;
  const size_t i = ((size_t)blockIdx.x * 16 + g) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < total) {
    for (int k = q; k < slices; k += 16) {
      const float4 v = *(const float4*)(slab + (size_t)k * total + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[q][g] = s;
  __syncthreads();
  if (q == 0 && i < total) {
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const float4 v = part[r][g];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int n = (int)(i % N);
    const size_t m = i / N;
    float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (bias) o[e] += bias[n + e];
      if (relu) o[e] = fmaxf(o[e], 0.f);
      out[m * ldo + n + e] = ElemOps<T>::store(o[e]);
    }
  }
}

// BNSTATS epilogue, second half: per (frame, channel) sum the tile partials in tile order (deterministic) and turn them
// into mean / biased variance.  partial [tilesM][3][2][C]; frame f covers pixels [off[f]*S, off[f+1]*S).
__global__ void bnstats_finalize_kernel(const float* __restrict__ partial, const int* __restrict__ frame_of,
                                        const int* __restrict__ frame_off, float* __restrict__ mean, float* __restrict__ var,
                                        int S, int bm, int C) {
  const int f = blockIdx.y;
  const int ch = blockIdx.x * blockDim.x + threadIdx.x;
  if (ch >= C) return;
  const long long lo = (long long)frame_off[f] * S, hi = (long long)frame_off[f + 1] * S;
  float s = 0.f, q = 0.f;
  for (int t = (int)(lo / bm); t <= (int)((hi - 1) / bm); ++t) {
    const int slot = f - frame_of[(int)(((long long)t * bm) / S)];
    if (slot < 0 || slot > 2) continue;
    s += partial[((size_t)t * 6 + slot * 2 + 0) * C + ch];
    q += partial[((size_t)t * 6 + slot * 2 + 1) * C + ch];
  }
  const float n = (float)(hi - lo);
  const float m = s / n;
  mean[(size_t)f * C + ch] = m;
  var[(size_t)f * C + ch] = fmaxf(q / n - m * m, 0.f);
}

}  // namespace

namespace {

int tile_bn(int tile) {
  switch (tile) {
    case VNQA_TILE_256x256: case VNQA_TILE_STEM_256x256: case VNQA_TILE_I5_256x256: case VNQA_TILE_STEM_I5_256x256: return 256;
    case VNQA_TILE_256x128: case VNQA_TILE_128x128: case VNQA_TILE_256x128_W24: case VNQA_TILE_512x128:
    case VNQA_TILE_P3_256x128: return 128;
    case VNQA_TILE_256x64: case VNQA_TILE_128x64: return 64;
    default: return 0;
  }
}

template <typename T>
__global__ void pack_tiled_kernel(const float* __restrict__ w, int c_out, int c_in, int taps, int c_in_pad,
                                  const float* __restrict__ out_scale, int BN, T* __restrict__ dst, size_t total) {
  constexpr int ES = (int)sizeof(T);
  constexpr int BK = 128 / ES, EPC = 16 / ES;
  const int KT = taps * (c_in_pad / BK);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % BK);
    const int row = (int)((i / BK) % BN);
    const size_t blk = i / ((size_t)BK * BN);
    const int kt = (int)(blk % KT);
    const int tile_n = (int)(blk / KT);
    const int pc = col / EPC, within = col - pc * EPC;
    const int lc = pc ^ ((row >> 1) & 7);
    const int kc = kt / taps, tap = kt - kc * taps;
    const int ch = kc * BK + lc * EPC + within;
    const int co = tile_n * BN + row;
    float v = 0.f;
    if (co < c_out && ch < c_in) {
      v = w[((size_t)co * c_in + ch) * taps + tap];
      if (out_scale) v *= out_scale[co];
    }
    dst[i] = ElemOps<T>::store(v);
  }
}

}  // namespace

extern "C" int64_t vnqa_conv_weight_tiled_bytes(int32_t c_out, int32_t c_in_pad, int32_t taps, int32_t tile, int32_t dtype) {
  const int bn = tile_bn(tile);
  if (bn == 0) return -1;
  const int es = dtype == VNQA_BF16 ? 2 : 4;
  const int64_t tiles_n = (c_out + bn - 1) / bn;
  return tiles_n * bn * (int64_t)taps * c_in_pad * es;
}

extern "C" int vnqa_pack_conv_weight_tiled(const float* w_oihw, int32_t c_out, int32_t c_in, int32_t taps,
                                           int32_t c_in_pad, const float* out_scale, int32_t tile, int32_t dtype,
                                           void* wt_tiled, void* stream) {
  VNQA_CHECK_ARG(w_oihw && wt_tiled, "pack_conv_weight_tiled: null pointer");
  const int bn = tile_bn(tile);
  VNQA_CHECK_ARG(bn > 0, "pack_conv_weight_tiled: tile id %d has no tiled weight layout", tile);
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "pack_conv_weight_tiled: bad dtype");
  const int bk = dtype == VNQA_BF16 ? 64 : 32;
  VNQA_CHECK_ARG(c_in_pad % bk == 0 && c_in_pad >= c_in, "pack_conv_weight_tiled: c_in_pad must be a multiple of %d", bk);
  const size_t total = (size_t)((c_out + bn - 1) / bn) * bn * taps * c_in_pad;
  size_t g = (total + 255) / 256;
  g = g > 4096 ? 4096 : g;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == VNQA_BF16)
    hipLaunchKernelGGL(pack_tiled_kernel<vnqa_bf16>, dim3((int)g), dim3(256), 0, st, w_oihw, c_out, c_in, taps, c_in_pad,
                       out_scale, bn, (vnqa_bf16*)wt_tiled, total);
  else
    hipLaunchKernelGGL(pack_tiled_kernel<float>, dim3((int)g), dim3(256), 0, st, w_oihw, c_out, c_in, taps, c_in_pad,
                       out_scale, bn, (float*)wt_tiled, total);
  VNQA_CHECK_LAUNCH();
  return VNQA_OK;
}

extern "C" int64_t vnqa_gemm_nt_workspace(int32_t m, int32_t n, int32_t k, int32_t dtype) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const int bk = dtype == VNQA_BF16 ? 64 : 32;
  const int bm = dtype == VNQA_BF16 ? 256 : 128;
  const int tiles = ((m + bm - 1) / bm) * ((n + 127) / 128);
  const int kt = k / bk;
  int slices = (512 + tiles - 1) / tiles;
  const int max_slices = kt / 8 > 0 ? kt / 8 : 1;
  slices = slices < 1 ? 1 : (slices > max_slices ? max_slices : slices);
  return slices <= 1 ? 0 : (int64_t)slices * m * n * 4;
}

// out[m][n] = act( sum_k a[m][k] * b[n][k] + bias[n] ): the igemm with 1x1 "images"; K is split
// over workgroups when the M x N tile grid alone cannot fill 256 CUs.
extern "C" int vnqa_gemm_nt(const void* a_mk, const void* b_nk, const float* bias, void* out, void* workspace,
                            int32_t m, int32_t n, int32_t k, int32_t ldo, int32_t relu, int32_t dtype,
                            void* stream) {
  const bool out_f32 = (dtype & VNQA_GEMM_OUT_F32) != 0;       // 16-bit operands, fp32 output (x3 products): per-call option bit
  const bool wrap2 = (dtype & VNQA_GEMM_X_WRAP2) != 0;         // a has k / 2 physical columns, read twice against b = [b_hi | b_lo]
  dtype &= ~(VNQA_GEMM_OUT_F32 | VNQA_GEMM_X_WRAP2);
  VNQA_CHECK_ARG(a_mk && b_nk && out, "gemm_nt: null pointer");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "gemm_nt: bad dtype %d", dtype);
  VNQA_CHECK_ARG(!out_f32 || (dtype == VNQA_BF16 && workspace != nullptr),
                 "gemm_nt: VNQA_GEMM_OUT_F32 needs 16-bit operands and a workspace of max(vnqa_gemm_nt_workspace, m*n*4) bytes");
  const int bk = dtype == VNQA_BF16 ? 64 : 32;
  VNQA_CHECK_ARG(m > 0 && n > 0 && k > 0 && k % bk == 0, "gemm_nt: k=%d must be a positive multiple of %d", k, bk);
  VNQA_CHECK_ARG(n % 8 == 0 && ldo >= n && ldo % 8 == 0, "gemm_nt: n=%d ldo=%d must be multiples of 8", n, ldo);
  // workspace == NULL opts out of split-K: one pass over K in a fixed order whatever m is (the frozen stem's ring GEMMs
  // use this so that a frame's features do not depend on how many other frames share the launch)
  const int64_t ws = workspace != nullptr ? vnqa_gemm_nt_workspace(m, n, k, dtype) : 0;
  ConvArgs a;
  a.x = (const char*)a_mk;
  a.wt = (const char*)b_nk;
  a.bias = bias;
  a.post_scale = nullptr;
  a.post_shift = nullptr;
  a.y = (char*)out;
  a.n_img = m; a.H = 1; a.W = 1; a.Hp = 1; a.Wp = 1;
  a.Cin = k; a.Cout = n; a.Cy = ldo;
  a.taps = 1; a.x_halo = 0; a.y_halo = 0; a.relu = relu; a.pool = 0;
  a.M = m; a.tilesN = 0; a.Hyp = 1; a.Wyp = 1; a.wt_tiled = 0; a.D = 0;
  a.slices = 1; a.kt_per_slice = 1 << 30; a.partial = nullptr; a.border_sub = nullptr; a.group_tiles = 0;
  a.epi = VNQA_EPI_NONE; a.ring_h = 0; a.ring_w = 0;
  a.x_wrap2 = wrap2 ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
  // bf16: 256-row tiles unless 128-row tiles waste fewer padded rows (e.g. m = 280: 384 instead of 512)
  int tile = VNQA_TILE_128x128;
  if (dtype == VNQA_BF16) {
    const int pad256 = (m + 255) / 256 * 256, pad128 = (m + 127) / 128 * 128;
    tile = (ws > 0 || pad128 >= pad256) ? VNQA_TILE_256x128 : VNQA_TILE_128x128;
    if (ws > 0 && m > 256 && m <= 320 && n <= 128) tile = VNQA_TILE_320x128;     // one row tile instead of two half-empty ones
    if (wrap2 && tile == VNQA_TILE_128x128) tile = VNQA_TILE_256x128;
    // (256x256 tiles for wide outputs — the stem's ring GEMM alone 112 -> 90 us — were measured and dropped: neutral end to
    // end at 224x224, -9 % at 160x208 where 412 such tiles fill 1.6 rounds of the chip)
  }
  if (ws > 0 || out_f32) {
    const int slices_req = ws > 0 ? (int)(ws / ((int64_t)m * n * 4)) : 1;
    const int kt = k / bk;
    a.kt_per_slice = (kt + slices_req - 1) / slices_req;
    a.slices = (kt + a.kt_per_slice - 1) / a.kt_per_slice;
    a.partial = (float*)workspace;
    a.bias = nullptr;
    const int rc = conv_dispatch(a, dtype, tile, st);
    if (rc != VNQA_OK) return rc;
    const size_t total = (size_t)m * n;
    const int g = (int)((total / 4 + 15) / 16);
    if (dtype == VNQA_BF16 && !out_f32)
      hipLaunchKernelGGL(splitk_reduce_kernel<vnqa_bf16>, dim3(g), dim3(256), 0, st, (const float*)workspace, bias,
                         (vnqa_bf16*)out, m, n, ldo, a.slices, relu);
    else
      hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)workspace, bias,
                         (float*)out, m, n, ldo, a.slices, relu);
    VNQA_CHECK_LAUNCH();
    return VNQA_OK;
  }
  return conv_dispatch(a, dtype, tile, st);
}

extern "C" int vnqa_gemm_nt_grouped(const void* a_gmk, const void* b_gnk, void* out, int32_t groups, int32_t m_group,
                                    int32_t n, int32_t k, int32_t ldo, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(a_gmk && b_gnk && out, "gemm_nt_grouped: null pointer");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "gemm_nt_grouped: bad dtype %d", dtype);
  const int bk = dtype == VNQA_BF16 ? 64 : 32, bm = dtype == VNQA_BF16 ? 256 : 128;
  VNQA_CHECK_ARG(groups > 0 && m_group > 0 && m_group % bm == 0,
                 "gemm_nt_grouped: m_group=%d must be a positive multiple of the %d-row tile", m_group, bm);
  VNQA_CHECK_ARG(n > 0 && k > 0 && k % bk == 0, "gemm_nt_grouped: k=%d must be a positive multiple of %d", k, bk);
  VNQA_CHECK_ARG(n % 8 == 0 && ldo >= n && ldo % 8 == 0, "gemm_nt_grouped: n=%d ldo=%d must be multiples of 8", n, ldo);
  VNQA_CHECK_ARG((int64_t)groups * m_group < (1ll << 31), "gemm_nt_grouped: too many rows");
  ConvArgs a;
  a.x = (const char*)a_gmk;
  a.wt = (const char*)b_gnk;
  a.bias = nullptr; a.post_scale = nullptr; a.post_shift = nullptr;
  a.y = (char*)out;
  a.n_img = groups * m_group; a.H = 1; a.W = 1; a.Hp = 1; a.Wp = 1;
  a.Cin = k; a.Cout = n; a.Cy = ldo;
  a.taps = 1; a.x_halo = 0; a.y_halo = 0; a.relu = 0; a.pool = 0;
  a.M = groups * m_group; a.tilesN = 0; a.Hyp = 1; a.Wyp = 1; a.wt_tiled = 0; a.D = 0;
  a.slices = 1; a.kt_per_slice = 1 << 30; a.partial = nullptr; a.border_sub = nullptr;
  a.epi = VNQA_EPI_NONE; a.ring_h = 0; a.ring_w = 0;
  a.group_tiles = m_group / bm;
  int tile = dtype == VNQA_BF16 ? (n % 256 == 0 ? VNQA_TILE_256x256 : VNQA_TILE_256x128) : VNQA_TILE_128x128;
  return conv_dispatch(a, dtype, tile, (hipStream_t)stream);
}

extern "C" int vnqa_conv2d_igemm_fwd(const vnqa_conv_desc* d, const void* x, const void* wt,
                                     const float* bias, const float* post_scale,
                                     const float* post_shift, void* y, void* stream) {
  return vnqa_conv2d_igemm_fwd_ex(d, x, wt, bias, post_scale, post_shift, nullptr, y, stream);
}

static int fill_conv_args(const vnqa_conv_desc* d, const void* x, const void* wt, const float* bias,
                          const float* post_scale, const float* post_shift, const void* border_sub, void* y,
                          ConvArgs& a) {
  VNQA_CHECK_ARG(d && x && wt && y, "conv2d_igemm_fwd: null pointer");
  VNQA_CHECK_ARG(d->dtype == VNQA_BF16 || d->dtype == VNQA_F32, "conv2d_igemm_fwd: bad dtype %d", d->dtype);
  const int bk = d->dtype == VNQA_BF16 ? 64 : 32;
  VNQA_CHECK_ARG(d->taps == 9 || d->taps == 1 || d->taps == 25 || (d->taps == 27 && d->depth > 0),
                 "conv2d_igemm_fwd: taps must be 9, 25 or 1, or 27 with depth > 0 (got %d)", d->taps);
  VNQA_CHECK_ARG(d->c_in > 0 && d->c_in % bk == 0, "conv2d_igemm_fwd: c_in=%d must be a multiple of %d", d->c_in, bk);
  VNQA_CHECK_ARG(d->c_out > 0 && d->c_out % 8 == 0 && d->c_y >= d->c_out && d->c_y % 8 == 0,
                 "conv2d_igemm_fwd: c_out=%d c_y=%d must be multiples of 8, c_y>=c_out", d->c_out, d->c_y);
  VNQA_CHECK_ARG(d->n_img > 0 && d->h > 0 && d->w > 0, "conv2d_igemm_fwd: empty problem");
  VNQA_CHECK_ARG((d->taps == 25 && d->x_halo == 2) || ((d->taps == 9 || d->taps == 27) && d->x_halo == 1) ||
                     (d->taps == 1 && (d->x_halo == 0 || d->x_halo == 1)),
                 "conv2d_igemm_fwd: x_halo=%d invalid for taps=%d", d->x_halo, d->taps);
  VNQA_CHECK_ARG(d->y_halo >= 0 && d->y_halo <= 2, "conv2d_igemm_fwd: y_halo must be 0, 1 or 2");
  VNQA_CHECK_ARG(border_sub == nullptr || (d->depth == 0 && d->h >= 2 && d->w >= 2),
                 "conv2d_igemm_fwd: border_sub needs a 2-D conv over images of at least 2x2");
  VNQA_CHECK_ARG(border_sub == nullptr || d->c_out % 4 == 0, "conv2d_igemm_fwd: border_sub needs c_out % 4 == 0");
  VNQA_CHECK_ARG(!d->pool2 || (d->h % 2 == 0 && d->w % 2 == 0), "conv2d_igemm_fwd: pool2 needs even h,w");
  if (d->flags & VNQA_CONV_RELU_FLOOR) {
    VNQA_CHECK_ARG(post_scale == nullptr && post_shift != nullptr && (d->tile == VNQA_TILE_STEM_256x256 || d->tile == VNQA_TILE_STEM_PS_224x256) &&
                       d->relu == 1 && d->c_out % 4 == 0 &&
                       !(d->flags & (VNQA_CONV_DUAL_OUT | VNQA_CONV_F32_EPILOGUE)),
                   "conv2d_igemm_fwd: VNQA_CONV_RELU_FLOOR takes post_shift (the per-channel floor) WITHOUT post_scale on the stem-tagged "
                   "256x256 tile with ReLU, c_out %% 4 == 0, plain epilogue");
  } else {
    VNQA_CHECK_ARG((post_scale == nullptr) == (post_shift == nullptr), "conv2d_igemm_fwd: post_scale/post_shift must come together");
  }
  VNQA_CHECK_ARG((long long)d->n_img * (d->depth > 0 ? d->depth : 1) * d->h * d->w < (1ll << 31), "conv2d_igemm_fwd: too many pixels");

  a.x = (const char*)x;
  a.wt = (const char*)wt;
  a.bias = bias;
  a.post_scale = post_scale;
  a.post_shift = post_shift;
  a.y = (char*)y;
  a.n_img = d->n_img;
  a.H = d->h;
  a.W = d->w;
  a.Hp = d->h + 2 * d->x_halo;
  a.Wp = d->w + 2 * d->x_halo;
  a.Cin = d->c_in;
  a.Cout = d->c_out;
  a.Cy = d->c_y;
  a.taps = d->taps;
  a.x_halo = d->x_halo;
  a.y_halo = d->y_halo;
  a.relu = d->relu;
  VNQA_CHECK_ARG(d->relu >= 0 && d->relu <= 2 && (d->relu != VNQA_ACT_ELU || !d->pool2),
                 "conv2d_igemm_fwd: relu must be 0 (none), 1 (ReLU) or 2 (ELU; not with pooling)");
  VNQA_CHECK_ARG(!(d->flags & VNQA_CONV_ZERO_HALO) || (d->y_halo == 1 && d->depth == 0),
                 "conv2d_igemm_fwd: VNQA_CONV_ZERO_HALO needs a 2-D conv with y_halo == 1");
  a.zero_halo = (d->flags & VNQA_CONV_ZERO_HALO) ? 1 : 0;
  a.xcd_split = (d->flags & VNQA_CONV_XCD_SPLIT_N) ? 1 : 0;
  a.x_wrap2 = (d->flags & VNQA_CONV_X_WRAP2) ? 1 : 0;
  a.dual_out = (d->flags & VNQA_CONV_DUAL_OUT) ? ((d->flags & VNQA_CONV_DUAL_HI2) ? 2 : 1)
                                                : ((d->flags & VNQA_CONV_F32_EPILOGUE) ? ((d->flags & VNQA_CONV_DUAL_HI2) ? 9 : 8) : 0);
  VNQA_CHECK_ARG(!a.dual_out || d->tile == VNQA_TILE_PS_224x256 || d->tile == VNQA_TILE_STEM_PS_224x256 ||
                     d->tile == VNQA_TILE_256x256 || d->tile == VNQA_TILE_STEM_256x256,
                 "conv2d_igemm_fwd: VNQA_CONV_DUAL_OUT / VNQA_CONV_F32_EPILOGUE are served by the patch-stationary tiles and the 256x256 implicit-GEMM tiles");
  a.pool = d->pool2;
  a.M = d->n_img * d->h * d->w;
  a.tilesN = 0;
  a.wt_tiled = d->wt_tiled;
  a.D = d->depth;
  if (d->depth > 0) {
    VNQA_CHECK_ARG(d->taps == 27 && d->x_halo == 1 && d->y_halo == 1, "conv2d_igemm_fwd: depth > 0 needs taps == 27 and halos == 1");
    a.M = d->n_img * d->depth * d->h * d->w;
  }
  a.slices = 1;
  a.kt_per_slice = 1 << 30;
  a.partial = nullptr;
  a.border_sub = border_sub;
  a.group_tiles = 0;
  a.epi = VNQA_EPI_NONE;
  a.ring_h = 0; a.ring_w = 0;
  a.frame_of = nullptr; a.stats_partial = nullptr; a.film_gamma = nullptr; a.film_beta = nullptr;
  a.film_ld = 0; a.film_c = 0; a.res = nullptr; a.y2 = nullptr;
  VNQA_CHECK_ARG(!d->wt_tiled || (d->tile != VNQA_TILE_AUTO && d->tile != VNQA_TILE_P4_256x256 &&
                                  d->tile != VNQA_TILE_P4_256x128 && d->tile != VNQA_TILE_P4_256x64 &&
                                  d->tile != VNQA_TILE_P3_256x128),
                 "conv2d_igemm_fwd: wt_tiled needs an explicit 128-byte-row tile id");
  const int ho = d->pool2 ? d->h / 2 : d->h, wo = d->pool2 ? d->w / 2 : d->w;
  a.Hyp = ho + 2 * d->y_halo;
  a.Wyp = wo + 2 * d->y_halo;
  return VNQA_OK;
}

extern "C" int vnqa_conv2d_igemm_fwd_ex(const vnqa_conv_desc* d, const void* x, const void* wt,
                                        const float* bias, const float* post_scale, const float* post_shift,
                                        const void* border_sub, void* y, void* stream) {
  ConvArgs a;
  const int rc = fill_conv_args(d, x, wt, bias, post_scale, post_shift, border_sub, y, a);
  if (rc != VNQA_OK) return rc;
  return conv_dispatch(a, d->dtype, d->tile, (hipStream_t)stream);
}

// conv11 evaluated at the OUTSIDE-RING positions of halo-2 images, straight from the image (implicit GEMM: the ring position
// decides the nine tap addresses) — the first operand of the composed conv's border correction (vnqa_conv2d_igemm_fwd_ex)
// without materialising the [n*R, 9*c_in] im2col matrix vnqa_ring_im2col + vnqa_gemm_nt needed (147 MB written and read
// back per 280-frame stem pass).  x: [n_img][h+4][w+4][c_in]; wt: [c_out][9][c_in]; y1: [n_img][R][c_out], R = 2(w+2) + 2h.
extern "C" int vnqa_conv2d_ring_fwd(const void* x, const void* wt, const float* bias, void* y1, int32_t n_img, int32_t h,
                                    int32_t w, int32_t c_in, int32_t c_out, int32_t padded, int32_t dtype, void* stream) {
  const bool wrap2 = (dtype & VNQA_GEMM_X_WRAP2) != 0;      // x has c_in / 2 physical channels, read twice against wt = [w_hi | w_lo]
  dtype &= ~VNQA_GEMM_X_WRAP2;
  VNQA_CHECK_ARG(x && wt && y1 && n_img > 0 && h >= 2 && w >= 2, "conv2d_ring_fwd: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "conv2d_ring_fwd: bad dtype %d", dtype);
  VNQA_CHECK_ARG(!wrap2 || dtype == VNQA_BF16, "conv2d_ring_fwd: VNQA_GEMM_X_WRAP2 needs the 16-bit format");
  const int bk = dtype == VNQA_BF16 ? 64 : 32;
  VNQA_CHECK_ARG(c_in > 0 && c_in % bk == 0 && c_out > 0 && c_out % 8 == 0, "conv2d_ring_fwd: c_in %% %d, c_out %% 8", bk);
  const int R = 2 * (w + 2) + 2 * h;
  VNQA_CHECK_ARG((long long)n_img * R < (1ll << 31), "conv2d_ring_fwd: too many ring positions");
  vnqa_conv_desc d;
  d.dtype = dtype; d.n_img = n_img; d.h = 1; d.w = R; d.c_in = c_in; d.c_out = c_out; d.c_y = c_out; d.taps = 9;
  d.x_halo = 1; d.y_halo = 0; d.relu = 0; d.pool2 = 0; d.tile = VNQA_TILE_AUTO; d.wt_tiled = 0; d.depth = 0;
  d.flags = wrap2 ? VNQA_CONV_X_WRAP2 : 0;
  ConvArgs a;
  const int rc = fill_conv_args(&d, x, wt, bias, nullptr, nullptr, nullptr, y1, a);
  if (rc != VNQA_OK) return rc;
  a.ring_h = h;
  a.ring_w = w;
  a.Hp = h + 4;        // the INPUT is the halo-2 image list; the output is [n_img][1][R] (fill_conv_args: Hyp = 1, Wyp = R)
  a.Wp = w + 4;
  if (padded) a.Wyp = R + 4;     // [n_img][R + 4]: zero separator rows around the left / right columns (see the store loop)
  // one pass over K in a fixed order whatever n_img is (a frame's features must not depend on the launch's other frames)
  int tile = VNQA_TILE_128x128;
  if (dtype == VNQA_BF16) {
    const long long m = (long long)n_img * R;
    const long long pad256 = (m + 255) / 256 * 256, pad128 = (m + 127) / 128 * 128;
    tile = (pad128 >= pad256 || wrap2) ? VNQA_TILE_256x128 : VNQA_TILE_128x128;      // (the wrap variant exists on the 256-row tiles)
  }
  return conv_dispatch(a, dtype, tile, (hipStream_t)stream);
}

// One edge product of the border correction as an implicit 1x3 conv along the padded ring rows of y1p [n][R+4][c_mid]
// (vnqa_conv2d_ring_fwd(padded = 1)): out[n][j][co] = sum_{slot, c} y1p[n][base(edge) + j + slot][c] * wt[co][slot][c],
// edge 0/1/2/3 = top / bottom / left / right, j = x (top, bottom: w outputs) or y (left, right: h outputs).  Replaces
// vnqa_ring_edge_gather + vnqa_gemm_nt (no [n*len, 3*c_mid] operand).
extern "C" int vnqa_ring_edge_conv_fwd(const void* y1p, const void* wt, void* out, int32_t n_img, int32_t h, int32_t w,
                                       int32_t c_mid, int32_t c_out, int32_t edge, int32_t dtype, void* stream) {
  const bool wrap2 = (dtype & VNQA_GEMM_X_WRAP2) != 0;      // c_mid counts the CONTRACTION channels: y1p has c_mid / 2 of them physically
  dtype &= ~VNQA_GEMM_X_WRAP2;
  VNQA_CHECK_ARG(y1p && wt && out && n_img > 0 && h >= 2 && w >= 2 && edge >= 0 && edge < 4, "ring_edge_conv_fwd: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "ring_edge_conv_fwd: bad dtype %d", dtype);
  VNQA_CHECK_ARG(!wrap2 || (dtype == VNQA_BF16 && c_mid % 128 == 0), "ring_edge_conv_fwd: VNQA_GEMM_X_WRAP2 needs the 16-bit format, c_mid %% 128 == 0");
  const int bk = dtype == VNQA_BF16 ? 64 : 32, es = dtype == VNQA_BF16 ? 2 : 4;
  VNQA_CHECK_ARG(c_mid > 0 && c_mid % bk == 0 && c_out > 0 && c_out % 8 == 0, "ring_edge_conv_fwd: c_mid %% %d, c_out %% 8", bk);
  const int R = 2 * (w + 2) + 2 * h, Rp = R + 4;
  const int len = edge < 2 ? w : h;
  const int base = edge == 0 ? 0 : (edge == 1 ? w + 2 : (edge == 2 ? 2 * (w + 2) : 2 * (w + 2) + h + 2));
  ConvArgs a;
  a.x = (const char*)y1p + (size_t)base * (wrap2 ? c_mid / 2 : c_mid) * es;
  a.x_wrap2 = wrap2 ? 1 : 0;
  a.xcd_split = 0;
  a.zero_halo = 0;
  a.wt = (const char*)wt;
  a.bias = nullptr; a.post_scale = nullptr; a.post_shift = nullptr;
  a.y = (char*)out;
  a.n_img = n_img; a.H = 1; a.W = len; a.Hp = 1; a.Wp = Rp;
  a.Cin = c_mid; a.Cout = c_out; a.Cy = c_out;
  a.taps = 3; a.x_halo = 0; a.y_halo = 0; a.relu = 0; a.pool = 0;
  a.M = n_img * len; a.tilesN = 0; a.Hyp = 1; a.Wyp = len; a.wt_tiled = 0; a.D = 0;
  a.slices = 1; a.kt_per_slice = 1 << 30; a.partial = nullptr; a.border_sub = nullptr; a.group_tiles = 0;
  a.epi = VNQA_EPI_NONE; a.ring_h = 0; a.ring_w = 0;
  a.frame_of = nullptr; a.stats_partial = nullptr; a.film_gamma = nullptr; a.film_beta = nullptr;
  a.film_ld = 0; a.film_c = 0; a.res = nullptr; a.y2 = nullptr;
  int tile = VNQA_TILE_128x128;
  if (dtype == VNQA_BF16) {
    const int pad256 = (a.M + 255) / 256 * 256, pad128 = (a.M + 127) / 128 * 128;
    tile = (pad128 >= pad256 || wrap2) ? VNQA_TILE_256x128 : VNQA_TILE_128x128;
  }
  return conv_dispatch(a, dtype, tile, (hipStream_t)stream);
}

// conv11 at ONE edge of the outside ring, with the three taps that can see the image: a ring position one pixel outside the image has
// six of its nine taps in the zero halo — the top row sees image row 0 through kernel row 2 only, the left column sees image column 0
// through kernel column 2 only — so the edge is a 1x3 (top / bottom) or 3x1 (left / right) conv over one image row / column:
// K = 3 c_in instead of vnqa_conv2d_ring_fwd's 9 c_in (two thirds of whose products are against zeros), same sums in the same
// order.  x: halo-2 images [n][h+4][w+4][c_in]; wt: [c_out][3][c_in] — kernel row 2 / row 0 / column 2 / column 0 of conv11 for
// edge 0 / 1 / 2 / 3 = top / bottom / left / right; y1p: the padded ring layout [n][R + 4][c_out] of vnqa_conv2d_ring_fwd(padded = 1),
// whose segment of this edge (w + 2 or h positions) is written.
extern "C" int vnqa_conv2d_ring_edge_fwd(const void* x, const void* wt, const float* bias, void* y1p, int32_t n_img, int32_t h, int32_t w,
                                         int32_t c_in, int32_t c_out, int32_t edge, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(x && wt && y1p && n_img > 0 && h >= 2 && w >= 2 && edge >= 0 && edge < 4, "conv2d_ring_edge_fwd: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "conv2d_ring_edge_fwd: bad dtype %d", dtype);
  const int bk = dtype == VNQA_BF16 ? 64 : 32, es = dtype == VNQA_BF16 ? 2 : 4;
  VNQA_CHECK_ARG(c_in > 0 && c_in % bk == 0 && c_out > 0 && c_out % 8 == 0, "conv2d_ring_edge_fwd: c_in %% %d, c_out %% 8", bk);
  const int R = 2 * (w + 2) + 2 * h, Rp = R + 4, Wp = w + 4;
  const int len = edge < 2 ? w + 2 : h;
  // first window of the edge in the halo-2 image (padded row, column of its first tap) and its segment in the padded ring layout
  const int row0 = edge == 0 ? 2 : (edge == 1 ? h + 1 : 1), col0 = edge < 2 ? 0 : (edge == 2 ? 2 : w + 1);
  const int base = edge == 0 ? 0 : (edge == 1 ? w + 2 : (edge == 2 ? 2 * (w + 2) + 1 : 2 * (w + 2) + h + 3));
  ConvArgs a;
  a.x = (const char*)x + ((size_t)row0 * Wp + col0) * c_in * es;
  a.x_wrap2 = 0;
  a.xcd_split = 0;
  a.zero_halo = 0;
  a.wt = (const char*)wt;
  a.bias = bias; a.post_scale = nullptr; a.post_shift = nullptr;
  a.y = (char*)y1p + (size_t)base * c_out * es;
  a.n_img = n_img;
  a.H = edge < 2 ? 1 : len; a.W = edge < 2 ? len : 1;       // positions run along the row (top / bottom) or down the column (left / right)
  a.Hp = h + 4; a.Wp = Wp;
  a.Cin = c_in; a.Cout = c_out; a.Cy = c_out;
  a.taps = 3; a.tap3_vertical = edge < 2 ? 0 : 1;
  a.x_halo = 0; a.y_halo = 0; a.relu = 0; a.pool = 0;
  a.M = n_img * len; a.tilesN = 0; a.wt_tiled = 0; a.D = 0;
  // output position j of image n -> y1p[n][base + j]: a row of Rp entries per image (positions along x), or Rp rows of one (along y)
  a.Hyp = edge < 2 ? 1 : Rp; a.Wyp = edge < 2 ? Rp : 1;
  a.slices = 1; a.kt_per_slice = 1 << 30; a.partial = nullptr; a.border_sub = nullptr; a.group_tiles = 0;
  a.epi = VNQA_EPI_NONE; a.ring_h = 0; a.ring_w = 0;
  a.frame_of = nullptr; a.stats_partial = nullptr; a.film_gamma = nullptr; a.film_beta = nullptr;
  a.film_ld = 0; a.film_c = 0; a.res = nullptr; a.y2 = nullptr;
  int tile = VNQA_TILE_128x128;
  if (dtype == VNQA_BF16) {
    const int pad256 = (a.M + 255) / 256 * 256, pad128 = (a.M + 127) / 128 * 128;
    tile = pad128 >= pad256 ? VNQA_TILE_256x128 : VNQA_TILE_128x128;
  }
  return conv_dispatch(a, dtype, tile, (hipStream_t)stream);
}

// The border correction of the composed conv11.conv12 pair, ONE EDGE, as a single 1x5 (top / bottom) or 5x1 (left / right) conv over the
// image's border row / column (round 6): conv12's taps that reach outside the image multiply conv11 evaluated on the outside ring, and
// conv11 there sees the image through ONE kernel row / column only — so edge e's correction of border pixel j is
//   out[n][j][co] = bias[co] + sum_{t < 5, c} x[n][border row / column, j + t - 2][c] * wt[co][t][c]
// with wt = the composition of conv12's edge taps and conv11's facing taps (made by the caller in float64: stem._compose_pair) —
// K = 5 c_in instead of the two-step form's 3 c_in + 3 c_mid (vnqa_conv2d_ring_edge_fwd + vnqa_ring_edge_conv_fwd), a quarter of its FLOPs
// at 128 -> 512 -> 512, one launch instead of two and no ring tensor.  The four corner pixels' double-counted term is the caller's
// (vnqa_ring_assemble_corners).  x: halo-2 images [n][h+4][w+4][c_in] (the halo supplies the window's two pixels beyond the image's
// ends); out: dense [n][w | h][c_out]; edge 0/1/2/3 = top/bottom/left/right.
extern "C" int vnqa_conv2d_border_edge_fwd(const void* x, const void* wt, const float* bias, void* out, int32_t n_img, int32_t h, int32_t w,
                                           int32_t c_in, int32_t c_out, int32_t edge, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(x && wt && out && n_img > 0 && h >= 2 && w >= 2 && edge >= 0 && edge < 4, "conv2d_border_edge_fwd: bad arguments");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "conv2d_border_edge_fwd: bad dtype %d", dtype);
  const int bk = dtype == VNQA_BF16 ? 64 : 32, es = dtype == VNQA_BF16 ? 2 : 4;
  VNQA_CHECK_ARG(c_in > 0 && c_in % bk == 0 && c_out > 0 && c_out % 8 == 0, "conv2d_border_edge_fwd: c_in %% %d, c_out %% 8", bk);
  const int Wp = w + 4;
  const int len = edge < 2 ? w : h;
  // first tap of the first window in the halo-2 image: (padded row of the border row, padded column 0) / (padded row 0, padded column of the border column)
  const int row0 = edge == 0 ? 2 : (edge == 1 ? h + 1 : 0), col0 = edge < 2 ? 0 : (edge == 2 ? 2 : w + 1);
  ConvArgs a;
  a.x = (const char*)x + ((size_t)row0 * Wp + col0) * c_in * es;
  a.x_wrap2 = 0;
  a.xcd_split = 0;
  a.zero_halo = 0;
  a.wt = (const char*)wt;
  a.bias = bias; a.post_scale = nullptr; a.post_shift = nullptr;
  a.y = (char*)out;
  a.n_img = n_img;
  a.H = edge < 2 ? 1 : len; a.W = edge < 2 ? len : 1;
  a.Hp = h + 4; a.Wp = Wp;
  a.Cin = c_in; a.Cout = c_out; a.Cy = c_out;
  a.taps = 5; a.tap3_vertical = edge < 2 ? 0 : 1;
  a.x_halo = 0; a.y_halo = 0; a.relu = 0; a.pool = 0;
  a.M = n_img * len; a.tilesN = 0; a.wt_tiled = 0; a.D = 0;
  a.Hyp = edge < 2 ? 1 : len; a.Wyp = edge < 2 ? len : 1;
  a.slices = 1; a.kt_per_slice = 1 << 30; a.partial = nullptr; a.border_sub = nullptr; a.group_tiles = 0;
  a.epi = VNQA_EPI_NONE; a.ring_h = 0; a.ring_w = 0;
  a.frame_of = nullptr; a.stats_partial = nullptr; a.film_gamma = nullptr; a.film_beta = nullptr;
  a.film_ld = 0; a.film_c = 0; a.res = nullptr; a.y2 = nullptr;
  int tile = VNQA_TILE_128x128;
  if (dtype == VNQA_BF16) {
    const int pad256 = (a.M + 255) / 256 * 256, pad128 = (a.M + 127) / 128 * 128;
    tile = pad128 >= pad256 ? VNQA_TILE_256x128 : VNQA_TILE_128x128;
  }
  return conv_dispatch(a, dtype, tile, (hipStream_t)stream);
}

// pixel rows per tile the fused-epilogue conv would use for this problem (0: no fused instantiation)
static int fused_rows_for(const vnqa_conv_desc* d) {
  ConvArgs a;
  a.Cout = d->c_out;
  return fused_tile_rows(d->dtype, resolve_tile(a, d->dtype, d->tile));
}

extern "C" int64_t vnqa_conv2d_bnstats_workspace(const vnqa_conv_desc* d, int32_t min_frame_images) {
  if (!d || d->pool2 || d->depth > 0 || d->wt_tiled) return -1;
  const int bm = fused_rows_for(d);
  if (bm == 0) return -1;
  // a tile may overlap at most 3 frames: two whole frames plus a pixel on either side must not fit into one tile
  if (2ll * min_frame_images * d->h * d->w + 2 <= bm) return -1;
  const int64_t tiles_m = ((int64_t)d->n_img * d->h * d->w + bm - 1) / bm;
  return tiles_m * 6 * d->c_out * 4;
}

extern "C" int vnqa_conv2d_igemm_fused_fwd(const vnqa_conv_desc* d, const void* x, const void* wt, const float* bias,
                                           const vnqa_conv_epilogue* e, void* y, void* stream) {
  VNQA_CHECK_ARG(e != nullptr, "conv2d_igemm_fused_fwd: null epilogue");
  ConvArgs a;
  // FILM_RES with y == NULL: forward-only (inference) form — the conv output z is needed by the BACKWARD only and is not stored
  const bool film_no_z = e->kind == VNQA_EPI_FILM_RES && y == nullptr && e->y2 != nullptr;
  const int rc = fill_conv_args(d, x, wt, bias, nullptr, nullptr, nullptr, film_no_z ? e->y2 : y, a);
  if (rc != VNQA_OK) return rc;
  if (film_no_z) a.y = nullptr;
  VNQA_CHECK_ARG(!d->pool2 && d->depth == 0 && !d->wt_tiled, "conv2d_igemm_fused_fwd: 2-D, un-pooled, K-major weights only");
  hipStream_t st = (hipStream_t)stream;
  if (e->kind == VNQA_EPI_SPLIT_OUT) {      // y = h16(v), y2 = h16(v - y): the fp32 accumulator kept as TWO plain 16-bit tensors
    VNQA_CHECK_ARG(e->y2 && y && (d->tile == VNQA_TILE_PS_224x256 || d->tile == VNQA_TILE_256x256) && !(d->flags & VNQA_CONV_DUAL_OUT),
                   "conv2d_igemm_fused_fwd(SPLIT_OUT): y, y2 and the patch-stationary tile (or the 256x256 implicit-GEMM tile) are required");
    a.dual_out = 4;
    a.y2 = (char*)e->y2;
    return conv_dispatch(a, d->dtype, d->tile, st);
  }
  // (the patch-stationary kernel carries FILM_RES and ADD_MASK in its own store loop: conv_ps.hip)
  const bool ps_tile = d->tile == VNQA_TILE_PS_224x256 && (e->kind == VNQA_EPI_FILM_RES || e->kind == VNQA_EPI_ADD_MASK);
  const int bm = ps_tile ? 224 : fused_rows_for(d);
  if (bm == 0) {
    vnqa_set_error("conv2d_igemm_fused_fwd: tile id %d has no fused-epilogue instantiation", d->tile);
    return VNQA_ERR_UNSUPPORTED;
  }
  if (e->kind == VNQA_EPI_BNSTATS) {
    VNQA_CHECK_ARG(e->frame_of && e->frame_off && e->n_frames > 0 && e->partial && e->mean && e->var,
                   "conv2d_igemm_fused_fwd(BNSTATS): null argument");
    if (vnqa_conv2d_bnstats_workspace(d, e->min_frame_images) < 0) {
      vnqa_set_error("conv2d_igemm_fused_fwd(BNSTATS): frames of %d image(s) x %dx%d are too small for a %d-pixel tile",
                     e->min_frame_images, d->h, d->w, bm);
      return VNQA_ERR_UNSUPPORTED;
    }
    a.epi = VNQA_EPI_BNSTATS;
    a.frame_of = e->frame_of;
    a.stats_partial = e->partial;
    const int rc2 = conv_dispatch(a, d->dtype, d->tile, st);
    if (rc2 != VNQA_OK) return rc2;
    dim3 grid((d->c_out + 127) / 128, e->n_frames);
    hipLaunchKernelGGL(bnstats_finalize_kernel, grid, dim3(128), 0, st, (const float*)e->partial, e->frame_of, e->frame_off,
                       e->mean, e->var, d->h * d->w, bm, d->c_out);
    VNQA_CHECK_LAUNCH();
    return VNQA_OK;
  }
  if (e->kind == VNQA_EPI_FILM_RES) {
    VNQA_CHECK_ARG(e->gamma && e->beta && e->res && e->y2 && e->film_ld > 0 && e->film_c > 0 && e->film_c <= d->c_out,
                   "conv2d_igemm_fused_fwd(FILM_RES): bad argument");
    VNQA_CHECK_ARG(!d->relu, "conv2d_igemm_fused_fwd(FILM_RES): the conv itself carries no ReLU (film_attn_pt_stem.py:224)");
    a.epi = VNQA_EPI_FILM_RES;
    a.film_gamma = e->gamma;
    a.film_beta = e->beta;
    a.film_ld = e->film_ld;
    a.film_c = e->film_c;
    a.res = (const char*)e->res;
    a.y2 = (char*)e->y2;
    return conv_dispatch(a, d->dtype, d->tile, st);
  }
  if (e->kind == VNQA_EPI_ADD_MASK) {
    VNQA_CHECK_ARG(e->res && e->y2, "conv2d_igemm_fused_fwd(ADD_MASK): res (the addend) and y2 (the mask source) are required");
    VNQA_CHECK_ARG(!d->relu && bias == nullptr, "conv2d_igemm_fused_fwd(ADD_MASK): no bias / ReLU on the conv itself");
    a.epi = VNQA_EPI_ADD_MASK;
    a.res = (const char*)e->res;
    a.y2 = (char*)e->y2;          // read-only here: the tensor whose sign is the mask
    return conv_dispatch(a, d->dtype, d->tile, st);
  }
  vnqa_set_error("conv2d_igemm_fused_fwd: unknown epilogue kind %d", e->kind);
  return VNQA_ERR_INVALID_ARG;
}
