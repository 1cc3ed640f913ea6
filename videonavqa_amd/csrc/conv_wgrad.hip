// conv_wgrad.hip — conv2d weight gradient on MFMA (gfx950).
//
//   dWt[co][tap][ci] = sum_P dY[P][co] * X[P + d_tap][ci],   d_tap = (r-1)*Wp + (s-1)
//
// X and dY are padded-NHWC buffers of the SAME geometry [n][H+2][W+2][C]; P is the flat
// padded pixel index.  Because dY's halo is zero, the sum may run over ALL padded positions:
// a tap is then nothing but a constant shift of the X row pointer, and the contraction index
// (pixels) is the contiguous-row direction of both operands.  Both tiles are therefore
// [pixels][channels] images in LDS (filled by global_load_lds, 16-byte chunks XOR-swizzled on
// the source side), and the MFMA operands — which need 8 consecutive PIXELS of one channel
// per lane — come out of LDS through the CDNA4 transposed read ds_read_b64_tr_b16.
// The exact-f32 variant uses v_mfma_f32_32x32x2_f32, whose operands are one element per lane
// (plain ds_read_b32 of a [pixel][channel] image, no transpose needed).
//
// The pixel dimension is split over workgroups (split-K); partial tiles go to fp32 slabs in a
// caller-provided workspace and a second kernel sums them in a fixed order (deterministic).
#include <cstdlib>

#include "vnqa_common.h"

namespace {

struct WgradArgs {
  const char* x;
  const char* dy;
  float* out;       // slab base: [slices][Cout][taps][Cin]
  long long Ptot;   // n*Hp*Wp  (n*Dp*Hp*Wp for 3-D)
  long long Vtot;   // contraction extent: Ptot, or n*H*W when only the VALID pixels are visited (2-D convs)
  int vrow, prow;   // H*W and Hp*Wp: compact index v -> image v / vrow, (y, x) = divmod(v % vrow, vw) -> padded pixel
                    // img * prow + (y + 1) * Wp + x + 1   (vrow = 0: identity, every padded position is visited)
  int vw;           // W
  unsigned vw_magic;  // ceil(2^32 / W): y = umulhi(rem, vw_magic) is exact for rem < 2^32 / W
  int Wp, Hp;
  int Cin, Cout, taps;
  int tilesCo, tilesCi;
  int ksteps_total, ksteps_per_slice, slices;
  int taps_real;    // SMALL form: `taps` counts tap GROUPS of 256 / Cin taps; this is the conv's tap count (27)
  int x_cs;         // pixel stride of x in elements (Cin; 2 / 3 Cin for a [hi | lo (| hi)] tensor whose first segment is contracted: VNQA_WGRAD_X_PAIR / _X_TRIPLE)
  float* final;     // != NULL: the LAST workgroup to finish a tile folds the slices' partial tiles into `final` (no reduce launch)
};

// arrival counters of the fused slab reduce, one per output tile, self-resetting (the last arriver writes 0 back).  Process-wide:
// two weight-gradient launches must not be in flight at once on different streams (they never are: the trunk is one stream).
__device__ int g_wgrad_arrivals[8192];

__device__ __forceinline__ void glds16w(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(4))) short s16x4;

// rows of the contraction range that do not exist read zeros from here
__device__ __attribute__((aligned(16))) char vnqa_zero_page[64];

__device__ __forceinline__ s16x4 lds_tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
// The same read as inline asm.  The compiler cannot tell that the transposed-read builtin does not alias the LDS-DMA of
// the NEXT stage and puts `s_waitcnt vmcnt(0)` in front of the first fragment read of every K-step (the plain ds_read_b128
// of conv_igemm.hip do not get one): the whole global->LDS transfer was exposed, 25 % of the kernel.  Reads issued from
// asm are invisible to that pass; the caller waits for them with lds_tr_wait() and ties the fragments to it.
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
template <int OFF>
__device__ __forceinline__ s16x4 lds_tr_read_asm(unsigned addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
__device__ __forceinline__ void lds_tr_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// 16-byte-chunk XOR swizzle of a [pixel row][256 channels] bf16 tile (512-byte rows) for ds_read_b64_tr_b16, which is
// served in two 32-lane groups over 64 banks x 4 B: the four rows q4 of a 16-lane group go to four different 64-byte
// units (row bits 0-1 -> chunk bits 2-3), and the two 16-lane groups of a 32-lane access, 8 rows apart in the 16x16x32
// fragment layout, to the two 32-byte halves of a unit (row bit 3 -> chunk bit 1).  Without the second term both groups
// hit the same 32 banks (2-way conflict on every fragment read).  Rows r and r + 4 (the lo / hi halves) swizzle alike.
__device__ __forceinline__ int swz_tr(int row) { return ((row & 3) << 2) | (((row >> 3) & 1) << 1); }

// BM = BN = 256 channels, 8 waves as 2 (co) x 4 (ci): wave tile 128 co x 64 ci.
// SMALL (bf16, C_out <= 128, C_in = 64 or 128 — the Conv3d layers of v_only_cnn3d): the 256 "ci" columns of a tile are
// 256 / C_in TAPS side by side (each its own row shift of X), and the two co halves of the wave grid, which would be idle,
// take the two 32-pixel substeps of a K-step instead (two partial slabs per slice): a tile is 128 co x (4 taps x 64 ci) at
// the full kernel's MFMA : LDS ratio, where the plain tiling would use 1/8 of its 256 x 256 tile.
template <typename T, bool SMALL = false>
__global__ void __launch_bounds__(512) conv_wgrad_kernel(const WgradArgs p) {
  static_assert(!SMALL || sizeof(T) == 2, "the small-channel form is 16-bit only");
  constexpr int ES = (int)sizeof(T);
  constexpr int BCH = 256;                 // channels per tile side
  constexpr int RB = BCH * ES;             // LDS row bytes
  constexpr int KP = 32768 / RB;           // pixels per K-step (64 bf16 / 32 f32)
  constexpr int CPR = RB / 16;             // 16-byte chunks per row
  constexpr int TILE_BYTES = 32768;
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  constexpr int TM = 4, TN = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // block -> (slice, co tile, tap, ci tile); ci fastest so neighbours share the dY tile.
  // XCD-aware bijective remap first (hardware deals blocks to the 8 XCDs round-robin): an XCD gets a CONTIGUOUS run of
  // this list, i.e. the tiles of one or two pixel slices, so the 18 workgroups that read the same dY rows and the same
  // (tap-shifted) X rows hit one 4 MiB L2 instead of each fetching them from the Infinity Cache.
  int bid;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
#ifndef VNQA_WGRAD_NO_XCD_REMAP
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
#else
    bid = b;
#endif
  }
  const int tile_ci = bid % p.tilesCi; bid /= p.tilesCi;
  const int tap = bid % p.taps; bid /= p.taps;
  const int tile_co = bid % p.tilesCo; bid /= p.tilesCo;
  const int slice = bid;

  int dtap = 0;
  auto tap_shift3d = [&](int t) {
    const int q = t / 9, rs = t - 9 * q, r = rs / 3, s = rs - 3 * r;
    return ((q - 1) * p.Hp + (r - 1)) * p.Wp + (s - 1);
  };
  if constexpr (SMALL) {
  } else if (p.taps == 9) {
    const int r = tap / 3, s = tap - 3 * r;
    dtap = (r - 1) * p.Wp + (s - 1);
  } else if (p.taps == 27) {
    const int q = tap / 9, rs = tap - 9 * q, r = rs / 3, s = rs - 3 * r;
    dtap = ((q - 1) * p.Hp + (r - 1)) * p.Wp + (s - 1);
  }

  // ---- per-lane staging geometry: instruction q = wave*4 + j covers 1 KiB of the tile ----
  int st_row[4], st_coff_a[4], st_coff_b[4], st_dtap[4];
  bool st_tap_ok[4], st_a_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int lin = (wave * 4 + j) * 64 + lane;   // 16-byte chunk index within the tile
    const int row = lin / CPR, phys = lin - row * CPR;
    const int logical = (ES == 2) ? (phys ^ swz_tr(row)) : phys;
    st_row[j] = row;
    // clamp channel tiles that stick out of the tensor (results for those rows/cols are dropped)
    int ca = tile_co * BCH + logical * (16 / ES);
    int cb = tile_ci * BCH + logical * (16 / ES);
    ca = ca < p.Cout ? ca : p.Cout - (16 / ES);
    cb = cb < p.Cin ? cb : p.Cin - (16 / ES);
    st_coff_a[j] = ca * ES;
    st_coff_b[j] = cb * ES;
    st_dtap[j] = dtap;
    st_tap_ok[j] = true;
    st_a_ok[j] = !SMALL || logical * (16 / ES) < p.Cout;     // SMALL: the tile's co columns past C_out are never read
    if constexpr (SMALL) {
      const int cpt = p.Cin >> 3;                         // 16-byte chunks per tap
      const int sub = logical / cpt, tapid = tap * (32 / cpt) + sub;
      st_tap_ok[j] = tapid < p.taps_real;
      st_dtap[j] = st_tap_ok[j] ? tap_shift3d(tapid) : 0;
      st_coff_b[j] = (logical - sub * cpt) * 16;
    }
  }
  const size_t rowA = (size_t)p.Cout * ES, rowB = (size_t)p.x_cs * ES;

  // Contraction position of this lane's four staging rows, advanced by KP per stage (no division in the loop):
  // v = compact index (halo rows skipped when vrow > 0), pa = padded pixel index of the dY row, rem = v % vrow.
  const int k_begin = slice * p.ksteps_per_slice;
  int k_end = k_begin + p.ksteps_per_slice;
  k_end = k_end < p.ksteps_total ? k_end : p.ksteps_total;
  long long st_v[4], st_pa[4];
  int st_rem[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long v = (long long)k_begin * KP + st_row[j];
    st_v[j] = v;
    st_pa[j] = v;
    st_rem[j] = 0;
    if (p.vrow > 0) {                         // 32-bit arithmetic: the 2-D plans are checked to stay below 2^31 pixels
      const unsigned vi = (unsigned)v;
      const unsigned img = vi / (unsigned)p.vrow;
      st_rem[j] = (int)(vi - img * (unsigned)p.vrow);
      st_pa[j] = (long long)img * p.prow + p.Wp + 1;      // padded index of the image's first valid pixel
    }
  }
  // stages are issued in K order: each call stages the current position and steps to the next
  auto stage = [&](int buf) {
    char* lds = smem + buf * STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // dY's halo is zero: the contraction visits the valid pixels only (a 16x16 padded map has 23 % halo); a tap is
      // still one constant shift of the padded pixel index
      long long pa = st_pa[j];
      if (p.vrow > 0) {
        const unsigned rem = (unsigned)st_rem[j];
        const unsigned y = __umulhi(rem, p.vw_magic);
        pa += (long long)(y * (unsigned)p.Wp + (rem - y * (unsigned)p.vw));
      }
      long long pb = pa + st_dtap[j];
      pb = pb < 0 ? 0 : (pb < p.Ptot ? pb : p.Ptot - 1);
      const char* srcA = st_v[j] < p.Vtot ? p.dy + (size_t)pa * rowA + st_coff_a[j] : (const char*)vnqa_zero_page;
#ifndef VNQA_WGRAD_DIAG_NO_A
      if (st_a_ok[j]) glds16w(srcA, lds + (wave * 4 + j) * 1024);
#endif
#ifndef VNQA_WGRAD_DIAG_NO_B
      glds16w(st_tap_ok[j] ? p.x + (size_t)pb * rowB + st_coff_b[j] : (const char*)vnqa_zero_page,
              lds + TILE_BYTES + (wave * 4 + j) * 1024);
#endif
      st_v[j] += KP;
      if (p.vrow > 0) {
        st_rem[j] += KP;
        while (st_rem[j] >= p.vrow) {          // crossed into the next image
          st_rem[j] -= p.vrow;
          st_pa[j] += p.prow;
        }
      } else {
        st_pa[j] += KP;
      }
    }
  };

#ifdef VNQA_WGRAD_MFMA32
  constexpr bool kMma16 = false;
#else
  constexpr bool kMma16 = (ES == 2);       // bf16: v_mfma_f32_16x16x32_bf16 (8 x 4 tiles of 16x16 per wave)
#endif
  vnqa_f32x16 acc[TM][TN];
  vnqa_f32x4 acc16[kMma16 ? 8 : 1][kMma16 ? 4 : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  if constexpr (kMma16) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
  }

  // (Measured and dropped: an L2 warm-up two stages ahead — one 4-byte LDS-DMA per lane and stage to each of the next-but-one
  // stage's 512 lines, left in flight by a vmcnt(1) wait.  The transfers are first touches (86 % L2 hits, 25 GB/s per CU,
  // 28 % of the kernel), but the 64 separate lines of such an instruction cost the texture addresser as much as eight
  // staging instructions: 0.347 -> 0.370 ms.  A cooperative form — wave 0 of each of the 18 workgroups that stream the same
  // rows touches only its 1/18 share, one instruction per workgroup and stage — was slower too: 0.355 -> 0.374 ms.)
  // end of a stage: the staged tile of the next stage has landed, every wave is done reading the current one
  auto stage_sync = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  };

  const int wmA = SMALL ? 0 : wm * 128;                   // first co column of this wave's fragments
  // fragment-read lane geometry
  const int g = lane >> 4, il = lane & 15, q4 = il >> 2, pp = il & 3;  // tr-read roles
  const int fh = lane >> 5, fr = lane & 31;                            // mfma roles

  if (k_begin < k_end) {
    stage(0);
    __syncthreads();
    for (int kt = k_begin; kt < k_end; ++kt) {
      const int cur = (kt - k_begin) & 1;
#ifdef VNQA_WGRAD_DIAG_NODMA   // timing-only build: the loop without its global->LDS transfers
      if (kt + 1 < k_end && kt == k_begin) stage(cur ^ 1);
#else
      if (kt + 1 < k_end) stage(cur ^ 1);
#endif
      const char* ldsA = smem + cur * STAGE_BYTES;
      const char* ldsB = ldsA + TILE_BYTES;
      if constexpr (kMma16) {
        // 16x16x32: the 16-lane group g supplies pixels 32 s + 8 g .. + 7 (two 4-row transposed reads) of 16 channels;
        // fragments of substep s+1 are read before the MFMAs of substep s are issued
        auto load16 = [&](int s, vnqa_bf16x8* af, vnqa_bf16x8* bf) {
          const int row0 = 32 * s + 8 * g + q4;
          const int sw = swz_tr(row0);
          const int sub = (pp & 1) << 3;
#ifdef VNQA_WGRAD_TR_BUILTIN
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int off = ((((wmA + i * 16) >> 3) + (pp >> 1)) ^ sw) << 4;
            const s16x4 lo = lds_tr_read(ldsA + row0 * RB + off + sub);
            const s16x4 hi = lds_tr_read(ldsA + (row0 + 4) * RB + off + sub);
            af[i] = vnqa_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int off = ((((wn * 64 + j * 16) >> 3) + (pp >> 1)) ^ sw) << 4;
            const s16x4 lo = lds_tr_read(ldsB + row0 * RB + off + sub);
            const s16x4 hi = lds_tr_read(ldsB + (row0 + 4) * RB + off + sub);
            bf[j] = vnqa_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#else
          const unsigned baseA = lds_addr(ldsA) + row0 * RB + sub, baseB = lds_addr(ldsB) + row0 * RB + sub;
          s16x4 alo[8], ahi[8], blo[4], bhi[4];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const unsigned a = baseA + (((((wmA + i * 16) >> 3) + (pp >> 1)) ^ sw) << 4);
            alo[i] = lds_tr_read_asm<0>(a);
            ahi[i] = lds_tr_read_asm<4 * RB>(a);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned a = baseB + (((((wn * 64 + j * 16) >> 3) + (pp >> 1)) ^ sw) << 4);
            blo[j] = lds_tr_read_asm<0>(a);
            bhi[j] = lds_tr_read_asm<4 * RB>(a);
          }
          lds_tr_wait();
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));      // the fragments exist only after the wait above
            af[i] = vnqa_bf16x8{alo[i][0], alo[i][1], alo[i][2], alo[i][3], ahi[i][0], ahi[i][1], ahi[i][2], ahi[i][3]};
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
            bf[j] = vnqa_bf16x8{blo[j][0], blo[j][1], blo[j][2], blo[j][3], bhi[j][0], bhi[j][1], bhi[j][2], bhi[j][3]};
          }
#endif
        };
#ifdef VNQA_WGRAD_PREFETCH   // register double-buffering of the fragments: 22 VGPR spills at 256 registers, measured slower
        vnqa_bf16x8 af[2][8], bf[2][4];
        load16(0, af[0], bf[0]);
#pragma unroll
        for (int s = 0; s < KP / 32; ++s) {
          if (s + 1 < KP / 32) load16(s + 1, af[(s + 1) & 1], bf[(s + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc16[i][j] = VNQA_MFMA_16x16x32(af[s & 1][i], bf[s & 1][j], acc16[i][j]);
        }
#elif defined(VNQA_WGRAD_TR_BUILTIN) || !defined(VNQA_WGRAD_ROLLING)
#pragma unroll
        for (int s = 0; s < KP / 32; ++s) {
          if (SMALL && s != wm) continue;                 // SMALL: this wave's co half is a pixel half instead
          vnqa_bf16x8 af[8], bf[4];
          load16(s, af, bf);
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc16[i][j] = VNQA_MFMA_16x16x32(af[i], bf[j], acc16[i][j]);
        }
#else
        // (-DVNQA_WGRAD_ROLLING, measured +-2 % for look-aheads 2..5 and left off: the loop is bound by its LDS-DMA, not by
        // fragment latency.)  Rolling fragment reads (asm-issued, counted lgkmcnt waits): the four ci fragments and two co fragments are
        // requested up front, then co fragment i+2 is requested before the 4 MFMAs of fragment i are issued — the LDS
        // latency hides behind this wave's own MFMAs with 3 co fragments live instead of 8 (no register double buffer).
        // LDS operations retire in order, so "at most N younger reads outstanding" identifies fragment i exactly; a
        // stray scalar load in between could only make the wait longer.
#pragma unroll
        for (int s = 0; s < KP / 32; ++s) {
          const int row0 = 32 * s + 8 * g + q4;
          const int sw = swz_tr(row0);
          const int sub = (pp & 1) << 3;
          const unsigned baseA = lds_addr(ldsA) + row0 * RB + sub, baseB = lds_addr(ldsB) + row0 * RB + sub;
          s16x4 alo[8], ahi[8], blo[4], bhi[4];
          auto req_a = [&](int i) {
            const unsigned a = baseA + (((((wmA + i * 16) >> 3) + (pp >> 1)) ^ sw) << 4);
            alo[i] = lds_tr_read_asm<0>(a);
            ahi[i] = lds_tr_read_asm<4 * RB>(a);
          };
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const unsigned a = baseB + (((((wn * 64 + j * 16) >> 3) + (pp >> 1)) ^ sw) << 4);
            blo[j] = lds_tr_read_asm<0>(a);
            bhi[j] = lds_tr_read_asm<4 * RB>(a);
          }
#ifndef VNQA_WGRAD_LOOKAHEAD
#define VNQA_WGRAD_LOOKAHEAD 2
#endif
          constexpr int LA = VNQA_WGRAD_LOOKAHEAD;
#pragma unroll
          for (int i = 0; i < LA; ++i) req_a(i);
          vnqa_bf16x8 bf[4];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            if (i + LA < 8) req_a(i + LA);
            const int younger = ((i + LA < 8 ? i + LA : 7) - i) * 2;      // reads issued after fragment i's
            switch (younger) {
              case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
              case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
              case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
              case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
              case 8: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
              default: asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); break;
            }
            if (i == 0) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
                bf[j] = vnqa_bf16x8{blo[j][0], blo[j][1], blo[j][2], blo[j][3], bhi[j][0], bhi[j][1], bhi[j][2], bhi[j][3]};
              }
            }
            asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
            const vnqa_bf16x8 af = vnqa_bf16x8{alo[i][0], alo[i][1], alo[i][2], alo[i][3], ahi[i][0], ahi[i][1], ahi[i][2], ahi[i][3]};
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc16[i][j] = VNQA_MFMA_16x16x32(af, bf[j], acc16[i][j]);
            __builtin_amdgcn_sched_barrier(0);     // keep request i+3 behind these MFMAs in program order
          }
        }
#endif
      } else if constexpr (ES == 2) {
#pragma unroll
        for (int s = 0; s < KP / 16; ++s) {
          // lane supplies row (16 s + 8 (g>>1) + q4 [+4]), 4 channels starting at 16(g&1) + 4 pp
          const int row0 = 16 * s + 8 * (g >> 1) + q4;
          const int sw = swz_tr(row0);
          vnqa_bf16x8 af[TM], bf[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int chunk = (wm * 128 + i * 32 + 16 * (g & 1)) / 8 + (pp >> 1);
            const int off = ((chunk ^ sw) << 4) + ((pp & 1) << 3);
            const s16x4 lo = lds_tr_read(ldsA + row0 * RB + off);
            const s16x4 hi = lds_tr_read(ldsA + (row0 + 4) * RB + off);
            af[i] = vnqa_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int chunk = (wn * 64 + j * 32 + 16 * (g & 1)) / 8 + (pp >> 1);
            const int off = ((chunk ^ sw) << 4) + ((pp & 1) << 3);
            const s16x4 lo = lds_tr_read(ldsB + row0 * RB + off);
            const s16x4 hi = lds_tr_read(ldsB + (row0 + 4) * RB + off);
            bf[j] = vnqa_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = VNQA_MFMA_32x32x16(af[i], bf[j], acc[i][j]);
        }
      } else {
#pragma unroll 4
        for (int s = 0; s < KP / 2; ++s) {
          const int row = 2 * s + fh;
          float af[TM], bf[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) af[i] = *(const float*)(ldsA + row * RB + (wm * 128 + i * 32 + fr) * 4);
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[j] = *(const float*)(ldsB + row * RB + (wn * 64 + j * 32 + fr) * 4);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
      }
      stage_sync();
    }
  }

  // ---- store the partial tile: D[co][ci], co = (reg&3)+8(reg>>2)+4 fh, ci = fr ----
  float* slab = p.out + (size_t)slice * p.Cout * p.taps * p.Cin;
  if constexpr (SMALL) {
    // D[co = 16 i + 4 (lane>>4) + e][column = 64 wn + 16 j + (lane&15)] of pixel half wm; column -> (tap, ci)
    const int r16 = lane & 15, h16 = lane >> 4;
    slab = p.out + (size_t)(slice * 2 + wm) * p.Cout * p.taps_real * p.Cin;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = wn * 64 + j * 16 + r16;
      const int sub = col / p.Cin, ci = col - sub * p.Cin;
      const int tapid = tap * (256 / p.Cin) + sub;
      if (tapid >= p.taps_real) continue;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = i * 16 + 4 * h16 + e;
          if (co < p.Cout) slab[((size_t)co * p.taps_real + tapid) * p.Cin + ci] = acc16[i][j][e];
        }
    }
    return;
  }
  if constexpr (kMma16) {
    // D[co = 16 i + 4 (lane>>4) + e][ci = 16 j + (lane&15)]
    const int r16 = lane & 15, h16 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = tile_ci * BCH + wn * 64 + j * 16 + r16;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = tile_co * BCH + wm * 128 + i * 16 + 4 * h16 + e;
          if (co < p.Cout && ci < p.Cin) slab[((size_t)co * p.taps + tap) * p.Cin + ci] = acc16[i][j][e];
        }
      }
    if (p.final != nullptr) {
      // Fused slab reduce: release this workgroup's partial tile, count arrivals; whoever arrives last acquires the others'
      // tiles and sums them IN SLICE ORDER (the result does not depend on which workgroup that is).
      __shared__ int s_last;
      __threadfence();
      __syncthreads();
      const int tile_id = (tile_co * p.taps + tap) * p.tilesCi + tile_ci;
      if (threadIdx.x == 0) {
        const int old = atomicAdd(&g_wgrad_arrivals[tile_id], 1);
        s_last = old == p.slices - 1;
        if (s_last) g_wgrad_arrivals[tile_id] = 0;
      }
      __syncthreads();
      if (s_last) {
        __threadfence();
        const size_t n_all = (size_t)p.Cout * p.taps * p.Cin;
        for (int e = threadIdx.x; e < BCH * (BCH / 4); e += 512) {
          const int co = tile_co * BCH + e / (BCH / 4), ci = tile_ci * BCH + (e % (BCH / 4)) * 4;
          if (co >= p.Cout || ci >= p.Cin) continue;
          const size_t off = ((size_t)co * p.taps + tap) * p.Cin + ci;
          float4 acc4 = *(const float4*)(p.out + off);
          for (int z = 1; z < p.slices; ++z) {
            const float4 v = *(const float4*)(p.out + (size_t)z * n_all + off);
            acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
          }
          *(float4*)(p.final + off) = acc4;
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ci = tile_ci * BCH + wn * 64 + j * 32 + fr;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = tile_co * BCH + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (co < p.Cout && ci < p.Cin) slab[((size_t)co * p.taps + tap) * p.Cin + ci] = acc[i][j][e];
      }
    }
}

#ifdef VNQA_H16_IS_F16
#define VNQA_WG_MFMA "v_mfma_f32_16x16x32_f16"
#else
#define VNQA_WG_MFMA "v_mfma_f32_16x16x32_bf16"
#endif
// accumulator pinned to the AGPR half ("+a", tied): the 4-wave form's 64 tiles fill all 256 AGPRs, and with the builtin (untied
// destination) the allocator shuffles tiles through VGPRs around every MFMA
__device__ __forceinline__ void wg_mfma(vnqa_f32x4& acc, const vnqa_bf16x8& a, const vnqa_bf16x8& b) {
  asm volatile(VNQA_WG_MFMA " %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// ---- 16-bit weight gradient, second form: 4 waves x 512 registers, ring of four 32-pixel stages ----
// Same tile (256 co x 256 ci of one tap), same staging images and the same transposed fragment reads as conv_wgrad_kernel<h16>,
// but laid out like the patch-stationary conv (conv_ps.hip): ONE wave per SIMD with a 128 co x 128 ci tile (64 accumulator tiles
// of 16 x 16 — 256 registers — beside two fragment sets), so that
//   * a half-step's 32 transposed reads are requested right before the 64 MFMAs of the half-step BEFORE it and land under them
//     (the 8-wave form waits for its reads with both waves of a SIMD stalled together: lgkmcnt(0), then 32 MFMAs);
//   * the LDS read volume per MFMA drops by a third (A + B fragments of a 128 x 128 wave tile against 128 x 64);
//   * the global -> LDS transfers of THREE later 32-pixel stages are in flight beside the one being consumed (counted vmcnt;
//     the 8-wave form issues one 64-pixel stage at the top of a K-step and drains it at the bottom).
// One barrier per half-step (64 MFMAs per wave).  Output layout, split-K slabs and the fused reduce are the 8-wave form's.
__global__ void __launch_bounds__(256) conv_wgrad4_kernel(const WgradArgs p) {
  constexpr int ES = 2, BCH = 256, RB = BCH * ES, KH = 32, CPR = RB / 16;
  constexpr int TILE_BYTES = KH * RB;            // 16 KiB: 32 pixels x 256 channels
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;    // dY tile, X tile
  constexpr int NSLOT = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int bid;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_ci = bid % p.tilesCi; bid /= p.tilesCi;
  const int tap = bid % p.taps; bid /= p.taps;
  const int tile_co = bid % p.tilesCo; bid /= p.tilesCo;
  const int slice = bid;

  int dtap = 0;
  if (p.taps == 9) {
    const int r = tap / 3, s = tap - 3 * r;
    dtap = (r - 1) * p.Wp + (s - 1);
  } else if (p.taps == 27) {
    const int q = tap / 9, rs = tap - 9 * q, r = rs / 3, s = rs - 3 * r;
    dtap = ((q - 1) * p.Hp + (r - 1)) * p.Wp + (s - 1);
  }

  // per-lane staging geometry: instruction q = wave*4 + j covers 1 KiB (two 512-byte rows) of a 16 KiB tile
  const unsigned rowA32 = (unsigned)(p.Cout * ES), rowB32 = (unsigned)(p.x_cs * ES);
  const int k_begin = slice * p.ksteps_per_slice;
  int k_end = k_begin + p.ksteps_per_slice;
  k_end = k_end < p.ksteps_total ? k_end : p.ksteps_total;
  const int nh = 2 * (k_end - k_begin);           // half-steps of 32 pixels (the plan counts 64-pixel K-steps)
  // Contraction position of this lane's four staging rows, advanced by 32 pixels per stage WITHOUT a branch (the staging code
  // is interleaved with the MFMAs of the half-step it is issued in: one scheduling region).  v = compact pixel index (valid
  // pixels only when vrow > 0), (y, x) its position in the image, pa the padded pixel index of the dY row:
  //   x += 32 % W, y += 32 / W, carry x -> y, carry y -> next image;  pa follows with constant increments.
  // The launcher sends geometries a 32-pixel step could carry twice (H < 32 / W + 1) to the 8-wave form; vrow = 0 (every padded
  // position is visited: 3-D convs, gemm_tn) is the same code with W = H = INT_MAX.
  const int W_ = p.vrow > 0 ? p.vw : 0x7fffffff, H_ = p.vrow > 0 ? p.vrow / p.vw : 0x7fffffff;
  const int q32 = p.vrow > 0 ? KH / W_ : 0, r32 = p.vrow > 0 ? KH % W_ : KH;
  const int delta = q32 * p.Wp + r32, cfix = p.vrow > 0 ? p.Wp - W_ : 0, dfix = p.vrow > 0 ? p.prow - H_ * p.Wp : 0;
  const int p_last = (int)(p.Ptot - 1), v_tot = (int)p.Vtot;
  int st_v[4], st_pa[4], st_x[4], st_y[4];
  const char* st_a[4];
  const char* st_b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int lin = (wave * 4 + j) * 64 + lane;
    const int row = lin / CPR, phys = lin - row * CPR;
    const int logical = phys ^ swz_tr(row);
    int ca = tile_co * BCH + logical * 8;
    int cb = tile_ci * BCH + logical * 8;
    ca = ca < p.Cout ? ca : p.Cout - 8;           // channel tiles that stick out of the tensor: clamped (their results are dropped)
    cb = cb < p.Cin ? cb : p.Cin - 8;
    st_a[j] = p.dy + ca * ES;
    st_b[j] = p.x + cb * ES;
    const int v = k_begin * (2 * KH) + row;
    st_v[j] = v;
    if (p.vrow > 0) {
      const unsigned img = (unsigned)v / (unsigned)p.vrow, rem = (unsigned)v - img * (unsigned)p.vrow;
      const unsigned y = rem / (unsigned)p.vw;
      st_y[j] = (int)y;
      st_x[j] = (int)(rem - y * (unsigned)p.vw);
      st_pa[j] = (int)img * p.prow + (st_y[j] + 1) * p.Wp + st_x[j] + 1;
    } else {
      st_y[j] = 0;
      st_x[j] = v;
      st_pa[j] = v;
    }
  }
  const char* zero_page = (const char*)vnqa_zero_page;
  asm volatile("" : "+s"(zero_page));       // (materialised once: re-derived inside the loop it is a scalar load + lgkmcnt(0) in front of the fragment reads)
  // 8 LDS-DMA instructions per wave and stage (the vmcnt waits below count them).  A piece (one of the wave's four row pairs) is
  // written as 7 micro-steps of 3-6 VALU instructions: the main loop places one after every other MFMA of a quadrant (the MFMAs
  // are inline asm — accumulators pinned to the AGPR half — so the interleave is by program order; `tie` keeps a micro-step
  // between the two MFMAs it was written between).
  struct Piece { bool live; int pb; const char* ba; const char* bb; unsigned oa, ob; bool c, d; int x, y; };
  auto tie_i = [](int& v) { asm volatile("" : "+v"(v)); };
  auto tie_u = [](unsigned& v) { asm volatile("" : "+v"(v)); };
  auto tie_p = [](const char*& v) { asm volatile("" : "+v"(v)); };
  auto micro = [&](Piece& t, int slot, int j, int k) {
    char* lds = smem + slot * STAGE_BYTES;
    if (k == 0) {
      tie_i(st_v[j]); tie_i(st_pa[j]);
      t.live = st_v[j] < v_tot;
      int pb = st_pa[j] + dtap;
      pb = pb < 0 ? 0 : (pb < p_last ? pb : p_last);
      t.oa = t.live ? (unsigned)st_pa[j] : 0u;      // (rows past the contraction range: base = the zero page, offset 0 — selects, no branch)
      t.ob = t.live ? (unsigned)pb : 0u;
      tie_u(t.oa); tie_u(t.ob);
    } else if (k == 1) {
      t.ba = t.live ? st_a[j] : zero_page;
      t.bb = t.live ? st_b[j] : zero_page;
      tie_p(t.ba); tie_p(t.bb);
    } else if (k == 2) {
      const char* src = t.ba + (unsigned long long)t.oa * rowA32;
      tie_p(src);
#if !defined(VNQA_WG4_DIAG) || VNQA_WG4_DIAG != 1       // timing-only builds: 1 = no global -> LDS transfers, 2 = no MFMAs
      glds16w(src, lds + (wave * 4 + j) * 1024);
#endif
    } else if (k == 3) {
      const char* src = t.bb + (unsigned long long)t.ob * rowB32;
      tie_p(src);
#if !defined(VNQA_WG4_DIAG) || VNQA_WG4_DIAG != 1
      glds16w(src, lds + TILE_BYTES + (wave * 4 + j) * 1024);
#endif
    } else if (k == 4) {
      tie_i(st_x[j]); tie_i(st_y[j]);
      st_v[j] += KH;
      t.x = st_x[j] + r32;
      t.y = st_y[j] + q32;
      t.c = t.x >= W_;
      t.x = t.c ? t.x - W_ : t.x;
      t.y += t.c ? 1 : 0;
      tie_i(t.x); tie_i(t.y); tie_i(st_v[j]);
    } else if (k == 5) {
      t.d = t.y >= H_;
      st_y[j] = t.d ? t.y - H_ : t.y;
      st_x[j] = t.x;
      tie_i(st_x[j]); tie_i(st_y[j]);
    } else if (k == 6) {
      st_pa[j] += delta + (t.c ? cfix : 0) + (t.d ? dfix : 0);
      tie_i(st_pa[j]);
    }
  };
  auto stage = [&](int slot) {       // the prologue's form: whole pieces back to back
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Piece t;
#pragma unroll
      for (int k = 0; k < 7; ++k) micro(t, slot, j, k);
    }
  };

  vnqa_f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

  // fragment-read lane geometry (conv_wgrad_kernel, 16x16x32 branch): 16-lane group g supplies pixels 8 g .. 8 g + 7
  const int g = lane >> 4, il = lane & 15, q4 = il >> 2, pp = il & 3;
  const int row0 = 8 * g + q4;
  const int sw = swz_tr(row0);
  const unsigned lane_base = lds_addr(smem) + row0 * RB + ((pp & 1) << 3);
  unsigned offA[8], offB[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    offA[i] = lane_base + (((((wm * 128 + i * 16) >> 3) + (pp >> 1)) ^ sw) << 4);
    offB[i] = lane_base + TILE_BYTES + (((((wn * 128 + i * 16) >> 3) + (pp >> 1)) ^ sw) << 4);
  }
  // The wave's 8 x 8 accumulator tiles are walked as four QUADRANTS of 4 x 4 (16 MFMAs, 256 matrix-pipe cycles) over four
  // fragment sets of 4 (A rows 0-3 / 4-7, B columns 0-3 / 4-7: 64 registers, no double buffer).  A quadrant needs one A and one
  // B set; while it runs, the set that the quadrant after the next needs first is requested into whichever set fell free:
  //   even half-step   Q00 (A0 B0) -> Q01 (A0 B1) -> Q11 (A1 B1) -> Q10 (A1 B0)     requests: B1, A1 | next stage's A0, B1
  //   odd half-step    Q01 (A0 B1) -> Q00 (A0 B0) -> Q10 (A1 B0) -> Q11 (A1 B1)     requests: B0, A1 | next stage's A0, B0
  // so every fragment read is issued a whole quadrant before its first use, and the pattern closes after two half-steps.
  s16x4 fa[2][4][2], fb[2][4][2];          // [set][fragment][rows q4 / q4 + 4]
  auto req_a = [&](int set, unsigned so) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      fa[set][i][0] = lds_tr_read_asm<0>(offA[4 * set + i] + so);
      fa[set][i][1] = lds_tr_read_asm<4 * RB>(offA[4 * set + i] + so);
    }
  };
  auto req_b = [&](int set, unsigned so) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      fb[set][j][0] = lds_tr_read_asm<0>(offB[4 * set + j] + so);
      fb[set][j][1] = lds_tr_read_asm<4 * RB>(offB[4 * set + j] + so);
    }
  };
  auto tie = [&](s16x4 (*f)[2]) {          // the fragments exist only after the lgkmcnt wait in front of this
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(f[i][0]), "+v"(f[i][1]));
  };
  // one quadrant: wait for every read issued so far, request the next set (REQ: 0 none, 1 A-set `rs`, 2 B-set `rs`, from LDS byte
  // offset `so`), then the 16 MFMAs with staging piece `piece` of slot `st_slot` in their shadow: one micro-step after every
  // other MFMA (one wave per SIMD: what is not issued between two MFMAs is issued instead of one)
  auto quadrant = [&](int as, int bs, int req, int rs, unsigned so, int st_slot, int piece) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tie(fa[as]);
    tie(fb[bs]);
    if (req == 1) req_a(rs, so);
    if (req == 2) req_b(rs, so);
    vnqa_bf16x8 bf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bf[j] = vnqa_bf16x8{fb[bs][j][0][0], fb[bs][j][0][1], fb[bs][j][0][2], fb[bs][j][0][3],
                          fb[bs][j][1][0], fb[bs][j][1][1], fb[bs][j][1][2], fb[bs][j][1][3]};
    Piece t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const vnqa_bf16x8 af = vnqa_bf16x8{fa[as][i][0][0], fa[as][i][0][1], fa[as][i][0][2], fa[as][i][0][3],
                                         fa[as][i][1][0], fa[as][i][1][1], fa[as][i][1][2], fa[as][i][1][3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#if !defined(VNQA_WG4_DIAG) || VNQA_WG4_DIAG != 2
        wg_mfma(acc[4 * as + i][4 * bs + j], af, bf[j]);
#else
        asm volatile("" :: "v"(af), "v"(bf[j]));
#endif
        const int n = 4 * i + j;
        if ((n & 1) == 1 && (n >> 1) < 7) micro(t, st_slot, piece, n >> 1);
      }
    }
  };
  // top of half-step h: stage h+1 has landed for every wave (the last two quadrants read it), every wave is done with slot (h-1) & 3
  // (stage h+3 goes there).  Stages past the slice's last half-step are issued all the same — rows past the contraction range stage
  // zeros, nothing consumes them — so the loop has no branch and every wait counts the same 8 transfers per stage.
  auto top = [&]() {
#if defined(VNQA_WG4_DIAG) && VNQA_WG4_DIAG == 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
    __builtin_amdgcn_s_barrier();
  };

  stage(0);
  stage(1);
  stage(2);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  req_a(0, 0u);
  req_b(0, 0u);
  __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
  for (int h = 0; h < nh; h += 2) {          // (nh is even)
    const unsigned s0 = (unsigned)(h & (NSLOT - 1)) * STAGE_BYTES, s1 = (unsigned)((h + 1) & (NSLOT - 1)) * STAGE_BYTES,
                   s2 = (unsigned)((h + 2) & (NSLOT - 1)) * STAGE_BYTES;
    const int t0 = (h + 3) & (NSLOT - 1), t1 = (h + 4) & (NSLOT - 1);
    top();
    quadrant(0, 0, 2, 1, s0, t0, 0);
    quadrant(0, 1, 1, 1, s0, t0, 1);
    quadrant(1, 1, 1, 0, s1, t0, 2);
    quadrant(1, 0, 2, 1, s1, t0, 3);
    top();
    quadrant(0, 1, 2, 0, s1, t1, 0);
    quadrant(0, 0, 1, 1, s1, t1, 1);
    quadrant(1, 0, 1, 0, s2, t1, 2);
    quadrant(1, 1, 2, 0, s2, t1, 3);
    // the accumulators are read right behind the loop (the compiler's AGPR -> VGPR copies sit on the exit edge) and the MFMAs
    // are inline asm, invisible to the hazard recogniser: the last trip ends with the wait states an 8-pass MFMA needs
    if (h + 2 >= nh) asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
  }
  __builtin_amdgcn_s_setprio(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the stages and the requests issued past the end
  // wait states between the last MFMAs (inline asm: invisible to the hazard recogniser) and the first read of an accumulator
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
  asm volatile("s_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));

  // ---- store the partial tile: D[co = 16 i + 4 (lane>>4) + e][ci = 16 j + (lane&15)] ----
  float* slab = p.out + (size_t)slice * p.Cout * p.taps * p.Cin;
  const int r16 = lane & 15, h16 = lane >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = tile_ci * BCH + wn * 128 + j * 16 + r16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = tile_co * BCH + wm * 128 + i * 16 + 4 * h16 + e;
        if (co < p.Cout && ci < p.Cin) slab[((size_t)co * p.taps + tap) * p.Cin + ci] = acc[i][j][e];
      }
    }
  if (p.final != nullptr) {      // fused slab reduce, as in conv_wgrad_kernel
    __shared__ int s_last4;
    __threadfence();
    __syncthreads();
    const int tile_id = (tile_co * p.taps + tap) * p.tilesCi + tile_ci;
    if (threadIdx.x == 0) {
      const int old = atomicAdd(&g_wgrad_arrivals[tile_id], 1);
      s_last4 = old == p.slices - 1;
      if (s_last4) g_wgrad_arrivals[tile_id] = 0;
    }
    __syncthreads();
    if (s_last4) {
      __threadfence();
      const size_t n_all = (size_t)p.Cout * p.taps * p.Cin;
      for (int e = threadIdx.x; e < BCH * (BCH / 4); e += 256) {
        const int co = tile_co * BCH + e / (BCH / 4), ci = tile_ci * BCH + (e % (BCH / 4)) * 4;
        if (co >= p.Cout || ci >= p.Cin) continue;
        const size_t off = ((size_t)co * p.taps + tap) * p.Cin + ci;
        float4 acc4 = *(const float4*)(p.out + off);
        for (int z = 1; z < p.slices; ++z) {
          const float4 v = *(const float4*)(p.out + (size_t)z * n_all + off);
          acc4.x += v.x; acc4.y += v.y; acc4.z += v.z; acc4.w += v.w;
        }
        *(float4*)(p.final + off) = acc4;
      }
    }
  }
}

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t n, int slices) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < slices; ++k) s += slabs[(size_t)k * n + i];
    out[i] = s;
  }
}
// the same sums (same order per element) four elements per thread and eight slabs' loads in flight: the small-channel 3-D
// weight gradients come as 72 slabs of < 1 MB, which the scalar form walks at 0.5 TB/s
__global__ void slab_reduce4_kernel(const float4* __restrict__ slabs, float4* __restrict__ out, size_t n4, int slices) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 s = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= slices; k += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(k + u) * n4 + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < slices; ++k) {
      const float4 v = slabs[(size_t)k * n4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    out[i] = s;
  }
}

// column sums of a [P][C] matrix (dbias): 16-byte loads, thread = (row lane 0..31, 8-channel chunk 0..7),
// partial[blk][c] per row block, then a final reduce with one thread per channel over <= 128 partials.
template <typename T>
__device__ __forceinline__ void cs_load8(const T* p, float v[8]);
template <>
__device__ __forceinline__ void cs_load8<vnqa_bf16>(const vnqa_bf16* p, float v[8]) {
  const uint4 u = *(const uint4*)p;
  v[0] = h16_lo(u.x); v[1] = h16_hi(u.x);
  v[2] = h16_lo(u.y); v[3] = h16_hi(u.y);
  v[4] = h16_lo(u.z); v[5] = h16_hi(u.z);
  v[6] = h16_lo(u.w); v[7] = h16_hi(u.w);
}
template <>
__device__ __forceinline__ void cs_load8<float>(const float* p, float v[8]) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <typename T>
__global__ void __launch_bounds__(256) colsum_partial_kernel(const T* __restrict__ m, float* __restrict__ partial,
                                                             long long P, int C, long long rows_per_block) {
  __shared__ float s_red[32 * 64];
  const int cg = blockIdx.x;                       // 64-channel group
  const int prow = threadIdx.x >> 3, chunk = threadIdx.x & 7;
  const long long r0 = (long long)blockIdx.y * rows_per_block;
  long long r1 = r0 + rows_per_block;
  r1 = r1 < P ? r1 : P;
  const int c0 = cg * 64 + chunk * 8;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (c0 < C) {
    for (long long r = r0 + prow; r < r1; r += 32) {
      float v[8];
      cs_load8<T>(m + (size_t)r * C + c0, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) s_red[prow * 64 + chunk * 8 + e] = s[e];
  __syncthreads();
  if (threadIdx.x < 64 && cg * 64 + threadIdx.x < C) {
    float t = 0.f;
    for (int r = 0; r < 32; ++r) t += s_red[r * 64 + threadIdx.x];
    partial[(size_t)blockIdx.y * C + cg * 64 + threadIdx.x] = t;
  }
}

struct Plan {
  int tilesCo, tilesCi, ksteps_total, slices, ksteps_per_slice, colsum_blocks;
  long long Ptot, Vtot;
  int vrow, prow, vw;
};

Plan make_plan_k(long long Ptot, int c_in, int c_out, int taps, int dtype, long long Vtot = -1, int vrow = 0, int prow = 0,
                 int vw = 0);

Plan make_plan(int n_img, int h, int w, int c_in, int c_out, int taps, int dtype) {
#ifdef VNQA_WGRAD_ALL_ROWS     // A/B: contract over every padded pixel
  return make_plan_k((long long)n_img * (h + 2) * (w + 2), c_in, c_out, taps, dtype);
#else
  // (the row decode's reciprocal multiply is exact for h * w * w < 2^32 and needs w >= 2; otherwise every padded pixel is visited)
  if (w < 2 || (long long)h * w * w >= (1ll << 32))
    return make_plan_k((long long)n_img * (h + 2) * (w + 2), c_in, c_out, taps, dtype);
  return make_plan_k((long long)n_img * (h + 2) * (w + 2), c_in, c_out, taps, dtype, (long long)n_img * h * w, h * w,
                     (h + 2) * (w + 2), w);
#endif
}

Plan make_plan_k(long long Ptot, int c_in, int c_out, int taps, int dtype, long long Vtot, int vrow, int prow, int vw) {
  Plan pl;
  pl.vw = vw;
  pl.Ptot = Ptot;
  pl.Vtot = Vtot < 0 ? Ptot : Vtot;
  pl.vrow = vrow;
  pl.prow = prow;
  const int KP = dtype == VNQA_BF16 ? 64 : 32;
  pl.tilesCo = (c_out + 255) / 256;
  pl.tilesCi = (c_in + 255) / 256;
  pl.ksteps_total = (int)((pl.Vtot + KP - 1) / KP);
  const int tiles = pl.tilesCo * pl.tilesCi * taps;
  // One workgroup per CU (128 KiB LDS): the launch runs in rounds of 256 workgroups, so pick the slice count by ROUND
  // EFFICIENCY e(s) = tiles*s / (256 * ceil(tiles*s / 256)): the smallest s <= 16 with e >= 0.9, else the best one.
  // 36 tiles (512x512x3x3): 7 slices = 252 workgroups (0.433 ms; 14 slices / two rounds 0.472 ms, 15 slices 0.478 ms);
  // 144 tiles (1024x1024x3x3): 5 slices = 720 workgroups in three 94 %-full rounds (1 slice would leave 112 CUs idle).
  // Every extra slice costs one more fp32 slab of the whole weight gradient, hence "smallest".
#if defined(VNQA_WGRAD_CEIL_SLICES)     // A/B: the first rule
  int slices = (512 + tiles - 1) / tiles;
#else
  int slices = 1;
  {
    double best = 0.0;
    for (int sl = 1; sl <= 16; ++sl) {
      const int wg = tiles * sl;
      const double e = (double)wg / (256.0 * ((wg + 255) / 256));
      if (e >= 0.9) { slices = sl; break; }
      if (e > best + 1e-9) { best = e; slices = sl; }
    }
  }
#endif
  const int max_slices = pl.ksteps_total / 8 > 0 ? pl.ksteps_total / 8 : 1;
  slices = slices < 1 ? 1 : (slices > max_slices ? max_slices : slices);
  pl.ksteps_per_slice = (pl.ksteps_total + slices - 1) / slices;
  pl.slices = (pl.ksteps_total + pl.ksteps_per_slice - 1) / pl.ksteps_per_slice;
  long long cb = pl.Ptot / 256;
  pl.colsum_blocks = (int)(cb < 1 ? 1 : (cb > 128 ? 128 : cb));
  return pl;
}

}  // namespace

extern "C" int64_t vnqa_conv2d_wgrad_workspace(int32_t n_img, int32_t h, int32_t w, int32_t c_in,
                                               int32_t c_out, int32_t taps) {
  if (n_img <= 0 || h <= 0 || w <= 0 || c_in <= 0 || c_out <= 0 || taps <= 0) return 0;     // (empty problem: the call itself is rejected)
  // sized for the larger (f32: smaller K-step -> never fewer slices than bf16) of both dtypes
  int64_t need = 0;
  for (int dt = 0; dt < 2; ++dt) {
    const Plan pl = make_plan(n_img, h, w, c_in, c_out, taps, dt);
    const int64_t b = ((int64_t)pl.slices * c_out * taps * c_in + (int64_t)pl.colsum_blocks * c_out) * 4;
    need = b > need ? b : need;
  }
  return need;
}

static int wgrad_run(const void* x, const void* dy, float* dwt, float* dbias, void* workspace, const Plan& pl,
                     int32_t w, int32_t c_in, int32_t c_out, int32_t taps, int32_t dtype, void* stream, int32_t h = 0,
                     bool small = false, bool fuse_reduce = false, int x_cs = 0, bool eight_waves = false);

// the small-channel form's plan: tap groups x slices, two partial slabs per slice
static bool small3d_ok(int c_in, int c_out, int dtype) {
#ifdef VNQA_WGRAD_NO_SMALL
  return false;
#else
  return dtype == VNQA_BF16 && c_out <= 128 && (c_in == 64 || c_in == 128);
#endif
}
static Plan small3d_plan(long long Ptot, int c_in) {
  Plan pl = make_plan_k(Ptot, 256, 256, 1, VNQA_BF16);
  const int groups = (27 + 256 / c_in - 1) / (256 / c_in);
  const int target_wgs = 256;
  int slices = target_wgs / groups;                               // one workgroup per CU (128 KiB LDS): ONE full round of 256 (two rounds: the
                                                                  // slab reduce reads twice the partials for nothing; config 2 -1.2 %)
  const int max_slices = pl.ksteps_total / 8 > 0 ? pl.ksteps_total / 8 : 1;
  slices = slices > max_slices ? max_slices : slices;
  pl.ksteps_per_slice = (pl.ksteps_total + slices - 1) / slices;
  pl.slices = (pl.ksteps_total + pl.ksteps_per_slice - 1) / pl.ksteps_per_slice;
  long long cb = Ptot / 2048;
  pl.colsum_blocks = (int)(cb < 1 ? 1 : (cb > 1024 ? 1024 : cb));
  return pl;
}

extern "C" int64_t vnqa_conv3d_wgrad_workspace(int32_t n_img, int32_t d, int32_t h, int32_t w, int32_t c_in, int32_t c_out) {
  if (n_img <= 0 || d <= 0 || h <= 0 || w <= 0 || c_in <= 0 || c_out <= 0) return 0;
  int64_t need = 0;
  if (small3d_ok(c_in, c_out, VNQA_BF16)) {
    const Plan pl = small3d_plan((long long)n_img * (d + 2) * (h + 2) * (w + 2), c_in);
    need = ((int64_t)2 * pl.slices * c_out * 27 * c_in + (int64_t)pl.colsum_blocks * c_out) * 4;
  }
  for (int dt = 0; dt < 2; ++dt) {
    const Plan pl = make_plan_k((long long)n_img * (d + 2) * (h + 2) * (w + 2), c_in, c_out, 27, dt);
    const int64_t b = ((int64_t)pl.slices * c_out * 27 * c_in + (int64_t)pl.colsum_blocks * c_out) * 4;
    need = b > need ? b : need;
  }
  return need;
}

extern "C" int vnqa_conv3d_wgrad(const void* x, const void* dy, float* dwt, float* dbias, void* workspace, int32_t n_img,
                                 int32_t d, int32_t h, int32_t w, int32_t c_in, int32_t c_out, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(x && dy && dwt && workspace, "conv3d_wgrad: null pointer");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "conv3d_wgrad: bad dtype %d", dtype);
  VNQA_CHECK_ARG(c_in % 8 == 0 && c_out % 8 == 0 && c_in >= 8 && c_out >= 8, "conv3d_wgrad: channels must be multiples of 8");
  VNQA_CHECK_ARG(n_img > 0 && d > 0 && h > 0 && w > 0, "conv3d_wgrad: empty problem");
  if (small3d_ok(c_in, c_out, dtype)) {
    const Plan pl = small3d_plan((long long)n_img * (d + 2) * (h + 2) * (w + 2), c_in);
    return wgrad_run(x, dy, dwt, dbias, workspace, pl, w, c_in, c_out, 27, dtype, stream, h, true);
  }
  const Plan pl = make_plan_k((long long)n_img * (d + 2) * (h + 2) * (w + 2), c_in, c_out, 27, dtype);
  return wgrad_run(x, dy, dwt, dbias, workspace, pl, w, c_in, c_out, 27, dtype, stream, h);
}

extern "C" int vnqa_conv2d_wgrad(const void* x, const void* dy, float* dwt, float* dbias, void* workspace,
                                 int32_t n_img, int32_t h, int32_t w, int32_t c_in, int32_t c_out,
                                 int32_t taps, int32_t dtype, void* stream) {
  const bool fuse_reduce = (dtype & VNQA_WGRAD_FUSED_REDUCE) != 0;      // per-call option bit (the library reads no environment)
  const int x_segs = (dtype & VNQA_WGRAD_X_TRIPLE) ? 3 : ((dtype & VNQA_WGRAD_X_PAIR) ? 2 : 1);   // x: [hi | lo (| hi)], x_segs c_in physical channels
  const bool eight_waves = (dtype & VNQA_WGRAD_EIGHT_WAVES) != 0;       // the first 16-bit form (conv_wgrad_kernel<h16>): kept as the cross-check
  dtype &= ~(VNQA_WGRAD_FUSED_REDUCE | VNQA_WGRAD_X_PAIR | VNQA_WGRAD_X_TRIPLE | VNQA_WGRAD_EIGHT_WAVES);
  VNQA_CHECK_ARG(x_segs == 1 || dtype == VNQA_BF16, "conv2d_wgrad: VNQA_WGRAD_X_PAIR / _X_TRIPLE need the 16-bit format");
  VNQA_CHECK_ARG(x && dy && dwt && workspace, "conv2d_wgrad: null pointer");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "conv2d_wgrad: bad dtype %d", dtype);
  VNQA_CHECK_ARG(taps == 9 || taps == 1, "conv2d_wgrad: taps must be 9 or 1");
  VNQA_CHECK_ARG(c_in % 8 == 0 && c_out % 8 == 0 && c_in >= 8 && c_out >= 8, "conv2d_wgrad: channels must be multiples of 8");
  VNQA_CHECK_ARG(n_img > 0 && h > 0 && w > 0, "conv2d_wgrad: empty problem");
  VNQA_CHECK_ARG((long long)n_img * (h + 2) * (w + 2) < (1ll << 31), "conv2d_wgrad: too many pixels");
  const Plan pl = make_plan(n_img, h, w, c_in, c_out, taps, dtype);
  return wgrad_run(x, dy, dwt, dbias, workspace, pl, w, c_in, c_out, taps, dtype, stream, 0, false, fuse_reduce, x_segs > 1 ? x_segs * c_in : 0,
                   eight_waves);
}

extern "C" int64_t vnqa_gemm_tn_workspace(int32_t m, int32_t n, int32_t k, int32_t dtype) {
  if (m <= 0 || n <= 0 || k <= 0) return 0;
  const Plan pl = make_plan_k(k, n, m, 1, dtype);
  return ((int64_t)pl.slices * m * n + (int64_t)pl.colsum_blocks * m) * 4;
}

// out[m][n] = sum_k a[k][m] * b[k][n]  ==  wgrad with taps = 1, "dy" = a, "x" = b, pixels = k
extern "C" int vnqa_gemm_tn(const void* a_km, const void* b_kn, float* out, void* workspace, int32_t m, int32_t n,
                            int32_t k, int32_t dtype, void* stream) {
  VNQA_CHECK_ARG(a_km && b_kn && out && workspace, "gemm_tn: null pointer");
  VNQA_CHECK_ARG(dtype == VNQA_BF16 || dtype == VNQA_F32, "gemm_tn: bad dtype %d", dtype);
  VNQA_CHECK_ARG(m % 8 == 0 && n % 8 == 0 && m >= 8 && n >= 8 && k > 0, "gemm_tn: m, n must be multiples of 8");
  const Plan pl = make_plan_k(k, n, m, 1, dtype);
  return wgrad_run(b_kn, a_km, out, nullptr, workspace, pl, 0, n, m, 1, dtype, stream);
}

static int wgrad_run(const void* x, const void* dy, float* dwt, float* dbias, void* workspace, const Plan& pl,
                     int32_t w, int32_t c_in, int32_t c_out, int32_t taps, int32_t dtype, void* stream, int32_t h, bool small,
                     bool fuse_reduce, int x_cs, bool eight_waves) {
  hipStream_t st = (hipStream_t)stream;
  WgradArgs a;
  a.x = (const char*)x;
  a.dy = (const char*)dy;
  a.out = pl.slices == 1 ? dwt : (float*)workspace;
  a.Ptot = pl.Ptot;
  a.Vtot = pl.Vtot;
  a.vrow = pl.vrow;
  a.prow = pl.prow;
  a.vw = pl.vw;
  a.vw_magic = pl.vw > 0 ? (unsigned)(((1ull << 32) + pl.vw - 1) / pl.vw) : 0u;
  a.Wp = w + 2;
  a.Hp = h + 2;
  a.Cin = c_in;
  a.Cout = c_out;
  a.taps = taps;
  a.tilesCo = pl.tilesCo;
  a.tilesCi = pl.tilesCi;
  a.ksteps_total = pl.ksteps_total;
  a.ksteps_per_slice = pl.ksteps_per_slice;
  a.slices = pl.slices;
  a.taps_real = taps;
  a.x_cs = x_cs > 0 ? x_cs : c_in;
  a.final = nullptr;
  // (opt-in through the VNQA_WGRAD_FUSED_REDUCE bit of vnqa_conv2d_wgrad's dtype argument: measured SLOWER end to end — same-box A/B 845 vs 869 clips/s over three rounds: the agent-scope release / acquire
  // fences write back and invalidate L2 lines across the 8 XCDs for every workgroup, which costs more than the 20 us reduce
  // launches it saves)
  if (fuse_reduce && !small && dtype == VNQA_BF16 && pl.slices > 1 && c_in % 4 == 0 &&
      (long long)pl.tilesCo * taps * pl.tilesCi <= 8192)
    a.final = dwt;
  const int lds = 2 * 65536;
  int grid = pl.slices * pl.tilesCo * taps * pl.tilesCi;
  int n_slabs = pl.slices;
  if (small) {
    a.taps = (taps + 256 / c_in - 1) / (256 / c_in);          // tap groups
    a.tilesCo = a.tilesCi = 1;
    a.out = (float*)workspace;
    grid = pl.slices * a.taps;
    n_slabs = 2 * pl.slices;
  }
  static std::atomic<bool> attr_done[2] = {{false}, {false}};   // idempotent attribute call: a race only repeats it
  if (small) {
    auto kern = conv_wgrad_kernel<vnqa_bf16, true>;
    static std::atomic<bool> small_done{false};
    if (!small_done) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        vnqa_set_error("conv3d_wgrad: cannot reserve %d B of LDS", lds);
        return VNQA_ERR_HIP;
      }
      small_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
  } else if (dtype == VNQA_BF16 && !eight_waves && pl.Ptot < (1ll << 31) &&
             (pl.vrow == 0 || pl.vrow / pl.vw >= 32 / pl.vw + 1)) {      // (its branch-free row stepping carries once per 32 pixels)
    auto kern = conv_wgrad4_kernel;
    static std::atomic<bool> four_done{false};
    if (!four_done) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        vnqa_set_error("conv2d_wgrad: cannot reserve %d B of LDS", lds);
        return VNQA_ERR_HIP;
      }
      four_done = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
  } else if (dtype == VNQA_BF16) {
    auto kern = conv_wgrad_kernel<vnqa_bf16>;
    if (!attr_done[0]) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        vnqa_set_error("conv2d_wgrad: cannot reserve %d B of LDS", lds);
        return VNQA_ERR_HIP;
      }
      attr_done[0] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
  } else {
    auto kern = conv_wgrad_kernel<float>;
    if (!attr_done[1]) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
        vnqa_set_error("conv2d_wgrad: cannot reserve %d B of LDS", lds);
        return VNQA_ERR_HIP;
      }
      attr_done[1] = true;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, st, a);
  }
  VNQA_CHECK_LAUNCH();
  const size_t n = (size_t)c_out * taps * c_in;
  if (n_slabs > 1 && a.final == nullptr) {
    if (n % 4 == 0 && n_slabs >= 16 && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dwt & 15) == 0) {
      int g = (int)((n / 4 + 255) / 256);
      g = g > 2048 ? 2048 : g;
      hipLaunchKernelGGL(slab_reduce4_kernel, dim3(g), dim3(256), 0, st, (const float4*)workspace, (float4*)dwt, n / 4, n_slabs);
    } else {
      int g = (int)((n + 255) / 256);
      g = g > 2048 ? 2048 : g;
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(g), dim3(256), 0, st, (const float*)workspace, dwt, n, n_slabs);
    }
    VNQA_CHECK_LAUNCH();
  }
  if (dbias != nullptr) {
    float* partial = (float*)workspace + (size_t)n_slabs * n;
    const long long rpb = (pl.Ptot + pl.colsum_blocks - 1) / pl.colsum_blocks;
    if (dtype == VNQA_BF16)
      hipLaunchKernelGGL(colsum_partial_kernel<vnqa_bf16>, dim3((c_out + 63) / 64, pl.colsum_blocks), dim3(256), 0, st,
                         (const vnqa_bf16*)dy, partial, pl.Ptot, c_out, rpb);
    else
      hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3((c_out + 63) / 64, pl.colsum_blocks), dim3(256), 0, st,
                         (const float*)dy, partial, pl.Ptot, c_out, rpb);
    VNQA_CHECK_LAUNCH();
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((c_out + 63) / 64), dim3(64), 0, st, (const float*)partial, dbias,
                       (size_t)c_out, pl.colsum_blocks);
    VNQA_CHECK_LAUNCH();
  }
  return VNQA_OK;
}
