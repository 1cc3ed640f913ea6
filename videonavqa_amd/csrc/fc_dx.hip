// fc_dx.hip — input gradient of a wide nn.Linear whose weight is held in the FORWARD layout only (gfx950).
//
//   dx[m][k] = sum_r dout[m][r] * nat[r][k]        m < M (320 rows per launch),  r < R = 128,  k < K (K % 128 == 0, here 131 072)
//
// fc_embed_attn (film_attn_pt_stem.py:56-57,244) maps the flattened [S*C] feature map of every packed image to 128
// attention features; its dX is a GEMM with a tiny contraction (128) and a huge output (280 x 131 072 x 2 B = 73 MB), i.e.
// a streaming problem.  Round 1 ran it on the implicit-GEMM kernel: 3072 tiles of 128 x 128 with a 2-step K loop (all
// prologue / epilogue, 83 us), fed by a second, transposed copy of the weight that cost another 42 us per step to make.
// This kernel needs neither:
//   * dout (<= 320 x 128, 80 KiB) is loaded ONCE per workgroup and stays in LDS as the MFMA B operand;
//   * a persistent workgroup walks 128-column slabs of `nat` ([128 r][128 k], 32 KiB, double-buffered LDS-DMA) — the slab is
//     [contraction][output] in memory, so the A fragments (8 consecutive r of one k) come out of LDS through the transposed
//     read ds_read_b64_tr_b16, exactly as conv_wgrad.hip reads its [pixel][channel] tiles;
//   * D[k][m] = sum_r A[k][r] B[r][m]: a lane ends up with 4 consecutive k of one m, i.e. 8 contiguous bytes of dx's row —
//     stored straight from the accumulators; a wave's four k-fragments complete a 128-byte line of each row.
// Traffic = nat once (33 MB) + dx once (73 MB).  16-bit storage formats only (the exact-f32 mode keeps the generic path).
#include "vnqa_common.h"

namespace {

struct FcDxArgs {
  const char* dout;   // [M][R] 16-bit
  const char* nat;    // [R][K] 16-bit
  char* dx;           // [M][K] 16-bit
  int M, K, tiles;
};

constexpr int R = 128;                 // contraction length (rows of nat)
constexpr int MPAD = 320;              // dout rows held in LDS (20 MFMA fragments)
constexpr int TK = 128;                // output columns per slab
constexpr int B_BYTES = MPAD * R * 2;  // 81920
constexpr int A_BYTES = R * TK * 2;    // 32768
constexpr int LDS_BYTES = B_BYTES + 2 * A_BYTES;

typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void glds16f(const char* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// asm-issued so that the compiler does not order the fragment reads behind the NEXT slab's LDS-DMA (conv_wgrad.hip)
template <int OFF>
__device__ __forceinline__ s16x4 tr_read(unsigned addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// 16-byte-chunk swizzle of the slab's 256-byte rows for the transposed reads (two 32-lane groups, 64 banks): the four rows
// of a 16-lane group go to four 64-byte units, the group 8 rows further to the other 32-byte half (as conv_wgrad.hip)
__device__ __forceinline__ int swz_tr(int row) { return ((row & 3) << 2) | (((row >> 3) & 1) << 1); }

__global__ void __launch_bounds__(512) fc_dx_kernel(const FcDxArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ldsB = smem;
  char* const ldsA = smem + B_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wk = wave & 1, wmg = wave >> 1;          // k half (64 columns = 4 fragments), m group (5 fragments of 16 rows)
  const int fr = lane & 15, fh = lane >> 4;
  const int g = lane >> 4, q4 = (lane & 15) >> 2, pp = lane & 3;

  // ---- dout -> LDS, once: 320 rows x 256 B = 80 instructions of 1 KiB (4 rows each), 10 per wave; chunk ^= row & 15 makes the
  // ds_read_b128 fragment reads (16 consecutive rows, k-chunks fh and fh + 1 within a lane group) conflict-free ----
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const int q = wave * 10 + j;
    const int row = q * 4 + (lane >> 4);
    const int logical = (lane & 15) ^ (row & 15);
    const int src_row = row < p.M ? row : p.M - 1;            // rows past M are computed and dropped
    glds16f(p.dout + (size_t)src_row * (R * 2) + logical * 16, ldsB + q * 1024);
  }
  auto stage = [&](int t, int buf) {                          // slab t: nat[0..127][t*128 .. +127] -> 32 instructions, 4 per wave
    char* dst = ldsA + buf * A_BYTES;
    const char* src0 = p.nat + (size_t)t * (TK * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = wave * 4 + j;
      const int row = q * 4 + (lane >> 4);
      const int logical = (lane & 15) ^ swz_tr(row);
      glds16f(src0 + (size_t)row * p.K * 2 + logical * 16, dst + q * 1024);
    }
  };

  int t = blockIdx.x;
  if (t < p.tiles) stage(t, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  int it = 0;
  for (; t < p.tiles; t += gridDim.x, ++it) {
    const int cur = it & 1;
    const int tn = t + gridDim.x;
    if (tn < p.tiles) stage(tn, cur ^ 1);
    const unsigned baseA = lds_addr_of(ldsA + cur * A_BYTES);

    vnqa_f32x4 acc[5][4];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;

#pragma unroll
    for (int s = 0; s < R / 32; ++s) {
      // A fragments: k-fragment j of this wave = slab columns wk*64 + 16 j .. +15, contraction rows 32 s + 8 g .. +7
      const int row0 = 32 * s + 8 * g + q4;
      const int sw = swz_tr(row0);
      const unsigned ra = baseA + row0 * (TK * 2) + ((pp & 1) << 3);
      s16x4 alo[4], ahi[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned a = ra + (((((wk * 64 + j * 16) >> 3) + (pp >> 1)) ^ sw) << 4);
        alo[j] = tr_read<0>(a);
        ahi[j] = tr_read<4 * TK * 2>(a);
      }
      // B fragments: rows (wmg*5 + i)*16 + fr of dout, contraction chunk 4 s + fh
      vnqa_f32x4 bf[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const int row = (wmg * 5 + i) * 16 + fr;
        bf[i] = *(const vnqa_f32x4*)(ldsB + row * (R * 2) + (((4 * s + fh) ^ fr) << 4));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      vnqa_bf16x8 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        asm volatile("" : "+v"(alo[j]), "+v"(ahi[j]));          // the fragments exist only after the wait above
        af[j] = vnqa_bf16x8{alo[j][0], alo[j][1], alo[j][2], alo[j][3], ahi[j][0], ahi[j][1], ahi[j][2], ahi[j][3]};
      }
#pragma unroll
      for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = VNQA_MFMA_16x16x32(af[j], __builtin_bit_cast(vnqa_bf16x8, bf[i]), acc[i][j]);
    }

    // acc[i][j][e] = dx[row (wmg*5+i)*16 + fr][t*128 + wk*64 + 16 j + 4 fh + e]: 8 bytes per lane, the four j of a lane's row
    // form one 128-byte line
    char* const out0 = p.dx + ((size_t)t * TK + wk * 64 + 4 * fh) * 2;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int row = (wmg * 5 + i) * 16 + fr;
      if (row < p.M) {
        char* o = out0 + (size_t)row * p.K * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint2 pk;
          pk.x = pack2_h16(acc[i][j][0], acc[i][j][1]);
          pk.y = pack2_h16(acc[i][j][2], acc[i][j][3]);
          *(uint2*)(o + j * 32) = pk;
        }
      }
    }
    // the next slab has landed and every wave is done reading the current one
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
}

}  // namespace

// dx[m][k] = sum_r dout[m][r] nat[r][k]; dout [m][128], nat [128][k], dx [m][k], all in the library's 16-bit format
// (include/vnqa_hip.h).  Replaces the `nat_t` operand of vnqa_pack_fc_weight + vnqa_gemm_nt for fc_embed_attn's dX.
extern "C" int vnqa_fc_dx(const void* dout, const void* nat, void* dx, int32_t m, int32_t r, int32_t k, int32_t dtype,
                          void* stream) {
  VNQA_CHECK_ARG(dout && nat && dx, "fc_dx: null pointer");
  VNQA_CHECK_ARG(dtype == VNQA_BF16, "fc_dx: 16-bit storage only (dtype %d)", dtype);
  VNQA_CHECK_ARG(r == R, "fc_dx: contraction length %d (this kernel is built for %d)", r, R);
  VNQA_CHECK_ARG(m > 0, "fc_dx: %d rows", m);
  VNQA_CHECK_ARG(k > 0 && k % TK == 0, "fc_dx: k=%d must be a positive multiple of %d", k, TK);
  static std::atomic<bool> attr_set{false};
  if (!attr_set.load(std::memory_order_acquire)) {
    (void)hipFuncSetAttribute((const void*)fc_dx_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set.store(true, std::memory_order_release);
  }
  // more than 320 rows (minibatches above 9 full-length clips): one launch per 320-row block of dout; `nat` (33 MB at the
  // headline size) is re-streamed per block from the Infinity Cache
  for (int m0 = 0; m0 < m; m0 += MPAD) {
    FcDxArgs a;
    a.dout = (const char*)dout + (size_t)m0 * R * 2;
    a.nat = (const char*)nat;
    a.dx = (char*)dx + (size_t)m0 * k * 2;
    a.M = (m - m0) < MPAD ? (m - m0) : MPAD;
    a.K = k; a.tiles = k / TK;
    const int grid = a.tiles < 256 ? a.tiles : 256;
    hipLaunchKernelGGL(fc_dx_kernel, dim3(grid), dim3(512), LDS_BYTES, (hipStream_t)stream, a);
    VNQA_CHECK_LAUNCH();
  }
  return VNQA_OK;
}
