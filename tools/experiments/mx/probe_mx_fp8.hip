// probe: v_mfma_scale_f32_16x16x128_f8f6f4 with fp8 (e4m3) operands and UNIFORM E8M0 scales.
// Each lane supplies 32 bytes of A (row = lane & 15, K group = lane >> 4) and 32 bytes of B (col = lane & 15, same group), read from
// row-major [16][128] byte matrices with the SAME addressing: bytes [32 g, 32 g + 32) of the row.  Checks D = 2^s * A . B^T.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void k(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  v8i a, b;
  const int* pa = (const int*)(A + r * 128 + g * 32);
  const int* pb = (const int*)(B + r * 128 + g * 32);
  for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
  v4f c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  // C/D layout of 16x16: col = lane & 15, row = (lane >> 4) * 4 + reg
  for (int e = 0; e < 4; ++e) D[(g * 4 + e) * 16 + r] = c[e];
}

static float e4m3(uint8_t v) {   // OCP e4m3fn
  int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x;
  if (e == 0) x = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) x = NAN;
  else x = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -x : x;
}

int main() {
  uint8_t hA[16 * 128], hB[16 * 128];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) { do { hA[i] = rand() & 0xff; } while ((hA[i] & 0x7f) == 0x7f || (hA[i] & 0x78) > 0x48);
                                       do { hB[i] = rand() & 0xff; } while ((hB[i] & 0x7f) == 0x7f || (hB[i] & 0x78) > 0x48); }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int trial = 0; trial < 3; ++trial) {
    const int ea = trial == 0 ? 127 : (trial == 1 ? 109 : 120), eb = trial == 2 ? 130 : 127;
    const int sa = ea * 0x01010101, sb = eb * 0x01010101;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
    float hD[256];
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
      double ref = 0;
      for (int kk = 0; kk < 128; ++kk) ref += (double)e4m3(hA[i * 128 + kk]) * e4m3(hB[j * 128 + kk]);
      ref = ldexp(ref, (ea - 127) + (eb - 127));
      // which orientation? try D[i][j] (A row i = output row) 
      double d1 = hD[i * 16 + j];
      maxerr = fmax(maxerr, fabs(d1 - ref)); maxref = fmax(maxref, fabs(ref));
    }
    printf("trial %d scales 2^%d 2^%d: max |D - ref| = %g of %g\n", trial, ea - 127, eb - 127, maxerr, maxref);
  }
  return 0;
}
