// experiment: does gfx950 execute scalar atomics (s_atomic_add with return)?  One wave-uniform draw per wave from a global counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned* c, unsigned* out) {
  unsigned v = 1;
  unsigned long long p = (unsigned long long)c;
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(p) : "memory");
  if (threadIdx.x % 64 == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = v;
}
int main() {
  unsigned *c, *out;
  const int blocks = 1024, waves = 4;
  hipMalloc(&c, 4); hipMalloc(&out, blocks * waves * 4);
  hipMemset(c, 0, 4);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), 0, 0, c, out);
  std::vector<unsigned> h(blocks * waves);
  unsigned total = 0;
  hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(&total, c, 4, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  bool ok = total == (unsigned)h.size();
  for (size_t i = 0; i < h.size(); ++i) ok = ok && h[i] == i;
  printf("counter %u (expected %zu), draws form 0..n-1: %s\n", total, h.size(), ok ? "yes" : "NO");
  return ok ? 0 : 1;
}
