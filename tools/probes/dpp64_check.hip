// v_mov_b64_dpp row_newbcast: does lane k of each 16-lane row reach all 16 lanes with both dwords?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* o) {
  unsigned long long x = ((unsigned long long)(1000 + threadIdx.x) << 32) | (unsigned)(threadIdx.x * 3 + 7), r;
  asm volatile("s_nop 1\n v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
  o[threadIdx.x] = r;
}
int main() {
  unsigned long long* d; unsigned long long h[64];
  (void)hipMalloc(&d, 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    const int src = (i & ~15) + 5;
    const unsigned long long want = ((unsigned long long)(1000 + src) << 32) | (unsigned)(src * 3 + 7);
    if (h[i] != want) ++bad;
  }
  printf("v_mov_b64_dpp row_newbcast:5 -> %s (%d lanes wrong)\n", bad ? "WRONG" : "ok", bad);
  return 0;
}
