// ds_read_b128 throughput by address pattern (all 8 waves of one workgroup reading): lane-linear (lane x 16 B: one contiguous KB per
// wave instruction) against four 256-byte runs 1 KB apart (the A-fragment pattern of the fused MLP: (lane >> 4) x 1024 + (lane & 15) x 16),
// 512 B apart, and 256-byte runs at 272-byte pitch.   hipcc --offload-arch=gfx950 -O2 tools/probes/lds_pattern.hip -o /tmp/lds_pattern && /tmp/lds_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(unsigned long long* out, unsigned* sink, int pattern, int iters) {
  extern __shared__ u32x4 lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (u32x4){(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned addr;
  if (pattern == 0) addr = lane * 16;
  else if (pattern == 1) addr = (lane >> 4) * 1024 + (lane & 15) * 16;
  else if (pattern == 2) addr = (lane >> 4) * 512 + (lane & 15) * 16;
  else if (pattern == 3) addr = (lane >> 4) * 272 + (lane & 15) * 16;
  else addr = (lane >> 4) * 4096 + (lane & 15) * 16;
  addr += wave * 8192;                                           // every wave its own 8 KB window (64 KB in all)
  u32x4 r0, r1, r2, r3;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:256\n ds_read_b128 %2, %4 offset:4096\n ds_read_b128 %3, %4 offset:4352\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(addr));
    acc ^= r0.x ^ r1.y ^ r2.z ^ r3.w;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[wave] = t1 - t0;
  sink[threadIdx.x] = acc;
}

// the fused MLP's fragment stream: batches of four reads (hi / lo part of two 16-row blocks), TWO batches in flight behind the one awaited
__global__ __launch_bounds__(512) void stream(unsigned long long* out, unsigned* sink, int inflight, int iters) {
  extern __shared__ u32x4 lds[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (u32x4){(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned base = (lane >> 4) * 1024 + (lane & 15) * 16;     // every wave reads the SAME 32-KB image (as the kernel's waves do)
  u32x4 r[12];
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const unsigned a = base + t * 4096;
      u32x4& d0 = r[(t % 3) * 4], &d1 = r[(t % 3) * 4 + 1], &d2 = r[(t % 3) * 4 + 2], &d3 = r[(t % 3) * 4 + 3];
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:512\n ds_read_b128 %2, %4 offset:256\n ds_read_b128 %3, %4 offset:768\n"
                   : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a));
      if (inflight == 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      else if (inflight == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int q = 0; q < 12; ++q) acc ^= r[q].x;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[wave] = t1 - t0;
  sink[threadIdx.x] = acc;
}

int main() {
  unsigned long long* dout; unsigned* sink;
  (void)hipMalloc(&dout, 64); (void)hipMalloc(&sink, 4096);
  (void)hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  const char* names[] = {"lane-linear (1 KB contiguous)", "4 x 256 B, 1 KB apart (fused MLP, W1 fragments)", "4 x 256 B, 512 B apart", "4 x 256 B at 272-byte pitch", "4 x 256 B, 4 KB apart"};
  for (int nw = 4; nw <= 8; nw += 4)
    for (int p = 0; p < 5; ++p) {
      const int iters = 4000;
      probe<<<1, 64 * nw, 136 * 1024>>>(dout, sink, p, iters);
      probe<<<1, 64 * nw, 136 * 1024>>>(dout, sink, p, iters);
      (void)hipDeviceSynchronize();
      unsigned long long h[8];
      (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
      double worst = 0;
      for (int w = 0; w < nw; ++w) worst = h[w] > worst ? (double)h[w] : worst;
      printf("%-52s %d waves: %.1f clocks per 4 reads and wave; %.0f B/clk for the CU\n", names[p], nw, worst / iters, nw * 4096.0 * iters / worst);
    }
  (void)hipFuncSetAttribute((const void*)stream, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  for (int nw = 4; nw <= 8; nw += 4)
    for (int inflight = 0; inflight <= 2; ++inflight) {
      const int iters = 1000;
      stream<<<1, 64 * nw, 136 * 1024>>>(dout, sink, inflight, iters);
      stream<<<1, 64 * nw, 136 * 1024>>>(dout, sink, inflight, iters);
      (void)hipDeviceSynchronize();
      unsigned long long h[8];
      (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
      double worst = 0;
      for (int w = 0; w < nw; ++w) worst = h[w] > worst ? (double)h[w] : worst;
      printf("fragment stream, %d batches in flight behind the awaited one, %d waves: %.1f clocks per batch of 4 reads and wave; %.0f B/clk for the CU\n", inflight, nw,
             worst / iters / 8, nw * 4096.0 * 8 * iters / worst);
    }
  return 0;
}
