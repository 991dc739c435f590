// The x stream of the three-product GEMM kernels in isolation: a wave reads tiles of 32 rows x 128 bytes per k-step (lane (j, g): row j,
// bytes 32 g .. 32 g + 31 of the k-step, two 16-byte loads -- the B-operand layout of v_mfma_f32_16x16x32_f16), RING k-steps ahead, all
// workgroups walking k from 0.  Question: does the ROW STRIDE (K * 4 bytes, a multiple of 512) camp the reads on a few L2 / fabric channels?
// The same bytes are read with the stride padded by 128 / 256 / ... bytes.   hipcc --offload-arch=gfx950 -O3 row_stride.hip -o /tmp/row_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int RING, bool FULL>
__global__ __launch_bounds__(512) void stream_rows(const char* __restrict__ x, float* __restrict__ sink, int M, int KS, long long stride,
                                                   int passes, int rot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
  const int ranges = gridDim.x / passes, range = blockIdx.x % ranges;   // pass-major: the passes of a row range share an XCD (ranges % 8 == 0)
  const int WT = (M + 31) / 32;
  const int t0 = (int)((long long)WT * range / ranges), t1 = (int)((long long)WT * (range + 1) / ranges);
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = t0 + wave; t < t1; t += 8) {
    // FULL: a wave-instruction reads 8 rows x 128 B (whole lines) instead of 16 rows x 64 B (the operand layout's half lines)
    const char* r0 = FULL ? x + (long long)min(t * 32 + (lane >> 3), M - 1) * stride + 16 * (lane & 7) - 0
                          : x + (long long)min(t * 32 + j, M - 1) * stride + 32 * g;
    const char* r1 = FULL ? x + (long long)min(t * 32 + 16 + (lane >> 3), M - 1) * stride + 16 * (lane & 7)
                          : x + (long long)min(t * 32 + 16 + j, M - 1) * stride + 32 * g;
    const long long o1 = FULL ? 8 * stride : 16;
    const int k0 = rot ? (int)(((unsigned)t * 2654435761u >> 8) % (unsigned)KS) : 0;   // rot: every tile starts at its own k-step
    f4 ring[RING][4];
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const int ks = (k0 + u) % KS;
      ring[u][0] = *(const f4*)(r0 + ks * 128); ring[u][1] = *(const f4*)(r0 + ks * 128 + o1);
      ring[u][2] = *(const f4*)(r1 + ks * 128); ring[u][3] = *(const f4*)(r1 + ks * 128 + o1);
    }
    for (int s = 0; s < KS; s += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        acc += ring[u][0] + ring[u][1] + ring[u][2] + ring[u][3];
        const int ks = (k0 + s + u + RING) % KS;
        ring[u][0] = *(const f4*)(r0 + ks * 128); ring[u][1] = *(const f4*)(r0 + ks * 128 + o1);
        ring[u][2] = *(const f4*)(r1 + ks * 128); ring[u][3] = *(const f4*)(r1 + ks * 128 + o1);
      }
    }
#pragma unroll
    for (int u = 0; u < RING; ++u) acc += ring[u][0];
  }
  if (acc.x == 123456.789f) sink[threadIdx.x] = acc.y;
}

int main() {
  const size_t cap = 1ull << 30;
  char* buf; float* sink;
  hipMalloc(&buf, cap); hipMalloc(&sink, 4096);
  hipMemset(buf, 0, cap);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct Case { const char* name; int M, K, passes, ranges; } cases[] = {
      {"s3_fc2 18400x1536 6 passes", 18400, 1536, 6, 40}, {"s3_fc2 1 pass", 18400, 1536, 1, 240}, {"s3_proj 18400x384 6 passes", 18400, 384, 6, 40},
      {"enc 96600x256 2 passes", 96600, 256, 2, 128},     {"enc 96600x256 1 pass", 96600, 256, 1, 256}, {"s4_fc2 4600x3072 12 passes", 4600, 3072, 12, 18},
      {"dec_kv 73600x256 6 passes", 73600, 256, 6, 40}};
  for (auto& c : cases) {
    for (int full = 0; full < 2; ++full)
      for (int pad = 0; pad <= 128; pad += 128) {
        const int rot = 0;
        const long long stride = (long long)c.K * 4 + pad;
        if ((size_t)c.M * stride > cap) continue;
        const int grid = c.passes * c.ranges;
        float best = 1e30f;
        for (int it = 0; it < 6; ++it) {
          hipEventRecord(e0);
          if (full) hipLaunchKernelGGL((stream_rows<4, true>), dim3(grid), dim3(512), 0, 0, buf, sink, c.M, c.K / 32, stride, c.passes, rot);
          else hipLaunchKernelGGL((stream_rows<4, false>), dim3(grid), dim3(512), 0, 0, buf, sink, c.M, c.K / 32, stride, c.passes, rot);
          hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (it > 0 && ms < best) best = ms;
        }
        const double bytes = (double)c.M * c.K * 4 * c.passes;
        printf("%-30s full-lines %d stride %6lld (+%3d): %8.1f us  %6.2f TB/s of L1 reads, %4d k-steps -> %6.0f clk/k-step at 2.4 GHz\n", c.name, full, stride, pad,
               best * 1e3, bytes / (best * 1e-3) / 1e12, c.K / 32, best * 1e-3 * 2.4e9 / (c.K / 32) / ((c.M + 31) / 32 / 8.0 / c.ranges + 1e-9));
      }
  }
  return 0;
}
