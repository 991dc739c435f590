// What the fp32 -> (hi, lo) fp16 split of the three-product kernels costs on gfx950, alone and between matrix instructions:
//   A: the shipped sequence (csrc/f16x3.h: l3_split8): per pair of values v_fma_mixlo / mixhi (h), v_fma_mixlo / mixhi (m = x s - h): 4 mix
//   B: two v_mul_f32 (x s), v_cvt_pk_f16_f32 (h), v_cvt_f32_f16 + its SDWA form on the high half (h back in fp32), two v_sub_f32 (x s - h),
//      v_cvt_pk_f16_f32 (m): 8 plain instructions per pair, the same bits (every step rounds to nearest even exactly as A does)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/split_cost.hip -o /tmp/split_cost && /tmp/split_cost
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void pairA(float x0, float x1, float s, unsigned& h, unsigned& m) {
  asm volatile("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
               : "=&v"(h), "=&v"(m) : "v"(x0), "v"(x1), "v"(s));
}
__device__ __forceinline__ void pairB2(float x0, float x1, float s, unsigned& h, unsigned& m) {
  float a, b, ha, hb, ra, rb;
  asm volatile("v_mul_f32 %0, %8, %10\n\t"
               "v_mul_f32 %1, %9, %10\n\t"
               "v_cvt_pk_f16_f32 %2, %0, %1\n\t"
               "v_cvt_f32_f16 %3, %2\n\t"
               "v_cvt_f32_f16_sdwa %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
               "v_sub_f32 %5, %0, %3\n\t"
               "v_sub_f32 %6, %1, %4\n\t"
               "v_cvt_pk_f16_f32 %7, %5, %6"
               : "=&v"(a), "=&v"(b), "=&v"(h), "=&v"(ha), "=&v"(hb), "=&v"(ra), "=&v"(rb), "=&v"(m)
               : "v"(x0), "v"(x1), "v"(s));
}

// A with the four pairs of a split interleaved: every instruction depends on the one issued four instructions earlier
__device__ __forceinline__ void quadA(const float (&x)[8], float s, unsigned (&h)[4], unsigned (&m)[4]) {
  asm volatile("v_fma_mixlo_f16 %0, %8, %16, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %1, %10, %16, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %2, %12, %16, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %3, %14, %16, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixhi_f16 %0, %9, %16, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixhi_f16 %1, %11, %16, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixhi_f16 %2, %13, %16, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixhi_f16 %3, %15, %16, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %4, %8, %16, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixlo_f16 %5, %10, %16, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixlo_f16 %6, %12, %16, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixlo_f16 %7, %14, %16, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixhi_f16 %4, %9, %16, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixhi_f16 %5, %11, %16, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixhi_f16 %6, %13, %16, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
               "v_fma_mixhi_f16 %7, %15, %16, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
               : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3])
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(s));
}
// C: sequence B with the four pairs of a split side by side (32 instructions, neighbours independent)
__device__ __forceinline__ void quadC(const float (&x)[8], float s, unsigned (&h)[4], unsigned (&m)[4]) {
  float p[8] = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
  float f[8];
  asm volatile("v_mul_f32 %4, %4, %28\n\t v_mul_f32 %5, %5, %28\n\t v_mul_f32 %6, %6, %28\n\t v_mul_f32 %7, %7, %28\n\t"
               "v_mul_f32 %8, %8, %28\n\t v_mul_f32 %9, %9, %28\n\t v_mul_f32 %10, %10, %28\n\t v_mul_f32 %11, %11, %28\n\t"
               "v_cvt_pk_f16_f32 %0, %4, %5\n\t v_cvt_pk_f16_f32 %1, %6, %7\n\t v_cvt_pk_f16_f32 %2, %8, %9\n\t v_cvt_pk_f16_f32 %3, %10, %11\n\t"
               "v_cvt_f32_f16 %12, %0\n\t v_cvt_f32_f16 %14, %1\n\t v_cvt_f32_f16 %16, %2\n\t v_cvt_f32_f16 %18, %3\n\t"
               "v_cvt_f32_f16_sdwa %13, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
               "v_cvt_f32_f16_sdwa %15, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
               "v_cvt_f32_f16_sdwa %17, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
               "v_cvt_f32_f16_sdwa %19, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
               "v_sub_f32 %4, %4, %12\n\t v_sub_f32 %5, %5, %13\n\t v_sub_f32 %6, %6, %14\n\t v_sub_f32 %7, %7, %15\n\t"
               "v_sub_f32 %8, %8, %16\n\t v_sub_f32 %9, %9, %17\n\t v_sub_f32 %10, %10, %18\n\t v_sub_f32 %11, %11, %19\n\t"
               "v_cvt_pk_f16_f32 %20, %4, %5\n\t v_cvt_pk_f16_f32 %21, %6, %7\n\t v_cvt_pk_f16_f32 %22, %8, %9\n\t v_cvt_pk_f16_f32 %23, %10, %11"
               : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]),
                 "+v"(p[6]), "+v"(p[7]), "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7]),
                 "=&v"(m[0]), "=&v"(m[1]), "=&v"(m[2]), "=&v"(m[3])
               : "v"(s), "v"(s), "v"(s), "v"(s), "v"(s));
}
// sixteen v_fma_mixlo into sixteen different registers: the issue rate without any dependence
__device__ __forceinline__ void indep16(const float (&x)[8], float s, unsigned (&h)[4], unsigned (&m)[4]) {
  unsigned d[16];
  asm volatile("v_fma_mixlo_f16 %0, %16, %24, 0 op_sel_hi:[0,0,0]\n\t v_fma_mixlo_f16 %1, %17, %24, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %2, %18, %24, 0 op_sel_hi:[0,0,0]\n\t v_fma_mixlo_f16 %3, %19, %24, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %4, %20, %24, 0 op_sel_hi:[0,0,0]\n\t v_fma_mixlo_f16 %5, %21, %24, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %6, %22, %24, 0 op_sel_hi:[0,0,0]\n\t v_fma_mixlo_f16 %7, %23, %24, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %8, %16, %24, 0 op_sel_hi:[0,0,0]\n\t v_fma_mixlo_f16 %9, %17, %24, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %10, %18, %24, 0 op_sel_hi:[0,0,0]\n\t v_fma_mixlo_f16 %11, %19, %24, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %12, %20, %24, 0 op_sel_hi:[0,0,0]\n\t v_fma_mixlo_f16 %13, %21, %24, 0 op_sel_hi:[0,0,0]\n\t"
               "v_fma_mixlo_f16 %14, %22, %24, 0 op_sel_hi:[0,0,0]\n\t v_fma_mixlo_f16 %15, %23, %24, 0 op_sel_hi:[0,0,0]"
               : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]),
                 "=&v"(d[10]), "=&v"(d[11]), "=&v"(d[12]), "=&v"(d[13]), "=&v"(d[14]), "=&v"(d[15])
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]), "v"(s));
  for (int i = 0; i < 4; ++i) { h[i] = d[i] ^ d[4 + i]; m[i] = d[8 + i] ^ d[12 + i]; }
}

template <int KIND>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, unsigned* sink, const float* in, int iters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = in[lane * 8 + i];
  const float s = in[600];
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    unsigned h[4], m[4];
    if (KIND == 4 || KIND == 5) {
      quadA(x, s, h, m);
      if (KIND == 5)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n\tv_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n\tv_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n\t"
                     "v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n\tv_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n\tv_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n\t"
                     "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n\tv_mfma_f32_16x16x32_f16 %7, %8, %9, %7"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b));
    } else if (KIND == 6) {
      indep16(x, s, h, m);
    } else if (KIND == 7 || KIND == 8) {
      quadC(x, s, h, m);
      if (KIND == 8)
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %8, %9, %0\n\tv_mfma_f32_16x16x32_f16 %1, %8, %9, %1\n\tv_mfma_f32_16x16x32_f16 %2, %8, %9, %2\n\t"
                     "v_mfma_f32_16x16x32_f16 %3, %8, %9, %3\n\tv_mfma_f32_16x16x32_f16 %4, %8, %9, %4\n\tv_mfma_f32_16x16x32_f16 %5, %8, %9, %5\n\t"
                     "v_mfma_f32_16x16x32_f16 %6, %8, %9, %6\n\tv_mfma_f32_16x16x32_f16 %7, %8, %9, %7"
                     : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "v"(a), "v"(b));
    } else
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      if (KIND == 0 || KIND == 2) pairA(x[2 * p], x[2 * p + 1], s, h[p], m[p]);
      else pairB2(x[2 * p], x[2 * p + 1], s, h[p], m[p]);
      if (KIND >= 2) {                                     // two matrix instructions per pair: 8 per split of eight values
        asm volatile("v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %1, %2, %3, %1"
                     : "+v"(c[2 * p]), "+v"(c[2 * p + 1]) : "v"(a), "v"(b));
      }
    }
    acc ^= h[0] ^ h[1] ^ h[2] ^ h[3] ^ m[0] ^ m[1] ^ m[2] ^ m[3];
    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[wave] = t1 - t0;
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) sum += c[i].x + c[i].y;
  sink[threadIdx.x] = acc + (unsigned)sum;
}

// same bits?
__global__ void check(const float* in, unsigned* out) {
  const int t = threadIdx.x;
  unsigned h1, m1, h2, m2;
  pairA(in[2 * t], in[2 * t + 1], in[600], h1, m1);
  pairB2(in[2 * t], in[2 * t + 1], in[600], h2, m2);
  out[4 * t] = h1; out[4 * t + 1] = m1; out[4 * t + 2] = h2; out[4 * t + 3] = m2;
}

template <int KIND>
void run(const char* name, int nw, unsigned long long* dout, unsigned* sink, const float* in) {
  const int iters = 2000;
  probe<KIND><<<1, 64 * nw>>>(dout, sink, in, iters);
  probe<KIND><<<1, 64 * nw>>>(dout, sink, in, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-72s nw=%d  clocks per split of eight values:", name, nw);
  for (int w = 0; w < nw; w += 4) printf(" wave%d %.1f", w, (double)h[w] / iters);
  printf("\n");
}

int main() {
  unsigned long long* dout; unsigned* sink; float* in; unsigned* chk;
  (void)hipMalloc(&dout, 64); (void)hipMalloc(&sink, 4096); (void)hipMalloc(&in, 4096); (void)hipMalloc(&chk, 16 * 256);
  float hin[1024];
  unsigned seed = 12345u;
  for (int i = 0; i < 1024; ++i) {
    seed = seed * 1664525u + 1013904223u;
    const float mag = (float)((seed >> 8) & 0xffff) / 65536.f;
    const int ex = (int)((seed >> 24) % 40) - 30;
    hin[i] = ((seed & 1) ? -1.f : 1.f) * ldexpf(0.5f + mag, ex);
  }
  hin[600] = 4096.f;
  hin[4] = 0.f; hin[5] = -0.f; hin[6] = 1e-30f; hin[7] = 3.0f;
  (void)hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
  check<<<1, 256>>>(in, chk);
  unsigned hc[1024];
  (void)hipMemcpy(hc, chk, sizeof(hc), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) bad += (hc[4 * t] != hc[4 * t + 2]) + (hc[4 * t + 1] != hc[4 * t + 3]);
  for (int t = 0; t < 256; ++t)
    if (hc[4 * t] != hc[4 * t + 2] || hc[4 * t + 1] != hc[4 * t + 3])
      printf("  pair %d: x = %g %g | A h %08x m %08x | B h %08x m %08x\n", t, hin[2 * t], hin[2 * t + 1], hc[4 * t], hc[4 * t + 1], hc[4 * t + 2], hc[4 * t + 3]);
  printf("sequence B == sequence A on 512 values of 40 binades (zeros, tiny values): %s (%d differences)\n", bad ? "NO" : "yes", bad);
  run<0>("A (4 v_fma_mix per pair)", 4, dout, sink, in);
  run<1>("B (mul, mul, cvt_pk, cvt, cvt_sdwa, sub, sub, cvt_pk per pair)", 4, dout, sink, in);
  run<4>("A interleaved (the four pairs side by side: dependence distance 4)", 4, dout, sink, in);
  run<6>("16 independent v_fma_mixlo (issue rate)", 4, dout, sink, in);
  run<5>("A interleaved, then 8 MFMA", 4, dout, sink, in);
  run<5>("A interleaved, then 8 MFMA, both waves", 8, dout, sink, in);
  run<7>("C (B with the four pairs side by side: 32 instructions)", 4, dout, sink, in);
  run<7>("C, both waves", 8, dout, sink, in);
  run<8>("C, then 8 MFMA", 4, dout, sink, in);
  run<8>("C, then 8 MFMA, both waves", 8, dout, sink, in);
  run<0>("A, both waves of the SIMD", 8, dout, sink, in);
  run<1>("B, both waves of the SIMD", 8, dout, sink, in);
  run<2>("A + 8 MFMA 16x16x32 (two behind every pair)", 4, dout, sink, in);
  run<3>("B + 8 MFMA", 4, dout, sink, in);
  run<2>("A + 8 MFMA, both waves", 8, dout, sink, in);
  run<3>("B + 8 MFMA, both waves", 8, dout, sink, in);
  return 0;
}
