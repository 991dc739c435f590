// Does vector-ALU / LDS work overlap the matrix pipe on gfx950, inside one wave and between the two waves of a SIMD?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
// One workgroup on one CU; NW waves (4 = one per SIMD, 8 = two per SIMD).  Every wave runs `iters` repetitions of a block chosen by its
// role; s_memtime clocks per block are printed per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define M16(acc) "v_mfma_f32_16x16x32_f16 %" #acc ", %8, %9, %" #acc "\n"
#define VMIX(d) "v_fma_mixlo_f16 %" #d ", %14, %15, 0 op_sel_hi:[0,0,0]\n"
#define VMAX(d) "v_max3_f32 %" #d ", |%14|, |%15|, |%14|\n"
#define VMUL(d) "v_mul_f32 %" #d ", %14, %15\n"
#define DSR(d) "ds_read_b128 %" #d ", %16\n"

// block kinds: 0: 8 MFMA | 1: 24 v_fma_mix | 2: 8 x (M + 3 mix) | 3: 8 x (M + 2 mix) | 4: 8 x (M + 1 mix) | 5: 8 x (M + 4 mix)
//              6: 8 x (M + ds_read_b128) then wait | 7: 8 ds_read_b128 then wait | 8: 8 x (M + 3 v_mul) | 9: 24 v_mul
//              10: 8 x (M + 1 ds_read + 2 mix) | 11: 8 dependent MFMAs on ONE accumulator | 12: 8 x (M + 3 v_max3)
template <int KIND>
__device__ __forceinline__ void block(f32x4 (&c)[8], f16x8 a, f16x8 b, unsigned (&t)[4], float x, float y, unsigned addr, u32x4 (&r)[2]) {
#define OPS "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) /*0-7*/ \
            : "v"(a), "v"(b) /*8,9*/, "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]) /*10-13 (placeholders)*/, "v"(x), "v"(y) /*14,15*/, "v"(addr) /*16*/
  if constexpr (KIND == 0) asm volatile(M16(0) M16(1) M16(2) M16(3) M16(4) M16(5) M16(6) M16(7) : OPS);
  else if constexpr (KIND == 11) asm volatile(M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) M16(0) : OPS);
  else if constexpr (KIND == 13) asm volatile(M16(0) M16(1) M16(0) M16(1) M16(0) M16(1) M16(0) M16(1) : OPS);      // two accumulators, alternating
  else if constexpr (KIND == 14) asm volatile(M16(0) M16(0) M16(0) M16(0) M16(1) M16(1) M16(1) M16(1) : OPS);      // two accumulators, runs of four
  else if constexpr (KIND == 15) asm volatile(M16(0) M16(1) M16(2) M16(0) M16(1) M16(2) M16(0) M16(1) : OPS);      // three accumulators, round-robin
  else if constexpr (KIND == 16) asm volatile(M16(0) M16(1) M16(2) M16(3) M16(0) M16(1) M16(2) M16(3) : OPS);      // four accumulators, round-robin
  else if constexpr (KIND == 1 || KIND == 9) {
    unsigned d0, d1, d2, d3;
    if constexpr (KIND == 1)
      asm volatile("v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %4, %5, 0 op_sel_hi:[0,0,0]\n"
                   "v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %4, %5, 0 op_sel_hi:[0,0,0]\n"
                   "v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %4, %5, 0 op_sel_hi:[0,0,0]\n"
                   "v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %4, %5, 0 op_sel_hi:[0,0,0]\n"
                   "v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %4, %5, 0 op_sel_hi:[0,0,0]\n"
                   "v_fma_mixlo_f16 %0, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %4, %5, 0 op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %4, %5, 0 op_sel_hi:[0,0,0]\n"
                   : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(x), "v"(y));
    else
      asm volatile("v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %5\n v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %5\n v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %5\n v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %5\n"
                   "v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %5\n v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %5\n v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %5\n v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %5\n"
                   "v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %5\n v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %5\n v_mul_f32 %0, %4, %5\n v_mul_f32 %1, %4, %5\n v_mul_f32 %2, %4, %5\n v_mul_f32 %3, %4, %5\n"
                   : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(x), "v"(y));
    t[0] ^= d0 ^ d1 ^ d2 ^ d3;
  } else if constexpr (KIND == 7) {
    asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:1024\n ds_read_b128 %0, %2 offset:2048\n ds_read_b128 %1, %2 offset:3072\n"
                 "ds_read_b128 %0, %2 offset:4096\n ds_read_b128 %1, %2 offset:5120\n ds_read_b128 %0, %2 offset:6144\n ds_read_b128 %1, %2 offset:7168\n s_waitcnt lgkmcnt(0)\n"
                 : "=&v"(r[0]), "=&v"(r[1]) : "v"(addr));
  } else {
    unsigned d0, d1, d2, d3;
    u32x4 r0, r1;
#define OUT2 "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) /*0-7*/, "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) /*8-11*/, "=&v"(r0), "=&v"(r1) /*12,13*/
#define IN2 "v"(a), "v"(b) /*14,15*/, "v"(x), "v"(y) /*16,17*/, "v"(addr) /*18*/
#define MM(i) "v_mfma_f32_16x16x32_f16 %" #i ", %14, %15, %" #i "\n"
#define X1 "v_fma_mixlo_f16 %8, %16, %17, 0 op_sel_hi:[0,0,0]\n"
#define X2 X1 "v_fma_mixlo_f16 %9, %16, %17, 0 op_sel_hi:[0,0,0]\n"
#define X3 X2 "v_fma_mixlo_f16 %10, %16, %17, 0 op_sel_hi:[0,0,0]\n"
#define X4 X3 "v_fma_mixlo_f16 %11, %16, %17, 0 op_sel_hi:[0,0,0]\n"
#define U3 "v_mul_f32 %8, %16, %17\n v_mul_f32 %9, %16, %17\n v_mul_f32 %10, %16, %17\n"
#define W3 "v_max3_f32 %8, |%16|, |%17|, |%16|\n v_max3_f32 %9, |%16|, |%17|, |%16|\n v_max3_f32 %10, |%16|, |%17|, |%16|\n"
#define R1(o) "ds_read_b128 %12, %18 offset:" #o "\n"
#define R2(o) "ds_read_b128 %13, %18 offset:" #o "\n"
    if constexpr (KIND == 2) asm volatile(MM(0) X3 MM(1) X3 MM(2) X3 MM(3) X3 MM(4) X3 MM(5) X3 MM(6) X3 MM(7) X3 : OUT2 : IN2);
    else if constexpr (KIND == 3) asm volatile(MM(0) X2 MM(1) X2 MM(2) X2 MM(3) X2 MM(4) X2 MM(5) X2 MM(6) X2 MM(7) X2 : OUT2 : IN2);
    else if constexpr (KIND == 4) asm volatile(MM(0) X1 MM(1) X1 MM(2) X1 MM(3) X1 MM(4) X1 MM(5) X1 MM(6) X1 MM(7) X1 : OUT2 : IN2);
    else if constexpr (KIND == 5) asm volatile(MM(0) X4 MM(1) X4 MM(2) X4 MM(3) X4 MM(4) X4 MM(5) X4 MM(6) X4 MM(7) X4 : OUT2 : IN2);
    else if constexpr (KIND == 6) asm volatile(MM(0) R1(0) MM(1) R2(1024) MM(2) R1(2048) MM(3) R2(3072) MM(4) R1(4096) MM(5) R2(5120) MM(6) R1(6144) MM(7) R2(7168) "s_waitcnt lgkmcnt(0)\n" : OUT2 : IN2);
    else if constexpr (KIND == 8) asm volatile(MM(0) U3 MM(1) U3 MM(2) U3 MM(3) U3 MM(4) U3 MM(5) U3 MM(6) U3 MM(7) U3 : OUT2 : IN2);
    else if constexpr (KIND == 12) asm volatile(MM(0) W3 MM(1) W3 MM(2) W3 MM(3) W3 MM(4) W3 MM(5) W3 MM(6) W3 MM(7) W3 : OUT2 : IN2);
    else if constexpr (KIND == 10) asm volatile(MM(0) R1(0) X2 MM(1) R2(1024) X2 MM(2) R1(2048) X2 MM(3) R2(3072) X2 MM(4) R1(4096) X2 MM(5) R2(5120) X2 MM(6) R1(6144) X2 MM(7) R2(7168) X2 "s_waitcnt lgkmcnt(0)\n" : OUT2 : IN2);
    t[0] ^= d0;
    r[0] = r0;
  }
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, float* sink, int iters) {
  __shared__ u32x4 lds[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = (u32x4){(unsigned)i, 1u, 2u, 3u};
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
  unsigned t[4] = {1u, 2u, 3u, 4u};
  u32x4 r[2] = {};
  float x = 1.0001f + lane, y = 0.5f;
  unsigned addr = lane * 16;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    for (int it = 0; it < iters; ++it) block<KA>(c, a, b, t, x, y, addr, r);
  } else {
    for (int it = 0; it < iters; ++it) block<KB>(c, a, b, t, x, y, addr, r);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[wave] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += c[i].x + c[i].y + c[i].z + c[i].w;
  sink[threadIdx.x] = s + t[0] + r[0].x + r[1].y;
}

template <int KA, int KB>
void run(const char* name, int nw, unsigned long long* dout, float* sink) {
  const int iters = 2000;
  probe<KA, KB><<<1, 64 * nw>>>(dout, sink, iters);
  probe<KA, KB><<<1, 64 * nw>>>(dout, sink, iters);
  (void)hipDeviceSynchronize();
  unsigned long long h[8];
  (void)hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-64s nw=%d  s_memtime clocks per block, per wave:", name, nw);
  for (int w = 0; w < nw; ++w) printf(" %.1f", (double)h[w] / iters);
  printf("\n");
}

int main() {
  unsigned long long* dout;
  float* sink;
  (void)hipMalloc(&dout, 64);
  (void)hipMalloc(&sink, 4096);
  run<0, 0>("8 MFMA 16x16x32 f16 (8 accumulators)", 4, dout, sink);
  run<0, 0>("8 MFMA, both waves of the SIMD", 8, dout, sink);
  run<11, 11>("8 MFMA on ONE accumulator", 4, dout, sink);
  run<13, 13>("8 MFMA, TWO accumulators alternating (a b a b ...)", 4, dout, sink);
  run<14, 14>("8 MFMA, two accumulators in runs (a a a a b b b b)", 4, dout, sink);
  run<15, 15>("8 MFMA, three accumulators round-robin", 4, dout, sink);
  run<16, 16>("8 MFMA, four accumulators round-robin", 4, dout, sink);
  run<13, 13>("8 MFMA, two accumulators alternating, both waves", 8, dout, sink);
  run<14, 14>("8 MFMA, two accumulators in runs, both waves", 8, dout, sink);
  run<11, 11>("8 MFMA on one accumulator, both waves", 8, dout, sink);
  run<1, 1>("24 v_fma_mixlo", 4, dout, sink);
  run<1, 1>("24 v_fma_mixlo, both waves", 8, dout, sink);
  run<9, 9>("24 v_mul_f32", 4, dout, sink);
  run<4, 4>("8 x (MFMA + 1 mix)", 4, dout, sink);
  run<3, 3>("8 x (MFMA + 2 mix)", 4, dout, sink);
  run<2, 2>("8 x (MFMA + 3 mix)", 4, dout, sink);
  run<5, 5>("8 x (MFMA + 4 mix)", 4, dout, sink);
  run<8, 8>("8 x (MFMA + 3 v_mul)", 4, dout, sink);
  run<12, 12>("8 x (MFMA + 3 v_max3)", 4, dout, sink);
  run<2, 2>("8 x (MFMA + 3 mix), both waves", 8, dout, sink);
  run<3, 3>("8 x (MFMA + 2 mix), both waves", 8, dout, sink);
  run<4, 4>("8 x (MFMA + 1 mix), both waves", 8, dout, sink);
  run<0, 1>("wave A: 8 MFMA | wave B: 24 mix", 8, dout, sink);
  run<0, 9>("wave A: 8 MFMA | wave B: 24 v_mul", 8, dout, sink);
  run<7, 7>("8 ds_read_b128 + wait", 4, dout, sink);
  run<7, 7>("8 ds_read_b128 + wait, both waves", 8, dout, sink);
  run<6, 6>("8 x (MFMA + ds_read_b128) + wait", 4, dout, sink);
  run<6, 6>("8 x (MFMA + ds_read_b128) + wait, both waves", 8, dout, sink);
  run<10, 10>("8 x (MFMA + ds_read_b128 + 2 mix) + wait", 4, dout, sink);
  run<10, 10>("8 x (MFMA + ds_read_b128 + 2 mix) + wait, both waves", 8, dout, sink);
  run<0, 7>("wave A: 8 MFMA | wave B: 8 ds_read_b128", 8, dout, sink);
  return 0;
}
