// Per-opcode VALU issue cost on gfx950 (clocks per wave-instruction per SIMD with 2 and 4 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/valu_ops.hip -o /tmp/vo && /tmp/vo
// Each kernel = 64 back-to-back instances of ONE instruction form over 8 rotating destination registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define R8(a, b, c, d, e, f, g, h) a b c d e f g h
#define BODY(OP, TAIL)                                                                                             \
  OP " %0, " TAIL "\n" OP " %1, " TAIL "\n" OP " %2, " TAIL "\n" OP " %3, " TAIL "\n" OP " %4, " TAIL "\n" OP " %5, " TAIL \
     "\n" OP " %6, " TAIL "\n" OP " %7, " TAIL "\n"
#define BODY64(OP, TAIL) BODY(OP, TAIL) BODY(OP, TAIL) BODY(OP, TAIL) BODY(OP, TAIL) BODY(OP, TAIL) BODY(OP, TAIL) BODY(OP, TAIL) BODY(OP, TAIL)

#define PROBE(NAME, OP, TAIL)                                                                                     \
  __global__ void k_##NAME(unsigned long long* out, float* sink, int iters) {                                     \
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;                  \
    float x = 1.0001f + threadIdx.x, y = 0.5f, z = 3.f;                                                           \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                         \
    for (int it = 0; it < iters; ++it)                                                                            \
      asm volatile(BODY64(OP, TAIL)                                                                               \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)               \
                   : "v"(x), "v"(y), "v"(z)                                                                       \
                   : "vcc", "s20", "s21");                                                                        \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                         \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                                 \
    sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                    \
  }

PROBE(fmac, "v_fmac_f32", "%8, %9")
PROBE(fma3, "v_fma_f32", "%8, %9, %10")
PROBE(mul, "v_mul_f32", "%8, %9")
PROBE(add, "v_add_f32", "%8, %9")
PROBE(sub_abs, "v_sub_f32_e64", "|%8|, %9")
PROBE(maxf, "v_max_f32", "%8, %9")
PROBE(minf, "v_min_f32", "%8, %9")
PROBE(med3f, "v_med3_f32", "%8, %9, %10")
PROBE(floorf_, "v_floor_f32", "%8")
PROBE(fract, "v_fract_f32", "%8")
PROBE(cvt_i32, "v_cvt_i32_f32", "%8")
PROBE(cvt_f32, "v_cvt_f32_i32", "%8")
PROBE(rcp, "v_rcp_f32", "%8")
PROBE(exp, "v_exp_f32", "%8")
PROBE(mov, "v_mov_b32", "%8")
PROBE(addu, "v_add_u32", "%8, %9")
PROBE(lshl_add, "v_lshl_add_u32", "%8, 3, %9")
PROBE(add3, "v_add3_u32", "%8, %9, %10")
PROBE(mad24, "v_mad_u32_u24", "%8, %9, %10")
PROBE(mullo, "v_mul_lo_u32", "%8, %9")
PROBE(and_, "v_and_b32", "%8, %9")
PROBE(maxi, "v_max_i32", "%8, %9")
PROBE(med3i, "v_med3_i32", "%8, %9, %10")
PROBE(cndvcc, "v_cndmask_b32", "%8, %9, vcc")
PROBE(mov_dpp, "v_mov_b32_dpp", "%8 row_newbcast:3 row_mask:0xf bank_mask:0xf")
PROBE(mov_dpp_qp, "v_mov_b32_dpp", "%8 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf")
PROBE(mul_dpp, "v_mul_f32_dpp", "%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf")
PROBE(fmac_sgpr, "v_fmac_f32", "s20, %9")
PROBE(mul_lit, "v_mul_f32", "0x40400000, %9")
PROBE(fma_mix, "v_fma_f32", "%8, %9, 1.0")

__global__ void k_mov64_dpp(unsigned long long* out, float* sink, int iters) {   // v_mov_b64_dpp row_newbcast (DPALU form)
  double a0 = threadIdx.x, a1 = 1., a2 = 2., a3 = 3., x = 1.5 + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b64_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b64_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b64_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                 "v_mov_b64_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                 : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(x));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (t1 - t0) * 4;
  sink[threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ void k_cmp(unsigned long long* out, float* sink, int iters) {   // v_cmp writing an SGPR pair
  float x = 1.0001f + threadIdx.x, y = 0.5f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
    asm volatile("v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %0, %1\n v_cmp_lt_f32 s[24:25], %0, %1\n v_cmp_lt_f32 s[26:27], %0, %1\n"
                 "v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %0, %1\n v_cmp_lt_f32 s[24:25], %0, %1\n v_cmp_lt_f32 s[26:27], %0, %1\n"
                 "v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %0, %1\n v_cmp_lt_f32 s[24:25], %0, %1\n v_cmp_lt_f32 s[26:27], %0, %1\n"
                 "v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %0, %1\n v_cmp_lt_f32 s[24:25], %0, %1\n v_cmp_lt_f32 s[26:27], %0, %1\n"
                 :: "v"(x), "v"(y) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (t1 - t0) * 4;   // 16 per iteration instead of 64
  sink[threadIdx.x] = x;
}
__global__ void k_salu(unsigned long long* out, float* sink, int iters) {   // s_and_b64 / s_add / s_cselect mix
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
    asm volatile("s_and_b64 s[20:21], s[22:23], s[24:25]\n s_add_i32 s26, s27, s28\n s_and_b64 s[20:21], s[22:23], s[24:25]\n s_add_i32 s26, s27, s28\n"
                 "s_and_b64 s[20:21], s[22:23], s[24:25]\n s_add_i32 s26, s27, s28\n s_and_b64 s[20:21], s[22:23], s[24:25]\n s_add_i32 s26, s27, s28\n"
                 "s_and_b64 s[20:21], s[22:23], s[24:25]\n s_add_i32 s26, s27, s28\n s_and_b64 s[20:21], s[22:23], s[24:25]\n s_add_i32 s26, s27, s28\n"
                 "s_and_b64 s[20:21], s[22:23], s[24:25]\n s_add_i32 s26, s27, s28\n s_and_b64 s[20:21], s[22:23], s[24:25]\n s_add_i32 s26, s27, s28\n"
                 ::: "s20", "s21", "s26", "scc");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = (t1 - t0) * 4;
  sink[threadIdx.x] = 0.f;
}

typedef void (*kfn)(unsigned long long*, float*, int);
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned long long* out; float* sink;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&sink, 4096 * 4);
  struct { const char* name; kfn f; } ks[] = {
#define E(N) {#N, k_##N}
      E(fmac), E(fma3), E(mul), E(add), E(sub_abs), E(maxf), E(minf), E(med3f), E(floorf_), E(fract), E(cvt_i32), E(cvt_f32), E(rcp), E(exp), E(mov),
      E(addu), E(lshl_add), E(add3), E(mad24), E(mullo), E(and_), E(maxi), E(med3i), E(cndvcc), E(mov_dpp), E(mov_dpp_qp), E(mul_dpp),
      E(fmac_sgpr), E(mul_lit), E(fma_mix), E(mov64_dpp), E(cmp), E(salu)};
  const int iters = 200;
  printf("%-12s %s\n", "op", "clk per instr per SIMD  @2 waves/SIMD  @4 waves/SIMD   (1 wave/SIMD: clk per instr per wave)");
  for (auto& k : ks) {
    double r[3];
    int wv[3] = {4, 8, 16};
    for (int i = 0; i < 3; ++i) {
      hipLaunchKernelGGL(k.f, dim3(1), dim3(64 * wv[i]), 0, 0, out, sink, iters);
      hipLaunchKernelGGL(k.f, dim3(1), dim3(64 * wv[i]), 0, 0, out, sink, iters);
      (void)hipDeviceSynchronize();
      std::vector<unsigned long long> h(16);
      (void)hipMemcpy(h.data(), out, wv[i] * 8, hipMemcpyDeviceToHost);
      unsigned long long mx = 0;
      for (int j = 0; j < wv[i]; ++j) mx = std::max(mx, h[j]);
      r[i] = (double)mx / (iters * 64.0) / (wv[i] / 4.0);
    }
    printf("%-12s %22.2f %14.2f   (%.2f)\n", k.name, r[1], r[2], r[0]);
  }
  return 0;
}
