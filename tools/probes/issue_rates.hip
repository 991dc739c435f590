// Issue-rate probe for gfx950 (run on the GPU box):  hipcc --offload-arch=gfx950 -O2 tools/probes/issue_rates.hip -o /tmp/issue_rates && /tmp/issue_rates
// Each kernel runs N repetitions of a block of 64 independent instructions of one kind in every wave and reports
// cycles (s_memtime) per wave-instruction for 1, 2, 4 waves per SIMD on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int KIND>
__global__ void probe(unsigned long long* out, float* sink, int iters) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)i;
  __syncthreads();
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  float w = 1.0001f, d = 0.5f;
  int ia = threadIdx.x, ib = 3;
  unsigned addr = (threadIdx.x & 63) * 8;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f}, pw = {1.0001f, 1.0002f}, pd = {0.5f, 0.25f};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {   // v_fmac_f32 (8 independent accumulators)
      asm volatile(REP4(REP4("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(d));
    } else if (KIND == 1) {   // v_pk_fma_f32
      asm volatile(REP4(REP4("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pw), "v"(pd));
    } else if (KIND == 2) {   // v_fmac_f32_dpp row_newbcast
      asm volatile(REP4(REP4("v_fmac_f32_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                             " v_fmac_f32_dpp %2, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(d));
    } else if (KIND == 3) {   // v_add_u32_dpp
      asm volatile(REP4(REP4("v_add_u32_dpp %0, %4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %5 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             " v_add_u32_dpp %2, %4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %5 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"))
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(ia), "v"(ib));
    } else if (KIND == 4) {   // s_and_b64 chain-free (SALU)
      asm volatile(REP64("s_and_b64 s[20:21], s[22:23], s[24:25]\n") ::: "s20", "s21");
    } else if (KIND == 5) {   // ds_read_b64, 16 in flight
      asm volatile(REP4("ds_read_b64 v[20:21], %0\n ds_read_b64 v[22:23], %0 offset:512\n ds_read_b64 v[24:25], %0 offset:1024\n ds_read_b64 v[26:27], %0 offset:1536\n"
                        "ds_read_b64 v[28:29], %0 offset:2048\n ds_read_b64 v[30:31], %0 offset:2560\n ds_read_b64 v[32:33], %0 offset:3072\n ds_read_b64 v[34:35], %0 offset:3584\n"
                        "ds_read_b64 v[36:37], %0 offset:4096\n ds_read_b64 v[38:39], %0 offset:4608\n ds_read_b64 v[40:41], %0 offset:5120\n ds_read_b64 v[42:43], %0 offset:5632\n"
                        "ds_read_b64 v[44:45], %0 offset:6144\n ds_read_b64 v[46:47], %0 offset:6656\n ds_read_b64 v[48:49], %0 offset:7168\n ds_read_b64 v[50:51], %0 offset:7680\n"
                        "s_waitcnt lgkmcnt(0)\n")
                   :: "v"(addr) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51");
    } else if (KIND == 6) {   // v_cndmask_b32
      asm volatile(REP4(REP4("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n"))
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(w), "v"(d) : "vcc");
    } else if (KIND == 8) {   // v_cndmask_b32 e64, SGPR-pair mask, 8 distinct destinations
      asm volatile(REP4(REP4("v_cndmask_b32 %0, %8, %9, s[20:21]\n v_cndmask_b32 %1, %8, %9, s[20:21]\n v_cndmask_b32 %2, %8, %9, s[22:23]\n v_cndmask_b32 %3, %8, %9, s[22:23]\n"))
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(w), "v"(d) : "s20", "s21", "s22", "s23");
    } else if (KIND == 9) {   // v_mov_b32_dpp
      asm volatile(REP4(REP4("v_mov_b32_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             " v_mov_b32_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"))
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(w));
    } else if (KIND == 10) {   // v_fmac_f32 with an SGPR multiplicand
      asm volatile(REP4(REP4("v_fmac_f32 %0, s20, %8\n v_fmac_f32 %1, s21, %8\n v_fmac_f32 %2, s22, %8\n v_fmac_f32 %3, s23, %8\n"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));
    } else if (KIND == 11) {   // v_min_i32 (plain int VALU)
      asm volatile(REP4(REP4("v_min_i32 %0, %4, %5\n v_min_i32 %1, %4, %5\n v_min_i32 %2, %4, %5\n v_min_i32 %3, %4, %5\n"))
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(ia), "v"(ib));
    } else if (KIND == 12) {   // 1 DPP : 2 plain interleave (v_mul_f32_dpp, v_fmac, v_fmac)
      asm volatile(REP4(REP4("v_mul_f32_dpp %0, %8, |%9| row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"))
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(d));
    } else if (KIND == 13) {   // v_readlane_b32
      asm volatile(REP4(REP4("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 5\n v_readlane_b32 s22, %0, 7\n v_readlane_b32 s23, %0, 9\n"))
                   :: "v"(w) : "s20", "s21", "s22", "s23");
    } else if (KIND == 7) {   // ds_read_b128, 16 in flight
      asm volatile(REP4("ds_read_b128 v[20:23], %0\n ds_read_b128 v[24:27], %0 offset:1024\n ds_read_b128 v[28:31], %0 offset:2048\n ds_read_b128 v[32:35], %0 offset:3072\n"
                        "ds_read_b128 v[36:39], %0 offset:4096\n ds_read_b128 v[40:43], %0 offset:5120\n ds_read_b128 v[44:47], %0 offset:6144\n ds_read_b128 v[48:51], %0 offset:7168\n"
                        "ds_read_b128 v[20:23], %0 offset:8192\n ds_read_b128 v[24:27], %0 offset:9216\n ds_read_b128 v[28:31], %0 offset:10240\n ds_read_b128 v[32:35], %0 offset:11264\n"
                        "ds_read_b128 v[36:39], %0 offset:12288\n ds_read_b128 v[40:43], %0 offset:13312\n ds_read_b128 v[44:47], %0 offset:14336\n ds_read_b128 v[48:51], %0 offset:15360\n"
                        "s_waitcnt lgkmcnt(0)\n")
                   :: "v"(addr * 2) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 64 + (threadIdx.x >> 6)] = t1 - t0;
  sink[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + lds[threadIdx.x];
}

template <int KIND>
void run(const char* name) {
  unsigned long long* out;
  float* sink;
  (void)hipMalloc(&out, 64 * 64 * 8);
  (void)hipMalloc(&sink, 4096 * 4);
  const int iters = 200;
  printf("%-28s", name);
  for (int waves : {4, 8, 16}) {
    hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64 * waves), 0, 0, out, sink, iters);
    hipLaunchKernelGGL(probe<KIND>, dim3(1), dim3(64 * waves), 0, 0, out, sink, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(64);
    (void)hipMemcpy(h.data(), out, 64 * 8, hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int i = 0; i < waves; ++i) mx = std::max(mx, h[i]);
    // s_memtime counts at a constant 100 MHz-ish reference on some parts; report raw ticks per wave-instruction and
    // per-CU instruction throughput relative to it
    printf("  %2d waves: %.3f ticks/instr/wave (%.3f ticks per CU-instr)", waves, (double)mx / (iters * 64.0), (double)mx / (iters * 64.0 * waves));
  }
  printf("\n");
  (void)hipFree(out); (void)hipFree(sink);
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  if (only < 0 || only == 0) run<0>("v_fmac_f32");
  if (only < 0 || only == 1) run<1>("v_pk_fma_f32");
  if (only < 0 || only == 2) run<2>("v_fmac_f32_dpp newbcast");
  if (only < 0 || only == 3) run<3>("v_add_u32_dpp newbcast");
  if (only < 0 || only == 6) run<6>("v_cndmask_b32 (vcc, same dst)");
  if (only < 0 || only == 8) run<8>("v_cndmask_b32 e64 sgpr mask");
  if (only < 0 || only == 9) run<9>("v_mov_b32_dpp newbcast");
  if (only < 0 || only == 10) run<10>("v_fmac_f32 sgpr operand");
  if (only < 0 || only == 11) run<11>("v_min_i32");
  if (only < 0 || only == 12) run<12>("1 mul_dpp : 3 fmac");
  if (only < 0 || only == 13) run<13>("v_readlane_b32");
  if (only < 0 || only == 4) run<4>("s_and_b64");
  if (only < 0 || only == 5) run<5>("ds_read_b64 (16 in flight)");
  if (only < 0 || only == 7) run<7>("ds_read_b128 (16 in flight)");
  // clock reference: v_fmac at 4 waves = one wave per SIMD, back-to-back dependent-free issue
  return 0;
}
