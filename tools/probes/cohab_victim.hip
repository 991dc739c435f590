// Victim kernels of the co-residency probe (tools/race_probe8.py): y[r][c] = x[r][c] * aff[r][0] + aff[r][1], written five ways, to find
// out WHAT goes wrong in transpose_f32_kernel's affine path when gemm_f16x3_tile runs beside it on another stream (the low half of
// `v_pk_fma_f32 v[2:3], v[2:3], v[8:9], v[8:9] op_sel:[0,0,1] op_sel_hi:[1,0,1]` comes out as x * scale, without the bias, in lanes
// 48..63: the register that receives the bias held 0 before the load).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/cohab_victim.hip -o /tmp/libcohab.so
#include <hip/hip_runtime.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));

// 256 threads: 16 rows x 16 lanes, 4 columns per lane (the load phase of transpose_f32_kernel: 16 lanes share a row's affine pair)
template <int VAR>
__global__ __launch_bounds__(256) void affine_rows(const float* __restrict__ x, const float* __restrict__ aff, float* __restrict__ y, int R,
                                                   int C) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int r = blockIdx.y * 16 + ty, c = blockIdx.x * 64 + 4 * tx;
  if (r >= R || c >= C) return;
  const float* px = x + (long long)r * C + c;
  const float* pa = aff + (long long)r * 2;
  float* py = y + (long long)r * C + c;
  if (VAR == 0) {                                      // what hipcc makes of the C++ (the shipped kernel's expression)
    float4 v = *reinterpret_cast<const float4*>(px);
    const float sc = pa[0], bi = pa[1];
    v = make_float4(fmaf(v.x, sc, bi), fmaf(v.y, sc, bi), fmaf(v.z, sc, bi), fmaf(v.w, sc, bi));
    *reinterpret_cast<float4*>(py) = v;
    return;
  }
  f32x2 a = {0.f, 0.f}, b = {0.f, 0.f};
  if (VAR == 1 || VAR == 3 || VAR == 4) {
    // the affine pair lands in a register pair that held (7, 1) before: a stale HIGH register shows up as x * scale + 1
    f32x2 ab = {7.0f, 1.0f};
    if (VAR == 1)
      asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %4, off\n\t"
                   "s_waitcnt vmcnt(0)\n\t"
                   "v_pk_fma_f32 %0, %0, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                   "v_pk_fma_f32 %1, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                   : "=&v"(a), "=&v"(b), "+v"(ab) : "v"(px), "v"(pa) : "memory");
    if (VAR == 3)                                      // the same with idle cycles between the wait and the first use
      asm volatile("global_load_dwordx2 %0, %3, off\n\tglobal_load_dwordx2 %1, %3, off offset:8\n\tglobal_load_dwordx2 %2, %4, off\n\t"
                   "s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\t"
                   "v_pk_fma_f32 %0, %0, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
                   "v_pk_fma_f32 %1, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]"
                   : "=&v"(a), "=&v"(b), "+v"(ab) : "v"(px), "v"(pa) : "memory");
    if (VAR == 4) {                                    // no cross-half operand selection: (scale, scale) and (bias, bias) built by moves first
      f32x2 s2, b2;
      asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %5, off offset:8\n\tglobal_load_dwordx2 %2, %6, off\n\t"
                   "s_waitcnt vmcnt(0)\n\t"
                   "v_pk_mul_f32 %3, %2, 1.0 op_sel:[0,0] op_sel_hi:[0,0]\n\t"
                   "v_pk_mul_f32 %4, %2, 1.0 op_sel:[1,0] op_sel_hi:[1,0]\n\t"
                   "v_pk_fma_f32 %0, %0, %3, %4\n\t"
                   "v_pk_fma_f32 %1, %1, %3, %4"
                   : "=&v"(a), "=&v"(b), "+v"(ab), "=&v"(s2), "=&v"(b2) : "v"(px), "v"(pa) : "memory");
    }
  }
  if (VAR == 2) {                                      // scalar-per-lane FMAs, scale and bias loaded separately into registers that held 7 and 1
    float sc = 7.0f, bi = 1.0f;
    float v0, v1, v2, v3;
    asm volatile("global_load_dword %0, %6, off\n\tglobal_load_dword %1, %6, off offset:4\n\tglobal_load_dword %2, %6, off offset:8\n\t"
                 "global_load_dword %3, %6, off offset:12\n\tglobal_load_dword %4, %7, off\n\tglobal_load_dword %5, %7, off offset:4\n\t"
                 "s_waitcnt vmcnt(0)\n\t"
                 "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "+v"(sc), "+v"(bi) : "v"(px), "v"(pa) : "memory");
    a = (f32x2){v0, v1};
    b = (f32x2){v2, v3};
  }
  *reinterpret_cast<float4*>(py) = make_float4(a.x, a.y, b.x, b.y);
}

// an LDS squatter: `lds_bytes` of dynamic LDS per workgroup, spins for `spin` clocks -- pushes other kernels' LDS allocations up the CU's 160 KB
__global__ __launch_bounds__(64) void lds_squatter(int spin, float* sink) {
  extern __shared__ float sq[];
  sq[threadIdx.x] = (float)threadIdx.x;
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (sq[threadIdx.x] < 0.f) *sink = 1.f;
}


// Single-instruction aggressors: 8 waves per workgroup, each repeating ONE kind of instruction `iters` x 16 times on live registers.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void one_instruction(int iters, float* sink) {
  __shared__ float lds[512 * 4];
  float a = (float)threadIdx.x, b = 1.5f, c = 0.25f, d = 3.0f;
  f32x2 p = {a, b}, q = {c, d};
  f16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(a + i); hb[i] = (_Float16)(b - i); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  unsigned u = threadIdx.x * 2654435761u, w = blockIdx.x + 17u;
  lds[threadIdx.x] = a;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(w));
      if (KIND == 1) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(w));
      if (KIND == 2) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(u) : "v"(w));
      if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[0,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %3 op_sel_hi:[0,0,0]" : "+v"(u) : "v"(a), "v"(b), "v"(c));
      if (KIND == 4) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);
      if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p) : "v"(q));
      if (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel_hi:[1,1,0]" : "+v"(p) : "v"(q));
      if (KIND == 7) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(u) : "v"(w));
      if (KIND == 8) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(acc) : "v"((threadIdx.x & 255) * 16) : "memory");
      if (KIND == 9) asm volatile("v_pk_fma_f32 %0, %0, %1, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "+v"(p) : "v"(q));
      if (KIND == 10) asm volatile("v_cmp_lt_i32_sdwa vcc, %0, %1 src0_sel:BYTE_0 src1_sel:DWORD" : : "v"(u), "v"(w) : "vcc");
      if (KIND == 11) asm volatile("v_pk_mul_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(p) : "v"(q));
    }
  }
  if (u + w + (unsigned)p.x + (unsigned)p.y + (unsigned)acc[0] + (unsigned)acc[3] == 0x12345u) *sink = 1.f;
}

extern "C" int cohab_one_instruction(int kind, int workgroups, int iters, float* sink, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 g(workgroups), b(512);
  switch (kind) {
#define K_(k) case k: hipLaunchKernelGGL(one_instruction<k>, g, b, 0, st, iters, sink); break
    K_(0); K_(1); K_(2); K_(3); K_(4); K_(5); K_(6); K_(7); K_(8); K_(9); K_(10); K_(11);
#undef K_
    default: return -1;
  }
  return (int)hipGetLastError();
}


// Packed-f32 victims by operand position: which cross-half selections lose their operand beside MFMA waves?  a = (x[2 i], x[2 i + 1]),
// b = the row's (scale, bias) pair; the expected values are formed on the host (tools/race_probe8.py: PK_FORMS).
template <int FORM>
__global__ __launch_bounds__(256) void pk_forms(const float* __restrict__ x, const float* __restrict__ aff, float* __restrict__ y, int R, int C) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int r = blockIdx.y * 16 + ty, c = blockIdx.x * 32 + 2 * tx;
  if (r >= R || c >= C) return;
  f32x2 a = *reinterpret_cast<const f32x2*>(x + (long long)r * C + c);
  const f32x2 b = *reinterpret_cast<const f32x2*>(aff + (long long)r * 2);
  f32x2 d = {0.f, 0.f};
  if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));       // (a0 b1, a1 b1)
  if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));       // (a1 b0, a1 b1)
  if (FORM == 2) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(b));       // (a0 + b1, a1 + b1)
  if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b));   // (a0 b1 + b0, a1 b1 + b0)
  if (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[1,0,0] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b));   // (a1 b0 + b0, a1 b0 + b1)
  if (FORM == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(b));   // (a0 b0 + b1, a1 b0 + b1): the known one
  if (FORM == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %2 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a), "v"(b));   // (a0 b0 + b0, a1 b1 + b0): high result <- low half
  if (FORM == 7) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));           // (a0 b0, a1 b0): high result <- low half
  *reinterpret_cast<f32x2*>(y + (long long)r * C + c) = d;
}

extern "C" int cohab_pk_form(int form, const float* x, const float* aff, float* y, int R, int C, void* stream) {
  dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 15) / 16)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (form) {
#define F_(k) case k: hipLaunchKernelGGL(pk_forms<k>, grid, block, 0, st, x, aff, y, R, C); break
    F_(0); F_(1); F_(2); F_(3); F_(4); F_(5); F_(6); F_(7);
#undef F_
    default: return -1;
  }
  return (int)hipGetLastError();
}

// MFMA aggressors by instruction kind (dependent chains, 8 waves per workgroup)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(512) void mfma_kinds(int iters, float* sink) {
  f16x8 ha, hb;
  bf16x8 ba, bb;
  f16x4 qa, qb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x + i); hb[i] = (_Float16)(1.5f - i); ba[i] = (__bf16)(float)(threadIdx.x + i); bb[i] = (__bf16)(1.5f - i); }
  for (int i = 0; i < 4; ++i) { qa[i] = ha[i]; qb[i] = hb[i]; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {1.f, 1.f, 1.f, 1.f};
  f32x16 big = {};
  float fa = (float)threadIdx.x, fb = 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (KIND == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0);                         // dependent chain (tools default)
      if (KIND == 1) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0); acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hb, ha, acc2, 0, 0, 0); }   // two independent chains
      if (KIND == 2) big = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, big, 0, 0, 0);
      if (KIND == 3) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc, 0, 0, 0);
      if (KIND == 4) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(qa, qb, acc, 0, 0, 0);
      if (KIND == 5) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc, 0, 0, 0);
      if (KIND == 6) { acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc, 0, 0, 0); asm volatile("s_nop 7\n\ts_nop 7"); }      // sparse issue
    }
  }
  if (acc[0] + acc2[1] + big[3] == 12345.f) *sink = 1.f;
}

extern "C" int cohab_mfma_kind(int kind, int workgroups, int iters, float* sink, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  dim3 g(workgroups), b(512);
  switch (kind) {
#define M_(k) case k: hipLaunchKernelGGL(mfma_kinds<k>, g, b, 0, st, iters, sink); break
    M_(0); M_(1); M_(2); M_(3); M_(4); M_(5); M_(6);
#undef M_
    default: return -1;
  }
  return (int)hipGetLastError();
}

extern "C" int cohab_affine(int var, const float* x, const float* aff, float* y, int R, int C, void* stream) {
  dim3 grid((unsigned)((C + 63) / 64), (unsigned)((R + 15) / 16)), block(256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  switch (var) {
    case 0: hipLaunchKernelGGL(affine_rows<0>, grid, block, 0, st, x, aff, y, R, C); break;
    case 1: hipLaunchKernelGGL(affine_rows<1>, grid, block, 0, st, x, aff, y, R, C); break;
    case 2: hipLaunchKernelGGL(affine_rows<2>, grid, block, 0, st, x, aff, y, R, C); break;
    case 3: hipLaunchKernelGGL(affine_rows<3>, grid, block, 0, st, x, aff, y, R, C); break;
    case 4: hipLaunchKernelGGL(affine_rows<4>, grid, block, 0, st, x, aff, y, R, C); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

extern "C" int cohab_squat(int workgroups, int lds_bytes, int spin, float* sink, void* stream) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_squatter), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipLaunchKernelGGL(lds_squatter, dim3(workgroups), dim3(64), lds_bytes, static_cast<hipStream_t>(stream), spin, sink);
  return (int)hipGetLastError();
}
