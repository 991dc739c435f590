// Shared by the three-product fp16 GEMM kernels (linear_f16x3.hip: W resident in LDS; gemm_f16x3_stream.hip: W pre-split in
// global memory and streamed through LDS -- wide-K Linears and the 3 x 3 convolution).
#pragma once
#include "common.h"

#include <type_traits>

namespace univs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// power of two that brings a row whose largest magnitude has the bit pattern `maxbits` into [2^t, 2^(t+1)), and its inverse.
// Zero rows and rows below 2^-100 keep a finite scale (2^(t+100)); an infinite maximum gives 2^(t-128).
__device__ __forceinline__ void l3_scale(unsigned maxbits, int t, float& s, float& inv) {
  int e = (int)((maxbits >> 23) & 255u) - 127;            // 2^e <= max < 2^(e+1)
  e = max(-100, min(e, 128));
  s = __builtin_bit_cast(float, (unsigned)(127 + t - e) << 23);
  inv = __builtin_bit_cast(float, (unsigned)(127 - t + e) << 23);
}

// Timing experiments only (instrumented builds, `python -m univs_amd.build --ablate nosplit|nomfma|nosplit_nomfma`; results WRONG):
//   UNIVS_ABLATE_NOSPLIT -- the x operand is taken as if it arrived pre-split (its 32 bytes per lane reinterpreted as the two fp16x8
//   parts, no row maximum, no conversions): what a consumer of producer-side (hi, lo) activations would execute;
//   UNIVS_ABLATE_NOMFMA  -- the matrix instructions are dropped (operands kept alive).
#ifdef UNIVS_ABLATE_NOMFMA
__device__ __forceinline__ f32x4 l3_fake_mfma(f16x8 a, f16x8 b, f32x4 c) {
  asm volatile("" : "+v"(c) : "v"(a), "v"(b));
  return c;
}
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) l3_fake_mfma(a, b, c)
#endif

// UNIVS_TRACE_GEMM (instrumented build `--ablate trace`): s_memtime stamps of a few (workgroup, wave) pairs at the phase boundaries of
// linear_f16x3 / gemm_f16x3_stream, read back through univs_debug_gemm_trace_{linear,stream} (tools/gemm_trace.py).
#ifdef UNIVS_TRACE_GEMM
#define UNIVS_GT_SLOTS 48
#define UNIVS_GT_STAMPS 64
#define UNIVS_GT_DECL(sym) static __device__ unsigned long long sym[UNIVS_GT_SLOTS * UNIVS_GT_STAMPS]
// slot: workgroups x = {0, mid, last} of passes y = {0, last}, all 8 waves; -1 elsewhere
#define UNIVS_GT_SLOT()                                                                                                     \
  ((((int)blockIdx.x == 0 || (int)blockIdx.x == (int)gridDim.x / 2 || (int)blockIdx.x == (int)gridDim.x - 1) &&                \
    ((int)blockIdx.y == 0 || (int)blockIdx.y == (int)gridDim.y - 1))                                                          \
       ? ((((int)blockIdx.x == 0 ? 0 : (int)blockIdx.x == (int)gridDim.x - 1 ? 2 : 1) * 2 + ((int)blockIdx.y == 0 ? 0 : 1)) * 8 + \
          (int)(threadIdx.x >> 6))                                                                                            \
       : -1)
#define UNIVS_GT(sym, slot, i)                                                                                  \
  do {                                                                                                          \
    if ((slot) >= 0 && (threadIdx.x & 63) == 0 && (i) < UNIVS_GT_STAMPS) sym[(slot) * UNIVS_GT_STAMPS + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define UNIVS_GT_REAL(sym, slot, i)                                                                             \
  do {                                                                                                          \
    if ((slot) >= 0 && (threadIdx.x & 63) == 0) sym[(slot) * UNIVS_GT_STAMPS + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define UNIVS_GT_VAL(sym, slot, i, v)                                                                           \
  do {                                                                                                          \
    if ((slot) >= 0 && (threadIdx.x & 63) == 0) sym[(slot) * UNIVS_GT_STAMPS + (i)] = (unsigned long long)(v);  \
  } while (0)
#else
#define UNIVS_GT_SLOT() (-1)
#define UNIVS_GT(sym, slot, i) do { } while (0)
#define UNIVS_GT_REAL(sym, slot, i) do { } while (0)
#define UNIVS_GT_VAL(sym, slot, i, v) do { } while (0)
#endif

// f(integral_constant<int, I>) for I = I0 .. N - 1: loops whose index has to be a compile-time constant (immediate offsets of asm
// statements; `#pragma unroll` is a hint and its index is not a constant expression)
template <int I, int N, class F>
__device__ __forceinline__ void l3_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    l3_static_for<I + 1, N>(f);
  }
}

// 8 consecutive k of one row, scaled -> the two fp16x8 parts (round to nearest even)
__device__ __forceinline__ void l3_split8(f32x4 v0, f32x4 v1, float s, f16x8& h, f16x8& m) {
#ifdef UNIVS_ABLATE_NOSPLIT
  h = __builtin_bit_cast(f16x8, v0);
  m = __builtin_bit_cast(f16x8, v1);
  return;
#endif
#ifdef UNIVS_SPLIT_PLAIN   // (A / B: the plain expression, ~3.4 vector instructions per value as hipcc compiles it; same values)
  {
    const float xp[8] = {v0.x * s, v0.y * s, v0.z * s, v0.w * s, v1.x * s, v1.y * s, v1.z * s, v1.w * s};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const _Float16 hh = (_Float16)xp[e];
      h[e] = hh;
      m[e] = (_Float16)(xp[e] - (float)hh);                  // exact difference, then rounded
    }
    return;
  }
#endif
  // Two instructions per value, written into the packed halves directly: h = fp16(v s) and m = fp16(v s - h) as mixed-precision FMAs
  // (v_fma_mix{lo,hi}_f16: fp32 sources v and s, the third source 0 or -h read as fp16 from the half just written; one rounding each).
  // The same values as the mul / cvt / cvt-back / sub / cvt chain hipcc makes of the plain expression -- s is a power of two, so v s
  // is exact, and v s - fp16(v s) is exact in fp32 -- at 2 instead of ~3.4 vector instructions per value: the split is a fifth of the
  // vector work of every three-product kernel, and vector work does not overlap the matrix pipe (profiles/r05_gemm_negative_results_v2.txt).
  const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
  u32x4 hu, mu;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned hp, mp;
    asm("v_fma_mixlo_f16 %0, %2, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0 op_sel_hi:[0,0,0]\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hp), "=&v"(mp)
        : "v"(x[2 * p]), "v"(x[2 * p + 1]), "v"(s));
    hu[p] = hp;
    mu[p] = mp;
  }
  h = __builtin_bit_cast(f16x8, hu);
  m = __builtin_bit_cast(f16x8, mu);
}

__device__ __forceinline__ unsigned l3_absmax8(f32x4 v0, f32x4 v1) {
#ifdef UNIVS_ABLATE_NOSPLIT
  return 0x3f800000u;
#endif
#ifdef UNIVS_SPLIT_PLAIN
  {
    const float a = fmaxf(fmaxf(fabsf(v0.x), fabsf(v0.y)), fmaxf(fabsf(v0.z), fabsf(v0.w)));
    const float b = fmaxf(fmaxf(fabsf(v1.x), fabsf(v1.y)), fmaxf(fabsf(v1.z), fabsf(v1.w)));
    return __builtin_bit_cast(unsigned, fmaxf(a, b));
  }
#endif
  // max |.| of eight values in four instructions (v_max3_f32 with |.| source modifiers)
  float t1, t2, r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t1) : "v"(v0.x), "v"(v0.y), "v"(v0.z));
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(t2) : "v"(v0.w), "v"(v1.x), "v"(v1.y));
  asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(v1.z), "v"(v1.w), "v"(t1));
  return __builtin_bit_cast(unsigned, fmaxf(r, t2));       // non-negative floats order like their bit patterns
}

// nn.GELU() (approximate = 'none'): x * 0.5 * (1 + erf(x / sqrt 2)) with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7;
// in fp32 the GELU differs from the exact value by <= 4.7e-7 absolute on [-12, 12], ATen's own fp32 GELU by 1.2e-6): one
// reciprocal, one exp2 and eight multiply-adds, branch-free -- the library's erff (three ranges under lane divergence) was
// a third of the fc1 + GELU kernels' time.
__device__ __forceinline__ float l3_gelu(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
  const float erf_abs = fmaf(-p, e, 1.0f);
  const float erf = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, erf_abs) | (__builtin_bit_cast(unsigned, x) & 0x80000000u));
  return 0.5f * x * (1.0f + erf);
}

// maximum over the four lanes (k-groups, lane >> 4) that hold one row of the B operand
__device__ __forceinline__ unsigned l3_row_max(unsigned v) {
#ifdef UNIVS_ABLATE_NOSPLIT
  return v;
#endif
  const auto s1 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  const unsigned r1 = max(s1[0], s1[1]);
  const auto s2 = __builtin_amdgcn_permlane16_swap(r1, r1, false, false);
  return max(s2[0], s2[1]);
}

}  // namespace univs
