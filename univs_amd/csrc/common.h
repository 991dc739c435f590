// Shared helpers for libunivs_hip.so (gfx950 only; no CUDA/compat paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/univs_hip.h"

namespace univs {

void set_error(const char* fmt, ...);

// Levels are passed by value in the kernarg segment (scalar loads), never re-read per thread
// from global memory as the reference does (ms_deform_im2col_cuda.cuh:277-280).
struct LevelTable {
  int H[UNIVS_MAX_LEVELS];
  int W[UNIVS_MAX_LEVELS];
  int start[UNIVS_MAX_LEVELS];
};

// MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only, never correctness).  Remap so
// that each XCD works on one contiguous chunk of the logical block range and neighbouring logical
// blocks (which share operand rows) hit the same private L2.  Bijective for any nblocks.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblocks) {
  const unsigned nx = 8;
  const unsigned q = nblocks / nx, r = nblocks % nx;
  const unsigned xcd = bid % nx, idx = bid / nx;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// gfx950 (measured, tools/race_probe8.py + tools/probes/cohab_victim.hip; DESIGN.md section 3, "co-residency"): a packed-f32 instruction
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose LOW result selects the HIGH half of src1 or src2 (op_sel:[.,1] / op_sel:[.,.,1])
// reads 0 for that operand in lanes 48..63 while ANOTHER wave of the same SIMD issues v_mfma_f32_16x16x32_f16 / _bf16 -- a kernel of
// another stream or process is enough.  hipcc forms exactly that instruction from scalar code: a (scale, bias) pair loaded as one
// 8-byte value and applied to a float4, a horizontal sum of a register pair.  These single operations stay out of its packed
// selection (same fused multiply-add, same rounding); tests/test_isa_cpu.py scans the built library for the form.
__device__ __forceinline__ float fma_single(float a, float b, float c) {
  float d;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float fnma_single(float a, float b, float c) {   // fma(-a, b, c)
  float d;
  asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float sub_single(float a, float b) {
  float d;
  asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ float add_single(float a, float b) {
  float d;
  asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// hipGetLastError() is sticky per thread: an unrelated earlier failure (e.g. a device query made before
// the runtime was initialised) would otherwise be reported as OUR launch failing.  Entry points call
// this first.
inline void clear_sticky_error() { (void)hipGetLastError(); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return UNIVS_ERR_LAUNCH;
  }
  return UNIVS_OK;
}

}  // namespace univs
