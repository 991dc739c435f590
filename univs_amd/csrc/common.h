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
