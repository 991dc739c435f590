// Bilinear-footprint arithmetic shared by the MSDA forward kernels (gfx950).
#pragma once
#include "common.h"

namespace univs {

typedef float v2f __attribute__((ext_vector_type(2)));

// a += s * v on 4 channels as two v_pk_fma_f32 (2 FMAs per VALU issue slot; the kernels here are
// VALU-issue-bound, not bandwidth-bound, so this halves their dominant instruction class)
__device__ __forceinline__ float4 fma4(float s, float4 v, float4 a) {
  const v2f s2 = {s, s};
  v2f lo = {a.x, a.y}, hi = {a.z, a.w};
  lo = __builtin_elementwise_fma(s2, (v2f){v.x, v.y}, lo);
  hi = __builtin_elementwise_fma(s2, (v2f){v.z, v.w}, hi);
  return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// Branch-free footprint of one sample: corner pixel coordinates clamped into the level (so every
// corner is always loadable) and the four corner weights, already multiplied by the attention
// weight and set to exactly 0 for corners outside the level (reference bilinear helper,
// ms_deform_im2col_cuda.cuh:38-89) or samples outside the (-1, size) band (cuh:293).
// No divergent control flow => all gathers of a level can be in flight together.
struct Footprint {
  int h0, h1, w0, w1;       // clamped corner rows / columns
  float w00, w01, w10, w11; // weights of (h0,w0) (h0,w1) (h1,w0) (h1,w1)
};

__host__ __device__ __forceinline__ Footprint footprint(int H, int W, float x, float y, float aw) {
  Footprint f;
  const float him = y * (float)H - 0.5f, wim = x * (float)W - 0.5f;
  const bool inb = him > -1.f && wim > -1.f && him < (float)H && wim < (float)W;
  const float hf = floorf(him), wf = floorf(wim);
  const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
  // clamp in float first so that the int conversion is defined for any input (inf / NaN / huge)
  const int h0 = (int)fminf(fmaxf(hf, -2.f), (float)H), w0 = (int)fminf(fmaxf(wf, -2.f), (float)W);
  const bool t = h0 >= 0, b = h0 + 1 <= H - 1, lft = w0 >= 0, rgt = w0 + 1 <= W - 1;
  // selects, not multiplications by 0: a NaN/inf location must contribute exactly nothing
  f.w00 = (inb && t && lft) ? aw * hh * hw : 0.f;
  f.w01 = (inb && t && rgt) ? aw * hh * lw : 0.f;
  f.w10 = (inb && b && lft) ? aw * lh * hw : 0.f;
  f.w11 = (inb && b && rgt) ? aw * lh * lw : 0.f;
  f.h0 = min(max(h0, 0), H - 1);
  f.h1 = min(max(h0 + 1, 0), H - 1);
  f.w0 = min(max(w0, 0), W - 1);
  f.w1 = min(max(w0 + 1, 0), W - 1);
  return f;
}

}  // namespace univs
