// Sample record of the third-generation LDS-tiled MSDA kernel (msda_tiled3.hip), host + device so that
// tools/t3_emulate.cpp can run exactly this arithmetic on the CPU against a plain bilinear reference.
//
// One record = one corner COLUMN (side 0: left pixel, side 1: right pixel) of one sample at one level:
//   slot  byte offset of (row rb, column cb) inside the staged window (row-major, 128 B per pixel-head)
//   wt    weight of (rb, cb), wb weight of (rb + 1, cb); both already include the attention weight
//   miss  the footprint is live but not inside the staged window -> the caller adds it from global memory
// Reference arithmetic: ms_deform_im2col_cuda.cuh:285-293 (h_im / w_im, the (-1, H) x (-1, W) band) and :38-89
// (corner validity).  Windows are clipped to the level, so the two staged rows are (rb, rb + 1) with
// rb = clamp(y0, 0, H - 2): for y0 == -1 the in-level row 0 is the BOTTOM corner (its weight moves to wt, wb = 0),
// for y0 == H - 1 the in-level row H - 1 is the TOP corner (its weight moves to wb, wt = 0); for any other y0 outside
// [0, H - 2] both rows are out of the level and both weights are exactly 0.  An out-of-level column gets weight 0
// and a clamped address.  The reference's band test (h_im > -1 && w_im > -1 && h_im < H && w_im < W) is implied:
// outside the band every corner is out of the level, on its edges (h_im == -1, ...) the only in-level corner has
// weight lh == 0 -- so no separate test is needed, and no boolean chains (each `&&` of two lane masks is an
// instruction on the CU's single scalar pipe, which is what the first version of the kernel was bound by).
// Weights are SELECTED, never multiplied by 0: a NaN / inf location must contribute exactly nothing.
#pragma once

namespace univs {

struct T3Record {
  int slot;
  float wt, wb;
  bool miss;
};

__host__ __device__ inline T3Record t3_record(float x, float y, float aw, int side, bool qvalid, int H, int W, int wx0,
                                              int wy0, int ww, int wh) {
  const float him = y * (float)H - 0.5f, wim = x * (float)W - 0.5f;
  const float hf = floorf(him), wf = floorf(wim);
  const float lh = him - hf, lw = wim - wf;
  // clamp in float first: the int conversion is then defined for any input (inf / NaN / huge -> -2: all corners out)
  const int y0 = (int)fminf(fmaxf(hf, -2.f), (float)H), x0 = (int)fminf(fmaxf(wf, -2.f), (float)W);
  const float awv = qvalid ? aw : 0.f;                        // lanes without a query: weight 0, never a miss
  const float wtop = awv * (1.f - lh), wbot = awv * lh;      // rows y0, y0 + 1
  const int rb = min(max(y0, 0), H - 2);
  const int dy = y0 - rb;                                    // 0: both rows staged as they are; -1 / +1: see above
  float wt = dy == 0 ? wtop : (dy == -1 ? wbot : 0.f);      // weight of row rb
  float wb = dy == 0 ? wbot : (dy == 1 ? wtop : 0.f);       // weight of row rb + 1
  const int c = x0 + side;
  const int cb = min(max(c, 0), W - 1);
  const float f = side ? lw : 1.f - lw;
  wt = c == cb ? wt * f : 0.f;                              // out-of-level column: nothing
  wb = c == cb ? wb * f : 0.f;
  const unsigned r0 = (unsigned)(rb - wy0), c0 = (unsigned)(cb - wx0);
  const bool inwin = r0 < (unsigned)(wh - 1) && c0 < (unsigned)ww;
  T3Record r;
  r.miss = !inwin && (fabsf(wt) + fabsf(wb)) != 0.f;       // (NaN weights count as live)
  r.slot = inwin ? (int)(r0 * (unsigned)ww + c0) * 128 : 0;
  r.wt = inwin ? wt : 0.f;
  r.wb = inwin ? wb : 0.f;
  return r;
}

}  // namespace univs
