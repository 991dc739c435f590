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
// for y0 == H - 1 the in-level row H - 1 is the TOP corner (its weight moves to wb, wt = 0).  An out-of-level
// column gets weight 0 and a clamped address.
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
  const bool inimg = him > -1.f && wim > -1.f && him < (float)H && wim < (float)W;
  const float hf = floorf(him), wf = floorf(wim);
  const float lh = him - hf, lw = wim - wf;
  // clamp in float first: the int conversion is then defined for any input (inf / NaN / huge)
  const int y0 = (int)fminf(fmaxf(hf, -2.f), (float)H), x0 = (int)fminf(fmaxf(wf, -2.f), (float)W);
  const float wtop = aw * (1.f - lh), wbot = aw * lh;       // rows y0, y0 + 1
  const int rb = min(max(y0, 0), H - 2);
  float wt = (y0 == rb) ? wtop : (y0 < rb ? wbot : 0.f);   // weight of row rb
  float wb = (y0 == rb) ? wbot : (y0 < rb ? 0.f : wtop);   // weight of row rb + 1
  const int c = x0 + side;
  const bool cvalid = (unsigned)c < (unsigned)W;
  const float f = side ? lw : 1.f - lw;
  const int cb = min(max(c, 0), W - 1);
  wt *= f;
  wb *= f;
  const int r0 = rb - wy0, c0 = cb - wx0;
  const bool inwin = (unsigned)r0 < (unsigned)(wh - 1) && (unsigned)c0 < (unsigned)ww;
  // selects, not multiplications by 0: a NaN / inf location or weight must contribute exactly nothing
  const bool live = inimg && qvalid && cvalid && aw != 0.f;
  const bool use = live && inwin;
  T3Record r;
  r.miss = live && !inwin;
  r.slot = use ? (r0 * ww + c0) * 128 : 0;
  r.wt = use ? wt : 0.f;
  r.wb = use ? wb : 0.f;
  return r;
}

}  // namespace univs
