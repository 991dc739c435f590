// Sample record of the third-generation LDS-tiled MSDA kernel (msda_tiled3.hip), host + device so that
// tools/t3_emulate.cpp can run exactly this arithmetic on the CPU against a plain bilinear reference.
//
// One record = one corner COLUMN (side 0: left pixel, side 1: right pixel) of one sample at one level:
//   slot  byte offset of (row y0, column x0 + side) inside the staged window (row-major, `pitch` pixels per row,
//         128 B per pixel-head); the bottom corner is one row below
//   wt    weight of the top corner, wb of the bottom corner; both already include the attention weight
//   miss  the sample is inside the reference's band but this column is not inside the staged window -> the caller
//         adds it from global memory
// Reference arithmetic: ms_deform_im2col_cuda.cuh:285-293 (h_im / w_im and the band test h_im > -1 && w_im > -1 &&
// h_im < H && w_im < W) and :38-89 (corner validity).  The staged window may include the one-pixel ring around the level
// (rows -1 and H, columns -1 and W), which the fill waves stage as zeros: an out-of-level corner then simply reads 0,
// exactly what the reference's corner tests amount to, and the record needs no border cases at all.
//
// Written for the issue costs of gfx950 (profiles/r02_gfx950_issue_costs.txt): fp32 mul / add / fma and integer add /
// shift issue at full rate, everything else (floor, cvt, compares, selects, min / max) at half rate -- so the band test
// is two |x - centre| < radius compares (NaN fails them, as it fails the reference's), the corner weights are three
// multiplies, and the only selects are the three that zero a record that must not contribute.
#pragma once

namespace univs {

struct T3Record {
  int slot;
  float wt, wb;
  bool miss;
};

// side_sign / side_one: (-1, 1) for the left column (weight 1 - lw), (+1, 0) for the right column (weight lw).
// WRAP (msda_tiled4.hip): the window is a row-circular buffer of `nr` rows -- window row r lives in LDS row
// (r + rot) mod nr, and LDS row `nr` duplicates LDS row 0 so that the bottom corner is always one row below the top one.
template <bool WRAP = false>
__host__ __device__ inline T3Record t3_record(float x, float y, float aw, int side, float side_sign, float side_one,
                                              bool qvalid, int H, int W, int wx0, int wy0, int ww, int wh, int pitch,
                                              int rot = 0, int nr = 0) {
  const float Hf = (float)H, Wf = (float)W;
  const float him = y * Hf - 0.5f, wim = x * Wf - 0.5f;
  const float hf = floorf(him), wf = floorf(wim);
  const float lh = him - hf, lw = wim - wf;
  // the band (-1, H) x (-1, W) as |v - centre| < radius; false for NaN / inf like the reference's four compares
  const bool inband = fabsf(him - 0.5f * (Hf - 1.f)) < 0.5f * (Hf + 1.f) && fabsf(wim - 0.5f * (Wf - 1.f)) < 0.5f * (Wf + 1.f);
  const float f = lw * side_sign + side_one;   // 1 - lw | lw
  const float t = aw * f;
  const float wb = t * lh;
  const float wt = t - wb;
  // top corner of my column relative to the window; the window must hold rows r0, r0 + 1 and column c0
  const int r0 = (int)hf - wy0, c0 = (int)wf - wx0 + side;
  const bool inwin = (unsigned)r0 < (unsigned)(wh - 1) && (unsigned)c0 < (unsigned)ww;
  const bool use = inband && inwin && qvalid;
  T3Record r;
  r.miss = inband && !inwin && qvalid && t != 0.f;
  int rl = r0;
  if (WRAP) {
    rl = r0 + rot;
    rl -= rl >= nr ? nr : 0;
  }
  // (a record that must not contribute still reads: it points at the window's first pixel, which is always staged --
  // LDS the strip has not filled yet may hold NaNs, and 0 * NaN is not 0)
  r.slot = use ? (rl * pitch + c0) * 128 : (WRAP ? rot * pitch * 128 : 0);
  r.wt = use ? wt : 0.f;
  r.wb = use ? wb : 0.f;
  return r;
}

}  // namespace univs
