// Host-side tables of the "strips" MSDA kernel (msda_strips.hip, generation 5): pure C++ (no HIP calls), so that the
// host emulator (tools/strips_emulate.cpp) builds the very tables the kernel reads.
//
// Geometry.  Tiles of TW x TH queries of the finest level (plus the queries of the coarser levels whose reference
// points fall into the tile), numbered column-major: tile = tx * tiles_y + ty, so consecutive tiles are vertical
// neighbours and a workgroup walks down a tile column.  Every level's window of a tile (bilinear footprints of samples
// within R pixels of the tile's box, plus the one-pixel zero ring around the level) is resident in LDS.
//
// LDS layout (one region per level).  A workgroup handles HALF a head (16 channels = 64 bytes per pixel).  A 128-byte
// LDS "super-pixel" holds pixel x of TWO consecutive level rows: with Y = y + 1 >= 0 (row -1 is the zero ring),
//     byte address = region + (((Y >> 1) mod NSR) * pitch + (x - wx0)) * 128 + (Y & 1) * 64 + chunk * 16.
// Rows are circular over the NSR super-rows: moving one tile down replaces only the rows that left the window.  The row
// below a sample's top row is at +64 (same super-row) or in the next super-row, which wraps to super-row 0 (no mirror
// row: it would cost 8 KB of the 80 KB a workgroup may use).  The four 16-byte chunk slots x {x parity} x {row parity} of a
// bilinear footprint are the 16 slots of the 256-byte LDS bank row: a lane visits its four corners and four chunks in a
// lane-specific order (msda_strips.hip) and the 16 lanes of a ds_read_b128 group never share a slot.  `pitch` is even
// (the x parity of the lower corners then equals that of the upper ones).
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

#include "msda_geometry.h"

namespace univs {

constexpr int S5_NW = 8;             // waves of a workgroup; wave w owns the queries [16 w, 16 w + 16) of an item
constexpr int S5_QCAP = 16 * S5_NW;
constexpr int S5_PC = 4;             // row pieces a wave stages in registers per pass
constexpr int S5_PCAP = 32;          // row pieces per wave and list (<= 64: a wave fetches its list with one load)
constexpr int S5_ROWS_MAX = 26, S5_PITCH_MAX = 32;   // window caps (rows, pixels)
constexpr int S5_PX_BIAS = 16;
constexpr int S5_LMAX = 4;
constexpr int S5_DH = 16;            // channels per pass (half a head)
constexpr int S5_LDS_MAX = 80 * 1024;   // two workgroups per CU

// Per level slot (visiting order: largest level first), the same for every tile: a kernel argument.
struct S5Levels {
  int H[S5_LMAX], W[S5_LMAX], start[S5_LMAX], l[S5_LMAX];
  int pitch[S5_LMAX], reg[S5_LMAX], nsr[S5_LMAX];   // LDS row pitch (pixels, even), region byte offset, super-rows
  float rW[S5_LMAX], rH[S5_LMAX];                   // 1 / W, 1 / H correctly rounded (exact division by two FMAs)
  int next_d[S5_LMAX], wrap_d[S5_LMAX];             // byte distance from a sample's top row in the ODD half of a super-row to
                                                    // the row below it, minus 64: pitch * 128 - 128, and the same when the
                                                    // next super-row wraps to super-row 0: -(nsr - 1) * pitch * 128 - 128
};
// Per tile, workgroup-uniform; 16 dwords, fetched with one vector load (lane k = dword k & 15).
struct S5Tile {
  // p0: (wx0 + 1) | (wy0 + 1) << 12 | rot << 24 | par << 30 -- first column / row of the tile's window (zero ring included:
  // >= -1), rot = ((wy0 + 1) >> 1) mod nsr = LDS super-row of the window's first row, par = (wy0 + 1) & 1 = the half of
  // that super-row it lives in;  p1: (ww - 2) | (wh - 2) << 8 -- the upper-left corner of a footprint may sit in window
  // columns [0, ww - 2], rows [0, wh - 2]
  unsigned p0[S5_LMAX], p1[S5_LMAX];
  int total;          // queries of the tile
  int n_cold;         // pieces per wave of this tile's "whole windows" list
  int n_enter_next;   // pieces per wave of the NEXT tile's "entering rows" list (next in the sequence, wrapping)
  int pad[5];
};
static_assert(sizeof(S5Tile) == 64, "16 dwords");
__host__ __device__ __forceinline__ int s5_wx0(unsigned p0) { return (int)(p0 & 0xfffu) - 1; }
__host__ __device__ __forceinline__ int s5_wy0(unsigned p0) { return (int)((p0 >> 12) & 0xfffu) - 1; }
__host__ __device__ __forceinline__ int s5_rot(unsigned p0) { return (int)((p0 >> 24) & 0x3fu); }
__host__ __device__ __forceinline__ int s5_par(unsigned p0) { return (int)((p0 >> 30) & 1u); }
// One (row, 16-pixel column block) of one level's window: what one wave instruction moves (4 lanes x 16 B per pixel).
struct S5Piece {
  unsigned a;   // S5_PX_BIAS + pixel index (start + y * W + x) of the block's first pixel within the frame (24 bits) |
                // level slot << 24
  unsigned b;   // byte offset of the block's first pixel in LDS (row half included)
  unsigned c;   // columns inside the level (load mask, 16 bits) | columns inside the window pitch (store mask) << 16
  unsigned d;   // 0
};

// ---- a lane's sample record at one level: shared by the kernel and the host emulator (tools/strips_emulate.cpp).
// Inputs: the sample's normalised location (x, y) and attention weight (FINITE: like the split-bf16 Linears that produce
// them, this path does not define results for inf / NaN activations), the level's size as floats, the tile's packed window
// words p0 / p1 (S5Tile), the level's nsr / pitch / next_d / wrap_d and the byte address of its LDS region, and the lane's
// low four bits.  Outputs: the LDS byte addresses of the four corners in the lane's visiting order (chunk rotation
// already in bits 4-5: read chunk slot j at a[k] ^ (j << 4)) with their weights, and `inwin`: the footprint lies inside
// the window (otherwise all weights are 0, the addresses point at the window's first pixel, and the caller checks whether
// the sample is inside the band and adds it from global memory).
struct S5Rec {
  unsigned a[4];
  float w[4];
  bool inwin;
};
__host__ __device__ __forceinline__ int s5_floor_to_int(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));   // floor and convert in one (saturating)
  return r;
#else
  return (int)floorf(fminf(fmaxf(v, -1e6f), 1e6f));
#endif
}
__host__ __device__ __forceinline__ unsigned s5_mul24(unsigned a, unsigned b) {   // both < 2^24
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return a * b;
#endif
}
__host__ __device__ __forceinline__ float s5_fract(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fractf(v);                      // v - floor(v), kept below 1
#else
  return v - floorf(v);
#endif
}
__host__ __device__ __forceinline__ S5Rec s5_record(float x, float y, float awt, float Hf, float Wf, unsigned p0, unsigned p1,
                                                    int nsr, int pitch, int next_d, int wrap_d, unsigned region, unsigned lane4) {
  // reference arithmetic: ms_deform_im2col_cuda.cuh:285-293 and :38-89; the window includes the one-pixel zero ring around
  // the level, so out-of-level corners simply read zeros
  const float him = fmaf(y, Hf, -0.5f), wim = fmaf(x, Wf, -0.5f);
  const int r0 = s5_floor_to_int(him) - s5_wy0(p0), c0 = s5_floor_to_int(wim) - s5_wx0(p0);
  const float lh = s5_fract(him), lw = s5_fract(wim);
  // Footprint inside the window?  A window never leaves the ring-extended level, so an in-window sample is inside the
  // reference's band (-1, H) x (-1, W) -- except exactly on its open edge (him == -1), where the bilinear weights of the
  // only in-level row are 0 anyway: the band test of cuh:293 is implied.
  S5Rec rec;
  rec.inwin = (unsigned)r0 <= ((p1 >> 8) & 0xffu) && (unsigned)c0 <= (p1 & 0xffu);
  // (a sample that must not contribute still reads: everything is masked to the window's first pixel, which is always
  // staged; its attention weight becomes an exact 0)
  const unsigned m = rec.inwin ? 0xffffffffu : 0u;
  const float aw = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, awt) & m);
  // window row r0 -> (super-row, half): rows are paired by the parity of y + 1
  const unsigned yrel = ((unsigned)r0 & m) + (unsigned)s5_par(p0);
  const unsigned hy = yrel & 1u;
  unsigned srl = (yrel >> 1) + (unsigned)s5_rot(p0);
  const unsigned srw = srl - (unsigned)nsr;
  srl = srl < srw ? srl : srw;                   // circular: srl - nsr underflows to a huge number unless srl >= nsr
  const unsigned idx = s5_mul24(srl, (unsigned)pitch) + ((unsigned)c0 & m);
  const unsigned r4 = ((lane4 >> 2) & 3u) << 4;
  const unsigned tl = ((idx << 7) + region) | (hy << 6) | r4;
  // the row below: the other half of the same super-pixel (+64), or the first half of the next super-row (circular)
  const int nd = (srl + 1u == (unsigned)nsr) ? wrap_d : next_d;
  const unsigned below = 64u + ((unsigned)nd & (0u - hy));
  const unsigned lb0 = lane4 & 1u, lb1 = (lane4 >> 1) & 1u;
  const unsigned fs = ((tl >> 7) ^ lb0) & 1u;   // which corner column I read first
  const unsigned ft = hy ^ lb1;                 // which corner row I read first
  const unsigned c00 = tl + (fs << 7), c01 = tl + ((fs ^ 1u) << 7);
  const unsigned rowd = below & (0u - ft), rowd2 = below - rowd;
  rec.a[0] = c00 + rowd; rec.a[1] = c01 + rowd; rec.a[2] = c00 + rowd2; rec.a[3] = c01 + rowd2;
  const float f0 = fs ? lw : 1.f - lw;          // column weight of the corner column read first
  const float g0 = ft ? lh : 1.f - lh;          // row weight of the corner row read first
  const float wr0 = aw * g0, wr1 = aw - wr0;
  rec.w[0] = wr0 * f0; rec.w[1] = wr0 - rec.w[0]; rec.w[2] = wr1 * f0; rec.w[3] = wr1 - rec.w[2];
  return rec;
}
// inside the reference's band (-1, H) x (-1, W)?  (only evaluated for samples outside the window: the rare path)
__host__ __device__ __forceinline__ bool s5_inband(float x, float y, float Hf, float Wf) {
  const float him = fmaf(y, Hf, -0.5f), wim = fmaf(x, Wf, -0.5f);
  return him > -1.f && wim > -1.f && him < Hf && wim < Wf;
}

struct S5Host {
  S5Levels lv;
  std::vector<S5Tile> tiles;       // [ntiles]
  std::vector<S5Piece> pieces;     // [ntiles][2][S5_NW][S5_PCAP]: list 0 = entering rows, list 1 = whole windows
  std::vector<int> qtab;           // [ntiles][S5_QCAP]: global query index of the tile's i-th query (padded with the last)
  int ntiles = 0, tiles_x = 0, tiles_y = 0;
  long long qmax = 0;              // max queries of a tile
  size_t lds = 0;                  // bytes of all the levels' circular windows
  bool ok = false;                 // the tables fit their caps
};

static inline int s5_pos_mod(int a, int b) { return ((a % b) + b) % b; }

// fine = index of the largest level.  Returns g.ok.
static bool s5_build_host(const LevelTable& lv, int L, int fine, int TH, int TW, int R, S5Host& g) {
  g = S5Host();
  if (L < 1 || L > S5_LMAX || TH < 1 || TW < 1) return false;
  const int tiles_y = (lv.H[fine] + TH - 1) / TH, tiles_x = (lv.W[fine] + TW - 1) / TW;
  g.tiles_x = tiles_x; g.tiles_y = tiles_y;
  std::vector<int4> ax((size_t)L * tiles_x), ay((size_t)L * tiles_y);
  int pitch[UNIVS_MAX_LEVELS] = {0, 0, 0, 0}, nsr[UNIVS_MAX_LEVELS] = {0, 0, 0, 0};
  for (int l = 0; l < L; ++l) {
    int mw = 2, ms = 1;
    for (int tx = 0; tx < tiles_x; ++tx) {
      int4& e = ax[(size_t)l * tiles_x + tx];
      axis_entry(tx, tiles_x, TW, lv.W[l], lv.W[fine], R, S5_PITCH_MAX, /*ring=*/1, e);
      mw = std::max(mw, e.w);
    }
    for (int ty = 0; ty < tiles_y; ++ty) {
      int4& e = ay[(size_t)l * tiles_y + ty];
      axis_entry(ty, tiles_y, TH, lv.H[l], lv.H[fine], R, S5_ROWS_MAX, /*ring=*/1, e);
      const int Yf = e.z + 1, Yl = e.z + e.w;          // first / last window row, shifted by the ring
      ms = std::max(ms, (Yl >> 1) - (Yf >> 1) + 1);
    }
    pitch[l] = (mw + 1) & ~1;                          // even
    nsr[l] = ms;
  }
  int ord[UNIVS_MAX_LEVELS];
  for (int l = 0; l < L; ++l) ord[l] = l;
  // slot order: by size, largest first, ties by index (ops.msda_level_order builds the projection layout with the same rule)
  std::sort(ord, ord + L, [&](int a, int b) {
    const long long sa = (long long)lv.H[a] * lv.W[a], sb = (long long)lv.H[b] * lv.W[b];
    return sa != sb ? sa > sb : a < b;
  });
  g.ntiles = tiles_y * tiles_x;
  g.ok = true;
  std::memset(&g.lv, 0, sizeof(g.lv));
  size_t lds = 0;
  for (int kk = 0; kk < L; ++kk) {
    const int l = ord[kk];
    g.lv.H[kk] = lv.H[l]; g.lv.W[kk] = lv.W[l]; g.lv.start[kk] = lv.start[l]; g.lv.l[kk] = l;
    g.lv.pitch[kk] = pitch[l]; g.lv.nsr[kk] = nsr[l]; g.lv.reg[kk] = (int)lds;
    g.lv.rW[kk] = 1.0f / (float)lv.W[l]; g.lv.rH[kk] = 1.0f / (float)lv.H[l];
    g.lv.next_d[kk] = pitch[l] * 128 - 128; g.lv.wrap_d[kk] = -(nsr[l] - 1) * pitch[l] * 128 - 128;
    lds += (size_t)nsr[l] * pitch[l] * 128;
  }
  g.lds = lds;
  g.tiles.assign((size_t)g.ntiles, S5Tile());
  g.pieces.assign((size_t)g.ntiles * 2 * S5_NW * S5_PCAP, S5Piece{0u, 0u, 0u, 0u});
  g.qtab.assign((size_t)g.ntiles * S5_QCAP, 0);
  std::vector<int> n_enter((size_t)g.ntiles, 0);
  for (int tx = 0; tx < tiles_x; ++tx)
    for (int ty = 0; ty < tiles_y; ++ty) {
      const size_t tile = (size_t)tx * tiles_y + ty;
      int pre[UNIVS_MAX_LEVELS + 1] = {0};
      for (int l = 0; l < L; ++l) pre[l + 1] = pre[l] + ax[(size_t)l * tiles_x + tx].y * ay[(size_t)l * tiles_y + ty].y;
      g.qmax = std::max<long long>(g.qmax, pre[L]);
      if (pre[L] >= 1 && pre[L] <= S5_QCAP) {
        int last = 0;
        for (int l = 0; l < L; ++l) {
          const int4 gx = ax[(size_t)l * tiles_x + tx], gy = ay[(size_t)l * tiles_y + ty];
          for (int i = 0; i < gx.y * gy.y; ++i)
            g.qtab[tile * S5_QCAP + pre[l] + i] = last = lv.start[l] + (gy.x + i / gx.y) * lv.W[l] + gx.x + i % gx.y;
        }
        for (int i = pre[L]; i < S5_QCAP; ++i) g.qtab[tile * S5_QCAP + i] = last;
      } else {
        g.ok = false;
      }
      S5Tile& t = g.tiles[tile];
      std::memset(&t, 0, sizeof(t));
      t.total = pre[L];
      for (int which = 0; which < 2; ++which) {   // 0: entering rows, 1: whole windows
        int count = 0;
        for (int kk = 0; kk < L; ++kk) {
          const int l = ord[kk];
          const int4 gx = ax[(size_t)l * tiles_x + tx], gy = ay[(size_t)l * tiles_y + ty];
          if (gx.z + 1 < 0 || gx.z + 1 > 0xfff || gy.z + 1 < 0 || gy.z + 1 > 0xfff || nsr[l] > 63 || gx.w < 2 || gy.w < 2 ||
              gx.w - 2 > 0xff || gy.w - 2 > 0xff) g.ok = false;
          t.p0[kk] = (unsigned)(gx.z + 1) | ((unsigned)(gy.z + 1) << 12) | ((unsigned)s5_pos_mod((gy.z + 1) >> 1, nsr[l]) << 24) |
                     ((unsigned)((gy.z + 1) & 1) << 30);
          t.p1[kk] = (unsigned)(gx.w - 2) | ((unsigned)(gy.w - 2) << 8);
          if (gy.z < -1 || gx.z < -1) g.ok = false;   // (axis_entry clips windows to the zero ring)
          int y0 = gy.z, n = gy.w;
          if (which == 0 && ty > 0) {
            const int4 py = ay[(size_t)l * tiles_y + ty - 1];
            y0 = std::max(gy.z, py.z + py.w);
            n = std::max(0, gy.z + gy.w - y0);
          }
          for (int r = 0; r < n; ++r) {
            const int y = y0 + r, Y = y + 1;
            const int sr = s5_pos_mod(Y >> 1, nsr[l]), hy = Y & 1;
            for (int b16 = 0; b16 * 16 < pitch[l]; ++b16) {
              S5Piece pc;
              int px = lv.start[l] + y * lv.W[l] + gx.z + 16 * b16;
              const int ldsoff = g.lv.reg[kk] + (sr * pitch[l] + 16 * b16) * 128 + hy * 64;
              unsigned ldmask = 0, stmask = 0;
              for (int t16 = 0; t16 < 16; ++t16) {
                const int cx = 16 * b16 + t16, x = gx.z + cx;
                if (cx < pitch[l]) stmask |= 1u << t16;
                if (cx < pitch[l] && y >= 0 && y < lv.H[l] && x >= 0 && x < lv.W[l]) ldmask |= 1u << t16;
              }
              px = ldmask ? px + S5_PX_BIAS : 0;
              if (px < 0 || px >= (1 << 24) || ldsoff >= (1 << 20)) g.ok = false;
              pc.a = ((unsigned)px & 0xffffffu) | ((unsigned)kk << 24);
              pc.b = (unsigned)ldsoff;
              pc.c = ldmask | (stmask << 16);
              pc.d = 0u;
              const int w = count % S5_NW, j = count / S5_NW;
              if (j < S5_PCAP) g.pieces[((tile * 2 + which) * S5_NW + w) * S5_PCAP + j] = pc;
              else g.ok = false;
              ++count;
            }
          }
        }
        const int per_wave = (count + S5_NW - 1) / S5_NW;
        if (which == 0) n_enter[tile] = per_wave;
        else t.n_cold = per_wave;
      }
    }
  for (size_t tile = 0; tile < (size_t)g.ntiles; ++tile) g.tiles[tile].n_enter_next = n_enter[(tile + 1) % g.ntiles];
  if (g.qmax < 1 || g.qmax > S5_QCAP) g.ok = false;
  return g.ok;
}

}  // namespace univs
