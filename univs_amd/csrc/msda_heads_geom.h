// Host-side tables of the "heads" MSDA kernel (msda_heads.hip, generation 6): pure C++ (no HIP calls), so that the host
// emulator (tools/heads_emulate.cpp) builds the very tables the kernel reads.
//
// Geometry.  Tiles of TW x TH queries of the finest level (plus the queries of the coarser levels whose reference points
// fall into the tile), numbered column-major: tile = tx * tiles_y + ty.  A workgroup walks down SEGMENTS of tile columns
// (S6Seg: plane = (frame, head), first tile, count); every level's window of the current tile (bilinear footprints of samples
// within R pixels of the tile's box, plus the one-pixel zero ring around the level) is resident in LDS.
//
// LDS layout (one region per level).  A workgroup handles a FULL head: 32 channels = 128 bytes per pixel = half a 256-byte
// LDS bank row.  With Y = y + 1 >= 0 (row -1 is the zero ring) and rows circular over the level's NR window rows,
//     byte address = region + ((Y mod NR) * pitch + (x - wx0)) * 128 + chunk * 16          (pitch even, region % 256 == 0)
// so bit 7 of a pixel's address is the parity of its window column.  The 16 lanes of a ds_read_b128 group have 16
// different low-four lane bits (MI355X_MICROARCH.md, LDS table): a lane reads the horizontal corner whose column parity
// equals its lane bit 3 first, and the eight 16-byte chunks of a pixel in the order j ^ (lane & 7): the 16 lanes of a group
// hit 16 different 16-byte slots of the 256-byte bank row whatever pixels they sample -- conflict-free by construction.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

#include "msda_geometry.h"

namespace univs {

constexpr int S6_NW = 8;             // waves of a workgroup; wave w owns the queries [16 w, 16 w + 16) of an item
constexpr int S6_QCAP = 16 * S6_NW;  // queries per item
constexpr int S6_PC = 8;             // row pieces a wave stages in registers per pass
constexpr int S6_PCAP = 32;          // row pieces per wave and list (<= 64: a wave fetches its list with one load)
constexpr int S6_ROWS_MAX = 40, S6_PITCH_MAX = 48;   // window caps (rows, pixels)
constexpr int S6_PX_BIAS = 8;
constexpr int S6_LMAX = 4;
constexpr int S6_DH = 32;            // channels per lane-sample (a full head)
constexpr int S6_LDS_MAX = 160 * 1024;   // one workgroup per CU

// Per level slot (visiting order: largest level first), the same for every tile: a kernel argument.
struct S6Levels {
  int H[S6_LMAX], W[S6_LMAX], start[S6_LMAX], l[S6_LMAX];
  int pitch[S6_LMAX], reg[S6_LMAX], nr[S6_LMAX];   // LDS row pitch (pixels, even), region byte offset, circular rows
  float rW[S6_LMAX], rH[S6_LMAX];                  // 1 / W, 1 / H correctly rounded (exact division by two FMAs)
  int next_d[S6_LMAX], wrap_d[S6_LMAX];            // byte distance from a row to the row below it: pitch * 128, and the same
                                                   // when the row below wraps to row 0: -(nr - 1) * pitch * 128
};
// Per tile, workgroup-uniform; 16 dwords, fetched with one vector load (lane k = dword k & 15).
struct S6Tile {
  // p0: (wx0 + 1) | (wy0 + 1) << 12 | rot << 24 -- first column / row of the tile's window (zero ring included: >= -1),
  // rot = (wy0 + 1) mod nr = LDS row of the window's first row;  p1: (ww - 2) | (wh - 2) << 8 -- the upper-left corner of a
  // footprint may sit in window columns [0, ww - 2], rows [0, wh - 2]
  unsigned p0[S6_LMAX], p1[S6_LMAX];
  int total;          // queries of the tile
  int n_cold;         // pieces per wave of this tile's "whole windows" list
  int n_enter;        // pieces per wave of this tile's "entering rows" list (rows its windows have and the windows of the
                      // tile above it -- ty - 1 of the same column -- have not; the whole windows at the top of a column)
  int pad[5];
};
static_assert(sizeof(S6Tile) == 64, "16 dwords");
__host__ __device__ __forceinline__ int s6_wx0(unsigned p0) { return (int)(p0 & 0xfffu) - 1; }
__host__ __device__ __forceinline__ int s6_wy0(unsigned p0) { return (int)((p0 >> 12) & 0xfffu) - 1; }
__host__ __device__ __forceinline__ int s6_rot(unsigned p0) { return (int)((p0 >> 24) & 0x3fu); }
// One (row, 8-pixel column block) of one level's window: what one wave instruction moves (8 lanes x 16 B per pixel).
struct S6Piece {
  unsigned a;   // S6_PX_BIAS + pixel index (start + y * W + x) of the block's first pixel within the frame (24 bits) |
                // columns inside the level (load mask, 8 bits) << 24
  unsigned b;   // byte offset of the block's first pixel in LDS (24 bits) | columns inside the window pitch (store mask) << 24
};
// A workgroup's work: `count` consecutive tiles of one column of one (frame, head) plane, first tile `tile0`.
struct S6Seg {
  int plane, tile0, count, pad;
};

// ---- a lane's sample record at one level: shared by the kernel and the host emulator (tools/heads_emulate.cpp).
// Inputs: the sample's normalised location (x, y) and attention weight (FINITE), the level's size as floats, the tile's
// packed window words p0 / p1 (S6Tile), the level's nr / pitch / next_d / wrap_d, the byte address of its LDS region and the
// lane's low four bits.  Outputs: the LDS byte addresses of the four corners in the lane's visiting order (first / second
// horizontal corner of the top row, then of the bottom row; the lane's chunk rotation already in bits 4-6: read chunk slot j at
// a[k] ^ (j << 4)) with their weights, and `inwin`: the footprint lies inside the window (otherwise all weights are 0, the
// addresses point at the window's first pixel, and the caller checks whether the sample is inside the band and adds it from
// global memory).
struct S6Rec {
  unsigned a[4];
  float w[4];
  bool inwin;
};
__host__ __device__ __forceinline__ int s6_floor_to_int(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));   // floor and convert in one (saturating)
  return r;
#else
  return (int)floorf(fminf(fmaxf(v, -1e6f), 1e6f));
#endif
}
__host__ __device__ __forceinline__ unsigned s6_mul24(unsigned a, unsigned b) {   // both < 2^24
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul24(a, b);
#else
  return a * b;
#endif
}
__host__ __device__ __forceinline__ float s6_fract(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_fractf(v);                      // v - floor(v), kept below 1
#else
  return v - floorf(v);
#endif
}
__host__ __device__ __forceinline__ S6Rec s6_record(float x, float y, float awt, float Hf, float Wf, unsigned p0, unsigned p1,
                                                    int nr, int pitch, int next_d, int wrap_d, unsigned region, unsigned lane4) {
  // reference arithmetic: ms_deform_im2col_cuda.cuh:285-293 and :38-89; the window includes the one-pixel zero ring around
  // the level, so out-of-level corners simply read zeros
  const float him = fmaf(y, Hf, -0.5f), wim = fmaf(x, Wf, -0.5f);
  const int r0 = s6_floor_to_int(him) - s6_wy0(p0), c0 = s6_floor_to_int(wim) - s6_wx0(p0);
  const float lh = s6_fract(him), lw = s6_fract(wim);
  // Footprint inside the window?  A window never leaves the ring-extended level, so an in-window sample is inside the
  // reference's band (-1, H) x (-1, W) -- except exactly on its open edge (him == -1), where the bilinear weights of the
  // only in-level row are 0 anyway: the band test of cuh:293 is implied.
  S6Rec rec;
  rec.inwin = (unsigned)r0 <= ((p1 >> 8) & 0xffu) && (unsigned)c0 <= (p1 & 0xffu);
  // (a sample that must not contribute still reads: everything is masked to the window's first pixel, which is always
  // staged; its attention weight becomes an exact 0)
  const unsigned m = rec.inwin ? 0xffffffffu : 0u;
  const float aw = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, awt) & m);
  const unsigned cm = (unsigned)c0 & m;
  unsigned row = ((unsigned)r0 & m) + (unsigned)s6_rot(p0);
  const unsigned roww = row - (unsigned)nr;
  row = row < roww ? row : roww;                 // circular: row - nr underflows to a huge number unless row >= nr
  const unsigned idx = s6_mul24(row, (unsigned)pitch) + cm;
  const unsigned fs = (cm ^ (lane4 >> 3)) & 1u;  // 1: the RIGHT corner's column parity equals my lane bit 3 -> I read it first
  const unsigned base = (idx << 7) + region + ((lane4 & 7u) << 4);
  const unsigned first = base + (fs << 7), second = base + ((fs ^ 1u) << 7);
  const unsigned nd = (unsigned)((row + 1u == (unsigned)nr) ? wrap_d : next_d);
  rec.a[0] = first; rec.a[1] = second; rec.a[2] = first + nd; rec.a[3] = second + nd;
  const float f0 = fs ? lw : 1.f - lw;           // column weight of the corner column read first
  const float wb = aw * lh, wt = aw - wb;        // row weights: bottom, top
  rec.w[0] = wt * f0; rec.w[1] = wt - rec.w[0]; rec.w[2] = wb * f0; rec.w[3] = wb - rec.w[2];
  return rec;
}
// inside the reference's band (-1, H) x (-1, W)?  (only evaluated for samples outside the window: the rare path)
__host__ __device__ __forceinline__ bool s6_inband(float x, float y, float Hf, float Wf) {
  const float him = fmaf(y, Hf, -0.5f), wim = fmaf(x, Wf, -0.5f);
  return him > -1.f && wim > -1.f && him < Hf && wim < Wf;
}

struct S6Host {
  S6Levels lv;
  std::vector<S6Tile> tiles;       // [ntiles]
  std::vector<S6Piece> pieces;     // [ntiles][2][S6_NW][S6_PCAP]: list 0 = entering rows, list 1 = whole windows
  std::vector<int> qtab;           // [ntiles][S6_QCAP]: global query index of the tile's i-th query (padded with the last)
  int ntiles = 0, tiles_x = 0, tiles_y = 0;
  long long qmax = 0;              // max queries of a tile
  size_t lds = 0;                  // bytes of all the levels' circular windows
  bool ok = false;                 // the tables fit their caps
};

static inline int s6_pos_mod(int a, int b) { return ((a % b) + b) % b; }

// fine = index of the largest level.  Returns g.ok.
static bool s6_build_host(const LevelTable& lv, int L, int fine, int TH, int TW, int R, S6Host& g) {
  g = S6Host();
  if (L < 1 || L > S6_LMAX || TH < 1 || TW < 1) return false;
  const int tiles_y = (lv.H[fine] + TH - 1) / TH, tiles_x = (lv.W[fine] + TW - 1) / TW;
  g.tiles_x = tiles_x; g.tiles_y = tiles_y;
  std::vector<int4> ax((size_t)L * tiles_x), ay((size_t)L * tiles_y);
  int pitch[UNIVS_MAX_LEVELS] = {0, 0, 0, 0}, nr[UNIVS_MAX_LEVELS] = {0, 0, 0, 0};
  for (int l = 0; l < L; ++l) {
    int mw = 2, mh = 2;
    for (int tx = 0; tx < tiles_x; ++tx) {
      int4& e = ax[(size_t)l * tiles_x + tx];
      axis_entry(tx, tiles_x, TW, lv.W[l], lv.W[fine], R, S6_PITCH_MAX, /*ring=*/1, e);
      mw = std::max(mw, e.w);
    }
    for (int ty = 0; ty < tiles_y; ++ty) {
      int4& e = ay[(size_t)l * tiles_y + ty];
      axis_entry(ty, tiles_y, TH, lv.H[l], lv.H[fine], R, S6_ROWS_MAX, /*ring=*/1, e);
      mh = std::max(mh, e.w);
    }
    pitch[l] = (mw + 1) & ~1;                          // even
    nr[l] = mh;
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
    g.lv.pitch[kk] = pitch[l]; g.lv.nr[kk] = nr[l]; g.lv.reg[kk] = (int)lds;
    g.lv.rW[kk] = 1.0f / (float)lv.W[l]; g.lv.rH[kk] = 1.0f / (float)lv.H[l];
    g.lv.next_d[kk] = pitch[l] * 128; g.lv.wrap_d[kk] = -(nr[l] - 1) * pitch[l] * 128;
    lds += (size_t)nr[l] * pitch[l] * 128;             // a multiple of 256: pitch is even
  }
  g.lds = lds;
  g.tiles.assign((size_t)g.ntiles, S6Tile());
  g.pieces.assign((size_t)g.ntiles * 2 * S6_NW * S6_PCAP, S6Piece{0u, 0u});
  g.qtab.assign((size_t)g.ntiles * S6_QCAP, 0);
  for (int tx = 0; tx < tiles_x; ++tx)
    for (int ty = 0; ty < tiles_y; ++ty) {
      const size_t tile = (size_t)tx * tiles_y + ty;
      int pre[UNIVS_MAX_LEVELS + 1] = {0};
      for (int l = 0; l < L; ++l) pre[l + 1] = pre[l] + ax[(size_t)l * tiles_x + tx].y * ay[(size_t)l * tiles_y + ty].y;
      g.qmax = std::max<long long>(g.qmax, pre[L]);
      if (pre[L] >= 1 && pre[L] <= S6_QCAP) {
        int last = 0;
        for (int l = 0; l < L; ++l) {
          const int4 gx = ax[(size_t)l * tiles_x + tx], gy = ay[(size_t)l * tiles_y + ty];
          for (int i = 0; i < gx.y * gy.y; ++i)
            g.qtab[tile * S6_QCAP + pre[l] + i] = last = lv.start[l] + (gy.x + i / gx.y) * lv.W[l] + gx.x + i % gx.y;
        }
        for (int i = pre[L]; i < S6_QCAP; ++i) g.qtab[tile * S6_QCAP + i] = last;
      } else {
        g.ok = false;
      }
      S6Tile& t = g.tiles[tile];
      std::memset(&t, 0, sizeof(t));
      t.total = pre[L];
      for (int which = 0; which < 2; ++which) {   // 0: entering rows, 1: whole windows
        int count = 0;
        for (int kk = 0; kk < L; ++kk) {
          const int l = ord[kk];
          const int4 gx = ax[(size_t)l * tiles_x + tx], gy = ay[(size_t)l * tiles_y + ty];
          if (gx.z + 1 < 0 || gx.z + 1 > 0xfff || gy.z + 1 < 0 || gy.z + 1 > 0xfff || nr[l] > 63 || gx.w < 2 || gy.w < 2 ||
              gx.w - 2 > 0xff || gy.w - 2 > 0xff) g.ok = false;
          t.p0[kk] = (unsigned)(gx.z + 1) | ((unsigned)(gy.z + 1) << 12) | ((unsigned)s6_pos_mod(gy.z + 1, nr[l]) << 24);
          t.p1[kk] = (unsigned)(gx.w - 2) | ((unsigned)(gy.w - 2) << 8);
          if (gy.z < -1 || gx.z < -1) g.ok = false;   // (axis_entry clips windows to the zero ring)
          int y0 = gy.z, n = gy.w;
          if (which == 0 && ty > 0) {
            const int4 py = ay[(size_t)l * tiles_y + ty - 1];
            if (gy.z < py.z) g.ok = false;            // windows move down monotonically
            y0 = std::max(gy.z, py.z + py.w);
            n = std::max(0, gy.z + gy.w - y0);
          }
          for (int r = 0; r < n; ++r) {
            const int y = y0 + r, Y = y + 1;
            const int rs = s6_pos_mod(Y, nr[l]);
            for (int b8 = 0; b8 * 8 < pitch[l]; ++b8) {
              S6Piece pc;
              int px = lv.start[l] + y * lv.W[l] + gx.z + 8 * b8;
              const int ldsoff = g.lv.reg[kk] + (rs * pitch[l] + 8 * b8) * 128;
              unsigned ldmask = 0, stmask = 0;
              for (int t8 = 0; t8 < 8; ++t8) {
                const int cx = 8 * b8 + t8, x = gx.z + cx;
                if (cx < pitch[l]) stmask |= 1u << t8;
                if (cx < pitch[l] && y >= 0 && y < lv.H[l] && x >= 0 && x < lv.W[l]) ldmask |= 1u << t8;
              }
              px = ldmask ? px + S6_PX_BIAS : 0;
              if (px < 0 || px >= (1 << 24) || ldsoff >= (1 << 20)) g.ok = false;
              pc.a = ((unsigned)px & 0xffffffu) | (ldmask << 24);
              pc.b = (unsigned)ldsoff | (stmask << 24);
              const int w = count % S6_NW, j = count / S6_NW;
              if (j < S6_PCAP) g.pieces[((tile * 2 + which) * S6_NW + w) * S6_PCAP + j] = pc;
              else g.ok = false;
              ++count;
            }
          }
        }
        const int per_wave = (count + S6_NW - 1) / S6_NW;
        if (which == 0) t.n_enter = per_wave;
        else t.n_cold = per_wave;
      }
    }
  if (g.qmax < 1 || g.qmax > S6_QCAP) g.ok = false;
  return g.ok;
}

// ---- who does what: `grid` workgroups, each a list of segments (workgroup b: segs[begin[b]] .. segs[begin[b + 1] - 1]), for
// `planes` (frame, head) planes of tiles_x columns of tiles_y tiles.  Workgroup b runs on XCD b % 8 (MI355X_MICROARCH.md: observed,
// used for speed only).
//   policy 0 -- contiguous ranges: the (plane, column, row) sequence is cut into `grid` equal ranges, the workgroups of an XCD
//     own neighbouring ranges (generation 5's rule).  Balanced to one tile, but two workgroups that work on horizontally
//     adjacent tiles do so at different times: each fetches its own halo columns from HBM.
//   policy 1 -- lockstep rounds: every XCD owns a contiguous eighth of the sequence and its W = grid / 8 workgroups walk W
//     ADJACENT columns top to bottom at the same time, round after round, so that the halo columns two neighbours share are
//     fetched from HBM once and found in that XCD's L2 by the other; the columns left over after the last full round are cut
//     into equal pieces or equal ranges, whichever estimates the smaller makespan (a cold start priced at `cold_steps` tiles).
static bool s6_build_segments_lists(int planes, int tiles_x, int tiles_y, int grid, int policy, double cold_steps,
                                    std::vector<std::vector<S6Seg>>& lists) {
  lists.assign((size_t)std::max(grid, 0), std::vector<S6Seg>());
  const long long ncol = (long long)planes * tiles_x, total = ncol * tiles_y;
  if (grid < 1 || total < 1) return false;
  auto push = [&](int wg, long long col, int y0, int n) -> bool {   // rows [y0, y0 + n) of linear column `col`
    if (n <= 0) return true;
    const int plane = (int)(col / tiles_x), tx = (int)(col % tiles_x);
    lists[(size_t)wg].push_back(S6Seg{plane, tx * tiles_y + y0, n, 0});
    return true;
  };
  auto range = [&](int wg, long long g0, long long g1) -> bool {   // the steps [g0, g1) of the sequence, cut at column ends
    while (g0 < g1) {
      const long long col = g0 / tiles_y;
      const int y0 = (int)(g0 % tiles_y), n = (int)std::min<long long>(g1 - g0, tiles_y - y0);
      if (!push(wg, col, y0, n)) return false;
      g0 += n;
    }
    return true;
  };
  const int nx = 8;
  if (policy == 0 || grid % nx != 0 || grid / nx < 2) {
    const int gq = grid / std::min(nx, grid), gr = grid % std::min(nx, grid), nxe = std::min(nx, grid);
    for (int b = 0; b < grid; ++b) {
      const int xcd = b % nxe, widx = b / nxe;
      const long long lw = (long long)xcd * gq + std::min(xcd, gr) + widx;
      if (!range(b, lw * total / grid, (lw + 1) * total / grid)) return false;
    }
    return true;
  }
  const int W = grid / nx;
  for (int x = 0; x < nx; ++x) {
    const long long s0 = (long long)x * total / nx, s1 = (long long)(x + 1) * total / nx;
    if (s0 >= s1) continue;
    struct Col { long long col; int y0, n; };
    std::vector<Col> cols;
    for (long long g0 = s0; g0 < s1;) {
      const int y0 = (int)(g0 % tiles_y), n = (int)std::min<long long>(s1 - g0, tiles_y - y0);
      cols.push_back(Col{g0 / tiles_y, y0, n});
      g0 += n;
    }
    const size_t full = cols.size() / W * W;
    for (size_t i = 0; i < full; ++i)
      if (!push((int)(x + nx * (i % W)), cols[i].col, cols[i].y0, cols[i].n)) return false;
    const size_t rem = cols.size() - full;
    if (!rem) continue;
    // The columns left over after the last full round, two ways; the smaller estimated makespan (tiles + `cold_steps` per
    // segment) wins:
    //  (a) every column cut into V equal pieces (V in 1..4), dealt longest-first: one segment per piece;
    //  (b) their tiles in sequence order cut into W equal ranges: balanced to one tile whatever the column count, a range that
    //      crosses a column end is two segments.
    struct Pc { long long col; int y0, n, wg; };
    auto plan_a = [&](int V, std::vector<Pc>& pcs) -> double {
      pcs.clear();
      for (size_t i = full; i < cols.size(); ++i)
        for (int v = 0; v < V; ++v) {
          const int a = cols[i].n * v / V, b = cols[i].n * (v + 1) / V;
          if (b > a) pcs.push_back(Pc{cols[i].col, cols[i].y0 + a, b - a, 0});
        }
      std::stable_sort(pcs.begin(), pcs.end(), [](const Pc& a, const Pc& b) { return a.n > b.n; });
      std::vector<double> load((size_t)W, 0.0);
      for (Pc& p : pcs) {
        p.wg = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        load[p.wg] += p.n + cold_steps;
      }
      return *std::max_element(load.begin(), load.end());
    };
    auto plan_b = [&](std::vector<Pc>& pcs) -> double {
      pcs.clear();
      long long rtot = 0;
      for (size_t i = full; i < cols.size(); ++i) rtot += cols[i].n;
      size_t ci = full;
      int used = 0;   // tiles of cols[ci] already dealt
      std::vector<double> load((size_t)W, 0.0);
      for (int j = 0; j < W; ++j) {
        long long want = rtot * (j + 1) / W - rtot * j / W;
        while (want > 0 && ci < cols.size()) {
          const int n = (int)std::min<long long>(want, cols[ci].n - used);
          pcs.push_back(Pc{cols[ci].col, cols[ci].y0 + used, n, j});
          load[j] += n + cold_steps;
          used += n;
          want -= n;
          if (used == cols[ci].n) { ++ci; used = 0; }
        }
      }
      return *std::max_element(load.begin(), load.end());
    };
    std::vector<Pc> best, cand;
    double bestms = plan_b(best);
    for (int V = 1; V <= 4; ++V) {
      const double ms = plan_a(V, cand);
      if (ms < bestms - 1e-9) { bestms = ms; best.swap(cand); }
    }
    for (const Pc& p : best)
      if (!push(x + nx * p.wg, p.col, p.y0, p.n)) return false;
  }
  return true;
}
static bool s6_build_segments(int planes, int tiles_x, int tiles_y, int grid, int policy, double cold_steps, std::vector<S6Seg>& segs,
                              std::vector<int>& begin) {
  std::vector<std::vector<S6Seg>> lists;
  segs.clear();
  begin.assign((size_t)std::max(grid, 0) + 1, 0);
  if (!s6_build_segments_lists(planes, tiles_x, tiles_y, grid, policy, cold_steps, lists)) return false;
  for (int b = 0; b < grid; ++b) {
    begin[b] = (int)segs.size();
    segs.insert(segs.end(), lists[b].begin(), lists[b].end());
  }
  begin[grid] = (int)segs.size();
  if (segs.empty()) segs.push_back(S6Seg{0, 0, 0, 0});   // (never read: keeps the upload non-empty)
  return true;
}

}  // namespace univs
