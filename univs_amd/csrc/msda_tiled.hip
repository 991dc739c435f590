// LDS-tiled multi-scale deformable attention forward (gfx950) for the pixel-decoder encoder geometry.
//
// Why: the direct-gather kernel touches 8 heads x 12 points x 4 corners x 128 B = 49 KB per query
// through the per-CU vector L1 against 3.2 KB/query of algorithmic traffic (SURVEY.md section 7
// "hard parts"), and every one of the 8 lanes that share a 128-B row repeats the same bilinear
// address arithmetic.  Here the gathers go to LDS (256 B/clk/CU for ds_read_b128) and the footprint
// arithmetic is done ONCE per sample.
//
// Preconditions (else the caller falls back to the generic kernel): Lq == S (every pixel of every
// level is a query, in level-major raster order -- the encoder self-attention of
// msdeformattn.py:61-89), D == 32 (one 128-B row per pixel-head), P == 4, L <= 4, levels >= 2x2.
//
// Decomposition: one workgroup = (frame n, head m, spatial tile).  A tile is a TH x TW block of the
// finest level; it owns every query of EVERY level whose centre falls inside the tile's normalised
// box (256 + 64 + 16 queries for a 2x pyramid and a 16x16 tile).  For each value level in turn:
//   phase A  (all threads)   copy the level's window [box * (H_l, W_l) +- R] of this head into LDS
//                            (coalesced 128-B rows, 8 x 16-B loads in flight per lane), and, one
//                            thread per SAMPLE, turn (x, y, attention weight) into a 20-byte record
//                            {LDS slot of the 2x2 footprint, 4 corner weights} in LDS;
//   phase B  (8 lanes/query) read the 4 records of the query by LDS broadcast, then 16 ds_read_b128
//                            (8 lanes = one 128-B row) + 16 x 4 FMAs; nothing else.
// A sample whose footprint is not fully inside the staged window (|offset| > R) is flagged in its
// record and taken straight from global memory by a branch-free 4-load fallback -- results never
// depend on R or on the tiling.  Accumulators (one float4 per lane per query) stay in registers
// across the level loop.
//
// Block order: logical id = ((n * tiles + tile) * M + m), XCD-chunked, so the 8 head-workgroups
// of a tile (which share the 128-B lines of sampling_loc / attn_weight) and neighbouring tiles
// (which share halo rows) run on the same XCD L2.
#include <algorithm>

#include "msda_common.h"

namespace univs {

struct TileGeom {
  int fine;      // index of the finest level (defines the tile grid)
  int TH, TW;    // tile size in finest-level pixels
  int tiles_y, tiles_x;
  int R;         // halo radius in pixels of each value level
  int cap_px;    // LDS window capacity in pixels (device clamps the window to it)
  int ablate;    // profiling only (UNIVS_MSDA_ABLATE): bit0 skip window copy, bit1 skip records, bit2 skip gathers
};

struct Window {  // staged window of the current level (workgroup-uniform)
  int x0, y0, w, h;
};

constexpr int TL_QCAP = 384;            // max queries per tile
constexpr int TL_NSMP = TL_QCAP * 4;    // sample records per level

__device__ __forceinline__ int ceil_div_i(int a, int b) {  // b > 0, any a
  return (a >= 0) ? (a + b - 1) / b : -((-a) / b);
}

// Sample record: weights of the LDS 2x2 block at (by, bx) and its slot; bit 31 of `slot` = "take
// this sample from global memory instead" (then the LDS weights are all zero).
__device__ __forceinline__ void make_record(const Window& win, int H, int W, float x, float y, float aw,
                                            float4& wout, int& slot) {
  const float him = y * (float)H - 0.5f, wim = x * (float)W - 0.5f;
  const bool inb = him > -1.f && wim > -1.f && him < (float)H && wim < (float)W;
  const float hf = floorf(him), wf = floorf(wim);
  const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
  const int h0 = (int)fminf(fmaxf(hf, -2.f), (float)H), w0 = (int)fminf(fmaxf(wf, -2.f), (float)W);
  // true corner weights (zero outside the image / outside the band), reference cuh:38-89, :293
  const bool t = h0 >= 0, b = h0 + 1 <= H - 1, lft = w0 >= 0, rgt = w0 + 1 <= W - 1;
  const float w00 = (inb && t && lft) ? aw * hh * hw : 0.f;
  const float w01 = (inb && t && rgt) ? aw * hh * lw : 0.f;
  const float w10 = (inb && b && lft) ? aw * lh * hw : 0.f;
  const float w11 = (inb && b && rgt) ? aw * lh * lw : 0.f;
  // window-local position of the true top-left corner and of the LDS 2x2 block that serves it
  const int r0 = h0 - win.y0, c0 = w0 - win.x0;
  const int by = min(max(r0, 0), win.h - 2), bx = min(max(c0, 0), win.w - 2);
  const int sy0 = r0 - by, sy1 = sy0 + 1, sx0 = c0 - bx, sx1 = sx0 + 1;  // in {0,1} when served
  float W00 = 0.f, W01 = 0.f, W10 = 0.f, W11 = 0.f;
  bool miss = false;
#define UNIVS_PLACE(sy, sx, wv)                                   \
  do {                                                            \
    const bool ok = ((unsigned)(sy) < 2u) && ((unsigned)(sx) < 2u); \
    if (ok) {                                                     \
      if ((sy) == 0 && (sx) == 0) W00 = (wv);                     \
      if ((sy) == 0 && (sx) == 1) W01 = (wv);                     \
      if ((sy) == 1 && (sx) == 0) W10 = (wv);                     \
      if ((sy) == 1 && (sx) == 1) W11 = (wv);                     \
    } else {                                                      \
      miss = miss || ((wv) != 0.f);                               \
    }                                                             \
  } while (0)
  UNIVS_PLACE(sy0, sx0, w00);
  UNIVS_PLACE(sy0, sx1, w01);
  UNIVS_PLACE(sy1, sx0, w10);
  UNIVS_PLACE(sy1, sx1, w11);
#undef UNIVS_PLACE
  slot = (by * win.w + bx) * 8;
  if (miss) {
    W00 = W01 = W10 = W11 = 0.f;
    slot |= (int)0x80000000;
  }
  wout = make_float4(W00, W01, W10, W11);
}

// THREADS in {512, 1024}; QMAX = TL_QCAP / (THREADS/8) queries per 8-lane group
template <int L, int THREADS>
__global__ __launch_bounds__(THREADS) void msda_fwd_tiled(const float* __restrict__ value,
                                                           LevelTable lv, TileGeom tg,
                                                           const float* __restrict__ loc,
                                                           const float* __restrict__ attn, int N, int S,
                                                           int M, float* __restrict__ out,
                                                           unsigned nblocks) {
  constexpr int D = 32, P = 4, OCTETS = THREADS / 8, QMAX = TL_QCAP / OCTETS;
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  // LDS carve: [records: weights float4 x NSMP][slots int x NSMP][query ids int x QCAP][window]
  float4* rec_w = lds;
  int* rec_s = reinterpret_cast<int*>(lds + TL_NSMP);
  int* qglob = rec_s + TL_NSMP;
  float4* win_lds = lds + TL_NSMP + (TL_NSMP + TL_QCAP) / 4;

  const unsigned bid = xcd_remap(blockIdx.x, nblocks);
  const int m = bid % M;
  const int ntiles = tg.tiles_y * tg.tiles_x;
  const int tile = (bid / M) % ntiles;
  const int n = bid / (M * ntiles);
  const int ty = tile / tg.tiles_x, tx = tile % tg.tiles_x;
  const int Hf = lv.H[tg.fine], Wf = lv.W[tg.fine];
  const int tid = threadIdx.x, lane8 = tid & 7, oct = tid >> 3;
  const int rowf4 = M * (D / 4);

  // ---- queries owned by this tile: per level the half-open index box of pixel centres in the tile
  int qx0[L], qy0[L], qnx[L], pre[L + 1];
  pre[0] = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int Hq = lv.H[l], Wq = lv.W[l];
    int xl = ceil_div_i(2 * tx * tg.TW * Wq - Wf, 2 * Wf);
    int xh = (tx + 1 == tg.tiles_x) ? Wq : ceil_div_i(2 * (tx + 1) * tg.TW * Wq - Wf, 2 * Wf);
    int yl = ceil_div_i(2 * ty * tg.TH * Hq - Hf, 2 * Hf);
    int yh = (ty + 1 == tg.tiles_y) ? Hq : ceil_div_i(2 * (ty + 1) * tg.TH * Hq - Hf, 2 * Hf);
    xl = max(xl, 0); yl = max(yl, 0); xh = min(max(xh, xl), Wq); yh = min(max(yh, yl), Hq);
    qx0[l] = xl; qy0[l] = yl; qnx[l] = xh - xl;
    pre[l + 1] = pre[l] + (xh - xl) * (yh - yl);
  }
  const int total = pre[L];  // <= TL_QCAP (host-checked)

  // global query index of every query of the tile, once
  for (int i = tid; i < total; i += THREADS) {
    int l = 0;
#pragma unroll
    for (int j = 1; j < L; ++j) l += (i >= pre[j]) ? 1 : 0;
    int li = i, x0 = qx0[0], y0 = qy0[0], nx = qnx[0], Wq = lv.W[0], st = lv.start[0];
#pragma unroll
    for (int j = 1; j < L; ++j)
      if (l == j) { li = i - pre[j]; x0 = qx0[j]; y0 = qy0[j]; nx = qnx[j]; Wq = lv.W[j]; st = lv.start[j]; }
    qglob[i] = st + (y0 + li / nx) * Wq + x0 + li % nx;
  }

  float4 acc[QMAX];
#pragma unroll
  for (int k = 0; k < QMAX; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);

  const float4* vn = reinterpret_cast<const float4*>(value + (long long)n * S * M * D + (long long)m * D) + lane8;
  const float* locn = loc + ((long long)n * S * M + m) * (L * P * 2);
  const float* attn_n = attn + ((long long)n * S * M + m) * (L * P);
  const float x0n = (float)(tx * tg.TW) / (float)Wf, x1n = fminf(1.f, (float)((tx + 1) * tg.TW) / (float)Wf);
  const float y0n = (float)(ty * tg.TH) / (float)Hf, y1n = fminf(1.f, (float)((ty + 1) * tg.TH) / (float)Hf);
  const float4* win_lane = win_lds + lane8;

  // Window of level l (workgroup-uniform, recomputed where needed: a handful of scalar ops).
  auto window_of = [&](int l) __attribute__((always_inline)) {
    const int H = lv.H[l], W = lv.W[l];
    Window win;
    win.x0 = max(0, (int)floorf(x0n * (float)W - 0.5f - (float)tg.R));
    win.y0 = max(0, (int)floorf(y0n * (float)H - 0.5f - (float)tg.R));
    const int x1 = min(W - 1, (int)floorf(x1n * (float)W - 0.5f + (float)tg.R) + 1);
    const int y1 = min(H - 1, (int)floorf(y1n * (float)H - 0.5f + (float)tg.R) + 1);
    win.w = max(x1 - win.x0 + 1, 2);                     // levels are >= 2x2 (host-checked)
    win.x0 = min(win.x0, W - win.w);
    win.h = max(y1 - win.y0 + 1, 2);
    win.y0 = min(win.y0, H - win.h);
    win.w = min(win.w, 32);                              // register-staged copy: <= 32 x 32 pixels
    win.h = max(2, min(min(win.h, 32), tg.cap_px / win.w)); // and never exceed the LDS carve
    return win;
  };

  // Register-staged software pipeline (issue early / write late): the global loads of level l+1's
  // window and sample inputs are issued right before phase B of level l and only written to LDS
  // after it, so HBM/L2 latency hides under the LDS gathers even at one workgroup per CU.
  constexpr int WR = 1024 / OCTETS;                 // window float4 per lane  (covers 32 x 32 pixels)
  constexpr int SR = TL_NSMP / THREADS;             // samples per thread
  typedef float v4f __attribute__((ext_vector_type(4)));  // native vector: plain loads/stores, no memcpy
  v4f wreg[WR];
  float2 sxy[SR];
  float sa[SR];
  auto issue = [&](int l) __attribute__((always_inline)) {
    const int W = lv.W[l];
    const Window win = window_of(l);
    // division-free 2-D mapping: 32 lane-groups per window row (windows are <= 32 x 32), ROWS rows
    // per step; out-of-window groups re-read a clamped pixel instead of branching
    const int rx = min(oct & 31, win.w - 1);
    const float4* src0 = vn + ((long long)lv.start[l] + (long long)win.y0 * W + win.x0 + rx) * rowf4;
    if (!(tg.ablate & 1)) {
#pragma unroll
      for (int u = 0; u < WR; ++u) {
        const int ry = min(u * (OCTETS / 32) + (oct >> 5), win.h - 1);
        wreg[u] = *reinterpret_cast<const v4f*>(src0 + (long long)ry * W * rowf4);
      }
    }
    if (!(tg.ablate & 2))
#pragma unroll
    for (int s = 0; s < SR; ++s) {
      const int i = min(tid + s * THREADS, total * 4 - 1);
      const long long e = ((long long)qglob[i >> 2] * M * L + l) * P + (i & 3);
      sxy[s] = reinterpret_cast<const float2*>(locn)[e];
      sa[s] = attn_n[e];
    }
  };
  auto commit = [&](int l) __attribute__((always_inline)) {
    const int H = lv.H[l], W = lv.W[l];
    const Window win = window_of(l);
    const int rx = oct & 31;
#pragma unroll
    for (int u = 0; u < WR; ++u) {
      const int ry = u * (OCTETS / 32) + (oct >> 5);
      if (rx < win.w && ry < win.h) *reinterpret_cast<v4f*>(win_lds + (ry * win.w + rx) * 8 + lane8) = wreg[u];
    }
#pragma unroll
    for (int s = 0; s < SR; ++s) {
      const int i = tid + s * THREADS;
      if (i < total * 4 && !(tg.ablate & 2)) {
        float4 wv;
        int slot;
        make_record(win, H, W, sxy[s].x, sxy[s].y, sa[s], wv, slot);
        rec_w[i] = wv;
        rec_s[i] = slot;
      }
    }
  };

  __syncthreads();  // qglob visible
  if (total > 0) issue(0);

#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    const Window win = window_of(l);
    const float4* vl = vn + (long long)lv.start[l] * rowf4;

    if (l > 0) __syncthreads();  // previous level's phase B is done with the window and the records
    if (total > 0) commit(l);
    __syncthreads();
    if (l + 1 < L && total > 0) issue(l + 1);   // in flight during phase B below

    // ---- phase B: 8 lanes per query, 4 samples x 4 corners from LDS
#pragma unroll
    for (int k = 0; k < QMAX; ++k) {
      const int qi = oct + k * OCTETS;
      if (qi < total && !(tg.ablate & 4)) {
        // SB samples per batch: 4 (all of the level) at 512 threads, 2 at 1024 threads (128-VGPR budget)
        constexpr int SB = (THREADS == 1024) ? 2 : 4;
        int sl[4];
        float4 a = acc[k];
#pragma unroll
        for (int p0 = 0; p0 < 4; p0 += SB) {
          float4 wv[SB];
          float4 v[SB][4];
#pragma unroll
          for (int p = 0; p < SB; ++p) {
            wv[p] = rec_w[qi * 4 + p0 + p];
            sl[p0 + p] = rec_s[qi * 4 + p0 + p];
          }
#pragma unroll
          for (int p = 0; p < SB; ++p) {
            const float4* b = win_lane + (sl[p0 + p] & 0x7fffffff);
            v[p][0] = b[0];
            v[p][1] = b[8];
            v[p][2] = b[win.w * 8];
            v[p][3] = b[win.w * 8 + 8];
          }
#pragma unroll
          for (int p = 0; p < SB; ++p) {
            a = fma4(wv[p].x, v[p][0], a);
            a = fma4(wv[p].y, v[p][1], a);
            a = fma4(wv[p].z, v[p][2], a);
            a = fma4(wv[p].w, v[p][3], a);
          }
        }
        if ((sl[0] | sl[1] | sl[2] | sl[3]) < 0) {
          // rare: footprint(s) outside the staged window -> those samples come from global memory,
          // one sample at a time (4 clamped, always-valid loads in flight; samples served from LDS
          // get weight 0).  Kept narrow on purpose: this path must not raise the register pressure of
          // the common path, which holds the next level's prefetch.
          const long long e = ((long long)qglob[qi] * M * L + l) * P;
#pragma unroll 1
          for (int p = 0; p < 4; ++p) {
            const int slp = (p == 0) ? sl[0] : (p == 1) ? sl[1] : (p == 2) ? sl[2] : sl[3];
            if (slp < 0) {
              const float2 xy = reinterpret_cast<const float2*>(locn)[e + p];
              const Footprint f = footprint(H, W, xy.x, xy.y, attn_n[e + p]);
              const float4 g0 = vl[(long long)(f.h0 * W + f.w0) * rowf4];
              const float4 g1 = vl[(long long)(f.h0 * W + f.w1) * rowf4];
              const float4 g2 = vl[(long long)(f.h1 * W + f.w0) * rowf4];
              const float4 g3 = vl[(long long)(f.h1 * W + f.w1) * rowf4];
              a = fma4(f.w00, g0, a);
              a = fma4(f.w01, g1, a);
              a = fma4(f.w10, g2, a);
              a = fma4(f.w11, g3, a);
            }
          }
        }
        acc[k] = a;
      }
    }
  }

#pragma unroll
  for (int k = 0; k < QMAX; ++k) {
    const int qi = oct + k * OCTETS;
    if (qi < total)
      reinterpret_cast<float4*>(out + (((long long)n * S + qglob[qi]) * M + m) * D)[lane8] = acc[k];
  }
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

static int ceil_div_h(long long a, long long b) {  // host, b > 0
  return (int)((a >= 0) ? (a + b - 1) / b : -((-a) / b));
}

// exact per-level maximum (over tiles) of the number of query rows / columns a tile owns; mirrors
// the device-side box computation
static int max_box(int ntile, int T, int Nq, int Nf) {
  int best = 0;
  for (int t = 0; t < ntile; ++t) {
    int lo = std::max(0, ceil_div_h(2LL * t * T * Nq - Nf, 2LL * Nf));
    int hi = (t + 1 == ntile) ? Nq : ceil_div_h(2LL * (t + 1) * T * Nq - Nf, 2LL * Nf);
    hi = std::min(std::max(hi, lo), Nq);
    best = std::max(best, hi - lo);
  }
  return best;
}

template <int L>
static void launch_tiled(int threads, unsigned nblocks, size_t lds, hipStream_t st, const float* value,
                         const LevelTable& lv, const TileGeom& tg, const float* loc, const float* attn,
                         int N, int S, int M, float* out) {
  if (threads == 1024) {
    auto k = msda_fwd_tiled<L, 1024>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(1024), lds, st, value, lv, tg, loc, attn, N, S, M, out, nblocks);
  } else {
    auto k = msda_fwd_tiled<L, 512>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(512), lds, st, value, lv, tg, loc, attn, N, S, M, out, nblocks);
  }
}

// returns 1 if launched, 0 if preconditions do not hold, <0 on error
int msda_forward_tiled_f32(const float* value, const LevelTable& lv, const float* loc,
                           const float* attn, int N, int S, int M, int D, int L, int Lq, int P,
                           float* out, hipStream_t st) {
  if (D != 32 || P != 4 || L < 1 || L > 4 || Lq != S || M < 1) return 0;
  // levels must tile [0, S) exactly, in order (the encoder's flatten+concat layout)
  long long expect = 0;
  int fine = 0;
  for (int l = 0; l < L; ++l) {
    if (lv.start[l] != expect || lv.H[l] < 2 || lv.W[l] < 2) return 0;
    expect += (long long)lv.H[l] * lv.W[l];
    if ((long long)lv.H[l] * lv.W[l] > (long long)lv.H[fine] * lv.W[fine]) fine = l;
  }
  if (expect != S) return 0;

  TileGeom tg;
  tg.fine = fine;
  tg.TH = env_int("UNIVS_MSDA_TILE_H", 16);
  tg.TW = env_int("UNIVS_MSDA_TILE_W", 16);
  tg.R = env_int("UNIVS_MSDA_HALO", 6);
  tg.ablate = env_int("UNIVS_MSDA_ABLATE", 0);
  const int threads = env_int("UNIVS_MSDA_THREADS", 512) == 1024 ? 1024 : 512;
  if (tg.TH < 1 || tg.TW < 1 || tg.R < 0) return 0;
  tg.tiles_y = (lv.H[fine] + tg.TH - 1) / tg.TH;
  tg.tiles_x = (lv.W[fine] + tg.TW - 1) / tg.TW;

  // static bounds: queries per tile (exact maximum) and LDS window (upper bound over all tiles)
  long long qmax = 0, win_px = 4;
  for (int l = 0; l < L; ++l) {
    const int nx = max_box(tg.tiles_x, tg.TW, lv.W[l], lv.W[fine]);
    const int ny = max_box(tg.tiles_y, tg.TH, lv.H[l], lv.H[fine]);
    qmax += (long long)nx * ny;
    // window extent: box * size +- R, plus the far bilinear corner and rounding slack
    const long long ww = std::min<long long>(lv.W[l], ((long long)tg.TW * lv.W[l] + lv.W[fine] - 1) / lv.W[fine] + 2 * tg.R + 3);
    const long long wh = std::min<long long>(lv.H[l], ((long long)tg.TH * lv.H[l] + lv.H[fine] - 1) / lv.H[fine] + 2 * tg.R + 3);
    win_px = std::max(win_px, ww * wh);
  }
  if (qmax > TL_QCAP) return 0;
  const size_t fixed = (size_t)TL_NSMP * 16 + (size_t)(TL_NSMP + TL_QCAP) * 4;
  const size_t budget = 160 * 1024 - fixed;
  if (win_px * 128 > (long long)budget) win_px = budget / 128;  // device clamps windows to cap_px
  if (win_px > 1024) win_px = 1024;  // register-staged copy covers WR * OCTETS = 1024 pixels
  if (win_px < 64) return 0;
  tg.cap_px = (int)win_px;
  const size_t lds = fixed + (size_t)win_px * 128;

  const long long nb = (long long)N * M * tg.tiles_y * tg.tiles_x;
  if (nb <= 0 || nb > 0x7fffffffLL) return 0;
  const unsigned nblocks = (unsigned)nb;
  switch (L) {
    case 1: launch_tiled<1>(threads, nblocks, lds, st, value, lv, tg, loc, attn, N, S, M, out); break;
    case 2: launch_tiled<2>(threads, nblocks, lds, st, value, lv, tg, loc, attn, N, S, M, out); break;
    case 3: launch_tiled<3>(threads, nblocks, lds, st, value, lv, tg, loc, attn, N, S, M, out); break;
    default: launch_tiled<4>(threads, nblocks, lds, st, value, lv, tg, loc, attn, N, S, M, out); break;
  }
  int rc = check_launch("msda_fwd_tiled");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
