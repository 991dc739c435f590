// LDS-tiled multi-scale deformable attention forward (gfx950) for the pixel-decoder encoder geometry.
//
// Why: the direct-gather kernel touches 8 heads x 12 points x 4 corners x 128 B = 49 KB per query
// through the per-CU vector L1 (~64 B/clk/CU, ~37 TB/s chip-wide) against 3.2 KB/query of
// algorithmic traffic (SURVEY.md section 7 "hard parts"), so it cannot get past ~30 % of the HBM
// roofline.  LDS delivers 256 B/clk/CU for ds_read_b128, so the gathers are moved there.
//
// Preconditions (else the caller falls back to the generic kernel): Lq == S (every pixel of every
// level is a query, in level-major raster order -- the encoder self-attention of
// msdeformattn.py:61-89), D == 32 (one 128-B row per pixel-head), P == 4, L <= 4.
//
// Decomposition: one workgroup = (frame n, head m, spatial tile).  A tile is a TH x TW block of the
// finest level; it owns every query of EVERY level whose centre falls inside the tile's normalised
// box (so 256 + 64 + 16 queries for a 2x pyramid).  For each value level in turn the workgroup
// stages the level's window [box * (H_l, W_l) +- R] for its head into LDS (coalesced 128-B rows),
// then every query takes its 4 samples of that level from LDS with ds_read_b128 (8 lanes = one
// 128-B row).  Samples that fall outside the staged window (|offset| > R) take the same corner
// straight from global memory -- results never depend on R or on the tiling.
// Accumulators (one float4 per lane per query) stay in registers across the level loop.
//
// Block order: logical id = ((n * tiles + tile) * M + m), XCD-chunked, so the 8 head-workgroups
// of a tile (which share the 128-B lines of sampling_loc / attn_weight) and neighbouring tiles
// (which share halo rows) run on the same XCD L2.
#include "common.h"

namespace univs {

struct TileGeom {
  int fine;      // index of the finest level (defines the tile grid)
  int TH, TW;    // tile size in finest-level pixels
  int tiles_y, tiles_x;
  int R;         // halo radius in pixels of each value level
};

constexpr int TL_THREADS = 256;
constexpr int TL_OCTETS = TL_THREADS / 8;
constexpr int TL_QMAX = 12;  // queries per 8-lane group (registers are statically indexed)

__device__ __forceinline__ int ceil_div_i(int a, int b) {  // b > 0, any a
  return (a >= 0) ? (a + b - 1) / b : -((-a) / b);
}

__device__ __forceinline__ float4 fma4t(float s, float4 v, float4 a) {
  a.x = fmaf(s, v.x, a.x);
  a.y = fmaf(s, v.y, a.y);
  a.z = fmaf(s, v.z, a.z);
  a.w = fmaf(s, v.w, a.w);
  return a;
}

struct Window {  // staged window of the current level (workgroup-uniform)
  int x0, y0, w, h;
};

// one bilinear sample; each corner comes from LDS when it is inside the staged window, else global
__device__ __forceinline__ float4 sample_tiled(const float4* __restrict__ lds,
                                               const float4* __restrict__ vl, int rowf4, int H, int W,
                                               const Window& win, int lane8, float x, float y,
                                               float aw, float4 acc) {
  const float him = y * (float)H - 0.5f, wim = x * (float)W - 0.5f;
  if (him > -1.f && wim > -1.f && him < (float)H && wim < (float)W) {
    const float hf = floorf(him), wf = floorf(wim);
    const int h0 = (int)hf, w0 = (int)wf;
    const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
    const int ly = h0 - win.y0, lx = w0 - win.x0;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v1 = z, v2 = z, v3 = z, v4 = z;
    const bool t = h0 >= 0, b = h0 + 1 <= H - 1, lft = w0 >= 0, rgt = w0 + 1 <= W - 1;
    // whole 2x2 footprint inside the window (the common case): 4 LDS reads, no further tests
    if (ly >= 0 && lx >= 0 && ly + 1 < win.h && lx + 1 < win.w) {
      const float4* p = lds + (ly * win.w + lx) * 8 + lane8;
      v1 = p[0];
      v2 = p[8];
      v3 = p[win.w * 8];
      v4 = p[win.w * 8 + 8];
      // window is clipped to the image, so "inside the window" implies "inside the image"
    } else {
      const long long p00 = ((long long)h0 * W + w0) * rowf4;
      const bool iy0 = ly >= 0 && ly < win.h, iy1 = ly + 1 >= 0 && ly + 1 < win.h;
      const bool ix0 = lx >= 0 && lx < win.w, ix1 = lx + 1 >= 0 && lx + 1 < win.w;
      if (t && lft) v1 = (iy0 && ix0) ? lds[(ly * win.w + lx) * 8 + lane8] : vl[p00];
      if (t && rgt) v2 = (iy0 && ix1) ? lds[(ly * win.w + lx + 1) * 8 + lane8] : vl[p00 + rowf4];
      if (b && lft)
        v3 = (iy1 && ix0) ? lds[((ly + 1) * win.w + lx) * 8 + lane8] : vl[p00 + (long long)W * rowf4];
      if (b && rgt)
        v4 = (iy1 && ix1) ? lds[((ly + 1) * win.w + lx + 1) * 8 + lane8]
                          : vl[p00 + (long long)W * rowf4 + rowf4];
    }
    acc = fma4t(aw * hh * hw, v1, acc);
    acc = fma4t(aw * hh * lw, v2, acc);
    acc = fma4t(aw * lh * hw, v3, acc);
    acc = fma4t(aw * lh * lw, v4, acc);
  }
  return acc;
}

template <int L>
__global__ __launch_bounds__(TL_THREADS) void msda_fwd_tiled(const float* __restrict__ value,
                                                              LevelTable lv, TileGeom tg,
                                                              const float* __restrict__ loc,
                                                              const float* __restrict__ attn, int N,
                                                              int S, int M, float* __restrict__ out,
                                                              unsigned nblocks) {
  extern __shared__ __attribute__((aligned(16))) float4 lds[];
  constexpr int D = 32, P = 4;
  const unsigned bid = xcd_remap(blockIdx.x, nblocks);
  const int m = bid % M;
  const int ntiles = tg.tiles_y * tg.tiles_x;
  const int tile = (bid / M) % ntiles;
  const int n = bid / (M * ntiles);
  const int ty = tile / tg.tiles_x, tx = tile % tg.tiles_x;
  const int Hf = lv.H[tg.fine], Wf = lv.W[tg.fine];
  const int lane8 = threadIdx.x & 7, oct = threadIdx.x >> 3;
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
  const int total = pre[L];

  // this lane-group's queries (global query index, -1 = none); static indexing only
  int qidx[TL_QMAX];
#pragma unroll
  for (int k = 0; k < TL_QMAX; ++k) {
    const int i = oct + k * TL_OCTETS;
    int q = -1;
    if (i < total) {
      int l = 0;
#pragma unroll
      for (int j = 1; j < L; ++j) l += (i >= pre[j]) ? 1 : 0;
      int li = i, x0 = qx0[0], y0 = qy0[0], nx = qnx[0], Wq = lv.W[0], st = lv.start[0];
#pragma unroll
      for (int j = 1; j < L; ++j)
        if (l == j) { li = i - pre[j]; x0 = qx0[j]; y0 = qy0[j]; nx = qnx[j]; Wq = lv.W[j]; st = lv.start[j]; }
      q = st + (y0 + li / nx) * Wq + x0 + li % nx;
    }
    qidx[k] = q;
  }

  float4 acc[TL_QMAX];
#pragma unroll
  for (int k = 0; k < TL_QMAX; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);

  const float4* vn = reinterpret_cast<const float4*>(value + (long long)n * S * M * D + (long long)m * D) + lane8;
  const float* locn = loc + ((long long)n * S * M + m) * (L * P * 2);
  const float* attn_n = attn + ((long long)n * S * M + m) * (L * P);
  const float x0n = (float)(tx * tg.TW) / (float)Wf, x1n = fminf(1.f, (float)((tx + 1) * tg.TW) / (float)Wf);
  const float y0n = (float)(ty * tg.TH) / (float)Hf, y1n = fminf(1.f, (float)((ty + 1) * tg.TH) / (float)Hf);

#pragma unroll 1
  for (int l = 0; l < L; ++l) {
    const int H = lv.H[l], W = lv.W[l];
    Window win;
    win.x0 = max(0, (int)floorf(x0n * (float)W - 0.5f - (float)tg.R));
    win.y0 = max(0, (int)floorf(y0n * (float)H - 0.5f - (float)tg.R));
    const int x1 = min(W - 1, (int)floorf(x1n * (float)W - 0.5f + (float)tg.R) + 1);
    const int y1 = min(H - 1, (int)floorf(y1n * (float)H - 0.5f + (float)tg.R) + 1);
    win.w = x1 - win.x0 + 1;
    win.h = y1 - win.y0 + 1;
    const float4* vl = vn + (long long)lv.start[l] * rowf4;

    __syncthreads();  // previous level's gathers are done with the LDS window
    {
      // flattened window copy, 8 independent 16-B loads in flight per lane (latency, not bandwidth,
      // bounds this phase at one workgroup per CU)
      constexpr int UN = 8;
      const int npx = win.w * win.h;
      const float inv_w = 1.0f / (float)win.w;
      const float4* src0 = vl + ((long long)win.y0 * W + win.x0) * rowf4;
      for (int base = 0; base < npx; base += TL_OCTETS * UN) {
        float4 tmp[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = base + u * TL_OCTETS + oct;
          if (i < npx) {
            int ry = (int)((float)i * inv_w);          // approximate i / win.w, then correct
            int rx = i - ry * win.w;
            if (rx < 0) { --ry; rx += win.w; }
            if (rx >= win.w) { ++ry; rx -= win.w; }
            tmp[u] = src0[((long long)ry * W + rx) * rowf4];
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int i = base + u * TL_OCTETS + oct;
          if (i < npx) lds[i * 8 + lane8] = tmp[u];
        }
      }
    }
    __syncthreads();

#pragma unroll
    for (int k = 0; k < TL_QMAX; ++k) {
      const int q = qidx[k];
      if (q >= 0) {
        const float* lp = locn + ((long long)q * M * L + l) * (P * 2);
        const float* ap = attn_n + ((long long)q * M * L + l) * P;
        const float4 c0 = reinterpret_cast<const float4*>(lp)[0];
        const float4 c1 = reinterpret_cast<const float4*>(lp)[1];
        const float4 aw = reinterpret_cast<const float4*>(ap)[0];
        float4 a = acc[k];
        a = sample_tiled(lds, vl, rowf4, H, W, win, lane8, c0.x, c0.y, aw.x, a);
        a = sample_tiled(lds, vl, rowf4, H, W, win, lane8, c0.z, c0.w, aw.y, a);
        a = sample_tiled(lds, vl, rowf4, H, W, win, lane8, c1.x, c1.y, aw.z, a);
        a = sample_tiled(lds, vl, rowf4, H, W, win, lane8, c1.z, c1.w, aw.w, a);
        acc[k] = a;
      }
    }
  }

#pragma unroll
  for (int k = 0; k < TL_QMAX; ++k) {
    const int q = qidx[k];
    if (q >= 0)
      reinterpret_cast<float4*>(out + (((long long)n * S + q) * M + m) * D)[lane8] = acc[k];
  }
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
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
    if (lv.start[l] != expect) return 0;
    expect += (long long)lv.H[l] * lv.W[l];
    if ((long long)lv.H[l] * lv.W[l] > (long long)lv.H[fine] * lv.W[fine]) fine = l;
  }
  if (expect != S) return 0;

  TileGeom tg;
  tg.fine = fine;
  tg.TH = env_int("UNIVS_MSDA_TILE_H", 16);
  tg.TW = env_int("UNIVS_MSDA_TILE_W", 16);
  tg.R = env_int("UNIVS_MSDA_HALO", 6);
  if (tg.TH < 1 || tg.TW < 1 || tg.R < 0) return 0;
  tg.tiles_y = (lv.H[fine] + tg.TH - 1) / tg.TH;
  tg.tiles_x = (lv.W[fine] + tg.TW - 1) / tg.TW;

  // static bounds: queries per tile and LDS window (upper bounds over all tiles)
  long long qmax = 0, win_px = 0;
  for (int l = 0; l < L; ++l) {
    const long long nx = ((long long)tg.TW * lv.W[l] + lv.W[fine] - 1) / lv.W[fine] + 1;
    const long long ny = ((long long)tg.TH * lv.H[l] + lv.H[fine] - 1) / lv.H[fine] + 1;
    qmax += nx * ny;
    const long long ww = std::min<long long>(lv.W[l], nx + 2 * tg.R + 3);
    const long long wh = std::min<long long>(lv.H[l], ny + 2 * tg.R + 3);
    win_px = std::max(win_px, ww * wh);
  }
  if (qmax > (long long)TL_QMAX * TL_OCTETS) return 0;
  const size_t lds = (size_t)win_px * 128;
  if (lds > 160 * 1024) return 0;

  const long long nb = (long long)N * M * tg.tiles_y * tg.tiles_x;
  if (nb <= 0 || nb > 0x7fffffffLL) return 0;
  const unsigned nblocks = (unsigned)nb;
#define UNIVS_LAUNCH_TILED(LL)                                                                     \
  do {                                                                                             \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_fwd_tiled<LL>),                      \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
    hipLaunchKernelGGL((msda_fwd_tiled<LL>), dim3(nblocks), dim3(TL_THREADS), lds, st, value, lv,  \
                       tg, loc, attn, N, S, M, out, nblocks);                                      \
  } while (0)
  switch (L) {
    case 1: UNIVS_LAUNCH_TILED(1); break;
    case 2: UNIVS_LAUNCH_TILED(2); break;
    case 3: UNIVS_LAUNCH_TILED(3); break;
    default: UNIVS_LAUNCH_TILED(4); break;
  }
#undef UNIVS_LAUNCH_TILED
  int rc = check_launch("msda_fwd_tiled");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
