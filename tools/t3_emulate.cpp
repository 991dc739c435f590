// Host emulation of msda_tiled3.hip's data flow (tile geometry -> staged windows -> sample records -> gathers ->
// global fallback) against a plain double-precision bilinear reference (ms_deform_im2col_cuda.cuh:38-89, 242-304).
// It compiles the SAME record builder (csrc/msda_tiled3_record.h), tile geometry (csrc/msda_geometry.h: axis_entry)
// and fallback footprint (csrc/msda_common.h) the kernel uses, so index / clipping / weight bugs show up without a GPU.
//   hipcc -O2 -std=c++17 -I include tools/t3_emulate.cpp -o /tmp/t3_emulate && /tmp/t3_emulate
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../univs_amd/csrc/msda_geometry.h"
#include "../univs_amd/csrc/msda_tiled3_record.h"

namespace univs { void set_error(const char*, ...) {} }
using namespace univs;

struct Case { const char* name; std::vector<std::pair<int, int>> shapes; int N, M, TH, TW, R; float off_std; };

static double ref_sample(const std::vector<float>& value, int S, int M, int n, int m, int start, int H, int W, float x, float y,
                         float aw, int ch) {
  const float him = y * H - 0.5f, wim = x * W - 0.5f;
  if (!(him > -1 && wim > -1 && him < H && wim < W)) return 0.0;
  const int h0 = (int)floorf(him), w0 = (int)floorf(wim);
  const double lh = him - h0, lw = wim - w0;
  auto v = [&](int h, int w) -> double {
    if (h < 0 || w < 0 || h >= H || w >= W) return 0.0;
    return value[(((size_t)n * S + start + (size_t)h * W + w) * M + m) * 32 + ch];
  };
  return aw * ((1 - lh) * (1 - lw) * v(h0, w0) + (1 - lh) * lw * v(h0, w0 + 1) + lh * (1 - lw) * v(h0 + 1, w0) + lh * lw * v(h0 + 1, w0 + 1));
}

int main() {
  std::vector<Case> cases = {
      {"cfg1", {{8, 14}, {16, 28}, {32, 56}}, 2, 8, 8, 16, 6, 2.0f},
      {"ragged", {{5, 7}, {9, 13}, {17, 25}}, 1, 8, 8, 16, 6, 2.0f},
      {"L4-fine-first", {{32, 48}, {16, 24}, {8, 12}, {4, 6}}, 1, 4, 8, 16, 6, 2.5f},
      {"L1", {{20, 33}}, 2, 1, 8, 16, 6, 3.0f},
      {"L2-tiny-halo", {{12, 20}, {24, 40}}, 1, 2, 8, 16, 1, 3.0f},
      {"two-px", {{2, 2}, {4, 4}, {8, 8}}, 1, 2, 8, 16, 6, 2.0f},
      {"cfg2-slice", {{23, 40}, {46, 80}, {92, 160}}, 1, 2, 8, 16, 6, 2.0f},
  };
  int bad_total = 0;
  for (const Case& c : cases) {
    const int L = (int)c.shapes.size();
    LevelTable lv{};
    int S = 0, fine = 0;
    for (int l = 0; l < L; ++l) {
      lv.H[l] = c.shapes[l].first; lv.W[l] = c.shapes[l].second; lv.start[l] = S;
      S += lv.H[l] * lv.W[l];
      if (lv.H[l] * lv.W[l] > lv.H[fine] * lv.W[fine]) fine = l;
    }
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(0.f, 1.f);
    std::vector<float> value((size_t)c.N * S * c.M * 32), loc((size_t)c.N * S * c.M * L * 4 * 2), attn((size_t)c.N * S * c.M * L * 4);
    for (auto& v : value) v = nd(rng);
    for (int n = 0; n < c.N; ++n)
      for (int q = 0; q < S; ++q) {
        int lq = 0;
        for (int l = 0; l < L; ++l) if (q >= lv.start[l]) lq = l;
        const int qi = q - lv.start[lq];
        const float rx = (qi % lv.W[lq] + 0.5f) / lv.W[lq], ry = (qi / lv.W[lq] + 0.5f) / lv.H[lq];
        for (int m = 0; m < c.M; ++m)
          for (int l = 0; l < L; ++l)
            for (int p = 0; p < 4; ++p) {
              const size_t e = ((((size_t)n * S + q) * c.M + m) * L + l) * 4 + p;
              float sc = c.off_std * ((q % 7 == 3) ? 6.f : 1.f);   // every 7th query: far offsets (window misses, out of level)
              loc[e * 2] = rx + nd(rng) * sc / lv.W[l];
              loc[e * 2 + 1] = ry + nd(rng) * sc / lv.H[l];
              attn[e] = (q % 11 == 5 && p == 2) ? 0.f : ud(rng);
            }
      }
    // geometry tables (ring = 0) exactly as geometry() builds them
    const int tiles_y = (lv.H[fine] + c.TH - 1) / c.TH, tiles_x = (lv.W[fine] + c.TW - 1) / c.TW;
    std::vector<int4> tab((size_t)L * (tiles_x + tiles_y));
    for (int l = 0; l < L; ++l) {
      for (int tx = 0; tx < tiles_x; ++tx) axis_entry(tx, tiles_x, c.TW, lv.W[l], lv.W[fine], c.R, 32, 1, tab[(size_t)l * tiles_x + tx]);
      for (int ty = 0; ty < tiles_y; ++ty) axis_entry(ty, tiles_y, c.TH, lv.H[l], lv.H[fine], c.R, 24, 1, tab[(size_t)L * tiles_x + (size_t)l * tiles_y + ty]);
    }
    std::vector<double> out((size_t)c.N * S * c.M * 32, 0.0);
    std::vector<int> covered((size_t)c.N * S * c.M, 0);
    long long nmiss = 0, nsamp = 0, qmax = 0;
    for (int n = 0; n < c.N; ++n)
      for (int ty = 0; ty < tiles_y; ++ty)
        for (int tx = 0; tx < tiles_x; ++tx)
          for (int m = 0; m < c.M; ++m) {
            int pre[UNIVS_MAX_LEVELS + 1] = {0};
            int4 gx[UNIVS_MAX_LEVELS], gy[UNIVS_MAX_LEVELS];
            for (int l = 0; l < L; ++l) {
              gx[l] = tab[(size_t)l * tiles_x + tx];
              gy[l] = tab[(size_t)L * tiles_x + (size_t)l * tiles_y + ty];
              pre[l + 1] = pre[l] + gx[l].y * gy[l].y;
              // window inside the level + its one-pixel zero ring, and inside the fill grid
              if (gx[l].z < -1 || gy[l].z < -1 || gx[l].z + gx[l].w > lv.W[l] + 1 || gy[l].z + gy[l].w > lv.H[l] + 1 || gx[l].w < 2 || gy[l].w < 2 ||
                  gx[l].w > 32 || gy[l].w > 24) {
                printf("%s: window of tile (%d,%d) level %d out of bounds: x %d+%d / %d, y %d+%d / %d\n", c.name, ty, tx, l, gx[l].z, gx[l].w, lv.W[l], gy[l].z, gy[l].w, lv.H[l]);
                ++bad_total;
              }
            }
            const int total = pre[L];
            qmax = std::max<long long>(qmax, total);
            for (int i = 0; i < total; ++i) {
              // qglob (msda_tiled3.hip)
              int li = i, qx0 = gx[0].x, qnx = gx[0].y, qy0 = gy[0].x, Wq = lv.W[0], st = lv.start[0];
              for (int jl = 1; jl < L; ++jl)
                if (i >= pre[jl]) { li = i - pre[jl]; qx0 = gx[jl].x; qnx = gx[jl].y; qy0 = gy[jl].x; Wq = lv.W[jl]; st = lv.start[jl]; }
              const int row = (int)(((float)li + 0.5f) * (1.0f / (float)qnx));
              const int qg = st + (qy0 + row) * Wq + qx0 + (li - row * qnx);
              covered[((size_t)n * S + qg) * c.M + m]++;
              for (int l = 0; l < L; ++l) {
                const int H = lv.H[l], W = lv.W[l], wx0 = gx[l].z, ww = gx[l].w, wy0 = gy[l].z, wh = gy[l].w;
                for (int p = 0; p < 4; ++p) {
                  const size_t e = ((((size_t)n * S + qg) * c.M + m) * L + l) * 4 + p;
                  const float x = loc[e * 2], y = loc[e * 2 + 1], aw = attn[e];
                  ++nsamp;
                  for (int side = 0; side < 2; ++side) {
                    const int pitch = (ww + 7) / 8 * 8;
                    const T3Record r = t3_record(x, y, aw, side, side ? 1.f : -1.f, side ? 0.f : 1.f, true, H, W, wx0, wy0, ww, wh, pitch);
                    const int px = r.slot / 128, wr = px / pitch, wc = px % pitch;
                    if (r.slot < 0 || wr + 1 >= wh || wc >= ww) { printf("%s: slot out of window\n", c.name); ++bad_total; }
                    auto staged = [&](int row, int col, int ch) -> float {   // what the fill waves put there: level data or the zero ring
                      const int gy_ = wy0 + row, gx_ = wx0 + col;
                      if (gy_ < 0 || gy_ >= H || gx_ < 0 || gx_ >= W) return 0.f;
                      return value[(((size_t)n * S + lv.start[l] + (size_t)gy_ * W + gx_) * c.M + m) * 32 + ch];
                    };
                    for (int ch = 0; ch < 32; ++ch) {
                      double acc = 0;
                      if (r.wt != 0.f || r.wb != 0.f) acc = (double)r.wt * staged(wr, wc, ch) + (double)r.wb * staged(wr + 1, wc, ch);
                      if (r.miss) {
                        const Footprint f = footprint(H, W, x, y, aw);
                        const int cc = side ? f.w1 : f.w0;
                        const float g0 = value[(((size_t)n * S + lv.start[l] + (size_t)f.h0 * W + cc) * c.M + m) * 32 + ch];
                        const float g1 = value[(((size_t)n * S + lv.start[l] + (size_t)f.h1 * W + cc) * c.M + m) * 32 + ch];
                        acc += (double)(side ? f.w01 : f.w00) * g0 + (double)(side ? f.w11 : f.w10) * g1;
                        if (ch == 0) ++nmiss;
                      }
                      out[(((size_t)n * S + qg) * c.M + m) * 32 + ch] += acc;
                    }
                  }
                }
              }
            }
          }
    // coverage + values
    int bad = 0;
    double maxerr = 0;
    for (size_t i = 0; i < covered.size(); ++i)
      if (covered[i] != 1) { if (bad < 5) printf("%s: query-head %zu covered %d times\n", c.name, i, covered[i]); ++bad; }
    for (int n = 0; n < c.N; ++n)
      for (int q = 0; q < S; ++q)
        for (int m = 0; m < c.M; ++m)
          for (int ch = 0; ch < 32; ch += 5) {
            double ref = 0;
            for (int l = 0; l < L; ++l)
              for (int p = 0; p < 4; ++p) {
                const size_t e = ((((size_t)n * S + q) * c.M + m) * L + l) * 4 + p;
                ref += ref_sample(value, S, c.M, n, m, lv.start[l], lv.H[l], lv.W[l], loc[e * 2], loc[e * 2 + 1], attn[e], ch);
              }
            const double err = fabs(ref - out[(((size_t)n * S + q) * c.M + m) * 32 + ch]);
            maxerr = std::max(maxerr, err);
            if (err > 2e-5 * std::max(1.0, fabs(ref))) { if (bad < 5) printf("%s: n%d q%d m%d ch%d ref %g got %g\n", c.name, n, q, m, ch, ref, out[(((size_t)n * S + q) * c.M + m) * 32 + ch]); ++bad; }
          }
    printf("%-14s S=%5d tiles %dx%d qmax=%lld  samples %lld  miss-sides %lld (%.3f%%)  max err %.2e  %s\n", c.name, S, tiles_y, tiles_x, qmax, nsamp, nmiss,
           100.0 * nmiss / (2.0 * nsamp), maxerr, bad ? "FAIL" : "ok");
    bad_total += bad;
  }
  return bad_total ? 1 : 0;
}
