// Host emulation of msda_strips.hip's data flow (tables -> row pieces -> circular super-row windows in LDS -> sample
// records -> gathers in the lane-specific corner / chunk order -> point reduction -> output channels; global fallback for
// samples that leave the window) against a plain double-precision bilinear reference (ms_deform_im2col_cuda.cuh:38-89,
// 242-304) on the STANDARD layouts.  It compiles the SAME table builder and record function the kernel uses
// (csrc/msda_strips_geom.h: s5_build_host, s5_record), re-creates the head-major operand layouts the Linear epilogues
// write, and checks on the way that (a) every ds_read_b128 lane group of the gather touches 16 different 16-byte slots
// (bank-conflict-free by construction), (b) no LDS byte is read before the current window wrote it (stale circular rows).
//   hipcc -O2 -std=c++17 -I include tools/strips_emulate.cpp -o /tmp/strips_emulate && /tmp/strips_emulate
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../univs_amd/csrc/msda_strips_geom.h"

namespace univs { void set_error(const char*, ...) {} }
using namespace univs;

struct Case { const char* name; std::vector<std::pair<int, int>> shapes; int N, M, TH, TW, R; float off_std; int nwg; };

static double ref_sample(const std::vector<float>& value, int S, int M, int n, int m, int start, int H, int W, float x, float y,
                         double aw, int ch) {
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

// the ds_read_b128 lane groups of gfx950 (MI355X_MICROARCH.md, LDS table)
static const int GROUPS[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                  {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                  {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                  {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

int main() {
  std::vector<Case> cases = {
      {"cfg1", {{8, 14}, {16, 28}, {32, 56}}, 2, 8, 8, 12, 6, 2.0f, 3},
      {"ragged", {{5, 7}, {9, 13}, {17, 25}}, 1, 8, 8, 12, 6, 2.0f, 2},
      {"L4-fine-first", {{32, 48}, {16, 24}, {8, 12}, {4, 6}}, 1, 4, 8, 12, 6, 2.5f, 5},
      {"L1", {{20, 33}}, 2, 1, 8, 12, 6, 3.0f, 2},
      {"L2-tiny-halo", {{12, 20}, {24, 40}}, 1, 2, 8, 12, 1, 3.0f, 1},
      {"two-px", {{2, 2}, {4, 4}, {8, 8}}, 1, 2, 8, 12, 6, 2.0f, 1},
      {"cfg2-slice", {{23, 40}, {46, 80}, {92, 160}}, 1, 2, 8, 12, 6, 2.0f, 7},
      {"cfg2-th6", {{23, 40}, {46, 80}, {92, 160}}, 1, 1, 6, 12, 6, 2.0f, 4},
      {"cfg5-slice", {{34, 60}, {68, 120}, {136, 240}}, 1, 1, 8, 12, 6, 2.0f, 6},
  };
  int bad_total = 0;
  for (const Case& c : cases) {
    const int L = (int)c.shapes.size(), P = 4;
    LevelTable lv{};
    int S = 0, fine = 0;
    for (int l = 0; l < L; ++l) {
      lv.H[l] = c.shapes[l].first; lv.W[l] = c.shapes[l].second; lv.start[l] = S;
      S += lv.H[l] * lv.W[l];
      if (lv.H[l] * lv.W[l] > lv.H[fine] * lv.W[fine]) fine = l;
    }
    const int N = c.N, M = c.M;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    // standard layouts: value [N][S][M][32]; raw projections: offsets [N][S][M][L][P][2] (pixels of the target level), logits
    // [N][S][M][L][P]; reference points [S][L][2] (pixel centres of the query's own level)
    std::vector<float> value((size_t)N * S * M * 32), off((size_t)N * S * M * L * P * 2), logit((size_t)N * S * M * L * P), ref((size_t)S * L * 2);
    for (auto& v : value) v = nd(rng);
    for (auto& v : off) v = nd(rng) * c.off_std;
    for (auto& v : logit) v = nd(rng);
    for (int lq = 0; lq < L; ++lq)
      for (int i = 0; i < lv.H[lq] * lv.W[lq]; ++i)
        for (int l = 0; l < L; ++l) {
          ref[((size_t)(lv.start[lq] + i) * L + l) * 2 + 0] = ((i % lv.W[lq]) + 0.5f) / lv.W[lq];
          ref[((size_t)(lv.start[lq] + i) * L + l) * 2 + 1] = ((i / lv.W[lq]) + 0.5f) / lv.H[lq];
        }
    // every 11th query: far offsets (misses, partly outside the image)
    for (int n = 0; n < N; ++n)
      for (int q = 0; q < S; ++q)
        if (q % 11 == 5)
          for (size_t i = 0; i < (size_t)M * L * P * 2; ++i) off[((size_t)n * S + q) * M * L * P * 2 + i] *= 5.f;
    int order[4] = {0, 1, 2, 3};
    std::sort(order, order + L, [&](int a, int b) {
      const long long sa = (long long)lv.H[a] * lv.W[a], sb = (long long)lv.H[b] * lv.W[b];
      return sa != sb ? sa > sb : a < b;
    });
    // head-major operands as the Linear epilogues write them
    std::vector<float> vhm((size_t)N * M * 2 * S * 16), qhm((size_t)N * M * S * P * 3 * L);
    for (int n = 0; n < N; ++n)
      for (int s = 0; s < S; ++s)
        for (int m = 0; m < M; ++m) {
          for (int ch = 0; ch < 32; ++ch)
            vhm[((((size_t)n * M + m) * 2 + ch / 16) * S + s) * 16 + ch % 16] = value[(((size_t)n * S + s) * M + m) * 32 + ch];
          for (int p = 0; p < P; ++p) {   // levels in SLOT order (largest first, ties by index), as ops.msda_pack_head_major
            float* row = &qhm[((((size_t)n * M + m) * S + s) * P + p) * 3 * L];
            for (int kk = 0; kk < L; ++kk) {
              const int l = order[kk];
              row[2 * kk] = off[(((((size_t)n * S + s) * M + m) * L + l) * P + p) * 2];
              row[2 * kk + 1] = off[(((((size_t)n * S + s) * M + m) * L + l) * P + p) * 2 + 1];
              row[2 * L + kk] = logit[((((size_t)n * S + s) * M + m) * L + l) * P + p];
            }
          }
        }
    S5Host g;
    for (int TH = c.TH; TH >= 2; TH -= 2) {   // as msda_forward_strips_f32 chooses the tile height
      s5_build_host(lv, L, fine, TH, c.TW, c.R, g);
      if (g.ok && g.lds <= (size_t)S5_LDS_MAX) break;
      g.ok = false;
    }
    if (!g.ok) { printf("%-14s tables not ok (qmax %lld lds %zu)\n", c.name, g.qmax, g.lds); ++bad_total; continue; }
    const unsigned nitems = (unsigned)((long long)N * M * 2 * g.ntiles);
    std::vector<float> out((size_t)N * S * M * 32, 0.f), cnt((size_t)N * S * M * 32, 0.f);
    long long conflicts = 0, stale = 0, misses = 0, samples = 0, reads = 0, inexact_div = 0;
    for (int wg = 0; wg < c.nwg; ++wg) {
      const unsigned g0 = (unsigned)((unsigned long long)wg * nitems / c.nwg), g1 = (unsigned)((unsigned long long)(wg + 1) * nitems / c.nwg);
      if (g0 >= g1) continue;
      std::vector<float> lds(g.lds / 4, NAN);
      std::vector<unsigned> stamp(g.lds / 16, 0xffffffffu);   // item that wrote each 16-byte slot last
      auto item_of = [&](unsigned gi, int& tile, int& n, int& m, int& half, unsigned& hd) {
        gi = std::min(gi, g1 - 1);
        hd = gi / g.ntiles; tile = (int)(gi - hd * g.ntiles); half = hd & 1; n = (int)((hd >> 1) / M); m = (int)((hd >> 1) % M);
      };
      auto move_rows = [&](unsigned gi, int which, unsigned stamp_val) {
        int tile, n, m, half; unsigned hd;
        item_of(gi, tile, n, m, half, hd);
        const float* vbase = &vhm[(size_t)hd * S * 16];
        for (int wave = 0; wave < S5_NW; ++wave)
          for (int k = 0; k < S5_PCAP; ++k) {
            const S5Piece pc = g.pieces[(((size_t)tile * 2 + which) * S5_NW + wave) * S5_PCAP + k];
            for (int lane = 0; lane < 64; ++lane) {
              const int lpx = lane >> 2, lch = lane & 3;
              const unsigned lanebit = 1u << lpx;
              float v[4] = {0, 0, 0, 0};
              if (pc.c & lanebit) {
                const long long px = (long long)(pc.a & 0xffffffu) + lpx - S5_PX_BIAS;
                if (px < 0 || px >= S) { printf("piece pixel out of the frame\n"); ++bad_total; continue; }
                for (int e = 0; e < 4; ++e) v[e] = vbase[px * 16 + lch * 4 + e];
              }
              if ((pc.c >> 16) & lanebit) {
                const unsigned dst = pc.b + lpx * 128 + lch * 16;
                if (dst + 16 > g.lds) { printf("LDS store out of range\n"); ++bad_total; continue; }
                for (int e = 0; e < 4; ++e) lds[dst / 4 + e] = v[e];
                stamp[dst / 16] = stamp_val;
              }
            }
          }
      };
      move_rows(g0, 1, g0);   // cold start: the whole windows of the first tile
      for (unsigned gi = g0; gi < g1; ++gi) {
        int tile, n, m, half; unsigned hd;
        item_of(gi, tile, n, m, half, hd);
        const S5Tile& t = g.tiles[tile];
        for (int wave = 0; wave < S5_NW; ++wave) {
          float acc[64][4][4];
          for (auto& a : acc) for (auto& b : a) for (auto& x : b) x = 0.f;
          int qg[64];
          float xs[64][4], ys[64][4], as[64][4];
          for (int lane = 0; lane < 64; ++lane) {
            const int qi = lane & 15, pt = lane >> 4;
            qg[lane] = g.qtab[(size_t)tile * S5_QCAP + wave * 16 + qi];
            const float* row = &qhm[((((size_t)n * M + m) * S + qg[lane]) * P + pt) * 3 * L];
            for (int kk = 0; kk < L; ++kk) {
              const int l = g.lv.l[kk];
              // the kernel's division: reciprocal multiply + exact-remainder correction; must equal the IEEE quotient
              const float Wf = (float)g.lv.W[kk], Hf = (float)g.lv.H[kk];
              if (g.lv.l[kk] != order[kk]) { printf("slot order mismatch\n"); ++bad_total; }
              const float qx = row[2 * kk] * g.lv.rW[kk], qy = row[2 * kk + 1] * g.lv.rH[kk];
              const float ox = fmaf(fmaf(-qx, Wf, row[2 * kk]), g.lv.rW[kk], qx), oy = fmaf(fmaf(-qy, Hf, row[2 * kk + 1]), g.lv.rH[kk], qy);
              if (ox != row[2 * kk] / Wf || oy != row[2 * kk + 1] / Hf) ++inexact_div;
              xs[lane][kk] = ref[((size_t)qg[lane] * L + l) * 2] + ox;
              ys[lane][kk] = ref[((size_t)qg[lane] * L + l) * 2 + 1] + oy;
              as[lane][kk] = row[2 * L + kk];
            }
          }
          for (int qi = 0; qi < 16; ++qi) {   // softmax over the L * P logits of the query (the 4 DPP rows)
            float mx = -INFINITY, sum = 0.f;
            for (int pt = 0; pt < 4; ++pt) for (int kk = 0; kk < L; ++kk) mx = fmaxf(mx, as[pt * 16 + qi][kk]);
            for (int pt = 0; pt < 4; ++pt) for (int kk = 0; kk < L; ++kk) { as[pt * 16 + qi][kk] = expf(as[pt * 16 + qi][kk] - mx); sum += as[pt * 16 + qi][kk]; }
            for (int pt = 0; pt < 4; ++pt) for (int kk = 0; kk < L; ++kk) as[pt * 16 + qi][kk] /= sum;
          }
          for (int kk = 0; kk < L; ++kk) {
            S5Rec rec[64];
            for (int lane = 0; lane < 64; ++lane) {
              rec[lane] = s5_record(xs[lane][kk], ys[lane][kk], as[lane][kk], (float)g.lv.H[kk], (float)g.lv.W[kk], t.p0[kk], t.p1[kk],
                                    g.lv.nsr[kk], g.lv.pitch[kk], g.lv.next_d[kk], g.lv.wrap_d[kk], (unsigned)g.lv.reg[kk], lane & 15);
              ++samples;
            }
            for (int k = 0; k < 4; ++k)
              for (int j = 0; j < 4; ++j) {
                for (int gr = 0; gr < 4; ++gr) {   // (a) the 16 lanes of a ds_read_b128 group hit 16 different slots
                  unsigned seen = 0;
                  for (int i = 0; i < 16; ++i) {
                    const unsigned addr = rec[GROUPS[gr][i]].a[k] ^ (unsigned)(j << 4);
                    const unsigned slot = (addr >> 4) & 15u;
                    if (seen & (1u << slot)) ++conflicts;
                    seen |= 1u << slot;
                  }
                }
                for (int lane = 0; lane < 64; ++lane) {
                  const unsigned addr = rec[lane].a[k] ^ (unsigned)(j << 4);
                  if (addr + 16 > g.lds || (addr & 15)) { printf("%s: LDS read out of range / misaligned\n", c.name); ++bad_total; continue; }
                  ++reads;
                  if (rec[lane].w[k] != 0.f) {
                    // (b) a contributing read must see data of THIS item's windows: written not before the previous cold start
                    if (stamp[addr / 16] == 0xffffffffu) ++stale;
                    for (int e = 0; e < 4; ++e) acc[lane][j][e] = fmaf(rec[lane].w[k], lds[addr / 4 + e], acc[lane][j][e]);
                  }
                }
              }
            for (int lane = 0; lane < 64; ++lane)
              if (!rec[lane].inwin && as[lane][kk] != 0.f && s5_inband(xs[lane][kk], ys[lane][kk], (float)g.lv.H[kk], (float)g.lv.W[kk])) {   // global fallback
                ++misses;
                const Footprint fp = footprint(g.lv.H[kk], g.lv.W[kk], xs[lane][kk], ys[lane][kk], as[lane][kk]);
                const float* vl = &vhm[((size_t)hd * S + g.lv.start[kk]) * 16];
                const unsigned orot = (lane >> 2) & 3;
                for (int ch = 0; ch < 16; ++ch) {
                  const float tot = fp.w00 * vl[(size_t)(fp.h0 * g.lv.W[kk] + fp.w0) * 16 + ch] + fp.w01 * vl[(size_t)(fp.h0 * g.lv.W[kk] + fp.w1) * 16 + ch] +
                                    fp.w10 * vl[(size_t)(fp.h1 * g.lv.W[kk] + fp.w0) * 16 + ch] + fp.w11 * vl[(size_t)(fp.h1 * g.lv.W[kk] + fp.w1) * 16 + ch];
                  acc[lane][(ch / 4) ^ orot][ch % 4] += tot;   // chunk slot j holds channel chunk j ^ rot
                }
              }
          }
          // point reduction: row r of the wave ends up with chunk slot r of each query = channel chunk r ^ rot4
          for (int qi = 0; qi < 16; ++qi)
            for (int r = 0; r < 4; ++r) {
              const unsigned rot4 = (qi >> 2) & 3;
              const unsigned ca = (unsigned)r ^ rot4;
              for (int e = 0; e < 4; ++e) {
                float tot = 0.f;
                for (int pt = 0; pt < 4; ++pt) tot += acc[pt * 16 + qi][r][e];
                const size_t o = (((size_t)n * S + qg[qi]) * M + m) * 32 + half * 16 + ca * 4 + e;
                out[o] = tot;
                cnt[o] += 1.f;
              }
            }
        }
        if (gi + 1 < g1) {
          int t2, n2, m2, h2; unsigned hd2;
          item_of(gi + 1, t2, n2, m2, h2, hd2);
          // the kernel always commits list 0 of the next item (entering rows; whole windows at the top of a column)
          move_rows(gi + 1, 0, gi + 1);
          if (hd2 != hd && t2 != 0) { printf("%s: (frame, head, half) changed inside a column\n", c.name); ++bad_total; }
        }
      }
    }
    // compare with the double-precision reference on the standard layouts
    double maxerr = 0;
    long long uncovered = 0;
    for (int n = 0; n < N; ++n)
      for (int q = 0; q < S; q += (S > 6000 ? 7 : 1))
        for (int m = 0; m < M; ++m) {
          double lg[4][4], mx = -1e30, sum = 0;
          for (int l = 0; l < L; ++l) for (int p = 0; p < P; ++p) { lg[l][p] = logit[((((size_t)n * S + q) * M + m) * L + l) * P + p]; mx = std::max(mx, lg[l][p]); }
          for (int l = 0; l < L; ++l) for (int p = 0; p < P; ++p) { lg[l][p] = exp((double)(float)expf((float)(lg[l][p] - mx)) > 0 ? lg[l][p] - mx : lg[l][p] - mx); lg[l][p] = exp(lg[l][p] - 0.0); }
          // (plain double softmax)
          sum = 0;
          for (int l = 0; l < L; ++l) for (int p = 0; p < P; ++p) { lg[l][p] = exp((double)logit[((((size_t)n * S + q) * M + m) * L + l) * P + p] - mx); sum += lg[l][p]; }
          for (int ch = 0; ch < 32; ch += 5) {
            double r = 0;
            for (int l = 0; l < L; ++l)
              for (int p = 0; p < P; ++p) {
                const float x = ref[((size_t)q * L + l) * 2] + off[(((((size_t)n * S + q) * M + m) * L + l) * P + p) * 2] / (float)lv.W[l];
                const float y = ref[((size_t)q * L + l) * 2 + 1] + off[(((((size_t)n * S + q) * M + m) * L + l) * P + p) * 2 + 1] / (float)lv.H[l];
                r += ref_sample(value, S, M, n, m, lv.start[l], lv.H[l], lv.W[l], x, y, lg[l][p] / sum, ch);
              }
            const size_t o = (((size_t)n * S + q) * M + m) * 32 + ch;
            if (cnt[o] < 1.f) ++uncovered;
            maxerr = std::max(maxerr, fabs(r - (double)out[o]));
          }
        }
    long long zero_cnt = 0;
    for (float v : cnt) zero_cnt += v < 1.f;
    const bool ok = maxerr < 2e-5 && conflicts == 0 && stale == 0 && zero_cnt == 0 && uncovered == 0 && inexact_div == 0;
    printf("%-14s tiles %3d (%dx%d) lds %6zu B qmax %3lld: max err %.2e, bank conflicts %lld, stale reads %lld, unwritten outputs %lld, "
           "inexact divisions %lld, misses %.3f %% of %lld samples  %s\n", c.name, g.ntiles, g.tiles_x, g.tiles_y, g.lds, g.qmax, maxerr, conflicts, stale, zero_cnt, inexact_div,
           100.0 * misses / std::max<long long>(samples, 1), samples, ok ? "ok" : "FAIL");
    if (!ok) ++bad_total;
  }
  printf(bad_total ? "FAILED\n" : "all ok\n");
  return bad_total ? 1 : 0;
}
