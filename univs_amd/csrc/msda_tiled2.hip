// LDS-tiled MSDA forward, second generation (gfx950): double-buffered windows + wave specialisation.
//
// Same decomposition, record format, zero-ring buffer loads and conflict-free 16-lane gather groups as
// msda_tiled.hip (read that file's header first).  What changes is the schedule.  msda_tiled.hip spends
// 45 % of an item outside the gathers because its single 115-KB window forces the sequence
//     barrier -> commit (LDS writes, records) -> barrier -> gathers
// with every wave in the same phase at the same time: the LDS pipe idles while the vector-memory pipe
// issues the window loads and vice versa.  Here a tile is 16 x 8 pixels of the finest level (168 queries),
// so two windows fit in LDS side by side (660 + 396 pixels at halo 6), and the 16 waves of a workgroup
// are split into
//     6 PRODUCER waves: wait for the staged loads of step s+1, write that window (into the region the
//                       consumers are not reading) and its sample records, issue the loads of step s+2;
//     10 CONSUMER waves: gathers of step s (LDS reads + packed FMAs), at the item's last level the
//                       column reduction and the output stores;
// with ONE workgroup barrier per step.  A step therefore costs max(producer, consumer) instead of their
// sum, and the two pipes run side by side.
//
// Region plan (host, per geometry): step parity picks the region, A = the largest window, B = the second
// largest.  With an odd number of levels the first step of every other item lands in B, so odd items visit
// their levels in the order (1, 0, 2, ...) instead of (0, 1, 2, ...) -- the accumulators do not care.
// Needs L >= 3 (the next item's query list is built two steps before its first sample loads).
#include "config.h"
#include "msda_geometry.h"

#ifdef UNIVS_MSDA_TRACE
// Debug builds only (tools/msda_trace.py): per-workgroup s_memtime stamps (own array: no relocatable device code)
__device__ unsigned long long g_msda_trace2[8192 * 16];
#define T2STAMP(i)                                                                                   \
  do {                                                                                               \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_msda_trace2[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#define T2STAMPC(i)                                                                                  \
  do {                                                                                               \
    if (threadIdx.x == 64 * UNIVS_MSDA_T2_PROD_WAVES && blockIdx.x < 8192) g_msda_trace2[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" __attribute__((visibility("default"))) int univs_msda_trace2_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_msda_trace2), sizeof(unsigned long long) * 16 * n);
}
#else
#define T2STAMP(i)
#define T2STAMPC(i)
#endif

namespace univs {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int T2_THREADS = 1024;
#ifndef UNIVS_MSDA_T2_PROD_WAVES
#define UNIVS_MSDA_T2_PROD_WAVES 6   // measured: 8 -> 175.6 us, 6 -> 169.7 us, 5 -> 194 us (spills)
#endif
constexpr int T2_PROD = 64 * UNIVS_MSDA_T2_PROD_WAVES;   // producer threads (waves 0 .. PROD_WAVES-1)
constexpr int T2_CONS = T2_THREADS - T2_PROD;            // consumer threads
constexpr int T2_QCAP = 192;                       // max queries per tile
constexpr int T2_NSMP = T2_QCAP * 4;               // sample records per level
constexpr int T2_GROUPS = T2_CONS / 16;            // gather groups (consumers)
constexpr int T2_QMAX = (T2_QCAP + T2_GROUPS - 1) / T2_GROUPS;   // queries per group
constexpr int T2_OCTETS = T2_PROD / 8;             // copy octets (producers)
constexpr int T2_WIN_PX = 704;                     // window capacity in pixels (a 30 x 22 window = 660)
constexpr int T2_WR = (T2_WIN_PX + T2_OCTETS - 1) / T2_OCTETS;   // staged 16-B rows per producer lane
constexpr int T2_SR = (T2_NSMP + T2_PROD - 1) / T2_PROD;         // sample records per producer thread
static_assert(T2_QCAP <= T2_PROD, "one query id per producer thread");

struct Tile2Geom {
  int tiles_y, tiles_x;
  int ord[2][UNIVS_MAX_LEVELS];      // level visited at step k of an even / odd item
  int reg[2][UNIVS_MAX_LEVELS];      // window region offset (in float4) of that step
  int ablate;
};

__device__ __forceinline__ v4f fma4w(float s, v4f v, v4f a) {
  const v4f s4 = {s, s, s, s};
  return __builtin_elementwise_fma(s4, v, a);
}

template <int L>
__global__ __launch_bounds__(T2_THREADS) void msda_fwd_tiled2(const float* __restrict__ value, LevelTable lv,
                                                              Tile2Geom tg, const int4* __restrict__ geo,
                                                              const float* __restrict__ loc,
                                                              const float* __restrict__ attn, int N, int S, int M,
                                                              float* __restrict__ out, unsigned nitems) {
  constexpr int D = 32, P = 4;
  extern __shared__ __attribute__((aligned(16))) v4f lds[];
  // LDS carve: [records v4f x NSMP, two buffers][query ids int x QCAP, two buffers][window regions A | B]
  v4f* recbuf = lds;
  int* qgbuf = reinterpret_cast<int*>(lds + 2 * T2_NSMP);
  v4f* win_lds = lds + 2 * T2_NSMP + 2 * T2_QCAP / 4;

  const int tid = threadIdx.x;
  const bool producer = tid < T2_PROD;             // the first waves (wave-uniform)
  const int ptid = producer ? tid : tid - T2_PROD;   // index inside the role
  const int lane8 = ptid & 7, oct = ptid >> 3;     // producers: 64 octets x 8 lanes
  const int ntiles = tg.tiles_y * tg.tiles_x;

  // ---- this workgroup's items (XCD-chunked, fixed stride; see msda_tiled.hip)
  const unsigned nxcd = min(8u, gridDim.x);
  const unsigned xcd = blockIdx.x % nxcd, widx = blockIdx.x / nxcd;
  const unsigned nw = gridDim.x / nxcd + (xcd < gridDim.x % nxcd ? 1u : 0u);
  const unsigned cq = nitems / nxcd, cr = nitems % nxcd;
  const unsigned cbase = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const unsigned csize = cq + (xcd < cr ? 1u : 0u);
  if (widx >= csize) return;   // uniform, before any barrier

  struct Item {   // workgroup-uniform
    long long nm;   // n * S * M + m
    int tx, ty, total;
  };
  auto make_item = [&](unsigned idx) __attribute__((always_inline)) {
    const unsigned item = cbase + idx;
    const int m = item % M;
    const int tile = (item / M) % ntiles;
    const int n = item / (M * ntiles);
    Item it;
    it.nm = (long long)n * S * M + m;
    it.ty = tile / tg.tiles_x;
    it.tx = tile % tg.tiles_x;
    int tot = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) tot += geo[l * tg.tiles_x + it.tx].y * geo[L * tg.tiles_x + l * tg.tiles_y + it.ty].y;
    it.total = tot;   // 1 .. T2_QCAP (host-checked)
    return it;
  };
  auto geo_x = [&](const Item& it, int l) __attribute__((always_inline)) { return geo[l * tg.tiles_x + it.tx]; };
  auto geo_y = [&](const Item& it, int l) __attribute__((always_inline)) {
    return geo[L * tg.tiles_x + l * tg.tiles_y + it.ty];
  };
  auto fill_qglob = [&](const Item& it, int* qg) __attribute__((always_inline)) {   // producers
    int pre[L + 1];
    int4 gxl[L], gyl[L];
    pre[0] = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      gxl[l] = geo_x(it, l);
      gyl[l] = geo_y(it, l);
      pre[l + 1] = pre[l] + gxl[l].y * gyl[l].y;
    }
    if (ptid < it.total) {
      const int i = ptid;
      int li = i, qx0 = gxl[0].x, qnx = gxl[0].y, qy0 = gyl[0].x, Wq = lv.W[0], st = lv.start[0];
#pragma unroll
      for (int j = 1; j < L; ++j)
        if (i >= pre[j]) { li = i - pre[j]; qx0 = gxl[j].x; qnx = gxl[j].y; qy0 = gyl[j].x; Wq = lv.W[j]; st = lv.start[j]; }
      const int row = (int)(((float)li + 0.5f) * __builtin_amdgcn_rcpf((float)qnx));
      qg[i] = st + (qy0 + row) * Wq + qx0 + (li - row * qnx);
    }
  };

  struct LevelGeo {   // workgroup-uniform
    int l, H, W, wx0, wy0, ww, npx, sx, sy;
    __amdgpu_buffer_rsrc_t rsrc;
  };
  auto level_geo = [&](const Item& it, int l) __attribute__((always_inline)) {
    const int4 gx = geo_x(it, l), gy = geo_y(it, l);
    LevelGeo q;
    q.l = l;
    q.H = lv.H[l]; q.W = lv.W[l];
    q.wx0 = gx.z; q.ww = gx.w; q.wy0 = gy.z; q.npx = gx.w * gy.w;
    q.sy = (int)(((float)T2_OCTETS + 0.5f) * __builtin_amdgcn_rcpf((float)gx.w));
    q.sx = T2_OCTETS - q.sy * gx.w;
    q.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(value + (it.nm + (long long)lv.start[l] * M) * D), 0,
                                               (int)(((long long)q.H * q.W - 1) * M * D * 4 + D * 4), 0x00020000);
    return q;
  };
  // ---- prologue: producers build step 0 completely and stage step 1
  const Item first = make_item(widx);
  T2STAMP(13);
  if (producer) fill_qglob(first, qgbuf);
  __syncthreads();

  // Two separate loops (not one loop with a role branch inside): register allocation is per loop, so the
  // producers' staging registers and the consumers' accumulators / gather temporaries do not add up.  Both
  // loops execute exactly one barrier per step.
  if (producer) {
    // =========================== producers ===========================
    v4f wreg[T2_WR];
    float2 sxy[T2_SR];
    float sa[T2_SR];
    auto load_windows = [&](const LevelGeo& q) __attribute__((always_inline)) {
      if (tg.ablate & 1) return;
      const unsigned pstride = (unsigned)(M * D * 4);
      int ry = (int)(((float)oct + 0.5f) * __builtin_amdgcn_rcpf((float)q.ww));
      int rx = oct - ry * q.ww;
      unsigned off = (unsigned)((q.wy0 + ry) * q.W + q.wx0 + rx) * pstride + (unsigned)lane8 * 16u;
      const unsigned step_n = (unsigned)(q.sy * q.W + q.sx) * pstride;
      const unsigned step_c = (unsigned)((q.sy + 1) * q.W + q.sx - q.ww) * pstride;
#pragma unroll
      for (int u = 0; u < T2_WR; ++u) {
        if (u * T2_OCTETS < q.npx) {   // uniform
          const bool xin = (unsigned)(q.wx0 + rx) < (unsigned)q.W;
          wreg[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(q.rsrc, xin ? off : 0x80000000u, 0, 0));
          rx += q.sx;
          const bool carry = rx >= q.ww;
          rx -= carry ? q.ww : 0;
          off += carry ? step_c : step_n;
        }
      }
    };
    auto load_samples = [&](const Item& it, const int* qg, int l) __attribute__((always_inline)) {
#pragma unroll
      for (int s = 0; s < T2_SR; ++s) {
        const int i = min(ptid + s * T2_PROD, it.total * 4 - 1);
        const unsigned e = (unsigned)((qg[i >> 2] * M * L + l) * P + (i & 3));
        sxy[s] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(loc + it.nm * (L * P * 2)) + e * 8u);
        sa[s] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(attn + it.nm * (L * P)) + e * 4u);
      }
    };
    auto commit = [&](const LevelGeo& q, int total, v4f* win, v4f* rec) __attribute__((always_inline)) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): explicit and unconditional, see msda_tiled.hip
#pragma unroll
      for (int u = 0; u < T2_WR; ++u) {
        if (u * T2_OCTETS < q.npx) {   // uniform
          const int j = oct + u * T2_OCTETS;
          if (j < q.npx) win[j * 8 + lane8] = wreg[u];
        }
      }
      const int wh = q.npx / max(q.ww, 1);
#pragma unroll
      for (int s = 0; s < T2_SR; ++s) {
        const int i = ptid + s * T2_PROD;
        if (i < total * 4) {
          // reference: ms_deform_attn_cuda.cuh:285-293 (h_im, w_im, the (-1, H) x (-1, W) band)
          const float him = sxy[s].y * (float)q.H - 0.5f, wim = sxy[s].x * (float)q.W - 0.5f;
          const bool inimg = him > -1.f && wim > -1.f && him < (float)q.H && wim < (float)q.W;
          const float hf = floorf(him), wf = floorf(wim);
          const float lh = him - hf, lw = wim - wf;
          const int r0 = (int)hf - q.wy0, c0 = (int)wf - q.wx0;
          const bool inwin = (unsigned)r0 < (unsigned)(wh - 1) && (unsigned)c0 < (unsigned)(q.ww - 1);
          const bool use = inimg && inwin;
          const bool miss = inimg && !inwin && sa[s] != 0.f;
          const int slot = (use ? (r0 * q.ww + c0) * 8 : 0) | (miss ? (int)0x80000000 : 0);
          rec[i] = (v4f){__int_as_float(slot), use ? sa[s] * (1.f - lh) : 0.f, use ? sa[s] * lh : 0.f, use ? lw : 0.f};
        }
      }
    };

    Item cur = first;
    {
      const LevelGeo g0 = level_geo(cur, tg.ord[0][0]);
      load_windows(g0);
      load_samples(cur, qgbuf, g0.l);
      commit(g0, cur.total, win_lds + tg.reg[0][0], recbuf);
    }
    LevelGeo geo_p = level_geo(cur, tg.ord[0][1]);          // the step held staged in registers
    load_windows(geo_p);
    load_samples(cur, qgbuf, geo_p.l);
    __syncthreads();   // step 0 is ready for the consumers

    int par = 0;
    unsigned step = 0;
#pragma unroll 1
    for (unsigned idx = widx, itn = 0;; idx += nw, ++itn) {
      if (itn == 1) T2STAMP(0);
      const bool has_next = idx + nw < csize;
      const Item nxt_item = make_item(has_next ? idx + nw : idx);
      const int* qg = qgbuf + par * T2_QCAP;
      int* qg_next = qgbuf + (par ^ 1) * T2_QCAP;
#pragma unroll
      for (int k = 0; k < L; ++k, ++step) {
        // while the consumers gather step (cur, k): commit step + 1 (staged) and load step + 2
        const bool valid1 = (k + 1 < L) || has_next, valid2 = (k + 2 < L) || has_next;
        Item it1, it2;
        if (k + 1 < L) it1 = cur; else it1 = nxt_item;
        if (k + 2 < L) it2 = cur; else it2 = nxt_item;
        const int par1 = (k + 1 < L) ? par : par ^ 1, k1 = (k + 1) % L;
        const int par2 = (k + 2 < L) ? par : par ^ 1, k2 = (k + 2) % L;
        const int* qg2 = (k + 2 < L) ? qg : qg_next;
        const LevelGeo geo_n = level_geo(it2, tg.ord[par2][k2]);
        if (itn == 1) T2STAMP(3 + 3 * k);
        if (valid1) commit(geo_p, it1.total, win_lds + tg.reg[par1][k1], recbuf + ((step + 1) & 1) * T2_NSMP);
        if (k == 0 && has_next) fill_qglob(nxt_item, qg_next);   // read two steps later (L >= 3)
        if (valid2) {
          load_windows(geo_n);
          load_samples(it2, qg2, geo_n.l);
        }
        if (itn == 1) T2STAMP(4 + 3 * k);
        __syncthreads();   // the one barrier of the step
        if (itn == 1) T2STAMP(5 + 3 * k);
        geo_p = geo_n;
      }
      if (itn == 1) T2STAMP(15);
      if (!has_next) break;
      cur = nxt_item;
      par ^= 1;
    }
  } else {
    // =========================== consumers ===========================
    // 16-lane gather groups = the ds_read_b128 hardware lane groups (see msda_tiled.hip)
    const int lane = tid & 63, hl = lane & 31;
    const unsigned long long postab = hl < 16 ? 0x7654765432103210ull : 0xFEDCFEDCBA98BA98ull;
    const int pos = (int)((postab >> ((hl & 15) * 4)) & 15);
    const int g = (0xF00F0FF0u >> hl) & 1;
    const int grp = (ptid >> 6) * 4 + (lane >> 5) * 2 + g;       // 0 .. T2_GROUPS-1
    const int side = pos >> 3, chunk = pos & 7;
    const float xw_c0 = side ? 0.f : 1.f, xw_c1 = side ? 1.f : -1.f;
    const int ppos = pos ^ 8;
    const int phl = g ? (ppos < 8 ? ppos + 4 : ppos < 12 ? ppos + 8 : ppos + 16)
                      : (ppos < 4 ? ppos : ppos < 8 ? ppos + 8 : ppos + 12);
    const int partner = (lane & 32) | phl;

    Item cur = first;
    __syncthreads();   // step 0 is ready
    int par = 0;
    unsigned step = 0;
#pragma unroll 1
    for (unsigned idx = widx, itn = 0;; idx += nw, ++itn) {
      if (itn == 1) T2STAMPC(1);
      const bool has_next = idx + nw < csize;
      const int* qg = qgbuf + par * T2_QCAP;
      v4f acc[T2_QMAX];
#pragma unroll
      for (int k = 0; k < T2_QMAX; ++k) acc[k] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < L; ++k, ++step) {
        // gathers of step (cur, k): 16 lanes per query, two 256-B spans per sample
        const int l = tg.ord[par][k];
        const int H = lv.H[l], W = lv.W[l];
        const int rowstride = geo_x(cur, l).w * 8;
        const v4f* rec = recbuf + (step & 1) * T2_NSMP;
        const v4f* wl = win_lds + tg.reg[par][k] + pos;
#pragma unroll
        for (int kq = 0; kq < T2_QMAX; ++kq) {
          const int qi = grp + kq * T2_GROUPS;
          if (qi < cur.total && !(tg.ablate & 4)) {
            v4f r[4], t[4], b[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) r[p] = rec[qi * 4 + p];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const v4f* base = wl + (__float_as_int(r[p].x) & 0x7fffffff);
              t[p] = base[0];
              b[p] = base[rowstride];
            }
            v4f a = acc[kq];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const float xw = fmaf(xw_c1, r[p].w, xw_c0);
              a = fma4w(r[p].y * xw, t[p], a);
              a = fma4w(r[p].z * xw, b[p], a);
            }
            const int s0 = __float_as_int(r[0].x), s1 = __float_as_int(r[1].x), s2 = __float_as_int(r[2].x),
                      s3 = __float_as_int(r[3].x);
            if ((s0 | s1 | s2 | s3) < 0 && !(tg.ablate & 8)) {
              // rare: footprint(s) outside the staged window -> straight from global memory
              const long long e = ((long long)qg[qi] * M * L + l) * P;
              const v4f* vl = reinterpret_cast<const v4f*>(value + cur.nm * D) + (long long)lv.start[l] * (M * (D / 4)) + chunk;
#pragma unroll 1
              for (int p = 0; p < 4; ++p) {
                const int slp = (p == 0) ? s0 : (p == 1) ? s1 : (p == 2) ? s2 : s3;
                if (slp < 0) {
                  const float2 xy = reinterpret_cast<const float2*>(loc + cur.nm * (L * P * 2))[e + p];
                  const Footprint f = footprint(H, W, xy.x, xy.y, (attn + cur.nm * (L * P))[e + p]);
                  const int wc = side ? f.w1 : f.w0;
                  const v4f g0 = vl[(long long)(f.h0 * W + wc) * (M * (D / 4))];
                  const v4f g1 = vl[(long long)(f.h1 * W + wc) * (M * (D / 4))];
                  a = fma4w(side ? f.w01 : f.w00, g0, a);
                  a = fma4w(side ? f.w11 : f.w10, g1, a);
                }
              }
            }
            acc[kq] = a;
          }
        }
        if (k == L - 1 && !(tg.ablate & 16)) {
          // add the two corner columns (lane <-> partner lane) and store; the left lane writes the row
#pragma unroll
          for (int kq = 0; kq < T2_QMAX; ++kq) {
            v4f o = acc[kq];
            o.x += __shfl(acc[kq].x, partner, 64);
            o.y += __shfl(acc[kq].y, partner, 64);
            o.z += __shfl(acc[kq].z, partner, 64);
            o.w += __shfl(acc[kq].w, partner, 64);
            const int qi = grp + kq * T2_GROUPS;
            if (qi < cur.total && side == 0)
              *reinterpret_cast<v4f*>(reinterpret_cast<char*>(out + cur.nm * D) +
                                      ((unsigned)(qg[qi] * M * D) * 4u + (unsigned)chunk * 16u)) = o;
          }
        }
        if (itn == 1 && k == 0) T2STAMPC(2);
        if (itn == 1 && k == 1) T2STAMPC(9);
        __syncthreads();   // the one barrier of the step
        if (itn == 1 && k == 0) T2STAMPC(6);
        if (itn == 1 && k == 1) T2STAMPC(12);
      }
      if (!has_next) break;
      cur = make_item(idx + nw);
      par ^= 1;
    }
  }
  T2STAMP(14);
}

template <int L>
static void launch_tiled2(unsigned grid, unsigned nitems, size_t lds, hipStream_t st, const float* value,
                          const LevelTable& lv, const Tile2Geom& tg, const int4* geo, const float* loc,
                          const float* attn, int N, int S, int M, float* out) {
  auto k = msda_fwd_tiled2<L>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(grid), dim3(T2_THREADS), lds, st, value, lv, tg, geo, loc, attn, N, S, M, out, nitems);
}

// returns 1 if launched, 0 if preconditions do not hold (caller tries the next implementation), <0 on error
int msda_forward_tiled2_f32(const float* value, const LevelTable& lv, const float* loc, const float* attn, int N,
                            int S, int M, int D, int L, int Lq, int P, float* out, hipStream_t st) {
  if (D != 32 || P != 4 || L < 3 || L > 4 || Lq != S || M < 1) return 0;
  if ((long long)S * M * D * 4 >= (1LL << 31) || (long long)S * M * L * P * 8 >= (1LL << 31)) return 0;
  long long expect = 0;
  int fine = 0;
  for (int l = 0; l < L; ++l) {
    if (lv.start[l] != expect || lv.H[l] < 2 || lv.W[l] < 2) return 0;
    expect += (long long)lv.H[l] * lv.W[l];
    if ((long long)lv.H[l] * lv.W[l] > (long long)lv.H[fine] * lv.W[fine]) fine = l;
  }
  if (expect != S) return 0;

  const UnivsConfig cfg = config();
  const int TH = 8, TW = 16;
  const int R = cfg.msda_halo > 0 ? cfg.msda_halo : 6;
  if (TH < 1 || TW < 1 || R < 0 || R > 64) return 0;
  const long long cap_px = std::min<long long>(T2_WIN_PX, (long long)T2_WR * T2_OCTETS);
  const std::shared_ptr<GeoEntry> ge = geometry(lv, L, fine, TH, TW, R, cap_px, st);
  if (!ge || ge->qmax > T2_QCAP) return 0;

  // region plan: step parity picks the region; an odd level count makes odd items start in B, so they
  // visit their two largest windows in swapped order
  Tile2Geom tg{};
  tg.tiles_y = ge->tiles_y;
  tg.tiles_x = ge->tiles_x;
  tg.ablate = 0;
  int by_size[UNIVS_MAX_LEVELS];
  for (int l = 0; l < L; ++l) by_size[l] = l;
  std::sort(by_size, by_size + L, [&](int a, int b) { return ge->lvl_px[a] > ge->lvl_px[b]; });
  long long capA = 0, capB = 0;
  for (int par = 0; par < 2; ++par) {
    for (int k = 0; k < L; ++k) tg.ord[par][k] = by_size[k];
    const int first_region = (par == 1 && (L & 1)) ? 1 : 0;
    if (first_region == 1) std::swap(tg.ord[par][0], tg.ord[par][1]);
    for (int k = 0; k < L; ++k) {
      const int region = (first_region + k) & 1;
      long long& cap = region ? capB : capA;
      cap = std::max(cap, ge->lvl_px[tg.ord[par][k]]);
      tg.reg[par][k] = region;   // resolved to an offset below
    }
  }
  const size_t fixed = (size_t)2 * T2_NSMP * 16 + (size_t)2 * T2_QCAP * 4;
  if (fixed + (size_t)(capA + capB) * 128 > 160 * 1024) return 0;
  for (int par = 0; par < 2; ++par)
    for (int k = 0; k < L; ++k) tg.reg[par][k] = tg.reg[par][k] ? (int)(capA * 8) : 0;
  const size_t lds = fixed + (size_t)(capA + capB) * 128;

  const long long nb = (long long)N * M * tg.tiles_y * tg.tiles_x;
  if (nb <= 0 || nb > 0x7fffffffLL) return 0;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  const unsigned grid = (unsigned)std::min<long long>(nb, std::max(cfg.msda_grid > 0 ? cfg.msda_grid : n_cu, 1));
  switch (L) {
    case 3: launch_tiled2<3>(grid, (unsigned)nb, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
    default: launch_tiled2<4>(grid, (unsigned)nb, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
  }
  int rc = check_launch("msda_fwd_tiled2");
  geo_mark_use(ge, st);
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
