// LDS-tiled multi-scale deformable attention forward (gfx950) for the pixel-decoder encoder geometry.
//
// Why: the direct-gather kernel touches 8 heads x 12 points x 4 corners x 128 B = 49 KB per query
// through the per-CU vector L1 (64 B/clk) against 3.2 KB/query of algorithmic traffic (SURVEY.md
// section 7 "hard parts"), and every lane that shares a 128-B row repeats the same bilinear address
// arithmetic.  Here the gathers go to LDS (256 B/clk/CU for ds_read_b128) and the footprint
// arithmetic is done ONCE per sample.
//
// Preconditions (else the caller falls back to the generic kernel): Lq == S (every pixel of every
// level is a query, in level-major raster order -- the encoder self-attention of
// msdeformattn.py:61-89), D == 32 (one 128-B row per pixel-head), P == 4, L <= 4, levels >= 2x2.
//
// Decomposition: one workgroup = (frame n, head m, spatial tile).  A tile is a TH x TW block of the
// finest level; it owns every query of EVERY level whose centre falls inside the tile's normalised
// box (256 + 64 + 16 queries for a 2x pyramid and a 16x16 tile).  All tile geometry (query boxes,
// windows) is computed once per geometry on the host and read from a small device table with scalar
// loads.  For each value level in turn:
//   commit   (all threads)    the level's window [box * (H_l, W_l) +- R] of this head goes to LDS,
//                             ZERO-PADDED where it sticks out of the image (that is the reference's
//                             zero padding, ms_deform_attn_cuda.cuh:38-89, done once per pixel instead
//                             of once per corner), and, one thread per SAMPLE, (x, y, attention
//                             weight) becomes a 16-byte record {LDS slot of the 2x2 footprint,
//                             aw*(1-lh), aw*lh, lw}.
//   phase B  (16 lanes/query) a 16-lane group reads the two horizontally adjacent corners of a
//                             sample as ONE contiguous 256-B span (pixel p and p+1 are adjacent in
//                             the window) -> every ds_read_b128 covers all 64 banks exactly once,
//                             whatever the sample positions are.  The groups are the hardware's
//                             ds_read_b128 lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32),
//                             so gathers are bank-conflict free by construction.  Each lane keeps a
//                             partial sum for its (left|right) corner column; the two halves are
//                             added once per query at the very end (one ds_bpermute per float).
// Global loads for level l+1 (window + sample inputs) are issued before phase B of level l and
// only written to LDS after it (register-staged pipeline), so their latency hides under the gathers.
// A sample whose footprint is not fully inside the staged window (|offset| > R) is flagged in its
// record and taken straight from global memory -- results never depend on R or on the tiling.
//
// Block order: logical id = ((n * tiles + tile) * M + m), XCD-chunked, so the 8 head-workgroups
// of a tile (which share the 128-B lines of sampling_loc / attn_weight) and neighbouring tiles
// (which share halo rows) run on the same XCD L2.
#include "msda_geometry.h"

#ifdef UNIVS_MSDA_TRACE
// Debug builds only (tools/msda_trace.py): per-workgroup s_memtime stamps of the kernel's phases.
__device__ unsigned long long g_msda_trace[8192 * 16];
#define TSTAMP(i)                                                                             \
  do {                                                                                        \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_msda_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
extern "C" __attribute__((visibility("default"))) int univs_msda_trace_read(unsigned long long* dst, int n) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_msda_trace), sizeof(unsigned long long) * 16 * n);
}
#else
#define TSTAMP(i)
#endif

namespace univs {

typedef float v4f __attribute__((ext_vector_type(4)));  // native vector: plain loads/stores, no memcpy

struct TileGeom {
  int tiles_y, tiles_x;
  int ablate;    // profiling only (UNIVS_MSDA_ABLATE): bit0 skip window copy, bit1 skip records, bit2 skip gathers, bit3 skip the global fallback
};

constexpr int TL_THREADS = 512;
constexpr int TL_QCAP = 384;                      // max queries per tile
constexpr int TL_NSMP = TL_QCAP * 4;              // sample records per level
constexpr int TL_GROUPS = TL_THREADS / 16;        // 16-lane gather groups per workgroup
constexpr int TL_QMAX = TL_QCAP / TL_GROUPS;      // queries per gather group
constexpr int TL_OCTETS = TL_THREADS / 8;         // 8-lane copy groups (one 128-B row each)
constexpr int TL_WR = 15;                         // window float4 per lane: 960 pixels (a 30 x 30 window = 16 + 2 x 6 + 2)
constexpr int TL_SR = TL_NSMP / TL_THREADS;       // samples per thread


__device__ __forceinline__ v4f fma4v(float s, v4f v, v4f a) {
  const v4f s4 = {s, s, s, s};
  return __builtin_elementwise_fma(s4, v, a);
}

// Geometry table (device memory, built by the host once per geometry):
//   geo[l * tiles_x + tx]                  = {qx0, qnx, wx0, ww}   query columns / window columns
//   geo[L * tiles_x + l * tiles_y + ty]    = {qy0, qny, wy0, wh}   query rows    / window rows
// Window coordinates are image coordinates and may start at -1 / end at H (W): the zero ring.
// Every tile owns at least one query (host-checked).
//
// Persistent: the grid is a multiple of the CU count, each workgroup walks a strided list of items
// (frame, tile, head) inside its XCD's chunk of the item space, and the loads of the NEXT item's first
// level are issued during the gathers of the current item's last level, so neither the launch gap nor
// the first window's latency is paid per item.
template <int L>
__global__ __launch_bounds__(TL_THREADS) void msda_fwd_tiled(const float* __restrict__ value,
                                                              LevelTable lv, TileGeom tg,
                                                              const int4* __restrict__ geo,
                                                              const float* __restrict__ loc,
                                                              const float* __restrict__ attn, int N, int S,
                                                              int M, float* __restrict__ out,
                                                              unsigned nitems) {
  constexpr int D = 32, P = 4;
  extern __shared__ __attribute__((aligned(16))) v4f lds[];
  // LDS carve: [records v4f x NSMP][query ids int x QCAP, two buffers][window]
  v4f* rec = lds;
  int* qgbuf = reinterpret_cast<int*>(lds + TL_NSMP);
  v4f* win_lds = lds + TL_NSMP + 2 * TL_QCAP / 4;

  const int tid = threadIdx.x, lane8 = tid & 7, oct = tid >> 3;
  const int rowf4 = M * (D / 4);
  const int ntiles = tg.tiles_y * tg.tiles_x;

  // ---- 16-lane gather groups = the ds_read_b128 hardware lane groups
  const int lane = tid & 63, hl = lane & 31;
  const unsigned long long postab = hl < 16 ? 0x7654765432103210ull : 0xFEDCFEDCBA98BA98ull;
  const int pos = (int)((postab >> ((hl & 15) * 4)) & 15);   // position in the 256-B span
  const int g = (0xF00F0FF0u >> hl) & 1;
  const int grp = (tid >> 6) * 4 + (lane >> 5) * 2 + g;        // 0 .. TL_GROUPS-1
  const int side = pos >> 3, chunk = pos & 7;                  // corner column (0 left, 1 right), 16-B chunk
  const float xw_c0 = side ? 0.f : 1.f, xw_c1 = side ? 1.f : -1.f;  // column weight = c0 + c1 * lw
  const int ppos = pos ^ 8;                                    // partner lane: same query, other column
  const int phl = g ? (ppos < 8 ? ppos + 4 : ppos < 12 ? ppos + 8 : ppos + 16)
                    : (ppos < 4 ? ppos : ppos < 8 ? ppos + 8 : ppos + 12);
  const int partner = (lane & 32) | phl;

  // ---- this workgroup's items: XCD x (= blockIdx % 8 in hardware dispatch order) owns a contiguous
  // chunk of the item space; its workgroups take every nw-th item of the chunk, so at any time the
  // workgroups of an XCD work on the heads of a few neighbouring tiles (shared halo rows and sample
  // lines in that XCD's L2).  (A ticket counter per XCD instead of the fixed stride was measured: no
  // gain -- the makespan is set by ceil(items / workgroups), not by variance.)
  const unsigned nxcd = min(8u, gridDim.x);
  const unsigned xcd = blockIdx.x % nxcd, widx = blockIdx.x / nxcd;
  const unsigned nw = gridDim.x / nxcd + (xcd < gridDim.x % nxcd ? 1u : 0u);
  const unsigned cq = nitems / nxcd, cr = nitems % nxcd;
  const unsigned cbase = xcd < cr ? xcd * (cq + 1) : cr * (cq + 1) + (xcd - cr) * cq;
  const unsigned csize = cq + (xcd < cr ? 1u : 0u);
  if (widx >= csize) return;   // uniform, before any barrier

  // workgroup-uniform item state, kept small on purpose (it lives in SGPRs twice: current + next):
  // element offset of (frame n, query 0, head m) and the tile's geometry rows
  struct Item {
    long long nm;   // n * S * M + m
    int tx, ty, total;
  };
  struct Coord { int n, m, tx, ty; };   // workgroup-uniform item coordinates
  auto make_item = [&](const Coord& c) __attribute__((always_inline)) {
    Item it;
    it.nm = (long long)c.n * S * M + c.m;
    it.tx = c.tx;
    it.ty = c.ty;
    int tot = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) tot += geo[l * tg.tiles_x + c.tx].y * geo[L * tg.tiles_x + l * tg.tiles_y + c.ty].y;
    it.total = tot;   // 1 .. TL_QCAP (host-checked)
    return it;
  };
  auto geo_x = [&](const Item& it, int l) __attribute__((always_inline)) { return geo[l * tg.tiles_x + it.tx]; };
  auto geo_y = [&](const Item& it, int l) __attribute__((always_inline)) {
    return geo[L * tg.tiles_x + l * tg.tiles_y + it.ty];
  };
  auto decode = [&](unsigned idx) __attribute__((always_inline)) {   // chunk-relative index -> coordinates
    const unsigned item = cbase + idx;
    Coord c;
    c.m = item % M;
    const int tile = (item / M) % ntiles;
    c.n = item / (M * ntiles);
    c.ty = tile / tg.tiles_x;
    c.tx = tile % tg.tiles_x;
    return c;
  };
  // global query index of every query of the item's tile
  auto fill_qglob = [&](const Item& it, int* qg) __attribute__((always_inline)) {
    int pre[L + 1];
    int4 gxl[L], gyl[L];   // scalar loads, selected per thread below (no per-thread global load)
    pre[0] = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      gxl[l] = geo_x(it, l);
      gyl[l] = geo_y(it, l);
      pre[l + 1] = pre[l] + gxl[l].y * gyl[l].y;
    }
    for (int i = tid; i < it.total; i += TL_THREADS) {
      int li = i, qx0 = gxl[0].x, qnx = gxl[0].y, qy0 = gyl[0].x, Wq = lv.W[0], st = lv.start[0];
#pragma unroll
      for (int j = 1; j < L; ++j)
        if (i >= pre[j]) { li = i - pre[j]; qx0 = gxl[j].x; qnx = gxl[j].y; qy0 = gyl[j].x; Wq = lv.W[j]; st = lv.start[j]; }
      const int row = (int)(((float)li + 0.5f) * __builtin_amdgcn_rcpf((float)qnx));   // exact: li < 2^10, margin 0.5/nx
      qg[i] = st + (qy0 + row) * Wq + qx0 + (li - row * qnx);
    }
  };

  // Register-staged software pipeline (issue early / write late): the global loads of the next
  // window and sample inputs are issued right before a gather loop and only written to LDS after it.
  struct LevelGeo {   // workgroup-uniform
    int H, W, wx0, wy0, ww, npx;
    int sx, sy;                    // 64 = sy * ww + sx: window-row / column step of one copy step
    __amdgpu_buffer_rsrc_t rsrc;   // this (frame, head, level)'s value rows: out-of-range offsets read 0
  };
  auto level_geo = [&](const Item& it, int l) __attribute__((always_inline)) {
    const int4 gx = geo_x(it, l), gy = geo_y(it, l);
    LevelGeo q;
    q.H = lv.H[l]; q.W = lv.W[l];
    q.wx0 = gx.z; q.ww = gx.w; q.wy0 = gy.z; q.npx = gx.w * gy.w;
    q.sy = (int)(((float)TL_OCTETS + 0.5f) * __builtin_amdgcn_rcpf((float)gx.w));   // exact, see below
    q.sx = TL_OCTETS - q.sy * gx.w;
    // rows of head m only: the last valid byte is the end of the last pixel's 128-B row
    q.rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(value + (it.nm + (long long)lv.start[l] * M) * D), 0,
        (int)(((long long)q.H * q.W - 1) * M * D * 4 + D * 4), 0x00020000);
    return q;
  };
  v4f wreg[TL_WR];
  float2 sxy[TL_SR];
  float sa[TL_SR];
  // Window copy: pixel j = oct + 64 u (row-major over the window), one 128-B row per octet and step.
  // The zero ring costs nothing: rows above / below the image fall outside the buffer resource (the
  // hardware returns 0), columns left / right of it get an out-of-range offset on purpose.  (ry, rx)
  // and the byte offset advance incrementally -- no division, no clamps.
  auto load_windows = [&](const LevelGeo& q) __attribute__((always_inline)) {
    if (tg.ablate & 1) return;
    const unsigned pstride = (unsigned)(M * D * 4);
    int ry = (int)(((float)oct + 0.5f) * __builtin_amdgcn_rcpf((float)q.ww));   // exact: oct < 64, margin 0.5/ww
    int rx = oct - ry * q.ww;
    unsigned off = (unsigned)((q.wy0 + ry) * q.W + q.wx0 + rx) * pstride + (unsigned)lane8 * 16u;
    const unsigned step_n = (unsigned)(q.sy * q.W + q.sx) * pstride;              // same window row + sy
    const unsigned step_c = (unsigned)((q.sy + 1) * q.W + q.sx - q.ww) * pstride;  // wrapped to the next row
#pragma unroll
    for (int u = 0; u < TL_WR; ++u) {
      if (u * TL_OCTETS < q.npx) {   // uniform
        const bool xin = (unsigned)(q.wx0 + rx) < (unsigned)q.W;
        wreg[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(q.rsrc, xin ? off : 0x80000000u, 0, 0));
        rx += q.sx;
        const bool carry = rx >= q.ww;
        rx -= carry ? q.ww : 0;
        off += carry ? step_c : step_n;
      }
    }
  };
  auto load_sample = [&](const Item& it, const int* qg, int l, int s) __attribute__((always_inline)) {
    if (!(tg.ablate & 2)) {
      const int i = min(tid + s * TL_THREADS, it.total * 4 - 1);
      const unsigned e = (unsigned)((qg[i >> 2] * M * L + l) * P + (i & 3));   // < 2^28 (host-checked)
      sxy[s] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(loc + it.nm * (L * P * 2)) + e * 8u);
      sa[s] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(attn + it.nm * (L * P)) + e * 4u);
    }
  };
  auto commit = [&](const LevelGeo& q, int total) __attribute__((always_inline)) {
    // All staged loads have landed from here on.  Explicit and unconditional on purpose: the loads and
    // their consumers sit in (uniform) conditional blocks, and without this hipcc's waitcnt pass assumes
    // a load of the previous level may still be pending on some path and puts s_waitcnt vmcnt(0) in
    // front of EVERY staged load of the next level (serialising them: 2x slower kernel).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), lgkmcnt/expcnt untouched
#pragma unroll
    for (int u = 0; u < TL_WR; ++u) {
      if (u * TL_OCTETS < q.npx) {   // uniform
        const int j = oct + u * TL_OCTETS;
        if (j < q.npx) win_lds[j * 8 + lane8] = wreg[u];
      }
    }
    const int wh = q.npx / max(q.ww, 1);
    if (!(tg.ablate & 2))
#pragma unroll
    for (int s = 0; s < TL_SR; ++s) {
      const int i = tid + s * TL_THREADS;
      if (i < total * 4) {
        // reference: ms_deform_attn_cuda.cuh:285-293 (h_im, w_im, the (-1, H) x (-1, W) band)
        const float him = sxy[s].y * (float)q.H - 0.5f, wim = sxy[s].x * (float)q.W - 0.5f;
        const bool inimg = him > -1.f && wim > -1.f && him < (float)q.H && wim < (float)q.W;
        const float hf = floorf(him), wf = floorf(wim);
        const float lh = him - hf, lw = wim - wf;
        const int r0 = (int)hf - q.wy0, c0 = (int)wf - q.wx0;   // v_cvt saturates; only used when inimg
        const bool inwin = (unsigned)r0 < (unsigned)(wh - 1) && (unsigned)c0 < (unsigned)(q.ww - 1);
        const bool use = inimg && inwin;
        const bool miss = inimg && !inwin && sa[s] != 0.f;
        const int slot = (use ? (r0 * q.ww + c0) * 8 : 0) | (miss ? (int)0x80000000 : 0);
        // selects, not multiplications by 0: a NaN/inf location must contribute exactly nothing
        rec[i] = (v4f){__int_as_float(slot), use ? sa[s] * (1.f - lh) : 0.f, use ? sa[s] * lh : 0.f,
                       use ? lw : 0.f};
      }
    }
  };

  // ---- prologue: first item's query list and first window
  Item cur = make_item(decode(widx));
  TSTAMP(13);
  fill_qglob(cur, qgbuf);
  __syncthreads();
  LevelGeo geo_cur = level_geo(cur, 0);
  load_windows(geo_cur);
#pragma unroll
  for (int s = 0; s < TL_SR; ++s) load_sample(cur, qgbuf, 0, s);

  int par = 0;
#pragma unroll 1
  for (unsigned idx = widx, itn = 0;; idx += nw, ++itn) {
    if (itn == 1) TSTAMP(0);
    const bool has_next = idx + nw < csize;
    const Item nxt_item = make_item(decode(has_next ? idx + nw : idx));
    const int* qg = qgbuf + par * TL_QCAP;
    int* qg_next = qgbuf + (par ^ 1) * TL_QCAP;

    v4f acc[TL_QMAX];
#pragma unroll
    for (int k = 0; k < TL_QMAX; ++k) acc[k] = (v4f){0.f, 0.f, 0.f, 0.f};

    // levels fully unrolled on purpose: with a rolled level loop hipcc's waitcnt pass serialises the
    // staged loads (s_waitcnt vmcnt(0) in front of each, see commit()); L <= 4, so the code stays small
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int H = geo_cur.H, W = geo_cur.W;
      const int rowstride = geo_cur.ww * 8;   // window row stride in float4
      const v4f* vl = reinterpret_cast<const v4f*>(value + cur.nm * D) + (long long)lv.start[l] * rowf4 + chunk;
      const v4f* wl = win_lds + pos;   // pixel p's chunk `chunk` (+8 float4 = pixel p+1 for the right column)

      if (l > 0 || itn > 0) __syncthreads();  // the previous gather loop is done with window + records
      if (itn == 1) TSTAMP(3 + 3 * l);
      commit(geo_cur, cur.total);
      if (l == L - 1 && has_next) fill_qglob(nxt_item, qg_next);
      // next staged loads (in flight during the gathers below): next level, or the next item's level 0.
      // (Spreading them over the gather steps instead was measured: slower.)  Their geometry is
      // fetched before the barrier so that the scalar loads' latency hides in the barrier wait.
      const bool do_loads = (l + 1 < L) || has_next;
      const Item& ld_item = (l + 1 < L) ? cur : nxt_item;
      const int ld_level = (l + 1 < L) ? l + 1 : 0;
      const int* ld_qg = (l + 1 < L) ? qg : qg_next;
      const LevelGeo geo_nxt = do_loads ? level_geo(ld_item, ld_level) : geo_cur;
      __syncthreads();
      if (itn == 1) TSTAMP(4 + 3 * l);
      if (do_loads) {
        load_windows(geo_nxt);
#pragma unroll
        for (int s = 0; s < TL_SR; ++s) load_sample(ld_item, ld_qg, ld_level, s);
      }

      // ---- gathers: 16 lanes per query; per sample two 256-B spans (top pair, bottom pair)
#pragma unroll
      for (int k = 0; k < TL_QMAX; ++k) {
        const int qi = grp + k * TL_GROUPS;
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
          v4f a = acc[k];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const float xw = fmaf(xw_c1, r[p].w, xw_c0);
            a = fma4v(r[p].y * xw, t[p], a);
            a = fma4v(r[p].z * xw, b[p], a);
          }
          const int s0 = __float_as_int(r[0].x), s1 = __float_as_int(r[1].x), s2 = __float_as_int(r[2].x),
                    s3 = __float_as_int(r[3].x);
          if ((s0 | s1 | s2 | s3) < 0 && !(tg.ablate & 8)) {
            // rare: footprint(s) outside the staged window -> those samples come from global memory,
            // one sample at a time (this lane's corner column: 2 clamped, always-valid loads).  Kept
            // narrow on purpose: this path must not raise the register pressure of the common path.
            const long long e = ((long long)qg[qi] * M * L + l) * P;
#pragma unroll 1
            for (int p = 0; p < 4; ++p) {
              const int slp = (p == 0) ? s0 : (p == 1) ? s1 : (p == 2) ? s2 : s3;
              if (slp < 0) {
                const float2 xy = reinterpret_cast<const float2*>(loc + cur.nm * (L * P * 2))[e + p];
                const Footprint f = footprint(H, W, xy.x, xy.y, (attn + cur.nm * (L * P))[e + p]);
                const int wc = side ? f.w1 : f.w0;
                const v4f g0 = vl[(long long)(f.h0 * W + wc) * rowf4];
                const v4f g1 = vl[(long long)(f.h1 * W + wc) * rowf4];
                a = fma4v(side ? f.w01 : f.w00, g0, a);
                a = fma4v(side ? f.w11 : f.w10, g1, a);
              }
            }
          }
          acc[k] = a;
        }
      }
      geo_cur = geo_nxt;
      if (itn == 1) TSTAMP(5 + 3 * l);
    }

    // ---- add the two corner columns (lane <-> partner lane, pos ^ 8) and store: of each pair of
    // queries (k, k+1) the left lane finishes k and the right lane k+1 -> every lane stores 16 B
    if (!(tg.ablate & 16))
#pragma unroll
    for (int k = 0; k < TL_QMAX; k += 2) {
      const v4f snd = side ? acc[k] : acc[k + 1], own = side ? acc[k + 1] : acc[k];
      v4f o;
      o.x = own.x + __shfl(snd.x, partner, 64);
      o.y = own.y + __shfl(snd.y, partner, 64);
      o.z = own.z + __shfl(snd.z, partner, 64);
      o.w = own.w + __shfl(snd.w, partner, 64);
      const int qi = grp + (k + side) * TL_GROUPS;
      if (qi < cur.total)
        *reinterpret_cast<v4f*>(reinterpret_cast<char*>(out + cur.nm * D) +
                                ((unsigned)(qg[qi] * M * D) * 4u + (unsigned)chunk * 16u)) = o;
    }
    if (itn == 1) TSTAMP(15);
    if (!has_next) break;
    cur = nxt_item;
    par ^= 1;
  }
  TSTAMP(14);
}

template <int L>
static void launch_tiled(unsigned grid, unsigned nblocks, size_t lds, hipStream_t st, const float* value,
                         const LevelTable& lv, const TileGeom& tg, const int4* geo,
                         const float* loc, const float* attn, int N, int S, int M, float* out) {
  auto k = msda_fwd_tiled<L>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(grid), dim3(TL_THREADS), lds, st, value, lv, tg, geo, loc, attn, N, S, M, out, nblocks);
}

// returns 1 if launched, 0 if preconditions do not hold, <0 on error
int msda_forward_tiled_f32(const float* value, const LevelTable& lv, const float* loc,
                           const float* attn, int N, int S, int M, int D, int L, int Lq, int P,
                           float* out, hipStream_t st) {
  if (D != 32 || P != 4 || L < 1 || L > 4 || Lq != S || M < 1) return 0;
  // per-frame byte offsets are kept in 32 bits on the device
  if ((long long)S * M * D * 4 >= (1LL << 31) || (long long)S * M * L * P * 8 >= (1LL << 31)) return 0;
  // levels must tile [0, S) exactly, in order (the encoder's flatten+concat layout)
  long long expect = 0;
  int fine = 0;
  for (int l = 0; l < L; ++l) {
    if (lv.start[l] != expect || lv.H[l] < 2 || lv.W[l] < 2) return 0;
    expect += (long long)lv.H[l] * lv.W[l];
    if ((long long)lv.H[l] * lv.W[l] > (long long)lv.H[fine] * lv.W[fine]) fine = l;
  }
  if (expect != S) return 0;

  const int TH = env_int("UNIVS_MSDA_TILE_H", 16), TW = env_int("UNIVS_MSDA_TILE_W", 16);
  const int R = env_int("UNIVS_MSDA_HALO", 6);
  if (TH < 1 || TW < 1 || R < 0 || R > 64) return 0;
  const size_t fixed = (size_t)TL_NSMP * 16 + (size_t)TL_QCAP * 4 * 2;
  const long long cap_px = std::min<long long>((160 * 1024 - (long long)fixed) / 128, TL_WR * TL_OCTETS);
  const GeoEntry* ge = geometry(lv, L, fine, TH, TW, R, cap_px);
  if (!ge) return 0;
  if (ge->qmax > TL_QCAP) return 0;

  TileGeom tg;
  tg.tiles_y = ge->tiles_y;
  tg.tiles_x = ge->tiles_x;
  tg.ablate = env_int("UNIVS_MSDA_ABLATE", 0);
  const size_t lds = fixed + (size_t)ge->win_px * 128;

  const long long nb = (long long)N * M * tg.tiles_y * tg.tiles_x;
  if (nb <= 0 || nb > 0x7fffffffLL) return 0;
  const unsigned nblocks = (unsigned)nb;
  // persistent grid: one workgroup per CU (LDS allows no more), each walking ~nb / CUs items
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  const int gwant = env_int("UNIVS_MSDA_GRID", n_cu);
  const unsigned grid = (unsigned)std::min<long long>(nb, std::max(gwant, 1));
  switch (L) {
    case 1: launch_tiled<1>(grid, nblocks, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
    case 2: launch_tiled<2>(grid, nblocks, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
    case 3: launch_tiled<3>(grid, nblocks, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
    default: launch_tiled<4>(grid, nblocks, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
  }
  int rc = check_launch("msda_fwd_tiled");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
