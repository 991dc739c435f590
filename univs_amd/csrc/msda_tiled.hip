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
#include <algorithm>
#include <mutex>
#include <vector>

#include "msda_common.h"

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
constexpr int TL_WR = 1024 / TL_OCTETS;           // window float4 per lane (covers 32 x 32 pixels)
constexpr int TL_SR = TL_NSMP / TL_THREADS;       // samples per thread
constexpr int TL_WIN_MAX = 64;                    // window edge limit (the product ww*wh is bounded by the LDS carve)

__device__ __forceinline__ v4f fma4v(float s, v4f v, v4f a) {
  const v4f s4 = {s, s, s, s};
  return __builtin_elementwise_fma(s4, v, a);
}

// Geometry table (device memory, built by the host once per geometry):
//   geo[l * tiles_x + tx]                  = {qx0, qnx, wx0, ww}   query columns / window columns
//   geo[L * tiles_x + l * tiles_y + ty]    = {qy0, qny, wy0, wh}   query rows    / window rows
// Window coordinates are image coordinates and may start at -1 / end at H (W): the zero ring.
template <int L>
__global__ __launch_bounds__(TL_THREADS) void msda_fwd_tiled(const float* __restrict__ value,
                                                              LevelTable lv, TileGeom tg,
                                                              const int4* __restrict__ geo,
                                                              const float* __restrict__ loc,
                                                              const float* __restrict__ attn, int N, int S,
                                                              int M, float* __restrict__ out,
                                                              unsigned nblocks) {
  constexpr int D = 32, P = 4;
  extern __shared__ __attribute__((aligned(16))) v4f lds[];
  // LDS carve: [records v4f x NSMP][query ids int x QCAP][window]
  v4f* rec = lds;
  int* qglob = reinterpret_cast<int*>(lds + TL_NSMP);
  v4f* win_lds = lds + TL_NSMP + TL_QCAP / 4;

  const unsigned bid = xcd_remap(blockIdx.x, nblocks);
  const int m = bid % M;
  const int ntiles = tg.tiles_y * tg.tiles_x;
  const int tile = (bid / M) % ntiles;
  const int n = bid / (M * ntiles);
  const int ty = tile / tg.tiles_x, tx = tile % tg.tiles_x;
  const int tid = threadIdx.x, lane8 = tid & 7, oct = tid >> 3;
  const int rowf4 = M * (D / 4);
  const int4* geox = geo + tx;
  const int4* geoy = geo + L * tg.tiles_x + ty;

  // ---- 16-lane gather groups = the ds_read_b128 hardware lane groups
  const int lane = tid & 63, hl = lane & 31;
  const unsigned long long postab = hl < 16 ? 0x7654765432103210ull : 0xFEDCFEDCBA98BA98ull;
  const int pos = (int)((postab >> ((hl & 15) * 4)) & 15);   // position in the 256-B span
  const int g = (0xF00F0FF0u >> hl) & 1;
  const int grp = (tid >> 6) * 4 + (lane >> 5) * 2 + g;        // 0 .. TL_GROUPS-1
  const int side = pos >> 3, chunk = pos & 7;                  // corner column (0 left, 1 right), 16-B chunk
  const float xw_c0 = side ? 0.f : 1.f, xw_c1 = side ? 1.f : -1.f;  // column weight = c0 + c1 * lw

  // ---- queries owned by this tile
  int pre[L + 1];
  pre[0] = 0;
#pragma unroll
  for (int l = 0; l < L; ++l) pre[l + 1] = pre[l] + geox[l * tg.tiles_x].y * geoy[l * tg.tiles_y].y;
  const int total = pre[L];  // <= TL_QCAP (host-checked)
  // Workgroup-uniform early exit (no barrier has been passed yet).  Also keeps every later load /
  // commit pair on ONE control-flow path: with separate `if (total > 0)` guards hipcc's waitcnt pass
  // sees an (infeasible) path "loads issued, commit skipped" and serialises the next level's loads.
  if (total == 0) return;
  TSTAMP(0);

  // global query index of every query of the tile, once
  for (int i = tid; i < total; i += TL_THREADS) {
    int l = 0;
#pragma unroll
    for (int j = 1; j < L; ++j) l += (i >= pre[j]) ? 1 : 0;
    const int4 gx = geox[l * tg.tiles_x], gy = geoy[l * tg.tiles_y];
    int li = i;
#pragma unroll
    for (int j = 1; j < L; ++j)
      if (l == j) li = i - pre[j];
    const int row = (int)(((float)li + 0.5f) * __builtin_amdgcn_rcpf((float)gx.y));   // exact: li < 2^10, margin 0.5/nx
    qglob[i] = lv.start[l] + (gy.x + row) * lv.W[l] + gx.x + (li - row * gx.y);
  }

  v4f acc[TL_QMAX];
#pragma unroll
  for (int k = 0; k < TL_QMAX; ++k) acc[k] = (v4f){0.f, 0.f, 0.f, 0.f};

  const v4f* vn = reinterpret_cast<const v4f*>(value + (long long)n * S * M * D + (long long)m * D);
  const float* locn = loc + ((long long)n * S * M + m) * (L * P * 2);
  const float* attn_n = attn + ((long long)n * S * M + m) * (L * P);

  // Register-staged software pipeline (issue early / write late): the global loads of level l+1's
  // window and sample inputs are issued right before the gather loop of level l and only written to
  // LDS after it.  (Spreading them over the gather steps was tried: hipcc then puts s_waitcnt vmcnt(0)
  // in front of every load and serialises them.)
  struct LevelGeo {   // workgroup-uniform
    int H, W, wx0, wy0, ww, npx;
    float rcp_ww;
    const v4f* src;
  };
  auto level_geo = [&](int l) __attribute__((always_inline)) {
    const int4 gx = geox[l * tg.tiles_x], gy = geoy[l * tg.tiles_y];
    LevelGeo q;
    q.H = lv.H[l]; q.W = lv.W[l];
    q.wx0 = gx.z; q.ww = gx.w; q.wy0 = gy.z; q.npx = gx.w * gy.w;
    q.rcp_ww = __builtin_amdgcn_rcpf((float)gx.w);   // 1 ulp is plenty, see load_window
    q.src = vn + (long long)lv.start[l] * rowf4 + lane8;
    return q;
  };
  v4f wreg[TL_WR];
  unsigned vmask = 0;   // bit u: wreg[u] is an image pixel (else: zero ring)
  float2 sxy[TL_SR];
  float sa[TL_SR];
  // window pixel j = oct + 64 u (row-major over the window) -> one 128-B row per octet
  auto load_window = [&](const LevelGeo& q, int u) __attribute__((always_inline)) {
    if (u * TL_OCTETS < q.npx && !(tg.ablate & 1)) {   // uniform
      const int j = min(oct + u * TL_OCTETS, q.npx - 1);
      const int ry = (int)(((float)j + 0.5f) * q.rcp_ww);   // exact: j < 2^10, margin 0.5/ww
      const int px = q.wx0 + (j - ry * q.ww), py = q.wy0 + ry;
      const bool in = (unsigned)px < (unsigned)q.W && (unsigned)py < (unsigned)q.H;
      vmask = in ? (vmask | (1u << u)) : (vmask & ~(1u << u));
      wreg[u] = q.src[((long long)min(max(py, 0), q.H - 1) * q.W + min(max(px, 0), q.W - 1)) * rowf4];
    }
  };
  auto load_sample = [&](int l, int s) __attribute__((always_inline)) {
    if (!(tg.ablate & 2)) {
      const int i = min(tid + s * TL_THREADS, total * 4 - 1);
      const long long e = ((long long)qglob[i >> 2] * M * L + l) * P + (i & 3);
      sxy[s] = reinterpret_cast<const float2*>(locn)[e];
      sa[s] = attn_n[e];
    }
  };
  auto commit = [&](const LevelGeo& q) __attribute__((always_inline)) {
    // All staged loads have landed from here on.  Explicit and unconditional on purpose: the loads and
    // their consumers sit in (uniform) conditional blocks, and without this hipcc's waitcnt pass assumes
    // a load of the previous level may still be pending on some path and puts s_waitcnt vmcnt(0) in
    // front of EVERY staged load of the next level (serialising them: 2x slower kernel).
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), lgkmcnt/expcnt untouched
#pragma unroll
    for (int u = 0; u < TL_WR; ++u) {
      if (u * TL_OCTETS < q.npx) {   // uniform
        const int j = oct + u * TL_OCTETS;
        if (j < q.npx) win_lds[j * 8 + lane8] = ((vmask >> u) & 1) ? wreg[u] : (v4f){0.f, 0.f, 0.f, 0.f};
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

  __syncthreads();  // qglob visible
  TSTAMP(1);
  LevelGeo cur = level_geo(0);
#pragma unroll
  for (int u = 0; u < TL_WR; ++u) load_window(cur, u);
#pragma unroll
  for (int s = 0; s < TL_SR; ++s) load_sample(0, s);
  TSTAMP(2);

  // fully unrolled on purpose: with a rolled level loop hipcc's waitcnt pass serialises the staged
  // loads (s_waitcnt vmcnt(0) in front of each, see commit()); L <= 4, so the code stays small
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const int H = cur.H, W = cur.W;
    const int rowstride = cur.ww * 8;   // window row stride in float4
    const v4f* vl = vn + (long long)lv.start[l] * rowf4 + chunk;
    const v4f* wl = win_lds + pos;   // pixel p's chunk `chunk` (+8 float4 = pixel p+1 for the right column)

    if (l > 0) __syncthreads();  // previous level's phase B is done with the window and the records
    TSTAMP(3 + 3 * l);
    commit(cur);
    __syncthreads();
    TSTAMP(4 + 3 * l);
    const bool more = l + 1 < L;
    const LevelGeo nxt = level_geo(min(l + 1, L - 1));

    if (more) {   // in flight during phase B below
#pragma unroll
      for (int u = 0; u < TL_WR; ++u) load_window(nxt, u);
#pragma unroll
      for (int s = 0; s < TL_SR; ++s) load_sample(l + 1, s);
    }

    // ---- phase B: 16 lanes per query; per sample two 256-B spans (top pair, bottom pair)
#pragma unroll
    for (int k = 0; k < TL_QMAX; ++k) {
      const int qi = grp + k * TL_GROUPS;
      if (qi < total && !(tg.ablate & 4)) {
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
          const long long e = ((long long)qglob[qi] * M * L + l) * P;
#pragma unroll 1
          for (int p = 0; p < 4; ++p) {
            const int slp = (p == 0) ? s0 : (p == 1) ? s1 : (p == 2) ? s2 : s3;
            if (slp < 0) {
              const float2 xy = reinterpret_cast<const float2*>(locn)[e + p];
              const Footprint f = footprint(H, W, xy.x, xy.y, attn_n[e + p]);
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
    cur = nxt;
    TSTAMP(5 + 3 * l);
  }

  // ---- add the two corner columns (lane <-> partner lane with pos ^ 8) and store: the left lane
  // finishes channels {0,1} of its chunk, the right lane channels {2,3} (two exchanges per query)
  const int ppos = pos ^ 8;
  const int phl = g ? (ppos < 8 ? ppos + 4 : ppos < 12 ? ppos + 8 : ppos + 16)
                    : (ppos < 4 ? ppos : ppos < 8 ? ppos + 8 : ppos + 12);
  const int partner = (lane & 32) | phl;
#pragma unroll
  for (int k = 0; k < TL_QMAX; ++k) {
    const int qi = grp + k * TL_GROUPS;
    const float r0 = __shfl(side ? acc[k].x : acc[k].z, partner, 64);
    const float r1 = __shfl(side ? acc[k].y : acc[k].w, partner, 64);
    const float2 o = side ? make_float2(acc[k].z + r0, acc[k].w + r1) : make_float2(acc[k].x + r0, acc[k].y + r1);
    if (qi < total)
      reinterpret_cast<float2*>(out + (((long long)n * S + qglob[qi]) * M + m) * D)[chunk * 2 + side] = o;
  }
  TSTAMP(15);
}

static int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

static long long floor_div(long long a, long long b) {  // b > 0
  return (a >= 0) ? a / b : -((-a + b - 1) / b);
}
static long long ceil_div(long long a, long long b) {  // b > 0
  return -floor_div(-a, b);
}

// ---- host-side geometry: built once per (device, level shapes, tile parameters), kept for the
// lifetime of the process (a few hundred bytes of device memory per distinct geometry).
struct GeoKey {
  int dev, L, TH, TW, R;
  int H[UNIVS_MAX_LEVELS], W[UNIVS_MAX_LEVELS];
  bool operator==(const GeoKey& o) const {
    if (dev != o.dev || L != o.L || TH != o.TH || TW != o.TW || R != o.R) return false;
    for (int l = 0; l < L; ++l)
      if (H[l] != o.H[l] || W[l] != o.W[l]) return false;
    return true;
  }
};
struct GeoEntry {
  GeoKey key;
  int4* table;       // device
  int tiles_y, tiles_x;
  long long qmax;    // max queries of a tile
  long long win_px;  // max window pixels of a (tile, level)
};

// one axis of one level: query interval and window interval of tile t
static void axis_entry(int t, int ntile, int T, int Nq, int Nf, int R, int cap, int4& e) {
  // queries: pixel centres (i + 0.5) / Nq inside [t*T/Nf, (t+1)*T/Nf)
  long long lo = std::max<long long>(0, ceil_div(2LL * t * T * Nq - Nf, 2LL * Nf));
  long long hi = (t + 1 == ntile) ? Nq : ceil_div(2LL * (t + 1) * T * Nq - Nf, 2LL * Nf);
  hi = std::min<long long>(std::max(hi, lo), Nq);
  // window: bilinear corners of samples within R pixels of the tile's box, clipped to the zero ring
  const long long num1 = std::min<long long>((long long)(t + 1) * T, Nf);
  long long w0 = std::max<long long>(-1, floor_div(2LL * t * T * Nq - (1 + 2LL * R) * Nf, 2LL * Nf));
  long long w1 = std::min<long long>(Nq, floor_div(2LL * num1 * Nq - (1 - 2LL * R) * Nf, 2LL * Nf) + 1);
  long long wn = std::max<long long>(w1 - w0 + 1, 2);
  wn = std::min<long long>(wn, cap);
  e.x = (int)lo; e.y = (int)(hi - lo); e.z = (int)w0; e.w = (int)wn;
}

static const GeoEntry* geometry(const LevelTable& lv, int L, int fine, int TH, int TW, int R, long long cap_px) {
  static std::mutex mu;
  static std::vector<GeoEntry*> cache;
  GeoKey key{};
  if (hipGetDevice(&key.dev) != hipSuccess) return nullptr;
  key.L = L; key.TH = TH; key.TW = TW; key.R = R;
  for (int l = 0; l < L; ++l) { key.H[l] = lv.H[l]; key.W[l] = lv.W[l]; }
  std::lock_guard<std::mutex> lock(mu);
  for (const GeoEntry* e : cache)
    if (e->key == key) return e;

  GeoEntry* ge = new GeoEntry();
  ge->key = key;
  ge->tiles_y = (lv.H[fine] + TH - 1) / TH;
  ge->tiles_x = (lv.W[fine] + TW - 1) / TW;
  std::vector<int4> tab((size_t)L * (ge->tiles_x + ge->tiles_y));
  for (int l = 0; l < L; ++l) {
    for (int tx = 0; tx < ge->tiles_x; ++tx)
      axis_entry(tx, ge->tiles_x, TW, lv.W[l], lv.W[fine], R, TL_WIN_MAX, tab[(size_t)l * ge->tiles_x + tx]);
    for (int ty = 0; ty < ge->tiles_y; ++ty)
      axis_entry(ty, ge->tiles_y, TH, lv.H[l], lv.H[fine], R, TL_WIN_MAX,
                 tab[(size_t)L * ge->tiles_x + (size_t)l * ge->tiles_y + ty]);
  }
  // windows must fit the LDS carve: shrink rows where a (tile, level) would not (samples beyond go
  // through the global fallback, results unchanged)
  ge->qmax = 0; ge->win_px = 4;
  for (int l = 0; l < L; ++l) {
    int mw = 2, mqx = 0, mqy = 0;
    for (int tx = 0; tx < ge->tiles_x; ++tx) {
      mw = std::max(mw, tab[(size_t)l * ge->tiles_x + tx].w);
      mqx = std::max(mqx, tab[(size_t)l * ge->tiles_x + tx].y);
    }
    for (int ty = 0; ty < ge->tiles_y; ++ty) {
      int4& e = tab[(size_t)L * ge->tiles_x + (size_t)l * ge->tiles_y + ty];
      e.w = (int)std::max<long long>(2, std::min<long long>(e.w, cap_px / mw));
      mqy = std::max(mqy, e.y);
      ge->win_px = std::max<long long>(ge->win_px, (long long)mw * e.w);
    }
    ge->qmax += (long long)mqx * mqy;
  }
  if (hipMalloc(reinterpret_cast<void**>(&ge->table), tab.size() * sizeof(int4)) != hipSuccess ||
      hipMemcpy(ge->table, tab.data(), tab.size() * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipGetLastError();
    delete ge;
    return nullptr;
  }
  cache.push_back(ge);
  return ge;
}

template <int L>
static void launch_tiled(unsigned nblocks, size_t lds, hipStream_t st, const float* value,
                         const LevelTable& lv, const TileGeom& tg, const int4* geo, const float* loc,
                         const float* attn, int N, int S, int M, float* out) {
  auto k = msda_fwd_tiled<L>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(nblocks), dim3(TL_THREADS), lds, st, value, lv, tg, geo, loc, attn, N, S, M, out, nblocks);
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

  const int TH = env_int("UNIVS_MSDA_TILE_H", 16), TW = env_int("UNIVS_MSDA_TILE_W", 16);
  const int R = env_int("UNIVS_MSDA_HALO", 6);
  if (TH < 1 || TW < 1 || R < 0 || R > 64) return 0;
  const size_t fixed = (size_t)TL_NSMP * 16 + (size_t)TL_QCAP * 4;
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
  switch (L) {
    case 1: launch_tiled<1>(nblocks, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
    case 2: launch_tiled<2>(nblocks, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
    case 3: launch_tiled<3>(nblocks, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
    default: launch_tiled<4>(nblocks, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); break;
  }
  int rc = check_launch("msda_fwd_tiled");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
