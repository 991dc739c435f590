// LDS-tiled MSDA forward, third generation (gfx950): LDS-DMA window fills, symmetric waves, sample records
// that never leave the register file.
//
// Operator: ms_deform_attn_forward (ops/src/ms_deform_attn.h:25-44; kernel ms_deform_im2col_cuda.cuh:242-304,
// bilinear helper :38-89) for the encoder geometry (Lq == S, D = 32, P = 4).  Decomposition as in msda_tiled2.hip:
// persistent workgroups (one per CU) walk (frame, tile, head) items inside their XCD's chunk; an item is a
// TW x TH tile of the finest level plus the queries of the coarser levels above it, visited level by level
// ("steps"); the window [tile box +- halo] of the level's value map for this head is staged in LDS, two regions
// alternate by step parity.  What changed against the second generation, and why (profiles/r01_msda_pmc_v3.txt:
// 48 % of the wave cycles parked at barriers / waitcnts, 6 of 16 waves writing LDS, 1/3 of the LDS read cycles
// spent on 16-byte sample records):
//
//   * Window fills are LDS-DMA (`global_load_lds_dwordx4`, 1 KiB per wave instruction): no staging registers, no
//     ds_write_b128 (13 LDS cycles per KiB on gfx950), issued by ALL waves one step ahead; the workgroup barrier
//     that ends a step is also the point where the DMA of the next window is known to have landed.
//   * Windows are clipped to the level (no zero ring): out-of-level bilinear corners are folded into the corner
//     weights (the 2-row footprint is moved inside the level and the weight of the missing row / column is 0), so
//     every staged pixel is real data and the DMA needs no bounds handling.
//   * No producer / consumer split: every wave builds the records of the samples it gathers, in registers, and
//     hands them to the gathering lanes with DPP `row_newbcast` folded into the consuming instruction:
//         v_add_u32_dpp   addr,  bcast_k(slot),   lane_offset        (x2: top / bottom row)
//         ds_read_b64     d,     addr                                (x2)
//         v_fmac_f32_dpp  acc,   bcast_k(weight), d                  (x4)
//     A DPP row (16 lanes) owns one corner COLUMN of a sample (left or right pixel, 32 channels x 2 rows, 8 bytes
//     per lane and row); rows 2i / 2i+1 of a wave work on the same sample, so the 32 lanes of one ds_read_b64
//     lane group read two horizontally adjacent pixels = 256 contiguous bytes = every LDS bank once (conflict-free
//     by construction, the bank is (addr / 4) mod 64 for b64).  Lane k of a row holds the record {slot, w_top,
//     w_bottom} of sample k of its row pair (4 queries x 4 points per level step) for ITS corner column.
//   * Samples whose footprint leaves the staged window are rare (halo 6: < 0.03 % at the bench geometry); their
//     record weights are 0 and a wave-uniform slow path adds them from global memory.
//
// LDS holds windows only (two regions).  One workgroup barrier per step.
#include "msda_geometry.h"
#include "msda_tiled3_record.h"

namespace univs {

typedef float t3v2 __attribute__((ext_vector_type(2)));
#define T3_LDS __attribute__((address_space(3)))
#define T3_GLB __attribute__((address_space(1)))

constexpr int T3_NB = 2;          // record batches per wave and step (a batch = 2 row pairs x 4 queries)
constexpr int T3_QPW = 8 * T3_NB; // queries per wave and step
constexpr int T3_WIN_PX = 768;    // cap on one window (pixels); the two regions together must fit 160 KiB

struct Tile3Geom {
  int tiles_y, tiles_x;
  int ord[2][UNIVS_MAX_LEVELS];   // level visited at step k of an even / odd item
  int reg[2][UNIVS_MAX_LEVELS];   // LDS byte offset of that step's window region
  int ablate;
};

// addr_top = bcast_k(slot) + off_top, addr_bottom = bcast_k(slot) + off_bottom   (k = K, lane K of my DPP row)
template <int K>
__device__ __forceinline__ void t3_addr(int slot, int off_t, int off_b, int& a_t, int& a_b) {
  if (K == 0)   // the record registers may have been written by the instruction just before: DPP read hazard
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %2, %3 row_newbcast:%c5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %2, %4 row_newbcast:%c5 row_mask:0xf bank_mask:0xf"
        : "=&v"(a_t), "=&v"(a_b)
        : "v"(slot), "v"(off_t), "v"(off_b), "n"(K));
  else
    asm("v_add_u32_dpp %0, %2, %3 row_newbcast:%c5 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %2, %4 row_newbcast:%c5 row_mask:0xf bank_mask:0xf"
        : "=&v"(a_t), "=&v"(a_b)
        : "v"(slot), "v"(off_t), "v"(off_b), "n"(K));
}

// acc += bcast_k(w) * d on my two channels.  NOP: pad the VALU-write -> DPP-read hazard (2 wait states) for the first
// use of a weight register; hipcc does not look inside the asm, so it cannot do it.
template <int K, bool NOP>
__device__ __forceinline__ void t3_fma(float& ax, float& ay, float w, t3v2 d) {
  if (NOP)
    asm("s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %2, %3 row_newbcast:%c5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %4 row_newbcast:%c5 row_mask:0xf bank_mask:0xf"
        : "+v"(ax), "+v"(ay)
        : "v"(w), "v"(d.x), "v"(d.y), "n"(K));
  else
    asm("v_fmac_f32_dpp %0, %2, %3 row_newbcast:%c5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %2, %4 row_newbcast:%c5 row_mask:0xf bank_mask:0xf"
        : "+v"(ax), "+v"(ay)
        : "v"(w), "v"(d.x), "v"(d.y), "n"(K));
}

template <int K>
__device__ __forceinline__ int t3_bcast(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, false);
}

template <int L, int NW>
__global__ __launch_bounds__(64 * NW) void msda_fwd_tiled3(const float* __restrict__ value, LevelTable lv,
                                                            Tile3Geom tg, const int4* __restrict__ geo,
                                                            const float* __restrict__ loc,
                                                            const float* __restrict__ attn, int N, int S, int M,
                                                            float* __restrict__ out, unsigned nitems) {
  constexpr int D = 32, P = 4, NB = T3_NB;
  constexpr int MAXPIECES = (T3_WIN_PX / 8 + NW - 1) / NW;
  extern __shared__ __attribute__((aligned(1024))) char lds3[];
  const unsigned lds_base = (unsigned)(unsigned long long)(T3_LDS char*)lds3;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int rp = lane >> 5;            // row pair of the wave
  const int side = (lane >> 4) & 1;    // corner column of my DPP row: 0 left, 1 right
  const int k = lane & 15;             // my record: sample k of my row pair = (query k >> 2, point k & 3)
  const int ch = (lane & 15) * 2;      // my two channels as a gathering lane
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
    int pre[L + 1];
    // per level: query box (qx0, qy0, qnx columns) and window (wx0, wy0, ww x wh) of this tile -- plain ints (arrays
    // of HIP int4 kept live across the loop back-edge are copied through scratch)
    int qx0[L], qy0[L], qnx[L], wx0[L], wy0[L], ww[L], wh[L];
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
    it.pre[0] = 0;
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const int4 gx = geo[l * tg.tiles_x + it.tx];
      const int4 gy = geo[L * tg.tiles_x + l * tg.tiles_y + it.ty];
      it.qx0[l] = gx.x; it.qnx[l] = gx.y; it.wx0[l] = gx.z; it.ww[l] = gx.w;
      it.qy0[l] = gy.x; it.wy0[l] = gy.z; it.wh[l] = gy.w;
      it.pre[l + 1] = it.pre[l] + gx.y * gy.y;
    }
    it.total = it.pre[L];   // 1 .. 16 * NW (host-checked)
    return it;
  };
  // global index of the item's query i (levels in order, raster inside the level's query box)
  auto qglob = [&](const Item& it, int i) __attribute__((always_inline)) {
    i = min(i, it.total - 1);
    int li = i, qx0 = it.qx0[0], qnx = it.qnx[0], qy0 = it.qy0[0], Wq = lv.W[0], st = lv.start[0];
#pragma unroll
    for (int jl = 1; jl < L; ++jl) {   // selects on values (an `if` with several assignments becomes a pointer phi into
      const bool c = i >= it.pre[jl];  // the struct, which then lives in scratch)
      li = c ? i - it.pre[jl] : li;
      qx0 = c ? it.qx0[jl] : qx0;
      qnx = c ? it.qnx[jl] : qnx;
      qy0 = c ? it.qy0[jl] : qy0;
      Wq = c ? lv.W[jl] : Wq;
      st = c ? lv.start[jl] : st;
    }
    const int row = (int)(((float)li + 0.5f) * __builtin_amdgcn_rcpf((float)qnx));
    return st + (qy0 + row) * Wq + qx0 + (li - row * qnx);
  };

  struct LevelGeo {   // workgroup-uniform
    int l, H, W, start, wx0, wy0, ww, wh, npx, reg;
  };
  auto level_geo = [&](const Item& it, int par, int kk) __attribute__((always_inline)) {
    const int l = tg.ord[par][kk];
    LevelGeo q;
    q.l = l;
    q.H = lv.H[0]; q.W = lv.W[0]; q.start = lv.start[0];
    q.wx0 = it.wx0[0]; q.ww = it.ww[0]; q.wy0 = it.wy0[0]; q.wh = it.wh[0];
#pragma unroll
    for (int jl = 1; jl < L; ++jl) {
      const bool c = l == jl;
      q.H = c ? lv.H[jl] : q.H; q.W = c ? lv.W[jl] : q.W; q.start = c ? lv.start[jl] : q.start;
      q.wx0 = c ? it.wx0[jl] : q.wx0; q.ww = c ? it.ww[jl] : q.ww;
      q.wy0 = c ? it.wy0[jl] : q.wy0; q.wh = c ? it.wh[jl] : q.wh;
    }
    q.npx = q.ww * q.wh;
    q.reg = tg.reg[par][kk];
    return q;
  };

  // ---- window fill: LDS-DMA, 8 pixels (1 KiB) per wave instruction, pieces dealt round-robin to the waves
  auto dma_window = [&](const Item& it, const LevelGeo& q) __attribute__((always_inline)) {
    if (tg.ablate & 1) return;
    const int npieces = (q.npx + 7) >> 3;
    const float* vb = value + (it.nm + (long long)q.start * M) * D;   // uniform
    const float rww = __builtin_amdgcn_rcpf((float)q.ww);
#pragma unroll
    for (int u = 0; u < MAXPIECES; ++u) {
      const int i = wave + u * NW;   // uniform
      if (i < npieces) {
        const int jpx = min(i * 8 + (lane >> 3), q.npx - 1);   // the tail piece repeats the last pixel
        const int ry = (int)(((float)jpx + 0.5f) * rww);
        const int rx = jpx - ry * q.ww;
        const unsigned off = (unsigned)(((q.wy0 + ry) * q.W + q.wx0 + rx) * (M * D) + (lane & 7) * 4);
        __builtin_amdgcn_global_load_lds((const T3_GLB void*)(vb + off), (T3_LDS void*)(lds3 + q.reg + i * 1024), 16, 0, 0);
      }
    }
  };

  // ---- my samples of a step: sampling location + attention weight of (query, point) for batch b
  struct Inputs {
    float x[NB], y[NB], a[NB];
  };
  auto load_inputs = [&](const Item& it, int l, const int (&qg)[NB], Inputs& in) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const unsigned e = (unsigned)((qg[b] * M * L + l) * P + (k & 3));
      const float2 xy = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(loc + it.nm * (L * P * 2)) + e * 8u);
      in.x[b] = xy.x; in.y[b] = xy.y;
      in.a[b] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(attn + it.nm * (L * P)) + e * 4u);
    }
  };
  auto my_queries = [&](const Item& it, int (&qg)[NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int b = 0; b < NB; ++b) qg[b] = qglob(it, ((wave * NB + b) * 2 + rp) * 4 + (k >> 2));
  };

  // ---- prologue: first window + first inputs
  Item cur = make_item(widx);
  int qg_cur[NB];
  my_queries(cur, qg_cur);
  Inputs in_cur;
  {
    const LevelGeo g0 = level_geo(cur, 0, 0);
    dma_window(cur, g0);
    load_inputs(cur, g0.l, qg_cur, in_cur);
  }
  __syncthreads();   // (the compiler drains vmcnt before the barrier: the DMA has landed)

  int par = 0;
#pragma unroll 1
  for (unsigned idx = widx;; idx += nw) {
    const bool has_next = idx + nw < csize;
    const Item nxt = make_item(has_next ? idx + nw : idx);
    int qg_nxt[NB];
    my_queries(nxt, qg_nxt);
    float ax[NB][4], ay[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) ax[b][jj] = ay[b][jj] = 0.f;

#pragma unroll
    for (int kk = 0; kk < L; ++kk) {
      const LevelGeo q = level_geo(cur, par, kk);
      // ---- A. records of this step for my corner column (inputs were loaded one step ago)
      int slot[NB];
      float wT[NB], wB[NB];
      bool miss[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const bool qvalid = ((wave * NB + b) * 2 + rp) * 4 + (k >> 2) < cur.total;
        const T3Record r = t3_record(in_cur.x[b], in_cur.y[b], in_cur.a[b], side, qvalid, q.H, q.W, q.wx0, q.wy0, q.ww, q.wh);
        miss[b] = r.miss;
        slot[b] = r.slot;
        wT[b] = r.wt;
        wB[b] = r.wb;
        // the records are complete HERE (not sunk next to their first DPP use, see t3_fma)
        asm volatile("" : "+v"(slot[b]), "+v"(wT[b]), "+v"(wB[b]));
      }

      // ---- B. next step: window DMA into the other region, inputs into registers
      const bool last = kk + 1 == L;
      const bool valid1 = !last || has_next;
      Inputs in_nxt = in_cur;
      if (valid1) {
        if (!last) {
          const LevelGeo q1 = level_geo(cur, par, (kk + 1) % L);
          dma_window(cur, q1);
          load_inputs(cur, q1.l, qg_cur, in_nxt);
        } else {
          const LevelGeo q1 = level_geo(nxt, par ^ 1, 0);
          dma_window(nxt, q1);
          load_inputs(nxt, q1.l, qg_nxt, in_nxt);
        }
      }

      // ---- C. gathers: 16 samples per row pair and batch
      const int off_t = (int)lds_base + q.reg + (lane & 15) * 8;
      const int off_b = off_t + q.ww * (D * 4);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if ((wave * NB + b) * 8 < cur.total && !(tg.ablate & 4)) {   // uniform
          int a_t[16], a_b[16];
#define T3_ADDR(K) t3_addr<K>(slot[b], off_t, off_b, a_t[K], a_b[K]);
          T3_ADDR(0) T3_ADDR(1) T3_ADDR(2) T3_ADDR(3) T3_ADDR(4) T3_ADDR(5) T3_ADDR(6) T3_ADDR(7)
          T3_ADDR(8) T3_ADDR(9) T3_ADDR(10) T3_ADDR(11) T3_ADDR(12) T3_ADDR(13) T3_ADDR(14) T3_ADDR(15)
#undef T3_ADDR
          t3v2 dt[16], db[16];
#pragma unroll
          for (int K = 0; K < 16; ++K) {
            dt[K] = *(const T3_LDS t3v2*)(unsigned long long)(unsigned)a_t[K];
            db[K] = *(const T3_LDS t3v2*)(unsigned long long)(unsigned)a_b[K];
          }
#define T3_FMA(K) t3_fma<K, K == 0>(ax[b][K >> 2], ay[b][K >> 2], wT[b], dt[K]); t3_fma<K, K == 0>(ax[b][K >> 2], ay[b][K >> 2], wB[b], db[K]);
          T3_FMA(0) T3_FMA(1) T3_FMA(2) T3_FMA(3) T3_FMA(4) T3_FMA(5) T3_FMA(6) T3_FMA(7)
          T3_FMA(8) T3_FMA(9) T3_FMA(10) T3_FMA(11) T3_FMA(12) T3_FMA(13) T3_FMA(14) T3_FMA(15)
#undef T3_FMA
          // rare: corner columns outside the staged window -> straight from global memory (wave-uniform loop)
          unsigned long long mm = __ballot(miss[b]);
          if (mm != 0 && !(tg.ablate & 8)) {
            const float* vl = value + (cur.nm + (long long)q.start * M) * D + ch;
#pragma unroll 1
            while (mm) {
              const int bl = __builtin_ctzll(mm);
              mm &= mm - 1;
              const float mx = __shfl(in_cur.x[b], bl, 64), my = __shfl(in_cur.y[b], bl, 64), ma = __shfl(in_cur.a[b], bl, 64);
              const Footprint fp = footprint(q.H, q.W, mx, my, ma);
              const int sd = (bl >> 4) & 1;
              const int wc = sd ? fp.w1 : fp.w0;
              const float w0 = sd ? fp.w01 : fp.w00, w1 = sd ? fp.w11 : fp.w10;
              if ((lane >> 4) == (bl >> 4)) {
                const t3v2 g0 = *reinterpret_cast<const t3v2*>(vl + (long long)(fp.h0 * q.W + wc) * (M * D));
                const t3v2 g1 = *reinterpret_cast<const t3v2*>(vl + (long long)(fp.h1 * q.W + wc) * (M * D));
                const float cx = fmaf(w0, g0.x, w1 * g1.x), cy = fmaf(w0, g0.y, w1 * g1.y);
                const int jb = (bl & 15) >> 2;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                  if (jb == jj) { ax[b][jj] += cx; ay[b][jj] += cy; }
              }
            }
          }
        }
      }

      // ---- D. last level: add the two corner columns (rows 2i / 2i+1) and store
      if (last && !(tg.ablate & 16)) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if ((wave * NB + b) * 8 < cur.total) {   // uniform
            const int qb = ((wave * NB + b) * 2 + rp) * 4;
#define T3_OUT(JJ)                                                                                                 \
  {                                                                                                                \
    const float sx = ax[b][JJ] + __shfl_xor(ax[b][JJ], 16, 64), sy = ay[b][JJ] + __shfl_xor(ay[b][JJ], 16, 64);   \
    const int qgj = t3_bcast<4 * JJ>(qg_cur[b]);                                                                   \
    if (side == 0 && qb + JJ < cur.total)                                                                          \
      *reinterpret_cast<t3v2*>(reinterpret_cast<char*>(out + cur.nm * D) + ((unsigned)(qgj * M * D + ch) * 4u)) =  \
          (t3v2){sx, sy};                                                                                          \
  }
            T3_OUT(0) T3_OUT(1) T3_OUT(2) T3_OUT(3)
#undef T3_OUT
          }
        }
      }
      __syncthreads();   // the one barrier of the step: everybody is done with this region, the next window has landed
      in_cur = in_nxt;
    }
    if (!has_next) break;
    cur = nxt;
#pragma unroll
    for (int b = 0; b < NB; ++b) qg_cur[b] = qg_nxt[b];
    par ^= 1;
  }
}

template <int L, int NW>
static void launch_tiled3(unsigned grid, unsigned nitems, size_t lds, hipStream_t st, const float* value,
                          const LevelTable& lv, const Tile3Geom& tg, const int4* geo, const float* loc,
                          const float* attn, int N, int S, int M, float* out) {
  auto kfn = msda_fwd_tiled3<L, NW>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * NW), lds, st, value, lv, tg, geo, loc, attn, N, S, M, out, nitems);
}

// returns 1 if launched, 0 if preconditions do not hold (caller tries the next implementation), <0 on error
int msda_forward_tiled3_f32(const float* value, const LevelTable& lv, const float* loc, const float* attn, int N,
                            int S, int M, int D, int L, int Lq, int P, float* out, hipStream_t st) {
  if (D != 32 || P != 4 || L < 1 || L > 4 || Lq != S || M < 1) return 0;
  if ((long long)S * M * D * 4 >= (1LL << 31) || (long long)S * M * L * P * 8 >= (1LL << 31)) return 0;
  long long expect = 0;
  int fine = 0;
  for (int l = 0; l < L; ++l) {
    if (lv.start[l] != expect || lv.H[l] < 2 || lv.W[l] < 2) return 0;
    expect += (long long)lv.H[l] * lv.W[l];
    if ((long long)lv.H[l] * lv.W[l] > (long long)lv.H[fine] * lv.W[fine]) fine = l;
  }
  if (expect != S) return 0;

  const int TH = env_int("UNIVS_MSDA_TILE3_H", 8), TW = env_int("UNIVS_MSDA_TILE3_W", 16);
  const int R = env_int("UNIVS_MSDA_HALO", 6);
  if (TH < 1 || TW < 1 || R < 0 || R > 64) return 0;
  const GeoEntry* ge = geometry(lv, L, fine, TH, TW, R, T3_WIN_PX, /*ring=*/0);
  if (!ge || ge->qmax > 12 * T3_QPW) return 0;
  const int NW = ge->qmax <= 11 * T3_QPW ? 11 : 12;

  // region plan: step parity picks the region; an odd level count makes odd items start in B, so they visit their
  // two largest windows in swapped order (the accumulators do not care)
  Tile3Geom tg{};
  tg.tiles_y = ge->tiles_y;
  tg.tiles_x = ge->tiles_x;
  tg.ablate = env_int("UNIVS_MSDA_ABLATE", 0);
  int by_size[UNIVS_MAX_LEVELS];
  for (int l = 0; l < L; ++l) by_size[l] = l;
  std::sort(by_size, by_size + L, [&](int a, int b) { return ge->lvl_px[a] > ge->lvl_px[b]; });
  long long capA = 0, capB = 0;
  for (int par = 0; par < 2; ++par) {
    for (int kk = 0; kk < L; ++kk) tg.ord[par][kk] = by_size[kk];
    const int first_region = (par == 1 && (L & 1)) ? 1 : 0;
    if (first_region == 1 && L >= 2) std::swap(tg.ord[par][0], tg.ord[par][1]);
    for (int kk = 0; kk < L; ++kk) {
      const int region = (first_region + kk) & 1;
      long long& cap = region ? capB : capA;
      cap = std::max(cap, ge->lvl_px[tg.ord[par][kk]]);
      tg.reg[par][kk] = region;   // resolved to an offset below
    }
  }
  capA = (capA + 7) & ~7LL;   // whole 1-KiB DMA pieces
  capB = (capB + 7) & ~7LL;
  const size_t lds = (size_t)(capA + capB) * 128;
  if (lds > 160 * 1024) return 0;
  for (int par = 0; par < 2; ++par)
    for (int kk = 0; kk < L; ++kk) tg.reg[par][kk] = tg.reg[par][kk] ? (int)(capA * 128) : 0;

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
  const unsigned grid = (unsigned)std::min<long long>(nb, std::max(env_int("UNIVS_MSDA_GRID", n_cu), 1));
#define T3_LAUNCH(LL)                                                                                          \
  if (NW == 11) launch_tiled3<LL, 11>(grid, (unsigned)nb, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out); \
  else launch_tiled3<LL, 12>(grid, (unsigned)nb, lds, st, value, lv, tg, ge->table, loc, attn, N, S, M, out);
  switch (L) {
    case 1: T3_LAUNCH(1) break;
    case 2: T3_LAUNCH(2) break;
    case 3: T3_LAUNCH(3) break;
    default: T3_LAUNCH(4) break;
  }
#undef T3_LAUNCH
  int rc = check_launch("msda_fwd_tiled3");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
