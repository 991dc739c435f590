// y[M, N] = x[M, K] * W[N, K]^T + bias[N]  (nn.Linear, fp32) on the bf16 matrix cores of gfx950, fp32-accurate.
//
// The token projections of the MSDeformAttn encoder (ms_deform_attn.py:95-113: value_proj, the merged sampling-offset /
// attention-weight projection, output_proj; M = T * 19 320 = 96 600 rows, K = 256, N = 256 / 288) run at 59 TF/s in
// hipBLASLt's best fp32 algorithm (211 / 137 us, 18 + 12 launches per clip = 18 % of the clip): fp32 MFMA issues at 1/16
// of the bf16 rate and the shapes are too skinny to hide it.  This kernel applies the arithmetic of the split mask-decode
// kernel (mask_decode.hip: every fp32 operand = h + m + l, three bf16 parts, exact by truncation; six bf16 x bf16 MFMA
// terms accumulated in fp32; dropped terms <= 3 * 2^-24 per product) to the transposed operand layout of a Linear:
//   * W (small, re-used by every row tile) is split once per workgroup into LDS in MFMA A-fragment order.  6 bytes per
//     element: at most 106 output features of K = 256 per pass, so N = 256 / 288 run as three passes (blockIdx.y) over the
//     same rows of x.  The grid's x extent is a multiple of 8, so the three workgroups that share a row range land on the
//     same XCD at the same time and two of the three reads of x are L2 hits;
//   * x is streamed: a wave tile is 32 rows; lane (j, g) of MFMA column tile c holds row 32 * tile + 16 c + j, k = 32 ks +
//     8 g .. + 7 -- 32 contiguous bytes, two 16-byte loads; four register stages (three k-steps in flight);
//   * D[i = feature][j = row]: a lane ends up with four consecutive features of one row -> one 16-byte store; the bias is
//     the accumulators' initial value; the epilogue optionally applies ReLU, exact (erf) GELU, or adds a residual tensor
//     of the output's shape -- the Swin block's fc1 -> GELU and shortcut + fc2 (swin.py:35-58, :291-293) without the
//     separate elementwise passes.
// K is a multiple of 128 (register ring of 4 k-steps) or of 96 (ring of 3: the Swin-T stage widths 96 and 192).
// Everything about pinning the ring (sched_barrier, the opaque split mask, straight-line tile body, unconditional buffer
// stores) is explained in mask_decode.hip / DESIGN.md "Toolchain hazards".
#include "common.h"
#include "config.h"

#include <algorithm>
#include <cstdlib>

namespace univs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int LS_THREADS = 512;   // 8 waves, two per SIMD
constexpr int LS_TILE_M = 32;     // rows of x per wave tile (two 16-column MFMA tiles)
constexpr int LS_MAX_RB = 7;
enum { LS_EPI_NONE = 0, LS_EPI_RELU = 1, LS_EPI_GELU = 2, LS_EPI_RESIDUAL = 3, LS_EPI_BLOCKED = 4 };
// LS_EPI_BLOCKED: the output is stored column-blocked per batch element, Y[M / blk_rows][N / blk_cols][blk_rows][blk_cols]
// (row m = b * blk_rows + s, feature n = c * blk_cols + i -> ((b * (N / blk_cols) + c) * blk_rows + s) * blk_cols + i): the
// head-major operand layouts of the MSDeformAttn sampling kernel (msda_strips.hip: value in blocks of 16 channels, the
// merged offset / logit projection in blocks of one head), written by the producing Linear at no extra cost.

// two fp32 bit patterns whose low halves are zero -> their bf16 pair (element 0 in the low half)
__device__ __forceinline__ unsigned ls_pack(unsigned lo, unsigned hi) { return (lo >> 16) | hi; }

// 8 consecutive k of one row (two float4, pairs as loaded) -> the three bf16x8 parts
__device__ __forceinline__ void ls_split8(f32x4 v0, f32x4 v1, unsigned hi_mask, bf16x8& h, bf16x8& m, bf16x8& l) {
  const u32x2 mask2 = {hi_mask, hi_mask};
  const f32x2 x[4] = {{v0.x, v0.y}, {v0.z, v0.w}, {v1.x, v1.y}, {v1.z, v1.w}};
  u32x4 hp, mp, lp;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const u32x2 hb = __builtin_bit_cast(u32x2, x[p]) & mask2;
    const f32x2 r = x[p] - __builtin_bit_cast(f32x2, hb);               // exact
    const u32x2 mb = __builtin_bit_cast(u32x2, r) & mask2;
    const u32x2 lb = __builtin_bit_cast(u32x2, r - __builtin_bit_cast(f32x2, mb));   // exact; a bf16 value
    hp[p] = ls_pack(hb.x, hb.y);
    mp[p] = ls_pack(mb.x, mb.y);
    lp[p] = ls_pack(lb.x, lb.y);
  }
  h = __builtin_bit_cast(bf16x8, hp);
  m = __builtin_bit_cast(bf16x8, mp);
  l = __builtin_bit_cast(bf16x8, lp);
}

// KSC = K / 32 (8 for K = 256: straight-line tile body), 0 = runtime; LS_RING = register stages of x (K / 32 is a multiple)
template <int RB, int KSC, int LS_RING, int EPI>
__global__ __launch_bounds__(LS_THREADS, 1) void linear_bf16x6(const float* __restrict__ X,      // [M, K]
                                                                const float* __restrict__ W,      // [N, K]
                                                                const float* __restrict__ bias,   // [N] or null
                                                                const float* __restrict__ Res,    // [M, N] (EPI == RESIDUAL)
                                                                float* __restrict__ Y,            // [M, N]
                                                                int M, int N, int K, int rows_per_pass, int blk_rows,
                                                                int blk_cols) {
  extern __shared__ __attribute__((aligned(16))) u32x4 Wsp[];   // [K/32][4 k-groups][R features][3 parts] | bias[R]
  const int n0 = blockIdx.y * rows_per_pass;
  const int R = min(rows_per_pass, N - n0);                      // a multiple of 4 (host-checked)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = K >> 5;
  const int j = lane & 15, g = lane >> 4;
  float* bias_lds = reinterpret_cast<float*>(Wsp + (size_t)(K >> 3) * R * 3);

  // ---- this wave's row tiles: the workgroup owns a contiguous run, its waves take tiles round-robin
  const int WT = (M + LS_TILE_M - 1) / LS_TILE_M;
  constexpr int NWV = LS_THREADS / 64;
  const int wg0 = (int)((long long)WT * blockIdx.x / gridDim.x), wg1 = (int)((long long)WT * (blockIdx.x + 1) / gridDim.x);
  const int wt0 = wg0 + wave;
  const int ntiles = wt0 < wg1 ? (wg1 - wt0 + NWV - 1) / NWV : 0;
  const int nsteps = max(ntiles, 1) * KS;                        // a multiple of LS_RING (idle waves: one dummy tile)

  const char* Xb = reinterpret_cast<const char*>(X);
  f32x4 raw[LS_RING][2][2];                                      // [stage][column tile][16-byte half]
  auto load_x = [&](f32x4 (&buf)[2][2], int step) __attribute__((always_inline)) {
    const int tile = step / KS, ks = step - tile * KS;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int m = min((wt0 + tile * NWV) * LS_TILE_M + 16 * c + j, M - 1);   // rows past the end repeat the last row
      const char* p = Xb + ((unsigned)m * (unsigned)K + (unsigned)(ks * 32 + 8 * g)) * 4u;   // < 2^31 (host-checked)
      buf[c][0] = *reinterpret_cast<const f32x4*>(p);
      buf[c][1] = *reinterpret_cast<const f32x4*>(p + 16);
    }
  };
#pragma unroll
  for (int u = 0; u < LS_RING; ++u) {
    load_x(raw[u], min(u, nsteps - 1));
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- split this pass's rows of W into LDS (fragment order) while the first four k-steps of x are in flight
  {
    const float* Wsrc = W + (size_t)n0 * K;
    const int kch = K >> 3;
    for (int idx = tid; idx < R * kch; idx += LS_THREADS) {
      const int r = idx / kch, kc = idx - r * kch;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(Wsrc + (size_t)r * K + kc * 8);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(Wsrc + (size_t)r * K + kc * 8 + 4);
      bf16x8 h, m, l;
      ls_split8(x0, x1, 0xFFFF0000u, h, m, l);
      u32x4* dst = Wsp + (((kc >> 2) * 4 + (kc & 3)) * R + r) * 3;
      dst[0] = __builtin_bit_cast(u32x4, h);
      dst[1] = __builtin_bit_cast(u32x4, m);
      dst[2] = __builtin_bit_cast(u32x4, l);
    }
    for (int r = tid; r < R; r += LS_THREADS) bias_lds[r] = bias ? bias[n0 + r] : 0.f;
    // Feature blocks of a SHORT last pass (R < rows_per_pass) read A fragments for rows >= R: those addresses belong to the
    // next k-group's rows, and for the last k-group to the bias area and up to 16 rows x 48 B behind it.  Their products
    // only reach features that are never stored (the f < R guard); zero the tail so that they are at least finite.
    for (int i = tid; i < 16 * 3 * 4; i += LS_THREADS) reinterpret_cast<unsigned*>(bias_lds + R)[i] = 0u;
  }
  __syncthreads();   // the only barrier
  if (ntiles == 0) return;

  // bias of this lane's features = the accumulators' initial value (features past R: 0, never stored)
  f32x4 binit[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int f = rb * 16 + 4 * g;
    binit[rb] = f < R ? *reinterpret_cast<const f32x4*>(bias_lds + f) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  f32x4 acc[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = binit[rb];

  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((long long)M * N * 4), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPI == LS_EPI_RESIDUAL ? Res : X), 0, (int)((long long)M * N * 4), 0x00020000);
  // LS_EPI_BLOCKED: this lane's features in the blocked layout, element offset of (block, column) within a batch element
  [[maybe_unused]] unsigned fblk[RB];
  if constexpr (EPI == LS_EPI_BLOCKED) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const unsigned fg = (unsigned)(n0 + rb * 16 + 4 * g);
      const unsigned cb = fg / (unsigned)blk_cols;
      fblk[rb] = cb * (unsigned)blk_rows * (unsigned)blk_cols + (fg - cb * (unsigned)blk_cols);
    }
  }
  auto epilogue = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int m = (wt0 + tile * NWV) * LS_TILE_M + 16 * c + j;
      [[maybe_unused]] unsigned rowpart = 0;
      if constexpr (EPI == LS_EPI_BLOCKED) {
        const unsigned b = (unsigned)m / (unsigned)blk_rows;
        rowpart = b * (unsigned)blk_rows * (unsigned)N + ((unsigned)m - b * (unsigned)blk_rows) * (unsigned)blk_cols;
      }
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const int f = rb * 16 + 4 * g;
        f32x4 v = acc[rb][c];
        const unsigned off = EPI == LS_EPI_BLOCKED ? (rowpart + fblk[rb]) * 4u
                                                   : ((unsigned)m * (unsigned)N + (unsigned)(n0 + f)) * 4u;
        const unsigned offc = (m < M && f < R) ? off : 0xFFFFFFF0u;   // out of range: loads return 0, stores are dropped
        if (EPI == LS_EPI_RELU) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
        if (EPI == LS_EPI_GELU) {   // x * 0.5 * (1 + erf(x / sqrt 2)): nn.GELU() (approximate = 'none')
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = v[e] * 0.5f * (1.0f + erff(v[e] * 0.70710678118654752440f));
        }
        if (EPI == LS_EPI_RESIDUAL) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, offc, 0, 0));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, offc, 0, 0);
      }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = binit[rb];
  };

  auto group = [&](int tile, int ks0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < LS_RING; ++u) {
      const int ks = ks0 + u;
      const int step = tile * KS + ks;
      unsigned hi_mask = 0xFFFF0000u;
      asm volatile("" : "+s"(hi_mask) : : "memory");            // pins the consumption of ring stage u here
      bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) ls_split8(raw[u][c][0], raw[u][c][1], hi_mask, bh[c], bm[c], bl[c]);
      __builtin_amdgcn_sched_barrier(0);
      load_x(raw[u], min(step + LS_RING, nsteps - 1));          // refill the stage just consumed (4 steps ahead)
      __builtin_amdgcn_sched_barrier(0);

      const u32x4* ap = Wsp + ((ks * 4 + g) * R + j) * 3;
      const u32x4* ap_last = Wsp + ((ks * 4 + g) * R + min((RB - 1) * 16 + j, R - 1)) * 3;
      auto afrag = [&](int rb, bf16x8 (&a)[3]) __attribute__((always_inline)) {
        const u32x4* p0 = (rb == RB - 1) ? ap_last : ap + rb * 48;
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) a[p3] = __builtin_bit_cast(bf16x8, p0[p3]);
      };
      bf16x8 abuf[2][3];                                         // h, m, l of the current / next feature block
      afrag(0, abuf[0]);
      // smallest terms first: l*h, h*l, m*m, m*h, h*m, h*h   (index pairs into {h, m, l})
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (rb + 1 < RB) afrag(rb + 1, abuf[(rb + 1) & 1]);
#pragma unroll
        for (int term = 0; term < 6; ++term) {
          const bf16x8 av = abuf[rb & 1][TA[term]];
          const bf16x8 b0 = TB[term] == 0 ? bh[0] : TB[term] == 1 ? bm[0] : bl[0];
          const bf16x8 b1 = TB[term] == 0 ? bh[1] : TB[term] == 1 ? bm[1] : bl[1];
          acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b0, acc[rb][0], 0, 0, 0);
          acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b1, acc[rb][1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (KSC > 0) {
    // the epilogue of tile i - 1 opens the body of tile i (the first iteration stores the bias to the first tile's own
    // rows, overwritten by its real epilogue later), so entry path and back edge issue identical sequences
#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
      epilogue(max(tile - 1, 0));
#pragma unroll
      for (int ks0 = 0; ks0 < KSC; ks0 += LS_RING) group(tile, ks0);
    }
    epilogue(ntiles - 1);
  } else {
#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
#pragma unroll 1
      for (int ks0 = 0; ks0 < KS; ks0 += LS_RING) group(tile, ks0);
      epilogue(tile);
    }
  }
}

// ---- wide-K variant: x-stationary.
// The kernel above keeps a slab of W (all of K for <= 112 output features) in LDS and makes one pass over x per slab: right
// for K <= 768, hopeless beyond (K = 1024 leaves room for 24 features: eleven passes over x).  Here a wave keeps its 16
// rows of x for ALL of K in flight and owns a [16 rows x NF * 16 features] block of accumulators; W streams through LDS in
// k-steps of 32 (all NF * 16 features x 32 k = 49 KB at 256 features), double-buffered, split to bf16 x 3 by the whole
// workgroup while the previous k-step is multiplied; one barrier per k-step.  Used for the encoder's second FFN Linear
// (K = 1024 -> 256, msdeformattn.py:87-91) and the fc2 of the deeper Swin stages (K = 1536).
template <int NF, int EPI>
__global__ __launch_bounds__(LS_THREADS, NF <= 8 ? 2 : 1) void linear_bf16x6_wide(const float* __restrict__ X,      // [M, K]
                                                                     const float* __restrict__ W,      // [N, K]
                                                                     const float* __restrict__ bias,   // [N] or null
                                                                     const float* __restrict__ Res,    // [M, N]
                                                                     float* __restrict__ Y, int M, int N, int K) {
  constexpr int RING = 4;
  constexpr int NFEAT = NF * 16;
  extern __shared__ __attribute__((aligned(16))) u32x4 Wst[];   // [2 buffers][4 k-groups][NFEAT][3 parts]
  const int n0 = blockIdx.y * NFEAT;
  const int R = min(NFEAT, N - n0);                              // a multiple of 4
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = K >> 5;
  const int j = lane & 15, g = lane >> 4;
  constexpr int NWV = LS_THREADS / 64;
  const int WT = (M + 15) / 16;                                  // 16-row wave tiles
  const int ST = (WT + NWV - 1) / NWV;                           // super-tiles of 8 wave tiles

  // W staging, two phases a k-step apart so that the global latency hides behind a whole k-step of MFMAs:
  //   fetch_w(ks): this pass's k-step `ks` of W -> registers (8 consecutive k of one feature per work item, <= 2 items);
  //   commit_w(buf): registers -> split -> LDS buffer `buf` (fragment order)
  constexpr int WITEMS = (NFEAT * 4 + LS_THREADS - 1) / LS_THREADS;
  f32x4 wraw[WITEMS][2];
  auto fetch_w = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < WITEMS; ++it) {
      const int idx = tid + it * LS_THREADS;
      const int r = idx >> 2, kg = idx & 3;
      wraw[it][0] = wraw[it][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (idx < NFEAT * 4 && r < R) {
        const float* src = W + (size_t)(n0 + r) * K + min(ks, KS - 1) * 32 + kg * 8;
        wraw[it][0] = *reinterpret_cast<const f32x4*>(src);
        wraw[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
      }
    }
  };
  auto commit_w = [&](int buf) __attribute__((always_inline)) {
    u32x4* dst0 = Wst + (size_t)buf * 4 * NFEAT * 3;
#pragma unroll
    for (int it = 0; it < WITEMS; ++it) {
      const int idx = tid + it * LS_THREADS;
      const int r = idx >> 2, kg = idx & 3;
      if (idx < NFEAT * 4) {
        bf16x8 h, m, l;
        ls_split8(wraw[it][0], wraw[it][1], 0xFFFF0000u, h, m, l);
        u32x4* dst = dst0 + (kg * NFEAT + r) * 3;
        dst[0] = __builtin_bit_cast(u32x4, h);
        dst[1] = __builtin_bit_cast(u32x4, m);
        dst[2] = __builtin_bit_cast(u32x4, l);
      }
    }
  };

  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((long long)M * N * 4), 0x00020000);
  [[maybe_unused]] const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(EPI == LS_EPI_RESIDUAL ? Res : X), 0, (int)((long long)M * N * 4), 0x00020000);
  const char* Xb = reinterpret_cast<const char*>(X);

#pragma unroll 1
  for (int st = blockIdx.x; st < ST; st += gridDim.x) {
    const int m = (st * NWV + wave) * 16 + j;                    // my row (lanes j; the 4 k-groups g share it)
    const int mc = min(m, M - 1);
    f32x4 raw[RING][2];
    auto load_x = [&](f32x4 (&buf)[2], int ks) __attribute__((always_inline)) {
      const char* p = Xb + ((unsigned)mc * (unsigned)K + (unsigned)(min(ks, KS - 1) * 32 + 8 * g)) * 4u;
      buf[0] = *reinterpret_cast<const f32x4*>(p);
      buf[1] = *reinterpret_cast<const f32x4*>(p + 16);
    };
#pragma unroll
    for (int u = 0; u < RING; ++u) load_x(raw[u], u);
    f32x4 acc[NF];
#pragma unroll
    for (int rb = 0; rb < NF; ++rb) {
      const int f = rb * 16 + 4 * g;
      acc[rb] = (bias && f < R) ? *reinterpret_cast<const f32x4*>(bias + n0 + f) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                                             // the previous super-tile's last k-step is done with the buffers
    fetch_w(0);
    commit_w(0);
    fetch_w(1);
    __syncthreads();

#pragma unroll 1
    for (int ks0 = 0; ks0 < KS; ks0 += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        const int ks = ks0 + u;
        commit_w((ks + 1) & 1);                                  // W of k-step ks + 1 (fetched a k-step ago) -> the free buffer
        fetch_w(ks + 2);                                         // ... and the one after it -> registers
        bf16x8 bh, bm, bl;
        ls_split8(raw[u][0], raw[u][1], 0xFFFF0000u, bh, bm, bl);
        load_x(raw[u], ks + RING);
        const u32x4* ap = Wst + (size_t)(ks & 1) * 4 * NFEAT * 3 + (g * NFEAT + j) * 3;
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};   // smallest terms first (see above)
#pragma unroll
        for (int rb = 0; rb < NF; ++rb) {
          bf16x8 a3[3];
#pragma unroll
          for (int p3 = 0; p3 < 3; ++p3) a3[p3] = __builtin_bit_cast(bf16x8, ap[rb * 48 + p3]);
#pragma unroll
          for (int term = 0; term < 6; ++term) {
            const bf16x8 b = TB[term] == 0 ? bh : TB[term] == 1 ? bm : bl;
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[TA[term]], b, acc[rb], 0, 0, 0);
          }
        }
        __syncthreads();                                         // buffer (ks + 1) & 1 is staged; buffer ks & 1 is free
      }
    }
    // ---- epilogue: D[i = feature][j = row]: a lane holds four consecutive features of its row
#pragma unroll
    for (int rb = 0; rb < NF; ++rb) {
      const int f = rb * 16 + 4 * g;
      f32x4 v = acc[rb];
      const unsigned off = ((unsigned)m * (unsigned)N + (unsigned)(n0 + f)) * 4u;
      const unsigned offc = (m < M && f < R) ? off : 0xFFFFFFF0u;
      if (EPI == LS_EPI_RELU) v = __builtin_elementwise_max(v, (f32x4){0.f, 0.f, 0.f, 0.f});
      if (EPI == LS_EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * 0.5f * (1.0f + erff(v[e] * 0.70710678118654752440f));
      }
      if (EPI == LS_EPI_RESIDUAL) v += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, offc, 0, 0));
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, offc, 0, 0);
    }
  }
}

// ---- 3 x 3 convolution (stride 1, padding 1, no bias) on NCHW tensors as the x-stationary GEMM above with tap addressing:
//   y[t][co][p] = sum over (tap, ci) of W2[co][tap * Cin + ci] * x[t][ci][p + tap offset]      (zero outside the image)
// i.e. M = T * H * W rows (pixels), K = 9 * Cin, N = Cout; a k-step of 32 is 32 input channels of one tap.  The FPN output
// convolution of the pixel decoder (msdeformattn.py:227-232, :352: 256 -> 256 at 1/4 resolution, 347 GFLOP per 5-frame clip,
// the single largest kernel of the clip at 2.7 ms in MIOpen's fp32 implicit GEMM).  A lane's B fragment is 8 input channels
// of its pixel: 8 dword loads a channel plane apart (16 consecutive pixels = 64 contiguous bytes per load and k-group);
// the output is stored NCHW, 4 dword stores per feature block.
template <int NF>
__global__ __launch_bounds__(LS_THREADS, 1) void conv3x3_bf16x6(const float* __restrict__ X,     // [T, Cin, H, W]
                                                                 const float* __restrict__ W2,    // [Cout, 9 * Cin], tap-major
                                                                 float* __restrict__ Y,           // [T, Cout, H, W]
                                                                 int T, int Cin, int Cout, int H, int Wd) {
  constexpr int RING = 4;
  constexpr int NFEAT = NF * 16;
  extern __shared__ __attribute__((aligned(16))) u32x4 Wst[];   // [2 buffers][4 k-groups][NFEAT][3 parts]
  const int n0 = blockIdx.y * NFEAT;
  const int R = min(NFEAT, Cout - n0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = 9 * Cin, KS = K >> 5, kspt = Cin >> 5;           // k-steps per tap
  const int j = lane & 15, g = lane >> 4;
  constexpr int NWV = LS_THREADS / 64;
  const int HW = H * Wd;
  const int M = T * HW;
  const int WT = (M + 15) / 16;
  const int ST = (WT + NWV - 1) / NWV;

  constexpr int WITEMS = (NFEAT * 4 + LS_THREADS - 1) / LS_THREADS;
  f32x4 wraw[WITEMS][2];
  auto fetch_w = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < WITEMS; ++it) {
      const int idx = tid + it * LS_THREADS;
      const int r = idx >> 2, kg = idx & 3;
      wraw[it][0] = wraw[it][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (idx < NFEAT * 4 && r < R) {
        const float* src = W2 + (size_t)(n0 + r) * K + min(ks, KS - 1) * 32 + kg * 8;
        wraw[it][0] = *reinterpret_cast<const f32x4*>(src);
        wraw[it][1] = *reinterpret_cast<const f32x4*>(src + 4);
      }
    }
  };
  auto commit_w = [&](int buf) __attribute__((always_inline)) {
    u32x4* dst0 = Wst + (size_t)buf * 4 * NFEAT * 3;
#pragma unroll
    for (int it = 0; it < WITEMS; ++it) {
      const int idx = tid + it * LS_THREADS;
      const int r = idx >> 2, kg = idx & 3;
      if (idx < NFEAT * 4) {
        bf16x8 h, m, l;
        ls_split8(wraw[it][0], wraw[it][1], 0xFFFF0000u, h, m, l);
        u32x4* dst = dst0 + (kg * NFEAT + r) * 3;
        dst[0] = __builtin_bit_cast(u32x4, h);
        dst[1] = __builtin_bit_cast(u32x4, m);
        dst[2] = __builtin_bit_cast(u32x4, l);
      }
    }
  };

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)((long long)T * Cin * HW * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((long long)T * Cout * HW * 4), 0x00020000);

#pragma unroll 1
  for (int st = blockIdx.x; st < ST; st += gridDim.x) {
    const int m = (st * NWV + wave) * 16 + j;                    // my pixel (lanes j; the 4 k-groups g share it)
    const bool valid = m < M;
    const int mc = min(m, M - 1);
    const int t = mc / HW, rem = mc - t * HW;
    const int py = rem / Wd, px = rem - py * Wd;
    float raw[RING][8];
    // 8 input channels of tap (ks / kspt) at my pixel; out-of-image taps and rows past M read 0 (offset out of the buffer)
    auto load_x = [&](float (&buf)[8], int ks) __attribute__((always_inline)) {
      const int kk = min(ks, KS - 1);
      const int tap = kk / kspt, cb = (kk - tap * kspt) * 32;    // uniform
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const int yy = py + dy, xx = px + dx;
      const bool inb = valid && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)Wd;
      const unsigned base = (unsigned)(((t * Cin + cb + 8 * g) * H + yy) * Wd + xx) * 4u;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        buf[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, inb ? base + (unsigned)(e * HW) * 4u : 0xFFFFFFF0u, 0, 0));
    };
#pragma unroll
    for (int u = 0; u < RING; ++u) load_x(raw[u], u);
    f32x4 acc[NF];
#pragma unroll
    for (int rb = 0; rb < NF; ++rb) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    fetch_w(0);
    commit_w(0);
    fetch_w(1);
    __syncthreads();

#pragma unroll 1
    for (int ks0 = 0; ks0 < KS; ks0 += RING) {
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        const int ks = ks0 + u;
        commit_w((ks + 1) & 1);
        fetch_w(ks + 2);
        bf16x8 bh, bm, bl;
        ls_split8((f32x4){raw[u][0], raw[u][1], raw[u][2], raw[u][3]}, (f32x4){raw[u][4], raw[u][5], raw[u][6], raw[u][7]},
                  0xFFFF0000u, bh, bm, bl);
        load_x(raw[u], ks + RING);
        const u32x4* ap = Wst + (size_t)(ks & 1) * 4 * NFEAT * 3 + (g * NFEAT + j) * 3;
        constexpr int TA[6] = {2, 0, 1, 1, 0, 0}, TB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int rb = 0; rb < NF; ++rb) {
          bf16x8 a3[3];
#pragma unroll
          for (int p3 = 0; p3 < 3; ++p3) a3[p3] = __builtin_bit_cast(bf16x8, ap[rb * 48 + p3]);
#pragma unroll
          for (int term = 0; term < 6; ++term) {
            const bf16x8 b = TB[term] == 0 ? bh : TB[term] == 1 ? bm : bl;
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[TA[term]], b, acc[rb], 0, 0, 0);
          }
        }
        __syncthreads();
      }
    }
    // ---- epilogue: a lane holds four consecutive output channels of its pixel -> NCHW
#pragma unroll
    for (int rb = 0; rb < NF; ++rb) {
      const int f = rb * 16 + 4 * g;
      // (R is a multiple of 16: one predicate per feature block; out-of-range lanes move their offset out of the buffer)
      const unsigned off0 = (valid && f < R) ? (unsigned)((t * Cout + n0 + f) * HW + rem) * 4u : 0xFFFFFFF0u - 3u * (unsigned)HW * 4u;
      // (each element through an opaque register: hipcc 7.2 otherwise emits all four stores with the FIRST element as data)
      float vx = acc[rb].x, vy = acc[rb].y, vz = acc[rb].z, vw = acc[rb].w;
      asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz), "+v"(vw));
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), yrs, off0, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), yrs, off0 + (unsigned)HW * 4u, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vz), yrs, off0 + 2u * (unsigned)HW * 4u, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vw), yrs, off0 + 3u * (unsigned)HW * 4u, 0, 0);
    }
  }
}

// K % (32 * 4) == 0 <=> (9 * Cin) % 128 == 0 <=> Cin % 128 == 0.  returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED
int conv3x3_split_f32(const float* x, const float* w2, float* y, int T, int Cin, int Cout, int H, int W, hipStream_t st) {
  if (T <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UNIVS_OK;
  const long long M = (long long)T * H * W;
  if (Cin % 128 != 0 || Cout % 16 != 0 || M < 4096 || M * std::max(Cin, Cout) * 4 >= 0x7FFFFFFFLL ||
      (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w2) & 15) || (reinterpret_cast<uintptr_t>(y) & 15))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  const long long ST = ((M + 15) / 16 + 7) / 8;
  const int passes = (Cout + 255) / 256;
  const int nfeat = ((Cout + passes - 1) / passes + 15) / 16 * 16;
  if (nfeat != 256 && nfeat != 128) return UNIVS_ERR_NOT_IMPLEMENTED;
  const unsigned gx = (unsigned)std::min<long long>(ST, std::max(1, n_cu / passes));
  const size_t lds = (size_t)2 * 4 * nfeat * 3 * 16;
  dim3 grid(gx, (unsigned)((Cout + nfeat - 1) / nfeat)), block(LS_THREADS);
  if (nfeat == 256) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16x6<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv3x3_bf16x6<16>), grid, block, lds, st, x, w2, y, T, Cin, Cout, H, W);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_bf16x6<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((conv3x3_bf16x6<8>), grid, block, lds, st, x, w2, y, T, Cin, Cout, H, W);
  }
  return check_launch("conv3x3_split_f32");
}

// returns 1 if launched, 0 if not covered
static int linear_split_wide_f32(const float* x, const float* w, const float* bias, const float* residual, float* y, long long M,
                                 int N, int K, int epi, hipStream_t st) {
  if (K % 128 != 0 || N % 16 != 0 || M < 4096) return 0;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  const long long ST = ((M + 15) / 16 + 7) / 8;
  // features per pass: 256 (one workgroup per CU, x read once) when the row super-tiles alone give every CU several
  // rounds of work; otherwise 128 (two workgroups per CU, x read once per pass from L2 / the memory-side cache) --
  // measured at 5 x 3680 x 1536 -> 384: 164 us against 211 us
  int passes = (N + 255) / 256;
  if (ST * passes < (n_cu * 3) / 2) passes = (N + 127) / 128;
  if (const int nf = config().linear_wide_nfeat; nf >= 16) passes = (N + nf - 1) / nf;
  const int nfeat = ((N + passes - 1) / passes + 15) / 16 * 16;
  const int NF = nfeat / 16;
  if (NF != 8 && NF != 12 && NF != 16) return 0;
  const unsigned gx = (unsigned)std::min<long long>(ST, std::max(1, n_cu / ((N + nfeat - 1) / nfeat)));
  const size_t lds = (size_t)2 * 4 * nfeat * 3 * 16;
  dim3 grid(gx, (unsigned)((N + nfeat - 1) / nfeat)), block(LS_THREADS);
#define UNIVS_LSW(nf, ep)                                                                                       \
  do {                                                                                                          \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bf16x6_wide<nf, ep>),                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
    hipLaunchKernelGGL((linear_bf16x6_wide<nf, ep>), grid, block, lds, st, x, w, bias, residual, y, (int)M, N, K); \
  } while (0)
#define UNIVS_LSW_EPI(nf)                                        \
  switch (epi) {                                                 \
    case LS_EPI_RELU: UNIVS_LSW(nf, LS_EPI_RELU); break;         \
    case LS_EPI_GELU: UNIVS_LSW(nf, LS_EPI_GELU); break;         \
    case LS_EPI_RESIDUAL: UNIVS_LSW(nf, LS_EPI_RESIDUAL); break; \
    default: UNIVS_LSW(nf, LS_EPI_NONE); break;                  \
  }
  if (NF == 8) { UNIVS_LSW_EPI(8) } else if (NF == 12) { UNIVS_LSW_EPI(12) } else { UNIVS_LSW_EPI(16) }
#undef UNIVS_LSW_EPI
#undef UNIVS_LSW
  const int rc = check_launch("linear_split_wide_f32");
  return rc == UNIVS_OK ? 1 : rc;
}

int linear_f16x3_f32(const float* x, const float* w, const float* bias, const float* residual, float* y, long long M, int N,
                     int K, int epi, hipStream_t st, int blk_rows, int blk_cols);   // linear_f16x3.hip

// returns 1 if launched, 0 if the shape is not covered (the caller uses the library GEMM), < 0 on error
int linear_split_f32(const float* x, const float* w, const float* bias, const float* residual, float* y, long long M, int N,
                     int K, int epi, hipStream_t st, int blk_rows, int blk_cols) {
  if (M <= 0 || N <= 0) return 1;
  if (epi < 0 || epi > LS_EPI_BLOCKED || (epi == LS_EPI_RESIDUAL) != (residual != nullptr)) return 0;
  if (epi == LS_EPI_BLOCKED && (K != 256 || blk_rows < 1 || blk_cols < 4 || blk_cols % 4 != 0 || N % blk_cols != 0 || M % blk_rows != 0))
    return 0;
  const int ring = K % 128 == 0 ? 4 : K % 96 == 0 ? 3 : 0;
  if (K < 96 || ring == 0 || N % 4 != 0) return 0;
  if (M * (long long)N * 4 >= 0x7FFFFFFFLL || M * (long long)K * 4 >= 0x7FFFFFFFLL) return 0;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(residual) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15))
    return 0;
  const int wide_kmin = config().linear_wide_kmin > 0 ? config().linear_wide_kmin : 768;
  // (... except the one K = 768 shape with many rows and few features, Swin stage 2's fc2 at 73 600 x 768 -> 192: 203 us
  // W-stationary against 221 us)
  if (K >= wide_kmin && !(K == 768 && M >= 32768 && N <= 256) && epi != LS_EPI_BLOCKED) {   // x-stationary variant (measured against the W-stationary one at the Swin-T widths: a tie at K = 384,
                    // 133 / 47 / 167 us against 159 / 61 / 228 us for the stage-4 qkv / proj / fc1 at K = 768)
    const int rc = linear_split_wide_f32(x, w, bias, residual, y, M, N, K, epi, st);
    if (rc != 0 || K > 768) return rc;
  }
  if (config().linear_terms != 6) return linear_f16x3_f32(x, w, bias, residual, y, M, N, K, epi, st, blk_rows, blk_cols);   // default: three products
  const long long lds_cap = 160 * 1024 - 2048;   // W slab + bias + the zeroed tail (see the staging loop)
  int r_cap = (int)std::min<long long>(lds_cap / ((long long)K * 6), 16 * LS_MAX_RB);
  r_cap -= r_cap % 4;
  if (r_cap < 16) return 0;
  const int passes = (N + r_cap - 1) / r_cap;
  int rows = (N + passes - 1) / passes;
  rows = (rows + 3) & ~3;
  const int RB = (rows + 15) / 16;
  const long long WT = (M + LS_TILE_M - 1) / LS_TILE_M;
  if (WT < 64) return 0;                                   // too few rows to amortise the split of W
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  // one workgroup per CU over (row ranges x passes); the x extent a multiple of 8 so that the passes of one row range
  // share an XCD (workgroups are dealt to the 8 XCDs round-robin by linear id)
  long long gx = std::max<long long>(1, n_cu / passes);
  gx = std::min(gx, std::max<long long>(1, WT / (2 * (LS_THREADS / 64))));
  // (... unless rounding down to a multiple of 8 would idle more than a tenth of the CUs: 17 passes -> 15 row ranges,
  // not 8; the passes of a row range then sit on different XCDs and x comes from the memory-side cache instead of L2)
  if (gx >= 8 && (gx - gx % 8) * 10 >= gx * 9) gx -= gx % 8;
  const size_t lds = (size_t)K * rows * 6 + 4 * (size_t)rows + 16 * 48 + 16;
  dim3 grid((unsigned)gx, (unsigned)passes), block(LS_THREADS);
#define UNIVS_LS(rb, ksc, rg, ep)                                                                                        \
  do {                                                                                                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_bf16x6<rb, ksc, rg, ep>),                            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                     \
    hipLaunchKernelGGL((linear_bf16x6<rb, ksc, rg, ep>), grid, block, lds, st, x, w, bias, residual, y, (int)M, N, K, rows, \
                       blk_rows, blk_cols);                                                                              \
  } while (0)
#define UNIVS_LS_EPI(rb, ksc, rg)                                   \
  switch (epi) {                                                    \
    case LS_EPI_RELU: UNIVS_LS(rb, ksc, rg, LS_EPI_RELU); break;    \
    case LS_EPI_GELU: UNIVS_LS(rb, ksc, rg, LS_EPI_GELU); break;    \
    case LS_EPI_RESIDUAL: UNIVS_LS(rb, ksc, rg, LS_EPI_RESIDUAL); break; \
    default: UNIVS_LS(rb, ksc, rg, LS_EPI_NONE); break;             \
  }
  // a straight-line tile body for K = 256 (MSDeformAttn), a runtime k loop for anything else (straight-line bodies for
  // the Swin widths 96 .. 768 were measured: no gain, 80 s of compile time)
#define UNIVS_LS_RB(rb)                                               \
  case rb:                                                            \
    if (epi == LS_EPI_BLOCKED) { UNIVS_LS(rb, 8, 4, LS_EPI_BLOCKED); } \
    else if (K == 256) { UNIVS_LS_EPI(rb, 8, 4) }                     \
    else if (ring == 4) { UNIVS_LS_EPI(rb, 0, 4) }                    \
    else { UNIVS_LS_EPI(rb, 0, 3) }                                   \
    break
  switch (RB) {
    UNIVS_LS_RB(1);
    UNIVS_LS_RB(2);
    UNIVS_LS_RB(3);
    UNIVS_LS_RB(4);
    UNIVS_LS_RB(5);
    UNIVS_LS_RB(6);
    default: UNIVS_LS_RB(7);
  }
#undef UNIVS_LS_RB
#undef UNIVS_LS_EPI
#undef UNIVS_LS
  const int rc = check_launch("linear_split_f32");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
