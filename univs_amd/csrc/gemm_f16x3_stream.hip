// Three-product fp16 GEMM (see linear_f16x3.hip for the arithmetic) for the shapes whose W does not fit LDS:
//   * wide-K Linears (K >= 768: the encoder's second FFN Linear, K = 1024 -> 256, msdeformattn.py:87-91; fc2 / stage-4
//     Linears of the Swin blocks, swin.py:35-58);
//   * the 3 x 3 convolution of the FPN output (msdeformattn.py:227-232, :352: 256 -> 256 at 1/4 resolution, K = 9 * 256), as
//     the same GEMM with tap addressing of x: M = T * H * W pixels, a k-step of 32 = 32 input channels of one tap.
// W is split ONCE (presplit_f16x3: row maxima, power-of-two row scales, two fp16 parts; the host side caches the result per
// weight tensor) into the very image the kernel keeps in LDS -- [k-step][k-group][part][feature] 16-byte units -- so that
// staging it is a copy: the six-product kernels (linear_split.hip) re-split the W slab in every workgroup for every 128
// rows of x, which cost more vector work than the split of x itself.
// A workgroup = 8 waves = 256 rows of x (32 per wave, two MFMA column tiles) x 128 output features per pass; W streams
// through two LDS buffers in groups of RING k-steps (64 KB each at 128 features), one barrier per group; x keeps the
// running per-row scale of linear_f16x3.
#include "common.h"
#include "config.h"
#include "f16x3.h"

#include <algorithm>
#include <cstdlib>

namespace univs {

#ifdef UNIVS_TRACE_GEMM
UNIVS_GT_DECL(g_gs_trace);
#endif
#ifndef UNIVS_GS_THREADS
#define UNIVS_GS_THREADS 512     // (256: timing experiment `--ablate gs256` -- two 4-wave workgroups per CU where their W buffers fit)
#endif
constexpr int GS_THREADS = UNIVS_GS_THREADS;
constexpr int GS_TILE_M = 32;
enum { GS_EPI_NONE = 0, GS_EPI_RELU = 1, GS_EPI_GELU = 2, GS_EPI_RESIDUAL = 3 };   // = LS_EPI_*

// ---- W [N, K] fp32 -> Wp [(K/32) * 4 * 2][N] 16-byte units + winv [N].  conv: W is [N, Cin, 3, 3] and k = tap * Cin + ci.
// kperm: the k-order inside a 32-wide k-step is (4 g + e, 16 + 4 g + e) for k-group g, e = 0..3 -- the order in which the
// fused MLP (mlp_f16x3.hip) holds its hidden activations when they come out of the first product's accumulators.
__global__ __launch_bounds__(256) void presplit_f16x3_kernel(const float* __restrict__ W, int N, int K, int conv_cin, int kperm,
                                                             u32x4* __restrict__ Wp, float* __restrict__ winv) {
  __shared__ unsigned smax;
  const int r = blockIdx.x;
  if (threadIdx.x == 0) smax = 0u;
  __syncthreads();
  const float* src = W + (size_t)r * K;
  auto elem = [&](int k) __attribute__((always_inline)) -> float {
    if (conv_cin == 0) return src[k];
    const int tap = k / conv_cin, ci = k - tap * conv_cin;
    return src[ci * 9 + tap];
  };
  unsigned mx = 0u;
  for (int k = threadIdx.x; k < K; k += blockDim.x) mx = max(mx, __builtin_bit_cast(unsigned, fabsf(elem(k))));
  atomicMax(&smax, mx);
  __syncthreads();
  float s, inv;
  l3_scale(smax, 14, s, inv);
  if (threadIdx.x == 0) winv[r] = inv;
  for (int kc = threadIdx.x; kc < (K >> 3); kc += blockDim.x) {
    f32x4 v0, v1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k0 = kperm ? (kc >> 2) * 32 + 4 * (kc & 3) + e : kc * 8 + e;
      v0[e] = elem(k0);
      v1[e] = elem(kperm ? k0 + 16 : k0 + 4);
    }
    f16x8 h, m;
    l3_split8(v0, v1, s, h, m);
    Wp[(size_t)(kc * 2) * N + r] = __builtin_bit_cast(u32x4, h);
    Wp[(size_t)(kc * 2 + 1) * N + r] = __builtin_bit_cast(u32x4, m);
  }
}

int presplit_f16x3(const float* w, int N, int K, int conv_cin, int kperm, void* wp, float* winv, hipStream_t st) {
  if (N <= 0) return UNIVS_OK;
  hipLaunchKernelGGL(presplit_f16x3_kernel, dim3(N), dim3(256), 0, st, w, N, K, conv_cin, kperm, reinterpret_cast<u32x4*>(wp), winv);
  return check_launch("presplit_f16x3");
}

struct GsArgs {
  const float* X;
  const u32x4* Wp;
  const float* winv;
  const float* bias;
  const float* Res;
  float* Y;
  int M, N, K, rows_per_pass, epi;
  int Cin, Cout, H, Wd, HW;      // conv (XMODE 1): X [T, Cin, H, W], Y [T, Cout, H, W]; XMODE 2: X [T, H, W, Cin] (channels last), Y as XMODE 1
  int tap0;                      // conv: first tap of the kernel (0: all nine taps of a 3 x 3; 4: the centre alone = a 1 x 1)
  int remap;                     // XCD-aware (row range, pass) order (UnivsConfig.linear_ablate == 5 switches it off: A/B)
};

// LDS: 2 x [RING][4 k-groups][2 parts][16 RB] 16 B | bias[Rp] | winv[Rp]
template <int RB, int RING, int XMODE>
__global__ __launch_bounds__(GS_THREADS, 512 / GS_THREADS) void gemm_f16x3_stream(const GsArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 Wst[];
  [[maybe_unused]] const int gts = UNIVS_GT_SLOT();
  UNIVS_GT(g_gs_trace, gts, 0);
  UNIVS_GT_REAL(g_gs_trace, gts, 62);
  constexpr int Rp = 16 * RB;
  constexpr int SLAB = RING * 4 * 2 * Rp;                        // 16-byte units per buffer
  // (row range, pass): every XCD takes a contiguous chunk of the (row range major, pass minor) sequence -- see linear_f16x3.hip
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (a.remap && gridDim.y > 1) {
    const unsigned lw = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    bx = lw / gridDim.y;
    by = lw - bx * gridDim.y;
  }
  const int n0 = by * a.rows_per_pass;
  const int R = min(a.rows_per_pass, a.N - n0);                  // a multiple of 4 (conv: of 16)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, N = a.N, K = a.K, epi = a.epi;
  const int KS = K >> 5, KG = KS / RING;
  const int j = lane & 15, g = lane >> 4;
  float* bias_lds = reinterpret_cast<float*>(Wst + 2 * SLAB);
  float* winv_lds = bias_lds + Rp;
  constexpr int NWV = GS_THREADS / 64;

  const int WT = (M + GS_TILE_M - 1) / GS_TILE_M;
  const int wg0 = (int)((long long)WT * bx / gridDim.x), wg1 = (int)((long long)WT * (bx + 1) / gridDim.x);
  const int rounds = (wg1 - wg0 + NWV - 1) / NWV;                // every wave runs all rounds (barriers); idle tiles store nothing
  if (rounds == 0) return;

  for (int r = tid; r < Rp; r += GS_THREADS) {
    bias_lds[r] = (a.bias && r < R) ? a.bias[n0 + r] : 0.f;
    winv_lds[r] = r < R ? a.winv[n0 + r] : 0.f;
  }

  // ---- W: copy of the pre-split image.  Slab q of this pass = RING * 8 runs of R consecutive 16-byte units; a thread moves
  // units tid + 512 v, two per k-step of the group before (fetch at stage u, commit at stage u + 1).
  // (The fetches sit under lane conditions, and hipcc answers loads under branches with `s_waitcnt vmcnt(0)` in front of every commit:
  // nothing is ever more than a k-step ahead.  A branch-free form with a full group of lead -- round 5, profiles/r05_gemm_tile_v1.txt --
  // measured the same at 64 features per pass and 10 % WORSE for the convolution (24 more registers at 128 features: spills), because
  // what bounds this kernel is the L2 -> CU stream of the passes' x re-reads, not the latency of the slab: the wide-K Linears went to
  // gemm_f16x3_tile.hip instead and this form stayed.)
  constexpr int UNITS = RING * 8 * Rp;                           // per slab (rows >= R of a short last pass are skipped)
  constexpr int UPT = (UNITS + GS_THREADS - 1) / GS_THREADS;     // units per thread and slab (8 at 128 features)
  constexpr int UPS = (UPT + RING - 1) / RING;                   // per stage
  u32x4 wreg[UPS];
  auto w_fetch = [&](int q, int u) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < UPS; ++v) {
      const int i = tid + GS_THREADS * (u * UPS + v);
      const int run = i / Rp, rr = i - run * Rp;
      wreg[v] = (u32x4){0u, 0u, 0u, 0u};
      if (u * UPS + v < UPT && i < UNITS && rr < R) wreg[v] = a.Wp[(size_t)(q * RING * 8 + run) * N + n0 + rr];
    }
  };
  auto w_commit = [&](int buf, int u) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < UPS; ++v) {
      const int i = tid + GS_THREADS * (u * UPS + v);
      if (u * UPS + v < UPT && i < UNITS) Wst[buf * SLAB + i] = wreg[v];
    }
  };
#pragma unroll
  for (int u = 0; u < RING; ++u) {                               // slab 0 -> buffer 0
    w_fetch(0, u);
    w_commit(0, u);
  }

  // ---- x
  const long long xbytes = XMODE == 0 ? (long long)M * K * 4 : (long long)(M / a.HW) * a.Cin * a.HW * 4;   // (XMODE 1 and 2: the same bytes)
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)xbytes, 0x00020000);
  const long long ybytes = XMODE == 0 ? (long long)M * N * 4 : (long long)(M / a.HW) * a.Cout * a.HW * 4;
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, (int)ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(epi == GS_EPI_RESIDUAL ? a.Res : a.X), 0, (int)ybytes, 0x00020000);

  // per tile and column tile: XMODE 0: byte offset of (row, k-group); XMODE 1: of (frame, channel 8 g, pixel) + pixel coords
  struct TileRows {
    unsigned vo[2];
    int py[2], px[2];
  };
  auto tile_rows = [&](int tt, TileRows& t) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int m = min(tt * GS_TILE_M + 16 * c + j, M - 1);    // rows past the end repeat the last row (not stored)
      if (XMODE == 0) {
        t.vo[c] = ((unsigned)m * (unsigned)K + (unsigned)(8 * g)) * 4u;
        t.py[c] = t.px[c] = 0;
      } else if (XMODE == 2) {
        const int f = m / a.HW, rem = m - f * a.HW;
        t.py[c] = rem / a.Wd;
        t.px[c] = rem - t.py[c] * a.Wd;
        t.vo[c] = ((unsigned)m * (unsigned)a.Cin + (unsigned)(8 * g)) * 4u;   // pixel m, channels 8 g ..: 32 contiguous bytes
      } else {
        const int f = m / a.HW, rem = m - f * a.HW;
        t.py[c] = rem / a.Wd;
        t.px[c] = rem - t.py[c] * a.Wd;
        t.vo[c] = (unsigned)((f * a.Cin + 8 * g) * a.HW + rem) * 4u;
      }
    }
  };
  f32x4 raw[RING][2][2];                                         // [stage][column tile][8 k-values]
  const int kspt = XMODE != 0 ? a.Cin >> 5 : 1;                  // k-steps per tap
  auto load_x = [&](f32x4 (&buf)[2][2], const TileRows& t, int ks) __attribute__((always_inline)) {
    if (XMODE == 0) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        buf[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, t.vo[c], ks * 128, 0));
        buf[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, t.vo[c] + 16u, ks * 128, 0));
      }
    } else if (XMODE == 2) {
      // channels last: 8 input channels of tap ks / kspt at my (shifted) pixel are 32 contiguous bytes -- two 16-byte loads per
      // column tile where the NCHW operand takes eight 4-byte loads; taps outside the image read 0 (offset out of range)
      const int tap_ = ks / kspt, cb = (ks - tap_ * kspt) * 32;  // uniform
      const int tap = tap_ + a.tap0;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bool inb = (unsigned)(t.py[c] + dy) < (unsigned)a.H && (unsigned)(t.px[c] + dx) < (unsigned)a.Wd;
        const unsigned vo = inb ? t.vo[c] + (unsigned)((dy * a.Wd + dx) * a.Cin * 4) : 0xFFFFFFE0u;
        buf[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, cb * 4, 0));
        buf[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo + 16u, cb * 4, 0));
      }
    } else {
      // 8 input channels (a plane apart) of tap ks / kspt at my pixel; taps outside the image read 0 (offset out of range)
      const int tap_ = ks / kspt, cb = (ks - tap_ * kspt) * 32;  // uniform
      const int tap = tap_ + a.tap0;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const bool inb = (unsigned)(t.py[c] + dy) < (unsigned)a.H && (unsigned)(t.px[c] + dx) < (unsigned)a.Wd;
        const unsigned vo = inb ? t.vo[c] + (unsigned)((dy * a.Wd + dx) * 4) : 0xFFFFFFF0u;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          buf[c][e >> 2][e & 3] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, vo, (cb + e) * a.HW * 4, 0));
      }
    }
  };

  f32x4 acc[RB][2];
  int eset[2];
  float sx[2], sx_inv[2];
  constexpr int NB = RB > 6 ? 4 : 2;
  constexpr int BSZ = (RB + NB - 1) / NB;
  u32x4 afr[2][BSZ][2];
  const unsigned a_lane = (unsigned)((g * 2 * Rp + j) * 16);
  const unsigned a_kstep = (unsigned)(8 * Rp * 16);
  const unsigned a_part = (unsigned)(Rp * 16);
  // (the block and part offsets of a read are immediates of the instruction: linear_f16x3.hip)
  auto read_batch = [&](u32x4 (&d)[BSZ][2], unsigned base, auto bc) __attribute__((always_inline)) {
    constexpr int b = decltype(bc)::value;
    l3_static_for<0, BSZ>([&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      constexpr int rb = b * BSZ + q < RB ? b * BSZ + q : RB - 1;
      u32x4& dh = d[q][0];
      u32x4& dm = d[q][1];
      const unsigned aa = base;
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                   : "=&v"(dh), "=&v"(dm)
                   : "v"(aa), "n"(rb * 256), "n"(rb * 256 + Rp * 16)
                   : "memory");
    });
  };
  auto wait_batch = [&](u32x4 (&d)[BSZ][2]) __attribute__((always_inline)) {
    if constexpr (BSZ == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0][0]), "+v"(d[0][1]) : : "memory");
    else if constexpr (BSZ == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[2][0]), "+v"(d[2][1]) : : "memory");
  };

  TileRows t_cur, t_next;
  tile_rows(wg0 + wave, t_cur);
  tile_rows(wg0 + min(1, rounds - 1) * NWV + wave, t_next);
#pragma unroll
  for (int u = 0; u < RING; ++u) load_x(raw[u], t_cur, u);

  UNIVS_GT(g_gs_trace, gts, 2);
  UNIVS_GT_VAL(g_gs_trace, gts, 63, rounds);
  int gq = 0;                                                    // k-groups done: slab gq is in buffer gq & 1
#pragma unroll 1
  for (int rd = 0; rd < rounds; ++rd) {
    const int tt = wg0 + rd * NWV + wave;
    const bool tile_ok = tt < wg1;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    eset[0] = eset[1] = -1000;
    sx[0] = sx[1] = sx_inv[0] = sx_inv[1] = 1.0f;
#pragma unroll 1
    for (int q = 0; q < KG; ++q, ++gq) {
      __syncthreads();                                           // slab gq is complete; the other buffer is free
      const int bufc = gq & 1, qn = (q + 1 == KG) ? 0 : q + 1;
      const bool last_group = q + 1 == KG;
      const unsigned a_buf = a_lane + (unsigned)(bufc * SLAB * 16);
      read_batch(afr[0], a_buf, std::integral_constant<int, 0>{});
#pragma unroll
      for (int u = 0; u < RING; ++u) {
        if (u > 0) w_commit(bufc ^ 1, u - 1);
        // ---- the running row scale (linear_f16x3.hip)
        bool need = false;
        int enew[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const unsigned mk = l3_row_max(l3_absmax8(raw[u][c][0], raw[u][c][1]));
          enew[c] = max(-100, min((int)((mk >> 23) & 255u) - 127, 128));
          need = need || (enew[c] > eset[c] + 2);
        }
        if (__builtin_amdgcn_ballot_w64(need) != 0) {
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const bool mine = enew[c] > eset[c] + 2;
            const int en = mine ? enew[c] : eset[c];
            const float ratio = __builtin_bit_cast(float, (unsigned)(127 + max(eset[c] - en, -126)) << 23);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) acc[rb][c] *= ratio;
            eset[c] = en;
            sx[c] = __builtin_bit_cast(float, (unsigned)(127 + 12 - en) << 23);
            sx_inv[c] = __builtin_bit_cast(float, (unsigned)(127 - 12 + en) << 23);
          }
        }
        float s0 = sx[0], s1 = sx[1];
        asm volatile("" : "+v"(s0), "+v"(s1) : : "memory");
        f16x8 bh[2], bm[2];
        l3_split8(raw[u][0][0], raw[u][0][1], s0, bh[0], bm[0]);
        l3_split8(raw[u][1][0], raw[u][1][1], s1, bh[1], bm[1]);
        __builtin_amdgcn_sched_barrier(0);
        w_fetch(qn, u);                                          // the next slab, two units per stage
        if (last_group) load_x(raw[u], t_next, u);               // the ring runs RING k-steps ahead, across tiles
        else load_x(raw[u], t_cur, q * RING + u + RING);
        __builtin_amdgcn_sched_barrier(0);
        l3_static_for<0, NB>([&](auto bc) __attribute__((always_inline)) {
          constexpr int b = decltype(bc)::value;
          wait_batch(afr[b & 1]);
          if constexpr (b + 1 < NB) read_batch(afr[(b + 1) & 1], a_buf + (unsigned)u * a_kstep, std::integral_constant<int, b + 1>{});
          else if (u + 1 < RING) read_batch(afr[0], a_buf + (unsigned)(u + 1) * a_kstep, std::integral_constant<int, 0>{});
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int qq = 0; qq < BSZ; ++qq) {
            const int rb = b * BSZ + qq;
            if (rb < RB) {
              const f16x8 ah = __builtin_bit_cast(f16x8, afr[b & 1][qq][0]);
              const f16x8 am = __builtin_bit_cast(f16x8, afr[b & 1][qq][1]);
              acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh[0], acc[rb][0], 0, 0, 0);
              acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, bh[1], acc[rb][1], 0, 0, 0);
              acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm[0], acc[rb][0], 0, 0, 0);
              acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bm[1], acc[rb][1], 0, 0, 0);
              acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[0], acc[rb][0], 0, 0, 0);
              acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[1], acc[rb][1], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      w_commit(bufc ^ 1, RING - 1);
    }
    UNIVS_GT(g_gs_trace, gts, 3 + 2 * rd);
    // ---- epilogue: D[i = feature][j = row]: a lane holds four consecutive features of its two rows
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int m = tt * GS_TILE_M + 16 * c + j;
      const bool row_ok = tile_ok && m < M;
      int fr = 0, rem = 0;
      if (XMODE != 0) {
        fr = min(m, M - 1) / a.HW;
        rem = min(m, M - 1) - fr * a.HW;
      }
      // the residual rows of this column tile, requested in batches (a load inside the per-block loop is compiled into load / wait /
      // store -- one memory latency per block)
      constexpr int RBAT = RB > 4 ? (RB + 1) / 2 : RB;           // (in two halves at RB > 4: registers)
      [[maybe_unused]] f32x4 resv[RBAT];
      auto load_res = [&](int rb0) __attribute__((always_inline)) {
        if (XMODE == 0 && epi == GS_EPI_RESIDUAL) {
#pragma unroll
          for (int q = 0; q < RBAT; ++q) {
            const int f = min(rb0 + q, RB - 1) * 16 + 4 * g;
            const unsigned off = ((unsigned)m * (unsigned)N + (unsigned)(n0 + f)) * 4u;
            resv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, (row_ok && f < R) ? off : 0xFFFFFFF0u, 0, 0));
          }
        } else {
#pragma unroll
          for (int q = 0; q < RBAT; ++q) resv[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      };
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (rb % RBAT == 0) load_res(rb);
        const int f = rb * 16 + 4 * g;
        const f32x4 wi = *reinterpret_cast<const f32x4*>(winv_lds + f);
        const f32x4 bi = *reinterpret_cast<const f32x4*>(bias_lds + f);
        f32x4 v = (acc[rb][c] * sx_inv[c]) * wi + bi;
        if (XMODE == 0) {
          const unsigned off = ((unsigned)m * (unsigned)N + (unsigned)(n0 + f)) * 4u;
          const unsigned offc = (row_ok && f < R) ? off : 0xFFFFFFF0u;
          if (epi == GS_EPI_RELU) v = __builtin_elementwise_maximum(v, (f32x4){0.f, 0.f, 0.f, 0.f})   /* NaN-propagating, as torch.relu */;
          if (epi == GS_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = l3_gelu(v[e]);
          }
          if (epi == GS_EPI_RESIDUAL) v += resv[rb % RBAT];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, offc, 0, 0);
        } else {
          // NCHW: four channel planes; out-of-range lanes move their offset out of the buffer
          const unsigned off0 = (row_ok && f < R) ? (unsigned)((fr * a.Cout + n0 + f) * a.HW + rem) * 4u
                                                  : 0xFFFFFFF0u - 3u * (unsigned)a.HW * 4u;
          float vx = v.x, vy = v.y, vz = v.z, vw = v.w;
          asm volatile("" : "+v"(vx), "+v"(vy), "+v"(vz), "+v"(vw));   // (hipcc 7.2 stored the first element four times without this)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vx), yrs, off0, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vy), yrs, off0 + (unsigned)a.HW * 4u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vz), yrs, off0 + 2u * (unsigned)a.HW * 4u, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vw), yrs, off0 + 3u * (unsigned)a.HW * 4u, 0, 0);
        }
      }
    }
    UNIVS_GT(g_gs_trace, gts, 4 + 2 * rd);
    t_cur = t_next;
    tile_rows(wg0 + min(rd + 2, rounds - 1) * NWV + wave, t_next);
  }
  UNIVS_GT_REAL(g_gs_trace, gts, 61);
}

static int gs_cus() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  return n_cu;
}

template <int XMODE>
static int gs_launch(const GsArgs& a0, int ring, hipStream_t st) {
  GsArgs a = a0;
  const UnivsConfig cfg_ = config();
  // output features per pass: 128, or 64 for short tall-K problems with a narrow output (Swin stage-3 / stage-4 proj and fc2:
  // few row tiles, N <= 768 <= K -- twice the passes fill the CUs; 172 -> 126 us at 18 400 x 1536 -> 384, 60 -> 43 us at
  // 18 400 x 384 -> 384: profiles/r04_kbench_smallm_v1.txt)
  const bool narrow = XMODE == 0 && a.N <= 768 && a.K >= a.N && a.M <= 32768;
  const int r_cap = cfg_.linear_rows_per_pass >= 16 ? std::min(128, cfg_.linear_rows_per_pass - cfg_.linear_rows_per_pass % 16)
                                                    : (narrow ? 64 : 128);
  const int passes = (a.N + r_cap - 1) / r_cap;
  int rows = (a.N + passes - 1) / passes;
  rows = (rows + 3) & ~3;
  if (XMODE != 0) rows = (rows + 15) & ~15;
  const int RB = (rows + 15) / 16;
  a.rows_per_pass = rows;
  a.remap = cfg_.linear_ablate == 5 ? 0 : 1;
  const long long WT = ((long long)a.M + GS_TILE_M - 1) / GS_TILE_M;
  long long gx = std::max<long long>(1, gs_cus() / passes);
  gx = std::min(gx, std::max<long long>(1, WT / (GS_THREADS / 64)));
  if (gx >= 8 && (gx - gx % 8) * 10 >= gx * 9) gx -= gx % 8;     // the passes of a row range share an XCD (linear_f16x3.hip)
  if (cfg_.linear_grid_x > 0) gx = std::min<long long>(cfg_.linear_grid_x, WT);
  const size_t lds = (size_t)2 * ring * 8 * (16 * RB) * 16 + 8 * (size_t)(16 * RB);
  if (GS_THREADS < 512 && cfg_.linear_grid_x <= 0 && lds * (512 / GS_THREADS) <= 156 * 1024)      // (experiment: several workgroups per CU)
    gx = std::min<long long>(gx * (512 / GS_THREADS), std::max<long long>(1, WT / (GS_THREADS / 64)));
  dim3 grid((unsigned)gx, (unsigned)passes), block(GS_THREADS);
#define UNIVS_GS(rb, rg)                                                                                          \
  do {                                                                                                            \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_stream<rb, rg, XMODE>),                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
    hipLaunchKernelGGL((gemm_f16x3_stream<rb, rg, XMODE>), grid, block, lds, st, a);                              \
  } while (0)
#define UNIVS_GS_RB(rb)                    \
  case rb:                                 \
    if (ring == 4) { UNIVS_GS(rb, 4); }    \
    else { UNIVS_GS(rb, 3); }              \
    break
  switch (RB) {
    UNIVS_GS_RB(1);
    UNIVS_GS_RB(2);
    UNIVS_GS_RB(3);
    UNIVS_GS_RB(4);
    UNIVS_GS_RB(5);
    UNIVS_GS_RB(6);
    UNIVS_GS_RB(7);
    default: UNIVS_GS_RB(8);
  }
#undef UNIVS_GS_RB
#undef UNIVS_GS
  return check_launch("gemm_f16x3_stream");
}

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED when the shape is not covered
int linear_f16x3_stream_f32(const float* x, const void* wp, const float* winv, const float* bias, const float* residual, float* y,
                            long long M, int N, int K, int epi, hipStream_t st) {
  if (M <= 0 || N <= 0) return UNIVS_OK;
  const int ring = K % 128 == 0 ? 4 : K % 96 == 0 ? 3 : 0;
  if (epi < 0 || epi > GS_EPI_RESIDUAL || (epi == GS_EPI_RESIDUAL) != (residual != nullptr) || ring == 0 || K < 96 || N % 4 != 0 ||
      M < 2048 || M * (long long)N * 4 >= 0x7FFFFFFFLL || M * (long long)K * 4 >= 0x7FFFFFFFLL ||
      (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wp) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(residual) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(winv) & 15))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  GsArgs a{};
  a.X = x; a.Wp = reinterpret_cast<const u32x4*>(wp); a.winv = winv; a.bias = bias; a.Res = residual; a.Y = y;
  a.M = (int)M; a.N = N; a.K = K; a.epi = epi;
  return gs_launch<0>(a, ring, st);
}

int conv3x3_f16x3_f32(const float* x, const void* wp, const float* winv, float* y, int T, int Cin, int Cout, int H, int W,
                      hipStream_t st) {
  if (T <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UNIVS_OK;
  const long long M = (long long)T * H * W;
  if (Cin % 128 != 0 || Cout % 16 != 0 || M < 4096 || M * std::max(Cin, Cout) * 4 >= 0x7FFFFFFFLL ||
      (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wp) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(winv) & 15))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  GsArgs a{};
  a.X = x; a.Wp = reinterpret_cast<const u32x4*>(wp); a.winv = winv; a.bias = nullptr; a.Res = nullptr; a.Y = y;
  a.M = (int)M; a.N = Cout; a.K = 9 * Cin; a.epi = GS_EPI_NONE;
  a.Cin = Cin; a.Cout = Cout; a.H = H; a.Wd = W; a.HW = H * W;
  a.tap0 = 0;
  return gs_launch<1>(a, 4, st);
}

// the same convolution on a CHANNELS-LAST operand x [T, H, W, Cin] (output NCHW as above): see XMODE 2 in the kernel
int conv3x3_nhwc_f16x3_f32(const float* x, const void* wp, const float* winv, float* y, int T, int Cin, int Cout, int H, int W,
                           hipStream_t st) {
  if (T <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UNIVS_OK;
  const long long M = (long long)T * H * W;
  if (Cin % 128 != 0 || Cout % 16 != 0 || M < 4096 || M * std::max(Cin, Cout) * 4 >= 0x7FFFFFFFLL ||
      (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wp) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(winv) & 15))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  GsArgs a{};
  a.X = x; a.Wp = reinterpret_cast<const u32x4*>(wp); a.winv = winv; a.bias = nullptr; a.Res = nullptr; a.Y = y;
  a.M = (int)M; a.N = Cout; a.K = 9 * Cin; a.epi = GS_EPI_NONE;
  a.Cin = Cin; a.Cout = Cout; a.H = H; a.Wd = W; a.HW = H * W;
  a.tap0 = 0;
  return gs_launch<2>(a, 4, st);
}

// y = conv2d(x, w [Cout, Cin, 1, 1], bias) on NCHW tensors: the same kernel with the centre tap alone (K = Cin; w pre-split as a
// Linear's [Cout, Cin]).  The lateral / mask-feature / input-projection convolutions of the pixel decoder
// (msdeformattn.py:214-232, :262-283): the library runs them as fp32 GEMMs at ~110 TF/s and adds the bias in a second pass.
int conv1x1_f16x3_f32(const float* x, const void* wp, const float* winv, const float* bias, float* y, int T, int Cin, int Cout, int H,
                      int W, hipStream_t st) {
  if (T <= 0 || Cout <= 0 || H <= 0 || W <= 0) return UNIVS_OK;
  const long long M = (long long)T * H * W;
  const int ring = Cin % 128 == 0 ? 4 : Cin % 96 == 0 ? 3 : 0;
  if (ring == 0 || Cout % 16 != 0 || M < 4096 || M * std::max(Cin, Cout) * 4 >= 0x7FFFFFFFLL ||
      (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wp) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(winv) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  GsArgs a{};
  a.X = x; a.Wp = reinterpret_cast<const u32x4*>(wp); a.winv = winv; a.bias = bias; a.Res = nullptr; a.Y = y;
  a.M = (int)M; a.N = Cout; a.K = Cin; a.epi = GS_EPI_NONE;
  a.Cin = Cin; a.Cout = Cout; a.H = H; a.Wd = W; a.HW = H * W;
  a.tap0 = 4;
  return gs_launch<1>(a, ring, st);
}

}  // namespace univs

#ifdef UNIVS_TRACE_GEMM
extern "C" int univs_debug_gemm_trace_stream(unsigned long long* out, int clear) {
  if (clear) {
    static unsigned long long zeros[UNIVS_GT_SLOTS * UNIVS_GT_STAMPS] = {};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(univs::g_gs_trace), zeros, sizeof(zeros));
  }
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(univs::g_gs_trace), sizeof(unsigned long long) * UNIVS_GT_SLOTS * UNIVS_GT_STAMPS);
}
#endif
