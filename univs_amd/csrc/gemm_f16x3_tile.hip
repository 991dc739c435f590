// Three-product fp16 GEMM (see linear_f16x3.hip for the arithmetic) for wide-K Linears, tiled in TWO dimensions: the
// stage-3 / stage-4 Linears of the Swin blocks (swin.py:35-58: fc2 K = 4 C, proj, qkv, fc1 at C = 768) and the patch-merging
// reductions (swin.py:339-357).
// Why a second kernel beside gemm_f16x3_stream.hip: there a workgroup owns 256 rows x 64 / 128 features and every pass over
// N streams the rows of x again -- 6 passes at 18 400 x 1536 -> 384, 12 at 4 600 x 3072 -> 768.  The bare x stream of those
// passes (tools/probes/row_stride.hip: the loads alone, no LDS, no matrix instructions, passes sharing an XCD) takes 85 of
// the kernel's 147 us: L2 hands the CUs 8 TB/s in the B-operand layout's half-line reads, and the kernel is bound by the
// bytes it pulls through L2 (x: passes x 113 MB + W slabs: 190 MB).  Here a workgroup owns TR = 32 CT rows x TF <= 64 RB
// features (160 x 192 at that shape: 230 workgroups, 2 feature tiles): x is split ONCE per workgroup and shared by the
// waves through LDS, W streams through LDS as before -- 497 MB instead of 870 MB through L2, and a third of the split work.
//   * waves 2 x 4: wave (wr, wf) accumulates rows [wr CT 16, +CT 16) x feature blocks [wf RB, +RB) -- CT x RB accumulator tiles;
//   * per k-step of 32: column tile t of x (16 rows) is loaded, scaled and split by wave t mod 8 (running per-row scale as in
//     linear_f16x3; the row's exponent goes to LDS beside the operand so that the waves holding that row's accumulators
//     rescale them by the same exact power of two), written as the B-fragment image [tile][part][k-group][row]; the W slab
//     of the k-step ([k-group][part][feature] 16-byte units of the pre-split image) is a copy by all 512 threads;
//   * two LDS stages, one barrier per k-step; global loads run NSLOT - 1 k-steps ahead in registers and every load of the
//     loop is unconditional (see gemm_f16x3_stream.hip on what hipcc does to loads under branches);
//   * the same products in the same order per (row, feature) as the other two kernels: results are bit-identical.
#include "common.h"
#include "config.h"
#include "f16x3.h"

#include <algorithm>
#include <cstdlib>

namespace univs {

constexpr int GT_THREADS = 512;
enum { GT_EPI_NONE = 0, GT_EPI_RELU = 1, GT_EPI_GELU = 2, GT_EPI_RESIDUAL = 3 };   // = LS_EPI_*

struct GtArgs {
  const float* X;
  const u32x4* Wp;
  const float* winv;
  const float* bias;
  const float* Res;
  float* Y;
  int M, N, K, epi;
  int tf, nf;                    // features per feature tile (a multiple of 4, <= 64 RB), feature tiles
};

// LDS (16-byte units): Xs[2][2 CT * 128] | Ws[2][8 * 64 RB] | Es[2][32 CT] ints | bias[64 RB] | winv[64 RB] floats
// OCC = 2: two workgroups per CU (<= 128 registers, <= 80 KB of LDS: the small tiles) -- a workgroup's waves run in lockstep between
// the barriers (vector phase, then matrix phase: measured additive), a second workgroup on the CU fills the other pipe
template <int CT, int RB, int NSLOT, int OCC>
__global__ __launch_bounds__(GT_THREADS, OCC) void gemm_f16x3_tile(const GtArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 lds[];
  constexpr int NT = 2 * CT;                                     // column tiles (16 rows) of the workgroup
  constexpr int TR = 32 * CT;
  constexpr int TFp = 64 * RB;
  constexpr int XS = NT * 128;                                   // units per stage of x: [tile][part][k-group][row]
  constexpr int WS = 8 * TFp;                                    // units per stage of W: [k-group][part][feature]
  constexpr int SLOTS = (NT + 7) / 8;                            // column tiles a wave loads and splits
  u32x4* Xs = lds;
  u32x4* Ws = lds + 2 * XS;
  int* Es = reinterpret_cast<int*>(lds + 2 * XS + 2 * WS);
  float* bias_lds = reinterpret_cast<float*>(Es + 2 * TR);
  float* winv_lds = bias_lds + TFp;

  // (row tile, feature tile): every XCD takes a contiguous chunk of the (row tile major, feature tile minor) sequence -- the
  // feature tiles of a row tile read the same rows of x and meet in one L2 (linear_f16x3.hip)
  const unsigned lw = xcd_remap(blockIdx.x, gridDim.x);
  const int bx = (int)(lw / (unsigned)a.nf), by = (int)(lw - (unsigned)bx * (unsigned)a.nf);
  const int M = a.M, N = a.N, K = a.K, epi = a.epi;
  const int row0 = bx * TR, n0 = by * a.tf;
  const int R = min(a.tf, N - n0);                               // a multiple of 4
  const int KS = K >> 5;                                         // a multiple of NSLOT (host-checked)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int wr = wave >> 2, wf = wave & 3;

  for (int r = tid; r < TFp; r += GT_THREADS) {
    bias_lds[r] = (a.bias && r < R) ? a.bias[n0 + r] : 0.f;
    winv_lds[r] = r < R ? a.winv[n0 + r] : 0.f;
  }

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((long long)M * K * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.Wp), 0, (int)((long long)(K >> 3) * 2 * N * 16), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, (int)((long long)M * N * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(epi == GT_EPI_RESIDUAL ? a.Res : a.X), 0, (int)((long long)M * N * 4), 0x00020000);

  // ---- producer side.  x: this wave's column tiles t = wave + 8 sl (rows past the end repeat the last row: never stored)
  unsigned xvo[SLOTS];                                           // byte offset of (row, k-group) in x
  unsigned xdst[SLOTS];                                          // unit of (tile, part 0, k-group, row) in a stage of Xs
  bool xok[SLOTS];                                               // wave-uniform
  int eset[SLOTS];                                               // exponent the scale of my row was set for (linear_f16x3.hip)
  float sx[SLOTS];
#pragma unroll
  for (int sl = 0; sl < SLOTS; ++sl) {
    const int t = wave + 8 * sl;
    xok[sl] = t < NT;
    const int m = min(row0 + min(t, NT - 1) * 16 + j, M - 1);
    xvo[sl] = ((unsigned)m * (unsigned)K + (unsigned)(8 * g)) * 4u;
    xdst[sl] = (unsigned)(min(t, NT - 1) * 128 + g * 16 + j);
    eset[sl] = -1000;
    sx[sl] = 1.0f;
  }
  // W: units tid + 512 v of the k-step's slab (v < RB: 8 runs of 64 RB units); features >= R read out of range = 0
  unsigned wvo[RB];
#pragma unroll
  for (int v = 0; v < RB; ++v) {
    const int i = tid + GT_THREADS * v;
    const int run = i / TFp, rr = i - run * TFp;
    wvo[v] = rr < R ? (unsigned)((run * N + n0 + rr) * 16) : 0xFFFFFFF0u;
  }
  const int wstep = 8 * N * 16;                                  // bytes of the image per k-step
  f32x4 xq[NSLOT][SLOTS][2];
  u32x4 wq[NSLOT][RB];
  auto issue = [&](int slot, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      xq[slot][sl][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo[sl], ks * 128, 0));
      xq[slot][sl][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xvo[sl] + 16u, ks * 128, 0));
    }
#pragma unroll
    for (int v = 0; v < RB; ++v) wq[slot][v] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo[v], ks * wstep, 0));
  };
  // slot -> stage `st` of LDS: W copied, x scaled and split, the rows' exponents beside it
  auto commit = [&](int slot, int st) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < RB; ++v) Ws[st * WS + tid + GT_THREADS * v] = wq[slot][v];
#pragma unroll
    for (int sl = 0; sl < SLOTS; ++sl) {
      const unsigned mk = l3_row_max(l3_absmax8(xq[slot][sl][0], xq[slot][sl][1]));
      const int enew = max(-100, min((int)((mk >> 23) & 255u) - 127, 128));      // 2^e <= max < 2^(e+1)
      if (enew > eset[sl] + 2) {
        eset[sl] = enew;
        sx[sl] = __builtin_bit_cast(float, (unsigned)(127 + 12 - enew) << 23);
      }
      f16x8 bh, bm;
      l3_split8(xq[slot][sl][0], xq[slot][sl][1], sx[sl], bh, bm);
      if (xok[sl]) {                                             // (wave-uniform; no load inside)
        Xs[st * XS + xdst[sl]] = __builtin_bit_cast(u32x4, bh);
        Xs[st * XS + xdst[sl] + 64] = __builtin_bit_cast(u32x4, bm);
        if (g == 0) Es[st * TR + (wave + 8 * sl) * 16 + j] = eset[sl];
      }
    }
  };

  // ---- consumer side
  f32x4 acc[RB][CT];
  int eprev[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    eprev[c] = -1000;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int b_unit = (wr * CT) * 128 + g * 16 + j;               // + c * 128 (+ 64 for the m part)
  const int a_unit = (g * 2) * TFp + wf * RB * 16 + j;           // + rb * 16 (+ TFp for the m part)
  const int e_idx = wr * CT * 16 + j;                            // + c * 16
  auto consume = [&](int st) __attribute__((always_inline)) {
    u32x4 bfr[CT][2];
    int e[CT];
    bool chg = false;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      bfr[c][0] = Xs[st * XS + b_unit + c * 128];
      bfr[c][1] = Xs[st * XS + b_unit + c * 128 + 64];
      e[c] = Es[st * TR + e_idx + c * 16];
      chg = chg || e[c] != eprev[c];
    }
    if (__builtin_amdgcn_ballot_w64(chg) != 0) {                 // rare after the first k-step
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float ratio = __builtin_bit_cast(float, (unsigned)(127 + max(eprev[c] - e[c], -126)) << 23);   // 2^(old - new) <= 1
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][c] *= ratio;
        eprev[c] = e[c];
      }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const f16x8 ah = __builtin_bit_cast(f16x8, Ws[st * WS + a_unit + rb * 16]);
      const f16x8 am = __builtin_bit_cast(f16x8, Ws[st * WS + a_unit + rb * 16 + TFp]);
      // smallest terms first: m*h', h*m', h*h'
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am, __builtin_bit_cast(f16x8, bfr[c][0]), acc[rb][c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, bfr[c][1]), acc[rb][c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < CT; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, __builtin_bit_cast(f16x8, bfr[c][0]), acc[rb][c], 0, 0, 0);
    }
  };

  // ---- pipeline: slot s mod NSLOT holds k-step s; at k-step s the loads of k-step s + NSLOT are requested into the slot the
  // previous k-step emptied, then k-step s + 1 goes to the other LDS stage, then the matrix work of k-step s
#pragma unroll
  for (int u = 0; u < NSLOT; ++u) issue(u, u);
  commit(0, 0);
#pragma unroll 1
  for (int s0 = 0; s0 < KS; s0 += NSLOT) {
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
      const int s = s0 + u;
      __syncthreads();                                           // stage s & 1 is complete; the other one is free
      int ksn = s + NSLOT;                                       // (past the end: wraps to valid addresses; never used)
      ksn = ksn >= KS ? ksn - KS : ksn;
      issue(u, ksn);
      commit((u + 1) % NSLOT, (s + 1) & 1);
      consume(s & 1);
    }
  }

  // ---- epilogue: D[i = feature][j = row]: a lane holds four consecutive features of its rows
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int m = row0 + (wr * CT + c) * 16 + j;
    const bool row_ok = m < M;
    const float sx_inv = __builtin_bit_cast(float, (unsigned)(127 - 12 + eprev[c]) << 23);
    auto out_off = [&](int rb) __attribute__((always_inline)) -> unsigned {
      const int f = (wf * RB + rb) * 16 + 4 * g;
      return (row_ok && f < R) ? ((unsigned)m * (unsigned)N + (unsigned)(n0 + f)) * 4u : 0xFFFFFFF0u;   // out of range: dropped / 0
    };
    f32x4 resv[RB];
    if (epi == GT_EPI_RESIDUAL) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) resv[rb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, out_off(rb), 0, 0));
    } else {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) resv[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int f = (wf * RB + rb) * 16 + 4 * g;
      const f32x4 wi = *reinterpret_cast<const f32x4*>(winv_lds + f);
      const f32x4 bi = *reinterpret_cast<const f32x4*>(bias_lds + f);
      f32x4 v = (acc[rb][c] * sx_inv) * wi + bi;                  // two exact unscalings, then the bias
      if (epi == GT_EPI_RELU) v = __builtin_elementwise_maximum(v, (f32x4){0.f, 0.f, 0.f, 0.f})   /* NaN-propagating, as torch.relu */;
      if (epi == GT_EPI_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = l3_gelu(v[e]);
      }
      if (epi == GT_EPI_RESIDUAL) v += resv[rb];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, out_off(rb), 0, 0);
    }
  }
}

static int gt_cus() {
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

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED when the shape is not covered (the caller then takes gemm_f16x3_stream)
int linear_f16x3_tile_f32(const float* x, const void* wp, const float* winv, const float* bias, const float* residual, float* y,
                          long long M, int N, int K, int epi, hipStream_t st) {
  if (M <= 0 || N <= 0) return UNIVS_OK;
  // k-steps in flight per workgroup (register slots of the loads): 4 where K allows -- with 2 a k-step took ~5 000 clocks at
  // 18 400 x 1536 -> 384, one memory latency under load: 44 KB in flight per CU, against the ~100 KB the L2 -> CU stream needs
  int nslot = K % 128 == 0 ? 4 : K % 96 == 0 ? 3 : 2;
  const UnivsConfig cfg_ = config();
  if (cfg_.linear_ablate >= 7 && cfg_.linear_ablate <= 9 && K % (32 * (cfg_.linear_ablate - 5)) == 0) nslot = cfg_.linear_ablate - 5;   // kernel benchmarks
  if (epi < 0 || epi > GT_EPI_RESIDUAL || (epi == GT_EPI_RESIDUAL) != (residual != nullptr) || K % 64 != 0 || K < 384 ||
      N % 4 != 0 || N < 128 || M < 2048 || M * (long long)N * 4 >= 0x7FFFFFFFLL || M * (long long)K * 4 >= 0x7FFFFFFFLL ||
      (long long)K * N * 4 >= 0x7FFFFFFFLL ||
      (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(wp) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(residual) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(winv) & 15))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  // tile shape: the (CT, RB) with the least estimated time.  Per k-step and workgroup: the matrix pipe (two waves per SIMD), the LDS port
  // (fragment reads of the 8 waves + the stage writes, 128 B / clock) and the L2 -> CU stream (~14 B / clock and CU measured for this
  // access pattern: tools/probes/row_stride.hip).  One workgroup per CU: vector / memory phase and matrix phase add up (measured:
  // 67 + 34 us at 18 400 x 1536 -> 384 on 160 x 192 tiles); two per CU (the small tiles): the slower pipe of the pair.  Workgroups
  // beyond the full rounds run with the CU to themselves.  (tools/gemm_tile_sweep.py measures every shape / depth.)
  const int ncu = gt_cus();
  int best_ct = 0, best_rb = 0, best_nf = 0, best_tf = 0, best_occ = 1;
  double best_t = 1e300;
  for (int ct = 3; ct <= 5; ++ct)
    for (int rb = 2; rb <= 4; ++rb) {
      if (cfg_.linear_grid_x >= 3 && cfg_.linear_grid_x <= 5 && ct != cfg_.linear_grid_x) continue;              // kernel benchmarks
      if (cfg_.linear_rows_per_pass >= 128 && rb != std::min(4, cfg_.linear_rows_per_pass / 64)) continue;
      const int nf = (N + 64 * rb - 1) / (64 * rb);
      int tf = (N + nf - 1) / nf;
      tf = (tf + 3) & ~3;
      if ((tf + 63) / 64 != rb) continue;                        // (a smaller RB covers this split)
      const int occ = ((ct == 3 && rb <= 3) || (ct == 4 && rb == 2)) ? 2 : 1;
      const long long wgs = ((M + 32 * ct - 1) / (32 * ct)) * nf;
      const double ks = K / 32;
      const double mfma = 2.0 * ct * rb * 3 * 16;
      const double ldsc = (8.0 * (2 * ct + 2 * rb) * 1024 + 2 * ct * 2048 + tf * 128) / 128.0;
      const double mem = (32.0 * ct * 128 + tf * 128) / 14.0;
      const double t_alone = 1.1 * ks * (std::max(ldsc, mem) + mfma) + 6000.0;    // (1.1: 101 us measured against 93 modelled)
      const double t_full = occ == 2 ? ks * 2.0 * std::max(mfma, std::max(ldsc, mem)) + 6000.0 : t_alone;
      const long long full = wgs / ((long long)ncu * occ), rem = wgs - full * ncu * occ;
      const double t = full * t_full + (rem == 0 ? 0.0 : rem > ncu ? t_full : t_alone);
      if (t < best_t) { best_t = t; best_ct = ct; best_rb = rb; best_nf = nf; best_tf = tf; best_occ = (occ == 2 && wgs > ncu) ? 2 : 1; }
    }
  if (best_ct == 0) return UNIVS_ERR_NOT_IMPLEMENTED;
  if (best_ct == 5 && best_rb == 4 && nslot == 4) nslot = K % 96 == 0 ? 3 : 2;                                    // (registers)
  // (two workgroups per CU only where there are more workgroups than CUs; otherwise the same tile with the deeper load pipeline:
  //  4 600 x 3072 -> 768 on 128 x 128 tiles, 216 workgroups: 95 us with 3-4 k-steps in flight, 105 with 2)
  if (best_occ == 2) nslot = 2;                                                                                    // (128 registers)
  GtArgs a{};
  a.X = x; a.Wp = reinterpret_cast<const u32x4*>(wp); a.winv = winv; a.bias = bias; a.Res = residual; a.Y = y;
  a.M = (int)M; a.N = N; a.K = K; a.epi = epi; a.tf = best_tf; a.nf = best_nf;
  const long long rt = (M + 32 * best_ct - 1) / (32 * best_ct);
  const dim3 grid((unsigned)(rt * best_nf)), block(GT_THREADS);
  const size_t lds = ((size_t)2 * (2 * best_ct * 128) + (size_t)2 * (8 * 64 * best_rb)) * 16 + (size_t)2 * 32 * best_ct * 4 +
                     (size_t)2 * 64 * best_rb * 4;
#define UNIVS_GT_K(ct, rb, ns, oc)                                                                                    \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f16x3_tile<ct, rb, ns, oc>),                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
    hipLaunchKernelGGL((gemm_f16x3_tile<ct, rb, ns, oc>), grid, block, lds, st, a);                                   \
  } while (0)
#define UNIVS_GT_RB(ct, rb)                                                        \
  do {                                                                             \
    constexpr int oc = ((ct == 3 && rb <= 3) || (ct == 4 && rb == 2)) ? 2 : 1;     \
    if (oc == 2 && best_occ == 2) UNIVS_GT_K(ct, rb, 2, oc);                       \
    else if (nslot == 4) UNIVS_GT_K(ct, rb, 4, 1);                                 \
    else if (nslot == 3) UNIVS_GT_K(ct, rb, 3, 1);                                 \
    else UNIVS_GT_K(ct, rb, 2, 1);                                                 \
  } while (0)
#define UNIVS_GT_CT(ct)                            \
  case ct:                                         \
    if (best_rb == 2) UNIVS_GT_RB(ct, 2);          \
    else if (best_rb == 3) UNIVS_GT_RB(ct, 3);     \
    else UNIVS_GT_RB(ct, 4);                       \
    break
  switch (best_ct) {
    UNIVS_GT_CT(3);
    UNIVS_GT_CT(4);
    default: UNIVS_GT_CT(5);
  }
#undef UNIVS_GT_RB
#undef UNIVS_GT_CT
#undef UNIVS_GT_K
  return check_launch("gemm_f16x3_tile");
}

}  // namespace univs
