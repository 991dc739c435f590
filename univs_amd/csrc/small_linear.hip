// y[M, N] = act((x [+ xadd]) W^T + b) [+ residual] [-> LayerNorm] for FEW rows (M <= a few thousand): the per-token Linears of the
// UniVS decoder on its Q' T = 500 ... 2 000 query tokens (univs/modeling/transformer_decoder/transformer_layers.py: the in / out
// projections of SelfAttentionLayer :30-46 and CrossAttentionLayer :95-115, FFNLayer :150-166, the mask-embedding MLP :205-217).
// The library runs each of them as a tuned GEMM of 8 - 13 us (launch- and latency-bound: 33 MFLOP) with the elementwise steps around
// it -- `tgt + query_pos` in front, `norm(tgt + .)` behind -- as launches of their own: ~100 GEMMs + ~60 elementwise / LayerNorm
// launches per clip.
//
// Arithmetic: the three-product fp16 scheme of linear_f16x3.hip (two fp16 parts per operand behind power-of-two row scales, three
// v_mfma_f32_16x16x32_f16 per fp32 product, fp32 accumulation: ~2^-22 per product); W pre-split once per weight tensor
// (gemm_f16x3_stream.hip: presplit_f16x3, the layout [(k/8) * 2 + part][feature] of 16-byte units), x split on the fly.
// Organisation: everything here is latency and co-residency: a workgroup = 8 waves = 16 rows x 256 features (32 per wave: two MFMA
// feature blocks).  The 16 x K tile of the operand is staged ONCE per workgroup: every thread loads one 8-value slot (+ x_add),
// the row maxima meet in LDS (atomic max on the magnitude bits), the slot is scaled, split into its two fp16 parts and written to
// LDS in B-fragment order -- a wave then reads its operand with two ds_read_b128 per k-step instead of holding a private copy of
// the tile (64 registers: with 256 registers per lane a workgroup filled a CU's register file, so every launch had to wait for the
// CU to drain: 12.3 us per call in the trace against 8.5 for the library GEMM).  The A fragments come straight from the pre-split
// image in global memory (L2-resident: the same 256 KB for every workgroup), four 16-byte loads per k-step and lane, SL_AHEAD
// k-steps ahead; the epilogue's operands are requested up front.  Epilogue: bias, ReLU, residual, and -- when the workgroup holds
// whole rows (N == 256) -- nn.LayerNorm with exact two-pass statistics across the eight waves (two LDS exchanges).
#include "common.h"
#include "f16x3.h"

#include <algorithm>
#include <cstdlib>

namespace univs {

constexpr int SL_WAVES = 8;
constexpr int SL_AHEAD = 4;     // k-steps of A fragments in flight
constexpr int SL_KMAX = 256;     // one 8-value slot of the operand tile per thread

struct SlArgs {
  const float* X;       // [M, K]
  const float* Xadd;    // [M, K] or null: the operand is X + Xadd (`with_pos_embed`)
  const u32x4* Wp;      // pre-split W [Nw, K]: unit ((k / 8) * 2 + part) * Nw + feature
  const float* winv;    // [Nw]
  const float* bias;    // [Nw] or null
  const float* Res;     // [M, N] or null
  const float* ln_g;    // [N] or null (then N == 256)
  const float* ln_b;    // [N] or null
  float ln_eps;
  float* Y;             // [M, N]
  int M, N, K, Nw, f_off, relu;   // features [f_off, f_off + N) of the pre-split matrix
  int out_T;                      // > 0: rows are (q, t) pairs, q-major with T = out_T; row (q, t) of the result is stored at (t, q) -- the
                                  // mask embeddings [Q', T, C] handed on as [T, Q', C] (...decoder_univs.py:527) without a copy of its own
  int add_features;               // Xadd applies to output features < add_features only (a multiple of 32; 0: to all) -- q, k = in_proj(tgt +
                                  // pos) and v = in_proj(tgt) of a self-attention in one launch
};

typedef unsigned sl_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sl_row_sum(float v) {          // over the four lanes (lane >> 4) that hold one row
  const sl_u2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float r1 = __uint_as_float(s1.x) + __uint_as_float(s1.y);
  const sl_u2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
  return __uint_as_float(s2.x) + __uint_as_float(s2.y);
}

__global__ __launch_bounds__(64 * SL_WAVES, 3) void small_linear_kernel(const SlArgs a) {
  // LDS: operand tiles [tile: with x_add / plain][part h, m][k-step][lane] 16-byte units | row maxima [2][16] | inverse scales [2][16]
  extern __shared__ __attribute__((aligned(16))) u32x4 xs[];
  __shared__ float red[2][SL_WAVES][16];
  __shared__ unsigned rmax[2][16];
  __shared__ float rinv[2][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int M = a.M, N = a.N, K = a.K, KS = K >> 5;
  const int row0 = blockIdx.x * 16;
  const int fbase = blockIdx.y * (32 * SL_WAVES);
  const int f0 = fbase + wave * 32;                             // this wave's 32 features (within [0, N))
  const bool with_ln = a.ln_g != nullptr;                      // uniform; then N == 256 and every wave has features
  const bool f_ok1 = f0 + 16 < N;                               // second feature block inside N (N % 16 == 0)
  // which operand tiles this workgroup's waves read (uniform): tile 0 = x + x_add, tile 1 = x
  const int fend = min(fbase + 32 * SL_WAVES, N);
  const bool need_add = a.Xadd != nullptr && (a.add_features == 0 || fbase < a.add_features);
  const bool need_plain = a.Xadd == nullptr || (a.add_features != 0 && fend > a.add_features);
  const int my_tile = (a.Xadd != nullptr && (a.add_features == 0 || f0 < a.add_features)) ? 0 : 1;   // wave-uniform
  const int tile_units = 2 * KS * 64;

  // ---- A fragments: units ((4 ks + g) * 2 + part) * Nw + f_off + f0 + 16 q + j; the first SL_AHEAD k-steps are requested now
  const u32x4* wl = a.Wp + (size_t)(g * 2) * a.Nw + a.f_off + min(f0, N - 16) + j;
  const size_t kstep_units = (size_t)8 * a.Nw;
  u32x4 afr[SL_AHEAD][2][2];                                    // [stage][feature block][part]
  auto load_a = [&](int ks, u32x4 (&d)[2][2]) __attribute__((always_inline)) {
    const u32x4* p = wl + (size_t)ks * kstep_units;
    d[0][0] = p[0];
    d[0][1] = p[a.Nw];
    d[1][0] = f_ok1 ? p[16] : p[0];
    d[1][1] = f_ok1 ? p[a.Nw + 16] : p[a.Nw];
  };
#pragma unroll
  for (int u = 0; u < SL_AHEAD; ++u)
    if (u < KS) load_a(u, afr[u]);

  // ---- the operand tile(s): slot s = ks * 64 + (g_ * 16 + j_) holds the 8 values of row j_ at columns 32 ks + 8 g_
  if (tid < 32) (&rmax[0][0])[tid] = 0u;
  __syncthreads();
  constexpr int SLOTS = (SL_KMAX / 32) * 64 / (64 * SL_WAVES);  // per thread (1)
  f32x4 xv[SLOTS][2], xw[SLOTS][2];
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int s_ = tid + 64 * SL_WAVES * i;
    if (s_ < KS * 64) {
      const int ks = s_ >> 6, l_ = s_ & 63, j_ = l_ & 15, g_ = l_ >> 4;
      const long long off = (long long)min(row0 + j_, M - 1) * K + 32 * ks + 8 * g_;
      xv[i][0] = *reinterpret_cast<const f32x4*>(a.X + off);
      xv[i][1] = *reinterpret_cast<const f32x4*>(a.X + off + 4);
      if (need_add) {
        xw[i][0] = xv[i][0] + *reinterpret_cast<const f32x4*>(a.Xadd + off);
        xw[i][1] = xv[i][1] + *reinterpret_cast<const f32x4*>(a.Xadd + off + 4);
        atomicMax(&rmax[0][j_], l3_absmax8(xw[i][0], xw[i][1]));
      }
      if (need_plain) atomicMax(&rmax[1][j_], l3_absmax8(xv[i][0], xv[i][1]));
    }
  }
  __syncthreads();                                              // the row maxima are complete
#pragma unroll
  for (int i = 0; i < SLOTS; ++i) {
    const int s_ = tid + 64 * SL_WAVES * i;
    if (s_ < KS * 64) {
      const int j_ = s_ & 15;
      float sc, inv;
      f16x8 h8, m8;
      if (need_add) {
        l3_scale(rmax[0][j_], 14, sc, inv);
        l3_split8(xw[i][0], xw[i][1], sc, h8, m8);
        xs[s_] = __builtin_bit_cast(u32x4, h8);
        xs[KS * 64 + s_] = __builtin_bit_cast(u32x4, m8);
        if (s_ < 16) rinv[0][s_] = inv;                          // (slot s_ < 16: k-step 0, g_ 0, row s_)
      }
      if (need_plain) {
        l3_scale(rmax[1][j_], 14, sc, inv);
        l3_split8(xv[i][0], xv[i][1], sc, h8, m8);
        xs[tile_units + s_] = __builtin_bit_cast(u32x4, h8);
        xs[tile_units + KS * 64 + s_] = __builtin_bit_cast(u32x4, m8);
        if (s_ < 16) rinv[1][s_] = inv;
      }
    }
  }
  __syncthreads();                                              // the tiles are in place
  if (f0 >= N && !with_ln) return;                              // (no barrier below without the LayerNorm)
  const float sx_inv = rinv[my_tile][j];
  // the epilogue's operands, requested ahead of the product
  const int m = min(row0 + j, M - 1);
  const int row = row0 + j;
  const bool row_ok = row < M;
  f32x4 e_wi[2], e_bi[2], e_res[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int f = f0 + 16 * q + 4 * g;
    const bool ok = f < N;                                      // (N % 4 == 0)
    const int fw = a.f_off + (ok ? f : 0);
    e_wi[q] = *reinterpret_cast<const f32x4*>(a.winv + fw);
    e_bi[q] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + fw) : (f32x4){0.f, 0.f, 0.f, 0.f};
    e_res[q] = (a.Res && ok) ? *reinterpret_cast<const f32x4*>(a.Res + (long long)m * N + f) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const u32x4* xt = xs + my_tile * tile_units + lane;

  // ---- the product
  f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
  for (int kb = 0; kb < KS; kb += SL_AHEAD) {
#pragma unroll
    for (int u = 0; u < SL_AHEAD; ++u) {
      const int ks = kb + u;
      if (ks < KS) {                                            // uniform
        const f16x8 bh = __builtin_bit_cast(f16x8, xt[ks * 64]), bm = __builtin_bit_cast(f16x8, xt[KS * 64 + ks * 64]);
        const f16x8 ah0 = __builtin_bit_cast(f16x8, afr[u][0][0]), am0 = __builtin_bit_cast(f16x8, afr[u][0][1]);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, afr[u][1][0]), am1 = __builtin_bit_cast(f16x8, afr[u][1][1]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am0, bh, acc[0], 0, 0, 0);   // smallest terms first
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am1, bh, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bm, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bm, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);                       // (the refill below overwrites the fragments just consumed: no copies)
        if (ks + SL_AHEAD < KS) load_a(ks + SL_AHEAD, afr[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: D[i = feature][j = row]: a lane holds features f0 + 16 q + 4 g ... + 3 of row j
  f32x4 v[2];
  float sm = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int f = f0 + 16 * q + 4 * g;
    const bool ok = f < N;
    f32x4 t = (acc[q] * sx_inv) * e_wi[q] + e_bi[q];
    if (a.relu) t = __builtin_elementwise_maximum(t, (f32x4){0.f, 0.f, 0.f, 0.f});   // NaN-propagating, as torch.relu
    t += e_res[q];
    v[q] = ok ? t : (f32x4){0.f, 0.f, 0.f, 0.f};
    sm += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
  }
  if (with_ln) {
    // nn.LayerNorm over the row's 256 values: 8 in this lane, 32 in this wave (its four lanes of the row), the rest in the other waves
    const float inv_n = 1.0f / (float)N;
    const float ws = sl_row_sum(sm);
    if (g == 0) red[0][wave][j] = ws;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < SL_WAVES; ++w) tot += red[0][w][j];
    const float mean = tot * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      v[q] -= mean;
#pragma unroll
      for (int e = 0; e < 4; ++e) sq = fmaf(v[q][e], v[q][e], sq);
    }
    const float wq = sl_row_sum(sq);
    if (g == 0) red[1][wave][j] = wq;
    __syncthreads();
    float tq = 0.f;
#pragma unroll
    for (int w = 0; w < SL_WAVES; ++w) tq += red[1][w][j];
    const float rstd = 1.0f / sqrtf(tq * inv_n + a.ln_eps);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = f0 + 16 * q + 4 * g;
      const f32x4 gm = *reinterpret_cast<const f32x4*>(a.ln_g + f);
      f32x4 y = (v[q] * rstd) * gm;
      if (a.ln_b) y += *reinterpret_cast<const f32x4*>(a.ln_b + f);
      v[q] = y;
    }
  }
  const int orow = a.out_T > 0 ? (row % a.out_T) * (M / a.out_T) + row / a.out_T : row;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int f = f0 + 16 * q + 4 * g;
    if (row_ok && f < N) *reinterpret_cast<f32x4*>(a.Y + (long long)orow * N + f) = v[q];
  }
}

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED when the shape is not covered
int small_linear_f32(const float* x, const float* xadd, const void* wp, const float* winv, const float* bias, int n_w, int f_off,
                     const float* residual, const float* ln_g, const float* ln_b, float ln_eps, float* y, long long M, int N, int K,
                     int relu, int add_features, int out_T, hipStream_t st) {
  if (M <= 0 || N <= 0) return UNIVS_OK;
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (K < 32 || K % 32 != 0 || K > SL_KMAX || N % 16 != 0 || n_w % 4 != 0 || f_off % 4 != 0 || f_off < 0 || f_off + N > n_w || M > 16LL * 65535 ||
      (ln_g && N != 32 * SL_WAVES) || (ln_b && !ln_g) || add_features < 0 || add_features % 32 != 0 || out_T < 0 || (out_T > 0 && (M % out_T != 0 || residual || ln_g)) || mis(x) || mis(xadd) || mis(wp) || mis(winv) || mis(bias) || mis(residual) ||
      mis(ln_g) || mis(ln_b) || mis(y) || M * (long long)std::max(N, K) * 4 >= 0x7FFFFFFFLL)
    return UNIVS_ERR_NOT_IMPLEMENTED;
  SlArgs a{};
  a.X = x; a.Xadd = xadd; a.Wp = reinterpret_cast<const u32x4*>(wp); a.winv = winv; a.bias = bias; a.Res = residual;
  a.ln_g = ln_g; a.ln_b = ln_b; a.ln_eps = ln_eps; a.Y = y;
  a.M = (int)M; a.N = N; a.K = K; a.Nw = n_w; a.f_off = f_off; a.relu = relu ? 1 : 0; a.add_features = add_features; a.out_T = out_T;
  dim3 grid((unsigned)((M + 15) / 16), (unsigned)((N + 32 * SL_WAVES - 1) / (32 * SL_WAVES)));
  const size_t lds = (size_t)2 * 2 * (K / 32) * 64 * 16;        // two operand tiles (x + x_add, x), two parts each
  hipLaunchKernelGGL(small_linear_kernel, grid, dim3(64 * SL_WAVES), lds, st, a);
  return check_launch("small_linear_f32");
}

// ---------------------------------------------------------------------------------------------------------------------------------
// A CHAIN of up to three 256 -> 256 Linears on few rows in ONE launch: y = L3(act(L2(act(L1(LN?(x)))))) -- the mask-embedding MLP of every
// prediction head (univs/modeling/transformer_decoder/transformer_layers.py:205-217 `MLP`, called at ...decoder_univs.py:520 on the
// `decoder_norm`-ed queries): three launches of small_linear_kernel (8 us each, all of it launch ramp and latency) and a LayerNorm launch
// per head, ten heads per clip.  Same workgroup shape (8 waves = 16 rows x 256 features) and the same arithmetic; between two stages the
// 16 x 256 result goes back to LDS AS THE NEXT OPERAND: every lane holds 2 x 4 consecutive features of one row, the row maxima meet in
// LDS (atomic max), the lane scales, splits and writes its two 8-byte halves of the B-fragment units -- the values, the row maximum, the
// scale and the parts are what the next launch would have read back from memory and computed, so the chain is bit-identical to the
// sequence of launches.  The next stage's first weight fragments are requested before the two barriers of the hand-over.
// Optional nn.LayerNorm on the INPUT rows (`decoder_norm`): exact two-pass statistics over the staged tile, summed in a fixed order.
struct SlChainArgs {
  const float* X;       // [M, 256]
  const u32x4* Wp[3];   // pre-split W_s [256, 256]
  const float* winv[3];
  const float* bias[3];
  int relu[3];
  int stages;
  const float* in_g;    // LayerNorm on x (weight / bias / eps) or null
  const float* in_b;
  float in_eps;
  float* Xn;            // [M, 256] or null: the normalised input rows (for the callers that need `decoder_norm(x)` itself)
  float* Y;             // [M, 256] (out_T > 0: row (q, t) stored at (t, q))
  int M, out_T;
};

__global__ __launch_bounds__(64 * SL_WAVES, 3) void small_chain_kernel(const SlChainArgs a) {
  constexpr int K = 256, N = 256, KS = K >> 5;
  __shared__ __attribute__((aligned(16))) u32x4 xs[2][2 * KS * 64];      // two operand tiles (ping-pong), [part][k-step][lane]
  __shared__ unsigned rmax[3][16];
  __shared__ float rstat[2][SL_WAVES][16];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int M = a.M;
  const int row0 = blockIdx.x * 16;
  const int f0 = wave * 32;

  u32x4 afr[SL_AHEAD][2][2];
  auto load_a = [&](const u32x4* wl, int ks, u32x4 (&d)[2][2]) __attribute__((always_inline)) {
    const u32x4* p = wl + (size_t)ks * (8 * N);
    d[0][0] = p[0];
    d[0][1] = p[N];
    d[1][0] = p[16];
    d[1][1] = p[N + 16];
  };
  const size_t wl_off = (size_t)(g * 2) * N + f0 + j;
#pragma unroll
  for (int u = 0; u < SL_AHEAD; ++u) load_a(a.Wp[0] + wl_off, u, afr[u]);

  if (tid < 48) (&rmax[0][0])[tid] = 0u;
  __syncthreads();
  // ---- stage-0 operand: slot s_ = tid (KS * 64 = 512 slots): row j_ = s_ & 15, columns 32 ks + 8 g_
  {
    const int s_ = tid;
    const int ks = s_ >> 6, l_ = s_ & 63, j_ = l_ & 15, g_ = l_ >> 4;
    const int mrow = min(row0 + j_, M - 1);
    const long long off = (long long)mrow * K + 32 * ks + 8 * g_;
    f32x4 x0 = *reinterpret_cast<const f32x4*>(a.X + off), x1 = *reinterpret_cast<const f32x4*>(a.X + off + 4);
    if (a.in_g) {
      // nn.LayerNorm over the row: mean, then the centred second moment (two passes).  A row's 32 slots sit in the four lanes j_ + 16 g_
      // of each of the 8 waves: lane sums by permlane swaps, wave sums through LDS, added in a fixed order (deterministic)
      const float s4 = sl_row_sum(((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3])));
      if (g_ == 0) rstat[0][ks][j_] = s4;
      __syncthreads();
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < SL_WAVES; ++w) tot += rstat[0][w][j_];
      const float mean = tot * (1.0f / K);
      x0 -= mean;
      x1 -= mean;
      float sq = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) sq = fmaf(x0[e], x0[e], fmaf(x1[e], x1[e], sq));
      const float q4 = sl_row_sum(sq);
      if (g_ == 0) rstat[1][ks][j_] = q4;
      __syncthreads();
      float tq = 0.f;
#pragma unroll
      for (int w = 0; w < SL_WAVES; ++w) tq += rstat[1][w][j_];
      const float rstd = 1.0f / sqrtf(tq * (1.0f / K) + a.in_eps);
      const int c = 32 * ks + 8 * g_;
      x0 = (x0 * rstd) * *reinterpret_cast<const f32x4*>(a.in_g + c);
      x1 = (x1 * rstd) * *reinterpret_cast<const f32x4*>(a.in_g + c + 4);
      if (a.in_b) {
        x0 += *reinterpret_cast<const f32x4*>(a.in_b + c);
        x1 += *reinterpret_cast<const f32x4*>(a.in_b + c + 4);
      }
      if (a.Xn && row0 + j_ < M) {
        *reinterpret_cast<f32x4*>(a.Xn + off) = x0;
        *reinterpret_cast<f32x4*>(a.Xn + off + 4) = x1;
      }
    }
    atomicMax(&rmax[0][j_], l3_absmax8(x0, x1));
    __syncthreads();
    float sc, inv;
    l3_scale(rmax[0][j_], 14, sc, inv);
    f16x8 h8, m8;
    l3_split8(x0, x1, sc, h8, m8);
    xs[0][s_] = __builtin_bit_cast(u32x4, h8);
    xs[0][KS * 64 + s_] = __builtin_bit_cast(u32x4, m8);
  }
  __syncthreads();

  const int row = row0 + j;
  const bool row_ok = row < M;
  f32x4 v[2];
#pragma unroll 1
  for (int st = 0; st < a.stages; ++st) {
    float sc_, sx_inv;
    l3_scale(rmax[st][j], 14, sc_, sx_inv);
    const u32x4* xt = xs[st & 1] + lane;
    const u32x4* wl = a.Wp[st] + wl_off;
    f32x4 e_wi[2], e_bi[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int f = f0 + 16 * q + 4 * g;
      e_wi[q] = *reinterpret_cast<const f32x4*>(a.winv[st] + f);
      e_bi[q] = a.bias[st] ? *reinterpret_cast<const f32x4*>(a.bias[st] + f) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int kb = 0; kb < KS; kb += SL_AHEAD) {
#pragma unroll
      for (int u = 0; u < SL_AHEAD; ++u) {
        const int ks = kb + u;
        const f16x8 bh = __builtin_bit_cast(f16x8, xt[ks * 64]), bm = __builtin_bit_cast(f16x8, xt[KS * 64 + ks * 64]);
        const f16x8 ah0 = __builtin_bit_cast(f16x8, afr[u][0][0]), am0 = __builtin_bit_cast(f16x8, afr[u][0][1]);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, afr[u][1][0]), am1 = __builtin_bit_cast(f16x8, afr[u][1][1]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am0, bh, acc[0], 0, 0, 0);   // smallest terms first
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am1, bh, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bm, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bm, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + SL_AHEAD < KS) load_a(wl, ks + SL_AHEAD, afr[u]);
        else if (st + 1 < a.stages) load_a(a.Wp[st + 1] + wl_off, ks + SL_AHEAD - KS, afr[u]);   // the next stage's first fragments
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 t = (acc[q] * sx_inv) * e_wi[q] + e_bi[q];
      if (a.relu[st]) t = __builtin_elementwise_maximum(t, (f32x4){0.f, 0.f, 0.f, 0.f});
      v[q] = t;
    }
    if (st + 1 < a.stages) {
      // hand-over: the result rows become the next operand tile (see the header)
      atomicMax(&rmax[st + 1][j], l3_absmax8(v[0], v[1]));
      __syncthreads();
      float sc, inv;
      l3_scale(rmax[st + 1][j], 14, sc, inv);
      _Float16* dst = reinterpret_cast<_Float16*>(xs[(st + 1) & 1]);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 hh, mm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xsv = v[q][e] * sc;
          hh[e] = (_Float16)xsv;
          mm[e] = (_Float16)(xsv - (float)hh[e]);
        }
        const int unit = wave * 64 + (2 * q + (g >> 1)) * 16 + j;           // k-step = this wave's 32 features
        *reinterpret_cast<h4*>(dst + (size_t)unit * 8 + 4 * (g & 1)) = hh;
        *reinterpret_cast<h4*>(dst + (size_t)(KS * 64 + unit) * 8 + 4 * (g & 1)) = mm;
      }
      __syncthreads();
    }
  }
  const int orow = a.out_T > 0 ? (row % a.out_T) * (M / a.out_T) + row / a.out_T : row;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int f = f0 + 16 * q + 4 * g;
    if (row_ok) *reinterpret_cast<f32x4*>(a.Y + (long long)orow * N + f) = v[q];
  }
}

int small_chain_f32(const float* x, int stages, const void* const* wp, const float* const* winv, const float* const* bias, const int* relu,
                    const float* in_g, const float* in_b, float in_eps, float* xn, float* y, long long M, int out_T, hipStream_t st) {
  if (M <= 0) return UNIVS_OK;
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (stages < 1 || stages > 3 || M > 16LL * 65535 || out_T < 0 || (out_T > 0 && M % out_T != 0) || mis(x) || mis(y) || mis(xn) || mis(in_g) ||
      mis(in_b) || (in_b && !in_g) || (xn && !in_g) || M * 256LL * 4 >= 0x7FFFFFFFLL)
    return UNIVS_ERR_NOT_IMPLEMENTED;
  SlChainArgs a{};
  a.X = x; a.stages = stages; a.in_g = in_g; a.in_b = in_b; a.in_eps = in_eps; a.Xn = xn; a.Y = y; a.M = (int)M; a.out_T = out_T;
  for (int s_ = 0; s_ < stages; ++s_) {
    if (!wp[s_] || !winv[s_] || mis(wp[s_]) || mis(winv[s_]) || mis(bias[s_])) return UNIVS_ERR_NOT_IMPLEMENTED;
    a.Wp[s_] = reinterpret_cast<const u32x4*>(wp[s_]); a.winv[s_] = winv[s_]; a.bias[s_] = bias[s_]; a.relu[s_] = relu[s_] ? 1 : 0;
  }
  hipLaunchKernelGGL(small_chain_kernel, dim3((unsigned)((M + 15) / 16)), dim3(64 * SL_WAVES), 0, st, a);
  return check_launch("small_chain_f32");
}

}  // namespace univs
