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

}  // namespace univs
