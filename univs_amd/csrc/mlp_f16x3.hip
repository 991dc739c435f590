// y[M, C] = act(x[M, C] W1^T + b1) W2^T + b2 (+ residual): a two-Linear MLP in ONE kernel, three-product fp16 arithmetic (see
// linear_f16x3.hip), the hidden activations never leave the CU.
//   * the encoder FFN 256 -> 1024 (ReLU) -> 256 (msdeformattn.py:87-91: linear2(dropout(activation(linear1(src)))));
//   * the Swin Mlp C -> 4C (GELU) -> C + shortcut (swin.py:35-58, :291-293) at C = 96 / 128 / 192 / 256.
// With two kernels the [M, Hd] activations are written and read back (396 MB each way per encoder layer at 720p, 452 MB at Swin
// stage 1) and split into fp16 parts a second time.  Here a wave owns 16 CT token rows for the whole MLP:
//   * its x tile is read once, scaled by the exact row maximum and kept as two fp16 parts in registers (the B operand of GEMM 1);
//   * the hidden dimension is walked in chunks of 32 features: h = act(x W1[chunk]^T + b1) comes out of the matrix cores as
//     D[i = hidden feature][j = token] -- a lane holds 4 + 4 consecutive features of one token -- which IS a B operand of the
//     second product if the k-order inside a 32-wide k-step is taken as (4 g + e, 16 + 4 g + e): W2 is pre-split ONCE with that
//     permutation (presplit mode 2), so h goes from accumulator to operand with no data movement: bias, activation, running
//     power-of-two row scale (as x in linear_f16x3: set by the first chunk, lowered with an exact rescaling of the output
//     accumulators), split into two fp16 parts;
//   * y accumulates over the chunks in registers (C / 16 blocks of 16 features) and is stored once.
// Both weight matrices stream through LDS in the chunk order, double-buffered: chunk = W1 rows [32 c, 32 c + 32) (all k) and
// the k-step c of W2 (all rows), 256 C bytes; a thread moves its 16-byte units of the next chunk L2 -> registers -> LDS on a
// static schedule spread over the chunk's batches (a unit is committed a quarter of a chunk after its fetch), one barrier
// per chunk.  A workgroup = NW waves (8; 4 for the narrow widths, so that two or three workgroups share a CU and one's
// load / store / activation phases overlap another's matrix phases), persistent over row groups.
// A fragments (two 16-row blocks x two parts = four ds_read_b128 per batch of six or twelve MFMAs) are requested TWO batches
// ahead into a ring of three; the wait before a batch is `s_waitcnt lgkmcnt(n)` with n = the LDS operations issued after
// that batch's reads (the next batch's four reads + this slot's commits): LDS operations complete in order and the loop
// holds no scalar memory operation (checked in the ISA), so the count is exact.
#include "common.h"
#include "config.h"
#include "f16x3.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace univs {

enum { ML_ACT_RELU = 1, ML_ACT_GELU = 2 };
// ABL (timing experiments, UnivsConfig.linear_ablate = 2 / 3 / 4; results are then WRONG): 1 no MFMAs, 2 no LDS reads of the A
// fragments, 3 no activation / scale / split of the hidden activations

struct MlpArgs {
  const float* X;       // [M, C]
  const u32x4* W1p;     // W1 [Hd, C] pre-split, standard k order: unit (kc * 2 + part) * Hd + r
  const float* w1inv;   // [Hd]
  const float* b1;      // [Hd] or null
  const u32x4* W2p;     // W2 [C, Hd] pre-split with the MLP k-permutation: unit (kc * 2 + part) * C + r
  const float* w2inv;   // [C]
  const float* b2;      // [C] or null
  const float* Res;     // [M, C] or null
  float* Y;             // [M, C]
  const float* ln_g;    // [C] or null: x is first normalised over its C channels (nn.LayerNorm: weight, bias, eps)
  const float* ln_b;    // [C]
  float ln_eps;
  const float* pln_g;   // [C] or null: nn.LayerNorm applied to the RESULT rows (after bias and residual): the post-norm layer
  const float* pln_b;   // [C]       `norm(x + mlp(x))` (msdeformattn.py:91-95)
  const float* padd;    // [padd_rows, C] or null: second output Y2 = Y + padd[row % padd_rows] (the next layer's `src + pos`)
  float* Y2;            // [M, C]
  float pln_eps;
  int padd_rows;
  int M, Hd, nwg;
  int dual;             // with pln_g and Y2: Y receives the finished rows UN-normalised and Y2 their post-LN (the Swin block's output and the
                        // next block's norm1 / the stage's output norm of it: swin.py:286-293, :236, :664-672); no padd
  int res_normed;       // with ln_g, Res == null: the residual is LN(x) -- the post-norm chain x1 = norm1(.), y = norm2(x1 + mlp(x1)) of the
                        // encoder layer (msdeformattn.py:124-133).  The normalised rows are parked in Y when the x tile is made and
                        // read back for the epilogue (same wave, through L2) instead of living in 16 CT C / 64 more registers.
};

// sum over the four lanes (k-groups, lane >> 4) that hold one row.  (The two results are taken through a typed vector and
// __uint_as_float: hipcc 7.2 evaluates `__builtin_bit_cast(float, s[1])` on the builtin's result as element 0.)
typedef unsigned ml_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float ml_row_sum(float v) {
  const ml_u2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float r1 = __uint_as_float(s1.x) + __uint_as_float(s1.y);
  const ml_u2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
  return __uint_as_float(s2.x) + __uint_as_float(s2.y);
}

template <int N>
__device__ __forceinline__ void ml_wait_lgkm(u32x4 (&d)[2][2]) {   // the operands tie the MFMAs behind this wait
  static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit counter");
  if constexpr (N == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
  else asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void ml_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ml_static_for<I + 1, N>(f);
  }
}

// static schedule of the weight stream: unit v of a thread is fetched in batch slot ml_fs(v) and committed ML_D slots later
constexpr int ml_dist(int nbat) { return nbat / 4 > 2 ? nbat / 4 : 2; }
constexpr int ml_fs(int v, int upt, int nbat) { return v * (nbat - ml_dist(nbat)) / upt; }
constexpr int ml_commits_in(int t, int upt, int nbat) {
  int n = 0;
  for (int v = 0; v < upt; ++v) n += (ml_fs(v, upt, nbat) + ml_dist(nbat) == t) ? 1 : 0;
  return n;
}
// LDS-DMA form of the stream (DB kernels): unit v goes global -> LDS directly (global_load_lds_dwordx4: no staging registers, no
// ds_write), requested in batch slot ml_dma_slot(v) of the chunk BEFORE the one that reads it -- the first half of the chunk's slots,
// so every request has at least half a chunk to land -- and awaited (vmcnt(0)) in front of the barrier that opens its chunk.
// Register staging held three 16-byte units per thread in flight: 24 KB per CU against an L2 latency of ~1.5k - 3k clocks
// = 8 - 16 bytes per clock and CU, i.e. 4k - 8k clocks for a chunk's 64 KB -- the stream's latency, not the matrix pipe, set the
// chunk time (7.5k clocks measured; 3k of MFMA per SIMD).
constexpr int ml_dma_slot(int v, int upt, int nbat) { return v * (nbat / 2) / upt; }
#ifdef UNIVS_MLP_REGSTAGE       // (A / B: the register-staged stream everywhere)
constexpr bool ML_GLDS = false;
#else
constexpr bool ML_GLDS = true;
#endif
// one wave's 1-KB piece: lane l's 16 bytes at `gsrc` -> LDS byte address `lds_dst` + 16 l (`lds_dst` wave-uniform).  M0 carries the
// destination and is written in the statement that uses it (the compiler does not preserve it around asm statements).
__device__ __forceinline__ void ml_glds16(const u32x4* gsrc, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);  // (an "s" operand the compiler holds in a vector register is passed as one)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

constexpr int ml_ring(int upt, int nbat) {          // most units in flight at once (after a slot's commits and fetches)
  int worst = 1;
  for (int t = 0; t < nbat; ++t) {
    int n = 0;
    for (int v = 0; v < upt; ++v) n += (ml_fs(v, upt, nbat) <= t && t < ml_fs(v, upt, nbat) + ml_dist(nbat)) ? 1 : 0;
    worst = n > worst ? n : worst;
  }
  return worst;
}

// LDS (16-byte units): 2 x { W1 part [C/8 k-chunks][2 parts][32 hidden rows] | W2 part [4 k-groups][2 parts][C rows] } |
//                      b1[Hd] | w1inv[Hd] | b2[C] | w2inv[C] | ln weight[C] | ln bias[C] | post-ln weight[C] | post-ln bias[C]
// DB: the chunk images double-buffered (one barrier per chunk).  !DB (C = 384, where two 96-KB images do not fit): ONE image, its
// W1 part refilled with the next chunk's rows during the second product and its W2 part with this chunk's k-step during the
// first, a barrier between the two products as well; a workgroup is then 4 waves, one per SIMD (512 registers per lane).
template <int KS1, int CT, int ACT, int NW, int ABL, bool DB>
__global__ __launch_bounds__(64 * NW, DB ? 2 : 1) void mlp_f16x3(const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 Lds[];
  constexpr int THREADS = 64 * NW;
  constexpr int C = 32 * KS1;
  constexpr int NOB = C / 16;                                    // output feature blocks
  constexpr int NBAT = 2 * KS1;                                  // A-fragment batches per chunk: KS1 k-steps of GEMM 1, KS1 block pairs of GEMM 2
  constexpr int W1U = 8 * C, W2U = 8 * C, CHU = W1U + W2U;       // units per chunk
  constexpr int UPT = CHU / THREADS;                             // units per thread and chunk
  constexpr int UPH = W1U / THREADS;                             // !DB: units per thread and phase (W1U == W2U)
  constexpr int WR = DB ? ml_ring(UPT, NBAT) : ml_ring(UPH, KS1), WD = DB ? ml_dist(NBAT) : ml_dist(KS1);
  constexpr int NBUF = DB ? 2 : 1;
  constexpr bool GLDS = DB && ML_GLDS;                           // the weight stream by LDS-DMA (see ml_dma_slot)
  static_assert(W1U % THREADS == 0, "phase geometry");
  static_assert(CHU % THREADS == 0 && NOB == 2 * KS1 && NBAT >= 4, "chunk geometry");
  constexpr int RG = NW * 16 * CT;                               // rows per workgroup round
  const int M = a.M, Hd = a.Hd, NCH = Hd >> 5, nwg = a.nwg;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  float* b1_lds = reinterpret_cast<float*>(Lds + NBUF * CHU);
  float* w1inv_lds = b1_lds + Hd;
  float* b2_lds = w1inv_lds + Hd;
  float* w2inv_lds = b2_lds + C;
  float* lng_lds = w2inv_lds + C;
  float* lnb_lds = lng_lds + C;
  float* plng_lds = lnb_lds + C;
  float* plnb_lds = plng_lds + C;
  const bool with_ln = a.ln_g != nullptr;                        // uniform
  const bool with_pln = a.pln_g != nullptr;                      // uniform
  const int ngroups = (M + RG - 1) / RG;
  if ((int)blockIdx.x >= ngroups) return;

  for (int r = tid; r < Hd; r += THREADS) {
    b1_lds[r] = a.b1 ? a.b1[r] : 0.f;
    w1inv_lds[r] = a.w1inv[r];
  }
  for (int r = tid; r < C; r += THREADS) {
    b2_lds[r] = a.b2 ? a.b2[r] : 0.f;
    w2inv_lds[r] = a.w2inv[r];
    lng_lds[r] = with_ln ? a.ln_g[r] : 1.f;
    lnb_lds[r] = (with_ln && a.ln_b) ? a.ln_b[r] : 0.f;
    plng_lds[r] = with_pln ? a.pln_g[r] : 1.f;
    plnb_lds[r] = (with_pln && a.pln_b) ? a.pln_b[r] : 0.f;
  }
  if (with_ln) __syncthreads();                                  // (the first x tile is normalised before the chunk loop's barrier)

  // ---- weight stream: unit i = tid + THREADS v of the chunk image
  auto w_src = [&](int c, int v) __attribute__((always_inline)) -> const u32x4* {
    const int i = tid + THREADS * v;
    return i < W1U ? a.W1p + ((size_t)(i >> 5) * Hd + 32 * c + (i & 31)) : a.W2p + ((size_t)c * W2U + (i - W1U));
  };
  {
    constexpr int U0 = DB ? UPT : UPH;                           // chunk 0 -> buffer 0 (!DB: its W1 part; W2 follows in the loop)
    u32x4 w0[U0];
#pragma unroll
    for (int v = 0; v < U0; ++v) w0[v] = *w_src(0, v);
#pragma unroll
    for (int v = 0; v < U0; ++v) Lds[tid + THREADS * v] = w0[v];
  }
  [[maybe_unused]] u32x4 wreg[WR];
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Lds);   // LDS byte address of the chunk images

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((long long)M * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, (int)((long long)M * C * 4), 0x00020000);
  const bool res_normed = a.res_normed != 0;                      // uniform
  const bool with_res = a.Res != nullptr || res_normed;
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(res_normed ? a.Y : a.Res ? a.Res : a.X), 0, (int)((long long)M * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t y2rs = __builtin_amdgcn_make_buffer_rsrc(a.Y2 ? a.Y2 : a.Y, 0, (int)((long long)M * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t pars = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.padd ? a.padd : a.X), 0, (int)((long long)(a.padd ? a.padd_rows : M) * C * 4), 0x00020000);

  u32x4 afr[3][2][2];                                            // [ring][block][part]
  if (ABL == 2) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int q = 0; q < 2; ++q) afr[r][q][0] = afr[r][q][1] = (u32x4){0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  }
  auto read_batch = [&](u32x4 (&d)[2][2], unsigned addr, unsigned pstride) __attribute__((always_inline)) {
    if (ABL == 2) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned ah = addr + (unsigned)(q * 256);
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(d[q][0]), "=&v"(d[q][1]) : "v"(ah), "v"(ah + pstride) : "memory");
    }
  };
  // byte addresses inside a chunk image: GEMM 1 (k-step t, block q): ((t*4 + g)*2 + part)*32 + 16 q + j units;
  // GEMM 2 (blocks 2 t + q): W1U + (g*2 + part)*C + 16 (2 t + q) + j units
  const unsigned a1_lane = (unsigned)((g * 64 + j) * 16);
  const unsigned a2_lane = (unsigned)((W1U + g * 2 * C + j) * 16);

  int gq = 0;                                                    // chunks done: chunk gq's image is in buffer gq & 1
#pragma unroll 1
  for (int grp = blockIdx.x; grp < ngroups; grp += nwg) {
    const int row0 = grp * RG + wave * 16 * CT;
    // ---- x tile: read once, exact row maximum, two fp16 parts in registers
    f16x8 xh[CT][KS1], xm[CT][KS1];
    float sx_inv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int m = min(row0 + 16 * ct + j, M - 1);              // rows past the end repeat the last row (not stored)
      const unsigned vo = ((unsigned)m * (unsigned)C + (unsigned)(8 * g)) * 4u;
      f32x4 raw[KS1][2];
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        raw[ks][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, ks * 128, 0));
        raw[ks][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo + 16u, ks * 128, 0));
      }
      if (with_ln) {
        // nn.LayerNorm over the row's C channels, exact two-pass statistics in registers (as layer_norm.hip), then weight / bias
        float sm = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
          for (int e = 0; e < 4; ++e) sm += raw[ks][0][e] + raw[ks][1][e];
        const float mean = ml_row_sum(sm) * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            raw[ks][h2] -= mean;
#pragma unroll
            for (int e = 0; e < 4; ++e) sq = fmaf(raw[ks][h2][e], raw[ks][h2][e], sq);
          }
        const float rstd = 1.0f / sqrtf(ml_row_sum(sq) * (1.0f / C) + a.ln_eps);
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2) {
            const f32x4 gm = *reinterpret_cast<const f32x4*>(lng_lds + 32 * ks + 8 * g + 4 * h2);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(lnb_lds + 32 * ks + 8 * g + 4 * h2);
            raw[ks][h2] = (raw[ks][h2] * rstd) * gm + bt;
            if (res_normed)                                      // park the residual row (rows past the end: dropped)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, raw[ks][h2]), yrs,
                                                     row0 + 16 * ct + j < M ? vo + (unsigned)(16 * h2) : 0xFFFFFFF0u, ks * 128, 0);
          }
      }
      unsigned mx = 0u;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) mx = max(mx, l3_absmax8(raw[ks][0], raw[ks][1]));
      mx = l3_row_max(mx);
      float s, inv;
      l3_scale(mx, 14, s, inv);
      sx_inv[ct] = inv;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) l3_split8(raw[ks][0], raw[ks][1], s, xh[ct][ks], xm[ct][ks]);
    }

    f32x4 acc2[NOB][CT];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc2[ob][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int eset[CT];                                                 // exponent the hidden rows' scale was set for
    float sh[CT], sh_inv[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      eset[ct] = -1000;
      sh[ct] = sh_inv[ct] = 1.0f;
    }

#pragma unroll 1
    for (int c = 0; c < NCH; ++c, ++gq) {
      if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of chunk gq's image have landed (asm loads are not in hipcc's count)
      __syncthreads();                                           // chunk gq's image is complete; the other buffer is free
      const int cn = (c + 1 == NCH) ? 0 : c + 1;
      const int bufc = DB ? (gq & 1) : 0;
      const unsigned a1 = a1_lane + (unsigned)(bufc * CHU * 16);
      const unsigned a2 = a2_lane + (unsigned)(bufc * CHU * 16);
      u32x4* const wdst = Lds + (DB ? (bufc ^ 1) * CHU : 0) + tid;
      [[maybe_unused]] const unsigned wdma = lds_base + (unsigned)(((bufc ^ 1) * CHU + 64 * wave) * 16);   // this wave's first piece of the other image
      read_batch(afr[0], a1, 512u);
      read_batch(afr[1], a1 + 4096u, 512u);

      f32x4 acc1[2][CT];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc1[q][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
      f16x8 hh[CT], hm[CT];

      // one batch slot: this slot's share of the weight stream, wait for batch T's fragments, request batch T + 2, six CT MFMAs
      auto slot = [&](auto tc) __attribute__((always_inline)) {
        constexpr int T = decltype(tc)::value;
        if constexpr (GLDS) {
          ml_static_for<0, UPT>([&](auto vc) __attribute__((always_inline)) {
            constexpr int V = decltype(vc)::value;
            if constexpr (ml_dma_slot(V, UPT, NBAT) == T) ml_glds16(w_src(cn, V), wdma + (unsigned)(THREADS * V * 16));
          });
          __builtin_amdgcn_sched_barrier(0);
          if (ABL != 2) ml_wait_lgkm<(T + 1 < NBAT ? 4 : 0)>(afr[T % 3]);     // (no LDS operation of mine behind the next batch's reads)
          if constexpr (T + 2 < NBAT) {
            if constexpr (T + 2 < KS1) read_batch(afr[(T + 2) % 3], a1 + (unsigned)((T + 2) * 4096), 512u);
            else read_batch(afr[(T + 2) % 3], a2 + (unsigned)((T + 2 - KS1) * 512), (unsigned)(C * 16));
          }
        } else if constexpr (DB) {
          ml_static_for<0, UPT>([&](auto vc) __attribute__((always_inline)) {
            constexpr int V = decltype(vc)::value;
            if constexpr (ml_fs(V, UPT, NBAT) + WD == T) wdst[THREADS * V] = wreg[V % WR];
          });
          ml_static_for<0, UPT>([&](auto vc) __attribute__((always_inline)) {
            constexpr int V = decltype(vc)::value;
            if constexpr (ml_fs(V, UPT, NBAT) == T) wreg[V % WR] = *w_src(cn, V);
          });
          __builtin_amdgcn_sched_barrier(0);
          if (ABL != 2) ml_wait_lgkm<(T + 1 < NBAT ? 4 : 0) + ml_commits_in(T, UPT, NBAT)>(afr[T % 3]);
          if constexpr (T + 2 < NBAT) {
            if constexpr (T + 2 < KS1) read_batch(afr[(T + 2) % 3], a1 + (unsigned)((T + 2) * 4096), 512u);
            else read_batch(afr[(T + 2) % 3], a2 + (unsigned)((T + 2 - KS1) * 512), (unsigned)(C * 16));
          }
        } else {
          // one image: during GEMM 1 (slots < KS1) the W2 part of THIS chunk is streamed in (units W1U + ...), during GEMM 2 the
          // W1 part of the NEXT chunk (units 0 ...); fragment requests do not cross the barrier between the two products
          constexpr int PT = T < KS1 ? T : T - KS1;                // slot within its phase
          constexpr int UOFF = T < KS1 ? UPH : 0;                  // first unit (per thread) of the part this phase refills
          ml_static_for<0, UPH>([&](auto vc) __attribute__((always_inline)) {
            constexpr int V = decltype(vc)::value;
            if constexpr (ml_fs(V, UPH, KS1) + WD == PT) wdst[THREADS * (UOFF + V)] = wreg[V % WR];
          });
          ml_static_for<0, UPH>([&](auto vc) __attribute__((always_inline)) {
            constexpr int V = decltype(vc)::value;
            if constexpr (ml_fs(V, UPH, KS1) == PT) wreg[V % WR] = *w_src(T < KS1 ? c : cn, UOFF + V);
          });
          __builtin_amdgcn_sched_barrier(0);
          constexpr bool next_requested = T < KS1 ? (T + 1 < KS1) : (T + 1 < NBAT);
          if (ABL != 2) ml_wait_lgkm<(next_requested ? 4 : 0) + ml_commits_in(PT, UPH, KS1)>(afr[T % 3]);
          if constexpr (T < KS1) {
            if constexpr (T + 2 < KS1) read_batch(afr[(T + 2) % 3], a1 + (unsigned)((T + 2) * 4096), 512u);
          } else if constexpr (T + 2 < NBAT) {
            read_batch(afr[(T + 2) % 3], a2 + (unsigned)((T + 2 - KS1) * 512), (unsigned)(C * 16));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ABL != 1) {
          const f16x8 ah0 = __builtin_bit_cast(f16x8, afr[T % 3][0][0]), am0 = __builtin_bit_cast(f16x8, afr[T % 3][0][1]);
          const f16x8 ah1 = __builtin_bit_cast(f16x8, afr[T % 3][1][0]), am1 = __builtin_bit_cast(f16x8, afr[T % 3][1][1]);
          // smallest terms first: m h', h m', h h'; the two blocks alternate (dependent accumulators two issues apart)
          if constexpr (T < KS1) {                               // GEMM 1: acc1[block][ct] += W1[chunk rows] x^T, k-step T
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              acc1[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am0, xh[ct][T], acc1[0][ct], 0, 0, 0);
              acc1[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am1, xh[ct][T], acc1[1][ct], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              acc1[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xm[ct][T], acc1[0][ct], 0, 0, 0);
              acc1[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xm[ct][T], acc1[1][ct], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              acc1[0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh[ct][T], acc1[0][ct], 0, 0, 0);
              acc1[1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh[ct][T], acc1[1][ct], 0, 0, 0);
            }
          } else {                                               // GEMM 2: acc2[blocks 2 t, 2 t + 1][ct] += W2[rows][k-step c] h^T
            constexpr int B0 = 2 * (T - KS1), B1 = B0 + 1;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              acc2[B0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am0, hh[ct], acc2[B0][ct], 0, 0, 0);
              acc2[B1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(am1, hh[ct], acc2[B1][ct], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              acc2[B0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, hm[ct], acc2[B0][ct], 0, 0, 0);
              acc2[B1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, hm[ct], acc2[B1][ct], 0, 0, 0);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              acc2[B0][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, hh[ct], acc2[B0][ct], 0, 0, 0);
              acc2[B1][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, hh[ct], acc2[B1][ct], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };

      ml_static_for<0, KS1>(slot);
      if constexpr (!DB) {
        __syncthreads();                                         // this chunk's W2 part is complete; everybody is done with the W1 part
        read_batch(afr[KS1 % 3], a2, (unsigned)(C * 16));
        read_batch(afr[(KS1 + 1) % 3], a2 + 512u, (unsigned)(C * 16));
      }

      // ---- hidden activations of this chunk: bias, activation, running row scale, two fp16 parts = the B operand of GEMM 2
      if (ABL == 1) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+v"(acc1[q][ct]));
      }
      if (ABL == 3) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          hh[ct] = __builtin_bit_cast(f16x8, acc1[0][ct]);
          hm[ct] = __builtin_bit_cast(f16x8, acc1[1][ct]);
        }
      } else {
        const int hf = 32 * c + 4 * g;
        const f32x4 wi0 = *reinterpret_cast<const f32x4*>(w1inv_lds + hf), wi1 = *reinterpret_cast<const f32x4*>(w1inv_lds + hf + 16);
        const f32x4 bi0 = *reinterpret_cast<const f32x4*>(b1_lds + hf), bi1 = *reinterpret_cast<const f32x4*>(b1_lds + hf + 16);
        f32x4 v0[CT], v1[CT];
        bool need = false;
        int enew[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          v0[ct] = (acc1[0][ct] * sx_inv[ct]) * wi0 + bi0;          // two exact unscalings, then the bias
          v1[ct] = (acc1[1][ct] * sx_inv[ct]) * wi1 + bi1;
          if (ACT == ML_ACT_RELU) {                               // NaN-propagating maximum (v_maximum3_f32), as torch's relu
            v0[ct] = __builtin_elementwise_maximum(v0[ct], (f32x4){0.f, 0.f, 0.f, 0.f});
            v1[ct] = __builtin_elementwise_maximum(v1[ct], (f32x4){0.f, 0.f, 0.f, 0.f});
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v0[ct][e] = l3_gelu(v0[ct][e]);
              v1[ct][e] = l3_gelu(v1[ct][e]);
            }
          }
          const unsigned mk = l3_row_max(l3_absmax8(v0[ct], v1[ct]));
          enew[ct] = max(-100, min((int)((mk >> 23) & 255u) - 127, 128));   // 2^e <= max < 2^(e+1)
          need = need || (enew[ct] > eset[ct] + 2);
        }
        if (__builtin_amdgcn_ballot_w64(need) != 0) {              // rare after the first chunk
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const bool mine = enew[ct] > eset[ct] + 2;
            const int en = mine ? enew[ct] : eset[ct];
            const float ratio = __builtin_bit_cast(float, (unsigned)(127 + max(eset[ct] - en, -126)) << 23);   // 2^(old - new) <= 1
#pragma unroll
            for (int ob = 0; ob < NOB; ++ob) acc2[ob][ct] *= ratio;
            eset[ct] = en;
            sh[ct] = __builtin_bit_cast(float, (unsigned)(127 + 12 - en) << 23);
            sh_inv[ct] = __builtin_bit_cast(float, (unsigned)(127 - 12 + en) << 23);
          }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) l3_split8(v0[ct], v1[ct], sh[ct], hh[ct], hm[ct]);
      }
      __builtin_amdgcn_sched_barrier(0);

      ml_static_for<KS1, NBAT>(slot);
    }

    // ---- y: D[i = feature][j = row]: a lane holds four consecutive features of its rows
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const int m = row0 + 16 * ct + j;
      const unsigned rowoff = m < M ? (unsigned)m * (unsigned)C * 4u : 0xFFFFFFF0u;   // out of range: loads give 0, stores are dropped
      // the residual values of a row come in batches of up to RBAT blocks, requested together: a load inside the per-block loop is
      // compiled into load / wait / store, one memory latency per block
      constexpr int RBAT = NOB < 8 ? NOB : 8;
      f32x4 resv[RBAT];
      auto load_res = [&](int ob0) __attribute__((always_inline)) {
        if (with_res) {
#pragma unroll
          for (int q = 0; q < RBAT; ++q) {
            const int f = min(ob0 + q, NOB - 1) * 16 + 4 * g;
            const unsigned offc = m < M ? rowoff + (unsigned)f * 4u : 0xFFFFFFF0u;
            // (parked rows: written by other lanes of this wave -> glc, from L2)
            resv[q] = res_normed ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, offc, 0, 1))
                                 : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, offc, 0, 0));
          }
        } else {
#pragma unroll
          for (int q = 0; q < RBAT; ++q) resv[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      };
      if (!with_pln) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
          if (ob % RBAT == 0) load_res(ob);
          const int f = ob * 16 + 4 * g;
          const f32x4 wi = *reinterpret_cast<const f32x4*>(w2inv_lds + f);
          const f32x4 bi = *reinterpret_cast<const f32x4*>(b2_lds + f);
          f32x4 v = (acc2[ob][ct] * sh_inv[ct]) * wi + bi;
          const unsigned offc = m < M ? rowoff + (unsigned)f * 4u : 0xFFFFFFF0u;
          if (with_res) v += resv[ob % RBAT];
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, offc, 0, 0);
        }
      } else {
        // post-norm: LayerNorm over the finished row (bias and residual added), two-pass statistics over the row's C values -- 4 NOB
        // in this lane, the rest in the three other lanes of the row --, then weight / bias; optionally a second output + padd
        float sm = 0.f;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
          if (ob % RBAT == 0) load_res(ob);
          const int f = ob * 16 + 4 * g;
          const f32x4 wi = *reinterpret_cast<const f32x4*>(w2inv_lds + f);
          const f32x4 bi = *reinterpret_cast<const f32x4*>(b2_lds + f);
          f32x4 v = (acc2[ob][ct] * sh_inv[ct]) * wi + bi;
          const unsigned offc = m < M ? rowoff + (unsigned)f * 4u : 0xFFFFFFF0u;
          if (with_res) v += resv[ob % RBAT];
          acc2[ob][ct] = v;
          if (a.dual) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, offc, 0, 0);
          sm += (v[0] + v[1]) + (v[2] + v[3]);
        }
        const float mean = ml_row_sum(sm) * (1.0f / C);
        float sq = 0.f;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
          acc2[ob][ct] -= mean;
#pragma unroll
          for (int e = 0; e < 4; ++e) sq = fmaf(acc2[ob][ct][e], acc2[ob][ct][e], sq);
        }
        const float rstd = 1.0f / sqrtf(ml_row_sum(sq) * (1.0f / C) + a.pln_eps);
        const unsigned prow = a.padd ? (unsigned)(min(m, M - 1) % a.padd_rows) * (unsigned)C * 4u : 0u;
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
          const int f = ob * 16 + 4 * g;
          const f32x4 gm = *reinterpret_cast<const f32x4*>(plng_lds + f);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(plnb_lds + f);
          const f32x4 y = (acc2[ob][ct] * rstd) * gm + bt;
          const unsigned offc = m < M ? rowoff + (unsigned)f * 4u : 0xFFFFFFF0u;
          if (a.dual) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), y2rs, offc, 0, 0);
            continue;
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), yrs, offc, 0, 0);
          if (a.Y2) {
            f32x4 y2 = y;
            if (a.padd) y2 += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pars, prow + (unsigned)f * 4u, 0, 0));
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y2), y2rs, offc, 0, 0);
          }
        }
      }
    }
  }
  if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last chunk requested chunk 0 again: nothing may land after the exit)
}

// -----------------------------------------------------------------------------------------------------------------------------------
// The same MLP with the two waves of every SIMD HALF A CHUNK APART (round 6; C = 128 / 192 / 256, 8 waves, 16 rows per wave).
// What the probes of round 6 measured (profiles/r06_mfma_overlap_probe_v1.txt): one wave saturates its SIMD's matrix pipe; vector work
// between a wave's OWN matrix instructions adds to its time; a vector-busy wave BESIDE a matrix-busy wave overlaps it.  In the kernel
// above both waves of a SIMD run the same phase at the same time (one barrier per chunk): first both want the matrix pipe, then both
// do the activation / split of the hidden values -- the chunk time was the SUM of the matrix, LDS-read and vector parts (7.2k clocks
// for 3.1k of matrix instructions).  Here a chunk is two PHASES -- P1 = GEMM 1 + activation, P2 = GEMM 2 -- with a barrier behind each,
// waves 0-3 (one per SIMD) run phase s in slot s and waves 4-7 phase s - 1: beside every P1 runs a P2.  The wave in P1 has the higher
// priority: its 6 KS1 matrix instructions go first, its activation then runs under the other wave's GEMM 2.
// The weights stream through a ring of FOUR half-images (W1 rows of a chunk | W2 k-step of a chunk, 8 C units each) by LDS-DMA: in
// slot s every wave requests its share of half-image s + 2 (its ring slot was last read in slot s - 1) and waits, in front of the
// slot's barrier, for everything but those requests (memory reads return in order).  Row groups follow each other without a gap:
// a wave stores group k's rows and loads group k + 1's at the head of its first phase of group k + 1, under its partner's GEMM 2.
#ifdef UNIVS_TRACE_MLP            // instrumented build (`--ablate mlp_trace`): s_memtime stamps of workgroup 0's waves in their first slots
constexpr int ML_TR_SLOTS = 24, ML_TR_ST = 6;
static __device__ unsigned long long g_ml_trace[8 * ML_TR_SLOTS * ML_TR_ST];
#define ML_TR(slot, k)                                                                                                  \
  do {                                                                                                                  \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (slot) < ML_TR_SLOTS)                                             \
      g_ml_trace[((threadIdx.x >> 6) * ML_TR_SLOTS + (slot)) * ML_TR_ST + (k)] = __builtin_amdgcn_s_memtime();          \
  } while (0)
#else
#define ML_TR(slot, k) do { } while (0)
#endif

#if defined(ML_PS_ABL) && (ML_PS_ABL & 1)
__device__ __forceinline__ f32x4 ml_ps_fake_mfma(f16x8 a_, f16x8 b_, f32x4 c_) {
  asm volatile("" : "+v"(c_) : "v"(a_), "v"(b_));
  return c_;
}
#define ML_PS_MMA(a_, b_, c_) ml_ps_fake_mfma(a_, b_, c_)
#else
#define ML_PS_MMA(a_, b_, c_) __builtin_amdgcn_mfma_f32_16x16x32_f16(a_, b_, c_, 0, 0, 0)
#endif

template <int KS1, int ACT>
__global__ __launch_bounds__(512, 1) void mlp_f16x3_ps(const MlpArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 Lds[];
  constexpr int THREADS = 512, NW = 8, C = 32 * KS1, NOB = C / 16, NRING = 4;
  constexpr int HU = 8 * C;                                      // 16-byte units of a half-image
  constexpr int UPH = HU / THREADS;                              // ... per thread
  constexpr int RG = NW * 16;                                    // rows per workgroup round
  static_assert(HU % THREADS == 0 && NOB == 2 * KS1 && KS1 >= 3 && UPH <= 8, "geometry");
  const int M = a.M, Hd = a.Hd, NCH = Hd >> 5, nwg = a.nwg;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  float* b1_lds = reinterpret_cast<float*>(Lds + NRING * HU);
  float* w1inv_lds = b1_lds + Hd;
  float* b2_lds = w1inv_lds + Hd;
  float* w2inv_lds = b2_lds + C;
  float* lng_lds = w2inv_lds + C;
  float* lnb_lds = lng_lds + C;
  float* plng_lds = lnb_lds + C;
  float* plnb_lds = plng_lds + C;
  const bool with_ln = a.ln_g != nullptr;                        // uniform
  const bool with_pln = a.pln_g != nullptr;                      // uniform
  const int ngroups = (M + RG - 1) / RG;
  if ((int)blockIdx.x >= ngroups) return;
  const int G = (ngroups - 1 - (int)blockIdx.x) / nwg + 1;       // row groups of this workgroup
  const int P = 2 * NCH * G;                                     // phases of a wave

  for (int r = tid; r < Hd; r += THREADS) {
    b1_lds[r] = a.b1 ? a.b1[r] : 0.f;
    w1inv_lds[r] = a.w1inv[r];
  }
  for (int r = tid; r < C; r += THREADS) {
    b2_lds[r] = a.b2 ? a.b2[r] : 0.f;
    w2inv_lds[r] = a.w2inv[r];
    lng_lds[r] = with_ln ? a.ln_g[r] : 1.f;
    lnb_lds[r] = (with_ln && a.ln_b) ? a.ln_b[r] : 0.f;
    plng_lds[r] = with_pln ? a.pln_g[r] : 1.f;
    plnb_lds[r] = (with_pln && a.pln_b) ? a.pln_b[r] : 0.f;
  }
  // half-image p: part p & 1 (0: W1 rows [32 c, 32 c + 32), 1: W2 k-step c) of chunk c = (p / 2) % NCH; unit i = tid + THREADS v
  auto w_src = [&](int p, int v) __attribute__((always_inline)) -> const u32x4* {
    const int i = tid + THREADS * v;
    const int c = (p >> 1) % NCH;
    return (p & 1) ? a.W2p + ((size_t)c * HU + i) : a.W1p + ((size_t)(i >> 5) * Hd + 32 * c + (i & 31));
  };
  {
    u32x4 w0[2 * UPH];                                           // half-images 0 and 1 (chunk 0) through registers
#pragma unroll
    for (int v = 0; v < 2 * UPH; ++v) w0[v] = *w_src(v / UPH, v % UPH);
#pragma unroll
    for (int v = 0; v < 2 * UPH; ++v) Lds[(v / UPH) * HU + tid + THREADS * (v % UPH)] = w0[v];
  }
  __syncthreads();                                               // parameters and half-images 0, 1 are in place
  const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Lds);

  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.X), 0, (int)((long long)M * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(a.Y, 0, (int)((long long)M * C * 4), 0x00020000);
  const bool res_normed = a.res_normed != 0;                      // uniform
  const bool with_res = a.Res != nullptr || res_normed;
  const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(res_normed ? a.Y : a.Res ? a.Res : a.X), 0, (int)((long long)M * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t y2rs = __builtin_amdgcn_make_buffer_rsrc(a.Y2 ? a.Y2 : a.Y, 0, (int)((long long)M * C * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t pars = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.padd ? a.padd : a.X), 0, (int)((long long)(a.padd ? a.padd_rows : M) * C * 4), 0x00020000);

  constexpr int AR = 3;                           // fragment batches in flight + 1 (C = 256: registers allow one in flight)
  u32x4 afr[AR][2][2];                                           // [ring][block][part]
  auto read_batch = [&](u32x4 (&d)[2][2], unsigned addr, unsigned pstride) __attribute__((always_inline)) {
#ifdef ML_PS_ABL
    if (ML_PS_ABL & 2) return;
#endif
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const unsigned ah = addr + (unsigned)(q * 256);
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(d[q][0]), "=&v"(d[q][1]) : "v"(ah), "v"(ah + pstride) : "memory");
    }
  };
  const unsigned a1_lane = (unsigned)((g * 64 + j) * 16);        // W1 half-image: ((t 4 + g) 2 + part) 32 + 16 q + j units
  const unsigned a2_lane = (unsigned)((g * 2 * C + j) * 16);     // W2 half-image: (g 2 + part) C + 16 (2 t + q) + j units

  // ---- the state of a row group
  f16x8 xh[KS1], xm[KS1], hh, hm;
  float sx_inv = 1.f, sh = 1.f, sh_inv = 1.f;
  int eset = -1000;
  f32x4 acc1[2], acc2[NOB];

  auto group_row0 = [&](int gi) __attribute__((always_inline)) { return ((int)blockIdx.x + gi * nwg) * RG + wave * 16; };

  // x tile of a group: read once, optional LayerNorm, exact row maximum, two fp16 parts in registers
  auto prologue = [&](int gi) __attribute__((always_inline)) {
    const int row0 = group_row0(gi);
    const int m = min(row0 + j, M - 1);                           // rows past the end repeat the last row (not stored)
    const unsigned vo = ((unsigned)m * (unsigned)C + (unsigned)(8 * g)) * 4u;
    f32x4 raw[KS1][2];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      raw[ks][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, ks * 128, 0));
      raw[ks][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo + 16u, ks * 128, 0));
    }
    if (with_ln) {
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int e = 0; e < 4; ++e) sm += raw[ks][0][e] + raw[ks][1][e];
      const float mean = ml_row_sum(sm) * (1.0f / C);
      float sq = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          raw[ks][h2] -= mean;
#pragma unroll
          for (int e = 0; e < 4; ++e) sq = fmaf(raw[ks][h2][e], raw[ks][h2][e], sq);
        }
      const float rstd = 1.0f / sqrtf(ml_row_sum(sq) * (1.0f / C) + a.ln_eps);
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const f32x4 gm = *reinterpret_cast<const f32x4*>(lng_lds + 32 * ks + 8 * g + 4 * h2);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(lnb_lds + 32 * ks + 8 * g + 4 * h2);
          raw[ks][h2] = (raw[ks][h2] * rstd) * gm + bt;
          if (res_normed)                                        // park the residual row (rows past the end: dropped)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, raw[ks][h2]), yrs,
                                                   row0 + j < M ? vo + (unsigned)(16 * h2) : 0xFFFFFFF0u, ks * 128, 0);
        }
    }
    unsigned mx = 0u;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) mx = max(mx, l3_absmax8(raw[ks][0], raw[ks][1]));
    mx = l3_row_max(mx);
    float s_, inv;
    l3_scale(mx, 14, s_, inv);
    sx_inv = inv;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) l3_split8(raw[ks][0], raw[ks][1], s_, xh[ks], xm[ks]);
  };

  // y of a group: D[i = feature][j = row]: a lane holds four consecutive features of its row
  auto epilogue = [&](int gi) __attribute__((always_inline)) {
    const int m = group_row0(gi) + j;
    const unsigned rowoff = m < M ? (unsigned)m * (unsigned)C * 4u : 0xFFFFFFF0u;   // out of range: loads give 0, stores are dropped
    constexpr int RBAT = NOB < 8 ? NOB : 8;
    f32x4 resv[RBAT];
    auto load_res = [&](int ob0) __attribute__((always_inline)) {
      if (with_res) {
#pragma unroll
        for (int q = 0; q < RBAT; ++q) {
          const int f = min(ob0 + q, NOB - 1) * 16 + 4 * g;
          const unsigned offc = m < M ? rowoff + (unsigned)f * 4u : 0xFFFFFFF0u;
          resv[q] = res_normed ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, offc, 0, 1))
                               : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, offc, 0, 0));
        }
      } else {
#pragma unroll
        for (int q = 0; q < RBAT; ++q) resv[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    };
    if (!with_pln) {
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        if (ob % RBAT == 0) load_res(ob);
        const int f = ob * 16 + 4 * g;
        const f32x4 wi = *reinterpret_cast<const f32x4*>(w2inv_lds + f);
        const f32x4 bi = *reinterpret_cast<const f32x4*>(b2_lds + f);
        f32x4 v = (acc2[ob] * sh_inv) * wi + bi;
        const unsigned offc = m < M ? rowoff + (unsigned)f * 4u : 0xFFFFFFF0u;
        if (with_res) v += resv[ob % RBAT];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, offc, 0, 0);
      }
    } else {
      float sm = 0.f;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        if (ob % RBAT == 0) load_res(ob);
        const int f = ob * 16 + 4 * g;
        const f32x4 wi = *reinterpret_cast<const f32x4*>(w2inv_lds + f);
        const f32x4 bi = *reinterpret_cast<const f32x4*>(b2_lds + f);
        f32x4 v = (acc2[ob] * sh_inv) * wi + bi;
        const unsigned offc = m < M ? rowoff + (unsigned)f * 4u : 0xFFFFFFF0u;
        if (with_res) v += resv[ob % RBAT];
        acc2[ob] = v;
        if (a.dual) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, offc, 0, 0);
        sm += (v[0] + v[1]) + (v[2] + v[3]);
      }
      const float mean = ml_row_sum(sm) * (1.0f / C);
      float sq = 0.f;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        acc2[ob] -= mean;
#pragma unroll
        for (int e = 0; e < 4; ++e) sq = fmaf(acc2[ob][e], acc2[ob][e], sq);
      }
      const float rstd = 1.0f / sqrtf(ml_row_sum(sq) * (1.0f / C) + a.pln_eps);
      const unsigned prow = a.padd ? (unsigned)(min(m, M - 1) % a.padd_rows) * (unsigned)C * 4u : 0u;
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) {
        const int f = ob * 16 + 4 * g;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(plng_lds + f);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(plnb_lds + f);
        const f32x4 y = (acc2[ob] * rstd) * gm + bt;
        const unsigned offc = m < M ? rowoff + (unsigned)f * 4u : 0xFFFFFFF0u;
        if (a.dual) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), y2rs, offc, 0, 0);
          continue;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y), yrs, offc, 0, 0);
        if (a.Y2) {
          f32x4 y2 = y;
          if (a.padd) y2 += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(pars, prow + (unsigned)f * 4u, 0, 0));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, y2), y2rs, offc, 0, 0);
        }
      }
    }
  };

  // half-image p -> ring slot p & 3 by LDS-DMA, requested by the FOUR waves that run a P2 in that slot (wave & 3 takes a quarter: 2 UPH
  // pieces of 1 KB): issuing a piece costs a wave 100 - 150 clocks of its instruction stream (traced), and the wave in P1 -- GEMM 1,
  // then the activation -- is the slot's critical path
  auto request_half = [&](int p) __attribute__((always_inline)) {
    const int c = (p >> 1) % NCH;
    const unsigned dst0 = lds_base + (unsigned)((p & (NRING - 1)) * HU * 16);
#pragma unroll
    for (int v = 0; v < 2 * UPH; ++v) {
      const int piece = (wave & 3) * 2 * UPH + v;
      const int i = piece * 64 + lane;
      const u32x4* src = (p & 1) ? a.W2p + ((size_t)c * HU + i) : a.W1p + ((size_t)(i >> 5) * Hd + 32 * c + (i & 31));
      ml_glds16(src, dst0 + (unsigned)(piece * 1024));
    }
  };

  // P1 of chunk c: acc1[block] = W1[chunk rows] x^T over KS1 k-steps, then bias, activation, running row scale, two fp16 parts
  int sl = 0;                                                    // the slot this wave is in
  auto phase1 = [&](int c, int ring) __attribute__((always_inline)) {
    ML_TR(sl, 0);
    const unsigned a1 = a1_lane + (unsigned)(ring * HU * 16);
    read_batch(afr[0], a1, 512u);
    if constexpr (AR == 3) read_batch(afr[1], a1 + 4096u, 512u);
    acc1[0] = acc1[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ml_static_for<0, KS1>([&](auto tc) __attribute__((always_inline)) {
      constexpr int T = decltype(tc)::value;
      // AR == 3: batch T + 2 is requested behind the wait for batch T; AR == 2: batch T + 1 in front of it (its slot was batch T - 1's)
      if constexpr (AR == 2 && T + 1 < KS1) read_batch(afr[(T + 1) % AR], a1 + (unsigned)((T + 1) * 4096), 512u);
      ml_wait_lgkm<(T + 1 < KS1 ? 4 : 0)>(afr[T % AR]);
      if constexpr (AR == 3 && T + 2 < KS1) read_batch(afr[(T + 2) % AR], a1 + (unsigned)((T + 2) * 4096), 512u);
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 ah0 = __builtin_bit_cast(f16x8, afr[T % AR][0][0]), am0 = __builtin_bit_cast(f16x8, afr[T % AR][0][1]);
      const f16x8 ah1 = __builtin_bit_cast(f16x8, afr[T % AR][1][0]), am1 = __builtin_bit_cast(f16x8, afr[T % AR][1][1]);
      acc1[0] = ML_PS_MMA(am0, xh[T], acc1[0]);   // smallest terms first: m h', h m', h h'
      acc1[1] = ML_PS_MMA(am1, xh[T], acc1[1]);
      acc1[0] = ML_PS_MMA(ah0, xm[T], acc1[0]);
      acc1[1] = ML_PS_MMA(ah1, xm[T], acc1[1]);
      acc1[0] = ML_PS_MMA(ah0, xh[T], acc1[0]);
      acc1[1] = ML_PS_MMA(ah1, xh[T], acc1[1]);
      __builtin_amdgcn_sched_barrier(0);
    });
    ML_TR(sl, 1);
    const int hf = 32 * c + 4 * g;
    const f32x4 wi0 = *reinterpret_cast<const f32x4*>(w1inv_lds + hf), wi1 = *reinterpret_cast<const f32x4*>(w1inv_lds + hf + 16);
    const f32x4 bi0 = *reinterpret_cast<const f32x4*>(b1_lds + hf), bi1 = *reinterpret_cast<const f32x4*>(b1_lds + hf + 16);
    f32x4 v0 = (acc1[0] * sx_inv) * wi0 + bi0;                   // two exact unscalings, then the bias
    f32x4 v1 = (acc1[1] * sx_inv) * wi1 + bi1;
    if (ACT == ML_ACT_RELU) {                                    // NaN-propagating maximum, as torch's relu
      v0 = __builtin_elementwise_maximum(v0, (f32x4){0.f, 0.f, 0.f, 0.f});
      v1 = __builtin_elementwise_maximum(v1, (f32x4){0.f, 0.f, 0.f, 0.f});
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = l3_gelu(v0[e]);
        v1[e] = l3_gelu(v1[e]);
      }
    }
    const unsigned mk = l3_row_max(l3_absmax8(v0, v1));
    const int enew = max(-100, min((int)((mk >> 23) & 255u) - 127, 128));   // 2^e <= max < 2^(e+1)
    const bool need = enew > eset + 2;
    if (__builtin_amdgcn_ballot_w64(need) != 0) {                // rare after the first chunk
      const int en = need ? enew : eset;
      const float ratio = __builtin_bit_cast(float, (unsigned)(127 + max(eset - en, -126)) << 23);   // 2^(old - new) <= 1
#pragma unroll
      for (int ob = 0; ob < NOB; ++ob) acc2[ob] *= ratio;
      eset = en;
      sh = __builtin_bit_cast(float, (unsigned)(127 + 12 - en) << 23);
      sh_inv = __builtin_bit_cast(float, (unsigned)(127 - 12 + en) << 23);
    }
    l3_split8(v0, v1, sh, hh, hm);
    ML_TR(sl, 2);
  };

  // P2 of chunk c: acc2[blocks] += W2[rows][k-step c] h^T
  auto phase2 = [&](int ring, int next_half) __attribute__((always_inline)) {
    ML_TR(sl, 0);
    const unsigned a2 = a2_lane + (unsigned)(ring * HU * 16);
    read_batch(afr[0], a2, (unsigned)(C * 16));
    if constexpr (AR == 3) read_batch(afr[1], a2 + 512u, (unsigned)(C * 16));
    if (next_half >= 0) request_half(next_half);
    ml_static_for<0, KS1>([&](auto tc) __attribute__((always_inline)) {
      constexpr int T = decltype(tc)::value;
      constexpr int B0 = 2 * T, B1 = B0 + 1;
      if constexpr (AR == 2 && T + 1 < KS1) read_batch(afr[(T + 1) % AR], a2 + (unsigned)((T + 1) * 512), (unsigned)(C * 16));
      ml_wait_lgkm<(T + 1 < KS1 ? 4 : 0)>(afr[T % AR]);
      if constexpr (AR == 3 && T + 2 < KS1) read_batch(afr[(T + 2) % AR], a2 + (unsigned)((T + 2) * 512), (unsigned)(C * 16));
      __builtin_amdgcn_sched_barrier(0);
      const f16x8 ah0 = __builtin_bit_cast(f16x8, afr[T % AR][0][0]), am0 = __builtin_bit_cast(f16x8, afr[T % AR][0][1]);
      const f16x8 ah1 = __builtin_bit_cast(f16x8, afr[T % AR][1][0]), am1 = __builtin_bit_cast(f16x8, afr[T % AR][1][1]);
      acc2[B0] = ML_PS_MMA(am0, hh, acc2[B0]);
      acc2[B1] = ML_PS_MMA(am1, hh, acc2[B1]);
      acc2[B0] = ML_PS_MMA(ah0, hm, acc2[B0]);
      acc2[B1] = ML_PS_MMA(ah1, hm, acc2[B1]);
      acc2[B0] = ML_PS_MMA(ah0, hh, acc2[B0]);
      acc2[B1] = ML_PS_MMA(ah1, hh, acc2[B1]);
      __builtin_amdgcn_sched_barrier(0);
    });
    ML_TR(sl, 1);
    ML_TR(sl, 2);
  };

  // a slot's end.  A wave requests weights only in its P2 slots (half-image slot + 2) and needs them landed one slot later: it waits
  // for all its requests at the end of its P1 slots and for nothing at the end of its P2 slots
  auto slot_end = [&](bool wait_all) __attribute__((always_inline)) {
    ML_TR(sl, 3);
    if (wait_all) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ML_TR(sl, 4);
    __builtin_amdgcn_s_barrier();
    ML_TR(sl, 5);
  };
  const int late = wave >> 2;                                    // waves 4-7 run one slot behind waves 0-3
  prologue(0);
  if (late) {                                                    // slot 0 of the late waves: half-image 2 (nobody runs a P2 in slot 0)
    if (2 < P) request_half(2);
    slot_end(true);
    ++sl;
  }
#pragma unroll 1
  for (int gi = 0; gi < G; ++gi) {
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) acc2[ob] = (f32x4){0.f, 0.f, 0.f, 0.f};
    eset = -1000;
    sh = sh_inv = 1.0f;
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
      const int ph = (gi * NCH + c) * 2;                         // = sl - late
      __builtin_amdgcn_s_setprio(1);                             // P1: my matrix instructions first, my activation under the partner's GEMM 2
      phase1(c, ph & (NRING - 1));
      slot_end(true);
      ++sl;
      const int nh = sl + 2 < P ? sl + 2 : -1;
      __builtin_amdgcn_s_setprio(0);
      phase2((ph + 1) & (NRING - 1), nh);
      slot_end(false);
      ++sl;
    }
    // the seam between two row groups sits at the head of this wave's next slot: store group gi, load group gi + 1
    epilogue(gi);
    if (gi + 1 < G) prologue(gi + 1);
  }
  if (!late) slot_end(true);                                     // slot P of the early waves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


static int ml_cus() {
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

template <int KS1, int CT, int ACT, int NW, int ABL, bool DB = true>
static int ml_launch1(MlpArgs a, size_t lds, int ngroups, hipStream_t st) {
  const void* fn = reinterpret_cast<const void*>(&mlp_f16x3<KS1, CT, ACT, NW, ABL, DB>);
  static int per_cu = 0;                                          // resident workgroups per CU (LDS and registers)
  if (per_cu == 0) {
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 64 * NW, lds) != hipSuccess || nb < 1) {
      (void)hipGetLastError();
      nb = 1;
    }
    per_cu = nb;
  }
  a.nwg = std::min(ml_cus() * per_cu, ngroups);
  hipLaunchKernelGGL((mlp_f16x3<KS1, CT, ACT, NW, ABL, DB>), dim3((unsigned)a.nwg), dim3(64 * NW), lds, st, a);
  return check_launch("mlp_f16x3");
}

template <int KS1, int CT, int NW, bool DB = true>
static int ml_launch(const MlpArgs& a, int act, hipStream_t st) {
  constexpr int C = 32 * KS1;
  constexpr int RG = NW * 16 * CT;
  const int ngroups = (a.M + RG - 1) / RG;
  const size_t lds = (size_t)(DB ? 2 : 1) * 16 * C * 16 + (size_t)(2 * a.Hd + 6 * C) * 4;
  if (lds > 160 * 1024) return UNIVS_ERR_NOT_IMPLEMENTED;
  const int abl = config().linear_ablate;                         // 2 / 3 / 4: timing experiments (encoder FFN and Swin stage 1 only)
  if (abl >= 2 && abl <= 4 && ((KS1 == 8 && act == ML_ACT_RELU) || (KS1 == 3 && act == ML_ACT_GELU))) {
    if constexpr (KS1 == 8) {
      if (abl == 2) return ml_launch1<KS1, CT, ML_ACT_RELU, NW, 1>(a, lds, ngroups, st);
      if (abl == 3) return ml_launch1<KS1, CT, ML_ACT_RELU, NW, 2>(a, lds, ngroups, st);
      return ml_launch1<KS1, CT, ML_ACT_RELU, NW, 3>(a, lds, ngroups, st);
    }
    if constexpr (KS1 == 3) {
      if (abl == 2) return ml_launch1<KS1, CT, ML_ACT_GELU, NW, 1>(a, lds, ngroups, st);
      if (abl == 3) return ml_launch1<KS1, CT, ML_ACT_GELU, NW, 2>(a, lds, ngroups, st);
      return ml_launch1<KS1, CT, ML_ACT_GELU, NW, 3>(a, lds, ngroups, st);
    }
  }
  if constexpr (DB && NW == 8 && CT == 1 && KS1 >= 4 && ML_GLDS) {
    // the phase-shifted kernel (waves 4-7 half a chunk behind waves 0-3): UnivsConfig.linear_ablate = 10.  NOT the default: it measured
    // 329 us against the lockstep kernel's 321 on the encoder FFN (profiles/r06_mlp_phase_shift_v1.txt) -- kept for A / B runs
    const size_t lds_ps = (size_t)4 * 8 * C * 16 + (size_t)(2 * a.Hd + 6 * C) * 4;
    if (abl == 10 && lds_ps <= 160 * 1024) {
      MlpArgs b = a;
      b.nwg = std::min(ml_cus(), ngroups);
      const void* fn = act == ML_ACT_RELU ? reinterpret_cast<const void*>(&mlp_f16x3_ps<KS1, ML_ACT_RELU>)
                                          : reinterpret_cast<const void*>(&mlp_f16x3_ps<KS1, ML_ACT_GELU>);
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
      if (act == ML_ACT_RELU) hipLaunchKernelGGL((mlp_f16x3_ps<KS1, ML_ACT_RELU>), dim3((unsigned)b.nwg), dim3(512), lds_ps, st, b);
      else hipLaunchKernelGGL((mlp_f16x3_ps<KS1, ML_ACT_GELU>), dim3((unsigned)b.nwg), dim3(512), lds_ps, st, b);
      return check_launch("mlp_f16x3_ps");
    }
  }
  if (act == ML_ACT_RELU) return ml_launch1<KS1, CT, ML_ACT_RELU, NW, 0, DB>(a, lds, ngroups, st);
  return ml_launch1<KS1, CT, ML_ACT_GELU, NW, 0, DB>(a, lds, ngroups, st);
}

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED when the shape is not covered
int mlp_f16x3_f32(const float* x, const void* w1p, const float* w1inv, const float* b1, const void* w2p, const float* w2inv,
                  const float* b2, const float* residual, const float* ln_w, const float* ln_b, float ln_eps, const float* pln_w,
                  const float* pln_b, float pln_eps, const float* post_add, long long post_add_rows, float* y2, float* y, long long M,
                  int C, int Hd, int act, int flags, hipStream_t st) {
  const int residual_is_normed_x = flags & 1, dual = (flags >> 1) & 1;
  if (M <= 0) return UNIVS_OK;
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if ((act != ML_ACT_RELU && act != ML_ACT_GELU) || Hd < 32 || Hd % 32 != 0 || M < 2048 || M * (long long)C * 4 >= 0x7FFFFFFFLL ||
      mis(x) || mis(w1p) || mis(w2p) || mis(y) || mis(residual) || mis(w1inv) || mis(w2inv) || mis(b1) || mis(b2) || (ln_b && !ln_w) ||
      mis(post_add) || mis(y2) || (pln_b && !pln_w) || ((post_add || y2) && !pln_w) || (residual_is_normed_x && (residual || !ln_w || x == y)) ||
      (dual && (!pln_w || !y2 || post_add || residual_is_normed_x)) || (flags & ~3) ||
      (post_add && (!y2 || post_add_rows < 1 ||
      post_add_rows * (long long)C * 4 >= 0x7FFFFFFFLL)))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  MlpArgs a{};
  a.X = x; a.W1p = reinterpret_cast<const u32x4*>(w1p); a.w1inv = w1inv; a.b1 = b1;
  a.W2p = reinterpret_cast<const u32x4*>(w2p); a.w2inv = w2inv; a.b2 = b2; a.Res = residual; a.Y = y;
  a.ln_g = ln_w; a.ln_b = ln_b; a.ln_eps = ln_eps;
  a.pln_g = pln_w; a.pln_b = pln_b; a.pln_eps = pln_eps; a.padd = post_add; a.padd_rows = (int)post_add_rows; a.Y2 = y2;
  a.M = (int)M; a.Hd = Hd; a.res_normed = residual_is_normed_x ? 1 : 0; a.dual = dual;
  switch (C) {
    case 96: return ml_launch<3, 2, 4>(a, act, st);
    case 128: return ml_launch<4, 1, 8>(a, act, st);
    case 192: return ml_launch<6, 1, 8>(a, act, st);
    case 256: return ml_launch<8, 1, 8>(a, act, st);
    case 384: return ml_launch<12, 1, 4, false>(a, act, st);
    default: return UNIVS_ERR_NOT_IMPLEMENTED;
  }
}

}  // namespace univs

#ifdef UNIVS_TRACE_MLP
extern "C" int univs_debug_mlp_trace(unsigned long long* out) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(univs::g_ml_trace), sizeof(unsigned long long) * 8 * univs::ML_TR_SLOTS * univs::ML_TR_ST);
}
#endif
