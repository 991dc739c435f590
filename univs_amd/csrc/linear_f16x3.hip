// y[M, N] = x[M, K] * W[N, K]^T + bias[N]  (nn.Linear, fp32) on the fp16 matrix cores of gfx950 in THREE passes.
//
// linear_split.hip multiplies fp32 operands as three bf16 parts each (8 + 8 + 8 mantissa bits) and keeps six of the nine
// part products.  bf16 was chosen there for its exponent range; fp16 parts carry 11 bits each, so TWO parts hold 22 - 23
// bits and THREE products (h h', h m', m h') do the work of six -- if the operands are first brought into fp16's range:
//   * every row of x and every row of W is scaled by a power of two that puts its largest magnitude in [2^14, 2^15)
//     (exact; the product is unscaled by the two inverse powers in the epilogue, exact again).  The row maximum of W is
//     taken while the workgroup stages its slab (LDS atomic max on the magnitude bits).  x is read ONCE: its rows carry a
//     RUNNING scale, set from the first k-step and lowered -- with an exact rescaling of the row's accumulators -- when a
//     later k-step would leave the range (see the kernel; a separate maximum pass cost 25 - 40 % of the kernel, a
//     look-ahead register ring no less: these Linears run close to the HBM time of x and y);
//   * x' = h + m + r with h = fp16(x') and m = fp16(x' - h), both round-to-nearest: |r| <= 2^-23 |x'| for elements within
//     2^-16 of their row's maximum, and 2^-25 absolute (scaled units, i.e. 2^-39 of the row maximum) below that (m becomes
//     subnormal);  dropped: m m' (<= 2^-24) and the representation errors (<= 2^-23 each): <= 2^-21.7 per product, against
//     3 x 2^-24 = 2^-22.4 for the six-term bf16 split.  Both are below the rounding error of an fp32 FMA chain over K >= 96
//     terms: measured against fp64 on the operator's shapes the result is as close as ATen's own fp32 GEMM or closer
//     (tests/test_ops_gpu.py: test_linear_split_matches_torch runs for both splits);
//   * products accumulate in fp32 on v_mfma_f32_16x16x32_f16; Inf / NaN inputs poison their own row only (as in a GEMM):
//     the row maximum ignores NaN (maxNum) and an infinite maximum scales the finite elements to ~0, the row's results
//     are Inf / NaN either way.
// Everything else is the organisation of linear_bf16x6 (linear_split.hip): W slab resident in LDS in A-fragment order
// (4 bytes per element instead of 6: 128 output features of K = 256 per pass, N = 256 in two passes instead of three),
// x streamed through a register ring, D[i = feature][j = row] so that a lane stores four consecutive features of one row,
// epilogues ReLU / GELU / residual / column-blocked output.
#include "common.h"
#include "config.h"
#include "f16x3.h"

#include <algorithm>
#include <cstdlib>

namespace univs {

#ifdef UNIVS_TRACE_GEMM
UNIVS_GT_DECL(g_l3_trace);
#endif
constexpr int L3_THREADS = 512;   // 8 waves, two per SIMD
constexpr int L3_TILE_M = 32;     // rows of x per wave tile (two 16-column MFMA tiles)
constexpr int L3_MAX_RB = 8;
constexpr int L3_WPT = 24;        // PRE: 16-byte units of the W slab per thread (the host side checks that the slab fits)
enum { L3_EPI_NONE = 0, L3_EPI_RELU = 1, L3_EPI_GELU = 2, L3_EPI_RESIDUAL = 3, L3_EPI_BLOCKED = 4 };   // = LS_EPI_*

// LDS: Wsp [K/32][4 k-groups][2 parts][16 RB features] 16 B | bias[R] | winv[R] | zero tail (16 x 16 B) | wmax[R]
// PRE: W is the pre-split image of presplit_f16x3 (gemm_f16x3_stream.hip: [(K / 8) x 2 parts][N] 16-byte units, `winv_g` its
// inverse row scales) and staging the slab is a COPY.  Splitting the slab inside every workgroup (row maxima by LDS atomics,
// then scale + split: two dependent sweeps over the slab) measured 28 - 32 k clocks = 13 - 15 us at the head of EVERY launch --
// a quarter of an 18 400-row Swin stage-3 Linear, 30 % of a 96 600-row encoder projection (profiles/r05_gemm_phase_trace_v1.txt).
template <int RB, int RING, bool PRE>   // RING: register stages of x (K / 32 is a multiple)
__global__ __launch_bounds__(L3_THREADS, 1) void linear_f16x3(const float* __restrict__ X,      // [M, K]
                                                               const float* __restrict__ W,      // [N, K] (PRE: the split image)
                                                               const float* __restrict__ bias,   // [N] or null
                                                               const float* __restrict__ Res,    // [M, N] (epi == RESIDUAL)
                                                               float* __restrict__ Y,            // [M, N]
                                                               int M, int N, int K, int rows_per_pass, int epi, int blk_rows,
                                                               int blk_cols, int ablate, const float* __restrict__ winv_g) {
  extern __shared__ __attribute__((aligned(16))) u32x4 Wsp[];
  [[maybe_unused]] const int gts = UNIVS_GT_SLOT();
  UNIVS_GT(g_l3_trace, gts, 0);
  UNIVS_GT_REAL(g_l3_trace, gts, 62);
  // (row range, pass) of this workgroup.  Workgroups go to the 8 XCDs round-robin by linear id; the remap gives every XCD a CONTIGUOUS
  // chunk of the sequence (row range major, pass minor): the passes of a row range -- which stream the same rows of x -- run on ONE XCD
  // and meet in its L2, whatever the grid extents (with 12 passes over 21 row ranges every XCD used to fetch nearly all of x itself).
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (ablate != 5 && gridDim.y > 1) {
    const unsigned lw = xcd_remap(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    bx = lw / gridDim.y;
    by = lw - bx * gridDim.y;
  }
  const int n0 = by * rows_per_pass;
  const int R = min(rows_per_pass, N - n0);                      // a multiple of 4 (host-checked)
  constexpr int Rp = 16 * RB;                                    // rows of the LDS image (rows >= R are never written: their
                                                                 // products only reach features that are never stored)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = K >> 5;
  const int j = lane & 15, g = lane >> 4;
  float* bias_lds = reinterpret_cast<float*>(Wsp + (size_t)(K >> 3) * Rp * 2);
  float* winv_lds = bias_lds + R;
  unsigned* tail = reinterpret_cast<unsigned*>(winv_lds + R);    // 64 dwords of zeros
  unsigned* wmax_lds = tail + 64;

  // ---- this wave's row tiles: the workgroup owns a contiguous run, its waves take tiles round-robin
  const int WT = (M + L3_TILE_M - 1) / L3_TILE_M;
  constexpr int NWV = L3_THREADS / 64;
  const int wg0 = (int)((long long)WT * bx / gridDim.x), wg1 = (int)((long long)WT * (bx + 1) / gridDim.x);
  const int wt0 = wg0 + wave;
  const int ntiles = wt0 < wg1 ? (wg1 - wt0 + NWV - 1) / NWV : 0;

  // x through buffer loads: the lane's byte offset (row, k-group) is computed once per tile, the k-step is the scalar
  // offset -- no vector arithmetic per load.  Rows past the end repeat the last row (their results are not stored).
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X), 0, (int)((long long)M * K * 4), 0x00020000);
  auto tile_voff = [&](int tile, unsigned (&vo)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int m = min((wt0 + tile * NWV) * L3_TILE_M + 16 * c + j, M - 1);
      vo[c] = ((unsigned)m * (unsigned)K + (unsigned)(8 * g)) * 4u;          // < 2^31 (host-checked)
    }
  };
  // PRE: this pass's slab of the split image, requested FIRST and all at once (<= L3_WPT 16-byte units per thread), committed to LDS
  // after the x ring has been requested behind it: vector memory returns in order, so the wait for the slab does not wait for x
  // (requested behind x, each sweep of the copy waited for the first tiles' rows to arrive from HBM: 14 - 17 k clocks instead of ~4 k).
  // run = (k-group of 8, part): R consecutive units of the image, N units apart; R threads per run, 512 / R runs per sweep.
  [[maybe_unused]] u32x4 wcp[L3_WPT];
  [[maybe_unused]] int wcp_runs = 0, wcp_rpi = 1, wcp_run0 = 0, wcp_rr = 0;
  [[maybe_unused]] bool wcp_active = false;
  if constexpr (PRE) {
    const u32x4* Wp = reinterpret_cast<const u32x4*>(W) + n0;
    wcp_runs = (K >> 3) * 2;
    wcp_rpi = L3_THREADS / R;
    wcp_run0 = tid / R;
    wcp_rr = tid - wcp_run0 * R;
    wcp_active = wcp_run0 < wcp_rpi;
#pragma unroll
    for (int u = 0; u < L3_WPT; ++u) wcp[u] = Wp[(size_t)min(wcp_run0 + u * wcp_rpi, wcp_runs - 1) * N + wcp_rr];
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned vo_cur[2], vo_next[2];
  tile_voff(0, vo_cur);
  tile_voff(min(1, max(ntiles, 1) - 1), vo_next);
  f32x4 raw[RING][2][2];                                         // [stage][column tile][16-byte half]
  auto load_x = [&](f32x4 (&buf)[2][2], const unsigned (&vo)[2], int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      buf[c][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo[c], ks * 128, 0));
      buf[c][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo[c] + 16u, ks * 128, 0));
    }
  };
#pragma unroll
  for (int u = 0; u < RING; ++u) {
    load_x(raw[u], vo_cur, u);                                   // RING <= KS
    __builtin_amdgcn_sched_barrier(0);
  }

  UNIVS_GT(g_l3_trace, gts, 1);
  // ---- this pass's rows of W -> LDS (fragment order), two sweeps: row maxima, then scale + split.  Work item = 8
  // consecutive k of one row; consecutive threads take consecutive ROWS (conflict-free 16-byte LDS writes; the reads of W
  // are 32-byte pieces of different rows, all L2 hits after the first workgroup)
  const float* Wsrc = W + (size_t)n0 * K;
  const int kch = K >> 3;
  for (int r = tid; r < R; r += L3_THREADS) {
    wmax_lds[r] = 0u;
    bias_lds[r] = bias ? bias[n0 + r] : 0.f;
    if constexpr (PRE) winv_lds[r] = winv_g[n0 + r];
  }
  for (int i = tid; i < 64; i += L3_THREADS) tail[i] = 0u;
  if constexpr (PRE) {                                             // commit the slab requested at the top of the kernel
#pragma unroll
    for (int u = 0; u < L3_WPT; ++u)
      if (wcp_active && wcp_run0 + u * wcp_rpi < wcp_runs) Wsp[(size_t)(wcp_run0 + u * wcp_rpi) * Rp + wcp_rr] = wcp[u];
  }
  if constexpr (!PRE) {
    __syncthreads();
    for (int idx = tid; idx < R * kch; idx += L3_THREADS) {
      const int kc = idx / R, r = idx - kc * R;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(Wsrc + (size_t)r * K + kc * 8);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(Wsrc + (size_t)r * K + kc * 8 + 4);
      atomicMax(&wmax_lds[r], l3_absmax8(x0, x1));
    }
    __syncthreads();
    for (int idx = tid; idx < R * kch; idx += L3_THREADS) {
      const int kc = idx / R, r = idx - kc * R;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(Wsrc + (size_t)r * K + kc * 8);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(Wsrc + (size_t)r * K + kc * 8 + 4);
      float s, inv;
      l3_scale(wmax_lds[r], 14, s, inv);
      f16x8 h, m;
      l3_split8(x0, x1, s, h, m);
      u32x4* dst = Wsp + (size_t)(kc * 2) * Rp + r;                // kc = ks * 4 + k-group
      dst[0] = __builtin_bit_cast(u32x4, h);
      dst[Rp] = __builtin_bit_cast(u32x4, m);
      if (kc == 0) winv_lds[r] = inv;
    }
  }
  __syncthreads();   // the only barrier of the main part
  UNIVS_GT(g_l3_trace, gts, 2);
  UNIVS_GT_VAL(g_l3_trace, gts, 63, ntiles);
  if (ntiles == 0) return;

  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (int)((long long)M * N * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rrs =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(epi == L3_EPI_RESIDUAL ? Res : X), 0, (int)((long long)M * N * 4), 0x00020000);

  f32x4 acc[RB][2];
  // Row scales of x WITHOUT a pass of their own: a running scale per row.  Before a k-step is split its row maximum (over
  // the four lanes that hold the row's k-groups) is compared with the exponent the current scale was set for; a row whose
  // new maximum would leave fp16's range gets a smaller power of two, and the accumulators of that row -- sums of
  // products under the old scale -- are multiplied by the ratio (exact).  The scale is set for 2^12 (three binades of room
  // for later k-steps), only ever shrinks, and typically changes once or twice per row.
  int eset[2];                                                    // exponent the scale of my two rows was set for
  float sx[2], sx_inv[2];

  auto epilogue = [&](int tile) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int m = (wt0 + tile * NWV) * L3_TILE_M + 16 * c + j;
      unsigned rowpart = 0;
      if (epi == L3_EPI_BLOCKED) {
        const unsigned b = (unsigned)m / (unsigned)blk_rows;
        rowpart = b * (unsigned)blk_rows * (unsigned)N + ((unsigned)m - b * (unsigned)blk_rows) * (unsigned)blk_cols;
      }
      auto out_off = [&](int rb) __attribute__((always_inline)) -> unsigned {
        const int f = rb * 16 + 4 * g;
        unsigned off = ((unsigned)m * (unsigned)N + (unsigned)(n0 + f)) * 4u;
        if (epi == L3_EPI_BLOCKED) {
          const unsigned fg = (unsigned)(n0 + f);
          const unsigned cb = fg / (unsigned)blk_cols;
          off = (rowpart + cb * (unsigned)blk_rows * (unsigned)blk_cols + (fg - cb * (unsigned)blk_cols)) * 4u;
        }
        return (m < M && f < R) ? off : 0xFFFFFFF0u;             // out of range: loads return 0, stores are dropped
      };
      // the residual values of this column tile, requested together: a load inside the per-block loop is compiled into load / wait /
      // store -- one memory latency per block, RB per column tile (enc_output_proj + residual: 91 us against 62 us without)
      constexpr int RBAT = RB > 4 ? (RB + 1) / 2 : RB;           // (in two halves at RB > 4: registers)
      f32x4 resv[RBAT];
      auto load_res = [&](int rb0) __attribute__((always_inline)) {
        if (epi == L3_EPI_RESIDUAL) {
#pragma unroll
          for (int q = 0; q < RBAT; ++q)
            resv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, out_off(min(rb0 + q, RB - 1)), 0, 0));
        } else {
#pragma unroll
          for (int q = 0; q < RBAT; ++q) resv[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      };
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        if (rb % RBAT == 0) load_res(rb);
        const int f = rb * 16 + 4 * g;
        const int fc = min(f, R - 4);
        const f32x4 wi = *reinterpret_cast<const f32x4*>(winv_lds + fc);
        const f32x4 bi = *reinterpret_cast<const f32x4*>(bias_lds + fc);
        f32x4 v = (acc[rb][c] * sx_inv[c]) * wi + bi;              // two exact unscalings, then the bias
        const unsigned offc = out_off(rb);
        if (epi == L3_EPI_RELU) v = __builtin_elementwise_maximum(v, (f32x4){0.f, 0.f, 0.f, 0.f})   /* NaN-propagating, as torch.relu */;
        if (epi == L3_EPI_GELU) {   // x * 0.5 * (1 + erf(x / sqrt 2)): nn.GELU() (approximate = 'none')
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = l3_gelu(v[e]);
        }
        if (epi == L3_EPI_RESIDUAL) v += resv[rb % RBAT];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, offc, 0, 0);
      }
    }
  };

  // A fragments (W parts) come from LDS in NB batches of BSZ feature blocks, software-pipelined by hand: the reads of batch
  // b + 1 are issued BEFORE the 6 BSZ MFMAs of batch b and waited for after them.  Left to the compiler, every ds_read_b128
  // sat directly in front of its first use behind an `s_waitcnt lgkmcnt(0)` (ISA of the first version): ~100 clocks of
  // exposed LDS latency per 3 MFMAs, and halving the MFMA work against the six-product kernel changed nothing.
  constexpr int NB = RB > 6 ? 4 : 2;                             // even: the two buffers alternate across k-steps too
  constexpr int BSZ = (RB + NB - 1) / NB;                        // 1 .. 3 blocks
  u32x4 afr[2][BSZ][2];                                          // [buffer][block of the batch][part h, m]
  // byte address of (k-step ks, my k-group g, part h, row j); blocks are 256 B apart, the m part Rp * 16 B further
  const unsigned a_lane = (unsigned)((g * 2 * Rp + j) * 16);
  const unsigned a_kstep = (unsigned)(8 * Rp * 16);
  const unsigned a_part = (unsigned)(Rp * 16);
  // the block (256 B) and part offsets of a read are IMMEDIATES of the instruction (batch index = a compile-time constant): one address
  // register per k-step instead of two vector adds per read (16 of ~100 vector instructions per k-step at 128 features)
  auto read_batch = [&](u32x4 (&d)[BSZ][2], int ks, auto bc) __attribute__((always_inline)) {
    constexpr int b = decltype(bc)::value;
    const unsigned a0 = a_lane + (unsigned)ks * a_kstep;
    l3_static_for<0, BSZ>([&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;
      constexpr int rb = b * BSZ + q < RB ? b * BSZ + q : RB - 1;   // a short last batch re-reads the last block
      u32x4& dh = d[q][0];                                       // (operands of an asm statement do not capture: name them first)
      u32x4& dm = d[q][1];
      const unsigned aa = a0;
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                   : "=&v"(dh), "=&v"(dm)
                   : "v"(aa), "n"(rb * 256), "n"(rb * 256 + Rp * 16)
                   : "memory");
    });
  };
  // all LDS traffic in flight is the batch about to be used; the operands tie the MFMAs below to this wait
  auto wait_batch = [&](u32x4 (&d)[BSZ][2]) __attribute__((always_inline)) {
    if constexpr (BSZ == 1)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0][0]), "+v"(d[0][1]) : : "memory");
    else if constexpr (BSZ == 2)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]) : : "memory");
    else
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(d[0][0]), "+v"(d[0][1]), "+v"(d[1][0]), "+v"(d[1][1]), "+v"(d[2][0]), "+v"(d[2][1]) : : "memory");
  };
  auto group = [&](int ks0, bool last_group) __attribute__((always_inline)) {
    unsigned vo_ref[2];                                           // rows of the k-steps requested in this group
#pragma unroll
    for (int c = 0; c < 2; ++c) vo_ref[c] = last_group ? vo_next[c] : vo_cur[c];
#pragma unroll
    for (int u = 0; u < RING; ++u) {
      const int ks = ks0 + u;
      // ---- the running row scale
      bool need = false;
      int enew[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const unsigned mk = l3_row_max(l3_absmax8(raw[u][c][0], raw[u][c][1]));
        enew[c] = max(-100, min((int)((mk >> 23) & 255u) - 127, 128));      // 2^e <= max < 2^(e+1)
        need = need || (enew[c] > eset[c] + 2);
      }
      if (__builtin_amdgcn_ballot_w64(need) != 0 && ablate != 1) {           // rare
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bool mine = enew[c] > eset[c] + 2;
          const int en = mine ? enew[c] : eset[c];
          const float ratio = __builtin_bit_cast(float, (unsigned)(127 + max(eset[c] - en, -126)) << 23);   // 2^(old - new) <= 1
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc[rb][c] *= ratio;
          eset[c] = en;
          sx[c] = __builtin_bit_cast(float, (unsigned)(127 + 12 - en) << 23);
          sx_inv[c] = __builtin_bit_cast(float, (unsigned)(127 - 12 + en) << 23);
        }
      }
      float s0 = sx[0], s1 = sx[1];
      asm volatile("" : "+v"(s0), "+v"(s1) : : "memory");      // pins the consumption of ring stage u here
      f16x8 bh[2], bm[2];
      l3_split8(raw[u][0][0], raw[u][0][1], s0, bh[0], bm[0]);
      l3_split8(raw[u][1][0], raw[u][1][1], s1, bh[1], bm[1]);
      __builtin_amdgcn_sched_barrier(0);
      {                                                           // refill the stage just consumed (RING steps ahead)
        const int ksn = last_group ? u : ks + RING;
        load_x(raw[u], vo_ref, ksn);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int ks_next = (ks + 1 == KS) ? 0 : ks + 1;           // the next k-step's first batch (W is the same for every tile)
      l3_static_for<0, NB>([&](auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        wait_batch(afr[b & 1]);
        if constexpr (b + 1 < NB) read_batch(afr[(b + 1) & 1], ks, std::integral_constant<int, b + 1>{});
        else read_batch(afr[0], ks_next, std::integral_constant<int, 0>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < BSZ; ++q) {
          const int rb = b * BSZ + q;
          if (rb < RB) {
            const f16x8 ah = __builtin_bit_cast(f16x8, afr[b & 1][q][0]);
            const f16x8 am = __builtin_bit_cast(f16x8, afr[b & 1][q][1]);
            // smallest terms first: m*h', h*m', h*h'
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
  };

  read_batch(afr[0], 0, std::integral_constant<int, 0>{});       // batch 0 of the first k-step

#pragma unroll 1
  for (int tile = 0; tile < ntiles; ++tile) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    eset[0] = eset[1] = ablate == 1 ? 12 : -1000;                // the first k-step sets the scale
    sx[0] = sx[1] = sx_inv[0] = sx_inv[1] = 1.0f;
#pragma unroll 1
    for (int ks0 = 0; ks0 < KS; ks0 += RING) group(ks0, ks0 + RING >= KS);
    UNIVS_GT(g_l3_trace, gts, 3 + 2 * tile);
    epilogue(tile);
    UNIVS_GT(g_l3_trace, gts, 4 + 2 * tile);
    vo_cur[0] = vo_next[0];
    vo_cur[1] = vo_next[1];
    tile_voff(min(tile + 2, ntiles - 1), vo_next);
  }
  UNIVS_GT_REAL(g_l3_trace, gts, 61);
}

// returns 1 if launched, 0 if the shape is not covered, < 0 on error.  Same contract as linear_split_f32 (W-stationary part).
// `winv` != nullptr: `w` is the pre-split image (univs_presplit_weights_f32, mode 0) and `winv` its inverse row scales
int linear_f16x3_f32(const float* x, const float* w, const float* bias, const float* residual, float* y, long long M, int N,
                     int K, int epi, hipStream_t st, int blk_rows, int blk_cols, const float* winv) {
  if (M <= 0 || N <= 0) return 1;
  if (epi < 0 || epi > L3_EPI_BLOCKED || (epi == L3_EPI_RESIDUAL) != (residual != nullptr)) return 0;
  if (epi == L3_EPI_BLOCKED && (blk_rows < 1 || blk_cols < 4 || blk_cols % 4 != 0 || N % blk_cols != 0 || M % blk_rows != 0))
    return 0;
  const int ring = K % 128 == 0 ? 4 : K % 96 == 0 ? 3 : 0;   // register stages of x: a divisor of the k-steps
  if (K < 96 || ring == 0 || N % 4 != 0) return 0;
  if (M * (long long)N * 4 >= 0x7FFFFFFFLL || M * (long long)K * 4 >= 0x7FFFFFFFLL) return 0;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15) || (reinterpret_cast<uintptr_t>(y) & 15) ||
      (reinterpret_cast<uintptr_t>(residual) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(winv) & 3))
    return 0;
  const long long lds_cap = 160 * 1024 - 2048;   // W slab + bias + inverse scales + zero tail + row maxima
  int r_cap = (int)std::min<long long>(lds_cap / ((long long)K * 4 + 12), 16 * L3_MAX_RB);
  r_cap -= r_cap % 16;                                     // the LDS image holds whole 16-feature blocks
  const UnivsConfig cfg_ = config();
  if (cfg_.linear_rows_per_pass >= 16) r_cap = std::min(r_cap, cfg_.linear_rows_per_pass - cfg_.linear_rows_per_pass % 16);
  if (r_cap < 16) return 0;
  const int passes = (N + r_cap - 1) / r_cap;
  int rows = (N + passes - 1) / passes;
  rows = (rows + 3) & ~3;
  const int RB = (rows + 15) / 16;
  const long long WT = (M + L3_TILE_M - 1) / L3_TILE_M;
  if (WT < 64) return 0;                                   // too few rows to amortise the staging of W
  if (winv && ((K >> 3) * 2 + (L3_THREADS / rows) - 1) / (L3_THREADS / rows) > L3_WPT) return 0;   // (cannot happen for K <= 768: <= 20 units)
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
  gx = std::min(gx, std::max<long long>(1, WT / (2 * (L3_THREADS / 64))));
  if (gx >= 8 && (gx - gx % 8) * 10 >= gx * 9) gx -= gx % 8;
  if (cfg_.linear_grid_x > 0) gx = std::min<long long>(cfg_.linear_grid_x, WT);
  const size_t lds = (size_t)K * (16 * RB) * 4 + 12 * (size_t)rows + 256 + 16;
  dim3 grid((unsigned)gx, (unsigned)passes), block(L3_THREADS);
#define UNIVS_L3(rb, rg, pre)                                                                                             \
  do {                                                                                                               \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linear_f16x3<rb, rg, pre>),                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
    hipLaunchKernelGGL((linear_f16x3<rb, rg, pre>), grid, block, lds, st, x, w, bias, residual, y, (int)M, N, K, rows, epi, \
                       blk_rows, blk_cols, config().linear_ablate, winv);                                                                          \
  } while (0)
#define UNIVS_L3_RB(rb)                                    \
  case rb:                                                 \
    if (winv) {                                            \
      if (ring == 4) { UNIVS_L3(rb, 4, true); }            \
      else { UNIVS_L3(rb, 3, true); }                      \
    } else {                                               \
      if (ring == 4) { UNIVS_L3(rb, 4, false); }           \
      else { UNIVS_L3(rb, 3, false); }                     \
    }                                                      \
    break
  switch (RB) {
    UNIVS_L3_RB(1);
    UNIVS_L3_RB(2);
    UNIVS_L3_RB(3);
    UNIVS_L3_RB(4);
    UNIVS_L3_RB(5);
    UNIVS_L3_RB(6);
    UNIVS_L3_RB(7);
    default: UNIVS_L3_RB(8);
  }
#undef UNIVS_L3_RB
#undef UNIVS_L3
  const int rc = check_launch("linear_f16x3_f32");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs

#ifdef UNIVS_TRACE_GEMM
extern "C" int univs_debug_gemm_trace_linear(unsigned long long* out, int clear) {
  if (clear) {
    static unsigned long long zeros[UNIVS_GT_SLOTS * UNIVS_GT_STAMPS] = {};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(univs::g_l3_trace), zeros, sizeof(zeros));
  }
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(univs::g_l3_trace), sizeof(unsigned long long) * UNIVS_GT_SLOTS * UNIVS_GT_STAMPS);
}
#endif
