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

int linear_f16x3_f32(const float* x, const float* w, const float* bias, const float* residual, float* y, long long M, int N,
                     int K, int epi, hipStream_t st, int blk_rows, int blk_cols, const float* winv);   // linear_f16x3.hip

// returns 1 if launched, 0 if the shape is not covered (the caller uses the library GEMM), < 0 on error
// `winv` != nullptr: `w` is the pre-split image of univs_presplit_weights_f32 (three-product kernel only)
int linear_split_f32(const float* x, const float* w, const float* bias, const float* residual, float* y, long long M, int N,
                     int K, int epi, hipStream_t st, int blk_rows, int blk_cols, const float* winv) {
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
  // K >= 768: the weights are split once per tensor and streamed (gemm_f16x3_stream.hip: univs_linear_presplit_f32); this
  // entry splits W in every workgroup and only pays while the whole K of a useful number of features fits LDS
  if (K > 768) return 0;
  if (config().linear_terms != 6 || winv)
    return linear_f16x3_f32(x, w, bias, residual, y, M, N, K, epi, st, blk_rows, blk_cols, winv);   // default: three products
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
