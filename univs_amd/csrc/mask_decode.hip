// Mask decode (mask-embed x pixel-feature contraction) for gfx950.  Three kernels, newest last in this file:
//   skinny_gemm_f32          v_mfma_f32_32x32x2_f32, bit-identical to a k-ordered fp32 fmaf chain (MFMA-f32-bound);
//   skinny_gemm_bf16x6       fp32 emulated on the bf16 matrix cores from an exact 3-way split, 64-column tiles;
//   skinny_gemm_bf16x6_n32   the same arithmetic with 32-column tiles and a four-stage register ring (the default
//                            for large feature maps: HBM-bound, 2.3x faster than the f32 kernel at config 2).
// The header below describes the f32 kernel; the split kernels carry their own.
//
// Reference semantics: torch.einsum("btqc,btchw->btqhw", mask_embed, mask_features).transpose(1,2)
// (univs/modeling/transformer_decoder/video_mask2former_transformer_decoder_univs.py:527-528) and,
// for the fused variant, the attention-mask rule of :555-566 plus the all-masked-row reset of :390.
//
// Shape of the problem per frame: C[Q, N] = A[Q, K] * B[K, N], Q ~ 100..350 ("skinny"), K = 256,
// N = H*W up to 130k.  B (the feature map) is streamed from HBM exactly once and never re-used by
// another wave, so it goes global -> VGPR directly (each MFMA B fragment is two 128-byte row
// segments).  A is tiny and re-used by every column tile, so it is transposed once into LDS
// ([K][Q] so that an A fragment is a conflict-free ds_read_b32) and the block then walks several
// column tiles.  v_mfma_f32_32x32x2_f32 is an exact fp32 fmaf chain (k-ordered), so results equal a
// scalar fp32 loop bit-for-bit -- what the 1e-3 / argmax-identical parity contract needs.
#include "common.h"
#include "config.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace univs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int MD_THREADS = 512;   // 8 waves = 2 per SIMD: one wave's HBM latency hides under the other's MFMAs
constexpr int MD_WAVE_N = 32;     // columns per wave per tile
constexpr int MD_BLOCK_N = 256;   // 8 waves x 32 columns

// Epilogue 1: write logits to out[(q*T + t)*N + n]
struct StoreLogits {
  float* out;
  int T;
  __device__ __forceinline__ void operator()(int t, int q, long long n, long long N, float v) const {
    out[((long long)q * T + t) * N + n] = v;
  }
  // four consecutive columns n .. n+3 (n % 4 == 0, N % 4 == 0, 16-B aligned base)
  __device__ __forceinline__ void store4(int t, int q, long long n, long long N, f32x4 v) const {
    *reinterpret_cast<f32x4*>(out + ((long long)q * T + t) * N + n) = v;
  }
};

// Epilogue 2: attention mask byte (1 = masked out <=> logit < 0 <=> sigmoid < 0.5) and a per-row
// "has a visible key" flag used by the all-masked-row reset.
struct StoreAttnMask {
  uint8_t* mask;       // [T, Q, N]
  unsigned* row_any;   // [T, Q]: `gen` is stored for a row with a visible key (the eager form zeroes the flags first and uses 1)
  int Q;
  unsigned gen;
  __device__ __forceinline__ void operator()(int t, int q, long long n, long long N, float v) const {
    const bool masked = v < 0.f;
    mask[((long long)t * Q + q) * N + n] = masked ? 1 : 0;
    if (!masked) row_any[t * Q + q] = gen;  // benign race: every writer stores the same value
  }
  __device__ __forceinline__ void store4(int t, int q, long long n, long long N, f32x4 v) const {
    const unsigned m = (v.x < 0.f ? 1u : 0u) | (v.y < 0.f ? 0x100u : 0u) | (v.z < 0.f ? 0x10000u : 0u) |
                       (v.w < 0.f ? 0x1000000u : 0u);
    *reinterpret_cast<unsigned*>(mask + ((long long)t * Q + q) * N + n) = m;
    if (m != 0x01010101u) row_any[t * Q + q] = gen;
  }
};

// MI = number of 32-row blocks of A handled by each wave (rows per block-tile = 32*MI).
// B fragments are fetched with raw buffer loads: the per-lane byte offset (column, k parity) is
// computed once, the k-row offset travels in an SGPR, out-of-range columns are clamped (their results
// are never stored) -- no per-load VALU address arithmetic and no divergent control flow, so the 16
// loads of a chunk are in flight together and the NEXT chunk is fetched while the current one feeds
// the MFMAs (register double buffer).
template <int MI, typename Epilogue>
__global__ __launch_bounds__(MD_THREADS, 1) void skinny_gemm_f32(const float* __restrict__ A,  // [T,Q,K]
                                                                  const float* __restrict__ B,  // [T,K,N]
                                                                  int Q, int K, long long N,
                                                                  int tiles_per_block, Epilogue ep) {
  extern __shared__ __attribute__((aligned(16))) float At[];  // [K][LDP]
  constexpr int QP = 32 * MI;
  constexpr int LDP = QP + 1;
  constexpr int UNR = 16;  // k-steps (of 2) per chunk
  const int t = blockIdx.z;
  const int q0 = blockIdx.y * QP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- stage A^T (rows q0..q0+QP) into LDS; coalesced along k, bank = (k + q) % 32 on the write
  const float* At_src = A + ((long long)t * Q) * K;
  for (int idx = tid; idx < QP * K; idx += MD_THREADS) {
    const int k = idx % K, q = idx / K;
    At[k * LDP + q] = (q0 + q < Q) ? At_src[(long long)(q0 + q) * K + k] : 0.f;
  }
  __syncthreads();

  const int khalf = lane >> 5;   // which of the two k's of an MFMA this lane feeds
  const int l31 = lane & 31;
  const int Ni = (int)N;
  // buffer resource over this frame's B matrix (K*N floats); wave-uniform by construction
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(B + (long long)t * K * N), 0, (int)((long long)K * N * 4), 0x00020000);
  const int Kc = (K + 2 * UNR - 1) / (2 * UNR);  // chunks; rows >= K read as 0 (buffer bounds check)

  for (int tile = 0; tile < tiles_per_block; ++tile) {
    const long long col0 = ((long long)blockIdx.x * tiles_per_block + tile) * MD_BLOCK_N + wave * MD_WAVE_N;
    if (col0 >= N) break;                    // wave-uniform
    const int col = (int)col0 + l31;
    const bool cv = col < Ni;
    const int voff = (min(col, Ni - 1) + khalf * Ni) * 4;  // bytes

    f32x16 acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float bcur[UNR], bnext[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u)
      bcur[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (2 * u) * Ni * 4, 0));

    for (int c = 0; c < Kc; ++c) {
      const int k0 = c * 2 * UNR;
      if (c + 1 < Kc) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
          bnext[u] = __builtin_bit_cast(
              float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (k0 + 2 * UNR + 2 * u) * Ni * 4, 0));
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = k0 + 2 * u + khalf;
        const float* arow = At + min(k, K - 1) * LDP + l31;
        const float bsel = (k < K) ? bcur[u] : 0.f;  // K not a multiple of 2: the odd tail row
#pragma unroll
        for (int i = 0; i < MI; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[32 * i], bsel, acc[i], 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) bcur[u] = bnext[u];
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (cv) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          if (q < Q) ep(t, q, col, N, acc[i][r]);
        }
    }
  }
}


// The same arithmetic for SMALL maps (the attention masks of the coarse levels: 23x40 and 46x80 pixels, K = 256, 100 rows:
// 5-19 MB per launch), where the chunked kernel above is a chain of latencies -- stage A, barrier, then eight times
// "request 16 rows, run 16 MFMAs": 29 / 44 us for what the memory system delivers in a few.  ONE SHOT: a wave requests all
// 128 k-row pairs of its 32 columns at once (128 registers), A is requested in front of them and committed to LDS while
// they fly, and the 128 MFMAs then run back to back as the rows arrive (the hardware returns loads in order; every MFMA
// waits for exactly its own row).  Same k order, same fmaf chain: bit-identical to skinny_gemm_f32<1>.
template <typename Epilogue>
__global__ __launch_bounds__(MD_THREADS, 1) void skinny_gemm_f32_oneshot(const float* __restrict__ A,  // [T,Q,256]
                                                                          const float* __restrict__ B,  // [T,256,N]
                                                                          int Q, int N, int tiles_per_block, Epilogue ep) {
  constexpr int K = 256, QP = 32, LDP = QP + 1, NL = K / 2;
  extern __shared__ __attribute__((aligned(16))) float At[];  // [K][LDP]
  const int t = blockIdx.z;
  const int q0 = blockIdx.y * QP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int khalf = lane >> 5, l31 = lane & 31;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(B + (long long)t * K * N), 0, (int)((long long)K * N * 4), 0x00020000);

  // ---- A^T of rows q0 .. q0+31: requested first (16 loads per thread, coalesced along k) ...
  constexpr int NA = QP * K / MD_THREADS;
  float areg[NA];
  {
    const float* At_src = A + ((long long)t * Q) * K;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int idx = tid + i * MD_THREADS, k = idx & (K - 1), q = idx >> 8;
      areg[i] = At_src[(long long)min(q0 + q, Q - 1) * K + k];     // rows past Q: a copy of the last row, never stored
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  float b[NL];
  auto request = [&](int tile) __attribute__((always_inline)) {
    const int col = min((blockIdx.x * tiles_per_block + tile) * MD_BLOCK_N + wave * MD_WAVE_N + l31, N - 1);
    const int voff = (col + khalf * N) * 4;
#pragma unroll
    for (int u = 0; u < NL; ++u) b[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (2 * u) * N * 4, 0));
  };
  // ... then the first tile's rows of B, then A goes to LDS (waits for the A loads only: they are the oldest)
  request(0);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + i * MD_THREADS, k = idx & (K - 1), q = idx >> 8;
    At[k * LDP + q] = areg[i];
  }
  __syncthreads();

  const float* arow = At + khalf * LDP + l31;
  for (int tile = 0; tile < tiles_per_block; ++tile) {
    const int col0 = (blockIdx.x * tiles_per_block + tile) * MD_BLOCK_N + wave * MD_WAVE_N;
    if (col0 >= N) break;                    // wave-uniform
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < NL; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * u * LDP], b[u], acc, 0, 0, 0);
    const int col = col0 + l31;
    if (tile + 1 < tiles_per_block) request(tile + 1);   // (uniform) the next tile's rows fly under this tile's stores
    if (col < N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int q = q0 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        if (q < Q) ep(t, q, col, N, acc[r]);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Second generation: fp32 emulated on the bf16 matrix cores ("bf16 x 6").
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (1/16 of the bf16 MFMA rate), which makes the contraction
// MFMA-bound at 42 % of the 157 TF/s f32 peak although its arithmetic intensity (36 flop/B) is far below the bf16
// ridge.  Here every fp32 operand is split EXACTLY into three bf16 parts by truncation,
//     x = h + m + l,   h = top 8 significant bits, m = the next 8, l = the last 8 (24 = 8 + 8 + 8),
// and the product is accumulated in fp32 from the six bf16 x bf16 terms of weight >= 2^-16 (each such product is
// exact in fp32): l*h + h*l + m*m + m*h + h*m + h*h.  The three dropped terms (m*l, l*m, l*l) are <= 3 * 2^-24
// relative per product -- below the rounding noise of an fp32 sum over K = 256 (measured: max error 7e-6 vs 1.1e-5
// for an fp32 GEMM on N(0, 0.25) inputs, logits up to 18).  Six bf16 MFMAs cost 6/16 of one f32 MFMA, so the
// kernel becomes HBM-bound -- what SURVEY 8d prices this contraction against.
//
// Shape: a workgroup = 8 waves (two per SIMD, 256 registers each); A^T is split once per workgroup into LDS
// ([k-step][part][k-group][row] x 16 B = the MFMA A fragments, 6 B per element: at most 106 rows of K = 256 -- more
// rows run as several row passes).  Each wave streams 64-column tiles of B: column 4j + c of the tile is column j
// of MFMA tile c, so the global loads (8 k-rows x 16 B per lane and k-step) AND the stores (4 consecutive columns
// per lane) are 16-byte vectors; B is split in registers (7 VALU per element, hidden behind 168 MFMAs per k-step).
// Any k-permutation inside a 32-deep step is harmless because A and B fragments are built with the same mapping.
constexpr int SB_THREADS = 512;   // 8 waves: two per SIMD
constexpr int SB_WAVE_N = 64;
constexpr int SB_MAX_RB = 7;

__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l, unsigned hi_mask = 0xFFFF0000u) {
  const unsigned hb = __float_as_uint(x) & hi_mask;
  const float r = x - __uint_as_float(hb);              // exact
  const unsigned mb = __float_as_uint(r) & hi_mask;
  const float r2 = r - __uint_as_float(mb);             // exact, <= 8 significant bits: a bf16 value
  h = hb; m = mb; l = __float_as_uint(r2);
}
// two fp32 bit patterns whose low halves are zero -> their bf16 pair (element 0 in the low half)
__device__ __forceinline__ unsigned pack_bf16(unsigned lo, unsigned hi) { return (lo >> 16) | hi; }

template <int RB, typename Epilogue>
__global__ __launch_bounds__(SB_THREADS, 1) void skinny_gemm_bf16x6(const float* __restrict__ A,   // [T,Q,K]
                                                                      const float* __restrict__ B,   // [T,K,N]
                                                                      int Q, int K, int N, int rows_per_pass,
                                                                      Epilogue ep) {
  extern __shared__ __attribute__((aligned(16))) u32x4 Asp[];   // [K/32][4 k-groups][R rows][3 parts]
  const int t = blockIdx.z;
  const int q0 = blockIdx.y * rows_per_pass;
  const int R = min(rows_per_pass, Q - q0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // tell the compiler it is wave-uniform (scalar loop, SGPR offsets)
  const int KS = K >> 5;

  // ---- split this pass's rows of A into LDS, in fragment order
  {
    const float* Asrc = A + ((long long)t * Q + q0) * K;
    const int kch = K >> 3;
    for (int idx = tid; idx < R * kch; idx += SB_THREADS) {
      const int r = idx / kch, kc = idx - r * kch;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(Asrc + (long long)r * K + kc * 8);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(Asrc + (long long)r * K + kc * 8 + 4);
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        split3(x0[e], h[e], m[e], l[e]);
        split3(x1[e], h[4 + e], m[4 + e], l[4 + e]);
      }
      u32x4* dst = Asp + (((kc >> 2) * 4 + (kc & 3)) * R + r) * 3;      // [k-step][k-group][row][part]
      dst[0] = (u32x4){pack_bf16(h[0], h[1]), pack_bf16(h[2], h[3]), pack_bf16(h[4], h[5]), pack_bf16(h[6], h[7])};
      dst[1] = (u32x4){pack_bf16(m[0], m[1]), pack_bf16(m[2], m[3]), pack_bf16(m[4], m[5]), pack_bf16(m[6], m[7])};
      dst[2] = (u32x4){pack_bf16(l[0], l[1]), pack_bf16(l[2], l[3]), pack_bf16(l[4], l[5]), pack_bf16(l[6], l[7])};
    }
  }
  __syncthreads();   // the only barrier

  // ---- this wave's contiguous run of 64-column tiles
  const int WT = (N + SB_WAVE_N - 1) / SB_WAVE_N;
  const int nw = gridDim.x * (SB_THREADS / 64), widx = blockIdx.x * (SB_THREADS / 64) + wave;
  const int wt0 = (int)((long long)WT * widx / nw), wt1 = (int)((long long)WT * (widx + 1) / nw);
  if (wt0 >= wt1) return;
  const int nsteps = (wt1 - wt0) * KS;

  const int j = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(B + (long long)t * K * N), 0, (int)((long long)K * N * 4), 0x00020000);
  const unsigned lane_off = (unsigned)((8 * g) * N + 4 * j) * 4u;

  f32x4 braw[8];
  auto load_b = [&](int step) __attribute__((always_inline)) {
    const int tile = step / KS, ks = step - tile * KS;
    const unsigned voff = lane_off + (unsigned)((wt0 + tile) * SB_WAVE_N) * 4u;
    const unsigned soff = (unsigned)(ks * 32) * (unsigned)N * 4u;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      braw[kk] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff + (unsigned)kk * (unsigned)N * 4u, 0));
  };

  f32x4 acc[RB][4];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[rb][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // One step = 32 k of one 64-column tile: wait for the staged rows, split them into the B fragments, re-issue the
  // loads of the NEXT step into the same registers, then 24 * RB MFMAs (which hide that latency).  Two waves per
  // SIMD: one wave's split / LDS reads / load waits run under the other's MFMAs.
  load_b(0);
#pragma unroll 1
  for (int step = 0; step < nsteps; ++step) {
    const int tile = step / KS, ks = step - tile * KS;
    // B fragments of the 4 column tiles: element e of tile c = k-row e of this lane's k-group, column 4j + c
    bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) split3(braw[kk][c], h[kk], m[kk], l[kk]);
      bh[c] = __builtin_bit_cast(bf16x8, (u32x4){pack_bf16(h[0], h[1]), pack_bf16(h[2], h[3]), pack_bf16(h[4], h[5]), pack_bf16(h[6], h[7])});
      bm[c] = __builtin_bit_cast(bf16x8, (u32x4){pack_bf16(m[0], m[1]), pack_bf16(m[2], m[3]), pack_bf16(m[4], m[5]), pack_bf16(m[6], m[7])});
      bl[c] = __builtin_bit_cast(bf16x8, (u32x4){pack_bf16(l[0], l[1]), pack_bf16(l[2], l[3]), pack_bf16(l[4], l[5]), pack_bf16(l[6], l[7])});
    }
    __builtin_amdgcn_sched_barrier(0);
    load_b(min(step + 1, nsteps - 1));      // (the load past the end re-reads the last step: one control-flow path)
    __builtin_amdgcn_sched_barrier(0);      // keep the prefetch HERE: hipcc otherwise sinks it below the MFMAs

    // two address registers in all: full row blocks are immediate offsets (768 B per block, 16 B per part) from the
    // lane's row, only the last block clamps (rows past R repeat the last row; never stored)
    const u32x4* ap = Asp + ((ks * 4 + g) * R + j) * 3;
    const u32x4* ap_last = Asp + ((ks * 4 + g) * R + min((RB - 1) * 16 + j, R - 1)) * 3;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const u32x4* a3 = (rb == RB - 1) ? ap_last : ap + rb * 48;
      const bf16x8 ah = __builtin_bit_cast(bf16x8, a3[0]);
      const bf16x8 am = __builtin_bit_cast(bf16x8, a3[1]);
      const bf16x8 al = __builtin_bit_cast(bf16x8, a3[2]);
      // smallest terms first; the four column tiles interleave so that no MFMA waits for its own accumulator
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[c], acc[rb][c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[c], acc[rb][c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[c], acc[rb][c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[c], acc[rb][c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[c], acc[rb][c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[c], acc[rb][c], 0, 0, 0);
    }
    if (ks == KS - 1) {
      // C/D layout of the 16x16 MFMA: column = lane & 15, row = 4 * (lane >> 4) + register
      const int col = (wt0 + tile) * SB_WAVE_N + 4 * j;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = rb * 16 + 4 * g + rr;
          if (row < R && col < N)
            ep.store4(t, q0 + row, col, N, (f32x4){acc[rb][0][rr], acc[rb][1][rr], acc[rb][2][rr], acc[rb][3][rr]});
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[rb][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  }
}


// ---- 32-column variant: half the tile, four k-steps of loads in flight -------------------------------------------
// Measured with the 64-column kernel above at config 2: 122 us = 3.4 TB/s with ONE k-step (8 KB per wave, 64 KB per
// CU) in flight -- latency-bound (~10 k cycles per step), and 920 tiles per frame over 408 waves quantise to 3 vs
// 2.25 tiles per wave.  Here a wave tile is 32 columns (column 2j + c of the tile = column j of MFMA tile c; 8-byte
// loads and stores), which halves accumulators and B fragments, so a ring of FOUR register stages fits: three k-steps
// (12 KB per wave, 96 KB per CU) are in flight while one is consumed, and 1840 tiles per frame balance to 4.5 -> 5.
constexpr int SB2_WAVE_N = 32;
constexpr int SB2_RING = 4;

// Epilogues of the 32-column kernel: unconditional buffer stores.  A lane with nothing to store (row past the pass,
// column past the end) moves its offset out of the buffer, where the hardware drops the write -- no exec-mask branches
// around the stores, so the compiler's vmcnt accounting sees exactly the stores the hardware counts.
struct Store2Logits {
  float* out;
  int T;
  unsigned bytes;
  struct Res { __amdgpu_buffer_rsrc_t o; };
  __device__ __forceinline__ Res resources() const {
    return {__builtin_amdgcn_make_buffer_rsrc(out, 0, (int)bytes, 0x00020000)};
  }
  __device__ __forceinline__ void store2(const Res& r, int t, int q, int n, int N, bool valid, float a, float b, bool) const {
    const unsigned off = (unsigned)(((unsigned)q * (unsigned)T + (unsigned)t) * (unsigned)N + (unsigned)n) * 4u;
    __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(a), __float_as_uint(b)}, r.o, valid ? off : 0xFFFFFFF0u, 0, 0);
  }
};
struct Store2AttnMask {
  uint8_t* mask;
  unsigned* row_any;
  int Q;
  unsigned mask_bytes, flag_bytes;
  unsigned gen;
  struct Res { __amdgpu_buffer_rsrc_t m, f; };
  __device__ __forceinline__ Res resources() const {
    return {__builtin_amdgcn_make_buffer_rsrc(mask, 0, (int)mask_bytes, 0x00020000),
            __builtin_amdgcn_make_buffer_rsrc(row_any, 0, (int)flag_bytes, 0x00020000)};
  }
  __device__ __forceinline__ void store2(const Res& r, int t, int q, int n, int N, bool valid, float a, float b, bool real) const {
    const unsigned short m = (unsigned short)((a < 0.f ? 1u : 0u) | (b < 0.f ? 0x100u : 0u));
    const unsigned row = (unsigned)t * (unsigned)Q + (unsigned)q;
    __builtin_amdgcn_raw_buffer_store_b16((short)m, r.m, valid ? row * (unsigned)N + (unsigned)n : 0xFFFFFFF0u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(gen, r.f, (valid && real && m != 0x0101u) ? row * 4u : 0xFFFFFFF0u, 0, 0);
  }
};

// ABL (profiling builds of one instantiation only, UNIVS_MASKDEC_ABLATE): 1 = no MFMAs (memory + split), 2 = no
// refills (split + MFMA on stale registers), 3 = neither.  Results are meaningless, the timing is the point.
template <int RB, int KSC, typename Epilogue, int ABL = 0>   // KSC = K / 32 when known at compile time (8), else 0
__global__ __launch_bounds__(SB_THREADS, 1) void skinny_gemm_bf16x6_n32(const float* __restrict__ A,   // [T,Q,K]
                                                                          const float* __restrict__ B,   // [T,K,N]
                                                                          int Q, int K, int N, int rows_per_pass,
                                                                          Epilogue ep) {
  extern __shared__ __attribute__((aligned(16))) u32x4 Asp[];   // [K/32][4 k-groups][R rows][3 parts]
  const int t = blockIdx.z;
  const int q0 = blockIdx.y * rows_per_pass;
  const int R = min(rows_per_pass, Q - q0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KS = K >> 5;                                         // a multiple of SB2_RING (host-checked)


  // The workgroup owns a contiguous run of 32-column tiles and its 8 waves take them round-robin: at any time the
  // waves read 8 adjacent 128-byte pieces of each k-row (1 KB contiguous per row and workgroup -- DRAM pages and
  // channels see long bursts), instead of 8 pieces five tiles apart.
  const int WT = (N + SB2_WAVE_N - 1) / SB2_WAVE_N;
  constexpr int NWV = SB_THREADS / 64;
  const int wg0 = (int)((long long)WT * blockIdx.x / gridDim.x), wg1 = (int)((long long)WT * (blockIdx.x + 1) / gridDim.x);
  const int wt0 = wg0 + wave;                                    // this wave's tiles: wt0, wt0 + 8, ...
  const int ntiles = wt0 < wg1 ? (wg1 - wt0 + NWV - 1) / NWV : 0;
  const int nsteps = max(ntiles, 1) * KS;                        // a multiple of SB2_RING (idle waves: one dummy tile)

  const int j = lane & 15, g = lane >> 4;
  const char* Bt = reinterpret_cast<const char*>(B + (long long)t * K * N);
  const unsigned rowb = (unsigned)N * 4u;

  f32x2 raw[SB2_RING][8];
  auto load_b = [&](f32x2 (&buf)[8], int step) __attribute__((always_inline)) {
    const int tile = step / KS, ks = step - tile * KS;
    // columns past the end (last tile only) re-read the last pair; their results are never stored
    const int col = min((wt0 + tile * NWV) * SB2_WAVE_N + 2 * j, N - 2);   // (idle waves read the last columns)
    const unsigned off = (unsigned)(ks * 32 + 8 * g) * rowb + (unsigned)col * 4u;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) buf[kk] = *reinterpret_cast<const f32x2*>(Bt + (off + (unsigned)kk * rowb));
  };

  f32x4 acc[RB][2];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) acc[rb][0] = acc[rb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};


  // (issue order = consumption order, pinned: the counted waits in the loop are the minimum over the entry path
  // and the back edge, so a reordered prologue would turn them into vmcnt(0))
#pragma unroll
  for (int u = 0; u < SB2_RING; ++u) {
    load_b(raw[u], min(u, nsteps - 1));
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- split this pass's rows of A into LDS (fragment order) while the first four k-steps of B are in flight
  {
    const float* Asrc = A + ((long long)t * Q + q0) * K;
    const int kch = K >> 3;
    for (int idx = tid; idx < R * kch; idx += SB_THREADS) {
      const int r = idx / kch, kc = idx - r * kch;
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(Asrc + (long long)r * K + kc * 8);
      const f32x4 x1 = *reinterpret_cast<const f32x4*>(Asrc + (long long)r * K + kc * 8 + 4);
      unsigned h[8], m[8], l[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        split3(x0[e], h[e], m[e], l[e]);
        split3(x1[e], h[4 + e], m[4 + e], l[4 + e]);
      }
      u32x4* dst = Asp + (((kc >> 2) * 4 + (kc & 3)) * R + r) * 3;      // [k-step][k-group][row][part]
      dst[0] = (u32x4){pack_bf16(h[0], h[1]), pack_bf16(h[2], h[3]), pack_bf16(h[4], h[5]), pack_bf16(h[6], h[7])};
      dst[1] = (u32x4){pack_bf16(m[0], m[1]), pack_bf16(m[2], m[3]), pack_bf16(m[4], m[5]), pack_bf16(m[6], m[7])};
      dst[2] = (u32x4){pack_bf16(l[0], l[1]), pack_bf16(l[2], l[3]), pack_bf16(l[4], l[5]), pack_bf16(l[6], l[7])};
    }
  }
  __syncthreads();
  if (ntiles == 0) return;


  // Vector-memory accounting (gfx9: loads AND stores count in vmcnt, in order).  Every wait below is emitted by the
  // compiler as "at most n younger operations in flight", n = the minimum over all paths reaching it.  With K / 32 = 8
  // known at compile time a tile is ONE straight-line block (28 stores of the previous tile + 8 steps), so every wait
  // leaves exactly the three younger ring stages (and the stores) in flight.  With a runtime k loop the wait after
  // an epilogue over-drains (it also waits for the newest loads) once per tile.
  const auto eres = ep.resources();
  auto epilogue = [&](int tile, bool real) __attribute__((always_inline)) {
    const int col = (wt0 + tile * NWV) * SB2_WAVE_N + 2 * j;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int row = rb * 16 + 4 * g + rr;
        ep.store2(eres, t, q0 + row, col, N, row < R && col < N, acc[rb][0][rr], acc[rb][1][rr], real);
      }
      acc[rb][0] = acc[rb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto group = [&](int tile, int ks0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < SB2_RING; ++u) {
      const int ks = ks0 + u;
      const int step = tile * KS + ks;
      // Pin the consumption of ring stage u HERE.  The splits are pure functions of registers loaded one loop
      // iteration earlier, and hipcc otherwise emits the splits of all four stages at the top of the loop body:
      // that waits for every load in flight and keeps four sets of fragments alive (spills).  An opaque copy of
      // the split mask, produced by a volatile asm at this point of the side-effect chain, ties them down.
      unsigned hi_mask = 0xFFFF0000u;
      asm volatile("" : "+s"(hi_mask) : : "memory");
      // B fragments of the 2 column tiles from ring stage u.  The split runs on the (column 0, column 1) pairs exactly
      // as the 8-byte loads delivered them (packed fp32 math on the loaded register pairs): any other pairing makes
      // hipcc shuffle the ring registers with copies at the loop latch, and those copies wait for every load.
      u32x2 H[8], M[8], Lo[8];
      const u32x2 mask2 = {hi_mask, hi_mask};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const f32x2 x = raw[u][kk];
        const u32x2 hb = __builtin_bit_cast(u32x2, x) & mask2;
        const f32x2 r = x - __builtin_bit_cast(f32x2, hb);               // exact
        const u32x2 mb = __builtin_bit_cast(u32x2, r) & mask2;
        const f32x2 r2 = r - __builtin_bit_cast(f32x2, mb);              // exact; a bf16 value
        H[kk] = hb; M[kk] = mb; Lo[kk] = __builtin_bit_cast(u32x2, r2);
      }
      bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        bh[c] = __builtin_bit_cast(bf16x8, (u32x4){pack_bf16(H[0][c], H[1][c]), pack_bf16(H[2][c], H[3][c]), pack_bf16(H[4][c], H[5][c]), pack_bf16(H[6][c], H[7][c])});
        bm[c] = __builtin_bit_cast(bf16x8, (u32x4){pack_bf16(M[0][c], M[1][c]), pack_bf16(M[2][c], M[3][c]), pack_bf16(M[4][c], M[5][c]), pack_bf16(M[6][c], M[7][c])});
        bl[c] = __builtin_bit_cast(bf16x8, (u32x4){pack_bf16(Lo[0][c], Lo[1][c]), pack_bf16(Lo[2][c], Lo[3][c]), pack_bf16(Lo[4][c], Lo[5][c]), pack_bf16(Lo[6][c], Lo[7][c])});
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!(ABL & 2)) load_b(raw[u], min(step + SB2_RING, nsteps - 1));   // refill the stage just consumed (4 steps ahead)
      __builtin_amdgcn_sched_barrier(0);

      const u32x4* ap = Asp + ((ks * 4 + g) * R + j) * 3;
      const u32x4* ap_last = Asp + ((ks * 4 + g) * R + min((RB - 1) * 16 + j, R - 1)) * 3;
      // One row block at a time, its three A fragments fetched from LDS while the previous block's 12 MFMAs issue
      // (compute-only ablation: 87 us against a 45 us MFMA-pipe bound -- the waves sat in lgkmcnt waits at the head of
      // every block group).  Two column tiles = two independent accumulators, re-used every second MFMA.
      auto afrag = [&](int rb, bf16x8 (&a)[3]) __attribute__((always_inline)) {
        const u32x4* p0 = (rb == RB - 1) ? ap_last : ap + rb * 48;
#pragma unroll
        for (int p3 = 0; p3 < 3; ++p3) a[p3] = __builtin_bit_cast(bf16x8, p0[p3]);
      };
      bf16x8 abuf[2][3];                                          // h, m, l of the current / next row block
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
          if constexpr (ABL & 1) {     // keep the operands alive without the matrix pipe
            acc[rb][0][0] += __builtin_bit_cast(f32x4, av)[0] + __builtin_bit_cast(f32x4, b0)[0];
            acc[rb][1][0] += __builtin_bit_cast(f32x4, b1)[1];
          } else {
            acc[rb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b0, acc[rb][0], 0, 0, 0);
            acc[rb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, b1, acc[rb][1], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if constexpr (KSC > 0) {
    // the epilogue of tile i - 1 opens the body of tile i (zeros to the first tile's own columns in the first
    // iteration, overwritten by its real epilogue later), so entry path and back edge issue identical sequences
#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
      epilogue(max(tile - 1, 0), tile > 0);
#pragma unroll
      for (int ks0 = 0; ks0 < KSC; ks0 += SB2_RING) group(tile, ks0);
    }
    epilogue(ntiles - 1, true);
  } else {
#pragma unroll 1
    for (int tile = 0; tile < ntiles; ++tile) {
#pragma unroll 1
      for (int ks0 = 0; ks0 < KS; ks0 += SB2_RING) group(tile, ks0);
      epilogue(tile, true);
    }
  }
}

static int maskdec_ablate() { return config().mask_decode_ablate & 3; }   // timing experiments (tools/prof_split.sh)
template <typename Epilogue>
static void launch_ablation(int abl, dim3 grid, dim3 block, size_t lds, hipStream_t st, const float* A, const float* B,
                            int Q, int K, int N, int rows, Epilogue ep) {
  if constexpr (std::is_same<Epilogue, Store2Logits>::value) {
#define UNIVS_ABL(a)                                                                                                   \
  case a:                                                                                                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_bf16x6_n32<7, 8, Epilogue, a>),              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
    hipLaunchKernelGGL((skinny_gemm_bf16x6_n32<7, 8, Epilogue, a>), grid, block, lds, st, A, B, Q, K, N, rows, ep);   \
    break
    switch (abl) {
      UNIVS_ABL(1);
      UNIVS_ABL(2);
      default: UNIVS_ABL(3);
    }
#undef UNIVS_ABL
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_bf16x6_n32<7, 8, Epilogue>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((skinny_gemm_bf16x6_n32<7, 8, Epilogue>), grid, block, lds, st, A, B, Q, K, N, rows, ep);
  }
}

// CT = 4: 64-column wave tiles (skinny_gemm_bf16x6), CT = 2: 32-column tiles with the four-stage ring (..._n32)
template <int CT, typename Epilogue>
static int launch_bf16x6(const float* A, const float* B, int T, int Q, int K, long long N, Epilogue ep,
                         hipStream_t st, const char* what) {
  // rows per pass: as many as the split A^T leaves room for in LDS, passes balanced
  const int r_cap = (int)std::min<long long>((160 * 1024) / ((long long)K * 6), 16 * SB_MAX_RB);
  const int passes = (Q + r_cap - 1) / r_cap;
  const int rows = (Q + passes - 1) / passes;
  const int RB = (rows + 15) / 16;
  const long long WT = (N + 16 * CT - 1) / (16 * CT);
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
  // one workgroup per CU over (frames x passes), as long as every wave has a tile (`wave_tiles` = 1).  Measured on the
  // attention-mask maps (T = 5, 100 rows; tools/kbench.py --only mask): 92 x 160 = 460 tiles: 51 workgroups 39.8 us, 29
  // workgroups (two tiles per wave) 44.4 us, and the round-2 rule (floor(460 / 16) = 28 workgroups of 16-17 tiles, i.e. a
  // THIRD tile for one wave of twelve of them) 56 us: the per-workgroup split of A costs less than an unbalanced tail.
  const int wave_tiles = config().mask_decode_wave_tiles > 0 ? config().mask_decode_wave_tiles : 1;
  long long gx = std::max<long long>(1, n_cu / std::max(1, T * passes));
  gx = std::min(gx, std::max<long long>(1, (WT + wave_tiles * (SB_THREADS / 64) - 1) / (wave_tiles * (SB_THREADS / 64))));
  const size_t lds = (size_t)K * rows * 6;
  dim3 grid((unsigned)gx, (unsigned)passes, (unsigned)T), block(SB_THREADS);
#define UNIVS_LAUNCH_RB(rb)                                                                                       \
  case rb:                                                                                                        \
    if constexpr (CT == 4) {                                                                                      \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_bf16x6<rb, Epilogue>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
      hipLaunchKernelGGL((skinny_gemm_bf16x6<rb, Epilogue>), grid, block, lds, st, A, B, Q, K, (int)N, rows, ep); \
    } else {                                                                                                      \
      if (K == 256 && rb == 7 && maskdec_ablate() > 0) {                                                          \
        launch_ablation<Epilogue>(maskdec_ablate(), grid, block, lds, st, A, B, Q, K, (int)N, rows, ep);          \
      } else if (K == 256) {                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_bf16x6_n32<rb, 8, Epilogue>),       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        hipLaunchKernelGGL((skinny_gemm_bf16x6_n32<rb, 8, Epilogue>), grid, block, lds, st, A, B, Q, K, (int)N, rows, ep); \
      } else {                                                                                                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_bf16x6_n32<rb, 0, Epilogue>),       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                         \
        hipLaunchKernelGGL((skinny_gemm_bf16x6_n32<rb, 0, Epilogue>), grid, block, lds, st, A, B, Q, K, (int)N, rows, ep); \
      }                                                                                                           \
    }                                                                                                             \
    break
  switch (RB) {
    UNIVS_LAUNCH_RB(1);
    UNIVS_LAUNCH_RB(2);
    UNIVS_LAUNCH_RB(3);
    UNIVS_LAUNCH_RB(4);
    UNIVS_LAUNCH_RB(5);
    UNIVS_LAUNCH_RB(6);
    default: UNIVS_LAUNCH_RB(7);
  }
#undef UNIVS_LAUNCH_RB
  return check_launch(what);
}

// column tiles per wave tile of the split kernel: UNIVS_MASKDEC_CT = 2 | 4 (default 2 where K allows the ring)
static int maskdec_ct(int K, long long out_bytes) {
  if (out_bytes >= 0x7FFFFFFFLL) return 4;      // the 32-column kernel stores through a 32-bit buffer range
  const int v = config().mask_decode_ct == 4 ? 4 : 2;
  return (v == 2 && K % (32 * SB2_RING) == 0) ? 2 : 4;
}

// 0 = by size (default), 1 = exact-f32 MFMA kernel, 2 = bf16 x 6 wherever its preconditions hold
static thread_local int g_maskdec_last = 0;
static int maskdec_impl() {
  const int v = config().mask_decode_impl;
  return (v < 0 || v > 2) ? 0 : v;
}
int mask_decode_last_impl() { return g_maskdec_last; }

static bool bf16x6_eligible(const void* A, const void* B, const void* out, int T, int Q, int K, long long N, int out_align) {
  const int impl = maskdec_impl();
  if (impl == 1) return false;
  if (K < 64 || K % 64 != 0 || N % 4 != 0 || (long long)K * N * 4 >= (1LL << 31)) return false;
  if ((long long)K * 16 * 6 > 160 * 1024) return false;           // not even one 16-row block of A fits
  if ((reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & (uintptr_t)(out_align - 1)))
    return false;
  if (impl == 2) return true;
  // by size: the A split (one LDS image per workgroup) has to be amortised over enough column tiles
  return N >= 8192;
}

template <typename Epilogue>
static int launch_skinny(const float* A, const float* B, int T, int Q, int K, long long N,
                         Epilogue ep, hipStream_t st, const char* what) {
  if (T == 0 || Q == 0 || N == 0) return UNIVS_OK;
  // rows per block-tile: 128 when Q is large, else the smallest multiple of 32 covering Q -- unless that leaves most of
  // the chip idle: the attention-mask maps of the coarse levels (23x40, 46x80) have 4 / 15 column tiles per frame, i.e.
  // 20 / 75 workgroups at 128 rows per workgroup, each running 512 dependent MFMAs per wave (measured 57-71 us for a 5-20 MB
  // problem: pure latency).  Fewer rows per workgroup = more workgroups and proportionally shorter MFMA chains; B is
  // re-read from L2 once per row block, which is noise at these sizes.
  const long long ctiles = (N + MD_BLOCK_N - 1) / MD_BLOCK_N;
  int MI = (Q + 31) / 32;
  if (MI > 4) MI = 4;
  while (MI > 1 && ctiles * ((Q + 32 * MI - 1) / (32 * MI)) * T < 192) --MI;
  if (MI == 3 && (Q + 63) / 64 == (Q + 95) / 96) MI = 2;   // same number of row blocks with less padding
  const int QP = 32 * MI;
  const int qtiles = (Q + QP - 1) / QP;
  // amortise the A staging and balance the grid: just under one block per CU (256 CUs) when the
  // problem is large enough, one tile per block otherwise
  long long tpb = (ctiles * qtiles * T + 255) / 256;
  if (tpb < 1) tpb = 1;
  if (tpb > 16) tpb = 16;
  if ((long long)K * N * 4 >= (1LL << 31)) {
    set_error("%s: K*N too large for a 32-bit buffer range", what);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const long long gx = (ctiles + tpb - 1) / tpb;
  const size_t lds = (size_t)K * (QP + 1) * sizeof(float);
  if (lds > 160 * 1024) {
    set_error("%s: K=%d too large for the LDS-resident A tile", what, K);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  dim3 grid((unsigned)gx, (unsigned)qtiles, (unsigned)T), block(MD_THREADS);
  if (MI == 1 && K == 256 && config().mask_decode_chunked == 0) {   // small maps: every row of B requested at once
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_f32_oneshot<Epilogue>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((skinny_gemm_f32_oneshot<Epilogue>), grid, block, lds, st, A, B, Q, (int)N, (int)tpb, ep);
    return check_launch(what);
  }
#define UNIVS_LAUNCH_MI(mi)                                                                         \
  do {                                                                                              \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_f32<mi, Epilogue>),            \
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                      \
    hipLaunchKernelGGL((skinny_gemm_f32<mi, Epilogue>), grid, block, lds, st, A, B, Q, K, N,        \
                       (int)tpb, ep);                                                               \
  } while (0)
  switch (MI) {
    case 1: UNIVS_LAUNCH_MI(1); break;
    case 2: UNIVS_LAUNCH_MI(2); break;
    case 3: UNIVS_LAUNCH_MI(3); break;
    default: UNIVS_LAUNCH_MI(4); break;
  }
#undef UNIVS_LAUNCH_MI
  return check_launch(what);
}

// rows with no visible key -> all keys visible (":390": attn_mask[all-True rows] = False)
__global__ __launch_bounds__(256) void attn_mask_row_reset(uint8_t* __restrict__ mask,
                                                            const unsigned* __restrict__ row_any, unsigned gen,
                                                            long long N) {
  const long long row = blockIdx.x;
  if (row_any[row] == gen) return;
  uint8_t* m = mask + row * N;
  for (long long i = threadIdx.x; i < N; i += blockDim.x) m[i] = 0;
}

int mask_decode_f32(const float* mask_embed, const float* mask_features, int T, int Q, int C,
                    long long HW, float* out, hipStream_t st) {
  if (T == 0 || Q == 0 || HW == 0) return UNIVS_OK;
  g_maskdec_last = 1;
  if (bf16x6_eligible(mask_embed, mask_features, out, T, Q, C, HW, 16) && (g_maskdec_last = 2))
    return maskdec_ct(C, (long long)Q * T * HW * 4) == 2
               ? launch_bf16x6<2>(mask_embed, mask_features, T, Q, C, HW,
                                  Store2Logits{out, T, (unsigned)((long long)Q * T * HW * 4)}, st, "mask_decode_bf16x6_n32")
               : launch_bf16x6<4>(mask_embed, mask_features, T, Q, C, HW, StoreLogits{out, T}, st, "mask_decode_bf16x6");
  return launch_skinny(mask_embed, mask_features, T, Q, C, HW, StoreLogits{out, T}, st,
                       "mask_decode_f32");
}

// generation == 0: the eager form -- flags zeroed, the contraction, the reset of the rows that stayed fully masked: attn_mask is the
// reference's tensor.  generation != 0 (the DEFERRED form): the contraction alone; row r of attn_mask is to be read as all-visible
// by the consumer wherever row_flags[r] != generation (univs_cross_attention_f32 takes the flags), or made explicit later by
// attn_mask_rows_reset.  The flags keep whatever older generations left in them: no memset, no second pass over the mask.
int mask_decode_attn_f32(const float* mask_embed, const float* feat_lowres, int T, int Q, int C,
                         long long hw, uint8_t* attn_mask, unsigned* row_any_ws, unsigned generation, hipStream_t st) {
  if (T == 0 || Q == 0 || hw == 0) return UNIVS_OK;
  const unsigned gen = generation ? generation : 1u;
  if (generation == 0) {
    hipError_t e = hipMemsetAsync(row_any_ws, 0, sizeof(unsigned) * (size_t)T * Q, st);
    if (e != hipSuccess) {
      set_error("mask_decode_attn_f32: memset failed: %s", hipGetErrorString(e));
      return UNIVS_ERR_LAUNCH;
    }
  }
  int rc;
  g_maskdec_last = 1;
  if (bf16x6_eligible(mask_embed, feat_lowres, attn_mask, T, Q, C, hw, 4) && (g_maskdec_last = 2))
    rc = maskdec_ct(C, (long long)T * Q * hw) == 2
             ? launch_bf16x6<2>(mask_embed, feat_lowres, T, Q, C, hw,
                                Store2AttnMask{attn_mask, row_any_ws, Q, (unsigned)((long long)T * Q * hw), (unsigned)((long long)T * Q * 4), gen},
                                st, "mask_decode_attn_bf16x6_n32")
             : launch_bf16x6<4>(mask_embed, feat_lowres, T, Q, C, hw, StoreAttnMask{attn_mask, row_any_ws, Q, gen}, st,
                                "mask_decode_attn_bf16x6");
  else
    rc = launch_skinny(mask_embed, feat_lowres, T, Q, C, hw, StoreAttnMask{attn_mask, row_any_ws, Q, gen},
                       st, "mask_decode_attn_f32");
  if (rc != UNIVS_OK || generation != 0) return rc;
  hipLaunchKernelGGL(attn_mask_row_reset, dim3((unsigned)(T * Q)), dim3(256), 0, st, attn_mask, row_any_ws, gen, hw);
  return check_launch("attn_mask_row_reset");
}

int attn_mask_rows_reset(uint8_t* attn_mask, const unsigned* row_flags, unsigned generation, long long rows, long long hw, hipStream_t st) {
  if (rows <= 0 || hw <= 0) return UNIVS_OK;
  hipLaunchKernelGGL(attn_mask_row_reset, dim3((unsigned)rows), dim3(256), 0, st, attn_mask, row_flags, generation, hw);
  return check_launch("attn_mask_row_reset");
}

}  // namespace univs
