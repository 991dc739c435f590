// Mask decode (mask-embed x pixel-feature contraction) for gfx950 on the exact-f32 MFMA path.
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

namespace univs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
};

// Epilogue 2: attention mask byte (1 = masked out <=> logit < 0 <=> sigmoid < 0.5) and a per-row
// "has a visible key" flag used by the all-masked-row reset.
struct StoreAttnMask {
  uint8_t* mask;       // [T, Q, N]
  unsigned* row_any;   // [T, Q]  (zeroed before the launch)
  int Q;
  __device__ __forceinline__ void operator()(int t, int q, long long n, long long N, float v) const {
    const bool masked = v < 0.f;
    mask[((long long)t * Q + q) * N + n] = masked ? 1 : 0;
    if (!masked) row_any[t * Q + q] = 1u;  // benign race: every writer stores the same value
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

template <typename Epilogue>
static int launch_skinny(const float* A, const float* B, int T, int Q, int K, long long N,
                         Epilogue ep, hipStream_t st, const char* what) {
  if (T == 0 || Q == 0 || N == 0) return UNIVS_OK;
  // rows per block-tile: 128 when Q is large, else the smallest multiple of 32 covering Q
  int MI = (Q + 31) / 32;
  if (MI > 4) MI = 4;
  const int QP = 32 * MI;
  const int qtiles = (Q + QP - 1) / QP;
  const long long ctiles = (N + MD_BLOCK_N - 1) / MD_BLOCK_N;
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
                                                            const unsigned* __restrict__ row_any,
                                                            long long N) {
  const long long row = blockIdx.x;
  if (row_any[row] != 0u) return;
  uint8_t* m = mask + row * N;
  for (long long i = threadIdx.x; i < N; i += blockDim.x) m[i] = 0;
}

int mask_decode_f32(const float* mask_embed, const float* mask_features, int T, int Q, int C,
                    long long HW, float* out, hipStream_t st) {
  return launch_skinny(mask_embed, mask_features, T, Q, C, HW, StoreLogits{out, T}, st,
                       "mask_decode_f32");
}

int mask_decode_attn_f32(const float* mask_embed, const float* feat_lowres, int T, int Q, int C,
                         long long hw, uint8_t* attn_mask, unsigned* row_any_ws, hipStream_t st) {
  if (T == 0 || Q == 0 || hw == 0) return UNIVS_OK;
  hipError_t e = hipMemsetAsync(row_any_ws, 0, sizeof(unsigned) * (size_t)T * Q, st);
  if (e != hipSuccess) {
    set_error("mask_decode_attn_f32: memset failed: %s", hipGetErrorString(e));
    return UNIVS_ERR_LAUNCH;
  }
  int rc = launch_skinny(mask_embed, feat_lowres, T, Q, C, hw, StoreAttnMask{attn_mask, row_any_ws, Q},
                         st, "mask_decode_attn_f32");
  if (rc != UNIVS_OK) return rc;
  hipLaunchKernelGGL(attn_mask_row_reset, dim3((unsigned)(T * Q)), dim3(256), 0, st, attn_mask,
                     row_any_ws, hw);
  return check_launch("attn_mask_row_reset");
}

}  // namespace univs
