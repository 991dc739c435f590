// Masked row softmax for multi-head attention scores, in place (gfx950, HBM-bound).
//
// Replaces the `masked_fill(attn_mask, -inf)` + `softmax(-1)` pair inside nn.MultiheadAttention as the
// decoder calls it (univs/modeling/transformer_decoder/transformer_layers.py:101-105 with the boolean
// per-frame attention mask of ...decoder_univs.py:390-405).  On the [5, 8, 100, 14720] scores of the
// finest level ATen needs a clone, a masked_fill and a softmax: 5 passes, 1.15 ms; this is one read
// and one write.  One workgroup per row, the row lives in registers (S <= 16384), otherwise three
// streaming passes.  Same formula as ATen: exp(x - max) / sum, masked entries contribute 0 (a fully
// masked row gives NaN there and here; the caller never produces one, ...decoder_univs.py:390).
#include "common.h"

namespace univs {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wg_reduce(float v, float* lds, bool is_max) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float u = __shfl_xor(v, o, 64);
    v = is_max ? fmaxf(v, u) : v + u;
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(lds[0], lds[1]), fmaxf(lds[2], lds[3])) : (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// scores [N, h, L, S]; mask [N, L, S] bytes (non-zero = masked out) or NULL
template <int KMAX>
__global__ __launch_bounds__(256) void masked_softmax_f32_kernel(float* __restrict__ scores,
                                                                 const unsigned char* __restrict__ mask, int h, int L,
                                                                 int S) {
  __shared__ float red[4];
  const long long row = blockIdx.x;                 // (n * h + head) * L + l
  const long long n = row / ((long long)h * L), l = row % L;
  float* p = scores + row * S;
  const unsigned char* m = mask ? mask + (n * L + l) * (long long)S : nullptr;
  const float NEG = -__builtin_inff();
  float v[KMAX];
  float mx = NEG;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int i = threadIdx.x + k * 256;
    v[k] = NEG;
    if (i < S) {
      const float x = p[i];
      v[k] = (m && m[i]) ? NEG : x;
    }
    mx = fmaxf(mx, v[k]);
  }
  mx = wg_reduce(mx, red, true);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    v[k] = expf(v[k] - mx);       // exp(-inf) = 0 for masked / out-of-range entries
    sum += v[k];
  }
  sum = wg_reduce(sum, red, false);
  const float inv = 1.f / sum;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < S) p[i] = v[k] * inv;
  }
}

__global__ __launch_bounds__(256) void masked_softmax_f32_stream_kernel(float* __restrict__ scores,
                                                                        const unsigned char* __restrict__ mask, int h,
                                                                        int L, int S) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const long long n = row / ((long long)h * L), l = row % L;
  float* p = scores + row * S;
  const unsigned char* m = mask ? mask + (n * L + l) * (long long)S : nullptr;
  const float NEG = -__builtin_inff();
  float mx = NEG;
  for (int i = threadIdx.x; i < S; i += 256) mx = fmaxf(mx, (m && m[i]) ? NEG : p[i]);
  mx = wg_reduce(mx, red, true);
  float sum = 0.f;
  for (int i = threadIdx.x; i < S; i += 256) sum += (m && m[i]) ? 0.f : expf(p[i] - mx);
  sum = wg_reduce(sum, red, false);
  const float inv = 1.f / sum;
  for (int i = threadIdx.x; i < S; i += 256) p[i] = (m && m[i]) ? 0.f : expf(p[i] - mx) * inv;
}

int masked_softmax_f32(float* scores, const unsigned char* mask, int N, int h, int L, int S, hipStream_t st) {
  const long long rows = (long long)N * h * L;
  if (rows > 0x7fffffffLL) return UNIVS_ERR_INVALID_ARGUMENT;
  const dim3 grid((unsigned)rows), block(256);
  if (S <= 256 * 4) hipLaunchKernelGGL(masked_softmax_f32_kernel<4>, grid, block, 0, st, scores, mask, h, L, S);
  else if (S <= 256 * 16) hipLaunchKernelGGL(masked_softmax_f32_kernel<16>, grid, block, 0, st, scores, mask, h, L, S);
  else if (S <= 256 * 64) hipLaunchKernelGGL(masked_softmax_f32_kernel<64>, grid, block, 0, st, scores, mask, h, L, S);
  else hipLaunchKernelGGL(masked_softmax_f32_stream_kernel, grid, block, 0, st, scores, mask, h, L, S);
  return check_launch("masked_softmax_f32");
}

}  // namespace univs
