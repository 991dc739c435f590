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

// scores [N, h, L, S]; mask [N, L, S] bytes (non-zero = masked out) or NULL.
// VEC = 4: S % 4 == 0, 16-B score loads and one 4-B mask load per 4 elements; VEC = 1: any S.
template <int KMAX, int VEC>
__global__ __launch_bounds__(256) void masked_softmax_f32_kernel(float* __restrict__ scores,
                                                                 const unsigned char* __restrict__ mask, int h, int L,
                                                                 int S) {
  __shared__ float red[4];
  const long long row = blockIdx.x;                 // (n * h + head) * L + l
  const long long n = row / ((long long)h * L), l = row % L;
  float* p = scores + row * S;
  const unsigned char* m = mask ? mask + (n * L + l) * (long long)S : nullptr;
  const float NEG = -__builtin_inff();
  float v[KMAX][VEC];
  float mx = NEG;
  const int nvec = S / VEC;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int i = threadIdx.x + k * 256;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[k][e] = NEG;
    if (i < nvec) {
      if (VEC == 4) {
        const v4f x = reinterpret_cast<const v4f*>(p)[i];
        const unsigned mm = m ? reinterpret_cast<const unsigned*>(m)[i] : 0u;
        v[k][0] = (mm & 0x000000ffu) ? NEG : x.x;
        v[k][1 % VEC] = (mm & 0x0000ff00u) ? NEG : x.y;
        v[k][2 % VEC] = (mm & 0x00ff0000u) ? NEG : x.z;
        v[k][3 % VEC] = (mm & 0xff000000u) ? NEG : x.w;
      } else {
        v[k][0] = (m && m[i]) ? NEG : p[i];
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) mx = fmaxf(mx, v[k][e]);
  }
  mx = wg_reduce(mx, red, true);
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      v[k][e] = expf(v[k][e] - mx);       // exp(-inf) = 0 for masked / out-of-range entries
      sum += v[k][e];
    }
  sum = wg_reduce(sum, red, false);
  const float inv = 1.f / sum;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int i = threadIdx.x + k * 256;
    if (i < nvec) {
      if (VEC == 4) reinterpret_cast<v4f*>(p)[i] = (v4f){v[k][0] * inv, v[k][1 % VEC] * inv, v[k][2 % VEC] * inv, v[k][3 % VEC] * inv};
      else p[i] = v[k][0] * inv;
    }
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
  // rows of S % 4 == 0 whose row starts (and mask rows) stay 16-B / 4-B aligned take the vector path
  const bool vec = (S % 4 == 0) && ((reinterpret_cast<uintptr_t>(scores) & 15) == 0) &&
                   (!mask || (reinterpret_cast<uintptr_t>(mask) & 3) == 0);
#define UNIVS_SM(K, V) hipLaunchKernelGGL((masked_softmax_f32_kernel<K, V>), grid, block, 0, st, scores, mask, h, L, S)
  if (vec && S <= 1024 * 4) UNIVS_SM(4, 4);
  else if (vec && S <= 1024 * 16) UNIVS_SM(16, 4);
  else if (!vec && S <= 256 * 4) UNIVS_SM(4, 1);
  else if (!vec && S <= 256 * 16) UNIVS_SM(16, 1);
  else if (!vec && S <= 256 * 64) UNIVS_SM(64, 1);
  else hipLaunchKernelGGL(masked_softmax_f32_stream_kernel, grid, block, 0, st, scores, mask, h, L, S);
#undef UNIVS_SM
  return check_launch("masked_softmax_f32");
}

}  // namespace univs
