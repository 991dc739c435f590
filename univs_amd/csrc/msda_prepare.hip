// Sampling locations and attention weights of MSDeformAttn from the raw projections, in one pass (gfx950).
//
// Replaces the elementwise tail of MSDeformAttn.forward
// (mask2former/modeling/pixel_decoder/ops/modules/ms_deform_attn.py:100-113):
//     attention_weights = softmax(attention_weights_raw.view(N, Lq, M, L*P), -1)
//     sampling_locations = reference_points[:, :, None, :, None, :] + sampling_offsets / (W_l, H_l)
// which ATen runs as softmax + div + add (+ a copy) over the 14.8 + 7.4 MB/frame tensors (4 kernels, 8 passes).
// One thread per (n, q, m): reads its L*P*2 offsets and L*P logits from the merged projection row
// [N, Lq, M*L*P*2 | M*L*P], writes loc [N, Lq, M, L, P, 2] and attn [N, Lq, M, L, P].
// Arithmetic in the reference's order: off / normalizer (IEEE division), then + reference point; softmax as
// exp(x - max) / sum.
#include "common.h"

namespace univs {

template <int L, int P>
__global__ __launch_bounds__(256) void msda_prepare_f32_kernel(const float* __restrict__ qp, int row_stride, int n_off,
                                                               const float* __restrict__ ref, long long ref_batch_stride,
                                                               LevelTable lv, int Lq, int M, long long total,
                                                               float* __restrict__ loc, float* __restrict__ attn) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // (n * Lq + q) * M + m
  if (idx >= total) return;
  const int m = (int)(idx % M);
  const long long nq = idx / M;
  const long long n = nq / Lq, q = nq % Lq;
  const float* row = qp + nq * row_stride;
  const float* off = row + (long long)m * (L * P * 2);
  const float* lg = row + n_off + (long long)m * (L * P);
  const float* rp = ref + n * ref_batch_stride + q * (L * 2);
  float* lo = loc + idx * (L * P * 2);
  float* ao = attn + idx * (L * P);
  float e[L * P];
  float mx = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < L * P; ++i) {
    e[i] = lg[i];
    mx = fmaxf(mx, e[i]);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < L * P; ++i) {
    e[i] = expf(e[i] - mx);
    sum += e[i];
  }
#pragma unroll
  for (int i = 0; i < L * P; ++i) ao[i] = e[i] / sum;
#pragma unroll
  for (int l = 0; l < L; ++l) {
    const float rx = rp[l * 2], ry = rp[l * 2 + 1];
    const float W = (float)lv.W[l], H = (float)lv.H[l];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      lo[(l * P + p) * 2] = rx + off[(l * P + p) * 2] / W;
      lo[(l * P + p) * 2 + 1] = ry + off[(l * P + p) * 2 + 1] / H;
    }
  }
}

// returns UNIVS_ERR_NOT_IMPLEMENTED for (L, P) combinations without an instantiation
int msda_prepare_f32(const float* qp, int row_stride, int n_off, const float* ref, long long ref_batch_stride,
                     const LevelTable& lv, int N, int Lq, int M, int L, int P, float* loc, float* attn, hipStream_t st) {
  const long long total = (long long)N * Lq * M;
  if (total == 0) return UNIVS_OK;
  const unsigned grid = (unsigned)((total + 255) / 256);
#define UNIVS_PREP(LL, PP)                                                                                        \
  hipLaunchKernelGGL((msda_prepare_f32_kernel<LL, PP>), dim3(grid), dim3(256), 0, st, qp, row_stride, n_off, ref, \
                     ref_batch_stride, lv, Lq, M, total, loc, attn)
  if (P == 4 && L == 3) UNIVS_PREP(3, 4);
  else if (P == 4 && L == 4) UNIVS_PREP(4, 4);
  else if (P == 4 && L == 2) UNIVS_PREP(2, 4);
  else if (P == 4 && L == 1) UNIVS_PREP(1, 4);
  else return UNIVS_ERR_NOT_IMPLEMENTED;
#undef UNIVS_PREP
  return check_launch("msda_prepare_f32");
}

}  // namespace univs
