// GroupNorm (+ optional ReLU) over NCHW planes, gfx950, HBM-bound.
//
// Replaces detectron2's Conv2d(norm=GroupNorm(32, C)[, activation=relu]) epilogues of the pixel decoder
// (mask2former/modeling/pixel_decoder/msdeformattn.py:214-232 input_proj, :262-283 lateral / output
// convs).  At [5, 256, 184, 320] ATen runs RowwiseMoments at 1.5 TB/s plus a separate apply (+ a separate
// ReLU): 0.55 ms per call.  Two passes are the minimum (a group = 8 x 58 880 floats does not fit on chip):
//   pass 1: every workgroup reduces one spatial chunk of one channel plane to (sum, sum of squares) of
//           x - shift, shift = the group's first element (keeps E[x^2] - E[x]^2 out of cancellation);
//   pass 2: every workgroup first folds the partials of its group (<= 1024 values), then normalises its
//           own chunk: y = (x - mean) * rstd * gamma[c] + beta[c] (biased variance, as ATen), optional ReLU.
#include "common.h"

namespace univs {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float block_sum_256(float v, float* lds) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[w] = v;
  __syncthreads();
  return lds[0] + lds[1] + lds[2] + lds[3];
}

// grid: (chunks, N * C); plane = blockIdx.y
__global__ __launch_bounds__(256) void gn_partials_kernel(const float* __restrict__ x, int C, int Cg, long long HW,
                                                          int chunks, float* __restrict__ partials) {
  __shared__ float red[4];
  const long long plane = blockIdx.y;
  const long long n = plane / C, c = plane % C, g = c / Cg;
  const float shift = x[(n * C + g * Cg) * HW];
  const long long per = (HW + chunks - 1) / chunks;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < HW ? lo + per : HW;
  const float* p = x + plane * HW;
  float s1 = 0.f, s2 = 0.f;
  if ((HW & 3) == 0 && (per & 3) == 0) {
    const v4f* p4 = reinterpret_cast<const v4f*>(p);
    for (long long i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
      const v4f d = p4[i] - shift;
      s1 += (d.x + d.y) + (d.z + d.w);
      s2 += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
      const float d = p[i] - shift;
      s1 += d;
      s2 += d * d;
    }
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  if (threadIdx.x == 0) {
    partials[(plane * chunks + blockIdx.x) * 2] = s1;
    partials[(plane * chunks + blockIdx.x) * 2 + 1] = s2;
  }
}

__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int C, int Cg, long long HW,
                                                       int chunks, float eps, int relu,
                                                       const float* __restrict__ partials, float* __restrict__ out) {
  __shared__ float red[4];
  const long long plane = blockIdx.y;
  const long long n = plane / C, c = plane % C, g = c / Cg;
  const long long g0 = n * C + g * Cg;          // first plane of the group
  const float shift = x[g0 * HW];
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < Cg * chunks; i += 256) {
    s1 += partials[(g0 * chunks + i) * 2];
    s2 += partials[(g0 * chunks + i) * 2 + 1];
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  const float inv_n = 1.f / ((float)Cg * (float)HW);
  const float m1 = s1 * inv_n;
  const float var = fmaxf(s2 * inv_n - m1 * m1, 0.f);
  const float mean = shift + m1;
  const float scale = gamma[c] / sqrtf(var + eps);
  const float bias = beta[c] - mean * scale;
  const long long per = (HW + chunks - 1) / chunks;
  const long long lo = (long long)blockIdx.x * per, hi = lo + per < HW ? lo + per : HW;
  const float* p = x + plane * HW;
  float* q = out + plane * HW;
  if ((HW & 3) == 0 && (per & 3) == 0) {
    const v4f* p4 = reinterpret_cast<const v4f*>(p);
    v4f* q4 = reinterpret_cast<v4f*>(q);
    for (long long i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
      v4f y = __builtin_elementwise_fma(p4[i], (v4f){scale, scale, scale, scale}, (v4f){bias, bias, bias, bias});
      if (relu) y = __builtin_elementwise_max(y, (v4f){0.f, 0.f, 0.f, 0.f});
      q4[i] = y;
    }
  } else {
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
      const float y = fmaf(p[i], scale, bias);
      q[i] = relu ? fmaxf(y, 0.f) : y;
    }
  }
}

// the per-plane affine form of the normalisation -- scale = gamma / sqrt(var + eps), bias = beta - mean * scale, the very values
// gn_apply_kernel uses (same sums in the same order) -- for a consumer that applies it while reading x (resample.hip:
// upsample2x_add_kernel).  One workgroup per plane.
__global__ __launch_bounds__(256) void gn_affine_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int C, int Cg, long long HW, int chunks,
                                                        float eps, const float* __restrict__ partials, float* __restrict__ affine) {
  __shared__ float red[4];
  const long long plane = blockIdx.x;
  const long long n = plane / C, c = plane % C, g = c / Cg;
  const long long g0 = n * C + g * Cg;
  const float shift = x[g0 * HW];
  float s1 = 0.f, s2 = 0.f;
  for (int i = threadIdx.x; i < Cg * chunks; i += 256) {
    s1 += partials[(g0 * chunks + i) * 2];
    s2 += partials[(g0 * chunks + i) * 2 + 1];
  }
  s1 = block_sum_256(s1, red);
  s2 = block_sum_256(s2, red);
  const float inv_n = 1.f / ((float)Cg * (float)HW);
  const float m1 = s1 * inv_n;
  const float var = fmaxf(s2 * inv_n - m1 * m1, 0.f);
  const float mean = shift + m1;
  const float scale = gamma[c] / sqrtf(var + eps);
  const float bias = beta[c] - mean * scale;
  if (threadIdx.x == 0) {
    affine[2 * plane] = scale;
    affine[2 * plane + 1] = bias;
  }
}

// workspace: N * C * chunks * 2 floats; chunks is chosen here and returned through *chunks_out when ws == NULL
static int gn_chunks(int C, int groups, long long HW) {
  const int Cg = C / groups;
  // enough workgroups to fill the chip, chunks of >= 4 K elements, partial count per group <= 1024
  int chunks = (int)((HW + 8191) / 8192);
  if (chunks < 1) chunks = 1;
  while ((long long)chunks * Cg > 1024 && chunks > 1) chunks = (chunks + 1) / 2;
  if (chunks > 1) {   // keep chunk boundaries 16-B aligned when the plane is
    const long long per = (HW + chunks - 1) / chunks;
    if ((HW & 3) == 0 && (per & 3) != 0) {
      const long long per4 = (per + 3) / 4 * 4;
      chunks = (int)((HW + per4 - 1) / per4);
    }
  }
  return chunks;
}

int group_norm_f32(const float* x, const float* gamma, const float* beta, int N, int C, long long HW, int groups,
                   float eps, int relu, float* ws, long long ws_floats, float* out, hipStream_t st) {
  const int Cg = C / groups;
  const int chunks = gn_chunks(C, groups, HW);
  if ((long long)N * C * chunks * 2 > ws_floats) {
    set_error("group_norm_f32: workspace too small (%lld floats, need %lld)", ws_floats, (long long)N * C * chunks * 2);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const dim3 grid((unsigned)chunks, (unsigned)(N * C));
  hipLaunchKernelGGL(gn_partials_kernel, grid, dim3(256), 0, st, x, C, Cg, HW, chunks, ws);
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(256), 0, st, x, gamma, beta, C, Cg, HW, chunks, eps, relu, ws, out);
  return check_launch("group_norm_f32");
}

// statistics only: affine [N * C][2] = (scale, bias) per plane, y = x * scale + bias being the normalised value
int group_norm_affine_f32(const float* x, const float* gamma, const float* beta, int N, int C, long long HW, int groups, float eps,
                          float* ws, long long ws_floats, float* affine, hipStream_t st) {
  const int Cg = C / groups;
  const int chunks = gn_chunks(C, groups, HW);
  if ((long long)N * C * chunks * 2 > ws_floats) {
    set_error("group_norm_affine_f32: workspace too small (%lld floats, need %lld)", ws_floats, (long long)N * C * chunks * 2);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL(gn_partials_kernel, dim3((unsigned)chunks, (unsigned)(N * C)), dim3(256), 0, st, x, C, Cg, HW, chunks, ws);
  hipLaunchKernelGGL(gn_affine_kernel, dim3((unsigned)(N * C)), dim3(256), 0, st, x, gamma, beta, C, Cg, HW, chunks, eps, ws, affine);
  return check_launch("group_norm_affine_f32");
}

}  // namespace univs
