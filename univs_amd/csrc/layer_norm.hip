// Row LayerNorm with an optional fused residual add (gfx950, HBM-bound).
//
// Replaces nn.LayerNorm / `norm(x + y)` on the token tensors of the path: Swin blocks (norm1 / norm2 and
// the residual in between, mask2former/modeling/backbone/swin.py:236-262), the MSDeformAttn encoder
// layers (`src = norm1(src + src2)`, msdeformattn.py:61-95) and the decoder layers.  ATen's
// vectorized_layer_norm_kernel reaches 0.8 TB/s at C = 96 (Swin stage 1) and 2 TB/s at C = 256; the
// residual add is a separate 3-pass kernel in front of it.
//
// One sub-wave of G lanes (G = 32 or 64, 16-B chunks) per row, the row stays in registers: exact
// two-pass statistics (mean, then sum of squared deviations), biased variance, rstd = 1/sqrt(var + eps),
// y = (x - mean) * rstd * gamma + beta -- the same expression order as ATen's kernel.
#include "common.h"

namespace univs {

typedef float v4f __attribute__((ext_vector_type(4)));

template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// NV: 16-B chunks per lane; G: lanes per row.  C % 4 == 0, C <= 4 * G * NV.
// POST_ADD: a second output out2 = y + addend (the encoder layer's `src + pos` for the next layer's query,
// msdeformattn.py:61-63, 85-95: the normalised row is still in registers).
template <int G, int NV, bool HAS_RES, bool WRITE_SUM, bool POST_ADD = false>
__global__ __launch_bounds__(256) void layer_norm_f32_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ res,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, long long rows, int C,
                                                             float eps, float* __restrict__ sum_out,
                                                             float* __restrict__ out, const float* __restrict__ addend = nullptr,
                                                             float* __restrict__ out2 = nullptr, long long addend_rows = 1) {
  constexpr int RPB = 256 / G;   // rows per block
  const int sub = threadIdx.x / G, lane = threadIdx.x % G;
  const int nchunk = C / 4;
  const float inv_c = 1.f / (float)C;
  for (long long row = (long long)blockIdx.x * RPB + sub; row < rows; row += (long long)gridDim.x * RPB) {
    const v4f* xr = reinterpret_cast<const v4f*>(x + row * C);
    v4f v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + j * G;
      if (c < nchunk) {
        v[j] = xr[c];
        if (HAS_RES) v[j] += reinterpret_cast<const v4f*>(res + row * C)[c];
        if (WRITE_SUM) reinterpret_cast<v4f*>(sum_out + row * C)[c] = v[j];
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      } else {
        v[j] = (v4f){0.f, 0.f, 0.f, 0.f};
      }
    }
    const float mean = group_sum<G>(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (lane + j * G < nchunk) {
        const v4f d = v[j] - mean;
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      }
    }
    const float rstd = 1.f / sqrtf(group_sum<G>(q) * inv_c + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + j * G;
      if (c < nchunk) {
        const v4f g = reinterpret_cast<const v4f*>(gamma)[c], b = reinterpret_cast<const v4f*>(beta)[c];
        const v4f y = (v[j] - mean) * rstd * g + b;
        reinterpret_cast<v4f*>(out + row * C)[c] = y;
        // (the addend may have fewer rows than x: [1, S, C] position embeddings against [N, S, C] tokens)
        if (POST_ADD) reinterpret_cast<v4f*>(out2 + row * C)[c] = y + reinterpret_cast<const v4f*>(addend + (row % addend_rows) * C)[c];
      }
    }
  }
}

template <int G, int NV>
static void launch_ln(const float* x, const float* res, const float* gamma, const float* beta, long long rows, int C,
                      float eps, float* sum_out, float* out, const float* addend, float* out2, long long addend_rows, hipStream_t st) {
  constexpr int RPB = 256 / G;
  const long long want = (rows + RPB - 1) / RPB;
  const unsigned grid = (unsigned)(want < 256LL * 32 ? want : 256LL * 32);   // grid-stride beyond 32 blocks per CU
#define UNIVS_LN_LAUNCH(R, S) \
  hipLaunchKernelGGL((layer_norm_f32_kernel<G, NV, R, S>), dim3(grid), dim3(256), 0, st, x, res, gamma, beta, rows, C, eps, sum_out, out)
  if (addend && res && !sum_out) {
    hipLaunchKernelGGL((layer_norm_f32_kernel<G, NV, true, false, true>), dim3(grid), dim3(256), 0, st, x, res, gamma, beta, rows, C, eps,
                       sum_out, out, addend, out2, addend_rows);
  } else if (res && sum_out) UNIVS_LN_LAUNCH(true, true);
  else if (res) UNIVS_LN_LAUNCH(true, false);
  else UNIVS_LN_LAUNCH(false, false);
#undef UNIVS_LN_LAUNCH
}

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED for row lengths this kernel does not cover
// addend / out2 (both or neither): the second output y + addend; only together with `res` and without `sum_out`
int layer_norm_f32(const float* x, const float* res, const float* gamma, const float* beta, long long rows, int C,
                   float eps, float* sum_out, float* out, const float* addend, float* out2, long long addend_rows, hipStream_t st) {
  const int nchunk = C / 4;
  if (C % 4 != 0 || nchunk > 64 * 12) return UNIVS_ERR_NOT_IMPLEMENTED;
  if ((addend != nullptr) != (out2 != nullptr) || (addend && (!res || sum_out || addend_rows < 1 || rows % addend_rows != 0)))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  if (nchunk <= 32) launch_ln<32, 1>(x, res, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend_rows, st);
  else if (nchunk <= 64) launch_ln<64, 1>(x, res, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend_rows, st);
  else if (nchunk <= 128) launch_ln<64, 2>(x, res, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend_rows, st);
  else if (nchunk <= 192) launch_ln<64, 3>(x, res, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend_rows, st);
  else if (nchunk <= 256) launch_ln<64, 4>(x, res, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend_rows, st);
  else if (nchunk <= 384) launch_ln<64, 6>(x, res, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend_rows, st);
  else if (nchunk <= 512) launch_ln<64, 8>(x, res, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend_rows, st);
  else launch_ln<64, 12>(x, res, gamma, beta, rows, C, eps, sum_out, out, addend, out2, addend_rows, st);
  return check_launch("layer_norm_f32");
}

// ---- Swin PatchMerging's gather + LayerNorm in one pass (mask2former/modeling/backbone/swin.py:341-386: pad H, W to even, concatenate
// the four pixels of every 2 x 2 patch along the channels in the order (0,0), (1,0), (0,1), (1,1) [dy, dx], `norm` over the 4 C
// channels).  ATen runs the concatenation as a strided copy of its own (82 us for Swin-T stage 1 at 720p x 5) in front of the
// LayerNorm; here the row is gathered straight into the registers that are normalised.  x [B, H, W, C], out [B, H2 * W2, 4 C].
template <int G, int NV>
__global__ __launch_bounds__(256) void patch_merge_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, int B, int H, int W, int C, float eps,
                                                               float* __restrict__ out) {
  constexpr int RPB = 256 / G;
  const int sub = threadIdx.x / G, lane = threadIdx.x % G;
  const int H2 = (H + 1) / 2, W2 = (W + 1) / 2;
  const int cq = C / 4, nchunk = C;                              // 16-byte chunks per source pixel / per output row
  const long long rows = (long long)B * H2 * W2;
  const float inv_c = 1.f / (float)(4 * C);
  for (long long row = (long long)blockIdx.x * RPB + sub; row < rows; row += (long long)gridDim.x * RPB) {
    const int b = (int)(row / ((long long)H2 * W2));
    const int rem = (int)(row - (long long)b * H2 * W2);
    const int y2 = rem / W2, x2 = rem - y2 * W2;
    v4f v[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + j * G;
      v[j] = (v4f){0.f, 0.f, 0.f, 0.f};
      if (c < nchunk) {
        const int seg = c / cq, w = c - seg * cq;
        const int yy = 2 * y2 + (seg & 1), xx = 2 * x2 + (seg >> 1);
        if (yy < H && xx < W) v[j] = reinterpret_cast<const v4f*>(x + (((long long)b * H + yy) * W + xx) * C)[w];   // (else: the zero padding)
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      }
    }
    const float mean = group_sum<G>(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      if (lane + j * G < nchunk) {
        const v4f d = v[j] - mean;
        q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
      }
    }
    const float rstd = 1.f / sqrtf(group_sum<G>(q) * inv_c + eps);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int c = lane + j * G;
      if (c < nchunk) {
        const v4f g = reinterpret_cast<const v4f*>(gamma)[c], bt = reinterpret_cast<const v4f*>(beta)[c];
        reinterpret_cast<v4f*>(out + row * 4 * C)[c] = (v[j] - mean) * rstd * g + bt;
      }
    }
  }
}

template <int NV>
static void launch_pm(const float* x, const float* gamma, const float* beta, int B, int H, int W, int C, float eps, float* out,
                      hipStream_t st) {
  const long long rows = (long long)B * ((H + 1) / 2) * ((W + 1) / 2);
  const long long want = (rows + 3) / 4;
  const unsigned grid = (unsigned)(want < 256LL * 32 ? want : 256LL * 32);
  hipLaunchKernelGGL((patch_merge_norm_kernel<64, NV>), dim3(grid), dim3(256), 0, st, x, gamma, beta, B, H, W, C, eps, out);
}

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED (C % 4 != 0 or C > 768)
int patch_merge_norm_f32(const float* x, const float* gamma, const float* beta, int B, int H, int W, int C, float eps, float* out,
                         hipStream_t st) {
  if (C % 4 != 0 || C > 768 || C < 4) return UNIVS_ERR_NOT_IMPLEMENTED;
  if ((long long)B * H * W <= 0) return UNIVS_OK;
  if (C <= 64) launch_pm<1>(x, gamma, beta, B, H, W, C, eps, out, st);
  else if (C <= 128) launch_pm<2>(x, gamma, beta, B, H, W, C, eps, out, st);
  else if (C <= 192) launch_pm<3>(x, gamma, beta, B, H, W, C, eps, out, st);
  else if (C <= 256) launch_pm<4>(x, gamma, beta, B, H, W, C, eps, out, st);
  else if (C <= 384) launch_pm<6>(x, gamma, beta, B, H, W, C, eps, out, st);
  else if (C <= 512) launch_pm<8>(x, gamma, beta, B, H, W, C, eps, out, st);
  else launch_pm<12>(x, gamma, beta, B, H, W, C, eps, out, st);
  return check_launch("patch_merge_norm_f32");
}

}  // namespace univs
