// Batched 2-D transpose  out[b][c][r] = in[b][r][c]  (float32), gfx950.
//
// The layout changes at the edges of the Swin backbone: the reference returns its stage outputs as NCHW tensors
// (mask2former/modeling/backbone/swin.py:676-683: `x_out.view(-1, H, W, C).permute(0, 3, 1, 2).contiguous()`) while the
// blocks work on tokens [B, H*W, C], and PatchEmbed flattens its convolution output the other way (:331-336).  ATen runs
// a permuted copy at ~0.6 TB/s (360 us for the 113-MB res2 map at 720p x 5 frames); this is the classic LDS tile
// transpose: 64 x 64 tiles, 16-byte loads along the input's fast axis, 16-byte stores along the output's.
#include "common.h"

namespace univs {

constexpr int TR_TILE = 64;

__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
  __shared__ float tile[TR_TILE][TR_TILE + 1];
  const long long base = (long long)blockIdx.z * R * C;
  const int r0 = blockIdx.y * TR_TILE, c0 = blockIdx.x * TR_TILE;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16 threads, 4 floats each along the fast axis
#pragma unroll
  for (int p = 0; p < TR_TILE; p += 16) {
    const int r = r0 + p + ty, c = c0 + 4 * tx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && c < C) v = *reinterpret_cast<const float4*>(in + base + (long long)r * C + c);   // C % 4 == 0 (host-checked)
    tile[p + ty][4 * tx + 0] = v.x;
    tile[p + ty][4 * tx + 1] = v.y;
    tile[p + ty][4 * tx + 2] = v.z;
    tile[p + ty][4 * tx + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < TR_TILE; p += 16) {
    const int c = c0 + p + ty, r = r0 + 4 * tx;
    if (c < C && r < R) {                                                                        // R % 4 == 0 (host-checked)
      const float4 v = make_float4(tile[4 * tx + 0][p + ty], tile[4 * tx + 1][p + ty], tile[4 * tx + 2][p + ty],
                                   tile[4 * tx + 3][p + ty]);
      *reinterpret_cast<float4*>(out + base + (long long)c * R + r) = v;
    }
  }
}

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED when R or C is not a multiple of 4 / the pointers are not 16-byte aligned
int transpose_f32(const float* in, float* out, long long B, int R, int C, hipStream_t st) {
  if (B <= 0 || R <= 0 || C <= 0) return UNIVS_OK;
  if (R % 4 != 0 || C % 4 != 0 || B > 65535 || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out) & 15))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  dim3 grid((unsigned)((C + TR_TILE - 1) / TR_TILE), (unsigned)((R + TR_TILE - 1) / TR_TILE), (unsigned)B);
  if (grid.y > 65535) return UNIVS_ERR_NOT_IMPLEMENTED;
  hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, st, in, out, R, C);
  return check_launch("transpose_f32");
}

}  // namespace univs
