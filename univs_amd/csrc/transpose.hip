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

// in_bstride / out_bstride: floats between consecutive matrices of the input / output (R * C when dense; larger: row ranges of a
// wider batch tensor).  row_affine [B * R][2] (optional): input row (b, r) is read as x * scale + bias -- the GroupNorm of an NCHW
// tensor (group_norm.hip: gn_affine_kernel, the expression of gn_apply_kernel) applied on the way through.  addend [C][R] + out2
// (optional, both or neither): a second output out2 = out + addend, same layout as out -- the encoder's `src + pos`.
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C,
                                                            long long in_bstride, long long out_bstride,
                                                            const float* __restrict__ row_affine, const float* __restrict__ addend,
                                                            float* __restrict__ out2) {
  __shared__ float tile[TR_TILE][TR_TILE + 1];
  const long long base = (long long)blockIdx.z * out_bstride, ibase = (long long)blockIdx.z * in_bstride;
  const int r0 = blockIdx.y * TR_TILE, c0 = blockIdx.x * TR_TILE;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16 threads, 4 floats each along the fast axis
#pragma unroll
  for (int p = 0; p < TR_TILE; p += 16) {
    const int r = r0 + p + ty, c = c0 + 4 * tx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < R && c < C) {
      v = *reinterpret_cast<const float4*>(in + ibase + (long long)r * C + c);   // C % 4 == 0 (host-checked)
      if (row_affine) {
        const float sc = row_affine[((long long)blockIdx.z * R + r) * 2], bi = row_affine[((long long)blockIdx.z * R + r) * 2 + 1];
        v = make_float4(fma_single(v.x, sc, bi), fma_single(v.y, sc, bi), fma_single(v.z, sc, bi), fma_single(v.w, sc, bi));   // (common.h)
      }
    }
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
      if (addend) {
        const float4 ad = *reinterpret_cast<const float4*>(addend + (long long)c * R + r);
        *reinterpret_cast<float4*>(out2 + base + (long long)c * R + r) = make_float4(v.x + ad.x, v.y + ad.y, v.z + ad.z, v.w + ad.w);
      }
    }
  }
}

// The decoder's cross-attention memory of one feature level, from the NCHW feature map in ONE pass:
//   mem[hw][t][c] = x[t][c][hw] + level_embed[c]                       (...decoder_univs.py:350-355: input_proj(x).flatten(2) + level_embed,
//   key[hw][t][c] = mem[hw][t][c] + (yx[hw][c] + pos_z[t][c])            permuted to [hw, bt, C]; :400-405: with_pos_embed(memory, pos))
// with the 3-D sine position embedding in its separable form (position_encoding.py: pos = yx + pos_z, the rounded sum first,
// as the reference adds `pos` to the memory).  ATen runs this as a broadcast add, two permuted copies and a strided add
// (108 + 122 + 107 us at the 1/8 level of a 720p clip); here the frame's [C, HW] tile is transposed through LDS and both
// outputs are written with 16-byte stores.
__global__ __launch_bounds__(256) void decoder_memory_kernel(const float* __restrict__ x, const float* __restrict__ level_embed,
                                                             const float* __restrict__ yx, const float* __restrict__ pos_z,
                                                             float* __restrict__ mem, float* __restrict__ key, int T, int C, int HW) {
  __shared__ float tile[TR_TILE][TR_TILE + 1];
  const int t = blockIdx.z;
  const long long base = (long long)t * C * HW;
  const int r0 = blockIdx.y * TR_TILE, c0 = blockIdx.x * TR_TILE;   // r = channel, c = pixel
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
  for (int p = 0; p < TR_TILE; p += 16) {
    const int r = r0 + p + ty, c = c0 + 4 * tx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < C && c < HW) v = *reinterpret_cast<const float4*>(x + base + (long long)r * HW + c);   // HW % 4 == 0 (host-checked)
    tile[p + ty][4 * tx + 0] = v.x;
    tile[p + ty][4 * tx + 1] = v.y;
    tile[p + ty][4 * tx + 2] = v.z;
    tile[p + ty][4 * tx + 3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < TR_TILE; p += 16) {
    const int hw = c0 + p + ty, ch = r0 + 4 * tx;
    if (hw < HW && ch < C) {                                                                       // C % 4 == 0 (host-checked)
      const float4 le = *reinterpret_cast<const float4*>(level_embed + ch);
      const float4 a = *reinterpret_cast<const float4*>(yx + (long long)hw * C + ch);
      const float4 z = *reinterpret_cast<const float4*>(pos_z + (long long)t * C + ch);
      const float4 m = make_float4(tile[4 * tx + 0][p + ty] + le.x, tile[4 * tx + 1][p + ty] + le.y, tile[4 * tx + 2][p + ty] + le.z,
                                   tile[4 * tx + 3][p + ty] + le.w);
      const float4 ps = make_float4(a.x + z.x, a.y + z.y, a.z + z.z, a.w + z.w);                   // pos, rounded as a tensor would be
      const long long o = ((long long)hw * T + t) * C + ch;
      *reinterpret_cast<float4*>(mem + o) = m;
      *reinterpret_cast<float4*>(key + o) = make_float4(m.x + ps.x, m.y + ps.y, m.z + ps.z, m.w + ps.w);
    }
  }
}

int decoder_memory_f32(const float* x, const float* level_embed, const float* yx, const float* pos_z, float* mem, float* key, int T, int C,
                       int HW, hipStream_t st) {
  if (T <= 0 || C <= 0 || HW <= 0) return UNIVS_OK;
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (C % 4 != 0 || HW % 4 != 0 || T > 65535 || mis(x) || mis(level_embed) || mis(yx) || mis(pos_z) || mis(mem) || mis(key))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  dim3 grid((unsigned)((HW + TR_TILE - 1) / TR_TILE), (unsigned)((C + TR_TILE - 1) / TR_TILE), (unsigned)T);
  if (grid.y > 65535) return UNIVS_ERR_NOT_IMPLEMENTED;
  hipLaunchKernelGGL(decoder_memory_kernel, grid, dim3(256), 0, st, x, level_embed, yx, pos_z, mem, key, T, C, HW);
  return check_launch("decoder_memory_f32");
}

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED when R or C is not a multiple of 4 / the pointers are not 16-byte aligned
int transpose_f32(const float* in, float* out, long long B, int R, int C, long long in_bstride, long long out_bstride,
                  const float* row_affine, const float* addend, float* out2, hipStream_t st) {
  if (B <= 0 || R <= 0 || C <= 0) return UNIVS_OK;
  if (in_bstride == 0) in_bstride = (long long)R * C;
  if (out_bstride == 0) out_bstride = (long long)R * C;
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (R % 4 != 0 || C % 4 != 0 || B > 65535 || in_bstride % 4 != 0 || in_bstride < (long long)R * C || out_bstride % 4 != 0 ||
      out_bstride < (long long)R * C || (addend != nullptr) != (out2 != nullptr) || mis(in) || mis(out) || mis(addend) || mis(out2))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  dim3 grid((unsigned)((C + TR_TILE - 1) / TR_TILE), (unsigned)((R + TR_TILE - 1) / TR_TILE), (unsigned)B);
  if (grid.y > 65535) return UNIVS_ERR_NOT_IMPLEMENTED;
  hipLaunchKernelGGL(transpose_f32_kernel, grid, dim3(256), 0, st, in, out, R, C, in_bstride, out_bstride, row_affine, addend, out2);
  return check_launch("transpose_f32");
}

}  // namespace univs

namespace univs {

// Swin PatchEmbed in one pass (mask2former/modeling/backbone/swin.py:307-339): 4 x 4 / stride-4 convolution of the 3-channel
// image + bias, tokens written directly as [T, H/4 * W/4, E] (the reference's `x.flatten(2).transpose(1, 2)`), LayerNorm
// (`patch_norm`) on the token while it is in registers.  The library runs this as layout changes around an implicit-GEMM kernel,
// a bias pass, our tile transpose and a LayerNorm launch (58 + 38 + 38 + 34 + 47 us at 720p x 5 frames).
// Eight lanes share a token: each loads the token's 48 inputs (3 channels x 4 rows of one 16-byte load) and owns E / 8 output
// channels, interleaved in groups of four (4 l + 32 i) so that the eight lanes' 16-byte stores of group i form one 128-byte line;
// W [E][48] sits transposed in LDS ([k][E]: the eight distinct addresses of a read are consecutive 16-byte slots); plain fp32
// FMAs in the k-order (channel, row, column) of the weight tensor; statistics by three xor-shuffles.
constexpr int PE_ROUNDS = 8;
template <int NG>   // E = 32 NG
__global__ __launch_bounds__(256) void patch_embed4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, const float* __restrict__ ln_g,
                                                           const float* __restrict__ ln_b, float eps, float* __restrict__ out, int T,
                                                           int H, int W) {
  constexpr int E = 32 * NG;
  __shared__ __attribute__((aligned(16))) float wl[48 * E];
  for (int i = threadIdx.x; i < 48 * E; i += 256) {
    const int k = i / E, e = i - k * E;
    wl[i] = w[e * 48 + k];
  }
  __syncthreads();
  const int Ht = H >> 2, Wt = W >> 2;
  const long long ntok = (long long)T * Ht * Wt;
  const int l8 = threadIdx.x & 7;
#pragma unroll 1
  for (int it = 0; it < PE_ROUNDS; ++it) {                       // the staged weights serve PE_ROUNDS x 32 tokens
  const long long tok = ((long long)blockIdx.x * PE_ROUNDS + it) * 32 + (threadIdx.x >> 3);
  if (tok >= ntok) return;                                       // (whole 8-lane groups leave together; no barrier follows)
  const int t = (int)(tok / ((long long)Ht * Wt));
  const int rem = (int)(tok - (long long)t * Ht * Wt);
  const int py = rem / Wt, px = rem - py * Wt;
  float4 in[12];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      in[c * 4 + r] = *reinterpret_cast<const float4*>(x + (((long long)t * 3 + c) * H + 4 * py + r) * W + 4 * px);
  float4 acc[NG];
#pragma unroll
  for (int i = 0; i < NG; ++i) acc[i] = bias ? *reinterpret_cast<const float4*>(bias + 4 * l8 + 32 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const float xin[4] = {in[q].x, in[q].y, in[q].z, in[q].w};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const float xv = xin[cc];
      const float* wr = wl + (q * 4 + cc) * E + 4 * l8;
#pragma unroll
      for (int i = 0; i < NG; ++i) {
        const float4 wv = *reinterpret_cast<const float4*>(wr + 32 * i);
        acc[i].x = fmaf(xv, wv.x, acc[i].x);
        acc[i].y = fmaf(xv, wv.y, acc[i].y);
        acc[i].z = fmaf(xv, wv.z, acc[i].z);
        acc[i].w = fmaf(xv, wv.w, acc[i].w);
      }
    }
  }
  if (ln_g) {   // LayerNorm over the token's E channels: two-pass statistics, the eight lanes of the token reduced by xor-shuffles
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i)   // (single adds: hipcc's packed horizontal sum is the form common.h describes)
      sm = add_single(sm, add_single(add_single(acc[i].x, acc[i].y), add_single(acc[i].z, acc[i].w)));
    sm += __shfl_xor(sm, 1, 64);
    sm += __shfl_xor(sm, 2, 64);
    sm += __shfl_xor(sm, 4, 64);
    const float mean = sm * (1.0f / E);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      acc[i].x -= mean; acc[i].y -= mean; acc[i].z -= mean; acc[i].w -= mean;
      sq = fmaf(acc[i].x, acc[i].x, sq); sq = fmaf(acc[i].y, acc[i].y, sq); sq = fmaf(acc[i].z, acc[i].z, sq); sq = fmaf(acc[i].w, acc[i].w, sq);
    }
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    sq += __shfl_xor(sq, 4, 64);
    const float rstd = 1.0f / sqrtf(sq * (1.0f / E) + eps);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      const float4 gm = *reinterpret_cast<const float4*>(ln_g + 4 * l8 + 32 * i);
      const float4 bt = ln_b ? *reinterpret_cast<const float4*>(ln_b + 4 * l8 + 32 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
      acc[i] = make_float4(acc[i].x * rstd * gm.x + bt.x, acc[i].y * rstd * gm.y + bt.y, acc[i].z * rstd * gm.z + bt.z,
                           acc[i].w * rstd * gm.w + bt.w);
    }
  }
  float* o = out + tok * E + 4 * l8;
#pragma unroll
  for (int i = 0; i < NG; ++i) *reinterpret_cast<float4*>(o + 32 * i) = acc[i];
  }
}

// x [T, 3, H, W] (H, W multiples of 4), w [E, 3, 4, 4], out [T, H/4 * W/4, E].  UNIVS_ERR_NOT_IMPLEMENTED: E not 96 / 128 / 192,
// H or W not a multiple of 4, pointers not 16-byte aligned.
int patch_embed4_f32(const float* x, const float* w, const float* bias, const float* ln_g, const float* ln_b, float eps, float* out, int T,
                     int H, int W, int E, hipStream_t st) {
  if (T <= 0 || H <= 0 || W <= 0) return UNIVS_OK;
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (H % 4 != 0 || W % 4 != 0 || mis(x) || mis(w) || mis(bias) || mis(ln_g) || mis(ln_b) || mis(out) || (ln_b && !ln_g))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  const long long ntok = (long long)T * (H / 4) * (W / 4);
  const long long nblk = (ntok + 32 * PE_ROUNDS - 1) / (32 * PE_ROUNDS);
  if (nblk > 0x7fffffffLL) return UNIVS_ERR_NOT_IMPLEMENTED;
  dim3 grid((unsigned)nblk), block(256);
  switch (E) {
    case 96: hipLaunchKernelGGL(patch_embed4_kernel<3>, grid, block, 0, st, x, w, bias, ln_g, ln_b, eps, out, T, H, W); break;
    case 128: hipLaunchKernelGGL(patch_embed4_kernel<4>, grid, block, 0, st, x, w, bias, ln_g, ln_b, eps, out, T, H, W); break;
    case 192: hipLaunchKernelGGL(patch_embed4_kernel<6>, grid, block, 0, st, x, w, bias, ln_g, ln_b, eps, out, T, H, W); break;
    default: return UNIVS_ERR_NOT_IMPLEMENTED;
  }
  return check_launch("patch_embed4_f32");
}

}  // namespace univs
