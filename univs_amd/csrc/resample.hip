// Bilinear plane resampling (align_corners = false), gfx950.
//
// Replaces the F.interpolate(mask_features, size, mode="bilinear", align_corners=False) calls that feed
// the attention-mask heads (reference: video_mask2former_transformer_decoder_univs.py:555-558 resizes the
// mask logits of every decoder layer; this build resamples the mask FEATURES once per level and
// contracts at the target resolution -- bilinear resampling commutes with the channel contraction).
// The optional `addend` makes it the FPN top-down step `lateral + upsample(coarser)`
// (mask2former/modeling/pixel_decoder/msdeformattn.py:350-351), which ATen runs as a channels-last
// upsample followed by a mixed-layout add (1.3 ms at 184x320).
// ATen's upsample_bilinear2d_out_frame runs this at ~220 GB/s on the [T*256, 184, 320] planes
// (1.35 ms per level); it is a pure HBM-bound gather: one thread per 4 output pixels, 16 taps, one
// 16-B store.
//
// Arithmetic follows ATen's kernel term by term (UpSampleBilinear2d.cu: area_pixel_compute_source_index,
// h1p / w1p edge handling, the lambda products) so that results agree to rounding.
#include "common.h"

namespace univs {

struct Tap {
  int i0, di;      // first tap, +1 or +0 (last row / column)
  float l0, l1;    // weights
};

__device__ __forceinline__ Tap make_tap(float scale, int dst, int in_size) {
  float src = scale * ((float)dst + 0.5f) - 0.5f;
  src = src < 0.f ? 0.f : src;
  Tap t;
  t.i0 = (int)src;
  t.di = (t.i0 < in_size - 1) ? 1 : 0;
  t.l1 = src - (float)t.i0;
  t.l0 = 1.f - t.l1;
  return t;
}

// l0y (l0x a + l1x b) + l1y (l0x c + l1x d) with the roundings spelled out (three products, three fused multiply-adds): left to the
// compiler's contraction the two kernels below rounded the same expression differently
__device__ __forceinline__ float bilerp(const Tap& ty, const Tap& tx, float a, float b, float c, float d) {
  const float top = fmaf(tx.l1, b, tx.l0 * a);
  const float bot = fmaf(tx.l1, d, tx.l0 * c);
  return fmaf(ty.l1, bot, ty.l0 * top);
}

template <int VEC>
__global__ __launch_bounds__(256) void bilinear_resample_f32_kernel(const float* __restrict__ in,
                                                                    const float* __restrict__ addend,
                                                                    float* __restrict__ out, int Hin, int Win,
                                                                    int Hout, int Wout, float rh, float rw,
                                                                    long long planes) {
  const int wq = Wout / VEC;                                  // VEC-wide groups per output row
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Hout * wq) return;
  const int oy = q / wq, ox = (q - oy * wq) * VEC;
  const Tap ty = make_tap(rh, oy, Hin);
  Tap tx[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) tx[v] = make_tap(rw, ox + v, Win);
  for (long long p = blockIdx.y; p < planes; p += gridDim.y) {
    const float* r0 = in + (p * Hin + ty.i0) * (long long)Win;
    const float* r1 = r0 + (long long)ty.di * Win;
    float a[VEC], b[VEC], c[VEC], d[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      a[v] = r0[tx[v].i0];
      b[v] = r0[tx[v].i0 + tx[v].di];
      c[v] = r1[tx[v].i0];
      d[v] = r1[tx[v].i0 + tx[v].di];
    }
    float o[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = bilerp(ty, tx[v], a[v], b[v], c[v], d[v]);
    const long long off = (p * Hout + oy) * (long long)Wout + ox;
    if (addend) {   // FPN top-down path: lateral + upsampled coarser level in one pass
#pragma unroll
      for (int v = 0; v < VEC; ++v) o[v] = addend[off + v] + o[v];
    }
    float* dst = out + off;
    if (VEC == 4) {
      *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) dst[v] = o[v];
    }
  }
}

// ---- the FPN top-down step for an exact 2x up-sampling: out = (addend [* scale_p + bias_p]) + interpolate(in) with Hout = 2 Hin,
// Wout = 2 Win (msdeformattn.py:350-351: `y = cur_fpn + F.interpolate(y, size=cur_fpn.shape[-2:], mode="bilinear")`).  Same taps,
// weights and expression as the kernel above (bit-identical), different data movement: the four outputs of a thread read input
// columns 2 X - 1 ... 2 X + 2 of two rows -- the thread loads (2 X, 2 X + 1) with ONE 8-byte load per row and takes the outer two
// from its neighbour lanes (they hold the adjacent column groups of the same row), the addend with one 16-byte load: 4 memory
// instructions per 16 output bytes instead of 21.  `affine` [planes][2] (optional): the addend is a convolution output whose
// GroupNorm is applied on the way in (group_norm.hip: gn_affine_kernel wrote scale = gamma / sqrt(var + eps), bias = beta - mean *
// scale per plane; `x * scale + bias` is gn_apply_kernel's expression) -- the normalised tensor is never written.
__global__ __launch_bounds__(256) void upsample2x_add_kernel(const float* __restrict__ in, const float* __restrict__ addend,
                                                             const float* __restrict__ affine, float* __restrict__ out, int Hin,
                                                             int Win, long long planes) {
  const int Wout = 2 * Win, Hout = 2 * Hin, wq = Wout / 4;
  const int total = Hout * wq;
  const int q0 = blockIdx.x * 256 + threadIdx.x;
  const bool active = q0 < total;
  const int q = active ? q0 : total - 1;                       // idle threads follow along (the lane exchanges are wave-wide)
  const int oy = q / wq, X = q - oy * wq, ox = 4 * X;
  const Tap ty = make_tap(0.5f, oy, Hin);
  Tap tx[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) tx[v] = make_tap(0.5f, ox + v, Win);
  const int lane = threadIdx.x & 63;
  const bool left_ok = lane > 0 && X > 0;                        // lane - 1 holds (oy, X - 1)
  const bool right_ok = lane < 63 && X + 1 < wq && q0 + 1 < total;   // lane + 1 holds (oy, X + 1)
  const int cl = max(2 * X - 1, 0), cr = min(2 * X + 2, Win - 1);
  auto pick = [&](float a0, float a1, float a2, float a3, int col) __attribute__((always_inline)) {
    const int rel = col - (2 * X - 1);
    return rel == 0 ? a0 : rel == 1 ? a1 : rel == 2 ? a2 : a3;
  };
  for (long long p = blockIdx.y; p < planes; p += gridDim.y) {
    const float* r0 = in + (p * Hin + ty.i0) * (long long)Win;
    const float* r1 = r0 + (long long)ty.di * Win;
    const float2 m0 = *reinterpret_cast<const float2*>(r0 + 2 * X), m1 = *reinterpret_cast<const float2*>(r1 + 2 * X);
    float e0 = __shfl_up(m0.y, 1, 64), e1 = __shfl_up(m1.y, 1, 64);
    float f0 = __shfl_down(m0.x, 1, 64), f1 = __shfl_down(m1.x, 1, 64);
    if (!left_ok) { e0 = r0[cl]; e1 = r1[cl]; }
    if (!right_ok) { f0 = r0[cr]; f1 = r1[cr]; }
    float o[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const float a = pick(e0, m0.x, m0.y, f0, tx[v].i0), b = pick(e0, m0.x, m0.y, f0, tx[v].i0 + tx[v].di);
      const float c = pick(e1, m1.x, m1.y, f1, tx[v].i0), d = pick(e1, m1.x, m1.y, f1, tx[v].i0 + tx[v].di);
      o[v] = bilerp(ty, tx[v], a, b, c, d);
    }
    const long long off = (p * Hout + oy) * (long long)Wout + ox;
    typedef float v4f_ __attribute__((ext_vector_type(4)));
    v4f_ ad = *reinterpret_cast<const v4f_*>(addend + off);
    if (affine) {
      const float scale = affine[2 * p], bias = affine[2 * p + 1];
      // as gn_apply_kernel; one fused multiply-add per component, kept out of the packed form (common.h: fma_single)
      ad = (v4f_){fma_single(ad.x, scale, bias), fma_single(ad.y, scale, bias), fma_single(ad.z, scale, bias), fma_single(ad.w, scale, bias)};
    }
    if (active) *reinterpret_cast<v4f_*>(out + off) = (v4f_){ad.x + o[0], ad.y + o[1], ad.z + o[2], ad.w + o[3]};
  }
}

// ---- the three attention-mask resolutions of the decoder (1/2, 1/4, 1/8 of the mask-feature resolution) in ONE pass.
// For an exact 2x / 4x / 8x reduction the taps of make_tap() are fixed: source index s * (dst + 0.5) - 0.5 = s * dst + s / 2 -
// 0.5, i.e. rows / columns (2 d, 2 d + 1), (4 d + 1, 4 d + 2), (8 d + 3, 8 d + 4) with weights (0.5, 0.5) -- every output is the
// same expression as in the kernel above (the multiplications by 0.5 are exact, so the results are bit-identical to three
// separate calls).  One thread owns an 8 x 8 block of the input (sixteen 16-byte loads) and writes its 4 x 4, 2 x 2 and 1 x 1
// outputs: the 301-MB feature tensor of a config-2 clip is read once instead of three times.
__global__ __launch_bounds__(256) void bilinear_pyramid3_f32_kernel(const float* __restrict__ in, float* __restrict__ out2,
                                                                    float* __restrict__ out4, float* __restrict__ out8, int H,
                                                                    int W, long long planes) {
  const int bw = W >> 3, bh = H >> 3;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= bh * bw) return;
  const int by = q / bw, bx = q - by * bw;
  auto avg = [](float a, float b, float c, float d) __attribute__((always_inline)) {
    return 0.5f * (0.5f * a + 0.5f * b) + 0.5f * (0.5f * c + 0.5f * d);      // l0 * (l0 a + l1 b) + l1 * (l0 c + l1 d)
  };
  for (long long p = blockIdx.y; p < planes; p += gridDim.y) {
    const float* src = in + (p * H + 8 * by) * (long long)W + 8 * bx;
    float r[8][8];
#pragma unroll
    for (int y = 0; y < 8; ++y) {
      const float4 u = *reinterpret_cast<const float4*>(src + (long long)y * W);
      const float4 v = *reinterpret_cast<const float4*>(src + (long long)y * W + 4);
      r[y][0] = u.x; r[y][1] = u.y; r[y][2] = u.z; r[y][3] = u.w;
      r[y][4] = v.x; r[y][5] = v.y; r[y][6] = v.z; r[y][7] = v.w;
    }
    float* d2 = out2 + (p * (H >> 1) + 4 * by) * (long long)(W >> 1) + 4 * bx;
#pragma unroll
    for (int y = 0; y < 4; ++y)
      *reinterpret_cast<float4*>(d2 + (long long)y * (W >> 1)) =
          make_float4(avg(r[2 * y][0], r[2 * y][1], r[2 * y + 1][0], r[2 * y + 1][1]), avg(r[2 * y][2], r[2 * y][3], r[2 * y + 1][2], r[2 * y + 1][3]),
                      avg(r[2 * y][4], r[2 * y][5], r[2 * y + 1][4], r[2 * y + 1][5]), avg(r[2 * y][6], r[2 * y][7], r[2 * y + 1][6], r[2 * y + 1][7]));
    float* d4 = out4 + (p * (H >> 2) + 2 * by) * (long long)(W >> 2) + 2 * bx;
#pragma unroll
    for (int y = 0; y < 2; ++y)
      *reinterpret_cast<float2*>(d4 + (long long)y * (W >> 2)) =
          make_float2(avg(r[4 * y + 1][1], r[4 * y + 1][2], r[4 * y + 2][1], r[4 * y + 2][2]),
                      avg(r[4 * y + 1][5], r[4 * y + 1][6], r[4 * y + 2][5], r[4 * y + 2][6]));
    out8[(p * bh + by) * (long long)bw + bx] = avg(r[3][3], r[3][4], r[4][3], r[4][4]);
  }
}

// H and W multiples of 8 (and W / 2 a multiple of 4 for the 16-byte stores): UNIVS_ERR_NOT_IMPLEMENTED otherwise
int bilinear_pyramid3_f32(const float* in, float* out2, float* out4, float* out8, long long planes, int H, int W, hipStream_t st) {
  if (H % 8 != 0 || W % 8 != 0 || (reinterpret_cast<uintptr_t>(in) & 15) || (reinterpret_cast<uintptr_t>(out2) & 15) ||
      (reinterpret_cast<uintptr_t>(out4) & 7))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  const unsigned gy = (unsigned)(planes < 65535 ? planes : 65535);
  const unsigned gx = (unsigned)(((long long)(H >> 3) * (W >> 3) + 255) / 256);
  hipLaunchKernelGGL(bilinear_pyramid3_f32_kernel, dim3(gx, gy), dim3(256), 0, st, in, out2, out4, out8, H, W, planes);
  return check_launch("bilinear_pyramid3_f32");
}

// returns UNIVS_ERR_NOT_IMPLEMENTED unless Win is even and the pointers are aligned (in: 8 bytes; addend, out: 16; affine: 4)
int upsample2x_add_f32(const float* in, const float* addend, const float* affine, float* out, long long planes, int Hin, int Win,
                       hipStream_t st) {
  if (Win % 2 != 0 || Win < 2 || Hin < 1 || (reinterpret_cast<uintptr_t>(in) & 7) || (reinterpret_cast<uintptr_t>(addend) & 15) ||
      (reinterpret_cast<uintptr_t>(out) & 15) || (long long)Hin * Win >= (1LL << 28))
    return UNIVS_ERR_NOT_IMPLEMENTED;
  const unsigned gy = (unsigned)(planes < 65535 ? planes : 65535);
  const unsigned gx = (unsigned)(((long long)(2 * Hin) * (Win / 2) + 255) / 256);
  hipLaunchKernelGGL(upsample2x_add_kernel, dim3(gx, gy), dim3(256), 0, st, in, addend, affine, out, Hin, Win, planes);
  return check_launch("upsample2x_add_f32");
}

int bilinear_resample_f32(const float* in, const float* addend, float* out, long long planes, int Hin, int Win,
                          int Hout, int Wout, hipStream_t st) {
  const float rh = (float)Hin / (float)Hout, rw = (float)Win / (float)Wout;
  const unsigned gy = (unsigned)(planes < 65535 ? planes : 65535);
  const bool vec4 = (Wout % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  if (vec4) {
    const unsigned gx = (unsigned)(((long long)Hout * (Wout / 4) + 255) / 256);
    hipLaunchKernelGGL(bilinear_resample_f32_kernel<4>, dim3(gx, gy), dim3(256), 0, st, in, addend, out, Hin, Win, Hout, Wout,
                       rh, rw, planes);
  } else {
    const unsigned gx = (unsigned)(((long long)Hout * Wout + 255) / 256);
    hipLaunchKernelGGL(bilinear_resample_f32_kernel<1>, dim3(gx, gy), dim3(256), 0, st, in, addend, out, Hin, Win, Hout, Wout,
                       rh, rw, planes);
  }
  return check_launch("bilinear_resample_f32");
}

// ---- (x - mean[c]) / std[c], zero-padded to [Hp, Wp]: the caller's pre-step of every clip (univs/inference/inference_video_entity.py:
// 246-250 `self.normalizer(...)`, `ImageList.from_tensors(images_norm, self.size_divisibility)`) -- in ATen a subtraction, a
// division, a fill and a strided copy (four passes, 75 us per 720p clip).  One pass: a thread writes four output pixels; the
// arithmetic is ATen's (an fp32 subtraction, then a true division).
__global__ __launch_bounds__(256) void normalize_pad_kernel(const float* __restrict__ in, float* __restrict__ out, int C, int H, int W, int Hp,
                                                            int Wp, float m0, float m1, float m2, float s0, float s1, float s2,
                                                            const float* __restrict__ mean, const float* __restrict__ stdv) {
  const int wq = (Wp + 3) >> 2;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Hp * wq) return;
  const int y = q / wq, x0 = (q - y * wq) * 4;
  const long long plane = blockIdx.y;                            // t * C + c
  const int c = (int)(plane % C);
  const float m = mean ? mean[c] : (c == 0 ? m0 : c == 1 ? m1 : m2);
  const float sd = stdv ? stdv[c] : (c == 0 ? s0 : c == 1 ? s1 : s2);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (y < H) {
    const float* src = in + (plane * H + y) * (long long)W + x0;
    if (x0 + 3 < W && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
      const float4 t = *reinterpret_cast<const float4*>(src);
      v[0] = (t.x - m) / sd; v[1] = (t.y - m) / sd; v[2] = (t.z - m) / sd; v[3] = (t.w - m) / sd;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (x0 + e < W) v[e] = (src[e] - m) / sd;
    }
  }
  float* dst = out + (plane * Hp + y) * (long long)Wp + x0;
  if (x0 + 3 < Wp && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (x0 + e < Wp) dst[e] = v[e];
  }
}

int normalize_pad_f32(const float* in, float* out, long long T, int C, int H, int W, int Hp, int Wp, const float* mean, const float* stdv,
                      hipStream_t st) {
  const long long planes = T * C;
  if (planes <= 0) return UNIVS_OK;
  if (planes > 65535 || Hp < H || Wp < W) return UNIVS_ERR_NOT_IMPLEMENTED;
  const int wq = (Wp + 3) >> 2;
  dim3 grid((unsigned)((Hp * wq + 255) / 256), (unsigned)planes);
  hipLaunchKernelGGL(normalize_pad_kernel, grid, dim3(256), 0, st, in, out, C, H, W, Hp, Wp, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, mean, stdv);
  return check_launch("normalize_pad_f32");
}

}  // namespace univs
