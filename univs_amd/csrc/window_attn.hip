// Swin window attention core for gfx950 on the exact-f32 MFMA path.
//
// Reference semantics (mask2former/modeling/backbone/swin.py:137-168, between the qkv and proj
// linears):   attn = softmax((q * scale) @ k^T + rel_pos_bias[h] (+ shift_mask[b % nW])) ;  x = attn @ v
// with qkv laid out [B_, Ntok, 3, nH, hd] as produced by the qkv Linear (the reference permutes it
// to [3, B_, nH, Ntok, hd] with a copy, materialises the [B_, nH, Ntok, Ntok] score tensor, and makes
// 5 more full passes over it: +bias, +mask, softmax, @v, transpose).  Here one wave64 owns one
// (window, head): Q/K fragments go global -> VGPR as 16-B loads, scores stay in registers, V is
// staged once in LDS, and the output is written once -- qkv is read exactly once, nothing else
// touches HBM.
//
// MFMA formulation (v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// C/D[row=(l>>4)*4+r][col=l&15]):
//   * scores are computed TRANSPOSED, S^T = K (Q*scale)^T, so that after the MFMA a lane holds
//     S[i = l&15][j = jb*16 + 4*(l>>4) + r]: the softmax axis j is spread over registers and the four
//     16-lane groups only (reduce = in-lane + 2 shuffles), and the same registers are already in
//     A-operand layout for the P @ V product (no LDS round trip, no transpose);
//   * the reduction index of both products is summed in a permuted order (lane group g takes
//     head-dim 8g..8g+7 / keys 4g..4g+3), which is exact up to fp32 re-association.
#include "common.h"

namespace univs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WA_WAVES = 4;
constexpr int WA_VSTRIDE = 36;  // floats per staged V row (32 + 4 pad): groups g, g+1 hit disjoint banks

// Image mode (IMG): qkv / out are in TOKEN order [B, H*W, ...] and the kernel does the reference's
// pad -> roll(-shift) -> window_partition on the way in and window_reverse -> roll(+shift) -> crop on
// the way out (swin.py:252-284) by index arithmetic: window b = (image, wy, wx), position j = (py, px)
// maps to pixel ((wy*ws + py + shift) mod Hp, (wx*ws + px + shift) mod Wp); pixels beyond (H, W) are the
// zero padding, whose q/k/v are the qkv Linear's bias (Linear(0) = bias) and which are not written back.
struct WinImage {
  int H, W, ws, shift, nWx, Hp, Wp;
};

template <bool IMG>
__device__ __forceinline__ long long win_token(const WinImage& wi, long long b, int nW, int Ntok, int j) {
  if (!IMG) return b * Ntok + j;
  const int w = (int)(b % nW);
  const long long img = b / nW;
  const int wy = w / wi.nWx, wx = w - wy * wi.nWx;
  const int py = j / wi.ws, px = j - py * wi.ws;
  int y = wy * wi.ws + py + wi.shift, x = wx * wi.ws + px + wi.shift;
  y -= (y >= wi.Hp) ? wi.Hp : 0;
  x -= (x >= wi.Wp) ? wi.Wp : 0;
  return (y < wi.H && x < wi.W) ? (img * wi.H + y) * wi.W + x : -1;
}

template <int NB, bool IMG>  // NB 16-token blocks: Ntok <= 16*NB
__global__ __launch_bounds__(64 * WA_WAVES) void window_attn_f32(const float* __restrict__ qkv,
                                                                  const float* __restrict__ qkv_bias,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ shift_mask,
                                                                  int B_, int nW, int Ntok, int nH,
                                                                  float scale, float* __restrict__ out,
                                                                  long long npairs, WinImage wi) {
  constexpr int HD = 32;
  constexpr int NP = 16 * NB;
  extern __shared__ __attribute__((aligned(16))) float vlds_all[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long pair = (long long)blockIdx.x * WA_WAVES + wave;  // (window b, head h), h fastest
  if (pair >= npairs) return;  // wave-uniform; no block-level barrier is used below
  const int h = (int)(pair % nH);
  const long long b = pair / nH;
  const int g = lane >> 4, n = lane & 15;
  float* vlds = vlds_all + wave * (NP * WA_VSTRIDE);

  const long long tok_stride = 3LL * nH * HD;
  const long long part = (long long)nH * HD;   // q | k | v parts of a token row
  // row of (window b, position j): its q part for head h; padding pixels read the qkv bias (or zeros)
  auto row = [&](int j) __attribute__((always_inline)) -> const float* {
    const long long t = win_token<IMG>(wi, b, nW, Ntok, j);
    if (t >= 0) return qkv + t * tok_stride + (long long)h * HD;
    return qkv_bias ? qkv_bias + (long long)h * HD : nullptr;
  };

  // ---- stage V[j][0..31] into LDS (8 lanes x 16 B per row, 8 rows per pass)
  {
    const int r8 = lane >> 3, c4 = (lane & 7) * 4;
    for (int j0 = 0; j0 < NP; j0 += 8) {
      const int j = j0 + r8;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < Ntok) {
        const float* rp = row(j);
        if (rp) v = *reinterpret_cast<const float4*>(rp + 2 * part + c4);
      }
      *reinterpret_cast<float4*>(vlds + j * WA_VSTRIDE + c4) = v;
    }
  }

  // ---- K fragments for all key blocks: lane holds K[jb*16 + n][8g .. 8g+7]
  float kf[NB][8];
#pragma unroll
  for (int jb = 0; jb < NB; ++jb) {
    const int j = jb * 16 + n;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (j < Ntok) {
      const float* rp = row(j);
      if (rp) {
        const float4* p = reinterpret_cast<const float4*>(rp + part + 8 * g);
        a = p[0];
        c = p[1];
      }
    }
    kf[jb][0] = a.x; kf[jb][1] = a.y; kf[jb][2] = a.z; kf[jb][3] = a.w;
    kf[jb][4] = c.x; kf[jb][5] = c.y; kf[jb][6] = c.z; kf[jb][7] = c.w;
  }
  const float* bias_h = bias + (long long)h * Ntok * Ntok;
  const float* mask_w = shift_mask ? shift_mask + (long long)(b % nW) * Ntok * Ntok : nullptr;

  // make this wave's LDS writes visible to its own later reads (single-wave scope)
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();

#pragma unroll 1
  for (int ib = 0; ib < NB; ++ib) {
    const int i = ib * 16 + n;  // this lane's query (column of S^T)
    if (ib * 16 >= Ntok) break;
    // Q fragment (scaled): Q[i][8g .. 8g+7]
    float qf[8];
    {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
      if (i < Ntok) {
        const float* rp = row(i);
        if (rp) {
          const float4* p = reinterpret_cast<const float4*>(rp + 8 * g);
          a = p[0];
          c = p[1];
        }
      }
      qf[0] = a.x * scale; qf[1] = a.y * scale; qf[2] = a.z * scale; qf[3] = a.w * scale;
      qf[4] = c.x * scale; qf[5] = c.y * scale; qf[6] = c.z * scale; qf[7] = c.w * scale;
    }
    // S^T blocks
    f32x4 s[NB];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jb][t], qf[t], acc, 0, 0, 0);
      s[jb] = acc;
    }
    // + bias (+ shift mask); padded keys -> -inf
    float mx = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jb * 16 + 4 * g + r;
        float v = -INFINITY;
        if (j < Ntok) {
          v = s[jb][r];
          if (i < Ntok) {
            v += bias_h[(long long)i * Ntok + j];
            if (mask_w) v += mask_w[(long long)i * Ntok + j];
          }
        }
        s[jb][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __expf(s[jb][r] - mx);  // exp(-inf) = 0 for padded keys
        s[jb][r] = e;
        sum += e;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) s[jb][r] *= inv;

    // out[ib] (16 x 32) = P[ib, :] @ V : two 16-column blocks
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jb = 0; jb < NB; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vr = vlds + (jb * 16 + 4 * g + r) * WA_VSTRIDE + n;
        o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s[jb][r], vr[0], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s[jb][r], vr[16], o1, 0, 0, 0);
      }
    // C layout: row = 4g + r -> query ib*16 + 4g + r ; col = n -> head-dim n (o0) / 16 + n (o1)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int io = ib * 16 + 4 * g + r;
      if (io < Ntok) {
        const long long t = win_token<IMG>(wi, b, nW, Ntok, io);
        if (t >= 0) {
          float* op = out + (t * nH + h) * HD;
          op[n] = o0[r];
          op[16 + n] = o1[r];
        }
      }
    }
  }
}

template <bool IMG>
static int launch_window_attn(const float* qkv, const float* qkv_bias, const float* bias, const float* shift_mask,
                              int B_, int nW, int Ntok, int nH, int hd, float scale, float* out, const WinImage& wi,
                              hipStream_t st) {
  if (hd != 32) {
    set_error("window_attention_f32: head_dim=%d (only 32, the Swin-T/B/L value)", hd);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const long long npairs = (long long)B_ * nH;
  if (npairs == 0) return UNIVS_OK;
  const unsigned nblocks = (unsigned)((npairs + WA_WAVES - 1) / WA_WAVES);
  if (Ntok <= 64) {
    const size_t lds = (size_t)WA_WAVES * 64 * WA_VSTRIDE * sizeof(float);
    hipLaunchKernelGGL((window_attn_f32<4, IMG>), dim3(nblocks), dim3(64 * WA_WAVES), lds, st, qkv, qkv_bias, bias,
                       shift_mask, B_, nW, Ntok, nH, scale, out, npairs, wi);
  } else if (Ntok <= 144) {
    const size_t lds = (size_t)WA_WAVES * 144 * WA_VSTRIDE * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_attn_f32<9, IMG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((window_attn_f32<9, IMG>), dim3(nblocks), dim3(64 * WA_WAVES), lds, st, qkv, qkv_bias, bias,
                       shift_mask, B_, nW, Ntok, nH, scale, out, npairs, wi);
  } else {
    set_error("window_attention_f32: %d tokens per window (max 144 = 12x12)", Ntok);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  return check_launch("window_attn_f32");
}

int window_attention_f32(const float* qkv, const float* bias, const float* shift_mask, int B_, int nW,
                         int Ntok, int nH, int hd, float scale, float* out, hipStream_t st) {
  WinImage wi{};
  return launch_window_attn<false>(qkv, nullptr, bias, shift_mask, B_, nW, Ntok, nH, hd, scale, out, wi, st);
}

// qkv [B, H*W, 3, nH, hd] in token order; out [B, H*W, nH*hd]
int window_attention_image_f32(const float* qkv, const float* qkv_bias, const float* bias, const float* shift_mask,
                               int B, int H, int W, int ws, int shift, int nH, int hd, float scale, float* out,
                               hipStream_t st) {
  WinImage wi;
  wi.H = H; wi.W = W; wi.ws = ws; wi.shift = shift;
  wi.Hp = (H + ws - 1) / ws * ws;
  wi.Wp = (W + ws - 1) / ws * ws;
  wi.nWx = wi.Wp / ws;
  const int nW = (wi.Hp / ws) * wi.nWx;
  return launch_window_attn<true>(qkv, qkv_bias, bias, shift_mask, B * nW, nW, ws * ws, nH, hd, scale, out, wi, st);
}

}  // namespace univs
