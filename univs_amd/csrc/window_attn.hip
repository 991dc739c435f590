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
#include "config.h"
#include "window_attn.h"

#include <algorithm>
#include <cstdlib>

namespace univs {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WA_WAVES = 4;
constexpr int WA_VSTRIDE = 36;  // floats per staged V row (32 + 4 pad): groups g, g+1 hit disjoint banks

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

// ---- image mode, 7 x 7 windows (NB = 4), second version.
// What bounded the first one (above, still used for 12 x 12 windows and the windowed layout): not the matrix cores
// (256 f32 MFMAs = 8.2k clocks per (window, head)) and not HBM (qkv is read once) but everything around them, ~34k
// clocks per pair at the 184 x 320 stage: the token index of every row recomputed with 64-bit divisions in every lane
// (the wave's (window, head) was not known to be uniform), and the relative-position bias + shift mask fetched as 32
// scattered 4-byte loads per lane and query block.  Here:
//   * a workgroup = 4 waves = ONE head, persistent over a range of windows: the head's bias table sits in LDS once per
//     workgroup (row stride 68 floats: a lane's four consecutive keys are one conflict-free 16-byte read);
//   * the wave's window is scalar (readfirstlane), its coordinates are scalar arithmetic; every lane computes the token
//     offsets of its 4 rows once per window (32-bit, multiply-shift division) and the other row sets (V staging, output)
//     are lane permutations of them;
//   * the shift mask is non-zero only in the last row / column of windows (swin.py:413-440): interior windows skip it.
constexpr int WA2_BSTRIDE = 68;

__global__ __launch_bounds__(64 * WA_WAVES) void window_attn_img7_f32(const float* __restrict__ qkv,
                                                                       const float* __restrict__ qkv_bias,
                                                                       const float* __restrict__ bias,
                                                                       const float* __restrict__ shift_mask, int B_, int nW,
                                                                       int nH, float scale, float* __restrict__ out,
                                                                       WinImage wi, int magic) {
  constexpr int HD = 32, NB = 4, NP = 64, Ntok = 49;
  extern __shared__ __attribute__((aligned(16))) float lds_wa2[];
  float* bias_lds = lds_wa2;                                  // [64][WA2_BSTRIDE]: bias of head h, -inf for keys >= 49
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  float* vlds = lds_wa2 + NP * WA2_BSTRIDE + wave * (NP * WA_VSTRIDE);
  const int h = blockIdx.y;
  const int g = lane >> 4, n = lane & 15;

  // ---- the head's bias table -> LDS (padded: key columns >= 49 are -inf, query rows >= 49 are 0)
  for (int idx = threadIdx.x; idx < NP * NP; idx += 64 * WA_WAVES) {
    const int i = idx >> 6, j = idx & 63;
    float v = j < Ntok ? 0.f : -INFINITY;
    if (i < Ntok && j < Ntok) v = bias[((long long)h * Ntok + i) * Ntok + j];
    bias_lds[i * WA2_BSTRIDE + j] = v;
  }
  __syncthreads();

  const int tok_stride = 3 * nH * HD, part = nH * HD;
  const float* qkvb = qkv_bias ? qkv_bias + h * HD : nullptr;
  const int nWy = wi.Hp / wi.ws;

#pragma unroll 1
  for (int b = blockIdx.x * WA_WAVES + wave; b < B_; b += gridDim.x * WA_WAVES) {   // scalar
    const int w = b % nW, img = b / nW;
    const int wy = w / wi.nWx, wx = w - wy * wi.nWx;
    const int y0 = wy * wi.ws + wi.shift, x0 = wx * wi.ws + wi.shift;
    const bool masked = shift_mask && (wy == nWy - 1 || wx == wi.nWx - 1);   // scalar
    // token (float offset of its qkv row; -1: padding) of my 4 rows j = 16 jb + n
    int tok[NB];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      const int j = jb * 16 + n;
      const int py = (j * magic) >> 16, px = j - py * wi.ws;
      int y = y0 + py, x = x0 + px;
      y -= (y >= wi.Hp) ? wi.Hp : 0;
      x -= (x >= wi.Wp) ? wi.Wp : 0;
      tok[jb] = (j < Ntok && y < wi.H && x < wi.W) ? (img * wi.H + y) * wi.W + x : -1;
    }
    auto row_ptr = [&](int t) __attribute__((always_inline)) -> const float* {
      return t >= 0 ? qkv + (long long)t * tok_stride + h * HD : qkvb;
    };

    // ---- every global load of the window is issued here, back to back (V rows, K and Q fragments: 24 x 16 B per lane):
    // one memory latency per window instead of six (the first version loaded V, waited, K, waited, and each of the four
    // Q fragments at the top of its query block)
    float4 vst[8];
    {
      const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {   // V[j][c4 .. c4+3], j = 8 jj + r8: row j's token sits in lane j & 15 of tok[j >> 4]
        const int j = jj * 8 + r8;
        const int t = __shfl(tok[jj >> 1], (jj & 1) * 8 + r8, 64);
        vst[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < Ntok) {
          const float* rp = row_ptr(t);
          if (rp) vst[jj] = *reinterpret_cast<const float4*>(rp + 2 * part + c4);
        }
      }
    }
    // K fragments: lane holds K[jb*16 + n][8g .. 8g+7]; Q fragments (unscaled) likewise
    float kf[NB][8], qraw[NB][8];
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a, qa = a, qc = a;
      if (jb * 16 + n < Ntok) {
        const float* rp = row_ptr(tok[jb]);
        if (rp) {
          const float4* p = reinterpret_cast<const float4*>(rp + part + 8 * g);
          a = p[0];
          c = p[1];
          const float4* pq = reinterpret_cast<const float4*>(rp + 8 * g);
          qa = pq[0];
          qc = pq[1];
        }
      }
      kf[jb][0] = a.x; kf[jb][1] = a.y; kf[jb][2] = a.z; kf[jb][3] = a.w;
      kf[jb][4] = c.x; kf[jb][5] = c.y; kf[jb][6] = c.z; kf[jb][7] = c.w;
      qraw[jb][0] = qa.x; qraw[jb][1] = qa.y; qraw[jb][2] = qa.z; qraw[jb][3] = qa.w;
      qraw[jb][4] = qc.x; qraw[jb][5] = qc.y; qraw[jb][6] = qc.z; qraw[jb][7] = qc.w;
    }
    {
      const int r8 = lane >> 3, c4 = (lane & 7) * 4;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) *reinterpret_cast<float4*>(vlds + (jj * 8 + r8) * WA_VSTRIDE + c4) = vst[jj];
    }
    const float* mask_w = masked ? shift_mask + (long long)w * Ntok * Ntok : nullptr;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are visible to its own reads
    __builtin_amdgcn_wave_barrier();

#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
      const int i = ib * 16 + n;
      float qf[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) qf[t] = qraw[ib][t] * scale;
      f32x4 s[NB];
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        // bias (and the -inf of the padded keys) is the accumulator's initial value
        f32x4 acc = *reinterpret_cast<const f32x4*>(bias_lds + i * WA2_BSTRIDE + jb * 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[jb][t], qf[t], acc, 0, 0, 0);
        s[jb] = acc;
      }
      if (mask_w) {   // scalar: last row / column of windows only
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // (unconditional loads at clamped indices + a select: with the load behind `if (i < Ntok && j < Ntok)` hipcc
            // 7.2 keeps part of s[] in AGPR copies made only by the lanes that take the branch and restores them in all
            // lanes -- the padded keys of the other lanes then hold stale values instead of -inf)
            const int j = jb * 16 + 4 * g + r;
            const float mv = mask_w[min(i, Ntok - 1) * Ntok + min(j, Ntok - 1)];
            s[jb][r] += (i < Ntok && j < Ntok) ? mv : 0.f;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[jb][r]);
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

      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* vr = vlds + (jb * 16 + 4 * g + r) * WA_VSTRIDE + n;
          o0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s[jb][r], vr[0], o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s[jb][r], vr[16], o1, 0, 0, 0);
        }
      // C layout: row = 4g + r -> query ib*16 + 4g + r (its token: lane 4g + r of tok[ib]); col = n -> head-dim n / 16 + n
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = __shfl(tok[ib], 4 * g + r, 64);
        if (t >= 0) {
          float* op = out + ((long long)t * nH + h) * HD;
          op[n] = o0[r];
          op[16 + n] = o1[r];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // the next window's V staging overwrites vlds
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
  const WinImage wi = win_image(H, W, ws, shift);
  const int nW = (wi.Hp / ws) * wi.nWx;
  const bool v1 = config().window_attn_v1 != 0;   // the first 7x7 kernel (kernel benchmarks)
  if (ws == 7 && hd == 32 && !v1 && (long long)B * H * W * 3 * nH * hd < 0x7FFFFFFFLL) {
    const int B_ = B * nW;
    if (B_ == 0) return UNIVS_OK;
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0, v = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
        (void)hipGetLastError();
        v = 256;
      }
      n_cu = v;
    }
    const size_t lds = (size_t)(64 * WA2_BSTRIDE + WA_WAVES * 64 * WA_VSTRIDE) * sizeof(float);   // 54 KB: 2-3 workgroups per CU
    int gx = std::max(1, (3 * n_cu + nH - 1) / nH);
    gx = std::min(gx, (B_ + WA_WAVES - 1) / WA_WAVES);
    hipLaunchKernelGGL(window_attn_img7_f32, dim3(gx, nH), dim3(64 * WA_WAVES), lds, st, qkv, qkv_bias, bias, shift_mask, B_, nW,
                       nH, scale, out, wi, (65536 + ws - 1) / ws);
    return check_launch("window_attn_img7_f32");
  }
  return launch_window_attn<true>(qkv, qkv_bias, bias, shift_mask, B * nW, nW, ws * ws, nH, hd, scale, out, wi, st);
}

}  // namespace univs
