// Swin window attention core with fp16 MFMA operands (BASELINE config 5: the reference runs the backbone under autocast,
// train_net.py:334, so WindowAttention.forward -- mask2former/modeling/backbone/swin.py:131-171 -- multiplies in fp16).
//
// What is fp16 here and what is not (UNIVS_MMA_F16 in include/univs_hip.h):
//   * q * scale, k, v and the un-normalised softmax probabilities exp(s - max) are ROUNDED to fp16 (RNE) as MFMA operands;
//   * both products accumulate in fp32 (v_mfma_f32_16x16x32_f16 for K (Q scale)^T over the 32 head channels in ONE
//     instruction, v_mfma_f32_16x16x16_f16 for P V);
//   * the relative-position bias, the shift mask, the softmax (max, exp2, sum) and the 1/sum normalisation (applied to the
//     fp32 output) are fp32; inputs and outputs are the fp32 tensors of the fp32 path.
// The reference's autocast rounds MORE (qkv and the scores themselves are fp16 tensors there), so this variant sits between
// the fp32 path and the reference's fp16 run; its tolerance against the fp32 result is stated in tests/test_ops_gpu.py.
//
// Organisation (image mode only: tokens in image order, pad / roll / partition / reverse / crop by index arithmetic):
//   * a workgroup = 8 waves = ONE head, persistent over a range of windows; the head's bias table (times log2 e) sits in
//     LDS in fp32 once per workgroup, row stride NP + 4 floats (a lane's four consecutive keys are one conflict-free
//     16-byte read and are the accumulator's initial value);
//   * a wave = one window at a time.  Scores are computed transposed, S^T = K (Q scale)^T, so that a lane holds
//     S[query n][keys jb*16 + 4g .. + 3]: softmax runs over registers and the four 16-lane groups, and the fp16
//     probabilities are already the A operand of P V;
//   * V is staged per wave in LDS as fp16, TRANSPOSED ([head channel][key], row stride NP + 4 halves = 4 x an odd number: the 16
//     lanes of one 8-byte read -- one channel row each -- then start in 16 distinct bank pairs; NP + 8, round 3's stride, is 4 x an even
//     number and rows n and n + 8 met in the same banks: 45 % of the LDS index cycles were conflicts, profiles/r04_pmc_by_kernel_v1.txt
//     reads): the B operand of P V is V[keys 4g .. 4g+3][channel n], four keys of ONE channel per lane.  The transpose is
//     done in registers on the way in: a lane loads 4 keys x 4 channels (four 16-byte loads) and writes four 8-byte rows;
//   * with 12 x 12 windows (config 5) every tile is full: 144 tokens = 9 blocks of 16, no padded key or query exists.
//     LDS: 144 x 148 x 4 (bias) + 8 x 32 x 152 x 2 (V) = 163 072 B of the CU's 163 840: one workgroup per CU.
#include "common.h"
#include "config.h"
#include "f16x3.h"
#include "window_attn.h"

#include <algorithm>

namespace univs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// waves per workgroup: 8; 4 for the three-product kernel on 12 x 12 windows, whose two fp16 planes of V leave room for four
// waves beside the bias table (144 x 148 x 4 + 4 x 2 x 32 x 152 x 2 = 163 072 B of the CU's 163 840)
constexpr int wh_waves(int nb, int terms) { return (terms == 3 && nb > 6) ? 4 : 8; }
#ifndef UNIVS_WH_VT_PAD
#define UNIVS_WH_VT_PAD 4          // (8: round 3's stride, kept for A/B builds -- python -m univs_amd.build --ablate vtpad8)
#endif
constexpr int WH_VT_PAD = UNIVS_WH_VT_PAD;

// TERMS = 1: fp16 operands (UNIVS_MMA_F16).  TERMS = 3: every operand as TWO fp16 parts (h = fp16(x), m = fp16(x - h)) and three
// of the four part products -- fp32-accurate (<= 2^-21.7 per product, see linear_f16x3.hip) at 3/16 of the exact-f32 MFMA time
// (UNIVS_MMA_F16X3).  Range: a window whose k, v or scaled q holds a magnitude >= 2^15 would turn into Inf - Inf = NaN in the
// split, and one whose largest magnitude is below 2^-4 (2^-6 for the scaled q) would lose the second part to fp16's subnormals;
// every wave therefore tests its operands (one max per two elements and two ballots) and, in those rare cases only, brings
// the operand into [2^14, 2^15) by a wave-uniform power of two -- exact, undone on the fp32 scores (bias scaled with them)
// and on the 1/sum normaliser -- so the result stays what the exact-f32 kernel returns for any finite input.
// rare path of the range test: the wave's largest magnitude -> (power of two that brings it into [2^14, 2^15), its inverse),
// the same value in every lane and in scalar registers.  Magnitudes below 2^-40 are scaled as if they were 2^-40 (the product
// of two scales then stays finite; operands that small contribute < 2^-40 to a score or an output in any case).
__device__ __forceinline__ void wh_wave_scale(float mx, float& s, float& inv) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  int e = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 255u) - 127;   // 2^e <= max < 2^(e+1)
  e = max(-40, min(e, 128));
  s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((127 + 14 - e) << 23));
  inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((127 - 14 + e) << 23));
}
// wave-uniform: some lane's largest magnitude is >= 2^15 (the split would overflow: Inf - Inf), or no lane reaches `lo` (the
// second fp16 part of every element would be subnormal: the operand would carry fewer than ~20 bits)
__device__ __forceinline__ bool wh_out_of_range(float mx, float lo) {
  return __builtin_amdgcn_ballot_w64(!(mx < 32768.0f)) != 0 || __builtin_amdgcn_ballot_w64(mx >= lo) == 0;
}

// NWV: waves per workgroup (wh_waves(NB, TERMS) by default; 12 for the three-product kernel on 7 x 7 windows where a launch's
// rounds come out fewer: 156 VGPRs allow three waves per SIMD, 17 + 12 x 8.7 KB of LDS one workgroup per CU)
template <int NB, bool MASK4, int TERMS, int NWV = wh_waves(NB, TERMS)>   // MASK4: ws*ws is a multiple of 4 (mask rows 16-byte aligned)
__global__ __launch_bounds__(64 * NWV) void window_attn_img_f16(const float* __restrict__ qkv,
                                                                      const float* __restrict__ qkv_bias,
                                                                      const float* __restrict__ bias,
                                                                      const float* __restrict__ shift_mask, int B_, int nW,
                                                                      int nH, float scale, float* __restrict__ out,
                                                                      WinImage wi, int magic) {
  constexpr int HD = 32, NP = 16 * NB, BS = NP + 4, VS = NP + WH_VT_PAD, WH_WAVES = NWV;
  constexpr int VR = (NB + 1) / 2;   // V staging rounds: lanes 0-31 take key block r, lanes 32-63 block r + VR
  constexpr float LOG2E = 1.4426950408889634f;
  const int Ntok = wi.ws * wi.ws;
  extern __shared__ __attribute__((aligned(16))) float lds_wh[];
  float* bias_lds = lds_wh;                                   // [NP][BS]: log2e * bias of head h, -inf for keys >= Ntok
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  constexpr int PL = TERMS == 3 ? 2 : 1;                      // planes of V (h, m)
  _Float16* vt = reinterpret_cast<_Float16*>(lds_wh + NP * BS) + wave * (PL * HD * VS);   // [PL][HD][VS]
  const int h = blockIdx.y;
  const int g = lane >> 4, n = lane & 15;

  for (int idx = threadIdx.x; idx < NP * NP; idx += 64 * WH_WAVES) {
    const int i = idx / NP, j = idx - i * NP;
    float v = j < Ntok ? 0.f : -INFINITY;
    if (i < Ntok && j < Ntok) v = bias[((long long)h * Ntok + i) * Ntok + j] * LOG2E;
    bias_lds[i * BS + j] = v;
  }
  __syncthreads();

  const int tok_stride = 3 * nH * HD, part = nH * HD;
  const float* qkvb = qkv_bias ? qkv_bias + h * HD : nullptr;
  const int nWy = wi.Hp / wi.ws;
  const float qscale = scale * LOG2E;    // scores in the exp2 domain
  // V staging roles: key group kgl (4 keys) of the half-wave's block, channel group hg (4 channels)
  const int half = lane >> 5, kgl = (lane >> 3) & 3, hg = lane & 7;

#pragma unroll 1
  for (int b = blockIdx.x * WH_WAVES + wave; b < B_; b += gridDim.x * WH_WAVES) {   // scalar
    const int w = b % nW, img = b / nW;
    const int wy = w / wi.nWx, wx = w - wy * wi.nWx;
    const int y0 = wy * wi.ws + wi.shift, x0 = wx * wi.ws + wi.shift;
    const bool masked = shift_mask && (wy == nWy - 1 || wx == wi.nWx - 1);   // scalar
    // token of row j: >= 0 its index, -1 a zero-padded pixel (q/k/v = the qkv bias), -2 beyond Ntok
    auto token = [&](int j) __attribute__((always_inline)) -> int {
      const int py = (j * magic) >> 16, px = j - py * wi.ws;
      int y = y0 + py, x = x0 + px;
      y -= (y >= wi.Hp) ? wi.Hp : 0;
      x -= (x >= wi.Wp) ? wi.Wp : 0;
      const int t = (y < wi.H && x < wi.W) ? (img * wi.H + y) * wi.W + x : -1;
      return j < Ntok ? t : -2;
    };
    int tok[NB];                     // of my rows j = 16 jb + n
#pragma unroll
    for (int jb = 0; jb < NB; ++jb) tok[jb] = token(jb * 16 + n);
    auto row_ptr = [&](int t) __attribute__((always_inline)) -> const float* {
      return t >= 0 ? qkv + (long long)t * tok_stride + h * HD : (t == -1 ? qkvb : nullptr);
    };

    // ---- V: 4 keys x 4 channels per lane and round, transposed into LDS as fp16
    float4 vraw[VR][4];
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const bool hi_ok = r + VR < NB;                          // compile time
      const int src = hi_ok ? (half ? tok[hi_ok ? r + VR : r] : tok[r]) : tok[r];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t = __shfl(src, half * 32 + kgl * 4 + e, 64);
        vraw[r][e] = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* rp = row_ptr(t);
        if (rp && (hi_ok || half == 0)) vraw[r][e] = *reinterpret_cast<const float4*>(rp + 2 * part + 4 * hg);
      }
    }
    float sv_inv = 1.0f;                                         // TERMS == 3: inverse of the power of two applied to V (rarely != 1)
    if (TERMS == 3) {
      float mv = 0.f;
#pragma unroll
      for (int r = 0; r < VR; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          mv = fmaxf(fmaxf(mv, fmaxf(fabsf(vraw[r][e].x), fabsf(vraw[r][e].y))), fmaxf(fabsf(vraw[r][e].z), fabsf(vraw[r][e].w)));
      if (wh_out_of_range(mv, 0.0625f)) {
        float sv;
        wh_wave_scale(mv, sv, sv_inv);
#pragma unroll
        for (int r = 0; r < VR; ++r)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            vraw[r][e].x *= sv;
            vraw[r][e].y *= sv;
            vraw[r][e].z *= sv;
            vraw[r][e].w *= sv;
          }
      }
    }
#pragma unroll
    for (int r = 0; r < VR; ++r) {
      const bool hi_ok = r + VR < NB;
      const float4(&v)[4] = vraw[r];
      if (hi_ok || half == 0) {
        const int jb = r + half * VR;
        _Float16* dst = vt + (4 * hg) * VS + jb * 16 + kgl * 4;
        const float vv[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                                {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};
#pragma unroll
        for (int c = 0; c < 4; ++c) {                          // channel 4 hg + c: four keys
          f16x4 ch, cm;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            ch[e] = (_Float16)vv[c][e];
            cm[e] = (_Float16)(vv[c][e] - (float)ch[e]);
          }
          *reinterpret_cast<f16x4*>(dst + c * VS) = ch;
          if (TERMS == 3) *reinterpret_cast<f16x4*>(dst + HD * VS + c * VS) = cm;
        }
      }
    }

    // ---- K fragments: lane holds K[jb*16 + n][8g .. 8g+7] as 8 halves (TERMS = 3: two parts)
    f16x8 kf[NB], kfm[TERMS == 3 ? NB : 1];
    float sk = 1.0f, sk_inv = 1.0f;                              // TERMS == 3: power of two applied to K (rarely != 1)
    {
      float kraw[NB][8];
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
        const float* rp = row_ptr(tok[jb]);
        if (rp) {
          const float4* p = reinterpret_cast<const float4*>(rp + part + 8 * g);
          a = p[0];
          c = p[1];
        }
        const float kk[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) kraw[jb][e] = kk[e];
      }
      if (TERMS == 3) {
        float mk = 0.f;
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
          for (int e = 0; e < 8; e += 2) mk = fmaxf(mk, fmaxf(fabsf(kraw[jb][e]), fabsf(kraw[jb][e + 1])));
        if (wh_out_of_range(mk, 0.0625f)) {
          wh_wave_scale(mk, sk, sk_inv);
#pragma unroll
          for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int e = 0; e < 8; ++e) kraw[jb][e] *= sk;
        }
      }
#pragma unroll
      for (int jb = 0; jb < NB; ++jb)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          kf[jb][e] = (_Float16)kraw[jb][e];
          if (TERMS == 3) kfm[jb][e] = (_Float16)(kraw[jb][e] - (float)kf[jb][e]);
        }
    }
    // Q of the first query block (the loop below requests block ib + 1 while it computes block ib)
    float4 qa = make_float4(0.f, 0.f, 0.f, 0.f), qc = qa;
    {
      const float* rp = row_ptr(tok[0]);
      if (rp) {
        const float4* pq = reinterpret_cast<const float4*>(rp + 8 * g);
        qa = pq[0];
        qc = pq[1];
      }
    }
    const float* mask_w = masked ? shift_mask + (long long)w * Ntok * Ntok : nullptr;
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes are visible to its own reads
    __builtin_amdgcn_wave_barrier();

    // the query loop stays rolled: unrolled, hipcc hoists the 81 bias reads and the addresses of all nine blocks to the
    // top and spills 600 registers
#pragma unroll 1
    for (int ib = 0; ib < NB; ++ib) {
      const int i = ib * 16 + n;        // this lane's query (column of S^T)
      const int tok_i = token(i);
      f16x8 qf, qfm;
      float ss = 1.0f, ss_inv = 1.0f;                             // wave-uniform (scalar registers)
      {
        float qq[8] = {qa.x * qscale, qa.y * qscale, qa.z * qscale, qa.w * qscale,
                       qc.x * qscale, qc.y * qscale, qc.z * qscale, qc.w * qscale};
        if (TERMS == 3) {
          // BOTH parts must come from the fp32-ROUNDED product: left alone, hipcc forms h with v_cvt_pk_f16_f32 of the rounded
          // product and m = fma(q, scale, -h') with h' = the UNROUNDED product rounded once (v_fma_mixlo_f16): where the fp32
          // product is an fp16 tie the two roundings differ by one fp16 ulp and h + m is off by 2^-11 (one query in ~3 000)
#pragma unroll
          for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(qq[e]));
        }
        if (TERMS == 3) {                                      // range test of the scaled queries of this block
          float mq = 0.f;
#pragma unroll
          for (int e = 0; e < 8; e += 2) mq = fmaxf(mq, fmaxf(fabsf(qq[e]), fabsf(qq[e + 1])));
          float sq = 1.0f, sq_inv = 1.0f;
          if (wh_out_of_range(mq, 0.015625f)) {
            wh_wave_scale(mq, sq, sq_inv);
#pragma unroll
            for (int e = 0; e < 8; ++e) qq[e] *= sq;
          }
          ss = sk * sq;                                         // the scores come out of the matrix cores times ss
          ss_inv = sk_inv * sq_inv;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          qf[e] = (_Float16)qq[e];
          qfm[e] = (_Float16)(qq[e] - (float)qf[e]);
        }
      }
      if (ib + 1 < NB) {                // scalar
        qa = make_float4(0.f, 0.f, 0.f, 0.f);
        qc = qa;
        const float* rp = row_ptr(token(i + 16));
        if (rp) {
          const float4* pq = reinterpret_cast<const float4*>(rp + 8 * g);
          qa = pq[0];
          qc = pq[1];
        }
      }
      f32x4 s[NB];
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        // bias (and the -inf of the padded keys) is the accumulator's initial value
        const f32x4 acc = *reinterpret_cast<const f32x4*>(bias_lds + i * BS + jb * 16 + 4 * g);
        f32x4 sc = acc;
        if (TERMS == 3) {                                      // smallest terms first
          if (ss != 1.0f) sc = acc * ss;                        // scalar: only windows that needed a range scale
          sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfm[jb], qf, sc, 0, 0, 0);
          sc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[jb], qfm, sc, 0, 0, 0);
        }
        s[jb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[jb], qf, sc, 0, 0, 0);
      }
      if (TERMS == 3 && ss != 1.0f) {                           // scalar
#pragma unroll
        for (int jb = 0; jb < NB; ++jb) s[jb] *= ss_inv;
      }
      if (mask_w) {   // scalar: last row / column of windows only
        // unconditional loads at clamped indices + a select (see window_attn.hip: a load behind a lane condition made
        // hipcc 7.2 restore stale AGPR copies of s[] in the lanes that skip it)
        const float* mrow = mask_w + min(i, Ntok - 1) * Ntok;
        if (MASK4) {                    // rows are 16-byte aligned: a lane's four keys are one load
#pragma unroll
          for (int jb = 0; jb < NB; ++jb) {
            const int j0 = jb * 16 + 4 * g;
            const f32x4 mv = *reinterpret_cast<const f32x4*>(mrow + min(j0, Ntok - 4));
            const float keep = (i < Ntok && j0 < Ntok) ? LOG2E : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) s[jb][r] += mv[r] * keep;
          }
        } else {
#pragma unroll
          for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = jb * 16 + 4 * g + r;
              const float mv = mrow[min(j, Ntok - 1)];
              s[jb][r] += (i < Ntok && j < Ntok) ? mv * LOG2E : 0.f;
            }
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
      f16x4 p[NB], pm[TERMS == 3 ? NB : 1];
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[jb][r] - mx);   // exp2(-inf) = 0 for padded keys
          sum += e;
          p[jb][r] = (_Float16)e;
          if (TERMS == 3) pm[jb][r] = (_Float16)(e - (float)p[jb][r]);
        }
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const float inv = (1.0f / sum) * sv_inv;     // of query n (the same value in the four lane groups); V's range scale undone

      // out[ib] (16 queries x 32 channels) = P[ib, :] @ V : two 16-channel halves
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const f16x4 b0 = *reinterpret_cast<const f16x4*>(vt + n * VS + jb * 16 + 4 * g);
        const f16x4 b1 = *reinterpret_cast<const f16x4*>(vt + (16 + n) * VS + jb * 16 + 4 * g);
        if (TERMS == 3) {
          const f16x4 b0m = *reinterpret_cast<const f16x4*>(vt + HD * VS + n * VS + jb * 16 + 4 * g);
          const f16x4 b1m = *reinterpret_cast<const f16x4*>(vt + HD * VS + (16 + n) * VS + jb * 16 + 4 * g);
          o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(pm[jb], b0, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(pm[jb], b1, o1, 0, 0, 0);
          o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(p[jb], b0m, o0, 0, 0, 0);
          o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(p[jb], b1m, o1, 0, 0, 0);
        }
        o0 = __builtin_amdgcn_mfma_f32_16x16x16f16(p[jb], b0, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x16f16(p[jb], b1, o1, 0, 0, 0);
      }
      // C layout: row = 4g + r -> query ib*16 + 4g + r (token and 1/sum: lane 4g + r); col = n -> channel n / 16 + n
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = __shfl(tok_i, 4 * g + r, 64);
        const float invr = __shfl(inv, 4 * g + r, 64);
        if (t >= 0) {
          float* op = out + ((long long)t * nH + h) * HD;
          op[n] = o0[r] * invr;
          op[16 + n] = o1[r] * invr;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();   // the next window's V staging overwrites vt
  }
}

template <int NB, bool MASK4, int TERMS, int NWV = wh_waves(NB, TERMS)>
static int launch_f16(const float* qkv, const float* qkv_bias, const float* bias, const float* shift_mask, int B_, int nW,
                      int nH, float scale, float* out, const WinImage& wi, int n_cu, hipStream_t st) {
  constexpr int NP = 16 * NB;
  constexpr int WH_WAVES = NWV;
  const size_t lds = (size_t)NP * (NP + 4) * sizeof(float) + (size_t)WH_WAVES * (TERMS == 3 ? 2 : 1) * 32 * (NP + WH_VT_PAD) * sizeof(_Float16);
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / lds));
  // workgroups = resident slots rounded DOWN to a multiple of the head count: one extra workgroup would double the tail
  int gx = std::max(1, n_cu * per_cu / nH);
  if constexpr (NB == 4 && TERMS == 3 && NWV == 8) {
    // 7 x 7 windows: a workgroup of 12 waves (three per SIMD) takes ~1.28 x the time of one of 8 per round of windows (measured
    // at the four Swin-T stages, gpurun_out/r06_u): it runs where it saves more rounds than that
    const long long r8 = ((long long)B_ + (long long)gx * 8 - 1) / ((long long)gx * 8), r12 = ((long long)B_ + (long long)gx * 12 - 1) / ((long long)gx * 12);
    if (1.28 * (double)r12 < (double)r8)
      return launch_f16<NB, MASK4, TERMS, 12>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
  }
  gx = std::min(gx, (B_ + WH_WAVES - 1) / WH_WAVES);
  const void* fn = reinterpret_cast<const void*>(&window_attn_img_f16<NB, MASK4, TERMS, NWV>);
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((window_attn_img_f16<NB, MASK4, TERMS, NWV>), dim3(gx, nH), dim3(64 * WH_WAVES), lds, st, qkv, qkv_bias, bias, shift_mask,
                     B_, nW, nH, scale, out, wi, (65536 + wi.ws - 1) / wi.ws);
  return check_launch("window_attn_img_f16");
}

// qkv [B, H*W, 3, nH, hd] in token order; out [B, H*W, nH*hd]
// terms = 1: fp16 operands; terms = 3: two fp16 parts per operand, three products (12 x 12 windows with four waves per
// workgroup)
int window_attention_image_f16mma(const float* qkv, const float* qkv_bias, const float* bias, const float* shift_mask, int B,
                                  int H, int W, int ws, int shift, int nH, int hd, float scale, int terms, float* out,
                                  hipStream_t st) {
  if (terms == 3 && hd != 32) return UNIVS_ERR_NOT_IMPLEMENTED;
  if (hd != 32) {
    set_error("window_attention_image (fp16 operands): head_dim=%d (only 32, the Swin-T/B/L value)", hd);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if (ws > 12) {
    set_error("window_attention_image (fp16 operands): %d x %d windows (max 12 x 12)", ws, ws);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  if ((long long)B * H * W * 3 * nH * hd >= 0x7FFFFFFFLL || nH > 65535) {
    set_error("window_attention_image (fp16 operands): qkv of %lld elements (32-bit offsets inside the kernel)",
              (long long)B * H * W * 3 * nH * hd);
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  const WinImage wi = win_image(H, W, ws, shift);
  const int nW = (wi.Hp / ws) * wi.nWx;
  const int B_ = B * nW;
  if (B_ == 0) return UNIVS_OK;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  const int ntok = ws * ws;
  if (terms == 3) {
    if (ntok <= 64) return launch_f16<4, false, 3>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
    if (ntok <= 96) return launch_f16<6, false, 3>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
    if (ntok % 4 == 0) return launch_f16<9, true, 3>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
    return launch_f16<9, false, 3>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
  }
  if (ntok <= 64) return launch_f16<4, false, 1>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
  if (ntok <= 96) return launch_f16<6, false, 1>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
  if (ntok % 4 == 0) return launch_f16<9, true, 1>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
  return launch_f16<9, false, 1>(qkv, qkv_bias, bias, shift_mask, B_, nW, nH, scale, out, wi, n_cu, st);
}

}  // namespace univs
