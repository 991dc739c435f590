// Masked multi-head cross-attention core in one pass over the keys: softmax(Q K^T * scale, masked) V per (batch entry, head),
// the [N h, L, S] scores never reach memory.
//
// Replaces, inside nn.MultiheadAttention as the UniVS decoder's CrossAttentionLayer uses it
// (univs/modeling/transformer_decoder/transformer_layers.py:95-115, called at ...decoder_univs.py:400-405): the scaled score
// GEMM, `masked_fill(attn_mask, -inf)`, the softmax over the S = H_l W_l keys and the product with V -- three launches and
// five passes over a [T 8, Q', H_l W_l] fp32 tensor (235 MB at the 1/8 level of a 720p clip) per decoder layer.  The
// in- and out-projections stay Linears.
//
// Arithmetic: the three-product fp16 form of the window attention (window_attn_f16.hip, TERMS = 3): q * scale * log2e, k, v
// and the un-normalised probabilities are each two fp16 parts (h = fp16(x), m = fp16(x - h)), every product is
// m h' + h m' + h h' on v_mfma_f32_16x16x32_f16 with fp32 accumulation (error <= 2^-21.7 per product); softmax in fp32 in the
// exp2 domain.  Range: as there, a wave tests its operands and applies a wave-uniform power of two only when a magnitude
// would leave fp16's range or the whole block is tiny.
//
// Organisation: one WAVE per workgroup handles one (batch entry n, head h) and one segment of the keys, all (<= 128) queries:
//   * S^T = K (Q scale)^T per 16-key block and 16-query block, so that a lane holds S[query j][keys 4 g .. 4 g + 3]: the
//     probabilities of TWO key blocks are, as they stand, the A operand of one 16x16x32 product P V over 32 keys with the
//     k-order (4 g + e, 16 + 4 g + e);
//   * V is staged per wave in LDS as fp16 parts, transposed ([plane][channel][32 keys]): the transpose is done in registers
//     on the way in (a lane loads 4 keys x 4 channels and writes four 8-byte rows), the B operand is two 8-byte reads;
//   * online softmax with a LAZY reference maximum per query: it is raised (and the accumulators rescaled: four cross-lane
//     reads per query block) only when a block's maximum exceeds it by more than 8 -- un-normalised probabilities up to 2^8
//     are harmless in fp32 accumulators and fp16 parts -- which after the first blocks is rare;
//   * the Q fragments (two fp16 parts, 16 NQB x 32 x 4 bytes) live in LDS in B-operand order and are re-read per query block
//     (two conflict-free 16-byte reads), which leaves the registers for a PREFETCH of the next iteration's K rows, V rows and
//     mask words: a wave has one partner per SIMD at most, so nothing else hides the memory latency of these loads;
//   * every wave writes (m, l, O[32]) per query for its segment; a second small kernel merges the segments
//     (out = sum_p O_p 2^(m_p - M) / sum_p l_p 2^(m_p - M)) and stores [L, N, h d].
#include "common.h"
#include "config.h"
#include "f16x3.h"

#include <algorithm>

namespace univs {

typedef _Float16 xa_h4 __attribute__((ext_vector_type(4)));
typedef unsigned xa_u2 __attribute__((ext_vector_type(2)));

struct XaArgs {
  const float* q;            // [L, N, H * 32]
  const float* k;            // [S, N, H * 32]
  const float* v;            // [S, N, H * 32]
  const unsigned char* mask; // [N, L, S] (non-zero = masked out) or null
  const unsigned* flags;     // [N, Lfull] or null: row (n, l) of the mask counts only where flags == gen; elsewhere every key is visible
  unsigned gen;              //   (the all-masked-row rule of ...decoder_univs.py:390, deferred: mask_decode.hip)
  float* ws;                 // [N * H][nseg][16 NQB][34]: O[32], m, l
  float* out;                // [L, N, H * 32]
  int L, S, N, H, nseg;      // L: queries; a workgroup (blockIdx.z = chunk) handles queries [128 chunk, 128 chunk + 128)
  int Lfull, l0;             // rows of the mask per batch entry (= L), 0
  int ldq, ldk, ldv;         // floats between consecutive batch entries of q / k / v (>= H * 32: slices of wider projections)
  float qscale;              // scale * log2(e)
};

constexpr int XA_PART = 34;
#ifndef UNIVS_XA_VS
#define UNIVS_XA_VS 36
#endif
// halves per channel row of the transposed V: 32 keys + 4 of padding = 4 x an odd number, so that the 16 lanes of an 8-byte read (one
// channel row each) start in 16 distinct bank pairs (40 = 4 x 10, round 4's stride: rows j and j + 8 shared their banks, 33 - 45 % conflict cycles)
constexpr int XA_VS = UNIVS_XA_VS;

// largest magnitude of the wave -> (power of two that brings it into [2^14, 2^15), its inverse), wave-uniform
__device__ __forceinline__ void xa_wave_scale(float mx, float& s, float& inv, int emin = -40, int emax = 128) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  int e = (int)((__builtin_bit_cast(unsigned, mx) >> 23) & 255u) - 127;
  e = max(emin, min(e, emax));
  s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((127 + 14 - e) << 23));
  inv = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane((127 - 14 + e) << 23));
}
__device__ __forceinline__ bool xa_out_of_range(float mx, float lo) {
  return __builtin_amdgcn_ballot_w64(!(mx < 32768.0f)) != 0 || __builtin_amdgcn_ballot_w64(mx >= lo) == 0;
}
__device__ __forceinline__ void xa_split8(const float (&x)[8], f16x8& h, f16x8& m) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const _Float16 hh = (_Float16)x[e];
    h[e] = hh;
    m[e] = (_Float16)(x[e] - (float)hh);
  }
}
// maximum / sum over the four lanes (lane >> 4) that share lane & 15
__device__ __forceinline__ float xa_col_max(float v) {
  const xa_u2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float r1 = fmaxf(__uint_as_float(s1.x), __uint_as_float(s1.y));
  const xa_u2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
  return fmaxf(__uint_as_float(s2.x), __uint_as_float(s2.y));
}
__device__ __forceinline__ float xa_col_sum(float v) {
  const xa_u2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float r1 = __uint_as_float(s1.x) + __uint_as_float(s1.y);
  const xa_u2 s2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r1), __float_as_uint(r1), false, false);
  return __uint_as_float(s2.x) + __uint_as_float(s2.y);
}

template <int NQB>
__global__ __launch_bounds__(64, 2) void xattn_partial(const XaArgs a) {
  constexpr int HD = 32;
  __shared__ __attribute__((aligned(16))) _Float16 vt[2 * HD * XA_VS];   // [plane][channel][XA_VS]
  __shared__ __attribute__((aligned(16))) f16x8 qlds[2 * NQB * 64];      // [part][query block][lane]
  const int lane = threadIdx.x;
  const int j = lane & 15, g = lane >> 4;
  const int seg = blockIdx.x, nh = blockIdx.y, chunk = blockIdx.z;
  const int n = nh / a.H, h = nh - n * a.H;
  const int l0 = 128 * chunk;                                    // first query of this chunk
  const int L = min(128, a.L - l0), S = a.S, N = a.N;            // queries of this chunk
  const int E = a.H * HD;
  const long long qrow = (long long)N * a.ldq, krow = (long long)N * a.ldk, vrow = (long long)N * a.ldv;   // floats between sequence positions
  const float* qb_ = a.q + (long long)l0 * qrow + (long long)n * a.ldq + h * HD;
  const float* kb_ = a.k + (long long)n * a.ldk + h * HD;
  const float* vb_ = a.v + (long long)n * a.ldv + h * HD;
  (void)E;

  // this segment's range of 32-key iterations
  const int nit = (S + 31) >> 5;
  const int it0 = (int)((long long)nit * seg / a.nseg), it1 = (int)((long long)nit * (seg + 1) / a.nseg);

  // ---- Q fragments: lane's Q[16 qb + j][8 g .. 8 g + 7] * scale * log2e as two fp16 parts -> LDS (B-operand order)
  float ss = 1.0f, ss_inv = 1.0f;                                // the scores come out of the matrix cores times ss (Q's and K's range scales)
  float sq = 1.0f, sq_inv = 1.0f;
  {
    float qraw[NQB][8];
    float mq = 0.f;
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      const int qi = 16 * qb + j;
      float4 x0 = make_float4(0.f, 0.f, 0.f, 0.f), x1 = x0;
      if (qi < L) {
        const float4* p = reinterpret_cast<const float4*>(qb_ + (long long)qi * qrow + 8 * g);
        x0 = p[0];
        x1 = p[1];
      }
      const float t[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = t[e] * a.qscale;
        asm volatile("" : "+v"(v));                              // both fp16 parts from the ROUNDED product (window_attn_f16.hip)
        qraw[qb][e] = v;
        mq = fmaxf(mq, fabsf(v));
      }
    }
    if (xa_out_of_range(mq, 0.015625f)) {
      xa_wave_scale(mq, sq, sq_inv);
#pragma unroll
      for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
        for (int e = 0; e < 8; ++e) qraw[qb][e] *= sq;
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      f16x8 h8, m8;
      xa_split8(qraw[qb], h8, m8);
      qlds[qb * 64 + lane] = h8;
      qlds[(NQB + qb) * 64 + lane] = m8;
    }
  }

  float mref[NQB], lsum[NQB];                                    // of query 16 qb + j (lsum: this lane's keys only until the end)
  f32x4 oacc[NQB][2];                                            // O[queries 16 qb + 4 g + r][channel 16 half + j]
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    mref[qb] = -1.0e30f;
    lsum[qb] = 0.f;
    oacc[qb][0] = oacc[qb][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // V staging roles: lanes 0-31 key block A (keys s0 .. s0 + 15), lanes 32-63 block B; key group kgl (4 keys), channel group hg
  const int half = lane >> 5, kgl = (lane >> 3) & 3, hg = lane & 7;

  // raw operands of one 32-key iteration: K rows (lane: key 16 kb + j, channels 8 g ..), V rows (lane: 4 keys x 4 channels), and
  // the mask dword of (query 16 qb + j, keys 16 kb + 4 g ..) per query block; loaded one iteration ahead
  float4 kn[2][2], vn[4];
  unsigned mwn[NQB][2];
  bool mrow[NQB];                                                // my query's mask row counts (see XaArgs.flags)
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb)
    mrow[qb] = a.mask != nullptr && (a.flags == nullptr || a.flags[(long long)n * a.Lfull + min(l0 + 16 * qb + j, a.Lfull - 1)] == a.gen);
  auto load_iter = [&](int it) __attribute__((always_inline)) {
    const int s0 = it << 5;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int s = s0 + 16 * kb + j;
      kn[kb][0] = kn[kb][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s < S) {
        const float4* p = reinterpret_cast<const float4*>(kb_ + (long long)s * krow + 8 * g);
        kn[kb][0] = p[0];
        kn[kb][1] = p[1];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int s = s0 + 16 * half + 4 * kgl + e;
      vn[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s < S) vn[e] = *reinterpret_cast<const float4*>(vb_ + (long long)s * vrow + 4 * hg);
    }
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        mwn[qb][kb] = 0u;
        if (mrow[qb]) {
          // a lane's four keys are one aligned dword of its query's mask row (rows are padded to a multiple of four bytes when S is not one)
          const int qi = min(l0 + 16 * qb + j, a.Lfull - 1);
          const int pitch = (S + 3) & ~3;
          const int sbc = min(s0 + 16 * kb + 4 * g, pitch - 4);
          mwn[qb][kb] = *reinterpret_cast<const unsigned*>(a.mask + ((long long)n * a.Lfull + qi) * pitch + sbc);
        }
      }
  };
  if (it0 < it1) load_iter(it0);
  __builtin_amdgcn_s_waitcnt(0xc07f);                            // lgkmcnt(0): the Q fragments are in LDS
  __builtin_amdgcn_wave_barrier();

#pragma unroll 1
  for (int it = it0; it < it1; ++it) {
    const int s0 = it << 5;
    // ---- this iteration's operands out of the prefetch registers; the next iteration's loads go out behind them
    float kraw[2][8];
    float4 vr[4];
    unsigned mw[NQB][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const float t[8] = {kn[kb][0].x, kn[kb][0].y, kn[kb][0].z, kn[kb][0].w, kn[kb][1].x, kn[kb][1].y, kn[kb][1].z, kn[kb][1].w};
#pragma unroll
      for (int e = 0; e < 8; ++e) kraw[kb][e] = t[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) vr[e] = vn[e];
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      mw[qb][0] = mwn[qb][0];
      mw[qb][1] = mwn[qb][1];
    }
    if (it + 1 < it1) load_iter(it + 1);                         // scalar

    // ---- K: two blocks, two fp16 parts
    f16x8 kh[2], km[2];
    float sk = 1.0f, sk_inv = 1.0f;
    {
      float mk = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 8; e += 2) mk = fmaxf(mk, fmaxf(fabsf(kraw[kb][e]), fabsf(kraw[kb][e + 1])));
      if (xa_out_of_range(mk, 0.0625f)) {
        xa_wave_scale(mk, sk, sk_inv);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 0; e < 8; ++e) kraw[kb][e] *= sk;
      }
      xa_split8(kraw[0], kh[0], km[0]);
      xa_split8(kraw[1], kh[1], km[1]);
    }
    ss = sq * sk;
    ss_inv = sq_inv * sk_inv;

    // ---- V: 4 keys x 4 channels per lane, transposed into LDS as two fp16 planes
    float sv_inv = 1.0f, sv_fwd = 1.0f;
    {
      float mv = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        mv = fmaxf(fmaxf(mv, fmaxf(fabsf(vr[e].x), fabsf(vr[e].y))), fmaxf(fabsf(vr[e].z), fabsf(vr[e].w)));
      // (an all-zero block -- zero-valued keys -- has nothing to scale: left alone, it no longer multiplies O by 2^54 and back; the
      // scale's exponent is clamped to [-30, 30], so O times the scale stays inside fp32 for any O the earlier blocks left: ADVICE r05)
      if (xa_out_of_range(mv, 0.0625f) && __builtin_amdgcn_ballot_w64(mv > 0.f) != 0) {
        float sv;
        xa_wave_scale(mv, sv, sv_inv, -16, 44);
        sv_fwd = sv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vr[e].x *= sv;
          vr[e].y *= sv;
          vr[e].z *= sv;
          vr[e].w *= sv;
        }
      }
      const float vv[4][4] = {{vr[0].x, vr[1].x, vr[2].x, vr[3].x}, {vr[0].y, vr[1].y, vr[2].y, vr[3].y},
                              {vr[0].z, vr[1].z, vr[2].z, vr[3].z}, {vr[0].w, vr[1].w, vr[2].w, vr[3].w}};
      __builtin_amdgcn_wave_barrier();                           // the previous iteration's reads of vt are done (one wave: program order)
      _Float16* dst = vt + (4 * hg) * XA_VS + 16 * half + 4 * kgl;
#pragma unroll
      for (int c = 0; c < 4; ++c) {                              // channel 4 hg + c: four keys
        xa_h4 ch, cm;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ch[e] = (_Float16)vv[c][e];
          cm[e] = (_Float16)(vv[c][e] - (float)ch[e]);
        }
        *reinterpret_cast<xa_h4*>(dst + c * XA_VS) = ch;
        *reinterpret_cast<xa_h4*>(dst + HD * XA_VS + c * XA_VS) = cm;
      }
    }

    const bool v_scaled = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sv_inv)) != 0x3f800000;   // scalar
    // ---- V^T fragments of this iteration: B[channel 16 half + j][k-slots: keys 4 g + e, 16 + 4 g + e], two parts
    __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): this wave's LDS writes are visible to its own reads
    __builtin_amdgcn_wave_barrier();
    f16x8 vh[2], vm[2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const _Float16* row = vt + (16 * hf + j) * XA_VS + 4 * g;
      const xa_h4 a0 = *reinterpret_cast<const xa_h4*>(row), a1 = *reinterpret_cast<const xa_h4*>(row + 16);
      const xa_h4 b0 = *reinterpret_cast<const xa_h4*>(row + HD * XA_VS), b1 = *reinterpret_cast<const xa_h4*>(row + HD * XA_VS + 16);
      vh[hf] = (f16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      vm[hf] = (f16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }

    // ---- per query block: scores of the two key blocks, mask, lazy maximum, probabilities, O += P V over the 32 keys
#pragma unroll
    for (int qb = 0; qb < NQB; ++qb) {
      const f16x8 qh = qlds[qb * 64 + lane], qm = qlds[(NQB + qb) * 64 + lane];
      f32x4 sc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(km[kb], qh, c, 0, 0, 0);   // smallest terms first
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[kb], qm, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(kh[kb], qh, c, 0, 0, 0);
        if (ss != 1.0f) c *= ss_inv;                              // scalar: only blocks that needed a range scale
        const int sb = s0 + 16 * kb + 4 * g;                     // my four keys
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (((mw[qb][kb] >> (8 * r)) & 0xffu) != 0u || sb + r >= S) c[r] = -INFINITY;
        sc[kb] = c;
      }
      float bm = fmaxf(fmaxf(fmaxf(sc[0][0], sc[0][1]), fmaxf(sc[0][2], sc[0][3])), fmaxf(fmaxf(sc[1][0], sc[1][1]), fmaxf(sc[1][2], sc[1][3])));
      bm = xa_col_max(bm);                                       // over the 32 keys, for query j
      if (__builtin_amdgcn_ballot_w64(bm > mref[qb] + 8.0f) != 0) {   // rare after the first blocks: raise reference maxima
        const float mnew = (bm > mref[qb] + 8.0f) ? bm : mref[qb];
        const float alpha = __builtin_amdgcn_exp2f(mref[qb] - mnew);   // of query j (1 when unchanged; 0 at the very first block)
        mref[qb] = mnew;
        lsum[qb] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                            // O's rows are queries 4 g + r: their factors sit in lanes 4 g + r
          const float ar = __shfl(alpha, 4 * g + r, 64);
          oacc[qb][0][r] *= ar;
          oacc[qb][1][r] *= ar;
        }
      }
      float pv[8];
      float part = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(sc[kb][r] - mref[qb]);   // exp2(-inf) = 0 for masked keys
          part += e;
          pv[4 * kb + r] = e;                                    // in (0, 2^8]: never scaled (V's range scale is undone in fp32 below)
        }
      lsum[qb] += part;
      f16x8 ph, pm;                                              // P[query 16 qb + j][k-slots: block A keys 4 g + e, block B keys 4 g + e]
      xa_split8(pv, ph, pm);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        // A range-scaled V block (wave-uniform): O is taken into the block's units (x scale, a power of two: exact), P V' accumulates
        // onto it, and O returns (x 1 / scale, exact) -- the fp32 accumulation of the unscaled products, in place.  Undoing the scale on
        // P instead pushed the probabilities into fp16's subnormals whenever |v| < 2^-4 over a whole block; a separate
        // accumulator for the block cost 14 - 30 registers and spilled at seven / eight query blocks (xattn_partial<7>: 49 -> 97 us).
        // (x scale <= 2^54 cannot overflow unless one head holds values 2^70 apart: O <= 2^8 x keys x max |v|.)
        f32x4 o = oacc[qb][hf];
        if (v_scaled) o *= sv_fwd;
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(pm, vh[hf], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vm[hf], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vh[hf], o, 0, 0, 0);
        if (v_scaled) o *= sv_inv;
        oacc[qb][hf] = o;
      }
    }
  }

  // ---- this segment's partial: O rows = queries 4 g + r, columns = channels 16 half + j; m and l from the lanes g == 0
  float* wsb = a.ws + (((long long)chunk * gridDim.y + nh) * a.nseg + seg) * (16 * NQB) * XA_PART;
#pragma unroll
  for (int qb = 0; qb < NQB; ++qb) {
    const float ltot = xa_col_sum(lsum[qb]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* p = wsb + (long long)(16 * qb + 4 * g + r) * XA_PART;
      p[j] = oacc[qb][0][r];
      p[16 + j] = oacc[qb][1][r];
    }
    if (g == 0) {
      float* p = wsb + (long long)(16 * qb + j) * XA_PART;
      p[32] = mref[qb];
      p[33] = ltot;
    }
  }
}

// out[q, n, h * 32 + c] = sum_p O_p[c] 2^(m_p - M) / sum_p l_p 2^(m_p - M); one thread per (n h, query, channel)
__global__ __launch_bounds__(256) void xattn_merge(const float* __restrict__ ws, float* __restrict__ out, int L, int Lp, int N, int H, int nseg) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)N * H * L * 32;
  if (idx >= total) return;
  const int c = (int)(idx & 31);
  const long long t = idx >> 5;
  const int q = (int)(t % L);
  const int nh = (int)(t / L);
  const int n = nh / H, h = nh - n * H;
  const int chunk = q >> 7, ql = q & 127;                        // (one chunk: Lp = 16 * query blocks; several: Lp = 128)
  const float* base = ws + ((((long long)chunk * N * H + nh) * nseg) * Lp + ql) * XA_PART;
  const long long pstride = (long long)Lp * XA_PART;
  float M = -INFINITY;
#pragma unroll 8                                                 // (independent loads in flight: the kernel is pure latency)
  for (int p = 0; p < nseg; ++p) M = fmaxf(M, base[p * pstride + 32]);
  float acc = 0.f, l = 0.f;
#pragma unroll 8
  for (int p = 0; p < nseg; ++p) {
    const float f = __builtin_amdgcn_exp2f(base[p * pstride + 32] - M);
    acc = fmaf(base[p * pstride + c], f, acc);
    l = fmaf(base[p * pstride + 33], f, l);
  }
  out[((long long)q * N + n) * (H * 32) + h * 32 + c] = acc / l;
}

static int xa_cus() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 1) {
      (void)hipGetLastError();
      v = 256;
    }
    n_cu = v;
  }
  return n_cu;
}

// segments per (batch entry, head, chunk of 128 queries): one round of the 8 waves per CU the kernel's registers allow, each wave
// with at least three 32-key iterations (the Q fragments cost about one)
static int xa_segments(int L, int S, int N, int H) {
  const int nit = (S + 31) / 32, nchunks = (L + 127) / 128;
  const int forced = config().xattn_segments;
  if (forced > 0) return std::min(forced, nit);
  const long long want = std::max<long long>(1, (8LL * xa_cus()) / ((long long)N * H * nchunks));
  // (short key sequences -- the decoder's self-attention over Q' T = 500 tokens -- are latency-bound: at least 8 segments, 22.8 vs 30.3 us)
  return (int)std::max<long long>(1, std::min<long long>(want, std::max(nit / 3, std::min(nit, 8))));
}
int cross_attention_segments(int S, int N, int H) { return xa_segments(1, S, N, H); }

size_t cross_attention_workspace_floats(int L, int S, int N, int H) {
  const int nqb = (std::min(L, 128) + 15) / 16, nchunks = (L + 127) / 128;
  return (size_t)nchunks * N * H * xa_segments(L, S, N, H) * (16 * nqb) * XA_PART;
}

// returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED when the shape is not covered.  Queries beyond 128 are handled in chunks.
int cross_attention_f32(const float* q, const float* k, const float* v, const unsigned char* mask, const unsigned* row_flags,
                        unsigned generation, int L, int S, int N, int H, int hd, int ldq, int ldk, int ldv, float scale, float* ws,
                        float* out, hipStream_t st) {
  if (L <= 0 || N <= 0 || H <= 0) return UNIVS_OK;
  auto mis = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; };
  if (hd != 32 || S < 32 || (mask && ((row_flags && S % 4 != 0) || (reinterpret_cast<uintptr_t>(mask) & 3))) || mis(q) || mis(k) || mis(v) || mis(out) ||
      mis(ws) || (long long)N * H > 65535)
    return UNIVS_ERR_NOT_IMPLEMENTED;
  const int nseg = xa_segments(L, S, N, H);
  const int E = H * 32;
  ldq = ldq > 0 ? ldq : E; ldk = ldk > 0 ? ldk : E; ldv = ldv > 0 ? ldv : E;
  if (ldq < E || ldk < E || ldv < E || ldq % 4 || ldk % 4 || ldv % 4) return UNIVS_ERR_NOT_IMPLEMENTED;
  const int nchunks = (L + 127) / 128;                           // a workgroup handles up to 128 queries
  const int nqb = (std::min(L, 128) + 15) / 16;
  if (nchunks > 65535) return UNIVS_ERR_NOT_IMPLEMENTED;
  XaArgs a{};
  a.q = q; a.k = k; a.v = v;
  a.mask = mask;
  a.flags = mask ? row_flags : nullptr;
  a.gen = generation;
  a.ws = ws; a.out = out;
  a.L = L; a.S = S; a.N = N; a.H = H; a.nseg = nseg;
  a.Lfull = L; a.l0 = 0;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.qscale = scale * 1.4426950408889634f;
  dim3 grid((unsigned)nseg, (unsigned)(N * H), (unsigned)nchunks);
  switch (nqb) {
    case 1: hipLaunchKernelGGL(xattn_partial<1>, grid, dim3(64), 0, st, a); break;
    case 2: hipLaunchKernelGGL(xattn_partial<2>, grid, dim3(64), 0, st, a); break;
    case 3: hipLaunchKernelGGL(xattn_partial<3>, grid, dim3(64), 0, st, a); break;
    case 4: hipLaunchKernelGGL(xattn_partial<4>, grid, dim3(64), 0, st, a); break;
    case 5: hipLaunchKernelGGL(xattn_partial<5>, grid, dim3(64), 0, st, a); break;
    case 6: hipLaunchKernelGGL(xattn_partial<6>, grid, dim3(64), 0, st, a); break;
    case 7: hipLaunchKernelGGL(xattn_partial<7>, grid, dim3(64), 0, st, a); break;
    default: hipLaunchKernelGGL(xattn_partial<8>, grid, dim3(64), 0, st, a); break;
  }
  int rc = check_launch("xattn_partial");
  if (rc != UNIVS_OK) return rc;
  const long long total = (long long)N * H * L * 32;
  hipLaunchKernelGGL(xattn_merge, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ws, out, L, 16 * nqb, N, H, nseg);
  return check_launch("xattn_merge");
}

}  // namespace univs
