// Device-side pieces shared by the register-record MSDA kernels (msda_tiled3.hip, msda_tiled4.hip).
#pragma once

namespace univs {

typedef float t3v2 __attribute__((ext_vector_type(2)));
typedef float t3v4 __attribute__((ext_vector_type(4)));
typedef unsigned t3u2 __attribute__((ext_vector_type(2)));
#define T3_LDS __attribute__((address_space(3)))

// lane K of my DPP row (row_newbcast); every lane has a source, so there is no "old" value to materialise
template <int K>
__device__ __forceinline__ int t3_bcast(int v) {
  return __builtin_amdgcn_mov_dpp(v, 0x150 + K, 0xf, 0xf, true);
}

// the same for a register PAIR: one v_mov_b64_dpp (4.7-5.4 clk for two dwords against 2 x 4.2 for two v_mov_b32_dpp;
// row_newbcast is the one DPP control the 64-bit form accepts)
template <int K>
__device__ __forceinline__ long long t3_bcast64(long long v) {
  // (inline asm: the builtin form materialises an "old" value -- two extra moves per broadcast; the s_nop pads the
  // VALU-write -> DPP-read hazard hipcc cannot see inside the asm)
  long long r;
  asm("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%c2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}

// FUSED: the sampling locations and attention weights are not read from memory but made from the raw projections of
// MSDeformAttn.forward (ms_deform_attn.py:100-113), exactly as csrc/msda_prepare.hip makes them:
//   loc  = reference_point + offset / (W_l, H_l)            (IEEE division, then the add)
//   attn = softmax over the L*P logits of (query, head)     (exp(x - max) / sum)
// `in.loc` / `in.attn` are then unused; `in.proj` [N, Lq, row_stride] holds the offsets in columns [0, M*L*P*2) and the
// logits in columns [n_off, n_off + M*L*P), `in.ref` [N or 1, Lq, L, 2] the reference points.
struct T3Inputs {
  const float* loc;
  const float* attn;
  const float* proj;
  const float* ref;
  int row_stride, n_off;
  long long ref_batch_stride;   // 0: one set of reference points for all frames
};

}  // namespace univs
