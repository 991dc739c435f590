// ProCA attention (gfx950): every prompt query attends ONLY to its own prompt tokens -- batch = Q_p * T entries, query length 1,
// key length 1 + L (the query's own state first, then its L dense prompt tokens of that frame).
//
// Replaces, inside forward_transformer_prompt_self_attention_layer (univs/modeling/transformer_decoder/
// video_mask2former_transformer_decoder_univs.py:456-496 -> CrossAttentionLayer.forward_post, transformer_layers.py:95-115 ->
// nn.MultiheadAttention), the concatenation of the query state and the dense tokens into `memory`, its transposition, the same for
// the position embeddings, the scaled q k^T batched GEMM, the softmax and the p v batched GEMM: ~12 launches per decoder layer,
// ten layers per clip.  The dense tokens' key / value projections are two tall Linears on the tokens where they lie
// ([Q_p, L, T, C]: no copy); the first key / value come out of the same few-rows launch as the query (univs_small_linear: q, k0, v0).
//
// One wave per (batch entry, head); head_dim 32.  Phase 1: a lane owns keys lane, lane + 64, ...: the 32-term dot product from eight
// 16-byte loads of its key's 128 contiguous bytes; scores in LDS.  Softmax by wave reductions.  Phase 2: lane = (key parity,
// channel): p v over half of the keys each, 128-byte coalesced rows, the two halves added by one permlane swap.
#include "common.h"

namespace univs {

typedef float pa4 __attribute__((ext_vector_type(4)));

struct ProcaArgs {
  const float* qkv0;   // [B, 3 E]: q, k0, v0 of every batch entry b = qp * T + t
  const float* kd;     // [Q_p, L, T, E] dense keys
  const float* vd;     // [Q_p, L, T, E] dense values
  float* out;          // [B, E]
  int Qp, L, T, h;
  float scale;
};

__global__ __launch_bounds__(64) void proca_attn_kernel(ProcaArgs a) {
  extern __shared__ float sc[];                 // 1 + L scores
  const int lane = threadIdx.x;
  const int hh = blockIdx.x % a.h, b = blockIdx.x / a.h;
  const int qp = b / a.T, t = b - qp * a.T;
  const int E = a.h * 32, S = 1 + a.L;
  const float* q = a.qkv0 + (long long)b * 3 * E + hh * 32;
  pa4 qv[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) qv[i] = reinterpret_cast<const pa4*>(q)[i] * a.scale;
  const long long tstride = (long long)a.T * E;                       // floats between consecutive dense tokens of one query
  const float* kbase = a.kd + ((long long)qp * a.L * a.T + t) * E + hh * 32;
  const float* vbase = a.vd + ((long long)qp * a.L * a.T + t) * E + hh * 32;
  float mx = -__builtin_inff();
  for (int key = lane; key < S; key += 64) {
    const float* kp = key == 0 ? q + E : kbase + (long long)(key - 1) * tstride;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const pa4 kv = reinterpret_cast<const pa4*>(kp)[i];
      s = fmaf(qv[i].x, kv.x, fmaf(qv[i].y, kv.y, fmaf(qv[i].z, kv.z, fmaf(qv[i].w, kv.w, s))));
    }
    sc[key] = s;
    mx = fmaxf(mx, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int key = lane; key < S; key += 64) {
    const float e = __expf(sc[key] - mx);
    sc[key] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
  __syncthreads();                               // (one wave: the LDS writes above are visible to every lane below)
  const int dd = lane & 31, par = lane >> 5;
  float acc = 0.f;
  for (int key = par; key < S; key += 2) {
    const float* vp = key == 0 ? q + 2 * E : vbase + (long long)(key - 1) * tstride;
    acc = fmaf(sc[key], vp[dd], acc);
  }
  acc += __shfl_xor(acc, 32, 64);
  if (par == 0) a.out[(long long)b * E + hh * 32 + dd] = acc / sum;
}

int proca_attention_f32(const float* qkv0, const float* kd, const float* vd, int Qp, int L, int T, int h, int hd, float scale,
                        float* out, hipStream_t st) {
  if (hd != 32 || h < 1 || Qp < 1 || T < 1 || L < 0 || (long long)(1 + L) * 4 > 64 * 1024) return 0;
  const long long B = (long long)Qp * T;
  if (B * h > 0x7fffffffLL) return 0;
  ProcaArgs a{qkv0, kd, vd, out, Qp, L, T, h, scale};
  hipLaunchKernelGGL(proca_attn_kernel, dim3((unsigned)(B * h)), dim3(64), (size_t)(1 + L) * 4, st, a);
  const int rc = check_launch("proca_attn_kernel");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
