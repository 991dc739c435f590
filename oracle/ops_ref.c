/*
 * ops_ref.c -- CPU restatement (plain C, scalar) of the operator semantics on the UniVS hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: it is compiled with gcc into
 * oracle/_build/liboracle.so and used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg -- never by the product path (univs_amd/), which has no CPU fallback.
 *
 * Pinned against the reference: tests/test_oracle_golden.py compares every function here with the
 * golden vectors in tests/golden/ that oracle/gen_golden.py produced by running the reference's own
 * Python (ms_deform_attn_core_pytorch, forward_prediction_heads, WindowAttention) in the dev container.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ----------------------------------------------------------------------------------------------
 * Multi-scale deformable attention forward.
 * Follows mask2former/modeling/pixel_decoder/ops/src/cuda/ms_deform_im2col_cuda.cuh:242-304 (loop
 * structure, h_im/w_im mapping at :290-291, in-range test at :293) and the bilinear helper at :38-89
 * (corner validity, weights); equals ms_deform_attn_core_pytorch (ops/functions/ms_deform_attn_func.py
 * :52-72), i.e. grid_sample(align_corners=False, padding_mode='zeros').
 * Accumulation in double when acc64 != 0 (used to measure fp32 rounding of the GPU kernels).
 * -------------------------------------------------------------------------------------------- */
#define MSDA_IMPL(NAME, T)                                                                          \
  void NAME(const T* value, const int64_t* shapes, const int64_t* starts, const T* loc,             \
            const T* attn, int N, int S, int M, int D, int L, int Lq, int P, T* out) {              \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                         \
    for (int n = 0; n < N; ++n)                                                                     \
      for (int q = 0; q < Lq; ++q)                                                                  \
        for (int m = 0; m < M; ++m) {                                                               \
          const long long item = ((long long)n * Lq + q) * M + m;                                   \
          const T* lp = loc + item * L * P * 2;                                                     \
          const T* ap = attn + item * L * P;                                                        \
          T* op = out + item * D;                                                                   \
          for (int c = 0; c < D; ++c) op[c] = 0;                                                    \
          for (int l = 0; l < L; ++l) {                                                             \
            const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];                           \
            const T* vl = value + (((long long)n * S + starts[l]) * M + m) * D;                     \
            const long long row = (long long)M * D;                                                 \
            for (int p = 0; p < P; ++p) {                                                           \
              const T x = lp[(l * P + p) * 2], y = lp[(l * P + p) * 2 + 1], w = ap[l * P + p];      \
              const T him = y * H - (T)0.5, wim = x * W - (T)0.5;                                   \
              if (!(him > -1 && wim > -1 && him < H && wim < W)) continue;                          \
              const int h0 = (int)floor((double)him), w0 = (int)floor((double)wim);                 \
              const T lh = him - h0, lw = wim - w0, hh = 1 - lh, hw = 1 - lw;                       \
              const T w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;                       \
              const int ok1 = h0 >= 0 && w0 >= 0, ok2 = h0 >= 0 && w0 + 1 <= W - 1;                 \
              const int ok3 = h0 + 1 <= H - 1 && w0 >= 0, ok4 = h0 + 1 <= H - 1 && w0 + 1 <= W - 1; \
              for (int c = 0; c < D; ++c) {                                                         \
                const T v1 = ok1 ? vl[((long long)h0 * W + w0) * row + c] : 0;                      \
                const T v2 = ok2 ? vl[((long long)h0 * W + w0 + 1) * row + c] : 0;                  \
                const T v3 = ok3 ? vl[((long long)(h0 + 1) * W + w0) * row + c] : 0;                \
                const T v4 = ok4 ? vl[((long long)(h0 + 1) * W + w0 + 1) * row + c] : 0;            \
                op[c] += (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4) * w;                               \
              }                                                                                     \
            }                                                                                       \
          }                                                                                         \
        }                                                                                           \
  }

MSDA_IMPL(oracle_msda_forward_f32, float)
MSDA_IMPL(oracle_msda_forward_f64, double)

/* ----------------------------------------------------------------------------------------------
 * Mask decode: out[q,t,n] = sum_c mask_embed[t,q,c] * mask_features[t,c,n]
 * == torch.einsum("btqc,btchw->btqhw").transpose(1,2) for b = 1
 * (univs/modeling/transformer_decoder/video_mask2former_transformer_decoder_univs.py:527-528).
 * c-ordered fp32 accumulation (what an fp32 fmaf chain computes).
 * -------------------------------------------------------------------------------------------- */
void oracle_mask_decode_f32(const float* mask_embed, const float* mask_features, int T, int Q, int C,
                            long long HW, float* out) {
  float* acc = (float*)malloc(sizeof(float) * (size_t)HW);
  for (int t = 0; t < T; ++t)
    for (int q = 0; q < Q; ++q) {
      const float* e = mask_embed + ((long long)t * Q + q) * C;
      memset(acc, 0, sizeof(float) * (size_t)HW);
      for (int c = 0; c < C; ++c) {
        const float ec = e[c];
        const float* f = mask_features + ((long long)t * C + c) * HW;
        for (long long n = 0; n < HW; ++n) acc[n] = fmaf(ec, f[n], acc[n]);
      }
      memcpy(out + ((long long)q * T + t) * HW, acc, sizeof(float) * (size_t)HW);
    }
  free(acc);
}

/* Attention-mask rule given logits at the target resolution:
 * mask = sigmoid(logit) < 0.5  (...decoder_univs.py:565), then rows that are entirely True are reset to
 * False (:390).  logits [T,Q,hw] -> mask bytes [T,Q,hw]. */
void oracle_attn_mask_from_logits(const float* logits, int T, int Q, long long hw, uint8_t* mask) {
  for (long long r = 0; r < (long long)T * Q; ++r) {
    const float* x = logits + r * hw;
    uint8_t* m = mask + r * hw;
    int any_visible = 0;
    for (long long i = 0; i < hw; ++i) {
      const float s = 1.0f / (1.0f + expf(-x[i]));
      m[i] = s < 0.5f;
      any_visible |= !m[i];
    }
    if (!any_visible) memset(m, 0, (size_t)hw);
  }
}

/* ----------------------------------------------------------------------------------------------
 * Swin window attention core (mask2former/modeling/backbone/swin.py:137-168 between qkv and proj):
 * attn = softmax((q*scale) k^T + bias[h] (+ mask[b % nW])); out = attn v.
 * qkv [B_, Ntok, 3, nH, hd]; bias [nH, Ntok, Ntok]; shift_mask [nW, Ntok, Ntok] or NULL;
 * out [B_, Ntok, nH*hd].
 * -------------------------------------------------------------------------------------------- */
void oracle_window_attention_f32(const float* qkv, const float* bias, const float* shift_mask, int B_,
                                 int nW, int Ntok, int nH, int hd, float scale, float* out) {
  float* s = (float*)malloc(sizeof(float) * (size_t)Ntok);
  const long long ts = 3LL * nH * hd;
  for (long long b = 0; b < B_; ++b)
    for (int h = 0; h < nH; ++h)
      for (int i = 0; i < Ntok; ++i) {
        const float* q = qkv + (b * Ntok + i) * ts + (long long)h * hd;
        float mx = -INFINITY;
        for (int j = 0; j < Ntok; ++j) {
          const float* k = qkv + (b * Ntok + j) * ts + (long long)(nH + h) * hd;
          float d = 0.f;
          for (int c = 0; c < hd; ++c) d += (q[c] * scale) * k[c];
          d += bias[((long long)h * Ntok + i) * Ntok + j];
          if (shift_mask) d += shift_mask[((b % nW) * Ntok + i) * (long long)Ntok + j];
          s[j] = d;
          if (d > mx) mx = d;
        }
        float sum = 0.f;
        for (int j = 0; j < Ntok; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
        float* o = out + ((b * Ntok + i) * nH + h) * hd;
        for (int c = 0; c < hd; ++c) o[c] = 0.f;
        for (int j = 0; j < Ntok; ++j) {
          const float p = s[j] / sum;
          const float* v = qkv + (b * Ntok + j) * ts + (long long)(2 * nH + h) * hd;
          for (int c = 0; c < hd; ++c) o[c] += p * v[c];
        }
      }
  free(s);
}

/* ---------------------------------------------------------------------------------------------------------------------
 * COCO run-length coding of a binary mask (result format of the VIS loop: the reference calls pycocotools'
 * mask.encode per object and frame, univs/inference/inference_video_entity.py:944-948).  pycocotools (third party, version
 * un-pinned by INSTALL.md) is absent from this image; this restates its published algorithm, maskApi.c:
 *   rleEncode:   column-major runs, the first run counts zeros (possibly 0 of them);
 *   rleToString: each count (from the 4th on: minus the count two places back) as 5-bit groups, low group first, bit 5 =
 *                "more groups follow", sign-extended from bit 4 of the last group, each group + 48 as a character.
 * Returns the string length (without the terminator); `out` must hold 7 * (h * w + 1) + 1 bytes.
 * ------------------------------------------------------------------------------------------------------------------- */
long long oracle_rle_encode(const uint8_t* mask, int h, int w, char* out, long long* counts_out, long long* n_counts) {
  const long long n = (long long)h * w;
  long long m = 0, run = 0, p = 0;
  uint8_t cur = 0;
  long long* cnts = counts_out;
  for (long long i = 0; i < n; ++i) {
    const long long col = i / h, row = i - col * h;      /* column-major walk over a row-major array */
    const uint8_t v = mask[row * w + col] ? 1 : 0;
    if (v != cur) {
      cnts[m++] = run;
      run = 0;
      cur = v;
    }
    ++run;
  }
  cnts[m++] = run;
  *n_counts = m;
  for (long long i = 0; i < m; ++i) {
    long long x = cnts[i];
    if (i > 2) x -= cnts[i - 2];
    int more = 1;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      out[p++] = (char)(c + 48);
    }
  }
  out[p] = 0;
  return p;
}
