// Multi-scale deformable attention backward (gfx950): gradients w.r.t. value, sampling locations and
// attention weights.  Completes the operator boundary B2 (ops/src/ms_deform_attn.h:46-66,
// cuda/ms_deform_attn_cuda.cu:88-153); training itself is outside the inference hot path, so this is a
// correctness-first kernel: one thread per sample (n, q, m, l, p), channel loop in registers, float
// atomics into grad_value.  The math is the derivative of the forward's bilinear sample
//   out[c] += aw * (w00 v00 + w01 v01 + w10 v10 + w11 v11)[c],  w00 = (1-lh)(1-lw), ...
// with corners outside the image contributing nothing (zero padding) and samples outside the band
// (-1, H) x (-1, W) skipped, exactly like the forward (ms_deform_im2col_cuda.cuh:285-293).
#include "msda_common.h"

namespace univs {

__global__ __launch_bounds__(256) void msda_bwd_f32_kernel(const float* __restrict__ value, LevelTable lv,
                                                           const float* __restrict__ loc,
                                                           const float* __restrict__ attn,
                                                           const float* __restrict__ grad_out, int N, int S, int M,
                                                           int D, int L, int Lq, int P, float* __restrict__ grad_value,
                                                           float* __restrict__ grad_loc, float* __restrict__ grad_attn,
                                                           long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  // idx = (((n * Lq + q) * M + m) * L + l) * P + p   (the layout of attn_weight)
  long long t = idx;
  t /= P;
  const int l = (int)(t % L);
  t /= L;
  const int m = (int)(t % M);
  t /= M;
  const long long nq = t;            // n * Lq + q
  const int n = (int)(nq / Lq);
  const int H = lv.H[l], W = lv.W[l];
  const float x = loc[idx * 2], y = loc[idx * 2 + 1], aw = attn[idx];
  const float him = y * (float)H - 0.5f, wim = x * (float)W - 0.5f;
  float g_x = 0.f, g_y = 0.f, g_a = 0.f;
  if (him > -1.f && wim > -1.f && him < (float)H && wim < (float)W) {
    const float hf = floorf(him), wf = floorf(wim);
    const int h0 = (int)hf, w0 = (int)wf, h1 = h0 + 1, w1 = w0 + 1;
    const float lh = him - hf, lw = wim - wf, hh = 1.f - lh, hw = 1.f - lw;
    const bool t_ok = h0 >= 0, b_ok = h1 <= H - 1, l_ok = w0 >= 0, r_ok = w1 <= W - 1;
    const long long vbase = ((long long)n * S + lv.start[l]) * M * D + (long long)m * D;
    const long long rs = (long long)M * D;   // stride between pixels
    const long long o00 = vbase + ((long long)h0 * W + w0) * rs, o01 = o00 + rs, o10 = o00 + (long long)W * rs,
                    o11 = o10 + rs;
    const float* go = grad_out + (nq * M + m) * D;
    float d_h = 0.f, d_w = 0.f;   // d(sample)/d(him), d(sample)/d(wim), contracted with grad_out
    for (int c = 0; c < D; ++c) {
      const float g = go[c];
      const float v00 = (t_ok && l_ok) ? value[o00 + c] : 0.f;
      const float v01 = (t_ok && r_ok) ? value[o01 + c] : 0.f;
      const float v10 = (b_ok && l_ok) ? value[o10 + c] : 0.f;
      const float v11 = (b_ok && r_ok) ? value[o11 + c] : 0.f;
      g_a += g * (hh * hw * v00 + hh * lw * v01 + lh * hw * v10 + lh * lw * v11);
      // (single subtractions: hipcc's packed pair of differences is the form common.h describes)
      d_h += g * (hw * sub_single(v10, v00) + lw * sub_single(v11, v01));
      d_w += g * (hh * sub_single(v01, v00) + lh * sub_single(v11, v10));
      const float ga = g * aw;
      if (t_ok && l_ok) atomicAdd(grad_value + o00 + c, ga * hh * hw);
      if (t_ok && r_ok) atomicAdd(grad_value + o01 + c, ga * hh * lw);
      if (b_ok && l_ok) atomicAdd(grad_value + o10 + c, ga * lh * hw);
      if (b_ok && r_ok) atomicAdd(grad_value + o11 + c, ga * lh * lw);
    }
    g_x = aw * d_w * (float)W;   // wim = x * W - 1/2
    g_y = aw * d_h * (float)H;
  }
  grad_loc[idx * 2] = g_x;
  grad_loc[idx * 2 + 1] = g_y;
  grad_attn[idx] = g_a;
}

int msda_backward_f32(const float* value, const LevelTable& lv, const float* loc, const float* attn,
                      const float* grad_out, int N, int S, int M, int D, int L, int Lq, int P, float* grad_value,
                      float* grad_loc, float* grad_attn, hipStream_t st) {
  if (hipMemsetAsync(grad_value, 0, sizeof(float) * (size_t)N * S * M * D, st) != hipSuccess) {
    set_error("msda_backward_f32: memset failed");
    (void)hipGetLastError();
    return UNIVS_ERR_LAUNCH;
  }
  const long long total = (long long)N * Lq * M * L * P;
  if (total == 0) return UNIVS_OK;
  const long long blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) {
    set_error("msda_backward_f32: problem too large");
    return UNIVS_ERR_INVALID_ARGUMENT;
  }
  hipLaunchKernelGGL(msda_bwd_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, st, value, lv, loc, attn, grad_out, N,
                     S, M, D, L, Lq, P, grad_value, grad_loc, grad_attn, total);
  return check_launch("msda_bwd_f32");
}

}  // namespace univs
