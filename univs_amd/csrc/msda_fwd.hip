// Multi-scale deformable attention forward for gfx950 (CDNA4, wave64).
//
// Semantics follow the reference operator (mask2former/modeling/pixel_decoder/ops/src/cuda/
// ms_deform_im2col_cuda.cuh:242-304 with the bilinear helper at :38-89, equivalently
// ms_deform_attn_core_pytorch in ops/functions/ms_deform_attn_func.py:52-72):
//   out[n,q,m,:] = sum_l sum_p A[n,q,m,l,p] * bilinear(V_l[n,:,m,:], x*W_l - 0.5, y*H_l - 0.5)
// with zero contribution from corners outside the level.
//
// The design is NOT the reference's (one thread per output scalar, 4 uncoalesced dword gathers per
// sample, shapes re-read from global memory by every thread).  Here:
//   * kernel "vec4": a group of D/4 lanes owns one (n, q, head); each lane holds 4 channels, so a
//     corner gather is ONE global_load_dwordx4 per lane and the group reads one contiguous
//     D*4-byte row (a full 128-B line at D=32).  A wave64 covers all 8 heads of a query at D=32, so
//     its output store is 1 KiB contiguous.  Level table lives in SGPRs (kernarg).  Blocks are
//     XCD-remapped so each private L2 sees one contiguous band of queries (neighbouring queries
//     re-use the same value rows).
//   * kernel "tiled" (msda_tiled.hip): encoder self-attention geometry, value tiles staged in LDS.
//   * kernel "scalar": any D / double precision, one thread per output element (KAT shapes of
//     ops/test.py use D=2).
#include "msda_common.h"

namespace univs {

// ------------------------------------------------------------------------------------------------
// scalar fallback (any D, float or double)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void msda_fwd_scalar(const T* __restrict__ value, LevelTable lv,
                                                        const T* __restrict__ loc,
                                                        const T* __restrict__ attn, int N, int S,
                                                        int M, int D, int L, int Lq, int P,
                                                        T* __restrict__ out) {
  const long long total = (long long)N * Lq * M * D;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % D);
    const long long item = idx / D;  // (n, q, m)
    const int m = (int)(item % M);
    const int n = (int)(item / ((long long)Lq * M));
    const T* lp = loc + item * L * P * 2;
    const T* ap = attn + item * L * P;
    const T* vb = value + (long long)n * S * M * D + (long long)m * D + c;
    const long long row = (long long)M * D;
    T acc = 0;
    for (int l = 0; l < L; ++l) {
      const int H = lv.H[l], W = lv.W[l];
      const T* vl = vb + (long long)lv.start[l] * row;
      for (int p = 0; p < P; ++p) {
        const T x = lp[(l * P + p) * 2], y = lp[(l * P + p) * 2 + 1];
        const T w = ap[l * P + p];
        const T him = y * H - (T)0.5, wim = x * W - (T)0.5;
        if (him > -1 && wim > -1 && him < H && wim < W) {
          const int h0 = (int)floor(him), w0 = (int)floor(wim);
          const T lh = him - h0, lw = wim - w0, hh = 1 - lh, hw = 1 - lw;
          T v1 = 0, v2 = 0, v3 = 0, v4 = 0;
          if (h0 >= 0 && w0 >= 0) v1 = vl[((long long)h0 * W + w0) * row];
          if (h0 >= 0 && w0 + 1 <= W - 1) v2 = vl[((long long)h0 * W + w0 + 1) * row];
          if (h0 + 1 <= H - 1 && w0 >= 0) v3 = vl[((long long)(h0 + 1) * W + w0) * row];
          if (h0 + 1 <= H - 1 && w0 + 1 <= W - 1) v4 = vl[((long long)(h0 + 1) * W + w0 + 1) * row];
          acc += (hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4) * w;
        }
      }
    }
    out[idx] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// vec4 direct-gather kernel (float, D % 4 == 0, D/4 in {1,2,4,8,16})
// ------------------------------------------------------------------------------------------------
// NS samples of one level: all 4*NS gathers are issued before the first use.
template <int NS>
__device__ __forceinline__ float4 gather_level(const float4* __restrict__ vl, int rowf4, int H, int W,
                                               const float* xs, const float* ys, const float* aws,
                                               float4 acc) {
  Footprint f[NS];
  float4 v[NS][4];
#pragma unroll
  for (int s = 0; s < NS; ++s) f[s] = footprint(H, W, xs[s], ys[s], aws[s]);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    v[s][0] = vl[(long long)(f[s].h0 * W + f[s].w0) * rowf4];
    v[s][1] = vl[(long long)(f[s].h0 * W + f[s].w1) * rowf4];
    v[s][2] = vl[(long long)(f[s].h1 * W + f[s].w0) * rowf4];
    v[s][3] = vl[(long long)(f[s].h1 * W + f[s].w1) * rowf4];
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    acc = fma4(f[s].w00, v[s][0], acc);
    acc = fma4(f[s].w01, v[s][1], acc);
    acc = fma4(f[s].w10, v[s][2], acc);
    acc = fma4(f[s].w11, v[s][3], acc);
  }
  return acc;
}

// LANES = D/4 lanes per (n,q,m) item; ITEMS_PER_GROUP consecutive passes per block so one block
// covers a run of neighbouring queries (L1 reuse of value rows).
template <int LANES, int L_CT, int P_CT, int PASSES>
__global__ __launch_bounds__(256) void msda_fwd_vec4(const float* __restrict__ value, LevelTable lv,
                                                      const float* __restrict__ loc,
                                                      const float* __restrict__ attn, int N, int S,
                                                      int M, int L, int Lq, int P,
                                                      float* __restrict__ out, unsigned nblocks) {
  constexpr int GROUPS = 256 / LANES;
  const int D = LANES * 4;
  const unsigned bid = xcd_remap(blockIdx.x, nblocks);
  const int g = threadIdx.x / LANES, lane = threadIdx.x % LANES;
  const long long items = (long long)N * Lq * M;
  const int rowf4 = M * LANES;
  const int Lr = L_CT > 0 ? L_CT : L, Pr = P_CT > 0 ? P_CT : P;

#pragma unroll 1
  for (int pass = 0; pass < PASSES; ++pass) {
    const long long item = ((long long)bid * PASSES + pass) * GROUPS + g;
    if (item >= items) return;
    const int m = (int)(item % M);
    const int n = (int)(item / ((long long)Lq * M));
    const float4* vb =
        reinterpret_cast<const float4*>(value + (long long)n * S * M * D + (long long)m * D) + lane;
    const float* lp = loc + item * (long long)(Lr * Pr * 2);
    const float* ap = attn + item * (long long)(Lr * Pr);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

    if constexpr (L_CT > 0 && P_CT == 4) {
      // compile-time geometry: fetch all locations/weights up front as 16-B loads
      float4 lc[L_CT * 2];
      float4 aw[L_CT];
#pragma unroll
      for (int l = 0; l < L_CT; ++l) {
        lc[2 * l] = reinterpret_cast<const float4*>(lp)[2 * l];
        lc[2 * l + 1] = reinterpret_cast<const float4*>(lp)[2 * l + 1];
        aw[l] = reinterpret_cast<const float4*>(ap)[l];
      }
#pragma unroll
      for (int l = 0; l < L_CT; ++l) {
        const int H = lv.H[l], W = lv.W[l];
        const float4* vl = vb + (long long)lv.start[l] * rowf4;
        const float xs[4] = {lc[2 * l].x, lc[2 * l].z, lc[2 * l + 1].x, lc[2 * l + 1].z};
        const float ys[4] = {lc[2 * l].y, lc[2 * l].w, lc[2 * l + 1].y, lc[2 * l + 1].w};
        const float ws[4] = {aw[l].x, aw[l].y, aw[l].z, aw[l].w};
        acc = gather_level<4>(vl, rowf4, H, W, xs, ys, ws, acc);
      }
    } else {
      for (int l = 0; l < Lr; ++l) {
        const int H = lv.H[l], W = lv.W[l];
        const float4* vl = vb + (long long)lv.start[l] * rowf4;
        for (int p = 0; p < Pr; ++p) {
          const float2 xy = reinterpret_cast<const float2*>(lp)[l * Pr + p];
          const float w1 = ap[l * Pr + p];
          acc = gather_level<1>(vl, rowf4, H, W, &xy.x, &xy.y, &w1, acc);
        }
      }
    }
    reinterpret_cast<float4*>(out + item * D)[lane] = acc;
  }
}

template <int LANES>
static int launch_vec4(const float* value, const LevelTable& lv, const float* loc,
                       const float* attn, int N, int S, int M, int L, int Lq, int P, float* out,
                       hipStream_t st) {
  constexpr int GROUPS = 256 / LANES;
  constexpr int PASSES = 4;
  const long long items = (long long)N * Lq * M;
  const long long per_block = (long long)GROUPS * PASSES;
  const unsigned nblocks = (unsigned)((items + per_block - 1) / per_block);
  if (L == 3 && P == 4)
    hipLaunchKernelGGL((msda_fwd_vec4<LANES, 3, 4, PASSES>), dim3(nblocks), dim3(256), 0, st, value,
                       lv, loc, attn, N, S, M, L, Lq, P, out, nblocks);
  else if (L == 4 && P == 4)
    hipLaunchKernelGGL((msda_fwd_vec4<LANES, 4, 4, PASSES>), dim3(nblocks), dim3(256), 0, st, value,
                       lv, loc, attn, N, S, M, L, Lq, P, out, nblocks);
  else
    hipLaunchKernelGGL((msda_fwd_vec4<LANES, 0, 0, PASSES>), dim3(nblocks), dim3(256), 0, st, value,
                       lv, loc, attn, N, S, M, L, Lq, P, out, nblocks);
  return check_launch("msda_fwd_vec4");
}

int msda_forward_generic_f32(const float* value, const LevelTable& lv, const float* loc,
                             const float* attn, int N, int S, int M, int D, int L, int Lq, int P,
                             float* out, hipStream_t st) {
  if (D % 4 == 0) {
    switch (D / 4) {
      case 1: return launch_vec4<1>(value, lv, loc, attn, N, S, M, L, Lq, P, out, st);
      case 2: return launch_vec4<2>(value, lv, loc, attn, N, S, M, L, Lq, P, out, st);
      case 4: return launch_vec4<4>(value, lv, loc, attn, N, S, M, L, Lq, P, out, st);
      case 8: return launch_vec4<8>(value, lv, loc, attn, N, S, M, L, Lq, P, out, st);
      case 16: return launch_vec4<16>(value, lv, loc, attn, N, S, M, L, Lq, P, out, st);
      default: break;
    }
  }
  const long long total = (long long)N * Lq * M * D;
  const unsigned nblocks = (unsigned)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256);
  hipLaunchKernelGGL(msda_fwd_scalar<float>, dim3(nblocks), dim3(256), 0, st, value, lv, loc, attn,
                     N, S, M, D, L, Lq, P, out);
  return check_launch("msda_fwd_scalar<float>");
}

int msda_forward_generic_f64(const double* value, const LevelTable& lv, const double* loc,
                             const double* attn, int N, int S, int M, int D, int L, int Lq, int P,
                             double* out, hipStream_t st) {
  const long long total = (long long)N * Lq * M * D;
  const unsigned nblocks = (unsigned)((total + 255) / 256 > 65535 * 16 ? 65535 * 16 : (total + 255) / 256);
  hipLaunchKernelGGL(msda_fwd_scalar<double>, dim3(nblocks), dim3(256), 0, st, value, lv, loc, attn,
                     N, S, M, D, L, Lq, P, out);
  return check_launch("msda_fwd_scalar<double>");
}

}  // namespace univs
