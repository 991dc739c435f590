// Per-plane statistics of mask logits in ONE pass: |{x > t_hi}|, |{x > t_lo}| and the bounding box of {x > t_box} over the valid
// region [0, hv) x [0, wv) of every [H, W] plane.
//
// The clip loop's book-keeping (univs/inference/inference_video_entity.py:452-652; univs/utils/comm.py:10-38, :104-112) asks these
// of every (query, frame) plane of a clip: `calculate_mask_quality_scores` = two compares + two sums, `convert_mask_to_box` = a compare,
// two `any` reductions, four `where` + min / max passes, a stack and a product -- ~25 ATen launches and five passes over the logits
// (118 MB per 720p clip at 100 queries), on a step that is bound by the HOST cost of its launches.  Here: three launches behind one
// call (initialise, accumulate, finish), one pass over the logits.  Integer results: counts and corners are exact whatever the order
// the atomics arrive in.
#include "common.h"

#include <limits.h>

#include <algorithm>

namespace univs {

// out[p][8] = {count_hi, count_lo, left, top, right, bottom, non-empty, 0}; corners are inclusive pixel indices, zeros for an empty
// plane (convert_mask_to_box's convention)
__global__ __launch_bounds__(256) void mask_stats_init_kernel(int* __restrict__ out, long long planes) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= planes) return;
  int* o = out + p * 8;
  o[0] = 0; o[1] = 0; o[2] = INT_MAX; o[3] = INT_MAX; o[4] = -1; o[5] = -1; o[6] = 0; o[7] = 0;
}

__global__ __launch_bounds__(256) void mask_stats_finish_kernel(int* __restrict__ out, long long planes) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= planes) return;
  int* o = out + p * 8;
  const bool ne = o[5] >= o[3];
  if (!ne) { o[2] = 0; o[3] = 0; o[4] = 0; o[5] = 0; }
  o[6] = ne ? 1 : 0;
}

__device__ __forceinline__ int wave_add(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
  return v;
}

// grid (segments, planes): a workgroup walks rows [y0, y1) of its plane, a thread 4 consecutive columns at a time (16-byte loads when
// the row pitch and the base allow, scalar otherwise)
template <bool VEC4>
__global__ __launch_bounds__(256) void mask_stats_kernel(const float* __restrict__ x, int* __restrict__ out, int W, int hv, int wv,
                                                         int rows_per_seg, float t_hi, float t_lo, float t_box, int inner,
                                                         long long stride_outer, long long stride_inner) {
  const long long p = blockIdx.y;
  const int y0 = blockIdx.x * rows_per_seg, y1 = min(hv, y0 + rows_per_seg);
  if (y0 >= y1) return;                                           // (the whole workgroup: no barrier is skipped by a part of it)
  // plane p = (outer index p / inner, inner index p % inner): a [N, T, H, W] view of a longer history has two plane strides
  const float* base = x + (p / inner) * stride_outer + (p % inner) * stride_inner;
  const int wq = (wv + 3) >> 2;                                   // column groups of 4 in the valid width
  const int n = (y1 - y0) * wq;
  int hi = 0, lo = 0, xmin = INT_MAX, ymin = INT_MAX, xmax = -1, ymax = -1;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int r = e / wq, c4 = e - r * wq;
    const int y = y0 + r, x0 = 4 * c4;
    const float* src = base + (long long)y * W + x0;
    float v[4];
    if (VEC4 && x0 + 3 < wv) {
      const float4 t = *reinterpret_cast<const float4*>(src);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = x0 + k < wv ? src[k] : -INFINITY;      // (-inf exceeds no threshold; NaN neither, as in ATen)
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi += v[k] > t_hi;
      lo += v[k] > t_lo;
      if (v[k] > t_box) {
        xmin = min(xmin, x0 + k);
        xmax = max(xmax, x0 + k);
        ymin = min(ymin, y);
        ymax = max(ymax, y);
      }
    }
  }
  hi = wave_add(hi);
  lo = wave_add(lo);
  xmin = wave_min(xmin);
  ymin = wave_min(ymin);
  xmax = wave_max(xmax);
  ymax = wave_max(ymax);
  // the four waves meet in LDS: ONE set of atomics per workgroup (with one per wave, a plane's ~2 000 atomics on a single cache line took
  // longer than the pass over its pixels when few planes are cut into many segments)
  __shared__ int part[4][6];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    part[wave][0] = hi; part[wave][1] = lo; part[wave][2] = xmin; part[wave][3] = ymin; part[wave][4] = xmax; part[wave][5] = ymax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      hi += part[k][0];
      lo += part[k][1];
      xmin = min(xmin, part[k][2]);
      ymin = min(ymin, part[k][3]);
      xmax = max(xmax, part[k][4]);
      ymax = max(ymax, part[k][5]);
    }
    int* o = out + p * 8;
    if (hi) atomicAdd(o + 0, hi);
    if (lo) atomicAdd(o + 1, lo);
    if (ymax >= 0) {
      atomicMin(o + 2, xmin);
      atomicMin(o + 3, ymin);
      atomicMax(o + 4, xmax);
      atomicMax(o + 5, ymax);
    }
  }
}

// plane p of the input starts at x + (p / inner) * stride_outer + (p % inner) * stride_inner floats (dense planes: inner = planes, stride_inner =
// H * W).  Returns UNIVS_OK, or UNIVS_ERR_NOT_IMPLEMENTED for more than 65 535 planes
int mask_stats_f32(const float* x, long long planes, int inner, long long stride_outer, long long stride_inner, int H, int W, int hv, int wv,
                   float t_hi, float t_lo, float t_box, int* out, hipStream_t st) {
  if (planes <= 0) return UNIVS_OK;
  if (planes > 65535 || inner < 1) return UNIVS_ERR_NOT_IMPLEMENTED;
  const unsigned pb = (unsigned)((planes + 255) / 256);
  hipLaunchKernelGGL(mask_stats_init_kernel, dim3(pb), dim3(256), 0, st, out, planes);
  if (hv > 0 && wv > 0) {
    // segments: enough workgroups to fill the chip (~8 per CU) when there are few planes, at least 8 rows each
    long long want = (2048 + planes - 1) / planes;
    int segs = (int)std::min<long long>(std::max<long long>(want, 1), std::max(1, hv / 8));
    const int rows_per_seg = (hv + segs - 1) / segs;
    segs = (hv + rows_per_seg - 1) / rows_per_seg;
    const bool vec4 = W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && stride_outer % 4 == 0 && stride_inner % 4 == 0;
    dim3 grid((unsigned)segs, (unsigned)planes);
    if (vec4) hipLaunchKernelGGL(mask_stats_kernel<true>, grid, dim3(256), 0, st, x, out, W, hv, wv, rows_per_seg, t_hi, t_lo, t_box, inner, stride_outer, stride_inner);
    else hipLaunchKernelGGL(mask_stats_kernel<false>, grid, dim3(256), 0, st, x, out, W, hv, wv, rows_per_seg, t_hi, t_lo, t_box, inner, stride_outer, stride_inner);
  }
  hipLaunchKernelGGL(mask_stats_finish_kernel, dim3(pb), dim3(256), 0, st, out, planes);
  return check_launch("mask_stats_f32");
}

}  // namespace univs
