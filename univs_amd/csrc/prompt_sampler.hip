// The visual-prompt sampler of a prompted clip as a handful of kernels (the host path was ~330 ATen micro-launches per clip:
// compares, reductions, scans, searches, gathers -- the clip every video runs was host-bound on them).
//
// What is computed (reference: univs/modeling/prompt_encoder/prompt_encoder.py, VisualPromptEncoder):
//   * get_mask_prompt (:168-263) for the F key frames x n entities of a clip at once, i = f * n + e:
//       - select_points_from_box_mask, mask branch (:362-442): candidate pixels = the central half of the box where the mask reaches
//         min(max, 0.75); entities without such a pixel fall back to their pixels >= min(max, 0.95); ONE candidate is drawn;
//       - the feature-resolution mask (nearest, 1 / scale), its binary form (>= min(max over the frame, 0.5)), validity (max > 0.5);
//       - get_dense_features (:445-497): R pixels of the binary feature mask -- a random R-subset when it has >= R pixels, its pixels
//         cyclically when it has fewer, the pooled token when it is empty -- and the features / position embeddings there;
//       - the cross-attention mask: everything outside the box, at the key frame only (comm.py:6-39 convert_box_to_mask).
// The draws themselves stay with torch's generators (the host passes uniform numbers, or explicit ranks in the reference's own
// random stream): these kernels turn them into pixels.  Bit-exact against the ATen formulation of univs_amd/modeling/prompt_encoder.py
// (tests/test_prompt_sampler_gpu.py), which stays the CPU path and the GPU oracle.
#include "common.h"

#include <algorithm>

// every product and sum below is rounded on its own, as the separate ATen kernels of the host formulation round them (hipcc
// contracts a * b + c into one fused multiply-add by default: a pixel centre on the border of a box's central half then falls
// on the other side)
#pragma clang fp contract(off)

namespace univs {

namespace {

constexpr int PS_NT = 512;                 // threads of the per-entity kernels
constexpr int PS_NW = PS_NT / 64;

// floats ordered as unsigned keys (0 = "nothing seen": below every float)
__device__ __forceinline__ unsigned ps_key(float v) {
  const unsigned b = __float_as_uint(v);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ps_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__device__ __forceinline__ unsigned ps_wave_max(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
  return v;
}
__device__ __forceinline__ int ps_wave_sum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// cxcywh of a normalised xyxy box, as box_xyxy_to_cxcywh (comm.py) computes it in fp32
struct PsBox {
  float cx, cy, bw, bh;
};
__device__ __forceinline__ PsBox ps_box(const float* __restrict__ boxes, int i) {
  const float x0 = boxes[4 * i], y0 = boxes[4 * i + 1], x1 = boxes[4 * i + 2], y1 = boxes[4 * i + 3];
  return PsBox{(x0 + x1) * 0.5f, (y0 + y1) * 0.5f, x1 - x0, y1 - y0};
}
// pixel centre k of an axis of n pixels inside the central half of the box: |(k + 0.5) / n - c| < 0.25 extent.  (ATen's GPU division of
// a tensor by a host scalar multiplies by the scalar's fp32 reciprocal -- BinaryDivTrueKernel.cu -- and so does this: `inv` = 1 / n.)
__device__ __forceinline__ bool ps_central(int k, float inv, float c, float quarter) {
  const float p = ((float)k + 0.5f) * inv;
  return fabsf(p - c) < quarter;
}

constexpr int PS_ROWS = 16;                // image rows per workgroup of the two passes over the masks (4 waves x 4 rows)

// ---- pass 1 over the masks: per entity the maximum and the maximum inside the central half of the box; the feature-resolution
// mask (nearest: pixel (s y, s x)) and its maximum per frame.  A wave walks whole image rows (16-byte loads when the row allows);
// one atomic per statistic and workgroup.
template <bool VEC>
__global__ __launch_bounds__(256) void ps_mask_stats(const float* __restrict__ masks, const float* __restrict__ boxes, int N, int n, int h,
                                                     int w, int s, float* __restrict__ feat_masks, unsigned* __restrict__ stats) {
  __shared__ unsigned red[3][4];
  const int i = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const PsBox b = ps_box(boxes, i);
  const float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h, qw = 0.25f * b.bw, qh = 0.25f * b.bh;
  const int w_img = w / s;
  unsigned mx = 0u, mxc = 0u, fmx = 0u;
  for (int q = 0; q < PS_ROWS / 4; ++q) {
    const int y = blockIdx.x * PS_ROWS + wave * (PS_ROWS / 4) + q;
    if (y >= h) break;
    const bool in_y = ps_central(y, inv_h, b.cy, qh);
    const float* row = masks + ((size_t)i * h + y) * w;
    if constexpr (VEC) {
      for (int x = 4 * lane; x < w; x += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + x);
        const unsigned k0 = ps_key(v.x), k1 = ps_key(v.y), k2 = ps_key(v.z), k3 = ps_key(v.w);
        mx = max(max(mx, max(k0, k1)), max(k2, k3));
        if (in_y) {
          if (ps_central(x, inv_w, b.cx, qw)) mxc = max(mxc, k0);
          if (ps_central(x + 1, inv_w, b.cx, qw)) mxc = max(mxc, k1);
          if (ps_central(x + 2, inv_w, b.cx, qw)) mxc = max(mxc, k2);
          if (ps_central(x + 3, inv_w, b.cx, qw)) mxc = max(mxc, k3);
        }
      }
    } else {
      for (int x = lane; x < w; x += 64) {
        const unsigned k = ps_key(row[x]);
        mx = max(mx, k);
        if (in_y && ps_central(x, inv_w, b.cx, qw)) mxc = max(mxc, k);
      }
    }
    if (y % s == 0) {                                          // a row of the feature-resolution mask (the row was just read: cache hits)
      float* frow = feat_masks + ((size_t)i * (h / s) + y / s) * w_img;
      for (int xi = lane; xi < w_img; xi += 64) {
        const float v = row[(size_t)xi * s];
        frow[xi] = v;
        fmx = max(fmx, ps_key(v));
      }
    }
  }
  mx = ps_wave_max(mx);
  mxc = ps_wave_max(mxc);
  fmx = ps_wave_max(fmx);
  if (lane == 0) {
    red[0][wave] = mx;
    red[1][wave] = mxc;
    red[2][wave] = fmx;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const unsigned v = max(max(red[threadIdx.x][0], red[threadIdx.x][1]), max(red[threadIdx.x][2], red[threadIdx.x][3]));
    if (v) atomicMax(&stats[threadIdx.x == 0 ? i : threadIdx.x == 1 ? N + i : 2 * N + i / n], v);
  }
}

// ---- pass 2: the candidate pixels of every entity and their count per image row
template <bool VEC>
__global__ __launch_bounds__(256) void ps_candidates(const float* __restrict__ masks, const float* __restrict__ boxes,
                                                     const unsigned* __restrict__ stats, int N, int h, int w, uint8_t* __restrict__ sel,
                                                     int* __restrict__ rowcnt) {
  const int i = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const PsBox b = ps_box(boxes, i);
  const float inv_w = 1.0f / (float)w, inv_h = 1.0f / (float)h, qw = 0.25f * b.bw, qh = 0.25f * b.bh;
  const float mx = ps_unkey(stats[i]);
  const unsigned mxc_k = stats[N + i];
  const float t_ctr = fminf(mx, 0.75f), t_hi = fminf(mx, 0.95f);
  const bool any_ctr = mxc_k != 0u && ps_unkey(mxc_k) >= t_ctr;   // some pixel of the central half reaches the threshold
  for (int q = 0; q < PS_ROWS / 4; ++q) {
    const int y = blockIdx.x * PS_ROWS + wave * (PS_ROWS / 4) + q;
    if (y >= h) break;
    const bool in_y = ps_central(y, inv_h, b.cy, qh);
    const float* row = masks + ((size_t)i * h + y) * w;
    uint8_t* srow = sel + ((size_t)i * h + y) * w;
    int cnt = 0;
    if constexpr (VEC) {
      for (int x = 4 * lane; x < w; x += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + x);
        bool c0, c1, c2, c3;
        if (any_ctr) {
          c0 = in_y && ps_central(x, inv_w, b.cx, qw) && v.x >= t_ctr;
          c1 = in_y && ps_central(x + 1, inv_w, b.cx, qw) && v.y >= t_ctr;
          c2 = in_y && ps_central(x + 2, inv_w, b.cx, qw) && v.z >= t_ctr;
          c3 = in_y && ps_central(x + 3, inv_w, b.cx, qw) && v.w >= t_ctr;
        } else {
          c0 = v.x >= t_hi;
          c1 = v.y >= t_hi;
          c2 = v.z >= t_hi;
          c3 = v.w >= t_hi;
        }
        *reinterpret_cast<unsigned*>(srow + x) = (c0 ? 1u : 0u) | (c1 ? 0x100u : 0u) | (c2 ? 0x10000u : 0u) | (c3 ? 0x1000000u : 0u);
        cnt += (int)c0 + (int)c1 + (int)c2 + (int)c3;
      }
    } else {
      for (int x = lane; x < w; x += 64) {
        const float v = row[x];
        const bool c = any_ctr ? (in_y && ps_central(x, inv_w, b.cx, qw) && v >= t_ctr) : (v >= t_hi);
        srow[x] = c ? 1 : 0;
        cnt += c ? 1 : 0;
      }
    }
    cnt = ps_wave_sum(cnt);
    if (lane == 0) rowcnt[(size_t)i * h + y] = cnt;
  }
}

// ---- per entity: the binary feature mask, the two draw sizes, validity
__global__ __launch_bounds__(256) void ps_finalize(const float* __restrict__ feat_masks, const unsigned* __restrict__ stats,
                                                   const int* __restrict__ rowcnt, int N, int n, int h, int HW, float feat_thresh,
                                                   uint8_t* __restrict__ fmb, int* __restrict__ counts, uint8_t* __restrict__ valid,
                                                   uint8_t* __restrict__ visible) {
  __shared__ int part[2][4];
  const int i = blockIdx.x, f = i / n, e = i - f * n, tid = threadIdx.x;
  const unsigned fk = stats[2 * N + f];
  const float thr = fminf(fk ? ps_unkey(fk) : -INFINITY, feat_thresh);
  const float* fm = feat_masks + (size_t)i * HW;
  uint8_t* out = fmb + (size_t)i * HW;
  int dc = 0, pc = 0;
  for (int p = tid; p < HW; p += 256) {
    const bool on = fm[p] >= thr;
    out[p] = on ? 1 : 0;
    dc += on ? 1 : 0;
  }
  for (int y = tid; y < h; y += 256) pc += rowcnt[(size_t)i * h + y];
  dc = ps_wave_sum(dc);
  pc = ps_wave_sum(pc);
  if ((tid & 63) == 0) {
    part[0][tid >> 6] = dc;
    part[1][tid >> 6] = pc;
  }
  __syncthreads();
  if (tid == 0) {
    const float mx = ps_unkey(stats[i]);
    counts[(size_t)f * 2 * n + e] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    counts[(size_t)f * 2 * n + n + e] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
    valid[i] = mx > feat_thresh ? 1 : 0;
    visible[i] = mx > 0.f ? 1 : 0;
  }
}

// ---- searches over cumulative counts without materialising them: thread t owns the elements [t CH, (t + 1) CH); P[t] = the sum of
// everything in front of its chunk (P[PS_NT] = the total)
template <class Get>
__device__ void ps_build_prefix(Get get, int L, int CH, int* P, int* wsum) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int s = 0;
  const int k0 = tid * CH, k1 = min(L, k0 + CH);
  for (int k = k0; k < k1; ++k) s += get(k);
  int inc = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int up = __shfl_up(inc, o, 64);
    if (lane >= o) inc += up;
  }
  __syncthreads();                                   // (the previous use of P / wsum is over)
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int q = 0; q < wave; ++q) base += wsum[q];
  P[tid] = base + inc - s;
  if (tid == PS_NT - 1) P[PS_NT] = base + inc;
  __syncthreads();
}
// first index whose inclusive cumulative count reaches q (q >= 1), or L when the total is smaller; `before` = the cumulative count
// in front of that index (for L: in front of index L - 1, what torch.gather(rc, y - 1) reads after the clamp)
template <class Get>
__device__ void ps_kth(Get get, int L, int CH, const int* P, int q, int& pos, int& before) {
  if (q > P[PS_NT]) {
    pos = L;
    before = L >= 1 ? P[PS_NT] - get(L - 1) : 0;
    return;
  }
  int lo = 0, hi = PS_NT - 1;                        // the last thread whose chunk starts below q
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (P[mid] < q) lo = mid;
    else hi = mid - 1;
  }
  int run = P[lo];
  const int k1 = min(L, (lo + 1) * CH);
  pos = L;
  before = run;
  for (int k = lo * CH; k < k1; ++k) {
    const int v = get(k);
    if (run + v >= q) {
      pos = k;
      before = run;
      return;
    }
    run += v;
  }
}

// ---- per entity: the drawn candidate pixel and the R pixels of the dense tokens.
// mode 0 (device draws): u [N] uniform in [0, 1) and keys [N, HW] uniform -- the point's rank = floor(u count), the dense pixels = the
// R largest keys of the mask's pixels, in decreasing order (equal keys: the lower pixel first); mode 1 (explicit ranks): tab [N, R + 2]
// int64 = R dense ranks, an "empty" flag, the point's rank.
__global__ __launch_bounds__(PS_NT) void ps_draw(const uint8_t* __restrict__ sel, const int* __restrict__ rowcnt,
                                                 const uint8_t* __restrict__ fmb, const int* __restrict__ counts,
                                                 const float* __restrict__ u, const float* __restrict__ keys,
                                                 const long long* __restrict__ tab, int n, int h, int w, int HW, int R,
                                                 long long* __restrict__ point_idx, long long* __restrict__ dense_idx,
                                                 uint8_t* __restrict__ empty, float* __restrict__ point_coords) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ps_lds[];
  int* P = reinterpret_cast<int*>(ps_lds);                    // [PS_NT + 1]
  int* wsum = P + PS_NT + 1;                                   // [PS_NW]
  int* bc = wsum + PS_NW;                                      // [4] broadcast slots
  float* wkey = reinterpret_cast<float*>(bc + 4);              // [PS_NW]
  int* widx = reinterpret_cast<int*>(wkey + PS_NW);            // [PS_NW]
  float* kl = reinterpret_cast<float*>(widx + PS_NW + 3);      // [HW] (mode 0)
  const int i = blockIdx.x, f = i / n, e = i - f * n, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cnt = counts[(size_t)f * 2 * n + e], dcnt = counts[(size_t)f * 2 * n + n + e];
  const bool explicit_ranks = tab != nullptr;

  // ---- the point: rank among the candidates -> image row -> column
  long long rank;
  if (explicit_ranks) rank = tab[(size_t)i * (R + 2) + R + 1];
  else {
    const long long c = max(cnt, 1);
    rank = (long long)(u[i] * (float)c);                      // (u * cnt).long(): the product in fp32, truncated
    rank = min(rank, c - 1);
  }
  const int r1 = (int)(rank + 1);
  const int* rc = rowcnt + (size_t)i * h;
  auto get_row = [&](int k) { return rc[k]; };
  ps_build_prefix(get_row, h, (h + PS_NT - 1) / PS_NT, P, wsum);
  if (tid == 0) {
    int pos, before;
    ps_kth(get_row, h, (h + PS_NT - 1) / PS_NT, P, r1, pos, before);
    bc[0] = min(pos, h - 1);
    bc[1] = before;
  }
  __syncthreads();
  const int y = bc[0], before_y = bc[1];
  const uint8_t* srow = sel + ((size_t)i * h + y) * w;
  auto get_col = [&](int k) { return (int)srow[k]; };
  ps_build_prefix(get_col, w, (w + PS_NT - 1) / PS_NT, P, wsum);
  if (tid == 0) {
    int pos, before;
    ps_kth(get_col, w, (w + PS_NT - 1) / PS_NT, P, r1 - before_y, pos, before);
    const int x = min(pos, w - 1);
    point_idx[i] = (long long)y * w + x;
    point_coords[2 * i] = ((float)x + 0.5f) * (1.0f / (float)w);
    point_coords[2 * i + 1] = ((float)y + 0.5f) * (1.0f / (float)h);
  }

  // ---- the dense tokens' pixels
  const uint8_t* m = fmb + (size_t)i * HW;
  auto get_m = [&](int k) { return (int)m[k]; };
  const int CHm = (HW + PS_NT - 1) / PS_NT;
  long long* dout = dense_idx + (size_t)i * R;
  const bool top = !explicit_ranks && dcnt >= R;
  if (!top) {
    ps_build_prefix(get_m, HW, CHm, P, wsum);
    for (int r = tid; r < R; r += PS_NT) {
      const long long q = explicit_ranks ? tab[(size_t)i * (R + 2) + r] : (long long)(r % max(dcnt, 1));
      int pos, before;
      ps_kth(get_m, HW, CHm, P, (int)(q + 1), pos, before);
      dout[r] = min(pos, HW - 1);
    }
    if (tid == 0) empty[i] = explicit_ranks ? (tab[(size_t)i * (R + 2) + R] != 0 ? 1 : 0) : (dcnt == 0 ? 1 : 0);
    return;
  }
  if (tid == 0) empty[i] = 0;
  const float* kg = keys + (size_t)i * HW;
  for (int p = tid; p < HW; p += PS_NT) kl[p] = m[p] ? kg[p] : -1.0f;
  __syncthreads();
  for (int r = 0; r < R; ++r) {
    float bk = -3.0f;
    int bi = 0x7fffffff;
    for (int p = tid; p < HW; p += PS_NT) {
      const float k = kl[p];
      if (k > bk) {                                           // ascending p: the lowest pixel among equal keys
        bk = k;
        bi = p;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ok = __shfl_xor(bk, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ok > bk || (ok == bk && oi < bi)) {
        bk = ok;
        bi = oi;
      }
    }
    if (lane == 0) {
      wkey[wave] = bk;
      widx[wave] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int q = 1; q < PS_NW; ++q)
        if (wkey[q] > bk || (wkey[q] == bk && widx[q] < bi)) {
          bk = wkey[q];
          bi = widx[q];
        }
      dout[r] = bi;
      kl[bi] = -2.0f;                                          // taken
    }
    __syncthreads();
  }
}

// ---- the dense tokens: features / position embeddings at the sampled pixels of the key frame's maps (the pooled token of an empty
// mask), zero for invalid entities, replicated over the clip's T frames.  One workgroup per (entity, token).
__global__ __launch_bounds__(256) void ps_tokens(const float* __restrict__ feats, long long f_sf, long long f_sc, long long f_sp,
                                                 const float* __restrict__ pos, long long p_sf, long long p_sc, long long p_sp,
                                                 const float* __restrict__ qfeat, const float* __restrict__ qpe,
                                                 const long long* __restrict__ dense_idx, const uint8_t* __restrict__ empty,
                                                 const uint8_t* __restrict__ valid, int n, int R, int T, int C,
                                                 float* __restrict__ fd, float* __restrict__ pd) {
  const int ir = blockIdx.x, i = ir / R, f = i / n;
  const long long p = dense_idx[ir];
  const bool em = empty[i] != 0;
  const float vf = valid[i] ? 1.0f : 0.0f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float a = (em ? qfeat[(size_t)i * C + c] : feats[f * f_sf + c * f_sc + p * f_sp]) * vf;
    const float b = (em ? qpe[(size_t)i * C + c] : pos[f * p_sf + c * p_sc + p * p_sp]) * vf;
    for (int t = 0; t < T; ++t) {
      fd[((size_t)ir * T + t) * C + c] = a;
      pd[((size_t)ir * T + t) * C + c] = b;
    }
  }
}

// ---- cross-attention masks [F, T, 1, n, HW]: at the key frame everything outside the box (convert_box_to_mask: x in (floor(x0 w),
// ceil(x1 w)], y in (floor(y0 h), ceil(y1 h)]) of a valid entity; False elsewhere
__global__ __launch_bounds__(256) void ps_attn_mask(const float* __restrict__ boxes, const uint8_t* __restrict__ valid,
                                                    const long long* __restrict__ kf, int n, int T, int h_img, int w_img,
                                                    uint8_t* __restrict__ attn) {
  const int HW = h_img * w_img;
  const int row = blockIdx.y;                                   // (f, t, e)
  const int e = row % n, t = (row / n) % T, f = row / (n * T);
  const int i = f * n + e;
  const bool live = valid[i] != 0 && kf[f] == t;
  const float bx0 = floorf(boxes[4 * i] * (float)w_img), by0 = floorf(boxes[4 * i + 1] * (float)h_img);
  const float bx1 = ceilf(boxes[4 * i + 2] * (float)w_img), by1 = ceilf(boxes[4 * i + 3] * (float)h_img);
  uint8_t* out = attn + (size_t)row * HW;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const float gx = (float)(p % w_img), gy = (float)(p / w_img);
    const bool inside = gx > bx0 && gx <= bx1 && gy > by0 && gy <= by1;
    out[p] = (live && !inside) ? 1 : 0;
  }
}

// ---- the position token of every sampled point: sin / cos embeddings of (y, x) at the frequencies dim_t [F] and of the frame
// coordinate z at the frequencies dim_tz [2 F], channel 2 k = sin, 2 k + 1 = cos (position_encoding.py:_points: cat(pos_y, pos_x) + pos_z;
// univs/modeling/transformer_decoder/position_encoding.py:170-236).  Same operations in the same order as the ATen formulation
// (scale, divide, sin / cos, add), so the same bits.
__global__ __launch_bounds__(256) void ps_point_pe(const float* __restrict__ xy, const float* __restrict__ z, const float* __restrict__ dim_t,
                                                   const float* __restrict__ dim_tz, float scale, int n, int F, float* __restrict__ out) {
  const int i = blockIdx.x, f = i / n;
  const float px = xy[2 * i] * scale, py = xy[2 * i + 1] * scale, pz = z[f];
  for (int c = threadIdx.x; c < 2 * F; c += 256) {
    const int k = c < F ? c : c - F;
    const float a = (c < F ? py : px) / dim_t[k];
    const float b = pz / dim_tz[c];
    const float v = (c & 1) ? cosf(a) : sinf(a);
    const float u = (c & 1) ? cosf(b) : sinf(b);
    out[(size_t)i * 2 * F + c] = v + u;
  }
}

// ---- mean over the non-blank tokens of every (entity, frame): x [n, L, T, C] -> out [n, T, C] = sum_l x / max(1, #{l : x[., l, ., :] != 0})
// (+ add [C]); ...decoder_univs.py:640-650.  One workgroup per (entity, frame), a thread per channel.
__global__ __launch_bounds__(256) void ps_token_mean(const float* __restrict__ x, const float* __restrict__ add, int L, int T, int C,
                                                     float* __restrict__ out) {
  extern __shared__ unsigned char tm_flag[];                      // [4][L]: wave w saw a non-zero channel of token l
  const int e = blockIdx.x / T, t = blockIdx.x - e * T, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                           // channels threadIdx.x + 256 q (C <= 1024)
  for (int l = 0; l < L; ++l) {
    bool nz = false;
    const float* row = x + (((size_t)e * L + l) * T + t) * C;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = threadIdx.x + 256 * q;
      if (c < C) {
        const float v = row[c];
        acc[q] += v;
        nz = nz || v != 0.f;
      }
    }
    const unsigned long long any = __ballot(nz);
    if (lane == 0) tm_flag[wave * L + l] = any != 0ull ? 1 : 0;
  }
  __syncthreads();
  int cnt = 0;
  for (int l = 0; l < L; ++l) cnt += (tm_flag[l] | tm_flag[L + l] | tm_flag[2 * L + l] | tm_flag[3 * L + l]) ? 1 : 0;
  const float d = (float)max(cnt, 1);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = threadIdx.x + 256 * q;
    if (c < C) {
      const float m = acc[q] / d;
      out[((size_t)e * T + t) * C + c] = add ? m + add[c] : m;
    }
  }
}

}  // namespace

int prompt_prefix_f32(const float* masks, const float* boxes, int Fk, int n, int h, int w, int scale, float feat_thresh,
                      float* feat_masks, unsigned* stats, uint8_t* sel, int* rowcnt, uint8_t* fmb, int* counts, uint8_t* valid,
                      uint8_t* visible, hipStream_t st) {
  const int N = Fk * n;
  dim3 grid((h + PS_ROWS - 1) / PS_ROWS, N);
  const bool vec = w % 4 == 0 && (reinterpret_cast<uintptr_t>(masks) & 15) == 0 && (reinterpret_cast<uintptr_t>(sel) & 3) == 0;
  if (vec) {
    hipLaunchKernelGGL(ps_mask_stats<true>, grid, dim3(256), 0, st, masks, boxes, N, n, h, w, scale, feat_masks, stats);
    hipLaunchKernelGGL(ps_candidates<true>, grid, dim3(256), 0, st, masks, boxes, stats, N, h, w, sel, rowcnt);
  } else {
    hipLaunchKernelGGL(ps_mask_stats<false>, grid, dim3(256), 0, st, masks, boxes, N, n, h, w, scale, feat_masks, stats);
    hipLaunchKernelGGL(ps_candidates<false>, grid, dim3(256), 0, st, masks, boxes, stats, N, h, w, sel, rowcnt);
  }
  hipLaunchKernelGGL(ps_finalize, dim3(N), dim3(256), 0, st, feat_masks, stats, rowcnt, N, n, h, (h / scale) * (w / scale), feat_thresh, fmb,
                     counts, valid, visible);
  return check_launch("prompt_prefix_f32");
}

int prompt_draw(const uint8_t* sel, const int* rowcnt, const uint8_t* fmb, const int* counts, const float* u, const float* keys,
                const long long* tab, int Fk, int n, int h, int w, int HW, int R, long long* point_idx, long long* dense_idx,
                uint8_t* empty, float* point_coords, hipStream_t st) {
  const size_t lds = (size_t)(PS_NT + 1 + PS_NW + 4 + PS_NW + PS_NW + 3) * 4 + (tab ? 0 : (size_t)HW * 4);
  if (lds > 150 * 1024) return 0;
  // (per launch, not once per process: the attribute belongs to the function ON THE CURRENT DEVICE)
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ps_draw), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipLaunchKernelGGL(ps_draw, dim3(Fk * n), dim3(PS_NT), lds, st, sel, rowcnt, fmb, counts, u, keys, tab, n, h, w, HW, R, point_idx,
                     dense_idx, empty, point_coords);
  const int rc = check_launch("prompt_draw");
  return rc == UNIVS_OK ? 1 : rc;
}

int prompt_tokens_f32(const float* feats, const long long* fs, const float* pos, const long long* ps, const float* qfeat,
                      const float* qpe, const long long* dense_idx, const uint8_t* empty, const uint8_t* valid, const float* boxes,
                      const long long* kf, int Fk, int n, int R, int T, int C, int h_img, int w_img, float* fd, float* pd, uint8_t* attn,
                      hipStream_t st) {
  const int N = Fk * n;
  hipLaunchKernelGGL(ps_tokens, dim3(N * R), dim3(256), 0, st, feats, fs[0], fs[1], fs[2], pos, ps[0], ps[1], ps[2], qfeat, qpe,
                     dense_idx, empty, valid, n, R, T, C, fd, pd);
  const int HW = h_img * w_img;
  hipLaunchKernelGGL(ps_attn_mask, dim3(std::min((HW + 255) / 256, 16), Fk * T * n), dim3(256), 0, st, boxes, valid, kf, n, T, h_img,
                     w_img, attn);
  return check_launch("prompt_tokens_f32");
}

}  // namespace univs

namespace univs {

int prompt_point_pe_f32(const float* xy, const float* z, const float* dim_t, const float* dim_tz, float scale, int Fk, int n, int F,
                        float* out, hipStream_t st) {
  hipLaunchKernelGGL(ps_point_pe, dim3(Fk * n), dim3(256), 0, st, xy, z, dim_t, dim_tz, scale, n, F, out);
  return check_launch("prompt_point_pe_f32");
}

int token_mean_f32(const float* x, const float* add, int n, int L, int T, int C, float* out, hipStream_t st) {
  if (C > 1024 || 4 * L > 60 * 1024) return 0;
  hipLaunchKernelGGL(ps_token_mean, dim3(n * T), dim3(256), (size_t)4 * L, st, x, add, L, T, C, out);
  const int rc = check_launch("token_mean_f32");
  return rc == UNIVS_OK ? 1 : rc;
}

}  // namespace univs
