// Stand-alone reproduction of DESIGN.md section 3, hazard 23 (no PyTorch, no libunivs_hip.so):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/cohab_repro.hip -o /tmp/cohab_repro && /tmp/cohab_repro
// Stream A runs y = x * scale + bias with the (scale, bias) pair in one register pair -- hipcc's own code for it is
//   v_pk_fma_f32 v[0:1], v[0:1], v[4:5], v[4:5] op_sel:[0,0,1] op_sel_hi:[1,0,1]
// (the LOW result takes the HIGH half of src2) -- and, for comparison, the same arithmetic as four v_fma_f32.  Stream B runs a kernel that
// does nothing but v_mfma_f32_16x16x32_f16.  With B idle every run is exact; with B busy the packed form returns x * scale (the bias
// read as 0) in lanes 48..63 of the low halves, the single form stays exact.  Exit code 1 when the packed form went wrong, 0 otherwise.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(e)                                                                        \
  do {                                                                                  \
    hipError_t err_ = (e);                                                              \
    if (err_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_));      \
      exit(2);                                                                          \
    }                                                                                   \
  } while (0)

template <bool SINGLE>
__global__ __launch_bounds__(256) void affine_rows(const float* __restrict__ x, const float* __restrict__ aff, float* __restrict__ y, int R,
                                                   int C) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;            // 16 lanes share a row: its (scale, bias) is one 8-byte load
  const int r = blockIdx.y * 16 + ty, c = blockIdx.x * 64 + 4 * tx;
  if (r >= R || c >= C) return;
  float4 v = *reinterpret_cast<const float4*>(x + (long long)r * C + c);
  const float sc = aff[2 * r], bi = aff[2 * r + 1];
  if (SINGLE) {
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(o[i]) : "v"(o[i]), "v"(sc), "v"(bi));
    v = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    v = make_float4(fmaf(v.x, sc, bi), fmaf(v.y, sc, bi), fmaf(v.z, sc, bi), fmaf(v.w, sc, bi));
  }
  *reinterpret_cast<float4*>(y + (long long)r * C + c) = v;
}

__global__ __launch_bounds__(512) void only_mfma(int iters, float* sink) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x + i); b[i] = (_Float16)(1.5f - i); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
  if (acc[0] == 12345.f) *sink = 1.f;
}

// The same arithmetic with the matrix instructions in the SAME wave (one stream, nothing else on the GPU): every lane issues 16 MFMAs, then the
// packed form on freshly loaded data, `rounds` times; mode 1: only the odd waves of a workgroup issue MFMAs (neighbours on the SIMD do, the
// wave itself does not).
template <int MODE>
__global__ __launch_bounds__(256) void mfma_then_affine(const float* __restrict__ x, const float* __restrict__ aff, float* __restrict__ y, int R, int C,
                                                        float* sink) {
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int r = blockIdx.y * 16 + ty, c = blockIdx.x * 64 + 4 * tx;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x + i); b[i] = (_Float16)(1.5f - i); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const bool mm = MODE == 0 || ((threadIdx.x >> 6) & 1);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int round = 0; round < 8; ++round) {
    if (mm) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    if (r < R && c < C && (MODE == 0 || !mm)) {
      const float* px = x + (long long)r * C + c;
      const float* pa = aff + 2 * r;
      asm volatile("" : "+v"(px), "+v"(pa));                        // (opaque: the loads and the packed FMAs stay inside the loop)
      v = *reinterpret_cast<const float4*>(px);
      const float sc = pa[0], bi = pa[1];
      v = make_float4(fmaf(v.x, sc, bi), fmaf(v.y, sc, bi), fmaf(v.z, sc, bi), fmaf(v.w, sc, bi));
      *reinterpret_cast<float4*>(y + (long long)r * C + c) = v;
    }
  }
  if (acc[0] == 12345.f) *sink = 1.f;
}

int main() {
  const int R = 1280, C = 14720, RUNS = 20;
  std::vector<float> hx((size_t)R * C), ha(2 * R), hy((size_t)R * C);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; };
  for (auto& v : hx) v = 4.0f * rnd();
  for (int r = 0; r < R; ++r) { ha[2 * r] = 1.0f + rnd(); ha[2 * r + 1] = 3.0f + rnd(); }
  float *x, *aff, *y, *sink;
  CHECK(hipMalloc(&x, hx.size() * 4));
  CHECK(hipMalloc(&aff, ha.size() * 4));
  CHECK(hipMalloc(&y, hy.size() * 4));
  CHECK(hipMalloc(&sink, 16));
  CHECK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(aff, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  hipStream_t sa, sb;
  CHECK(hipStreamCreate(&sa));
  CHECK(hipStreamCreate(&sb));
  const dim3 grid((C + 63) / 64, (R + 15) / 16), block(256);
  int packed_wrong_busy = 0;
  for (int busy = 0; busy < 2; ++busy)
    for (int single = 0; single < 2; ++single) {
      long long wrong = 0, no_bias = 0, lanes48 = 0, low_half = 0;
      int bad_runs = 0;
      for (int run = 0; run < RUNS; ++run) {
        if (busy) hipLaunchKernelGGL(only_mfma, dim3(1024), dim3(512), 0, sb, 600, sink);
        if (single) hipLaunchKernelGGL(affine_rows<true>, grid, block, 0, sa, x, aff, y, R, C);
        else hipLaunchKernelGGL(affine_rows<false>, grid, block, 0, sa, x, aff, y, R, C);
        CHECK(hipStreamSynchronize(sa));
        CHECK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipStreamSynchronize(sb));
        long long w = 0;
        for (int r = 0; r < R; ++r)
          for (int c = 0; c < C; ++c) {
            const float xv = hx[(size_t)r * C + c], got = hy[(size_t)r * C + c];
            if (got != __builtin_fmaf(xv, ha[2 * r], ha[2 * r + 1])) {
              ++w;
              no_bias += got == xv * ha[2 * r];
              lanes48 += (r % 16) % 4 == 3;                          // row r % 16 = threadIdx.x >> 4: rows 3, 7, 11, 15 are lanes 48..63
              low_half += c % 2 == 0;
            }
          }
        wrong += w;
        bad_runs += w != 0;
      }
      printf("MFMA kernel on the other stream: %-3s  %-44s %2d of %d runs wrong; %lld elements (of them: == x * scale %lld, in lanes 48..63 %lld, "
             "low half of a packed pair %lld)\n",
             busy ? "yes" : "no", single ? "four v_fma_f32 (inline asm)" : "hipcc's v_pk_fma_f32 op_sel:[0,0,1]", bad_runs, RUNS, wrong, no_bias,
             lanes48, low_half);
      if (busy && !single) packed_wrong_busy = bad_runs;
    }
  // ---- one stream: the MFMAs in the same wave / in the neighbouring waves of the same workgroup
  for (int mode = 0; mode < 2; ++mode) {
    long long wrong = 0;
    int bad_runs = 0;
    for (int run = 0; run < RUNS; ++run) {
      CHECK(hipMemset(y, 0, hy.size() * 4));
      if (mode == 0) hipLaunchKernelGGL(mfma_then_affine<0>, grid, block, 0, sa, x, aff, y, R, C, sink);
      else hipLaunchKernelGGL(mfma_then_affine<1>, grid, block, 0, sa, x, aff, y, R, C, sink);
      CHECK(hipStreamSynchronize(sa));
      CHECK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
      long long w = 0;
      for (int r = 0; r < R; ++r) {
        if (mode == 1 && (((r % 16) >> 2) & 1)) continue;                 // (rows of the waves that issue MFMAs instead)
        for (int c = 0; c < C; ++c) w += hy[(size_t)r * C + c] != __builtin_fmaf(hx[(size_t)r * C + c], ha[2 * r], ha[2 * r + 1]);
      }
      wrong += w;
      bad_runs += w != 0;
    }
    printf("one stream, one kernel, %-66s %2d of %d runs wrong; %lld elements\n",
           mode == 0 ? "every wave: 16 MFMAs, then the packed form (the compiler's waits apply):" : "even waves: the packed form; odd waves of the workgroup: MFMAs only:",
           bad_runs, RUNS, wrong);
  }
  return packed_wrong_busy ? 1 : 0;
}
