// LDS gather probe for gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_gather_b128.hip -o /tmp/lds_gather && /tmp/lds_gather
// Question: a lane that owns a bilinear sample reads its four corner pixels (128 B each) from an LDS window with
// ds_read_b128, eight 16-byte chunks per pixel.  With every lane on the same chunk the 64 banks collapse onto the 8
// banks that chunk lives in (two groups of 4, by pixel parity).  Schemes that rotate the chunk per lane and pick the
// left / right corner by pixel parity make the (parity, chunk) pairs of 16 consecutive lanes distinct -- IF the LDS
// processes a b128 read in passes of 16 consecutive lanes.  This measures it: clocks per 32-read sample evaluation
// per wave and bytes per clock per CU, 16 waves on one CU, with and without the 64 v_pk_fma_f32 that consume the data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4 __attribute__((ext_vector_type(4)));
typedef float v2 __attribute__((ext_vector_type(2)));
#define LDSP __attribute__((address_space(3)))

// SCHEME 0: sequential (lane * 16 + j * 1024): the conflict-free ceiling
//        1: random pixels, every lane on chunk j
//        2: random pixels, chunk j ^ ((lane >> 1) & 7), corner side by parity == lane & 1
//        3: random pixels, chunk j ^ (lane & 7), corner side by parity == (lane >> 3) & 1
//        4: random pixels, chunk j ^ (lane & 7), no parity choice
template <int SCHEME, bool FMA>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, float* sink, int iters, int npix, int pitch) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  for (int i = threadIdx.x; i < npix * 32; i += blockDim.x) ((LDSP float*)lds)[i] = (float)(i & 1023);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned h = threadIdx.x * 2654435761u + 12345u;
  v2 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (v2){0.f, 0.f};
  const v2 w = {1.0001f, 0.9999f};
  const unsigned lds_base = (unsigned)(unsigned long long)(LDSP char*)lds;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    h = h * 1664525u + 1013904223u;
    // top-left corner of my sample: row in [0, rows-2], column in [0, pitch-2]
    const int rows = npix / pitch;
    const int r0 = (int)((h >> 8) % (unsigned)(rows - 1)), c0 = (int)((h >> 20) % (unsigned)(pitch - 1));
    unsigned rot, first_side;
    if (SCHEME == 2) { rot = (lane >> 1) & 7; first_side = ((unsigned)c0 ^ lane) & 1; }
    else if (SCHEME == 3) { rot = lane & 7; first_side = ((unsigned)c0 ^ (lane >> 3)) & 1; }
    else if (SCHEME == 4) { rot = lane & 7; first_side = 0; }
    else { rot = 0; first_side = 0; }
    unsigned base[4];
    if (SCHEME == 0) {
      for (int q = 0; q < 4; ++q) base[q] = lds_base + lane * 16 + q * 8192;
    } else {
      const unsigned tl = lds_base + (unsigned)(r0 * pitch + c0) * 128u;
      base[0] = tl + first_side * 128u;             // top, first side
      base[1] = tl + (first_side ^ 1u) * 128u;      // top, other side
      base[2] = base[0] + (unsigned)pitch * 128u;   // bottom
      base[3] = base[1] + (unsigned)pitch * 128u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned a = SCHEME == 0 ? base[q] + j * 1024 : base[q] + (((unsigned)j ^ rot) << 4);
        const v4 d = *(const LDSP v4*)(unsigned long long)a;
        if (FMA) {
          acc[2 * j] = __builtin_elementwise_fma(w, (v2){d.x, d.y}, acc[2 * j]);
          acc[2 * j + 1] = __builtin_elementwise_fma(w, (v2){d.z, d.w}, acc[2 * j + 1]);
        } else {
          acc[(2 * j) & 15] += (v2){d.x, d.w};
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  if (s == 123.456f) sink[threadIdx.x] = s;
  if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
}

template <int SCHEME, bool FMA>
static void run(const char* name, int waves) {
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, 16 * sizeof(unsigned long long));
  hipMalloc(&sink, 1024 * sizeof(float));
  const int pitch = 26, npix = 26 * 48, iters = 200;
  const size_t lds = (size_t)npix * 128;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<SCHEME, FMA>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((probe<SCHEME, FMA>), dim3(1), dim3(64 * waves), lds, 0, out, sink, iters, npix, pitch);
    hipDeviceSynchronize();
  }
  std::vector<unsigned long long> h(16);
  hipMemcpy(h.data(), out, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double mx = 0;
  for (int i = 0; i < waves; ++i) mx = std::max<double>(mx, (double)h[i]);
  const double per_eval = mx / iters;                                  // clocks for all `waves` waves to do one evaluation each
  printf("%-44s waves %2d fma %d: %8.1f clk per round of evaluations, %6.1f clk per wave-evaluation, %6.1f B/clk/CU\n", name, waves,
         (int)FMA, per_eval, per_eval / waves, waves * 32.0 * 1024.0 / per_eval);
  hipFree(out);
  hipFree(sink);
}

int main() {
  for (int waves : {4, 16}) {
    if (waves == 4) {
      run<0, false>("0 sequential", 4); run<1, false>("1 same chunk", 4); run<2, false>("2 rot lane>>1, parity lane&1", 4);
      run<3, false>("3 rot lane&7, parity lane>>3", 4); run<4, false>("4 rot lane&7, no parity", 4);
    } else {
      run<0, false>("0 sequential", 16); run<1, false>("1 same chunk", 16); run<2, false>("2 rot lane>>1, parity lane&1", 16);
      run<3, false>("3 rot lane&7, parity lane>>3", 16); run<4, false>("4 rot lane&7, no parity", 16);
      run<0, true>("0 sequential", 16); run<2, true>("2 rot lane>>1, parity lane&1", 16); run<3, true>("3 rot lane&7, parity lane>>3", 16);
    }
  }
  return 0;
}
