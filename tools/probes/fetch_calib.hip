// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts (the guide calibrates
// FETCH_SIZE = 1/2 of the bytes on 16-B/lane streaming reads only).  Every kernel touches each 64-byte line of a 1-GiB buffer
// (4 x the 256-MiB Infinity Cache) exactly ONCE: 2^30 bytes of compulsory traffic per launch, in contiguous runs of RUN lines placed
// by an odd-multiplier permutation of the run index (RUN = 1: isolated 64-B lines -- the gather granularity of msda_strips'
// super-pixel rows when a window row is short; RUN = 16: 1 KB; calib_stream: the whole buffer in order).  Four lanes share a line
// (16 B per lane), as in msda_strips.hip / msda_tiled2.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -o p --output-format csv -- /tmp/fetch_calib      (then WRITE_SIZE in a pass of its own)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int RUN>
__global__ __launch_bounds__(256) void calib_gather(const f4* __restrict__ buf, f4* __restrict__ sink, unsigned nlines, unsigned mult) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  const unsigned piece = gid >> 2, q = gid & 3u;
  const unsigned nruns = nlines / RUN;                       // a power of two: any odd multiplier permutes the runs
  const unsigned run = piece / RUN, within = piece % RUN;
  const unsigned prun = (run * mult) & (nruns - 1u);
  const size_t line = (size_t)prun * RUN + within;
  const f4 v = buf[line * 4 + q];
  if (v.x == 123456.789f && v.y == 42.0f) sink[gid & 1023u] = v;   // never true: keeps the load
}

__global__ __launch_bounds__(256) void calib_stream(const f4* __restrict__ buf, f4* __restrict__ sink, unsigned nlines) {
  const size_t gid = (size_t)blockIdx.x * 256u + threadIdx.x;
  const f4 v = buf[gid];
  if (v.x == 123456.789f && v.y == 42.0f) sink[gid & 1023u] = v;
}

// writes: every 64-B line once, runs of RUN lines as above
template <int RUN>
__global__ __launch_bounds__(256) void calib_scatter(f4* __restrict__ buf, unsigned nlines, unsigned mult) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  const unsigned piece = gid >> 2, q = gid & 3u;
  const unsigned nruns = nlines / RUN;
  const unsigned run = piece / RUN, within = piece % RUN;
  const unsigned prun = (run * mult) & (nruns - 1u);
  const size_t line = (size_t)prun * RUN + within;
  buf[line * 4 + q] = (f4){(float)gid, 1.f, 2.f, 3.f};
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t bytes = (size_t)1 << 30;
  const unsigned nlines = (unsigned)(bytes / 64);
  f4 *buf = nullptr, *sink = nullptr;
  CK(hipMalloc(&buf, bytes));
  CK(hipMalloc(&sink, 1024 * sizeof(f4)));
  CK(hipMemset(buf, 0, bytes));
  const unsigned grid = (unsigned)(bytes / 16 / 256);
  const unsigned mult = 2654435761u | 1u;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto report = [&](const char* name, float ms) { printf("%-22s %8.3f ms  %7.1f GB/s of compulsory bytes\n", name, ms, bytes / ms / 1e6); };
  float ms;
#define RUN_K(name, launch)                                  \
  for (int rep = 0; rep < 3; ++rep) {                        \
    CK(hipEventRecord(a));                                   \
    launch;                                                  \
    CK(hipEventRecord(b));                                   \
    CK(hipEventSynchronize(b));                              \
    CK(hipEventElapsedTime(&ms, a, b));                      \
  }                                                          \
  report(name, ms);
  RUN_K("calib_stream", hipLaunchKernelGGL(calib_stream, dim3(grid), dim3(256), 0, 0, buf, sink, nlines));
  RUN_K("calib_gather<1>", hipLaunchKernelGGL(calib_gather<1>, dim3(grid), dim3(256), 0, 0, buf, sink, nlines, mult));
  RUN_K("calib_gather<2>", hipLaunchKernelGGL(calib_gather<2>, dim3(grid), dim3(256), 0, 0, buf, sink, nlines, mult));
  RUN_K("calib_gather<4>", hipLaunchKernelGGL(calib_gather<4>, dim3(grid), dim3(256), 0, 0, buf, sink, nlines, mult));
  RUN_K("calib_gather<16>", hipLaunchKernelGGL(calib_gather<16>, dim3(grid), dim3(256), 0, 0, buf, sink, nlines, mult));
  RUN_K("calib_gather<64>", hipLaunchKernelGGL(calib_gather<64>, dim3(grid), dim3(256), 0, 0, buf, sink, nlines, mult));
  RUN_K("calib_scatter<1>", hipLaunchKernelGGL(calib_scatter<1>, dim3(grid), dim3(256), 0, 0, buf, nlines, mult));
  RUN_K("calib_scatter<16>", hipLaunchKernelGGL(calib_scatter<16>, dim3(grid), dim3(256), 0, 0, buf, nlines, mult));
  RUN_K("calib_scatter<4096>", hipLaunchKernelGGL(calib_scatter<4096>, dim3(grid), dim3(256), 0, 0, buf, nlines, mult));
  CK(hipDeviceSynchronize());
  return 0;
}
