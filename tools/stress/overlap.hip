// tools/stress/overlap.hip -- do the tile copies of a pipelined host-memory call overlap with its kernels?  A kernel that keeps every CU busy for
// ~20 ms on one stream; on another, the copy-in of one tile as the library issues it: 20 arrays of (137 rows x 12288 columns) doubles out of
// host arrays of 100 000 columns (hipMemcpy2DAsync, host pitch 800 000 bytes), page-locked or pageable -- and the same bytes as ONE linear copy.
//   hipcc --offload-arch=gfx950 -O2 -o overlap overlap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ __launch_bounds__(256) void spin(long long cycles, double* sink) {
  const long long t0 = wall_clock64();
  double x = threadIdx.x;
  while (wall_clock64() - t0 < cycles) x = x * 1.0000001 + 1e-9;
  if (x == 12345.0) sink[0] = x;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t ncol = 100000, nlev = 137, tile = 12288, narr = 20;
  const size_t abytes = ncol * nlev * 8, tbytes = tile * nlev * 8;
  double *hp, *hg, *d, *sink;
  CK(hipHostMalloc((void**)&hp, abytes * 2, hipHostMallocDefault));      // (two arrays' worth: the 20 "arrays" alternate between them)
  hg = (double*)malloc(abytes * 2); memset(hg, 0, abytes * 2); memset(hp, 0, abytes * 2);
  CK(hipMalloc(&d, tbytes * narr)); CK(hipMalloc(&sink, 8));
  hipStream_t sk, sc;
  CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
  int wc_khz = 100000;
  (void)hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
  const long long cycles = (long long)wc_khz * 20;      // 20 ms
  auto kernel = [&] { hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(256), 0, sk, cycles, sink); };
  auto copy2d = [&](double* host, int dir) {
    for (size_t a = 0; a < narr; ++a) {
      double* h = host + (a & 1) * ncol * nlev + 3 * tile;
      if (dir == 0) CK(hipMemcpy2DAsync(d + a * tile * nlev, tile * 8, h, ncol * 8, tile * 8, nlev, hipMemcpyHostToDevice, sc));
      else CK(hipMemcpy2DAsync(h, ncol * 8, d + a * tile * nlev, tile * 8, tile * 8, nlev, hipMemcpyDeviceToHost, sc));
    }
  };
  auto copy1d = [&](double* host, int dir) {
    if (dir == 0) CK(hipMemcpyAsync(d, host, tbytes * narr > abytes * 2 ? abytes * 2 : tbytes * narr, hipMemcpyHostToDevice, sc));
    else CK(hipMemcpyAsync(host, d, tbytes * narr > abytes * 2 ? abytes * 2 : tbytes * narr, hipMemcpyDeviceToHost, sc));
  };
  const double gb = (tbytes * narr > abytes * 2 ? abytes * 2 : tbytes * narr) / 1e9;
  kernel(); CK(hipDeviceSynchronize());
  double t0 = now_ms(); kernel(); CK(hipStreamSynchronize(sk)); printf("kernel alone: %.2f ms\n", now_ms() - t0);
  struct { const char* name; double* host; int form; } cases[] = {{"page-locked, 2-D", hp, 2}, {"page-locked, linear", hp, 1}, {"pageable, 2-D", hg, 2}, {"pageable, linear", hg, 1}};
  for (int dir = 0; dir < 2; ++dir)
    for (auto& c : cases) {
      auto copy = [&] { if (c.form == 2) copy2d(c.host, dir); else copy1d(c.host, dir); };
      copy(); CK(hipStreamSynchronize(sc));
      t0 = now_ms(); copy(); CK(hipStreamSynchronize(sc)); const double alone = now_ms() - t0;
      t0 = now_ms(); kernel(); copy(); CK(hipStreamSynchronize(sc)); const double with_k = now_ms() - t0; CK(hipStreamSynchronize(sk)); const double both = now_ms() - t0;
      printf("%s %-20s: alone %6.2f ms (%5.1f GB/s); next to the kernel: copy done after %6.2f ms, both after %6.2f ms\n", dir ? "out" : "in ", c.name, alone, gb / alone * 1e3,
             with_k, both);
    }
  return 0;
}
