// hbm_calibrate.hip -- known-byte-count streaming kernels in the SAME access width the ecrad kernels
// use (8 bytes per lane, 512 B per wave64 instruction), for calibrating rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes.
//   read8  : reads  N doubles, writes ~nothing      -> FETCH_SIZE  vs N*8
//   write8 : writes N doubles, reads nothing        -> WRITE_SIZE  vs N*8
// N*8 = 2 GiB so that the 256 MiB Infinity Cache cannot absorb the stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void read8(const double* __restrict__ a, size_t n, double* out) {
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
  if (acc == 12345.678) out[0] = acc;   // never true: keeps the loads alive
}
__global__ void write8(double* __restrict__ a, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}

int main() {
  const size_t n = (size_t)1 << 28;   // 2 GiB of doubles
  double *a, *o;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&o, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0, n * 8);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0); hipLaunchKernelGGL(write8, dim3(2048), dim3(256), 0, 0, a, n); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("write8 bytes=%zu ms=%.3f GB/s=%.1f\n", n * 8, ms, n * 8 / ms / 1e6);
    hipEventRecord(e0); hipLaunchKernelGGL(read8, dim3(2048), dim3(256), 0, 0, a, n, o); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("read8  bytes=%zu ms=%.3f GB/s=%.1f\n", n * 8, ms, n * 8 / ms / 1e6);
  }
  return 0;
}
