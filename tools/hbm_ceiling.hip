// hbm_ceiling.hip -- what streaming rate can this box sustain?  (tuning aid, not part of the library)
// read / write / copy at 8 and 16 bytes per lane with several loads in flight per lane, and the
// access pattern of the sweep scratch: every block writes a private slab level by level and reads it
// back in reverse order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T, int U>
__global__ void rd(const T* __restrict__ a, size_t n, double* out) {
  double acc = 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += ((const double*)&v[u])[0];
  }
  if (acc == 12345.678) out[0] = acc;
}
template <typename T>
__global__ void wr(T* __restrict__ a, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  T v; for (unsigned k = 0; k < sizeof(T) / 8; ++k) ((double*)&v)[k] = 1.0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) a[i] = v;
}
template <typename T, int U>
__global__ void cp(const T* __restrict__ a, T* __restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    T v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) b[i + u * stride] = v[u];
  }
}
// scratch pattern: slab of nlev records of REC doubles per lane; groups = how many times each block re-uses it
template <int REC>
__global__ void slab(double* __restrict__ base, int nlev, int groups, double* out) {
  double* s = base + (size_t)blockIdx.x * nlev * REC * blockDim.x;
  double acc = 0.0;
  for (int g = 0; g < groups; ++g) {
    for (int l = 0; l < nlev; ++l)
#pragma unroll
      for (int r = 0; r < REC; ++r) s[((size_t)l * REC + r) * blockDim.x + threadIdx.x] = acc + l + r;
    for (int l = nlev - 1; l >= 0; --l)
#pragma unroll
      for (int r = 0; r < REC; ++r) acc += s[((size_t)l * REC + r) * blockDim.x + threadIdx.x];
  }
  if (acc == 12345.678) out[0] = acc;
}

#define TIME(label, bytes, ...)                                                        \
  do {                                                                                 \
    float best = 1e30f;                                                                \
    for (int rep = 0; rep < 3; ++rep) {                                                \
      hipEventRecord(e0); __VA_ARGS__; hipEventRecord(e1); hipEventSynchronize(e1);    \
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;            \
    }                                                                                  \
    printf("%-28s %8.3f ms  %7.1f GB/s\n", label, best, (double)(bytes) / best / 1e6); \
  } while (0)

int main() {
  const size_t nbytes = (size_t)4 << 30;
  double *a, *b, *o;
  if (hipMalloc(&a, nbytes) != hipSuccess || hipMalloc(&b, nbytes) != hipSuccess || hipMalloc(&o, 8) != hipSuccess) return 1;
  hipMemset(a, 0, nbytes); hipMemset(b, 0, nbytes); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t n8 = nbytes / 8, n16 = nbytes / 16;
  for (int grid : {768, 2048, 8192}) {
    printf("grid %d x 256\n", grid);
    TIME("read 8B x1", nbytes, hipLaunchKernelGGL((rd<double, 1>), dim3(grid), dim3(256), 0, 0, a, n8, o));
    TIME("read 8B x4", nbytes, hipLaunchKernelGGL((rd<double, 4>), dim3(grid), dim3(256), 0, 0, a, n8, o));
    TIME("read 16B x1", nbytes, hipLaunchKernelGGL((rd<double2, 1>), dim3(grid), dim3(256), 0, 0, (const double2*)a, n16, o));
    TIME("read 16B x4", nbytes, hipLaunchKernelGGL((rd<double2, 4>), dim3(grid), dim3(256), 0, 0, (const double2*)a, n16, o));
    TIME("write 8B", nbytes, hipLaunchKernelGGL((wr<double>), dim3(grid), dim3(256), 0, 0, a, n8));
    TIME("write 16B", nbytes, hipLaunchKernelGGL((wr<double2>), dim3(grid), dim3(256), 0, 0, (double2*)a, n16));
    TIME("copy 16B x4 (r+w bytes)", 2 * nbytes, hipLaunchKernelGGL((cp<double2, 4>), dim3(grid), dim3(256), 0, 0, (const double2*)a, (double2*)b, n16));
  }
  // slab pattern: 768 blocks, 137 levels, 5 doubles: 1.4 MB per block, 1.07 GB footprint
  for (int grid : {512, 768, 1024}) {
    const int groups = 16;
    const size_t bytes = (size_t)grid * 137 * 5 * 256 * 8 * 2 * groups;
    char label[64]; snprintf(label, sizeof label, "slab w+r 5x8B, %d blocks", grid);
    TIME(label, bytes, hipLaunchKernelGGL((slab<5>), dim3(grid), dim3(256), 0, 0, a, 137, groups, o));
  }
  return 0;
}
