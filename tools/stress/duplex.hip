// tools/stress/duplex.hip -- can the two directions of the host link be busy at once?  (round 5: two page-locked copies in opposite directions from two
// threads came out at the ONE-way rate, 57 GB/s for both together.)  Page-locked host buffers of 1 GiB, every combination of
//   in : hipMemcpyAsync host->device (copy engine)  |  a kernel that READS the host buffer (zero-copy over the link)
//   out: hipMemcpyAsync device->host (copy engine)  |  a kernel that WRITES the host buffer
// alone and in pairs on two streams (two host threads for the copy-engine pairs, as the tile pipeline issues them).
//   hipcc --offload-arch=gfx950 -O2 -o duplex duplex.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void pull(v2d* __restrict__ dst, const v2d* __restrict__ host, size_t n) {      // host -> device by loads
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const v2d a = host[i], b = host[i + stride], c = host[i + 2 * stride], d = host[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
}
__global__ __launch_bounds__(256) void push(v2d* __restrict__ host, const v2d* __restrict__ src, size_t n) {      // device -> host by stores
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const v2d a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    __builtin_nontemporal_store(a, host + i); __builtin_nontemporal_store(b, host + i + stride);
    __builtin_nontemporal_store(c, host + i + 2 * stride); __builtin_nontemporal_store(d, host + i + 3 * stride);
  }
}

int main(int argc, char** argv) {
  const size_t nbytes = (size_t)1 << 30, n = nbytes / 16;
  const int grid = argc > 1 ? atoi(argv[1]) : 64;      // blocks of the copy kernels (they share the device with the solver kernels: keep them small)
  void *hin, *hout, *din, *dout;
  CK(hipHostMalloc(&hin, nbytes, hipHostMallocDefault)); CK(hipHostMalloc(&hout, nbytes, hipHostMallocDefault));
  CK(hipMalloc(&din, nbytes)); CK(hipMalloc(&dout, nbytes));
  memset(hin, 1, nbytes); memset(hout, 0, nbytes);
  CK(hipMemset(dout, 2, nbytes));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  auto in_op = [&](int how) {
    if (how == 0) CK(hipMemcpyAsync(din, hin, nbytes, hipMemcpyHostToDevice, s1));
    else hipLaunchKernelGGL(pull, dim3(grid), dim3(256), 0, s1, (v2d*)din, (const v2d*)hin, n);
  };
  auto out_op = [&](int how) {
    if (how == 0) CK(hipMemcpyAsync(hout, dout, nbytes, hipMemcpyDeviceToHost, s2));
    else hipLaunchKernelGGL(push, dim3(grid), dim3(256), 0, s2, (v2d*)hout, (const v2d*)dout, n);
  };
  const char* name[2] = {"copy engine", "kernel"};
  for (int rep = 0; rep < 2; ++rep) {
    for (int how = 0; how < 2; ++how) {
      auto t0 = std::chrono::steady_clock::now();
      in_op(how); CK(hipStreamSynchronize(s1));
      double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("in  alone, %-11s: %6.1f GB/s\n", name[how], nbytes / ms / 1e6);
      t0 = std::chrono::steady_clock::now();
      out_op(how); CK(hipStreamSynchronize(s2));
      ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("out alone, %-11s: %6.1f GB/s\n", name[how], nbytes / ms / 1e6);
    }
    for (int hi = 0; hi < 2; ++hi)
      for (int ho = 0; ho < 2; ++ho) {
        double ms_in = 0, ms_out = 0;
        const auto t0 = std::chrono::steady_clock::now();
        std::thread other([&] {
          CK(hipSetDevice(0));
          out_op(ho); CK(hipStreamSynchronize(s2));
          ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        });
        in_op(hi); CK(hipStreamSynchronize(s1));
        ms_in = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        other.join();
        const double ms = ms_in > ms_out ? ms_in : ms_out;
        if (rep) printf("both: in by %-11s (%6.1f GB/s), out by %-11s (%6.1f GB/s): together %6.1f GB/s\n", name[hi], nbytes / ms_in / 1e6, name[ho],
                        nbytes / ms_out / 1e6, 2.0 * nbytes / ms / 1e6);
      }
  }
  // did the kernel's stores arrive?
  const unsigned char* p = (const unsigned char*)hout;
  printf("host buffer after the last out: %d %d (expect 2 2)\n", p[0], p[nbytes - 1]);
  return 0;
}
