// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the
// persistent update kernels (MI355X_MICROARCH.md: the 1/2 factor of FETCH_SIZE is established for
// 16-byte-per-lane streaming reads only; "other access widths and WRITE_SIZE are uncalibrated"):
// every kernel moves exactly `bytes` bytes; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./pmc_calib   and   --pmc WRITE_SIZE ...
// and compare the counters (KB) with the known byte count.
// Build: hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void read_f4(const float4* p, size_t n4, float* sink) {          // 16 B per lane, cached
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) *sink = acc;
}
__global__ void read_sc1_b32(const unsigned* p, size_t n, float* sink) {    // 4 B per lane around the L2
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (acc == 0x12345678u) *sink = 1.f;
}
__global__ void read_sc1_b64(const unsigned long long* p, size_t n, float* sink) {   // 8 B per lane around the L2
  unsigned long long acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    acc += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (acc == 0x12345678ull) *sink = 1.f;
}
__global__ void read_b32(const float* p, size_t n, float* sink) {           // 4 B per lane, cached
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678f) *sink = acc;
}
__global__ void write_f4(float4* p, size_t n4) {                            // 16 B per lane, cached
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void write_sc1_b32(unsigned* p, size_t n) {                      // 4 B per lane, written through
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __hip_atomic_store(p + i, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void write_b32(float* p, size_t n) {                             // 4 B per lane, cached
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)i;
}

int main() {
  const size_t bytes = (size_t)1 << 30;   // 1 GiB per kernel: far beyond the 256 MB Infinity Cache
  void *buf; float* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 4);
  hipMemset(buf, 1, bytes);
  const dim3 g(2048), b(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read_f4, g, b, 0, 0, (const float4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(read_b32, g, b, 0, 0, (const float*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(read_sc1_b32, g, b, 0, 0, (const unsigned*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(read_sc1_b64, g, b, 0, 0, (const unsigned long long*)buf, bytes / 8, sink);
    hipLaunchKernelGGL(write_f4, g, b, 0, 0, (float4*)buf, bytes / 16);
    hipLaunchKernelGGL(write_b32, g, b, 0, 0, (float*)buf, bytes / 4);
    hipLaunchKernelGGL(write_sc1_b32, g, b, 0, 0, (unsigned*)buf, bytes / 4);
    hipDeviceSynchronize();
  }
  printf("each kernel moved %zu bytes = %zu KB\n", bytes, bytes / 1024);
  return 0;
}
