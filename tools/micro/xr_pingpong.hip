// Micro-benchmark for a data-parallel rank that stays RESIDENT across the gradient exchange
// (DESIGN.md section 5): a persistent kernel on stream A hands "gradients are out" to a second
// stream B through a counter the command processor waits on (hipStreamWaitValue32), B runs the
// exchange (here: nothing | an empty kernel | a 4.3 MB copy kernel, standing in for ncclAllReduce's
// launch) and releases the kernel with hipStreamWriteValue32; the kernel polls that word around its
// caches.  Prints the round trip as the resident kernel sees it (wall_clock64, 100 MHz).
// Every poll is bounded: a mechanism that does not work here ends in "timed out", not in a hang.
// Build: hipcc --offload-arch=gfx950 -O3 -o xr_pingpong xr_pingpong.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void resident(int iters, unsigned* ready, const unsigned* done, long long* stamps,
                                                int* failed, float* grads, int floats_per_wg, unsigned* count) {
  extern __shared__ float lds[];
  const int G = gridDim.x;
  for (int it = 0; it < iters; ++it) {
    // "gradients": this workgroup's share, written through
    for (int i = threadIdx.x; i < floats_per_wg; i += blockDim.x)
      __hip_atomic_store(reinterpret_cast<unsigned*>(grads) + (size_t)blockIdx.x * floats_per_wg + i, (unsigned)it,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (sc1: written through, as the kernel's exchanges)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    long long t0 = 0;
    if (threadIdx.x == 0) {
      t0 = wall_clock64();
      // the workgroups count themselves in device memory; the last one raises the word the command
      // processor waits on (one system-scope store: 248 atomics on the signal word itself cost ~1 us EACH)
      const unsigned old = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (no cache write-back: the data went through)
      if (old + 1 == (unsigned)(it + 1) * G)
        __hip_atomic_store(ready, (unsigned)(it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      int spins = 0;
      while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned)(it + 1)) {
        if (++spins > 4000000) { *failed = 1; break; }
        __builtin_amdgcn_s_sleep(8);      // (248 workgroups spinning flat out on one word: 24 us instead of 9)
      }
      if (blockIdx.x == 0) stamps[it] = wall_clock64() - t0;
    }
    __syncthreads();
    if (*(volatile int*)failed) return;
  }
}

__global__ void empty_kernel() {}
__global__ void copy_kernel(const float4* a, float4* b, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b[i] = a[i];
}

int main(int argc, char** argv) {
  const int iters = 200;
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  unsigned *ready, *done, *count; long long* stamps; int* failed; float *grads, *grads2;
  const int n_floats = 260 * 4096 + 260;
  CK(hipExtMallocWithFlags((void**)&ready, 8, hipMallocSignalMemory));
  CK(hipMalloc(&done, 256)); CK(hipMalloc(&count, 256)); CK(hipMalloc(&stamps, iters * 8)); CK(hipMalloc(&failed, 4));
  CK(hipMalloc(&grads, (size_t)n_floats * 4 + 4096)); CK(hipMalloc(&grads2, (size_t)n_floats * 4 + 4096));
  for (int G : {1, 248}) {
    for (int mode = 0; mode < 3; ++mode) {
      CK(hipMemset(ready, 0, 8)); CK(hipMemset(done, 0, 256)); CK(hipMemset(count, 0, 256)); CK(hipMemset(failed, 0, 4));
      CK(hipDeviceSynchronize());
      const int fpw = n_floats / G / 4 * 4;
      CK(hipFuncSetAttribute((const void*)resident, hipFuncAttributeMaxDynamicSharedMemorySize, 129 * 1024));
      hipLaunchKernelGGL(resident, dim3(G), dim3(512), 129 * 1024, A, iters, ready, done, stamps, failed, grads, fpw, count);
      CK(hipGetLastError());
      for (int it = 0; it < iters; ++it) {
        CK(hipStreamWaitValue32(B, ready, (unsigned)(it + 1), hipStreamWaitValueGte, 0xFFFFFFFFu));
        if (mode == 1) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, B);
        if (mode == 2) hipLaunchKernelGGL(copy_kernel, dim3(8), dim3(1024), 0, B, (const float4*)grads, (float4*)grads2, n_floats / 4);
        CK(hipStreamWriteValue32(B, done, (unsigned)(it + 1), 0));
      }
      CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
      int f = 0; CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
      std::vector<long long> h(iters); CK(hipMemcpy(h.data(), stamps, iters * 8, hipMemcpyDeviceToHost));
      std::vector<long long> s(h.begin() + 20, h.end()); std::sort(s.begin(), s.end());
      printf("G=%3d workgroups, exchange = %-28s: %s round trip median %.2f us, p10 %.2f, p90 %.2f (updates 20..%d)\n", G,
             mode == 0 ? "nothing" : mode == 1 ? "empty kernel" : "4.3 MB copy kernel (8 WGs)", f ? "TIMED OUT;" : "",
             s[s.size() / 2] / 100.0, s[s.size() / 10] / 100.0, s[s.size() * 9 / 10] / 100.0, iters);
      fflush(stdout);
    }
  }
  return 0;
}
