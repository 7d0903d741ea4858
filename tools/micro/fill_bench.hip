// What does the chip take as PURE writes when the footprint is far beyond the 256 MiB Infinity
// Cache?  (The summarizers write 2.4 - 9.4 GB per call and read almost nothing.)
//   hipcc --offload-arch=gfx950 -O3 -o fill_bench fill_bench.hip
// Variants: plain / nontemporal 16-byte stores; one contiguous 1 KB run per wavefront instruction;
// workgroups own contiguous ROWS of `row_kb` KB (the summarizers: one trajectory row per workgroup)
// or interleave (grid-stride over 16-byte quads); hipMemsetAsync for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_rows(f4* out, size_t n_rows, size_t quads_per_row, float v) {
  for (size_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    f4* o = out + r * quads_per_row;
    const f4 x = {v, v + 1.f, v + 2.f, (float)r};
    for (size_t q = threadIdx.x; q < quads_per_row; q += 256) {
      if (NT) __builtin_nontemporal_store(x, o + q); else o[q] = x;
    }
  }
}
// the cross-correlation store loop's geometry: rows of `row_bytes` (a multiple of 16, not of 128),
// `act` of the 256 threads store one quad each per sweep; `align`: the thread <-> quad map is
// rotated per row so that every wavefront's 1 KB run starts on a 128-byte line
template <bool NT>
__global__ __launch_bounds__(256) void fill_rows_cc(f4* out, size_t n_rows, size_t row_quads, int act, int align, float v) {
  for (size_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const size_t base = r * row_quads;
    f4* o = out + base;
    const f4 x = {v, v + 1.f, v + 2.f, (float)r};
    const int shift = align ? (int)((8 - (base & 7)) & 7) : 0;
    if ((int)threadIdx.x < shift) { if (NT) __builtin_nontemporal_store(x, o + threadIdx.x); else o[threadIdx.x] = x; }
    if ((int)threadIdx.x < act)
      for (size_t q = shift + threadIdx.x; q < row_quads; q += act) {
        if (NT) __builtin_nontemporal_store(x, o + q); else o[q] = x;
      }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void fill_stride(f4* out, size_t n_quads, float v) {
  const f4 x = {v, v + 1.f, v + 2.f, v + 3.f};
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < n_quads; q += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(x, out + q); else out[q] = x;
  }
}

template <typename F> static double time_us(F&& f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3 / reps;
}

int main() {
  for (double gb : {0.1, 0.5, 2.36, 9.44}) {
    const size_t bytes = (size_t)(gb * 1e9) / 65536 * 65536;
    f4* buf; CK(hipMalloc(&buf, bytes));
    const size_t nq = bytes / 16;
    printf("footprint %.2f GB\n", bytes / 1e9);
    for (int row_kb : {47, 189}) {
      const size_t qpr = (size_t)row_kb * 1024 / 16, rows = nq / qpr;
      for (int grid : {2048, 16384}) {
        double t0 = time_us([&] { hipLaunchKernelGGL(fill_rows<false>, dim3(grid), dim3(256), 0, 0, buf, rows, qpr, 1.f); }, 10);
        double t1 = time_us([&] { hipLaunchKernelGGL(fill_rows<true>, dim3(grid), dim3(256), 0, 0, buf, rows, qpr, 1.f); }, 10);
        printf("  rows of %3d KB, grid %5d: plain %8.1f us = %.2f TB/s | nontemporal %8.1f us = %.2f TB/s\n", row_kb, grid,
               t0, rows * qpr * 16 / t0 / 1e6, t1, rows * qpr * 16 / t1 / 1e6);
      }
    }
    {
      const size_t rq = 47216 / 16, rows = nq / rq;
      for (int act : {250, 240, 256}) for (int al : {0, 1}) {
        double t0 = time_us([&] { hipLaunchKernelGGL(fill_rows_cc<false>, dim3(16384), dim3(256), 0, 0, buf, rows, rq, act, al, 1.f); }, 10);
        double t1 = time_us([&] { hipLaunchKernelGGL(fill_rows_cc<true>, dim3(16384), dim3(256), 0, 0, buf, rows, rq, act, al, 1.f); }, 10);
        printf("  rows of 47216 B, %d threads per sweep, %s: plain %8.1f us = %.2f TB/s | nontemporal %8.1f us = %.2f TB/s\n", act,
               al ? "line-aligned runs" : "runs as they fall", t0, rows * rq * 16 / t0 / 1e6, t1, rows * rq * 16 / t1 / 1e6);
      }
    }
    for (int grid : {2048, 16384}) {
      double t0 = time_us([&] { hipLaunchKernelGGL(fill_stride<false>, dim3(grid), dim3(256), 0, 0, buf, nq, 1.f); }, 10);
      double t1 = time_us([&] { hipLaunchKernelGGL(fill_stride<true>, dim3(grid), dim3(256), 0, 0, buf, nq, 1.f); }, 10);
      printf("  grid-stride quads, grid %5d: plain %8.1f us = %.2f TB/s | nontemporal %8.1f us = %.2f TB/s\n", grid,
             t0, bytes / t0 / 1e6, t1, bytes / t1 / 1e6);
    }
    double tm = time_us([&] { CK(hipMemsetAsync(buf, 0, bytes, 0)); }, 10);
    printf("  hipMemsetAsync: %8.1f us = %.2f TB/s\n", tm, bytes / tm / 1e6);
    CK(hipFree(buf));
  }
  return 0;
}
