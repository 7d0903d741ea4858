// Micro-benchmark for the signature half of f2 (summarizers.py:144-168 into
// mdnn.py:71,108): could a first-layer tile workgroup of the persistent MDNN kernel REBUILD the 256
// level-3 signature columns of its k-slice for the 104 minibatch rows from the rows' [11, 22] paths,
// instead of reading them from a materialised 45 KB summary row?
//
// One workgroup of 512 threads (the kernel's), 104 rows, a k-slice of 256 consecutive level-3 columns
// col = (i*d + j)*d + k of d = 22 channels: 12-13 (i, j) pairs.  Thread <-> (row, pair): the pair's
// S2[i,j] and its S3[i,j,0..22) in registers over the 10 segments, in signature3_kernel's operation
// order (Chen / Horner: coef = S2 + (S1_i + D_i/3) * D_j/2;  S3_k = fma(coef, D_k, S3_k);
// S2 = fma(S1_i + D_i/2, D_j, S2)), the result written into a [104][260] tile in LDS like the one
// the forward product reads.  Two variants of where the rows' paths come from:
//   mode 0: the rows' paths (11 x 22 floats each) are staged in LDS first, 26 rows at a time (27 KB:
//           the 104 rows' 110 KB do not fit next to the 106 KB minibatch tile, and the real kernel
//           has ~20 KB to spare beside its minibatch and weight tiles) -- four passes per rebuild;
//   mode 1: every lane reads its row's path values from global memory (L2) as it needs them.
// Prints the time of one rebuild (HIP events over `iters` rebuilds by every workgroup of a 256-wide
// grid: one per CU, as in the kernel).
// Build: hipcc --offload-arch=gfx950 -O3 -o sig3_tile sig3_tile_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kD = 22, kL = 11, kRows = 104, kCols = 256, kPitch = 260, kT = 512;
constexpr int kDP = 24;                       // increment rows padded to float4
constexpr int kRP = 26;                       // mode 0: rows staged per pass

template <int MODE>
__global__ __launch_bounds__(kT) void rebuild(const float* __restrict__ paths, int iters, int col0, float* sink) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                                   // [kRows][kPitch]
  float* pl = tile + kRows * kPitch;                    // mode 0: [kRP][kL][kDP] paths of the pass's rows
  const int tid = threadIdx.x;
  // the pairs this k-slice touches
  const int p_lo = col0 / kD, p_hi = (col0 + kCols - 1) / kD, n_pairs = p_hi - p_lo + 1;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const float* src = paths + (size_t)((blockIdx.x * 7 + it) % 64) * kRows * kL * kD;   // (another minibatch each time)
    for (int r0 = 0; r0 < kRows; r0 += (MODE == 0 ? kRP : kRows)) {
      const int nr = MODE == 0 ? kRP : kRows;
      if (MODE == 0) {
        __syncthreads();
        for (int e = tid; e < nr * kL * kD; e += kT) {
          const int r = e / (kL * kD), rem = e - r * (kL * kD), l = rem / kD, c = rem - l * kD;
          pl[(r * kL + l) * kDP + c] = src[(size_t)(r0 + r) * kL * kD + rem];
        }
        __syncthreads();
      }
      for (int item = tid; item < nr * n_pairs; item += kT) {
        const int rr = item / n_pairs, r = r0 + rr, pr = p_lo + (item - rr * n_pairs);
        const int i = pr / kD, j = pr - i * kD;
        float s2 = 0.f, s3[kDP];
#pragma unroll
        for (int k = 0; k < kDP; ++k) s3[k] = 0.f;
        const float* row = MODE == 0 ? pl + rr * kL * kDP : src + (size_t)r * kL * kD;
        constexpr int RP = MODE == 0 ? kDP : kD;
        const float x0i = row[i];
        for (int l = 0; l + 1 < kL; ++l) {
          const float* a = row + l * RP;
          const float* b = a + RP;
          const float di = b[i] - a[i], dj = b[j] - a[j];
          const float s1i = a[i] - x0i;
          const float coef = s2 + (s1i + di * (1.0f / 3.0f)) * dj * 0.5f;
#pragma unroll
          for (int k = 0; k < kD; ++k) s3[k] = fmaf(coef, b[k] - a[k], s3[k]);
          s2 = fmaf(s1i + di * 0.5f, dj, s2);
        }
        // the pair's columns that fall into this k-slice -> the tile
#pragma unroll
        for (int k = 0; k < kD; ++k) {
          const int col = pr * kD + k - col0;
          if (col >= 0 && col < kCols) tile[r * kPitch + col] = s3[k];
        }
      }
    }
    __syncthreads();
    acc += tile[(tid % kRows) * kPitch + (tid * 7) % kCols];
    __syncthreads();
  }
  if (acc == 12345.678f) sink[0] = acc;
}

int main() {
  const int grid = 256, iters = 50;
  std::vector<float> h((size_t)64 * kRows * kL * kD);
  for (size_t e = 0; e < h.size(); ++e) h[e] = (float)((e * 2654435761u) % 1000) * 1e-3f;
  float *paths, *sink;
  hipMalloc(&paths, h.size() * 4); hipMalloc(&sink, 4);
  hipMemcpy(paths, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    const size_t lds = ((size_t)kRows * kPitch + (mode == 0 ? (size_t)kRP * kL * kDP : 0)) * 4;
    auto k = mode == 0 ? rebuild<0> : rebuild<1>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
      printf("mode %d: cannot get %zu B of LDS\n", mode, lds);
      continue;
    }
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(grid), dim3(kT), lds, 0, paths, iters, 5000, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
      if (hipGetLastError() != hipSuccess) { printf("mode %d: launch failed\n", mode); break; }
      best = ms < best ? ms : best;
    }
    printf("mode %d (%s): %.2f us per rebuild of a [104 x 256] level-3 tile (LDS %zu B, %d workgroups at once)\n",
           mode, mode == 0 ? "paths staged in LDS" : "paths read from global / L2", best * 1e3f / iters, lds, grid);
  }
  return 0;
}
