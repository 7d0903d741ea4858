// How fast can 206 workgroups stream W, m, v [128, I] in and out (the Adam pass of the streamed
// first layer, fit_persistent_mdnn_stream.hip) for different tile shapes and access widths?
//   hipcc --offload-arch=gfx950 -O3 -o stream_pattern_bench stream_pattern_bench.hip
// Each workgroup owns a contiguous range of columns and walks tiles of R rows x C columns;
// lanes run along the columns (VW floats each).  Reports TB/s of (3 reads + 3 writes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int VW> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<4> { typedef float4 T; };

__device__ inline float upd(float w, float m, float v) { return w + 0.001f * m * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + 1e-8f); }

// mode 0: read W,m,v write W,m,v; 1: read only (sum into a sink); 2: write only
template <int VW, int MODE>
__global__ __launch_bounds__(512) void pass_kernel(float* W, float* M, float* V, int I, int cols_per_wg,
                                                  int R, int C, float* sink) {
  typedef typename Vec<VW>::T T;
  const int tid = threadIdx.x;
  const int k_lo = blockIdx.x * cols_per_wg, k_hi = min(k_lo + cols_per_wg, I);
  const int lanes_per_row = C / VW;            // threads along a row of the tile
  const int rows_per_pass = 512 / lanes_per_row;
  const int lc = (tid % lanes_per_row) * VW, lr = tid / lanes_per_row;
  float acc = 0.f;
  for (int k0 = k_lo; k0 < k_hi; k0 += C) {
    for (int r0 = 0; r0 < 128; r0 += R) {
      constexpr int MAXU = 16;
      T w[MAXU], m[MAXU], v[MAXU];
      const int nu = R / rows_per_pass;        // <= MAXU
      const bool ok = k0 + lc + VW <= k_hi;
#pragma unroll
      for (int u = 0; u < MAXU; ++u) {
        if (u < nu && ok && MODE != 2) {
          const size_t off = (size_t)(r0 + u * rows_per_pass + lr) * I + k0 + lc;
          w[u] = *reinterpret_cast<const T*>(W + off);
          m[u] = *reinterpret_cast<const T*>(M + off);
          v[u] = *reinterpret_cast<const T*>(V + off);
        }
      }
#pragma unroll
      for (int u = 0; u < MAXU; ++u) {
        if (u < nu && ok) {
          const size_t off = (size_t)(r0 + u * rows_per_pass + lr) * I + k0 + lc;
          if (MODE == 1) {
            const float* a = reinterpret_cast<const float*>(&w[u]);
            const float* b = reinterpret_cast<const float*>(&m[u]);
            const float* c = reinterpret_cast<const float*>(&v[u]);
            for (int e = 0; e < VW; ++e) acc += a[e] + b[e] + c[e];
          } else {
            T wo, mo, vo;
            float* a = reinterpret_cast<float*>(&wo); float* b = reinterpret_cast<float*>(&mo); float* c = reinterpret_cast<float*>(&vo);
            for (int e = 0; e < VW; ++e) {
              if (MODE == 0) {
                const float wi = reinterpret_cast<const float*>(&w[u])[e], mi = reinterpret_cast<const float*>(&m[u])[e],
                            vi = reinterpret_cast<const float*>(&v[u])[e];
                b[e] = mi * 0.9f + 0.1f; c[e] = vi * 0.999f + 0.001f; a[e] = upd(wi, b[e], c[e]);
              } else { a[e] = 1.f; b[e] = 2.f; c[e] = 3.f; }
            }
            *reinterpret_cast<T*>(W + off) = wo; *reinterpret_cast<T*>(M + off) = mo; *reinterpret_cast<T*>(V + off) = vo;
          }
        }
      }
    }
  }
  if (MODE == 1 && acc == 12345.678f) sink[0] = acc;
}

template <int VW, int MODE>
double run(float* W, float* M, float* V, int I, int wgs, int R, int C, float* sink, int reps) {
  const int cols = ((I + wgs - 1) / wgs + 63) / 64 * 64;   // (tiles clip at the range end)
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((pass_kernel<VW, MODE>), dim3(wgs), dim3(512), 0, 0, W, M, V, I, cols, R, C, sink);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pass_kernel<VW, MODE>), dim3(wgs), dim3(512), 0, 0, W, M, V, I, cols, R, C, sink);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const int I = argc > 1 ? atoi(argv[1]) : 105004;     // multiple of 4: aligned rows
  const int wgs = argc > 2 ? atoi(argv[2]) : 206;
  const size_t n = (size_t)128 * I;
  float *W, *M, *V, *sink;
  CK(hipMalloc(&W, n * 4 + 64)); CK(hipMalloc(&M, n * 4 + 64)); CK(hipMalloc(&V, n * 4 + 64)); CK(hipMalloc(&sink, 64));
  CK(hipMemset(W, 0, n * 4)); CK(hipMemset(M, 0, n * 4)); CK(hipMemset(V, 0x3c, n * 4));
  const double bytes_rw = 6.0 * n * 4, bytes_r = 3.0 * n * 4;
  printf("I = %d (%0.1f MB per array), %d workgroups of 512 threads\n", I, n * 4 / 1e6, wgs);
  struct Cfg { int vw, R, C; };
  const Cfg cfgs[] = {{1, 128, 64}, {1, 32, 256}, {2, 128, 64}, {2, 32, 256}, {4, 128, 64}, {4, 32, 256}, {4, 32, 512}, {4, 128, 256}};
  for (const Cfg& c : cfgs) {
    double t0, t1, t2;
    if (c.vw == 1) { t0 = run<1, 0>(W, M, V, I, wgs, c.R, c.C, sink, 20); t1 = run<1, 1>(W, M, V, I, wgs, c.R, c.C, sink, 20); t2 = run<1, 2>(W, M, V, I, wgs, c.R, c.C, sink, 20); }
    else if (c.vw == 2) { t0 = run<2, 0>(W, M, V, I, wgs, c.R, c.C, sink, 20); t1 = run<2, 1>(W, M, V, I, wgs, c.R, c.C, sink, 20); t2 = run<2, 2>(W, M, V, I, wgs, c.R, c.C, sink, 20); }
    else { t0 = run<4, 0>(W, M, V, I, wgs, c.R, c.C, sink, 20); t1 = run<4, 1>(W, M, V, I, wgs, c.R, c.C, sink, 20); t2 = run<4, 2>(W, M, V, I, wgs, c.R, c.C, sink, 20); }
    printf("VW %d tile %3d x %3d: read+write %7.1f us = %5.2f TB/s | read %7.1f us = %5.2f TB/s | write %7.1f us = %5.2f TB/s\n",
           c.vw, c.R, c.C, t0, bytes_rw / t0 / 1e6, t1, bytes_r / t1 / 1e6, t2, bytes_r / t2 / 1e6);
  }
  return 0;
}
