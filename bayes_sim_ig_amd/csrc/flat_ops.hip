// Flat-buffer helpers of the fit loop: fused Adam over the flat parameter
// buffer (torch.optim.Adam defaults, mdnn.py:203,234), column sums for the
// bias gradients, theta normalisation (mdnn.py:245-248), strided row copies.
#include "common.h"

#include <algorithm>
#include <cmath>

namespace bsig {

struct AdamScalars {  // resolved on the host or by the fit engine's step kernel
  float lr, beta1, beta2, eps;
  float step_size;       // lr / (1 - beta1^t)
  float inv_bc2_sqrt;    // 1 / sqrt(1 - beta2^t)
};

// torch's single-tensor Adam (torch/optim/adam.py, no amsgrad / weight decay):
//   m <- m + (g - m) * (1 - b1)        (lerp)
//   v <- v * b2 + (1 - b2) * g * g
//   p <- p - step_size * m / (sqrt(v) / sqrt(bc2) + eps)
// `dyn` (device, optional) overrides step_size / inv_bc2_sqrt so that a HIP
// graph can replay the kernel for every step.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p,
                                                   const float* __restrict__ g,
                                                   float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n,
                                                   AdamScalars sc,
                                                   const float* __restrict__ dyn) {
  float step_size = sc.step_size, inv_bc2_sqrt = sc.inv_bc2_sqrt;
  if (dyn) { step_size = dyn[0]; inv_bc2_sqrt = dyn[1]; }
  const float omb1 = 1.0f - sc.beta1, omb2 = 1.0f - sc.beta2;
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
#define BSIG_ADAM1(c)                                                         \
    mm.c = mm.c + (gg.c - mm.c) * omb1;                                       \
    vv.c = vv.c * sc.beta2 + omb2 * gg.c * gg.c;                              \
    pp.c = pp.c - step_size * (mm.c / (sqrtf(vv.c) * inv_bc2_sqrt + sc.eps));
    BSIG_ADAM1(x) BSIG_ADAM1(y) BSIG_ADAM1(z) BSIG_ADAM1(w)
#undef BSIG_ADAM1
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += stride) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * omb1;
    const float vi = v[i] * sc.beta2 + omb2 * gi * gi;
    p[i] = p[i] - step_size * (mi / (sqrtf(vi) * inv_bc2_sqrt + sc.eps));
    m[i] = mi;
    v[i] = vi;
  }
}

int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                float beta1, float beta2, float eps, int64_t t, const float* dyn,
                hipStream_t st) {
  BSIG_REQUIRE(p && g && m && v && n >= 0, "adam: bad args");
  BSIG_REQUIRE(aligned(p, 16) && aligned(g, 16) && aligned(m, 16) && aligned(v, 16),
               "adam: buffers must be 16-byte aligned");
  if (n == 0) return BSIG_OK;
  AdamScalars sc;
  sc.lr = lr; sc.beta1 = beta1; sc.beta2 = beta2; sc.eps = eps;
  const double bc1 = 1.0 - std::pow((double)beta1, (double)t);
  const double bc2 = 1.0 - std::pow((double)beta2, (double)t);
  sc.step_size = (float)((double)lr / bc1);
  sc.inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
  const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>(ceil_div<int64_t>(n, 4), 256), 2048);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, p, g, m, v, n, sc, dyn);
  BSIG_CHECK_LAUNCH("adam");
  return BSIG_OK;
}

// out[j] = sum_i x[i*ld + j]; block = 64 columns x 4 row lanes, grid.y = row
// slabs; two deterministic stages for tall inputs.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld,
                                                     int64_t rows, int64_t cols,
                                                     int64_t rows_per_slab,
                                                     float* __restrict__ out) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t j = (int64_t)blockIdx.x * 64 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r1 = (r0 + rows_per_slab < rows) ? r0 + rows_per_slab : rows;
  float acc = 0.f;
  if (j < cols)
    for (int64_t i0 = r0 + rl; i0 < r1; i0 += 32) {   // 8 rows in flight per thread
      float q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t i = (i0 + 4 * u < r1) ? i0 + 4 * u : r1 - 1;
        q[u] = x[i * ld + j];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + 4 * u < r1) acc += q[u];
    }
  part[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && j < cols)
    out[(int64_t)blockIdx.y * cols + j] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

int colsum_launch(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                  void* workspace, size_t workspace_bytes, hipStream_t st) {
  BSIG_REQUIRE(x && out && rows >= 0 && cols >= 1 && ld >= cols, "colsum: bad args");
  int64_t slabs = 1;
  if (rows > 1024) slabs = std::min<int64_t>(ceil_div<int64_t>(rows, 512), 64);
  if (slabs > 1 && (!workspace || workspace_bytes < (size_t)slabs * cols * sizeof(float)))
    slabs = 1;
  const int64_t per = ceil_div<int64_t>(std::max<int64_t>(rows, 1), slabs);
  float* stage = slabs > 1 ? reinterpret_cast<float*>(workspace) : out;
  hipLaunchKernelGGL(colsum_kernel, dim3((int)ceil_div<int64_t>(cols, 64), (int)slabs),
                     dim3(256), 0, st, x, ld, rows, cols, per, stage);
  BSIG_CHECK_LAUNCH("colsum");
  if (slabs > 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3((int)ceil_div<int64_t>(cols, 64), 1), dim3(256), 0,
                       st, stage, cols, slabs, cols, slabs, out);
    BSIG_CHECK_LAUNCH("colsum2");
  }
  return BSIG_OK;
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(
    const float* __restrict__ theta, int64_t ld_in, const float* __restrict__ lows,
    const float* __restrict__ highs, float* __restrict__ out, int64_t ld_out, int64_t rows,
    int64_t cols) {
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x)
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x)
      out[r * ld_out + c] = (theta[r * ld_in + c] - lows[c]) / (highs[c] - lows[c]);
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src,
                                                        int64_t ld_src,
                                                        const int32_t* __restrict__ rows,
                                                        float* __restrict__ dst,
                                                        int64_t ld_dst, int64_t n_rows,
                                                        int64_t cols) {
  for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const float* s = src + (rows ? (int64_t)rows[r] : r) * ld_src;
    float* d = dst + r * ld_dst;
    for (int64_t c = threadIdx.x; c < cols; c += blockDim.x) d[c] = s[c];
  }
}

// The same for rows that are 16-byte aligned on both sides (every staging copy of the fit: the
// summarizers write rows with a leading dimension that is a multiple of four): float4 moves, four
// in flight per thread.  The scalar kernel moved the 0.9 GB of a 100k-row ShadowHand chunk at
// 0.2 TB/s (8.7 ms, 11 % of a scaled-batch fit).
__global__ __launch_bounds__(256) void copy_rows_vec_kernel(const float* __restrict__ src,
                                                            int64_t ld_src,
                                                            const int32_t* __restrict__ rows,
                                                            float* __restrict__ dst,
                                                            int64_t ld_dst, int64_t n_rows,
                                                            int64_t cols) {
  const int64_t c4n = cols >> 2;
  for (int64_t r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const float* s = src + (rows ? (int64_t)rows[r] : r) * ld_src;
    float* d = dst + r * ld_dst;
    const float4* s4 = reinterpret_cast<const float4*>(s);
    float4* d4 = reinterpret_cast<float4*>(d);
    for (int64_t c0 = threadIdx.x; c0 < c4n; c0 += 4 * blockDim.x) {
      float4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t c = c0 + u * blockDim.x;
        if (c < c4n) q[u] = s4[c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t c = c0 + u * blockDim.x;
        if (c < c4n) d4[c] = q[u];
      }
    }
    for (int64_t c = (c4n << 2) + threadIdx.x; c < cols; c += blockDim.x) d[c] = s[c];
  }
}

}  // namespace bsig

using namespace bsig;

extern "C" int bsig_adam_flat(float* params, const float* grads, float* exp_avg,
                              float* exp_avg_sq, int64_t n, float lr, float beta1,
                              float beta2, float eps, int64_t t, bsig_stream_t stream) {
  BSIG_REQUIRE(t >= 1, "adam: step number is 1-based");
  return adam_launch(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, t, nullptr,
                     as_stream(stream));
}

extern "C" int bsig_colsum(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                           void* workspace, size_t workspace_bytes, bsig_stream_t stream) {
  return colsum_launch(x, ld, rows, cols, out, workspace, workspace_bytes, as_stream(stream));
}

extern "C" int bsig_normalize_rows(const float* theta, int64_t ld_in, const float* lows,
                                   const float* highs, float* out, int64_t ld_out,
                                   int64_t rows, int64_t cols, bsig_stream_t stream) {
  BSIG_REQUIRE(theta && lows && highs && out && rows >= 0 && cols >= 1, "normalize: bad args");
  if (rows == 0) return BSIG_OK;
  hipLaunchKernelGGL(normalize_rows_kernel, dim3((int)std::min<int64_t>(rows, 4096)),
                     dim3(cols >= 192 ? 256 : 64), 0, as_stream(stream), theta, ld_in, lows,
                     highs, out, ld_out, rows, cols);
  BSIG_CHECK_LAUNCH("normalize_rows");
  return BSIG_OK;
}

extern "C" int bsig_copy_rows(const float* src, int64_t ld_src, const int32_t* rows,
                              float* dst, int64_t ld_dst, int64_t n_rows, int64_t cols,
                              bsig_stream_t stream) {
  BSIG_REQUIRE(src && dst && n_rows >= 0 && cols >= 0, "copy_rows: bad args");
  if (n_rows == 0 || cols == 0) return BSIG_OK;
  const bool vec = cols >= 256 && ld_src % 4 == 0 && ld_dst % 4 == 0 &&
                   ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
  if (vec)
    hipLaunchKernelGGL(copy_rows_vec_kernel, dim3((int)std::min<int64_t>(n_rows, 16384)), dim3(256), 0,
                       as_stream(stream), src, ld_src, rows, dst, ld_dst, n_rows, cols);
  else
    hipLaunchKernelGGL(copy_rows_kernel, dim3((int)std::min<int64_t>(n_rows, 8192)),
                       dim3(cols >= 192 ? 256 : 64), 0, as_stream(stream), src, ld_src, rows,
                       dst, ld_dst, n_rows, cols);
  BSIG_CHECK_LAUNCH("copy_rows");
  return BSIG_OK;
}
