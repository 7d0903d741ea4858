// Device helpers shared by the persistent update kernels (fit_persistent.hip:
// linear heads on cached features; fit_persistent_mdnn.hip: two-layer trunk).
#pragma once
#include "head_device.h"

namespace bsig {

// row of element i of a 32x32 MFMA accumulator held by lane half h
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// One Adam step of a weight element / a bias element.  Written with explicit
// fused operations so that every call site (the resident run, a data-parallel
// rank's pending step) rounds identically whatever the surrounding code.
struct AdamK { float ob1, b2f, ob2, eps; };
__device__ __forceinline__ float adam_weight(float g, float& m, float& v, float w, float a0,
                                             float a1, const AdamK& k) {
  m = __builtin_fmaf(g - m, k.ob1, m);
  v = __builtin_fmaf(k.ob2 * g, g, v * k.b2f);
  // v_sqrt_f32 / v_rcp_f32 (1 ulp each) instead of the IEEE sequences
  const float r = __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_sqrtf(v), a1, k.eps));
  return __builtin_fmaf(-a0, m * r, w);
}
__device__ __forceinline__ float adam_bias(float g, float& m, float& v, float w, float a0,
                                           float a1, const AdamK& k) {
  m = __builtin_fmaf(g - m, k.ob1, m);
  v = __builtin_fmaf(k.ob2 * g, g, v * k.b2f);
  return __builtin_fmaf(-a0, m / __builtin_fmaf(sqrtf(v), a1, k.eps), w);
}

// workgroup-uniform test of the time-out bit (set by any bounded poll on the chip)
__device__ __forceinline__ bool run_aborted(int32_t* flagp, float* red, int tid) {
  if (tid == 0)
    red[63] = (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) ? 1.f : 0.f;
  __syncthreads();
  return red[63] != 0.f;
}


// 8-byte cache-bypassing load (two adjacent floats) for data that crosses workgroups
__device__ inline float2 xwg_load2(const float* p) {
  const unsigned long long x = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32)));
}

}  // namespace bsig
