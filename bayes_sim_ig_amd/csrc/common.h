// Shared helpers for libbsig_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/bsig.h"

namespace bsig {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(bsig_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// roctx range around a C-ABI entry point (api.cpp: bound at run time, a no-op without roctx)
void range_push(const char* name);
void range_pop();
struct Range {
  explicit Range(const char* name) { range_push(name); }
  ~Range() { range_pop(); }
  Range(const Range&) = delete;
  Range& operator=(const Range&) = delete;
};

// Report a launch failure without synchronising (works under stream capture).
#define BSIG_CHECK_LAUNCH(what)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      ::bsig::set_error("%s: %s", what, hipGetErrorString(e__));             \
      return BSIG_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

#define BSIG_HIP(call)                                                       \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess) {                                                 \
      ::bsig::set_error("%s: %s", #call, hipGetErrorString(e__));            \
      return BSIG_ELAUNCH;                                                   \
    }                                                                        \
  } while (0)

#define BSIG_REQUIRE(cond, ...)                                              \
  do {                                                                       \
    if (!(cond)) {                                                           \
      ::bsig::set_error(__VA_ARGS__);                                        \
      return BSIG_EINVAL;                                                    \
    }                                                                        \
  } while (0)

#define BSIG_TRY(call)                                                       \
  do {                                                                       \
    int rc__ = (call);                                                       \
    if (rc__ != BSIG_OK) return rc__;                                        \
  } while (0)

template <typename T>
__host__ __device__ inline T ceil_div(T a, T b) { return (a + b - 1) / b; }
template <typename T>
__host__ __device__ inline T round_up(T a, T b) { return ceil_div(a, b) * b; }

inline bool aligned(const void* p, size_t bytes) {
  return (reinterpret_cast<uintptr_t>(p) % bytes) == 0;
}

// ---- wave / block reductions (64-lane wavefronts) -------------------------
__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// Same sum through the DPP row shifts / broadcasts of GFX9 (no LDS crossbar
// round trips): inclusive scan inside the rows of 16, then row 0 -> 1, 2 -> 3,
// rows 0-1 -> 2-3; lane 63 holds the total.  Deterministic, but a different
// association than wave_sum.
__device__ inline float wave_sum_dpp(float v) {
#define BSIG_DPP_ADD(ctrl, row_mask)                                                        \
  v += __builtin_bit_cast(                                                                  \
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
  BSIG_DPP_ADD(0x111, 0xf);   // row_shr:1
  BSIG_DPP_ADD(0x112, 0xf);   // row_shr:2
  BSIG_DPP_ADD(0x114, 0xf);   // row_shr:4
  BSIG_DPP_ADD(0x118, 0xf);   // row_shr:8
  BSIG_DPP_ADD(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
  BSIG_DPP_ADD(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef BSIG_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ inline float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}
// Sum over the whole block; `scratch` holds >= blockDim.x/64 floats.
// Every thread gets the result.  Deterministic order.
__device__ inline float block_sum(float v, float* scratch) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}

// ---- Philox4x32-10 counter RNG (jitter noise, mdnn.py:116 rand_like) ------
struct Philox4 {
  uint32_t v[4];
};
__device__ inline Philox4 philox4x32_10(uint64_t seed, uint64_t stream_id, uint64_t ctr) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32);
  uint32_t c2 = (uint32_t)stream_id, c3 = (uint32_t)(stream_id >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{{c0, c1, c2, c3}};
}
__device__ inline float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

}  // namespace bsig
