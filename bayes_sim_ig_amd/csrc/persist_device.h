// Device helpers shared by the persistent update kernels (fit_persistent.hip:
// linear heads on cached features; fit_persistent_mdnn.hip: two-layer trunk).
#pragma once
#include "head_device.h"

namespace bsig {

// row of element i of a 32x32 MFMA accumulator held by lane half h
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
// The compile-time part of acc_row: row(i, h) = acc_row0(i) + 4 * h.
__device__ __forceinline__ constexpr int acc_row0(int i) { return (i & 3) + 8 * (i >> 2); }
// A pointer the compiler must treat as new at this point.  The 16 row addresses of an accumulator
// store are invariant across the update loop; hoisted out of it they do not fit the register file and
// come back as scratch reloads, and the `s_waitcnt vmcnt(0)` in front of each reload's use makes every
// write-through store wait for the acknowledgement of the one before (0.4 us each, measured).  Deriving
// them from a fresh base keeps them two VALU instructions next to the store.
template <typename T>
__device__ __forceinline__ T* fresh_ptr(T* p) { asm volatile("" : "+v"(p)); return p; }
__device__ __forceinline__ int fresh_value(int v) { asm volatile("" : "+v"(v)); return v; }

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
// Two elements at a time on the packed fp32 instructions (v_pk_add / v_pk_mul / v_pk_fma: one issue slot
// for both): the same operations, rounded the same way, as adam_weight on each -- bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void adam_weight2(float g0, float g1, float& m0, float& m1, float& v0, float& v1,
                                             float& w0, float& w1, float a0, float a1, const AdamK& k) {
  const f32x2 g = {g0, g1};
  f32x2 m = {m0, m1}, v = {v0, v1}, w = {w0, w1};
  const f32x2 ob1 = {k.ob1, k.ob1}, ob2 = {k.ob2, k.ob2}, b2f = {k.b2f, k.b2f}, a1v = {a1, a1}, eps = {k.eps, k.eps},
              na0 = {-a0, -a0};
  m = __builtin_elementwise_fma(g - m, ob1, m);
  v = __builtin_elementwise_fma(ob2 * g, g, v * b2f);
  const f32x2 sq = {__builtin_amdgcn_sqrtf(v[0]), __builtin_amdgcn_sqrtf(v[1])};
  const f32x2 d = __builtin_elementwise_fma(sq, a1v, eps);
  const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  w = __builtin_elementwise_fma(na0, m * r, w);
  m0 = m[0]; m1 = m[1]; v0 = v[0]; v1 = v[1]; w0 = w[0]; w1 = w[1];
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


// Workgroup barrier that waits for this wavefront's LDS traffic only.  __syncthreads() lowers to
// s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier: with global loads (a prefetched tile) or write-through
// stores in flight every wavefront would first wait for their round trip.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// 8-byte cache-bypassing load (two adjacent floats) for data that crosses workgroups
__device__ inline float2 xwg_load2(const float* p) {
  const unsigned long long x = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((uint32_t)x), __uint_as_float((uint32_t)(x >> 32)));
}

// 16-byte write-through stores / cache-bypassing loads for the large cross-workgroup payloads
// (slabs of partial products, gradient rows).  tools/micro/handoff_bench.hip: a 3200-dword payload
// crosses workgroups in 2.3 us with four dwords per lane against 4.5 us with one (same cache policy,
// `sc1`, same coherence -- 0 stale words); the time is in the number of memory transactions, not in
// the bytes.  Buffer instructions through the clang builtins, so the compiler tracks them (several
// in flight per lane, exact `s_waitcnt`s).  `base` must be workgroup-uniform; byte offsets are 32-bit.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kXwgPolicy = 16;          // gfx940+: sc1
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xwg_buffer(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
// ... a buffer of `bytes` bytes: loads beyond it return zeros, stores beyond it are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xwg_buffer_n(const float* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
}
// ... 16 bytes at byte offset voff (per lane) + soff (uniform)
__device__ __forceinline__ void xwg_store4s(__amdgpu_buffer_rsrc_t r, int voff, int soff, f32x4 q) {
  const u32x4 v = {__float_as_uint(q[0]), __float_as_uint(q[1]), __float_as_uint(q[2]), __float_as_uint(q[3])};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff, soff, kXwgPolicy);
}
__device__ __forceinline__ f32x4 xwg_load4s(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, kXwgPolicy);
  f32x4 f = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  return f;
}
__device__ __forceinline__ void xwg_store4(__amdgpu_buffer_rsrc_t r, int float_off, float a, float b,
                                           float c, float d) {
  const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, float_off * 4, 0, kXwgPolicy);
}
__device__ __forceinline__ f32x4 xwg_load4(__amdgpu_buffer_rsrc_t r, int float_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, float_off * 4, 0, kXwgPolicy);
  f32x4 f = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
  return f;
}

// A data-parallel rank RESIDENT across the gradient exchange (persist.h: PersistBuffers::xr_*), the
// hand-off of one update by one workgroup whose gradients are out and acknowledged (s_waitcnt(0) +
// barrier before the call): thread 0 counts the workgroup in, the last of the n_wg raises *ready to
// base + done_upd with ONE system-scope store, and waits (bounded) until the exchange stream has
// written the same number behind its all-reduce.  The caller follows with a workgroup barrier.
__device__ __forceinline__ void xr_hand_off(unsigned* count, unsigned* ready, const unsigned* done, unsigned base,
                                            unsigned done_upd, unsigned n_wg, int32_t* flagp) {
  const unsigned target = base + done_upd;
  const unsigned old = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old + 1u == done_upd * n_wg)
    __hip_atomic_store(ready, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  int spins = 0;
  while ((int)(__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target) < 0) {
    if (++spins > (1 << 23)) { atomicOr(flagp, 2); break; }
    // (a run that gave up elsewhere: the workgroup that would release the exchange stream may be this one)
    if ((spins & 1023) == 0 && (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2)) break;
    __builtin_amdgcn_s_sleep(8);
  }
}

// 4 x 4 transpose across the four lanes of a quad (lanes 4q .. 4q+3, a = lane & 3): lane a comes in with
// column a of a 4 x 4 block, c[r] = M[r][a], and leaves with row a, M[a][0..3].  A 32x32 accumulator lane
// holds four consecutive ROWS of one column; a cross-workgroup payload is stored 16 bytes per lane along a
// row (it costs by the number of memory transactions): two butterfly steps turn the one into the other.
__device__ __forceinline__ f32x4 quad_transpose4(float c0, float c1, float c2, float c3, int a) {
  const bool odd = a & 1, hi = a & 2;
  // step 1, partner a ^ 1: even lanes end with rows 0 and 2 of columns (a, a + 1), odd ones with rows 1 and 3
  const float r0 = __shfl_xor(odd ? c0 : c1, 1, 64), r1 = __shfl_xor(odd ? c2 : c3, 1, 64);
  const float pA0 = odd ? r0 : c0, pA1 = odd ? c1 : r0;      // row 0 / 1, the lane pair's two columns
  const float pB0 = odd ? r1 : c2, pB1 = odd ? c3 : r1;      // row 2 / 3
  // step 2, partner a ^ 2: lanes 0, 1 keep their row 0 / 1 pair and take the other pair's, lanes 2, 3 rows 2 / 3
  const float q0 = __shfl_xor(hi ? pA0 : pB0, 2, 64), q1 = __shfl_xor(hi ? pA1 : pB1, 2, 64);
  f32x4 out;
  out[0] = hi ? q0 : pA0; out[1] = hi ? q1 : pA1; out[2] = hi ? pB0 : q0; out[3] = hi ? pB1 : q1;
  return out;
}

}  // namespace bsig
