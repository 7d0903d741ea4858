// Micro-benchmark: ONE payload read by MANY workgroups (the head matrix every row owner reloads each
// update, the activations every small-weight workgroup reads) on gfx950.  P producer workgroups write
// a payload of `floats` (write-through stores), raise flags; C consumer workgroups wait for all the
// flags and each read the WHOLE payload into LDS; then the consumers raise flags and the producers
// write the next version.  Consumer loads:
//   mode 0: cache-bypassing 16-byte loads (`sc1`: what the persistent kernels use) -- every consumer
//           pulls every byte through its own CU's path to memory;
//   mode 1: `buffer_inv sc1` (agent-scope acquire: drops the stale lines of this XCD's L2 and of the
//           L1) after the flags, then PLAIN 16-byte loads -- the first consumer of an XCD misses, the
//           others hit in that L2.
// Every word carries its version, so stale reads are counted.
// Build: hipcc --offload-arch=gfx950 -O3 -o fanout fanout_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t rsrc(const unsigned* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p), 0, 0x7fffffff, 0x00020000);
}
__device__ inline void put(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline unsigned get(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void wait_all(const unsigned* flags, int n, unsigned epoch, int lane) {
  bool ok[4];
  for (int u = 0; u < 4; ++u) ok[u] = lane + 64 * u >= n;
  for (unsigned spin = 0; spin < (1u << 14); ++spin) {
    for (int u = 0; u < 4; ++u)
      if (!ok[u]) ok[u] = get(flags + (size_t)(lane + 64 * u) * 16) >= epoch;
    if (__all(ok[0] && ok[1] && ok[2] && ok[3])) break;
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void fanout(unsigned* data, unsigned* fp, unsigned* fc, int P, int C, int words,
                                              int iters, unsigned* stale, long long* cycles) {
  extern __shared__ unsigned lds[];
  const int wg = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const bool producer = wg < P;
  long long t0 = 0, tload = 0;
  unsigned bad = 0;
  const __amdgpu_buffer_rsrc_t r = rsrc(data);
  for (int it = 1; it <= iters; ++it) {
    if (it == 2 && tid == 0) t0 = wall_clock64();
    if (producer) {
      // wait until every consumer has read version it - 1
      if (tid < 64 && it > 1) wait_all(fc, C, (unsigned)(it - 1), lane);
      __syncthreads();
      const int per = words / P;
      for (int i = tid * 4; i < per; i += blockDim.x * 4) {
        const unsigned b = (unsigned)it * 0x10000u + (unsigned)((wg * per + i) & 0xffff);
        const u32x4 v = {b, b + 1, b + 2, b + 3};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, (wg * per + i) * 4, 0, 16);
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0) put(fp + (size_t)wg * 16, (unsigned)it);
    } else {
      if (tid < 64) wait_all(fp, P, (unsigned)it, lane);
      __syncthreads();
      const long long a = wall_clock64();
      if (MODE == 1) asm volatile("buffer_inv sc1" ::: "memory");
      for (int base = 0; base < words; base += blockDim.x * 4 * 8) {
        u32x4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + (u * blockDim.x + tid) * 4;
          const u32x4 z = {0, 0, 0, 0};
          q[u] = i < words ? __builtin_amdgcn_raw_buffer_load_b128(r, i * 4, 0, MODE == 0 ? 16 : 0) : z;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = base + (u * blockDim.x + tid) * 4;
          if (i < words) {
            const unsigned b = (unsigned)it * 0x10000u + (unsigned)(i & 0xffff);
            bad += (q[u].x != b) + (q[u].y != b + 1) + (q[u].z != b + 2) + (q[u].w != b + 3);
            *reinterpret_cast<u32x4*>(lds + (i & 0x3fff)) = q[u];
          }
        }
      }
      __syncthreads();
      if (tid == 0) { tload += wall_clock64() - a; put(fc + (size_t)(wg - P) * 16, (unsigned)it); }
    }
  }
  if (tid == 0) { cycles[wg] = wall_clock64() - t0; cycles[256 + wg] = tload; }
  if (bad) atomicAdd(stale, bad);
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int iters = 100;
  unsigned *data, *fp, *fc, *stale; long long* cycles;
  hipMalloc(&data, 1 << 20); hipMalloc(&fp, 256 * 64); hipMalloc(&fc, 256 * 64); hipMalloc(&stale, 4);
  hipMalloc(&cycles, 512 * 8);
  for (int C : {13, 25, 100})
    for (int words : {4096, 16384, 36864})      // 16 KB, 64 KB (h2 + a gradient block), 144 KB (a 288-row head matrix)
      for (int mode = 0; mode < 2; ++mode) {
        const int P = 16;
        hipMemset(fp, 0, 256 * 64); hipMemset(fc, 0, 256 * 64); hipMemset(stale, 0, 4); hipMemset(data, 0, 1 << 20);
        if (mode == 0) hipLaunchKernelGGL(fanout<0>, dim3(P + C), dim3(512), 64 * 1024 + 4096, 0, data, fp, fc, P, C, words, iters, stale, cycles);
        else hipLaunchKernelGGL(fanout<1>, dim3(P + C), dim3(512), 64 * 1024 + 4096, 0, data, fp, fc, P, C, words, iters, stale, cycles);
        hipDeviceSynchronize();
        std::vector<long long> hc(512);
        unsigned hs;
        hipMemcpy(hc.data(), cycles, 512 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost);
        double mx = 0, ld = 0;
        for (int i = 0; i < P + C; ++i) mx = hc[i] > mx ? hc[i] : mx;
        for (int i = P; i < P + C; ++i) ld = hc[256 + i] > ld ? hc[256 + i] : ld;
        printf("C=%3d consumers x %6d KB  mode %d: %.2f us per round, slowest consumer's load %.2f us  stale=%u\n", C,
               words * 4 / 1024, mode, mx / 100.0 / (iters - 1), ld / 100.0 / iters, hs);
      }
  return 0;
}
