// Micro-benchmark: the many-to-many hand-off of the persistent update kernels on gfx950 --
// P producer workgroups each raise a flag (write-through store), C consumer workgroups each wait
// for ALL P flags (one wavefront polls them around the L2), then the roles swap -- as a function
// of how the flags are laid out in memory:
//   stride 1  : P consecutive dwords (what round 1 shipped: every poller hammers the same few
//               cache lines of one memory channel)
//   stride 32 : one flag per 128-byte line;  stride 64: one per 256 bytes
// and of a two-level scheme: consumer 0 polls the P flags and raises ONE word the others poll.
// Build: hipcc --offload-arch=gfx950 -O3 -o fanin fanin_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ inline void put(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned get(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one wavefront waits until flags[i * stride] >= epoch for i < n (n <= 256)
__device__ inline void wait_all(const unsigned* flags, int n, int stride, unsigned epoch, int lane, int sleep) {
  bool ok[4];
  for (int u = 0; u < 4; ++u) ok[u] = lane + 64 * u >= n;
  for (unsigned spin = 0; spin < (1u << 14); ++spin) {
    for (int u = 0; u < 4; ++u)
      if (!ok[u]) ok[u] = get(flags + (size_t)(lane + 64 * u) * stride) >= epoch;
    if (__all(ok[0] && ok[1] && ok[2] && ok[3])) break;
    if (sleep) __builtin_amdgcn_s_sleep(1);
  }
}

// workgroups [0, P) and [P, P + C) alternate as producers / consumers: one iteration = two fan-ins
__global__ __launch_bounds__(256) void fanin(unsigned* fa, unsigned* fb, unsigned* go, int P, int C,
                                             int stride, int two_level, int sleep, int iters,
                                             long long* cycles) {
  extern __shared__ float pad[];
  const int wg = blockIdx.x, lane = threadIdx.x & 63;
  const bool grp_a = wg < P;
  const int me = grp_a ? wg : wg - P;
  long long t0 = 0;
  for (int it = 1; it <= iters; ++it) {
    if (it == 2 && threadIdx.x == 0) t0 = wall_clock64();
    // phase 1: A raises, B waits for all of A
    if (grp_a) {
      __syncthreads();
      if (threadIdx.x == 0) put(fa + (size_t)me * stride, (unsigned)it);
    } else {
      if (threadIdx.x < 64) {
        if (!two_level || me == 0) {
          wait_all(fa, P, stride, (unsigned)it, lane, sleep);
          if (two_level && lane == 0) put(go, (unsigned)it);
        } else if (lane == 0) {
          for (unsigned spin = 0; spin < (1u << 14) && get(go) < (unsigned)it; ++spin) {}
        }
      }
      __syncthreads();
    }
    // phase 2: B raises, A waits for all of B
    if (!grp_a) {
      __syncthreads();
      if (threadIdx.x == 0) put(fb + (size_t)me * stride, (unsigned)it);
    } else {
      if (threadIdx.x < 64) {
        if (!two_level || me == 0) {
          wait_all(fb, C, stride, (unsigned)it, lane, sleep);
          if (two_level && lane == 0) put(go + 64, (unsigned)it);
        } else if (lane == 0) {
          for (unsigned spin = 0; spin < (1u << 14) && get(go + 64) < (unsigned)it; ++spin) {}
        }
      }
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) cycles[wg] = wall_clock64() - t0;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int iters = 200;
  unsigned *fa, *fb, *go; long long* cycles;
  const size_t fbytes = (size_t)256 * 64 * 4;
  hipMalloc(&fa, fbytes); hipMalloc(&fb, fbytes); hipMalloc(&go, 4096); hipMalloc(&cycles, 256 * 8);
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int P = cfg == 0 ? 144 : 188, C = cfg == 0 ? 100 : 25;
    for (int two_level = 0; two_level < 2; ++two_level)
      for (int sleep = 0; sleep < 2; ++sleep)
        for (int stride : {1, 2, 16, 32, 64}) {
          hipMemset(fa, 0, fbytes); hipMemset(fb, 0, fbytes); hipMemset(go, 0, 4096);
          hipLaunchKernelGGL(fanin, dim3(P + C), dim3(256), 100 * 1024, 0, fa, fb, go, P, C, stride,
                             two_level, sleep, iters, cycles);
          hipDeviceSynchronize();
          std::vector<long long> hc(256);
          hipMemcpy(hc.data(), cycles, 256 * 8, hipMemcpyDeviceToHost);
          double mx = 0;
          for (int i = 0; i < P + C; ++i) mx = hc[i] > mx ? hc[i] : mx;
          printf("P=%3d C=%3d two_level=%d sleep=%d stride=%2d dwords: %.2f us per fan-in\n", P, C, two_level,
                 sleep, stride, mx / 100.0 / (2.0 * (iters - 1)));
        }
  }
  return 0;
}
