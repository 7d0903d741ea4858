// Micro-benchmark: cost of a grid-wide barrier (one workgroup per CU, agent-scope
// fences + one atomic counter) on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 -o gb grid_barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// mode 0/1: one counter (relaxed RMW + relaxed polling), 1 = with agent fences
// mode 2/3: flag per workgroup ({epoch} stores, wave 0 polls all G flags), 3 = with fences
__device__ inline void grid_barrier(unsigned* ctr, unsigned target, int mode, unsigned epoch) {
  const bool fences = mode & 1;
  if (fences) __threadfence();
  __syncthreads();
  if (mode < 2) {
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) break;
      }
    }
  } else {
    unsigned* flags = ctr + 16;
    if (threadIdx.x == 0)
      __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {
      const unsigned G = gridDim.x;
      unsigned spins = 0;
      bool ok[4];
      for (int u = 0; u < 4; ++u) ok[u] = threadIdx.x + 64 * u >= G;
      for (;;) {
        for (int u = 0; u < 4; ++u)
          if (!ok[u]) ok[u] = __hip_atomic_load(flags + threadIdx.x + 64 * u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT) >= epoch;
        if (__all(ok[0] && ok[1] && ok[2] && ok[3])) break;
        if (++spins > (1u << 22)) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  __syncthreads();
  if (fences) __threadfence();
}

__global__ void bench(unsigned* ctr, float* buf, int n_iter, int payload, int fences) {
  const unsigned G = gridDim.x;
  float acc = 0.f;
  for (int it = 0; it < n_iter; ++it) {
    // payload: every WG writes `payload` floats, after the barrier reads its neighbour's
    float* wr = buf + (size_t)(it & 1) * 256 * 65536 / 2 + (size_t)blockIdx.x * payload;
    if (fences >= 4) {   // mode 4/5: write-through payload (agent-scope relaxed atomics), no fences
      for (int i = threadIdx.x * 2; i < payload; i += blockDim.x * 2) {
        const unsigned long long v = ((unsigned long long)__float_as_uint((float)(it + i + 1)) << 32) |
                                     __float_as_uint((float)(it + i));
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(wr + i), v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_s_waitcnt(0);
    } else {
      for (int i = threadIdx.x; i < payload; i += blockDim.x) wr[i] = (float)(it + i);
    }
    grid_barrier(ctr, G * (unsigned)(it + 1), fences >= 4 ? (fences == 4 ? 0 : 2) : fences, (unsigned)(it + 1));
    const unsigned nb = (blockIdx.x + 37) % G;
    float* rd = buf + (size_t)(it & 1) * 256 * 65536 / 2 + (size_t)nb * payload;
    if (fences >= 4) {
      for (int i0 = threadIdx.x * 2; i0 < payload; i0 += blockDim.x * 8) {
        unsigned long long q[4];
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * blockDim.x * 2;
          q[u] = i < payload ? __hip_atomic_load(reinterpret_cast<unsigned long long*>(rd + i),
                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * blockDim.x * 2;
          if (i < payload)
            acc += fabsf(__uint_as_float((unsigned)q[u]) - (float)(it + i)) +
                   fabsf(__uint_as_float((unsigned)(q[u] >> 32)) - (float)(it + i + 1));
        }
      }
    } else {
      for (int i = threadIdx.x; i < payload; i += blockDim.x) acc += fabsf(rd[i] - (float)(it + i));
    }
  }
  if (acc != 0.f) buf[0] = -1234.5f;   // stale read detector
  if (acc != 0.f && threadIdx.x == 0) atomicAdd(ctr + 1, 1u);
}

int main(int argc, char** argv) {
  int n_iter = 400;
  unsigned* ctr; float* buf;
  hipMalloc(&ctr, 4096); hipMalloc(&buf, (size_t)256 * 65536 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int G : {64, 128, 252}) for (int payload : {1024, 4096, 16384}) for (int fences : {0, 2, 4, 5}) {
    float best = 1e9;
    unsigned bad = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipMemset(ctr, 0, 4096);
      hipEventRecord(e0);
      hipLaunchKernelGGL(bench, dim3(G), dim3(256), 0, 0, ctr, buf, n_iter, payload, fences);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
      unsigned h[2]; hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost); bad += h[1];
    }
    printf("G=%3d payload=%6d floats/WG mode=%d : %.2f us per (write + barrier + read)  stale=%u\n", G, payload, fences,
           best * 1e3 / n_iter, bad);
  }
  return 0;
}
