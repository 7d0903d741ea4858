// Micro-benchmark: one-way latency of a producer -> consumer hand-off between two workgroups on
// gfx950 (payload + flag), for partners on the SAME XCD and on DIFFERENT XCDs, with
//   mode 0: write-through stores + loads around the L2 (agent-scope relaxed atomics) -- what the
//           persistent update kernels use for everything that crosses workgroups;
//   mode 2: as mode 0 with 16-byte stores and loads (four dwords per lane);
//   mode 3: as mode 2 through buffer_{load,store}_dwordx4 sc1 (clang builtins: loads pipelined);
//   mode 1: plain stores (the L1 writes through to the XCD's L2) + L1-bypassing `sc0` loads that may
//           hit in the L2: coherent only between workgroups that share an L2 (same XCD).
// Build: hipcc --offload-arch=gfx950 -O3 -o handoff handoff_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ inline unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ inline unsigned load_sc0(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ inline void store_plain(unsigned* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory");
}

// 16-byte write-through store / cache-bypassing load (the same cache policy bits as the agent-scope
// relaxed atomics above, four dwords per lane)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ inline void put4(unsigned* p, uint4 v) {
  const u32x4 r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(r) : "memory");
}
__device__ inline uint4 get4(const unsigned* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return make_uint4(r.x, r.y, r.z, r.w);
}

// the same through buffer instructions (compiler-tracked: several loads in flight per lane), aux 16 = sc1
__device__ inline __amdgpu_buffer_rsrc_t rsrc(const unsigned* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(p), 0, 0x7fffffff, 0x00020000);
}

template <int MODE>
__device__ inline void put(unsigned* p, unsigned v) {
  if (MODE != 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else store_plain(p, v);
}
template <int MODE>
__device__ inline unsigned get(const unsigned* p) {
  if (MODE != 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return load_sc0(p);
}

// workgroup `me` and `partner[me]` play ping-pong: payload floats + flag each way
template <int MODE>
__global__ __launch_bounds__(256) void pingpong(unsigned* xcc, const int* partner, unsigned* flags,
                                                unsigned* data, int payload, int iters,
                                                unsigned* stale, long long* cycles) {
  extern __shared__ float pad[];   // forces one workgroup per CU
  const int me = blockIdx.x, other = partner[me];
  if (threadIdx.x == 0) xcc[me] = xcc_id();
  if (other < 0) return;
  const bool first = me < other;
  unsigned* mine = data + (size_t)me * payload;
  const unsigned* theirs = data + (size_t)other * payload;
  unsigned bad = 0;
  long long t0 = 0;
  for (int it = 1; it <= iters; ++it) {
    if (it == 2 && threadIdx.x == 0) t0 = wall_clock64();
    for (int half = 0; half < 2; ++half) {
      const bool sender = (half == 0) == first;
      if (sender) {
        if (MODE == 2) {
          for (int i = threadIdx.x * 4; i < payload; i += blockDim.x * 4) {
            const unsigned b = (unsigned)(it * 2 + half + i);
            put4(mine + i, make_uint4(b, b + 1, b + 2, b + 3));
          }
        } else if (MODE == 3) {
          const __amdgpu_buffer_rsrc_t r = rsrc(mine);
          for (int i = threadIdx.x * 4; i < payload; i += blockDim.x * 4) {
            const unsigned b = (unsigned)(it * 2 + half + i);
            const u32x4 v = {b, b + 1, b + 2, b + 3};
            __builtin_amdgcn_raw_buffer_store_b128(v, r, i * 4, 0, 16);
          }
        } else
        for (int i = threadIdx.x; i < payload; i += blockDim.x) put<MODE>(mine + i, (unsigned)(it * 2 + half + i));
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (threadIdx.x == 0) put<MODE>(flags + me, (unsigned)(it * 2 + half));
      } else {
        if (threadIdx.x == 0) {
          unsigned spins = 0;
          while (get<MODE>(flags + other) != (unsigned)(it * 2 + half) && ++spins < (1u << 13)) {}
        }
        __syncthreads();
        if (MODE == 3) {
          const __amdgpu_buffer_rsrc_t r = rsrc(theirs);
#pragma unroll 4
          for (int i = threadIdx.x * 4; i < payload; i += blockDim.x * 4) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, i * 4, 0, 16);
            const unsigned b = (unsigned)(it * 2 + half + i);
            bad += (v.x != b) + (v.y != b + 1) + (v.z != b + 2) + (v.w != b + 3);
          }
        } else if (MODE == 2) {
          for (int i = threadIdx.x * 4; i < payload; i += blockDim.x * 4) {
            const uint4 v = get4(theirs + i);
            const unsigned b = (unsigned)(it * 2 + half + i);
            bad += (v.x != b) + (v.y != b + 1) + (v.z != b + 2) + (v.w != b + 3);
          }
        } else
        for (int i = threadIdx.x; i < payload; i += blockDim.x)
          bad += get<MODE>(theirs + i) != (unsigned)(it * 2 + half + i);
      }
    }
  }
  if (threadIdx.x == 0) cycles[me] = wall_clock64() - t0;
  if (bad) atomicAdd(stale, bad);
}

int main() {
  const int G = 256, iters = 100;
  setvbuf(stdout, nullptr, _IONBF, 0);
  unsigned *xcc, *flags, *data, *stale; int* partner; long long* cycles;
  hipMalloc(&xcc, G * 4); hipMalloc(&flags, G * 4); hipMalloc(&data, (size_t)G * 16384 * 4);
  hipMalloc(&stale, 4); hipMalloc(&partner, G * 4); hipMalloc(&cycles, G * 8);
  std::vector<int> none(G, -1);
  hipMemcpy(partner, none.data(), G * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(pingpong<0>, dim3(G), dim3(256), 100 * 1024, 0, xcc, partner, flags, data, 0, 0, stale, cycles);
  std::vector<unsigned> hx(G);
  hipMemcpy(hx.data(), xcc, G * 4, hipMemcpyDeviceToHost);
  printf("xcc of workgroups 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %u", hx[i]);
  printf("\n");
  int wrong = 0;
  for (int i = 0; i < G; ++i) wrong += hx[i] != (unsigned)(i % 8);
  printf("workgroups whose XCD != id %% 8: %d of %d\n", wrong, G);
  for (int same = 1; same >= 0; --same) {
    // pairs: (i, i + 8) on the same XCD if round-robin holds, (i, i + 1) on different ones
    std::vector<int> pr(G, -1);
    int npairs = 0;
    for (int i = 0; i + (same ? 8 : 1) < G && npairs < 64; i += 16) {
      const int j = i + (same ? 8 : 1);
      pr[i] = j; pr[j] = i; ++npairs;
    }
    hipMemcpy(partner, pr.data(), G * 4, hipMemcpyHostToDevice);
    for (int mode : {0, 2, 3})     // (mode 1 -- plain stores + sc0 loads -- is not coherent: see profiles/r02_handoff_bench.txt)
      for (int payload : {0, 64, 1024, 3200}) {
        hipMemset(flags, 0, G * 4); hipMemset(stale, 0, 4); hipMemset(cycles, 0, G * 8);
        if (mode == 3)
          hipLaunchKernelGGL(pingpong<3>, dim3(G), dim3(256), 100 * 1024, 0, xcc, partner, flags, data, payload, iters, stale, cycles);
        else if (mode == 2)
          hipLaunchKernelGGL(pingpong<2>, dim3(G), dim3(256), 100 * 1024, 0, xcc, partner, flags, data, payload, iters, stale, cycles);
        else if (mode == 0)
          hipLaunchKernelGGL(pingpong<0>, dim3(G), dim3(256), 100 * 1024, 0, xcc, partner, flags, data, payload, iters, stale, cycles);
        else
          hipLaunchKernelGGL(pingpong<1>, dim3(G), dim3(256), 100 * 1024, 0, xcc, partner, flags, data, payload, iters, stale, cycles);
        hipDeviceSynchronize();
        std::vector<long long> hc(G);
        unsigned hs;
        hipMemcpy(hc.data(), cycles, G * 8, hipMemcpyDeviceToHost);
        hipMemcpy(&hs, stale, 4, hipMemcpyDeviceToHost);
        double sum = 0; int n = 0;
        for (int i = 0; i < G; ++i) if (pr[i] >= 0) { sum += (double)hc[i]; ++n; }
        // wall_clock64: 100 MHz
        printf("%s XCD  mode %d  payload %5d dwords: %.2f us one way (%d pairs)  stale=%u\n",
               same ? "same" : "diff", mode, payload, sum / n / 100.0 / (2.0 * (iters - 1)), npairs, hs);
      }
  }
  return 0;
}
