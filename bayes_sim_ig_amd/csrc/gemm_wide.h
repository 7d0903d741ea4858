// The two products of a LARGE-minibatch update of linear mixture-density heads (the scaled-batch
// fit: thousands of rows per update; mdnn.py:108-119 forward, the weight gradient of its backward)
//   forward          O[b, n]  = sum_f X[ids[b], f] W[n, f]         b < B (thousands), n < Nh (260)
//   weight gradient  dW[n, f] = sum_b dO[b, n]    X[ids[b], f]
// on tiles that span the WHOLE head width.  Nh = K (1 + 2 D) is a few hundred and no multiple of
// 32: on 32x32x2 MFMA tiles the 260 outputs of cfg5 pad to 288 (96-wide tiles: 11 % of every
// product is zeros) or 384.  Here the head dimension is NWT tiles of 16 (v_mfma_f32_16x16x4_f32,
// the same 64 flop/clk/SIMD): 272 for cfg5, 4.6 % padding, and the long operand X -- the gathered
// minibatch rows, 134 MB per update -- crosses the chip once per product instead of once per
// column tile.
//
// Workgroup = 8 wavefronts on a (64 narrow) x (NWT*16 wide) tile of one K slice; wavefront w owns
// narrow tile w & 3 and one half of the wide tiles (w >> 2: ceil / floor of NWT / 2 -- wavefronts
// w and w + 4 share a SIMD, so every SIMD carries NWT tiles).  LDS: two images of the K step
// (BK = 32), filled from registers (loads of step t+2 issued in the middle of step t, committed
// in the middle of step t+1: one barrier per step), no vector-ALU work in the loop beyond the
// pointer increments (gemm_lean.h).
//   forward (KMAJ = false): both operands k-contiguous, LDS rows of 40 floats (conflict-free
//     ds_read_b128 of 4 k per lane: lane (c16, g) takes k = 16 kb + 4 g .. + 3 of row c16).
//   gradient (KMAJ = true): both operands k-major in memory (dO [B, NWT*16]: a row pitch of exactly
//     NWT*16 floats, padding columns finite; X rows gathered through ids).  They are TRANSPOSED on
//     their way into LDS -- a thread loads 16 bytes of ONE k row and writes its four floats to four
//     LDS rows, consecutive lanes = consecutive k: conflict-free 4-byte stores -- so that the LDS
//     images, the fragment reads and the MFMA sequence are those of the forward product.  (Read
//     k-major with ds_read_b32, the compiler sank every read next to its MFMA and waited for each:
//     +15 us per launch.)
// The accumulator of a lane is four ADJACENT outputs along the contiguous axis of the result (head
// columns of one minibatch row / feature columns of one head row): 16-byte stores.
// Split K writes one slab per slice: [z][B][NWT*16] (forward; every column stored, the padding
// columns carry the products of clamped rows) or [z][Nh][F] (gradient).
#pragma once
#include "gemm_kernel.h"
#include "persist_device.h"

#include <type_traits>

namespace bsig {

typedef float floatx4 __attribute__((ext_vector_type(4)));

struct WideParams {
  const float* wide = nullptr; int64_t ld_wide = 0; int n_wide = 0;   // W [Nh, F] / dO [B, NWT*16]
  const float* x = nullptr; int64_t ldx = 0;                          // X [rows, F]
  const int32_t* ids = nullptr;                                        // minibatch row -> row of X (may be null)
  const int32_t* dyn = nullptr; int dyn_delta = 0; int64_t dyn_stride = 0, dyn_base = 0;
  int n_narrow = 0;     // forward: B; gradient: F
  int k = 0, k_chunk = 0, splits = 1;
  float* out = nullptr; int64_t ld_out = 0; int64_t slab = 0;   // floats between two K slices' slabs
  // forward only: combine the K slices INSIDE the launch.  Every workgroup writes its slab through,
  // draws a ticket of its narrow tile; the last of the tile's `splits` workgroups sums the slabs in
  // slice order (its own from its registers), adds the bias and stores final[b][0 .. NWT*16) -- no
  // reduce kernel, and with expsum one partial sum of exp(final[b][c]) over c in [expsum_col0,
  // expsum_col0 + expsum_ncols) per narrow tile (the jitter scale of the head, mdnn.py:115).
  // tickets: one zeroed int32 per narrow tile; the last arriver leaves it zeroed.
  int32_t* tickets = nullptr;
  const float* bias = nullptr;      // [n_wide]
  float* final_out = nullptr; int64_t ld_final = 0;
  float* expsum = nullptr; int expsum_col0 = 0, expsum_ncols = 0;
};

constexpr int kWidePitch = 40;   // k-contiguous LDS rows (BK = 32 + 8): conflict-free 16x16x4 fragment reads
constexpr int kWideNP = 80;      // k-major narrow rows (64 + 16)

template <int NWT, bool KMAJ>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wide_kernel(WideParams p) {
#ifndef BSIG_HOST_SAN_BUILD
  constexpr int NT = 512;
  constexpr int WIDE = NWT * 16;
  // 16-byte items of the wide image per thread.  forward: item = (row, k quad), rows 64 apart;
  // gradient: item = (k row, column quad), column quads 16 apart = rows 64 apart again.  The image
  // has 64 * kWItems rows: the last item of a thread needs no predicate (rows beyond WIDE are written,
  // from clamped addresses, and never read).
  constexpr int kWItems = (WIDE + 63) / 64;
  // forward: k-contiguous images [rows][40]; gradient: k-major images [32 k][80] and [32 k][WIDE],
  // the wide one one contiguous block of dO per step (pitch = 16 mod 32: conflict-free ds_read_b32 of
  // the four k rows the lane groups of a wavefront take), padded to a whole number of 512-thread items
  constexpr int kWLin = (WIDE * BK / 4 + NT - 1) / NT;  // gradient: linear 16-byte items per thread
  constexpr int kNs = KMAJ ? BK * kWideNP : 64 * kWidePitch;
  constexpr int kWs = KMAJ ? 4 * NT * kWLin : 64 * kWItems * kWidePitch;
  constexpr int kStage = kNs + kWs;
  constexpr int H0 = (NWT + 1) / 2, H1 = NWT / 2;                 // wide tiles of the two wavefront halves
  extern __shared__ __attribute__((aligned(16))) float wsm[];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c16 = lane & 15, g = lane >> 4;
  const int nt = w & 3, wh = __builtin_amdgcn_readfirstlane(w >> 2);   // (uniform: the halves branch apart)
  // Workgroup -> (narrow tile, K slice).  Consecutive workgroup ids go round the 8 XCDs (each with an
  // L2 of its own); with the remap an XCD works on a contiguous run of (slice, tile) pairs, i.e. on
  // ONE K slice where there are at most 8: its L2 then holds that slice of the shared operand only
  // (a quarter of dO, half of W) instead of all of it.  A bijection on [0, nwg): speed only.
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, nwg = gx * (int)gridDim.y;
    const int lin = bx + gx * by;
    const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    bx = wgid % gx; by = wgid / gx;
  }
  const int n0 = bx * 64;                 // first narrow index of this workgroup
  const int kbeg = by * p.k_chunk;
  const int kend = min(p.k, kbeg + p.k_chunk);
  const int nkt = (kend - kbeg) / BK;             // (whole steps only: the launcher checks)
  const int64_t dstep = p.dyn ? (int64_t)(p.dyn[0] + p.dyn_delta) : 0;
  const int64_t row_off = dstep * p.dyn_stride + p.dyn_base;

  // ---- loaders: one pointer per item, advanced by a constant per step -----------------------
  const float* np_;              // narrow item
  constexpr int kWRegs = KMAJ ? kWLin : kWItems;
  const float* wp[kWRegs];      // wide items
  int64_t nstep, wstep;
  const int32_t* ip = nullptr;   // gradient: &ids[k row of this thread's narrow item, tile after next]
  int nidx = 0, kleft = 0;
  if constexpr (!KMAJ) {
    const int q = tid & 7;
    const int64_t gr = min(n0 + (tid >> 3), p.n_narrow - 1) + row_off;
    const int64_t srow = p.ids ? (int64_t)p.ids[gr] : gr;
    np_ = p.x + srow * p.ldx + kbeg + 4 * q;
    nstep = BK;
#pragma unroll
    for (int it = 0; it < kWItems; ++it) {
      const int row = min((tid >> 3) + 64 * it, p.n_wide - 1);
      wp[it] = p.wide + (int64_t)row * p.ld_wide + kbeg + 4 * q;
    }
    wstep = BK;
  } else {
    const int kr = tid >> 4, q = tid & 15;        // narrow item: 16 lanes cover 256 contiguous bytes of a row
    // (the contraction rows are always gathered: the launcher insists on ids -- a run-time "ids or
    // not" inside the K loop split it into blocks the MFMAs cannot be spread over)
    np_ = p.x + n0 + 4 * q;                       // column of the item; the row comes from ids
    nstep = 0;
    nidx = p.ids[min(kbeg + kr, p.k - 1) + row_off];
    ip = p.ids + row_off + kbeg + BK + kr;
    kleft = p.k - 1 - (kbeg + BK + kr);
#pragma unroll
    for (int it = 0; it < kWLin; ++it)   // (dO: this minibatch's rows, a step of it one contiguous block)
      wp[it] = p.wide + (int64_t)kbeg * WIDE + 4 * min(tid + NT * it, WIDE * BK / 4 - 1);
    wstep = (int64_t)BK * WIDE;
  }
  floatx4 rn, rw[kWRegs];
  auto fetch = [&]() {
    if constexpr (KMAJ) {
      rn = *reinterpret_cast<const floatx4*>(np_ + (int64_t)nidx * p.ldx);
      nidx = ip[min(0, kleft)];   // the row of the step after: a whole step ahead of its use
      ip += BK; kleft -= BK;
    } else {
      rn = *reinterpret_cast<const floatx4*>(np_);
      np_ += nstep;
    }
#pragma unroll
    for (int it = 0; it < kWRegs; ++it) {
      rw[it] = *reinterpret_cast<const floatx4*>(wp[it]);
      wp[it] += wstep;
    }
  };
  auto commit = [&](float* st) {
    if constexpr (!KMAJ) {
      *reinterpret_cast<floatx4*>(st + (tid >> 3) * kWidePitch + 4 * (tid & 7)) = rn;
#pragma unroll
      for (int it = 0; it < kWItems; ++it)
        *reinterpret_cast<floatx4*>(st + kNs + ((tid >> 3) + 64 * it) * kWidePitch + 4 * (tid & 7)) = rw[it];
    } else {
      *reinterpret_cast<floatx4*>(st + (tid >> 4) * kWideNP + 4 * (tid & 15)) = rn;
#pragma unroll
      for (int it = 0; it < kWLin; ++it)
        *reinterpret_cast<floatx4*>(st + kNs + 4 * (tid + NT * it)) = rw[it];
    }
  };

  if (nkt > 0) {
    fetch();
    commit(wsm);
    if (nkt > 1) fetch();
  }
  __syncthreads();

  // ---- the K loop of one wavefront half: CNT wide tiles from tile wt0.  With an odd NWT the halves
  // differ by a tile: each gets its own copy of the loop (branched to ONCE -- a branch around the odd
  // tile inside the loop cut every step into blocks that the loads and LDS stores of the next tiles
  // could not be spread over; the workgroup's barriers count arrivals, not program counters) --------
  auto run = [&](auto cnt_c) {
    constexpr int CNT = decltype(cnt_c)::value;
    const int wt0 = wh * H0;
    floatx4 acc[CNT];
#pragma unroll
    for (int j = 0; j < CNT; ++j) acc[j] = floatx4{0.f, 0.f, 0.f, 0.f};
    // MFMAs of one half (16 k) of a staged K step: lane (c16, g) takes k = 16 kb + 4 g .. + 3 of row
    // c16 of its narrow tile and of each of its wide tiles, one k per MFMA.  forward: D[i = head
    // column][j = minibatch row]; gradient: D[i = feature column][j = head row] -- either way a lane
    // ends up with four ADJACENT outputs along the contiguous axis of the result.
    auto compute = [&](const float* __restrict__ st, int kb) {
      if constexpr (KMAJ) {
        // k-major images: lane (c16, g) of MFMA u takes k row 16 kb + 4 u + g.  Every operand of the
        // 16 k is read first: the groups below pin "reads, then MFMAs" (left to itself the scheduler
        // sinks each ds_read_b32 next to its MFMA and waits for every one of them)
        float nf[4], wf[4][CNT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kr = kb * 16 + 4 * u + g;
          nf[u] = st[kr * kWideNP + nt * 16 + c16];
          const float* wrow = st + kNs + kr * WIDE + wt0 * 16 + c16;
#pragma unroll
          for (int j = 0; j < CNT; ++j) wf[u][j] = wrow[j * 16];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < CNT; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(nf[u], wf[u][j], acc[j], 0, 0, 0);
        return;
      }
      const float4 nv = *reinterpret_cast<const float4*>(st + (nt * 16 + c16) * kWidePitch + kb * 16 + 4 * g);
      const float nf[4] = {nv.x, nv.y, nv.z, nv.w};
      const float* wrow = st + kNs + (wt0 * 16 + c16) * kWidePitch + kb * 16 + 4 * g;
      float4 wv[CNT];
#pragma unroll
      for (int j = 0; j < CNT; ++j) wv[j] = *reinterpret_cast<const float4*>(wrow + j * 16 * kWidePitch);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
          const float wf = u == 0 ? wv[j].x : (u == 1 ? wv[j].y : (u == 2 ? wv[j].z : wv[j].w));
          if constexpr (!KMAJ) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf, nf[u], acc[j], 0, 0, 0);
          else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(nf[u], wf, acc[j], 0, 0, 0);
        }
    };
    auto k_step = [&](int kt, auto par) {
      constexpr int P = decltype(par)::value;
      float* cur = wsm + P * kStage;
      float* nxt = wsm + (1 - P) * kStage;
      compute(cur, 0);
      if (kt + 1 < nkt) {
        commit(nxt);
        if (kt + 2 < nkt) fetch();
      }
      compute(cur, 1);
      __syncthreads();
    };
    // steady state: both steps of a pair commit and fetch unconditionally -- one basic block per step
    auto k_step_full = [&](auto par) {
      constexpr int P = decltype(par)::value;
      float* cur = wsm + P * kStage;
      float* nxt = wsm + (1 - P) * kStage;
      compute(cur, 0);
      commit(nxt);
      fetch();
      compute(cur, 1);
      // The order of the step, pinned (sched_group_barrier: MFMA 0x008, DS read 0x100, DS write 0x200,
      // VMEM read 0x020).  Left to itself the scheduler put the six LDS stores of the next tile, each
      // behind its own wait, right after the first MFMA (and, k-major, every ds_read_b32 next to its
      // MFMA): the matrix pipe idled through them at the top of every step.
      //   first 16 k : operands | MFMAs with one LDS store of the next tile after every 4th |
      //                the rest of the MFMAs with the second 16 k's operand reads between them
      //   second 16 k: MFMAs with one global load of the tile after next after every 4th | the rest
      constexpr int NR = KMAJ ? 4 * (CNT + 1) : CNT + 1;   // (upper bound on the LDS reads of 16 k)
      constexpr int NM = 4 * CNT, NW = 1 + kWRegs, NL = NW + (KMAJ ? 1 : 0);
      constexpr int REST0 = NM - 4 * NW;
      static_assert(REST0 > 0, "fewer MFMAs than LDS stores to spread");
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
      for (int i = 0; i < NW; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < REST0; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + REST0 - 1) / REST0, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
#pragma unroll
      for (int i = 0; i < NL; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
      __syncthreads();
    };
    int kt = 0;
    for (; kt + 3 < nkt; kt += 2) {
      k_step_full(std::integral_constant<int, 0>{});
      k_step_full(std::integral_constant<int, 1>{});
    }
    for (; kt < nkt; kt += 2) {
      k_step(kt, std::integral_constant<int, 0>{});
      if (kt + 1 < nkt) k_step(kt + 1, std::integral_constant<int, 1>{});
    }

    // ---- slab of this K slice --------------------------------------------------------------
    float* out = p.out + (int64_t)by * p.slab;
    if constexpr (!KMAJ) {
      const int b = n0 + nt * 16 + c16;
      if (p.tickets) {
        // ---- in-launch combine of the K slices (see WideParams) ----
        const int nsp = (int)gridDim.y;
        const int boff = (int)((int64_t)min(b, p.n_narrow - 1) * p.ld_out) + wt0 * 16 + 4 * g;   // < 2^31 floats: the launcher checks
        if (nsp > 1) {
          const __amdgpu_buffer_rsrc_t rs = xwg_buffer(out);
          if (b < p.n_narrow) {
#pragma unroll
            for (int j = 0; j < CNT; ++j) xwg_store4(rs, boff + 16 * j, acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();           // (also: nobody reads an LDS image any more)
        if (tid == 0) {
          int old = nsp - 1;
          if (nsp > 1) {
            old = __hip_atomic_fetch_add(p.tickets + bx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == nsp - 1) __hip_atomic_store(p.tickets + bx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          reinterpret_cast<int*>(wsm)[0] = old;
        }
        __syncthreads();
        const bool last = reinterpret_cast<const int*>(wsm)[0] == nsp - 1;
        float esum = 0.f;
        if (last && b < p.n_narrow) {
          floatx4 v[CNT];
#pragma unroll
          for (int j = 0; j < CNT; ++j) v[j] = floatx4{0.f, 0.f, 0.f, 0.f};
          for (int z = 0; z < nsp; ++z) {        // slice order, whoever arrives last
            if (z == by) {
#pragma unroll
              for (int j = 0; j < CNT; ++j) v[j] += acc[j];
            } else {
              const __amdgpu_buffer_rsrc_t rz = xwg_buffer(p.out + (int64_t)z * p.slab);
              f32x4 q[CNT];
#pragma unroll
              for (int j = 0; j < CNT; ++j) q[j] = xwg_load4(rz, boff + 16 * j);
#pragma unroll
              for (int j = 0; j < CNT; ++j) v[j] += q[j];
            }
          }
          float* dst = p.final_out + (int64_t)b * p.ld_final + wt0 * 16 + 4 * g;
#pragma unroll
          for (int j = 0; j < CNT; ++j) {
            const int col = (wt0 + j) * 16 + 4 * g;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float o = v[j][c] + (col + c < p.n_wide ? p.bias[col + c] : 0.f);
              v[j][c] = o;
              if (p.expsum && col + c >= p.expsum_col0 && col + c < p.expsum_col0 + p.expsum_ncols) esum += expf(o);
            }
            *reinterpret_cast<floatx4*>(dst + 16 * j) = v[j];
          }
        }
        if (p.expsum && reinterpret_cast<const int*>(wsm)[0] == nsp - 1) {
          const float s = block_sum(esum, wsm + 16);   // fixed order: bitwise reproducible
          if (tid == 0) p.expsum[bx] = s;
        }
      } else if (b < p.n_narrow) {
#pragma unroll
        for (int j = 0; j < CNT; ++j)
          *reinterpret_cast<floatx4*>(out + (int64_t)b * p.ld_out + (wt0 + j) * 16 + 4 * g) = acc[j];
      }
    } else {
      const int f = n0 + nt * 16 + 4 * g;
#pragma unroll
      for (int j = 0; j < CNT; ++j) {
        const int n = (wt0 + j) * 16 + c16;
        if (n < p.n_wide) *reinterpret_cast<floatx4*>(out + (int64_t)n * p.ld_out + f) = acc[j];
      }
    }
  };
  if constexpr (H0 == H1) {
    run(std::integral_constant<int, H0>{});
  } else {
    if (wh == 0) run(std::integral_constant<int, H0>{});
    else run(std::integral_constant<int, H1>{});
  }
#endif
}

template <int NWT>
constexpr size_t wide_lds_bytes(bool kmaj) {
  return 2 * sizeof(float) * (kmaj ? (size_t)BK * kWideNP + (size_t)4 * 512 * ((NWT * 16 * BK / 4 + 511) / 512)
                                   : (size_t)64 * kWidePitch + (size_t)64 * ((NWT * 16 + 63) / 64) * kWidePitch);
}

// Launchers (gemm_wide.hip).  BSIG_EUNSUPPORTED: shape not covered (use the generic kernels).
// forward: out[z][b][NWT*16]; gradient: out[z][n][F].
int gemm_wide_forward(const WideParams& p, hipStream_t st);
int gemm_wide_gradient(const WideParams& p, hipStream_t st);
// the head widths the wide kernels are instantiated for
bool gemm_wide_covers(int n_wide);

}  // namespace bsig
