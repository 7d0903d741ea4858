// The fp32 MFMA GEMM with a LEAN main loop (round 5).  Same tiles, same LDS images, same MFMA
// sequence as gemm_mfma_kernel (gemm_kernel.h) -- every output element sees the same fma chain in
// the same k order, so the two kernels are bit-identical for the same K split -- but the K loop
// carries (almost) no vector-ALU instructions:
//   * v_mfma_f32_32x32x2_f32 runs at the rate of the fp32 vector pipe and on its issue port
//     (kernel 4): every VALU instruction of the loop is MFMA time lost.  The old
//     loop spent ~55 VALU instructions per 64 MFMAs on addresses (row * ld + k as 64-bit
//     arithmetic per item per step) and on zeroing a contraction tail that only the LAST step
//     can have -- and the zeroing made every step WAIT for the loads it had just issued.
//   * here every item keeps ONE pointer that advances by a constant per step; only the last,
//     partial K step goes through a guarded fetch; the loads of step t+2 are issued in the middle
//     of step t's MFMAs and committed to the other LDS image in the middle of step t+1's: one
//     barrier per step, a whole step of MFMAs between a load and its first use.
// 16-byte operand items only (rows 16-byte aligned); anything else stays on gemm_mfma_kernel.
#pragma once
#include "gemm_kernel.h"

#include <type_traits>

namespace bsig {

template <int ROWS, bool KMAJOR, int NT>
struct LeanLoader {
  static constexpr int kItems = ROWS * BK / (NT * 4);
  static_assert(ROWS * BK % (NT * 4) == 0, "tile not divisible");
  // items per LDS row (k-contiguous: quads of a row's BK columns; k-major: quads of a k row)
  static constexpr int kPer = KMAJOR ? ROWS / 4 : BK / 4;
  static constexpr bool kSame = NT % kPer == 0;   // every item of a thread has the same column
  struct Regs { float4 r[kItems]; };

  const float* ptr[kItems];   // address of the item in the next tile to fetch (gathered k-major:
                              // of its column in row 0)
  int64_t step;               // floats from one tile to the next
  int64_t ld;
  // k-major operand whose contraction rows are gathered (dW = dO^T X[ids]): the looked-up rows of
  // the next tile to fetch, one step ahead of the loads that use them
  const int32_t* ip;          // &idx[k of item 0 in the tile AFTER the next one]
  int nidx[kItems];
  int kleft;                  // k rows from item 0's next looked-up row to the last valid one
  bool gathered;

  // k0: first contraction index of this workgroup; row0 / nrows: the tile's rows in the other
  // dimension; row_off: device-resolved offset of the gathered dimension
  __device__ inline void init(const float* __restrict__ g, int64_t ld_, const int32_t* __restrict__ idx,
                              int row0, int nrows, int k0, int ktot, int tid, int64_t row_off) {
    ld = ld_;
    gathered = false;
    if constexpr (!KMAJOR) {
      step = BK;
      const int q = tid % kPer;
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const int64_t gr = min(row0 + (tid + it * NT) / kPer, nrows - 1) + row_off;
        const int64_t srow = idx ? (int64_t)idx[gr] : gr;
        ptr[it] = g + srow * ld + k0 + q * 4;
      }
    } else {
      step = (int64_t)BK * ld;
      gathered = idx != nullptr;
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const int item = tid + it * NT;
        const int col = min(row0 + (item % kPer) * 4, (int)ld - 4);
        const int kr = item / kPer;
        if (gathered) {
          ptr[it] = g + col;
          nidx[it] = idx[min(k0 + kr, ktot - 1) + row_off];
        } else {
          ptr[it] = g + (int64_t)(k0 + kr + row_off) * ld + col;
        }
      }
      if (gathered) {
        ip = idx + row_off + k0 + BK + tid / kPer;
        kleft = ktot - 1 - (k0 + BK + tid / kPer);
      }
    }
  }

  // look the NEXT tile's rows up (gathered k-major), clamped to the last valid contraction row
  __device__ inline void lookup_next() {
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int d = it * (NT / kPer);   // (gathered operands: kSame tiles only)
      nidx[it] = ip[min(d, kleft)];
    }
    ip += BK;
    kleft -= BK;
  }

  // a full K step: no guards
  __device__ inline void fetch(Regs& t) {
    if constexpr (KMAJOR) {
      if (gathered) {
#pragma unroll
        for (int it = 0; it < kItems; ++it)
          t.r[it] = *reinterpret_cast<const float4*>(ptr[it] + (int64_t)nidx[it] * ld);
        lookup_next();
        return;
      }
    }
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      t.r[it] = *reinterpret_cast<const float4*>(ptr[it]);
      ptr[it] += step;
    }
  }

  // the last, partial K step: `kv` (1..BK-1) valid contraction indices; `krem` = ld - (first k of
  // the tile) for a k-contiguous operand (how far a row's pitch reaches)
  __device__ inline void fetch_tail(Regs& t, int kv, int64_t krem, int tid) {
    if constexpr (!KMAJOR) {
      const int kq = (tid % kPer) * 4;
      const int back = max(0, kq - (int)min<int64_t>(krem - 4, BK));   // stay inside the row pitch
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        float4 q = *reinterpret_cast<const float4*>(ptr[it] - back);
        if (back != 0) q = make_float4(0.f, 0.f, 0.f, 0.f);   // (then kq >= kv: all four beyond K)
        if (kq + 0 >= kv) q.x = 0.f;
        if (kq + 1 >= kv) q.y = 0.f;
        if (kq + 2 >= kv) q.z = 0.f;
        if (kq + 3 >= kv) q.w = 0.f;
        t.r[it] = q;
      }
    } else {
#pragma unroll
      for (int it = 0; it < kItems; ++it) {
        const int kr = (tid + it * NT) / kPer;
        const bool dead = kr >= kv;
        float4 q;
        if (gathered) q = *reinterpret_cast<const float4*>(ptr[it] + (int64_t)nidx[it] * ld);
        else q = *reinterpret_cast<const float4*>(ptr[it] - (dead ? (int64_t)(kr - (kv - 1)) * ld : 0));
        if (dead) q = make_float4(0.f, 0.f, 0.f, 0.f);
        t.r[it] = q;
      }
    }
  }

  __device__ inline void commit(const Regs& t, float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int it = 0; it < kItems; ++it) {
      const int item = tid + it * NT;
      int off;
      if constexpr (!KMAJOR) off = (item / kPer) * BKP + (item % kPer) * 4;
      else off = item * 4;
      *reinterpret_cast<float4*>(lds + off) = t.r[it];
    }
  }
};

template <int WM, int WN, int TM, int TN, bool AKM, bool BKM>
__global__ __launch_bounds__(WM * WN * 64) void gemm_lean_kernel(GemmParams p) {
#ifndef BSIG_HOST_SAN_BUILD
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int kAs = lds_floats<BM, AKM>(), kBs = lds_floats<BN, BKM>();
  constexpr int kStage = kAs + kBs;
  constexpr int kEpi = WM * WN * 32 * 32;
  __shared__ __attribute__((aligned(16))) float smem[2 * kStage > kEpi ? 2 * kStage : kEpi];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;
  const int l31 = lane & 31, h = lane >> 5;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_swz) {   // as gemm_mfma_kernel: an XCD works on a contiguous run of tiles
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * (int)gridDim.z;
    const int lin = bx + gx * (by + gy * bz);
    const int xcd = lin & 7, q = nwg >> 3, r = nwg & 7;
    const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);
    bx = wgid % gx; by = (wgid / gx) % gy; bz = wgid / (gx * gy);
  }
  p.bid_z = bz;
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * p.k_chunk;
  const int kend = min(p.k, kbeg + p.k_chunk);
  const int nkt = (kend - kbeg + BK - 1) / BK;
  const int ktail = (kend - kbeg) % BK;   // != 0: the last step is partial

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  float adam_ss = 0.f, adam_ib = 0.f;
  if (p.epilogue == EPI_ADAM) { adam_ss = p.adam_dyn[0]; adam_ib = p.adam_dyn[1]; }
  const int64_t dstep = p.dyn ? (int64_t)(p.dyn[0] + p.dyn_delta) : 0;
  LeanLoader<BM, AKM, NT> la;
  LeanLoader<BN, BKM, NT> lb;
  la.init(p.a, p.lda, p.a_rows, m0, p.m, kbeg, p.k, tid, dstep * p.a_dyn_stride + p.a_dyn_base);
  lb.init(p.b, p.ldb, p.b_rows, n0, p.n, kbeg, p.k, tid, dstep * p.b_dyn_stride + p.b_dyn_base);
  typename LeanLoader<BM, AKM, NT>::Regs ra;
  typename LeanLoader<BN, BKM, NT>::Regs rb;

  auto fetch_tile = [&](int kt) {   // block-uniform branch: only the last step can be partial
    if (ktail != 0 && kt == nkt - 1) {
      la.fetch_tail(ra, ktail, p.lda - (kbeg + kt * BK), tid);
      lb.fetch_tail(rb, ktail, p.ldb - (kbeg + kt * BK), tid);
    } else {
      la.fetch(ra);
      lb.fetch(rb);
    }
  };
  // MFMAs of k groups [kg0, kg1) of one staged K tile (a group: 8 k, TM*TN*4 MFMAs)
  auto compute = [&](const float* __restrict__ At, const float* __restrict__ Bt, int kg0, int kg1) {
#pragma unroll
    for (int kg = kg0; kg < kg1; ++kg) {
      float af[TM][4], bf[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = (wm * TM + i) * 32 + l31;
        if constexpr (!AKM) {
          const float4 q = *reinterpret_cast<const float4*>(&At[row * BKP + kg * 8 + h * 4]);
          af[i][0] = q.x; af[i][1] = q.y; af[i][2] = q.z; af[i][3] = q.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) af[i][u] = At[(kg * 8 + h * 4 + u) * BM + row];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = (wn * TN + j) * 32 + l31;
        if constexpr (!BKM) {
          const float4 q = *reinterpret_cast<const float4*>(&Bt[col * BKP + kg * 8 + h * 4]);
          bf[j][0] = q.x; bf[j][1] = q.y; bf[j][2] = q.z; bf[j][3] = q.w;
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) bf[j][u] = Bt[(kg * 8 + h * 4 + u) * BN + col];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][u], bf[j][u], acc[i][j],
                                                             0, 0, 0);
    }
  };

  if (nkt > 0) {
    fetch_tile(0);
    la.commit(ra, smem, tid);
    lb.commit(rb, smem + kAs, tid);
    if (nkt > 1) fetch_tile(1);
  }
  __syncthreads();
  // one K step on LDS image PAR (compile-time offsets), the next tile committed to the other one
  auto k_step = [&](int kt, auto par) {
    constexpr int P = decltype(par)::value;
    float* cur = smem + P * kStage;
    float* nxt = smem + (1 - P) * kStage;
    compute(cur, cur + kAs, 0, BK / 16);
    if (kt + 1 < nkt) {
      la.commit(ra, nxt, tid);
      lb.commit(rb, nxt + kAs, tid);
      if (kt + 2 < nkt) fetch_tile(kt + 2);
    }
    compute(cur, cur + kAs, BK / 16, BK / 8);
    __syncthreads();
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    k_step(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < nkt) k_step(kt + 1, std::integral_constant<int, 1>{});
  }

  float* patch = smem + wid * (32 * 32);
  float exp_acc = 0.f;
  const bool vec_epi = epilogue_vec_ok(p);
#define BSIG_TILE_EPILOGUE(I, J)                                                          \
  if constexpr ((I) < TM && (J) < TN) {                                                   \
    _Pragma("unroll") for (int q = 0; q < 16; ++q)                                        \
        patch[((q & 3) + 8 * (q >> 2) + 4 * h) * 32 + l31] = acc[I][J][q];                \
    __builtin_amdgcn_wave_barrier();                                                      \
    run_tile_epilogue(p, patch, m0 + (wm * TM + (I)) * 32, n0 + (wn * TN + (J)) * 32,     \
                      lane, vec_epi, exp_acc, adam_ss, adam_ib);                          \
    __builtin_amdgcn_wave_barrier();                                                      \
  }
  BSIG_TILE_EPILOGUE(0, 0) BSIG_TILE_EPILOGUE(0, 1) BSIG_TILE_EPILOGUE(0, 2)
  BSIG_TILE_EPILOGUE(1, 0) BSIG_TILE_EPILOGUE(1, 1) BSIG_TILE_EPILOGUE(1, 2)
  BSIG_TILE_EPILOGUE(2, 0) BSIG_TILE_EPILOGUE(2, 1) BSIG_TILE_EPILOGUE(2, 2)
#undef BSIG_TILE_EPILOGUE
  static_assert(TM <= 3 && TN <= 3, "extend the tile enumeration");
  if (p.expsum && p.splits == 1) {
    const float s = block_sum(exp_acc, smem);
    if (tid == 0) p.expsum[by * gridDim.x + bx] = s;
  }
#endif
}

template <int WM, int WN, int TM, int TN>
inline int launch_lean(const GemmParams& p, bool akm, bool bkm, hipStream_t st) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const dim3 grid(ceil_div(p.n, BN), ceil_div(p.m, BM), p.splits);
  const dim3 block(WM * WN * 64);
  // a gathered k-major operand needs one column per thread (LeanLoader::lookup_next)
  if (akm && p.a_rows && !LeanLoader<BM, true, WM * WN * 64>::kSame) return BSIG_EUNSUPPORTED;
  if (bkm && p.b_rows && !LeanLoader<BN, true, WM * WN * 64>::kSame) return BSIG_EUNSUPPORTED;
#define BSIG_LEAN_CASE(AK, BKm)                                                              \
  if (akm == AK && bkm == BKm) {                                                             \
    hipLaunchKernelGGL((gemm_lean_kernel<WM, WN, TM, TN, AK, BKm>), grid, block, 0, st, p);  \
    return BSIG_OK;                                                                          \
  }
  BSIG_LEAN_CASE(false, false) BSIG_LEAN_CASE(false, true)
  BSIG_LEAN_CASE(true, false) BSIG_LEAN_CASE(true, true)
#undef BSIG_LEAN_CASE
  return BSIG_EUNSUPPORTED;
}

// one per tile shape (gemm_lean_*.hip); BSIG_EUNSUPPORTED: not covered, use gemm_mfma_kernel
int launch_lean_64(const GemmParams& p, bool akm, bool bkm, hipStream_t st);
int launch_lean_128(const GemmParams& p, bool akm, bool bkm, hipStream_t st);
int launch_lean_128x32(const GemmParams& p, bool akm, bool bkm, hipStream_t st);
int launch_lean_128x64(const GemmParams& p, bool akm, bool bkm, hipStream_t st);
int launch_lean_128x96(const GemmParams& p, bool akm, bool bkm, hipStream_t st);
int launch_lean_96x128(const GemmParams& p, bool akm, bool bkm, hipStream_t st);

}  // namespace bsig
