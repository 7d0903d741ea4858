// Device-side pieces of the mixture-density head shared by mdn_head.hip and the
// persistent update kernel (fit_persistent.hip).
#pragma once
#include "head.h"

namespace bsig {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr int kSigMax = 4096;       // capacity of the exp(pre) partial-sum array
constexpr int kSigBlocks = 512;     // blocks of the stand-alone sigma0_sum kernel
constexpr int kElemsPerLane = 8;    // wave kernel: register-cached elements per lane
constexpr int kXwgMax = 256;        // persistent kernel: at most this many workgroups
// Cross-workgroup flags / granules sit one per 64-byte block: a fan-in of 144 flags polled by
// 100 workgroups costs 2.75 us when the flags are consecutive dwords (every poller hammers the
// same few lines of one memory channel) and 1.4 us at this spacing
// (tools/micro/fanin_bench.hip, profiles/r02_fanin_bench.txt).
#ifndef BSIG_XWG_STRIDE_BYTES
#define BSIG_XWG_STRIDE_BYTES 64
#endif
constexpr int kFlagStride = BSIG_XWG_STRIDE_BYTES / 4;   // dwords between two flags
constexpr int kGranStride = BSIG_XWG_STRIDE_BYTES / 8;   // 8-byte granules between two granules
constexpr int kFlagArr = kXwgMax * kFlagStride;       // dwords of one flag array
constexpr int kGranArr = kXwgMax * kGranStride;       // granule slots of one granule array
// The per-update loss granules alternate between two arrays by update parity (arrays 2 and 5 of a
// plan's six: Sigma exp | Sigma u dL/dsigma | loss (even updates) | evaluation x 2 | loss (odd)):
// owner 0 gathers them by exact tag AFTER it has released the tile workgroups, so with one array a
// delayed owner 0 could find a slot already carrying the next update's tag; the slot of update t is
// now rewritten by update t + 2, which cannot start before owner 0 has published update t + 1.
__device__ __forceinline__ unsigned long long* loss_granules(unsigned long long* gran, unsigned epoch) {
  return gran + ((epoch & 1u) ? 5 : 2) * kGranArr;
}

struct HeadArgs {
  const float* seg_w; int64_t ld_w;    // logits (fused) or weights (tuple), K / row
  const float* seg_mu; int64_t ld_mu;  // D*K / row
  const float* seg_sg; int64_t ld_sg;  // pre-activation (fused) or L_d (tuple)
  const float* seg_lo; int64_t ld_lo;  // Ls*K / row or nullptr
  const float* y; int64_t ldy; const int32_t* y_rows;
  const int32_t* y_dyn; int64_t y_dyn_stride;  // y_rows index offset = y_dyn[0]*stride (replay)
  int batch; float inv_norm;
  int D, K, Ls, Nh, R;
  int from_tuple;
  const float* noise; uint64_t seed, stream_id;
  const uint64_t* dyn_rng;             // device {seed, stream}: overrides (graph replay)
  float eps_noise, min_w, ll_limit;
  const float* sig_partials; int n_sig;  // partial sums of exp(pre) over the minibatch
  float* d_out; int64_t ld_dout;       // nullptr: forward only
  float* block_lse;                    // [gridDim.x]
  float* block_uds;                    // [gridDim.x]  sum u * dL/dsigma
  int32_t* nonfinite;
};

// ---- diagonal covariance, one wavefront per row ------------------------------
// A wave never depends on another wave's LDS data here (each row is private),
// so the phases are ordered by the in-order LDS pipeline of the wave itself
// (wave_barrier only pins the compiler).  Per-component sums over the dimensions
// are strided wavefront shuffles; the jitter draws use all four Philox outputs.
//
// diag_row: forward + backward of one row by one wavefront.  In: tile[Nh] raw
// head outputs, yv[D] target.  Out: gradients in tile[K..Nh) and dlg[K]
// (without the jitter-scale term), the row's logsumexp, sum(u * dL/dsigma),
// and exp(pre) of the lane's elements (for the jitter-scale correction).
struct RowOut {
  float lse, uds; bool bad; float esg0[kElemsPerLane];
#ifdef BSIG_ROW_PROF
  long long ts[10];            // (profiling build) wall-clock stamps inside diag_row
#endif
};
#ifdef BSIG_ROW_PROF
#define BSIG_ROW_STAMP(i) out.ts[i] = wall_clock64()
#else
#define BSIG_ROW_STAMP(i)
#endif

// `eps_fn()` delivers the jitter scale; it is called (by the whole wave) after
// everything that does not depend on it -- the Philox draws, exp(pre), the
// mixture weights -- so that a caller can hide a cross-workgroup wait there.
// The row's jitter draws u ~ U[0, 1) (mdnn.py:116), element q of the lane as in diag_row: they
// depend on (seed, stream, row) only, so a caller that waits for the row's head outputs anyway can
// draw them during the wait and hand them to diag_row (`eu_pre`).
__device__ __forceinline__ void diag_row_noise(const HeadArgs& a, int groups, int k, int d0, int row,
                                               bool active, int lane, float (&eu)[kElemsPerLane]) {
  const int D = a.D, K = a.K;
  const int TPR = groups * K;
  const bool elem = active && lane < TPR;
  const bool jitter = !a.from_tuple && a.eps_noise != 0.f;
  const bool draw = jitter && a.noise == nullptr;
  const uint64_t rng_seed = a.dyn_rng ? a.dyn_rng[0] : a.seed;
  const uint64_t rng_sid = a.dyn_rng ? a.dyn_rng[1] : a.stream_id;
  Philox4 ph{{0u, 0u, 0u, 0u}};
#pragma unroll
  for (int q = 0; q < kElemsPerLane; ++q) {
    const int d = d0 + q * groups;
    eu[q] = 0.f;
    if ((q & 3) == 0 && draw && elem && d < D)
      ph = philox4x32_10(rng_seed, rng_sid, ((uint64_t)row * 64 + lane) * 2 + (q >> 2));
    if (elem && d < D && jitter) eu[q] = a.noise ? a.noise[((int64_t)row * D + d) * K + k] : u01(ph.v[q & 3]);
  }
}

#ifdef BSIG_DIAG_ROW_V1
template <typename EpsFn>
__device__ __forceinline__ void diag_row(const HeadArgs& a, int row, bool active, int lane,
                                         float* tile, const float* yv, float* rk, float* lpk,
                                         float* dlg, EpsFn&& eps_fn, RowOut& out,
                                         const float* eu_pre = nullptr) {
  const int D = a.D, K = a.K;
  const int DK = D * K;
  const int groups = 64 / K;               // d-slots per sweep
  const int TPR = groups * K;
  const int k = lane % K, d0 = lane / K;
  const bool elem = active && lane < TPR;
  bool bad = false;
  // per-element values kept for the backward half
  float ez[kElemsPerLane], esg[kElemsPerLane], eu[kElemsPerLane];
  float (&esg0)[kElemsPerLane] = out.esg0;
  float quad = 0.f, logdet = 0.f;
  Philox4 ph{{0u, 0u, 0u, 0u}};
  const bool jitter = !a.from_tuple && a.eps_noise != 0.f;
  const bool draw = jitter && a.noise == nullptr;
  const uint64_t rng_seed = a.dyn_rng ? a.dyn_rng[0] : a.seed;
  const uint64_t rng_sid = a.dyn_rng ? a.dyn_rng[1] : a.stream_id;
#pragma unroll
  for (int q = 0; q < kElemsPerLane; ++q) {
    const int d = d0 + q * groups;
    ez[q] = 0.f; esg[q] = 1.f; esg0[q] = 1.f; eu[q] = 0.f;
    if (eu_pre == nullptr && (q & 3) == 0 && draw && elem && d < D)
      ph = philox4x32_10(rng_seed, rng_sid, ((uint64_t)row * 64 + lane) * 2 + (q >> 2));
    if (elem && d < D && !a.from_tuple) {
      esg0[q] = expf(tile[K + DK + d * K + k]);
      if (jitter) eu[q] = eu_pre ? eu_pre[q] : (a.noise ? a.noise[((int64_t)row * D + d) * K + k] : u01(ph.v[q & 3]));
    }
  }
  // mixture weights (mdnn.py:109-111): lane j < K owns component j, the sums over
  // the components are wavefront reductions
  const bool comp = active && lane < K;
  float s_own = 0.f, w_own = 0.f, csum = 1.f;
  if (active && !a.from_tuple) {
    float mx = tile[0];
    for (int j = 1; j < K; ++j) mx = fmaxf(mx, tile[j]);
    const float e_own = comp ? expf(tile[lane] - mx) : 0.f;
    s_own = e_own / wave_sum_dpp(e_own);
    const float c_own = comp ? fminf(fmaxf(s_own, a.min_w), 1.0f) : 0.f;
    csum = wave_sum_dpp(c_own);
    w_own = c_own / csum;
  } else if (comp) {
    w_own = tile[lane];
  }
  const float eps = eps_fn();
#pragma unroll
  for (int q = 0; q < kElemsPerLane; ++q) {
    const int d = d0 + q * groups;
    if (elem && d < D) {
      const float mu = tile[K + d * K + k];
      float sg;
      if (a.from_tuple) sg = tile[K + DK + d * K + k];
      else {
        sg = esg0[q];
        if (eps != 0.f) sg += eu[q] * eps;
        else eu[q] = 0.f;
      }
      bad |= !(isfinite(mu) && isfinite(sg));
      const float z = (yv[d] - mu) / sg;
      quad += z * z;
      logdet += logf(sg);
      ez[q] = z; esg[q] = sg;
    }
  }
  // sum over the dimensions of each component: lanes k, k+K, k+2K, ...
  for (int off = 32; off >= 1; off >>= 1) {
    if (off < groups || off == 1) {
      const float tq = __shfl_down(quad, off * K, 64);
      const float tl = __shfl_down(logdet, off * K, 64);
      if (d0 + off < groups && lane + off * K < 64) { quad += tq; logdet += tl; }
    }
  }
  if (comp) {
    const float logp = -0.5f * quad - logdet - (float)D * kHalfLog2Pi;
    const float w = w_own;
    const float lp = fminf(fmaxf(logp, -a.ll_limit), a.ll_limit);
    const float rv = lp + logf(fminf(fmaxf(w, a.min_w), 1.0f));
    bad |= !(isfinite(w) && isfinite(logp) && isfinite(rv));
    rk[lane] = rv;
    lpk[lane] = logp;
  }
  __builtin_amdgcn_wave_barrier();
  float lse = 0.f;
  if (active) {
    float m2 = rk[0];
    for (int j = 1; j < K; ++j) m2 = fmaxf(m2, rk[j]);
    float se = 0.f;
    for (int j = 0; j < K; ++j) se += expf(rk[j] - m2);
    lse = m2 + logf(se);
  }

  const bool bwd = a.d_out != nullptr;
  float uds = 0.f;
  if (bwd && elem) {
    const float sc = -expf(rk[k] - lse) * a.inv_norm;
    const float lp0 = lpk[k];
    const float g_lp = (lp0 >= -a.ll_limit && lp0 <= a.ll_limit) ? sc : 0.f;
#pragma unroll
    for (int q = 0; q < kElemsPerLane; ++q) {
      const int d = d0 + q * groups;
      if (d < D) {
        const float dsg = g_lp * (ez[q] * ez[q] - 1.0f) / esg[q];
        uds += eu[q] * dsg;
        tile[K + d * K + k] = g_lp * ez[q] / esg[q];
        tile[K + DK + d * K + k] = dsg * esg0[q];
      }
    }
  }
  if (bwd && active) {                      // mixture-weight path, lane = component
    float gw = 0.f;
    if (comp) {
      const float sc = -expf(rk[lane] - lse) * a.inv_norm;
      gw = (w_own >= a.min_w && w_own <= 1.0f) ? sc / fminf(fmaxf(w_own, a.min_w), 1.0f) : 0.f;
    }
    float dlogit = gw;
    if (!a.from_tuple) {                    // through the renormalisation, the clamp, the softmax
      const float s1 = wave_sum_dpp(gw * w_own);
      const float gs = (comp && s_own >= a.min_w && s_own <= 1.0f) ? (gw - s1) / csum : 0.f;
      const float s2 = wave_sum_dpp(gs * s_own);
      dlogit = s_own * (gs - s2);
    }
    if (comp) dlg[lane] = dlogit;           // separate slot: the logits stay readable
  }
  __builtin_amdgcn_wave_barrier();
  out.lse = lse; out.uds = uds; out.bad = bad;
}

#else
// Wavefront maximum through the DPP row shifts / broadcasts (cf. wave_sum_dpp, common.h); every
// lane gets the result.
__device__ inline float wave_max_dpp(float v) {
#define BSIG_DPP_MAX(ctrl, row_mask)                                                        \
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), \
                                             __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false)))
  BSIG_DPP_MAX(0x111, 0xf);   // row_shr:1
  BSIG_DPP_MAX(0x112, 0xf);   // row_shr:2
  BSIG_DPP_MAX(0x114, 0xf);   // row_shr:4
  BSIG_DPP_MAX(0x118, 0xf);   // row_shr:8
  BSIG_DPP_MAX(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
  BSIG_DPP_MAX(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef BSIG_DPP_MAX
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// wave_sum_dpp(a * b) with the product's first addition fused (the first version's compile of that
// expression; with contraction off it has to be spelled out to keep the logit gradients' bits)
__device__ inline float wave_sum_dpp_prod(float a, float b) {
  const float p = a * b;
  float v = __builtin_fmaf(a, b, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                                      0, __builtin_bit_cast(int, p), 0x111, 0xf, 0xf, false)));   // row_shr:1
#define BSIG_DPP_ADD(ctrl, row_mask)                                                        \
  v += __builtin_bit_cast(                                                                  \
      float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
  BSIG_DPP_ADD(0x112, 0xf);   // row_shr:2
  BSIG_DPP_ADD(0x114, 0xf);   // row_shr:4
  BSIG_DPP_ADD(0x118, 0xf);   // row_shr:8
  BSIG_DPP_ADD(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
  BSIG_DPP_ADD(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef BSIG_DPP_ADD
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// The jitter scale as an object: issue() may start a cross-workgroup gather early (its loads stay
// in flight over the arithmetic that does not depend on the scale), get() delivers the value.
template <typename F>
struct EpsNow {
  F f;
  __device__ __forceinline__ void issue() {}
  __device__ __forceinline__ float get() { return f(); }
};

// Round 4: written for LATENCY.  The row is one wavefront's dependent chain (its 2 x D x K elements
// are 1-8 per lane), and the first version spent it in waits: ~40 LDS round trips, 39 IEEE
// divisions, four rounds of two ds_bpermutes for the per-component sums, K exponentials per lane for
// the logsumexp (4.3 us per row at D = 32, K = 4, the longest piece of every persistent update).  Now:
// every LDS read of the row is issued before anything waits; the lane geometry (integer divisions
// by K) comes from the caller; the sums over the dimensions of a component go through LDS (the
// component lanes read their `groups` partial sums back in one batch, at constant offsets); the
// logsumexp's K exponentials are one per component lane; the element loops are specialised on the
// number of sweeps; the jitter scale is requested before the arithmetic that does not need it.
// The ARITHMETIC stays the reference's: IEEE divisions (a cheaper reciprocal moved cfg3 -- whose
// 11802-wide first layer turns an ulp into 1e-4 of held-out NLL within a chunk -- from 7e-6 to
// 1.7e-4 off the oracle), the first version's reduction trees and summation orders, and no
// contraction of multiplies and adds (below).
// Where a lane sits in the row: component k, dimension slot d0 (element (d, k), d = d0 + q * groups,
// of sweep q); groups = 64 / K, TPR = groups * K lanes take part, nq sweeps cover the D dimensions.
// Four integer divisions by run-time values (~150 instructions): a kernel that runs many rows per
// lane computes this once (row_geom) and calls diag_row_g.
struct RowGeom { int k, d0, groups, TPR, nq; };
__device__ __forceinline__ RowGeom row_geom(int D, int K, int lane) {
  RowGeom g;
  g.groups = 64 / K;
  g.TPR = g.groups * K;
  g.k = lane % K;
  g.d0 = lane / K;
  g.nq = (D + g.groups - 1) / g.groups;
  return g;
}

// NQ: compile-time bound of the sweeps (nq <= NQ): the element loops have no run-time test but the
// validity of the last sweep.
template <int NQ, typename Eps>
__device__ __forceinline__ void diag_row_body(const HeadArgs& a, const RowGeom& rg, int row, bool active,
                                              int lane, float* tile, const float* yv, float* rk,
                                              float* dlg, Eps& eps_src, RowOut& out, const float* eu_pre) {
  // Contraction of multiplies and adds is OFF in this function and the three fused operations it
  // has (sigma0 + u * eps, z^2 - 1, the sum of u * dL/dsigma) are written out: left to the compiler,
  // the bits of a row changed with unrelated edits of the surrounding code -- which the wide
  // cross-correlation configs amplify to 1e-4 of held-out NLL within a chunk.  The choices are the
  // first version's (tools/micro/diag_row_ab.py compares two builds bit by bit).
#pragma clang fp contract(off)
  const int D = a.D, K = a.K;
  const int DK = D * K;
  const int groups = rg.groups, TPR = rg.TPR, k = rg.k, d0 = rg.d0;
  const bool elem = active && lane < TPR;
  const bool comp = active && lane < K;
  const bool tup = a.from_tuple != 0;
  const bool jitter = !tup && a.eps_noise != 0.f;
  const bool draw = jitter && a.noise == nullptr && eu_pre == nullptr;
  bool bad = false;
  float (&esg0)[kElemsPerLane] = out.esg0;
  float pre[NQ], ydf[NQ], eu[NQ], rsg[NQ], ez[NQ], muv[NQ];   // (rsg: the element's sigma)
  bool valid[NQ];

  BSIG_ROW_STAMP(0);
  // ---- every LDS read of the row, then the jitter-scale request ------------------------------
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int e = min(lane + q * TPR, DK - 1), d = min(d0 + q * groups, D - 1);
    valid[q] = elem && d0 + q * groups < D;
    pre[q] = tile[K + DK + e];
    muv[q] = tile[K + e];
    ydf[q] = yv[d];
  }
  const float lg_own = tile[min(lane, K - 1)];
  eps_src.issue();
  BSIG_ROW_STAMP(1);

  // ---- what does not depend on the jitter scale ---------------------------------------------
  Philox4 ph{{0u, 0u, 0u, 0u}};
  const uint64_t rng_seed = a.dyn_rng ? a.dyn_rng[0] : a.seed;
  const uint64_t rng_sid = a.dyn_rng ? a.dyn_rng[1] : a.stream_id;
#pragma unroll
  for (int q = 0; q < kElemsPerLane; ++q) esg0[q] = 1.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    eu[q] = 0.f; rsg[q] = 0.f; ez[q] = 0.f;
    if ((q & 3) == 0 && draw && valid[q])
      ph = philox4x32_10(rng_seed, rng_sid, ((uint64_t)row * 64 + lane) * 2 + (q >> 2));
    ydf[q] -= muv[q];
    if (!tup) {
      const float ev = expf(pre[q]);
      esg0[q] = valid[q] ? ev : 1.f;
      if (jitter && valid[q])
        eu[q] = eu_pre ? eu_pre[q] : (a.noise ? a.noise[((int64_t)row * D + d0 + q * groups) * K + k] : u01(ph.v[q & 3]));
    }
  }
  BSIG_ROW_STAMP(2);
  // mixture weights (mdnn.py:109-111): lane j < K owns component j
  float s_own = 0.f, w_own = 0.f, csum = 1.f;
  if (active && !tup) {
    const float mx = wave_max_dpp(comp ? lg_own : -INFINITY);     // (a loop over tile[j]: K dependent LDS round trips)
    const float e_own = comp ? expf(lg_own - mx) : 0.f;
    s_own = e_own / wave_sum_dpp(e_own);
    const float c_own = comp ? fminf(fmaxf(s_own, a.min_w), 1.0f) : 0.f;
    csum = wave_sum_dpp(c_own);
    w_own = c_own / csum;
  } else if (comp) {
    w_own = lg_own;
  }
  const float wc = fminf(fmaxf(w_own, a.min_w), 1.0f);     // the second clamp, mdnn.py:160
  const float lw = comp ? logf(wc) : 0.f;

  // ---- the row with the jitter scale ----------------------------------------------------------
  BSIG_ROW_STAMP(3);
  const float eps = eps_src.get();
  BSIG_ROW_STAMP(4);
  float quad = 0.f, logdet = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    float sg;
    if (tup) sg = pre[q];
    else {
      sg = esg0[q];
      if (eps != 0.f) sg = __builtin_fmaf(eu[q], eps, sg);
      else eu[q] = 0.f;
    }
    if (!valid[q]) sg = 1.f;
    bad |= valid[q] && !(isfinite(muv[q]) && isfinite(sg));
    const float z = valid[q] ? ydf[q] / sg : 0.f;
    quad += z * z;
    logdet += logf(sg);          // (log 1 = 0 for the sweeps beyond D)
    ez[q] = z; rsg[q] = sg;
  }
  // sums over the dimensions of each component: the lanes' partial sums go where the row's
  // mu / pre values were (all in registers by now), component by component (slot d0 of component k
  // at k * gp + d0), and lane k < K adds its component's gp values in slot order -- reads at constant
  // offsets from one address per lane
  const int gp = min(groups, D);
  BSIG_ROW_STAMP(5);
  __builtin_amdgcn_wave_barrier();
  if (elem && d0 < gp) { tile[K + k * gp + d0] = quad; tile[K + DK + k * gp + d0] = logdet; }
  __builtin_amdgcn_wave_barrier();
  float rv = -INFINITY, logp = 0.f;
  if (comp) {
    float qs = 0.f, ls = 0.f;
    const float* bq = tile + K + lane * gp;
    const float* bl = bq + DK;
    if (groups <= 16) {
      // (slots beyond gp: another component's values or the next segment, still inside the row's LDS; zeroed)
      float tq[16], tl[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { tq[u] = bq[u]; tl[u] = bl[u]; }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (u >= gp) { tq[u] = 0.f; tl[u] = 0.f; }
      // the order of the first version's shuffle tree (slot i takes slot i + off, off = 8, 4, 2, 1 below
      // `groups`, all slots at once): the same bits as before
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        if (off < groups || off == 1) {
#pragma unroll
          for (int i = 0; i + off < 16; ++i)
            if (i + off < groups) { tq[i] += tq[i + off]; tl[i] += tl[i + off]; }
        }
      }
      qs = tq[0]; ls = tl[0];
    } else {
      for (int g0 = 0; g0 < gp; g0 += 16) {
        float tq[16], tl[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { tq[u] = bq[g0 + u]; tl[u] = bl[g0 + u]; }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (g0 + u < gp) { qs += tq[u]; ls += tl[u]; }
      }
    }
    logp = -0.5f * qs - ls - (float)D * kHalfLog2Pi;
    const float lp = fminf(fmaxf(logp, -a.ll_limit), a.ll_limit);
    rv = lp + lw;
    bad |= !(isfinite(w_own) && isfinite(logp) && isfinite(rv));
  }
  BSIG_ROW_STAMP(6);
  // logsumexp over the components (lanes < K), mdnn.py:163-178
  float lse = 0.f, sc = 0.f;
  if (active) {
    const float m2 = wave_max_dpp(rv);
    const float ek = comp ? expf(rv - m2) : 0.f;
    float se = 0.f;                                   // (added in component order, as the first version did)
    for (int j = 0; j < K; ++j) se += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ek), j));
    lse = m2 + logf(se);
    if (comp) sc = -expf(rv - lse) * a.inv_norm;
  }

  BSIG_ROW_STAMP(7);
  const bool bwd = a.d_out != nullptr;
  float uds = 0.f;
  if (bwd && active) {
    // d loss / d logp_k (zero where the clamp of mdnn.py:159 is active) for the element lanes
    __builtin_amdgcn_wave_barrier();
    if (comp) rk[lane] = (logp >= -a.ll_limit && logp <= a.ll_limit) ? sc : 0.f;
    __builtin_amdgcn_wave_barrier();
    const float g_lp = rk[k];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (valid[q]) {
        const int e = lane + q * TPR;
        const float dsg = g_lp * __builtin_fmaf(ez[q], ez[q], -1.0f) / rsg[q];
        uds = __builtin_fmaf(eu[q], dsg, uds);
        tile[K + e] = g_lp * ez[q] / rsg[q];
        tile[K + DK + e] = dsg * esg0[q];
      }
    }
    // mixture-weight path, lane = component
    const float gw = (comp && w_own >= a.min_w && w_own <= 1.0f) ? sc / wc : 0.f;
    float dlogit = gw;
    if (!tup) {                             // through the renormalisation, the clamp, the softmax
      const float s1 = wave_sum_dpp_prod(gw, w_own);
      const float gs = (comp && s_own >= a.min_w && s_own <= 1.0f) ? (gw - s1) / csum : 0.f;
      const float s2 = wave_sum_dpp_prod(gs, s_own);
      dlogit = s_own * (gs - s2);
    }
    if (comp) dlg[lane] = dlogit;           // separate slot: the logits stay readable
  }
  __builtin_amdgcn_wave_barrier();
  BSIG_ROW_STAMP(8);
  out.lse = lse; out.uds = uds; out.bad = bad;
}

// NQCAP: what the caller knows about the sweeps (rg.nq <= NQCAP): a kernel built for NQCAP = 2 carries
// only the two-sweep body and its registers (the wave kernel: 109 -> VGPRs of the <2> body alone).
template <int NQCAP = kElemsPerLane, typename Eps>
__device__ __forceinline__ void diag_row_impl(const HeadArgs& a, const RowGeom& rg, int row, bool active,
                                              int lane, float* tile, const float* yv, float* rk,
                                              float* lpk, float* dlg, Eps& eps_src, RowOut& out,
                                              const float* eu_pre) {
  (void)lpk;
  if constexpr (NQCAP <= 2) {
    diag_row_body<2>(a, rg, row, active, lane, tile, yv, rk, dlg, eps_src, out, eu_pre);
  } else if constexpr (NQCAP <= 4) {
    if (rg.nq <= 2) diag_row_body<2>(a, rg, row, active, lane, tile, yv, rk, dlg, eps_src, out, eu_pre);
    else diag_row_body<4>(a, rg, row, active, lane, tile, yv, rk, dlg, eps_src, out, eu_pre);
  } else {
    if (rg.nq <= 2) diag_row_body<2>(a, rg, row, active, lane, tile, yv, rk, dlg, eps_src, out, eu_pre);
    else if (rg.nq <= 4) diag_row_body<4>(a, rg, row, active, lane, tile, yv, rk, dlg, eps_src, out, eu_pre);
    else diag_row_body<kElemsPerLane>(a, rg, row, active, lane, tile, yv, rk, dlg, eps_src, out, eu_pre);
  }
}

// `eps_fn()` form: the jitter scale from a plain callable; the lane geometry computed here
template <typename EpsFn>
__device__ __forceinline__ void diag_row(const HeadArgs& a, int row, bool active, int lane,
                                         float* tile, const float* yv, float* rk, float* lpk,
                                         float* dlg, EpsFn&& eps_fn, RowOut& out,
                                         const float* eu_pre = nullptr) {
  EpsNow<EpsFn&> e{eps_fn};
  const RowGeom rg = row_geom(a.D, a.K, lane);
  diag_row_impl(a, rg, row, active, lane, tile, yv, rk, lpk, dlg, e, out, eu_pre);
}
// ... for a kernel built for at most NQCAP sweeps
template <int NQCAP, typename EpsFn>
__device__ __forceinline__ void diag_row_capped(const HeadArgs& a, int row, bool active, int lane,
                                                float* tile, const float* yv, float* rk, float* lpk,
                                                float* dlg, EpsFn&& eps_fn, RowOut& out) {
  EpsNow<EpsFn&> e{eps_fn};
  const RowGeom rg = row_geom(a.D, a.K, lane);
  diag_row_impl<NQCAP>(a, rg, row, active, lane, tile, yv, rk, lpk, dlg, e, out, nullptr);
}
#endif

// ---- diagonal covariance: the FAST row of the persistent kernels' owners (round 6) -------------------
// One row on TWO wavefronts (wavefront h in {0, 1} takes the sweeps h, h + 2, ...), lane = d-slot * KP + k
// with the component count padded to KP in {4, 8, 16} (the lanes k >= K of a group idle and enter the
// butterflies with the neutral element): the lanes of a component's dimensions sit at a stride of KP inside
// the 16-lane DPP rows, and KP adjacent lanes hold the components.  Sums over the dimensions: DPP row
// rotations, then ONE exchange through LDS (xq / xl: 8 floats per component -- 4 DPP rows x 2 wavefronts
// --, `bar()` a barrier both wavefronts pass), read back by EVERY lane; mixture weights, logsumexp and
// the logit gradients by DPP butterflies over the KP adjacent lanes, redundantly in every lane: no
// broadcast step, no second exchange.  Same formulas and IEEE divisions as diag_row_body; the summation
// orders differ.  In: per sweep i of this wavefront valid / mu / y / jitter draw u / exp(pre); the
// lane's component's logit.  Out: d mu, d pre (without the jitter-scale term) per sweep, the component's
// logit gradient, the row's logsumexp, this wavefront's lanes' partial sums of u * dL/dsigma.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// all-reduce over the K adjacent lanes of an aligned group (K = 4, 8, 16): xor 1, xor 2, mirror of 8, mirror of 16
template <int K>
__device__ __forceinline__ float kgroup_sum(float v) {
  v += dpp_mov<0xB1>(v);                            // quad_perm:[1,0,3,2]
  v += dpp_mov<0x4E>(v);                            // quad_perm:[2,3,0,1]
  if constexpr (K >= 8) v += dpp_mov<0x141>(v);     // row_half_mirror
  if constexpr (K >= 16) v += dpp_mov<0x140>(v);    // row_mirror
  return v;
}
template <int K>
__device__ __forceinline__ float kgroup_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  if constexpr (K >= 8) v = fmaxf(v, dpp_mov<0x141>(v));
  if constexpr (K >= 16) v = fmaxf(v, dpp_mov<0x140>(v));
  return v;
}
// all-reduce over the 16 / K lanes of a DPP row that hold the same component (stride K): rotations by 8, 4
template <int K>
__device__ __forceinline__ float dslots_sum(float v) {
  if constexpr (K <= 8) v += dpp_mov<0x128>(v);     // row_ror:8
  if constexpr (K <= 4) v += dpp_mov<0x124>(v);     // row_ror:4
  return v;
}
// the four values of the lane's quad
__device__ __forceinline__ f32x4 quad_gather(float v) {
  f32x4 q = {dpp_mov<0x00>(v), dpp_mov<0x55>(v), dpp_mov<0xAA>(v), dpp_mov<0xFF>(v)};
  return q;
}

// (Contraction of multiplies and adds is off, as in diag_row_body.)
template <int KP, int NQH, typename Eps, typename Bar>
__device__ __forceinline__ void diag_row_fast_core(int K, int D, bool active, bool kok, const bool (&valid)[NQH],
                                                   const float (&muv)[NQH], const float (&yd)[NQH], float (&eu)[NQH],
                                                   const float (&ev)[NQH], float lg_own, float min_w, float ll_limit,
                                                   float inv_norm, float* xq, float* xl, int h, int lane,
                                                   Eps& eps_src, Bar&& bar, float (&dmu)[NQH], float (&dpre)[NQH],
                                                   float& dlogit, float& lse, float& uds, bool& bad) {
#pragma clang fp contract(off)
  (void)K;
  // ---- mixture weights (mdnn.py:109-111), every lane its component -------------------------------
  const float mx = kgroup_max<KP>(kok ? lg_own : -INFINITY);
  const float e_own = kok ? expf(lg_own - mx) : 0.f;
  const float s_own = e_own / kgroup_sum<KP>(e_own);
  const float c_own = kok ? fminf(fmaxf(s_own, min_w), 1.0f) : 0.f;
  const float csum = kgroup_sum<KP>(c_own);
  const float w_own = c_own / csum;
  const float wc = fminf(fmaxf(w_own, min_w), 1.0f);     // the second clamp, mdnn.py:160
  const float lw = logf(wc);
  const float eps = eps_src.get();
  // ---- elements: sigma, z, log sigma ------------------------------------------------------------
  float quad = 0.f, logdet = 0.f, ez[NQH], rsg[NQH];
#pragma unroll
  for (int i = 0; i < NQH; ++i) {
    float sg = ev[i];
    if (eps != 0.f) sg = __builtin_fmaf(eu[i], eps, sg);
    else eu[i] = 0.f;
    if (!valid[i]) sg = 1.f;
    bad |= valid[i] && !(isfinite(muv[i]) && isfinite(sg));
    const float z = valid[i] ? (yd[i] - muv[i]) / sg : 0.f;
    quad += z * z;
    logdet += logf(sg);
    ez[i] = z; rsg[i] = sg;
  }
  // sums over the dimensions of each component: inside the DPP rows, then 4 rows x 2 wavefronts through LDS
  quad = dslots_sum<KP>(quad);
  logdet = dslots_sum<KP>(logdet);
  if (active && (lane & 15) < KP) { xq[4 * h + (lane >> 4)] = quad; xl[4 * h + (lane >> 4)] = logdet; }
  bar();
  float qs, ls;
  {
    const f32x4 a0 = *reinterpret_cast<const f32x4*>(xq), a1 = *reinterpret_cast<const f32x4*>(xq + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(xl), b1 = *reinterpret_cast<const f32x4*>(xl + 4);
    qs = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a1[0] + a1[1]) + (a1[2] + a1[3]));
    ls = ((b0[0] + b0[1]) + (b0[2] + b0[3])) + ((b1[0] + b1[1]) + (b1[2] + b1[3]));
  }
  const float logp = -0.5f * qs - ls - (float)D * kHalfLog2Pi;
  const float lp = fminf(fmaxf(logp, -ll_limit), ll_limit);
  const float rv = lp + lw;
  bad |= active && kok && !(isfinite(w_own) && isfinite(logp) && isfinite(rv));
  // logsumexp over the components (mdnn.py:163-178)
  const float m2 = kgroup_max<KP>(kok ? rv : -INFINITY);
  const float se = kgroup_sum<KP>(kok ? expf(rv - m2) : 0.f);
  lse = m2 + logf(se);
  const float sc = -expf(rv - lse) * inv_norm;
  const float g_lp = (logp >= -ll_limit && logp <= ll_limit) ? sc : 0.f;
  // ---- backward ----------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < NQH; ++i) {
    const float dsg = g_lp * __builtin_fmaf(ez[i], ez[i], -1.0f) / rsg[i];
    uds = valid[i] ? __builtin_fmaf(eu[i], dsg, uds) : uds;
    dmu[i] = g_lp * ez[i] / rsg[i];
    dpre[i] = dsg * ev[i];
  }
  const float gw = (kok && w_own >= min_w && w_own <= 1.0f) ? sc / wc : 0.f;
  const float s1 = kgroup_sum<KP>(gw * w_own);
  const float gsv = (kok && s_own >= min_w && s_own <= 1.0f) ? (gw - s1) / csum : 0.f;
  const float s2 = kgroup_sum<KP>(gsv * s_own);
  dlogit = s_own * (gsv - s2);

}

// jitter draw of element (row, d, k) for the thread-per-component kernels (full covariance)
__device__ inline float jitter_u(const HeadArgs& a, int row, int d, int k) {
  const int64_t e = ((int64_t)row * a.D + d) * a.K + k;
  if (a.noise) return a.noise[e];
  const uint64_t seed = a.dyn_rng ? a.dyn_rng[0] : a.seed;
  const uint64_t sid = a.dyn_rng ? a.dyn_rng[1] : a.stream_id;
  return u01(philox4x32_10(seed, sid, (uint64_t)e).v[0]);
}

// ---- full covariance, one wavefront per row (the persistent MDNN kernel's owners) ---------
// Lane k < K runs component k: forward substitution with T_k = diag(sigma_k) + strict lower
// (mdnn.py:152-158), back substitution for the gradients -- the same
// operations in the same order as mdn_nll_kernel<true> (mdn_head.hip), which the per-phase
// path runs with one thread per (row, component).  In: tile[Nh] raw head outputs, yv[D].  Out:
// gradients in tile[K..Nh) and dlg[K] (without the jitter-scale term), the row's logsumexp,
// the lane's sum(u * dL/dsigma); vq [3][D][K] scratch: v, q, and exp(pre) (left for the caller's
// jitter-scale correction).  `eps_fn()` as in diag_row.
template <typename EpsFn>
__device__ __forceinline__ void full_row(const HeadArgs& a, int row, bool active, int lane,
                                         float* T, const float* yv, float* rk, float* dlg,
                                         float* vq, EpsFn&& eps_fn, RowOut& out) {
  const int D = a.D, K = a.K, DK = D * K;
  const int k = lane < K ? lane : 0;
  const bool comp = active && lane < K;
  float* v_ = vq + k;              // v_[i * K]
  float* q_ = vq + DK + k;         // q_[i * K]
  float* s0 = vq + 2 * DK + k;     // exp(pre)[i * K]
  float logp = 0.f, w_k = 0.f, s_k = 0.f, csum = 1.f, mx = 0.f, den = 1.f;
  bool bad = false;
  if (comp) {   // softmax -> clamp -> renormalise, mdnn.py:109-111
    mx = T[0];
    for (int j = 1; j < K; ++j) mx = fmaxf(mx, T[j]);
    den = 0.f;
    for (int j = 0; j < K; ++j) den += expf(T[j] - mx);
    csum = 0.f;
    for (int j = 0; j < K; ++j) csum += fminf(fmaxf(expf(T[j] - mx) / den, a.min_w), 1.0f);
    s_k = expf(T[k] - mx) / den;
    w_k = fminf(fmaxf(s_k, a.min_w), 1.0f) / csum;
    bad |= !isfinite(w_k);
  }
  const float eps = eps_fn();
  if (comp) {
    float quad = 0.f, logdet = 0.f;
    for (int d = 0; d < D; ++d) {
      const float mu = T[K + d * K + k];
      const float sg0 = expf(T[K + DK + d * K + k]);
      s0[d * K] = sg0;
      float sg = sg0;
      if (eps != 0.f) sg += jitter_u(a, row, d, k) * eps;
      bad |= !(isfinite(mu) && isfinite(sg));
      float res = yv[d] - mu;
      const int base = K + 2 * DK + (d * (d - 1) / 2) * K + k;
      for (int j = 0; j < d; ++j) {
        const float lij = T[base + j * K];
        bad |= !isfinite(lij);
        res -= lij * v_[j * K];
      }
      const float vi = res / sg;
      v_[d * K] = vi;
      quad += vi * vi;
      logdet += logf(sg);
    }
    logp = -0.5f * quad - logdet - (float)D * kHalfLog2Pi;
    const float lp = fminf(fmaxf(logp, -a.ll_limit), a.ll_limit);   // mdnn.py:159
    const float wc = fminf(fmaxf(w_k, a.min_w), 1.0f);              // mdnn.py:160
    const float rv = lp + logf(wc);
    bad |= !(isfinite(logp) && isfinite(rv));
    rk[k] = rv;
  }
  __builtin_amdgcn_wave_barrier();
  float lse = 0.f;
  if (active) {
    float m2 = rk[0];
    for (int j = 1; j < K; ++j) m2 = fmaxf(m2, rk[j]);
    float se = 0.f;
    for (int j = 0; j < K; ++j) se += expf(rk[j] - m2);
    lse = m2 + logf(se);
  }
  float uds = 0.f, dlogit = 0.f;
  if (a.d_out != nullptr && comp) {
    const float sc = -expf(rk[k] - lse) * a.inv_norm;
    const float g_lp = (logp >= -a.ll_limit && logp <= a.ll_limit) ? sc : 0.f;
    for (int i = D - 1; i >= 0; --i) {   // q = T^{-T} v by back substitution
      float acc = v_[i * K];
      for (int j = i + 1; j < D; ++j) acc -= T[K + 2 * DK + (j * (j - 1) / 2 + i) * K + k] * q_[j * K];
      float sg = s0[i * K];
      if (eps != 0.f) sg += jitter_u(a, row, i, k) * eps;
      q_[i * K] = acc / sg;
    }
    for (int d = 0; d < D; ++d) {
      const float sg0 = s0[d * K];
      float sg = sg0, u = 0.f;
      if (eps != 0.f) { u = jitter_u(a, row, d, k); sg += u * eps; }
      const float qi = q_[d * K], vi = v_[d * K];
      const float dsg = g_lp * (qi * vi - 1.0f / sg);
      const int base = K + 2 * DK + (d * (d - 1) / 2) * K + k;
      for (int j = 0; j < d; ++j) T[base + j * K] = g_lp * qi * v_[j * K];
      uds += u * dsg;
      T[K + d * K + k] = g_lp * qi;
      T[K + DK + d * K + k] = dsg * sg0;
    }
    // mixture-weight path: second clamp, renormalisation, first clamp, softmax
    float s1 = 0.f;
    for (int j = 0; j < K; ++j) {
      const float sj = expf(T[j] - mx) / den;
      const float wj = fminf(fmaxf(sj, a.min_w), 1.0f) / csum;
      const float wcj = fminf(fmaxf(wj, a.min_w), 1.0f);
      const float scj = -expf(rk[j] - lse) * a.inv_norm;
      const float gwj = (wj >= a.min_w && wj <= 1.0f) ? scj / wcj : 0.f;
      s1 += gwj * wj;
    }
    float s2 = 0.f, gs_k = 0.f;
    for (int j = 0; j < K; ++j) {
      const float sj = expf(T[j] - mx) / den;
      const float wj = fminf(fmaxf(sj, a.min_w), 1.0f) / csum;
      const float wcj = fminf(fmaxf(wj, a.min_w), 1.0f);
      const float scj = -expf(rk[j] - lse) * a.inv_norm;
      const float gwj = (wj >= a.min_w && wj <= 1.0f) ? scj / wcj : 0.f;
      const float gcj = (gwj - s1) / csum;
      const float gsj = (sj >= a.min_w && sj <= 1.0f) ? gcj : 0.f;
      s2 += gsj * sj;
      if (j == k) gs_k = gsj;
    }
    dlogit = s_k * (gs_k - s2);
  }
  __builtin_amdgcn_wave_barrier();
  if (a.d_out != nullptr && comp) dlg[k] = dlogit;   // separate slot: the logits stay readable
  __builtin_amdgcn_wave_barrier();
  out.lse = lse; out.uds = uds; out.bad = bad;
}

// ---- cross-workgroup sums inside one launch -----------------------------------
// 8-byte {tag, value} granules: one relaxed agent-scope atomic store each
// (written through to memory: the per-XCD L2s are not coherent with each other
// inside a kernel), polled by lane g (+64, +128, +192) of every consuming wave
// and summed in granule order (bitwise reproducible).  The poll is bounded and
// raises bit 1 of `flag` instead of hanging.
// (array `g`, slot `slot`)
__device__ inline void granule_publish(unsigned long long* g, int slot, uint32_t tag, float v) {
  __hip_atomic_store(g + slot * kGranStride, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (All the polls of a round are issued before the first is looked at: with a test between them the
// compiler waits for each load in turn, and a 188-flag fan-in cost three memory round trips -- 1.6 us
// on flags that were already up -- instead of one.)
template <int NU>
__device__ __forceinline__ void granule_poll(const unsigned long long* g, int G, uint32_t tag, int lane,
                                             bool (&ok)[4], float (&v)[4]) {
  unsigned long long x[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u)
    x[u] = __hip_atomic_load(g + min(lane + 64 * u, G - 1) * kGranStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (!ok[u] && (uint32_t)(x[u] >> 32) == tag) { ok[u] = true; v[u] = __uint_as_float((uint32_t)x[u]); }
}
__device__ inline float granule_gather(unsigned long long* g, int G, uint32_t tag, int lane,
                                       int32_t* flag) {
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  bool ok[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) ok[u] = lane + 64 * u >= G;
  const int nu = (G + 63) >> 6;
  for (unsigned spin = 0;; ++spin) {
    if (nu <= 1) granule_poll<1>(g, G, tag, lane, ok, v);
    else if (nu == 2) granule_poll<2>(g, G, tag, lane, ok, v);
    else if (nu == 3) granule_poll<3>(g, G, tag, lane, ok, v);
    else granule_poll<4>(g, G, tag, lane, ok, v);
    if (__all(ok[0] && ok[1] && ok[2] && ok[3])) break;
    if (spin > (1u << 18)) {
      if (flag && lane == 0) atomicOr(flag, 2);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return (wave_sum_dpp(v[0]) + wave_sum_dpp(v[1])) + (wave_sum_dpp(v[2]) + wave_sum_dpp(v[3]));
}

__device__ inline void flag_raise(unsigned* flags, int slot, unsigned epoch) {
  __hip_atomic_store(flags + slot * kFlagStride, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until flags[0..G) >= epoch (G <= 256; one wavefront polls; the flags only grow)
template <int NU>
__device__ __forceinline__ bool flags_poll(const unsigned* flags, int G, unsigned epoch, int lane) {
  unsigned x[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u)
    x[u] = __hip_atomic_load(flags + min(lane + 64 * u, G - 1) * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool ok = true;
#pragma unroll
  for (int u = 0; u < NU; ++u) ok = ok && x[u] >= epoch;
  return __all(ok);
}
__device__ inline void flags_wait(unsigned* flags, int G, unsigned epoch, int lane, int32_t* flag) {
  const int nu = (G + 63) >> 6;
  for (unsigned spin = 0;; ++spin) {
    const bool done = nu <= 1 ? flags_poll<1>(flags, G, epoch, lane)
                    : nu == 2 ? flags_poll<2>(flags, G, epoch, lane)
                    : nu == 3 ? flags_poll<3>(flags, G, epoch, lane) : flags_poll<4>(flags, G, epoch, lane);
    if (done) break;
    if (spin > (1u << 18)) {
      if (flag && lane == 0) atomicOr(flag, 2);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// Quiet waiting.  A workgroup that reaches a fan-in long before its producers (a tile workgroup
// waiting ~10 us for the row owners, a row owner waiting for the next forward product) would poll all
// G flags every ~0.1 us for that long -- 255 workgroups x 100 lines of cache-bypassing reads per round,
// on the memory channels the producers' own latency-critical exchanges go through.  It first waits
// for ONE of the words (every lane reads the same address: one request per round), sleeping in
// between, and only then gathers the rest, which are up or nearly up by then.  Bounded like the others.
__device__ inline void flag_wait_one(const unsigned* flags, int slot, unsigned epoch, int32_t* flag) {
  for (unsigned spin = 0;; ++spin) {
    const unsigned x = __hip_atomic_load(flags + slot * kFlagStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (x >= epoch) break;
    if (spin > (1u << 17)) {
      if (flag) atomicOr(flag, 2);
      break;
    }
    __builtin_amdgcn_s_sleep(3);
  }
}
__device__ inline void granule_wait_one(const unsigned long long* g, int slot, uint32_t tag, int32_t* flag) {
  for (unsigned spin = 0;; ++spin) {
    const unsigned long long x = __hip_atomic_load(g + slot * kGranStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(x >> 32) == tag) break;
    if (spin > (1u << 17)) {
      if (flag) atomicOr(flag, 2);
      break;
    }
    __builtin_amdgcn_s_sleep(3);
  }
}

// The jitter scale EPS_NOISE * mean(exp(pre)) of an update from the owners' sum(exp(pre)) granules,
// for diag_row_impl: the granule loads are issued before the row arithmetic that does not need the
// scale and looked at after it -- by then the other owners' granules (published about when this
// owner published its own) have arrived, and the gather costs no round trip of its own.  A granule
// that was not up yet sends get() through the bounded polling loop.
struct GranuleEps {
  const unsigned long long* g; int G; uint32_t tag; int lane; int32_t* flag; float eps_noise, norm;
  unsigned long long x[4];
  __device__ __forceinline__ void issue() {
    if (eps_noise != 0.f) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        x[u] = __hip_atomic_load(g + min(lane + 64 * u, G - 1) * kGranStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __device__ __forceinline__ float get() {
    if (eps_noise == 0.f) return 0.f;
    float v[4];
    bool all = true;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool mine = lane + 64 * u < G, up = (uint32_t)(x[u] >> 32) == tag;
      v[u] = mine && up ? __uint_as_float((uint32_t)x[u]) : 0.f;
      all = all && (!mine || up);
    }
    float sum;
    if (__all(all)) sum = (wave_sum_dpp(v[0]) + wave_sum_dpp(v[1])) + (wave_sum_dpp(v[2]) + wave_sum_dpp(v[3]));
    else sum = granule_gather(const_cast<unsigned long long*>(g), G, tag, lane, flag);
    return eps_noise * (sum / norm);
  }
};

// write-through store / cache-bypassing load for data that crosses workgroups
// inside one launch
__device__ inline void xwg_store(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline float xwg_load(const float* p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT));
}

}  // namespace bsig
