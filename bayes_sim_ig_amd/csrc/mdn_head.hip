// Mixture-density head: forward() tuple, NLL and its backward, fused.
// Replaces bayes_sim_ig/models/mdnn.py:108-178 (softmax -> clamp -> renorm
// weights, exp + jitter diagonals, per-component MultivariateNormal.log_prob,
// clamp, + log clamp(w), -logsumexp, mean) and the autograd backward of that
// graph (closed forms: SURVEY.md Appendix A.3, checked against the reference's
// autograd in tests/test_oracle_mdn.py).
//
// Diagonal covariance (every BASELINE config): one WAVEFRONT per minibatch
// row.  The row's raw head outputs are staged in LDS with lane-contiguous
// loads; lane l < (64/K)*K owns component k = l % K and the dimensions
// d = l/K + q*(64/K); its per-element values stay in registers between the
// forward and the backward half; the per-component sums meet in LDS for the
// logsumexp; gradients overwrite the staged row in place and leave with
// lane-contiguous stores.  Full covariance: a workgroup owns R rows, thread
// (r, k) runs the forward / back substitution of component k from LDS.
// Row sums (loss, jitter terms) are reduced with wavefront shuffles, one
// partial per workgroup, summed in a fixed order by the finishing kernel
// (bitwise reproducible, no atomics).
#include "head.h"
#include "head_device.h"

#include <algorithm>

namespace bsig {

// eps = EPS_NOISE * mean(exp(pre)) from the partial sums (every workgroup
// adds the same values in the same order)
__device__ inline float jitter_eps(const HeadArgs& a, float* red) {
  float s = 0.f;
  for (int i0 = threadIdx.x; i0 < a.n_sig; i0 += 4 * blockDim.x) {
    float q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * blockDim.x;
      q[u] = i < a.n_sig ? a.sig_partials[i] : 0.f;
    }
    s += (q[0] + q[1]) + (q[2] + q[3]);
  }
  s = block_sum(s, red);
  return a.eps_noise * (s / ((float)a.batch * (float)(a.D * a.K)));
}

// sum_{b,d,k} exp(pre[b, d*K+k]) in kSigBlocks partials (for eps = EPS*mean(L_d),
// mdnn.py:115) when no producer kernel delivered them
__global__ __launch_bounds__(256) void sigma0_sum_kernel(const float* __restrict__ pre,
                                                         int64_t ld, int batch, int dk,
                                                         float* __restrict__ partials) {
  __shared__ float red[8];
  const int64_t total = (int64_t)batch * dk;
  float acc = 0.f;
  for (int64_t e0 = (int64_t)blockIdx.x * 256 + threadIdx.x; e0 < total;
       e0 += (int64_t)8 * gridDim.x * 256) {
    float q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t e = std::min<int64_t>(e0 + (int64_t)u * gridDim.x * 256, total - 1);
      const int64_t r = e / dk;
      q[u] = pre[r * ld + (e - r * dk)];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + (int64_t)u * gridDim.x * 256 < total) acc += expf(q[u]);
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = acc;
}

// ---- generic kernel: thread (r, k); full covariance and very wide diagonals ----
template <bool FULL>
__global__ void mdn_nll_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, K = a.K, Ls = a.Ls, Nh = a.Nh, R = a.R;
  const int DK = D * K;
  float* tile = smem;                 // [R][Nh]
  float* ybuf = tile + R * Nh;        // [R][D]
  float* rk = ybuf + R * D;           // [R][K]
  float* red = rk + R * K;            // [16]
  float* vq = red + 16;               // FULL: [2][D][blockDim.x]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int row0 = blockIdx.x * R;
  const int nrows = min(R, a.batch - row0);

  // ---- stage the R rows (four segments, each lane-contiguous) -------------
  const int64_t yoff = a.y_dyn ? (int64_t)a.y_dyn[0] * a.y_dyn_stride : 0;
  for (int r = 0; r < nrows; ++r) {
    const int64_t row = row0 + r;
    float* t = tile + r * Nh;
    for (int j = tid; j < K; j += nt) t[j] = a.seg_w[row * a.ld_w + j];
    for (int j = tid; j < DK; j += nt) t[K + j] = a.seg_mu[row * a.ld_mu + j];
    for (int j = tid; j < DK; j += nt) t[K + DK + j] = a.seg_sg[row * a.ld_sg + j];
    if (FULL)
      for (int j = tid; j < Ls * K; j += nt) t[K + 2 * DK + j] = a.seg_lo[row * a.ld_lo + j];
    const int64_t yrow = a.y_rows ? (int64_t)a.y_rows[row + yoff] : row + yoff;
    for (int j = tid; j < D; j += nt) ybuf[r * D + j] = a.y[yrow * a.ldy + j];
  }
  float eps = 0.f;
  if (!a.from_tuple && a.eps_noise != 0.f) eps = jitter_eps(a, red);
  __syncthreads();

  const int r = tid / K, k = tid - r * K;
  const bool active = r < nrows;
  const int row = row0 + r;
  float* T = tile + r * Nh;
  const float* yv = ybuf + r * D;
  float* v_ = vq + tid;                       // v_[i*nt]
  float* q_ = vq + (size_t)D * nt + tid;      // q_[i*nt]

  float logp = 0.f, w_k = 0.f, s_k = 0.f, csum = 1.f, mx = 0.f, den = 1.f;
  bool bad = false;
  if (active) {
    if (a.from_tuple) {
      w_k = T[k];
    } else {  // softmax -> clamp -> renormalise, mdnn.py:109-111
      mx = T[0];
      for (int j = 1; j < K; ++j) mx = fmaxf(mx, T[j]);
      den = 0.f;
      for (int j = 0; j < K; ++j) den += expf(T[j] - mx);
      csum = 0.f;
      for (int j = 0; j < K; ++j)
        csum += fminf(fmaxf(expf(T[j] - mx) / den, a.min_w), 1.0f);
      s_k = expf(T[k] - mx) / den;
      w_k = fminf(fmaxf(s_k, a.min_w), 1.0f) / csum;
    }
    bad |= !isfinite(w_k);
    float quad = 0.f, logdet = 0.f;
    for (int d = 0; d < D; ++d) {
      const float mu = T[K + d * K + k];
      const float sraw = T[K + DK + d * K + k];
      float sg;
      if (a.from_tuple) sg = sraw;
      else {
        sg = expf(sraw);
        if (eps != 0.f) sg += jitter_u(a, row, d, k) * eps;
      }
      bad |= !(isfinite(mu) && isfinite(sg));
      float res = yv[d] - mu;
      if (FULL) {  // forward substitution with T = diag(sigma) + strict lower
        const int base = K + 2 * DK + (d * (d - 1) / 2) * K + k;
        for (int j = 0; j < d; ++j) {
          const float lij = T[base + j * K];
          bad |= !isfinite(lij);
          res -= lij * v_[j * nt];
        }
        const float vi = res / sg;
        v_[d * nt] = vi;
        quad += vi * vi;
      } else {
        const float z = res / sg;
        quad += z * z;
      }
      logdet += logf(sg);
    }
    logp = -0.5f * quad - logdet - (float)D * kHalfLog2Pi;
    const float lp = fminf(fmaxf(logp, -a.ll_limit), a.ll_limit);   // mdnn.py:159
    const float wc = fminf(fmaxf(w_k, a.min_w), 1.0f);              // mdnn.py:160
    const float rv = lp + logf(wc);
    bad |= !(isfinite(logp) && isfinite(rv));
    rk[r * K + k] = rv;
  }
  __syncthreads();

  float lse = 0.f;
  if (active) {
    const float* rr = rk + r * K;
    float m2 = rr[0];
    for (int j = 1; j < K; ++j) m2 = fmaxf(m2, rr[j]);
    float se = 0.f;
    for (int j = 0; j < K; ++j) se += expf(rr[j] - m2);
    lse = m2 + logf(se);
  }
  const float lse_sum = block_sum((active && k == 0) ? lse : 0.f, red);
  if (tid == 0) a.block_lse[blockIdx.x] = lse_sum;

  float uds = 0.f, dlogit = 0.f;
  const bool bwd = a.d_out != nullptr;
  if (bwd && active) {
    const float* rr = rk + r * K;
    const float sc = -expf(rr[k] - lse) * a.inv_norm;
    const float g_lp = (logp >= -a.ll_limit && logp <= a.ll_limit) ? sc : 0.f;
    if (FULL) {  // q = T^{-T} v by back substitution
      for (int i = D - 1; i >= 0; --i) {
        float acc = v_[i * nt];
        for (int j = i + 1; j < D; ++j)
          acc -= T[K + 2 * DK + (j * (j - 1) / 2 + i) * K + k] * q_[j * nt];
        const float sraw = T[K + DK + i * K + k];
        float sg = a.from_tuple ? sraw : expf(sraw);
        if (!a.from_tuple && eps != 0.f) sg += jitter_u(a, row, i, k) * eps;
        q_[i * nt] = acc / sg;
      }
    }
    for (int d = 0; d < D; ++d) {
      const float mu = T[K + d * K + k];
      const float sraw = T[K + DK + d * K + k];
      float sg0 = 1.f, sg, u = 0.f;
      if (a.from_tuple) sg = sraw;
      else {
        sg0 = expf(sraw);
        sg = sg0;
        if (eps != 0.f) { u = jitter_u(a, row, d, k); sg += u * eps; }
      }
      float dmu, dsg;
      if (FULL) {
        const float qi = q_[d * nt], vi = v_[d * nt];
        dmu = g_lp * qi;
        dsg = g_lp * (qi * vi - 1.0f / sg);
        const int base = K + 2 * DK + (d * (d - 1) / 2) * K + k;
        for (int j = 0; j < d; ++j) T[base + j * K] = g_lp * qi * v_[j * nt];
      } else {
        const float z = (yv[d] - mu) / sg;
        dmu = g_lp * z / sg;
        dsg = g_lp * (z * z - 1.0f) / sg;
      }
      uds += u * dsg;
      T[K + d * K + k] = dmu;
      T[K + DK + d * K + k] = dsg * sg0;
    }
    // mixture-weight path: second clamp, renormalisation, first clamp, softmax
    if (a.from_tuple) {
      const float wc = fminf(fmaxf(w_k, a.min_w), 1.0f);
      dlogit = (w_k >= a.min_w && w_k <= 1.0f) ? sc / wc : 0.f;   // d/d weights
    } else {
      float s1 = 0.f;  // sum_j g_w[j] * w[j]
      for (int j = 0; j < K; ++j) {
        const float sj = expf(T[j] - mx) / den;
        const float wj = fminf(fmaxf(sj, a.min_w), 1.0f) / csum;
        const float wcj = fminf(fmaxf(wj, a.min_w), 1.0f);
        const float scj = -expf(rr[j] - lse) * a.inv_norm;
        const float gwj = (wj >= a.min_w && wj <= 1.0f) ? scj / wcj : 0.f;
        s1 += gwj * wj;
      }
      float s2 = 0.f, gs_k = 0.f;  // sum_j g_s[j] * s[j]
      for (int j = 0; j < K; ++j) {
        const float sj = expf(T[j] - mx) / den;
        const float wj = fminf(fmaxf(sj, a.min_w), 1.0f) / csum;
        const float wcj = fminf(fmaxf(wj, a.min_w), 1.0f);
        const float scj = -expf(rr[j] - lse) * a.inv_norm;
        const float gwj = (wj >= a.min_w && wj <= 1.0f) ? scj / wcj : 0.f;
        const float gcj = (gwj - s1) / csum;
        const float gsj = (sj >= a.min_w && sj <= 1.0f) ? gcj : 0.f;
        s2 += gsj * sj;
        if (j == k) gs_k = gsj;
      }
      dlogit = s_k * (gs_k - s2);
    }
  }
  const float uds_sum = block_sum(uds, red);  // also orders logits reads before writes
  if (tid == 0 && a.block_uds) a.block_uds[blockIdx.x] = uds_sum;
  if (bwd) {
    if (active) T[k] = dlogit;
    __syncthreads();
    for (int r2 = 0; r2 < nrows; ++r2) {
      float* o = a.d_out + (int64_t)(row0 + r2) * a.ld_dout;
      const float* t = tile + r2 * Nh;
      for (int j = tid; j < Nh; j += nt) o[j] = t[j];
    }
  }
  if (bad && a.nonfinite) atomicOr(a.nonfinite, 1);
}

template <int WPB, int NQCAP = kElemsPerLane>
__global__ __launch_bounds__(WPB * 64) void mdn_nll_diag_wave_kernel(HeadArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int D = a.D, K = a.K, Nh = a.Nh;
  const int DK = D * K;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int per_wave = Nh + D + 3 * K;
  float* wsum = smem;                      // [2][WPB] per-wave lse / uds
  float* tile = smem + 32 + wid * per_wave;  // [Nh]
  float* yv = tile + Nh;                   // [D]
  float* rk = yv + D;                      // [K] clamp(logp)+log clamp(w)
  float* lpk = rk + K;                     // [K] raw logp (clamp indicator)
  float* dlg = lpk + K;                    // [K] d loss / d logits
  const int row = blockIdx.x * WPB + wid;
  const bool active = row < a.batch;

  // every independent global load is issued before anything waits
  const bool want_eps = !a.from_tuple && a.eps_noise != 0.f;
  float sp[4] = {0.f, 0.f, 0.f, 0.f};
  if (want_eps) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = lane + 64 * u;
      sp[u] = i < a.n_sig ? a.sig_partials[i] : 0.f;
    }
  }
  if (active) {
    const int64_t yoff = a.y_dyn ? (int64_t)a.y_dyn[0] * a.y_dyn_stride : 0;
    const int64_t yrow = a.y_rows ? (int64_t)a.y_rows[row + yoff] : row + yoff;
    const bool one_run = a.seg_mu == a.seg_w + K && a.seg_sg == a.seg_mu + DK &&
                         a.ld_mu == a.ld_w && a.ld_sg == a.ld_w && Nh <= 64 * 8;
    if (one_run) {        // the row is one contiguous run: eight loads in flight
      const float* src = a.seg_w + (int64_t)row * a.ld_w;
      float r[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = src[min(lane + 64 * u, Nh - 1)];
      const float yq = a.y[yrow * a.ldy + min(lane, D - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (lane + 64 * u < Nh) tile[lane + 64 * u] = r[u];
      if (lane < D) yv[lane] = yq;
      for (int j = lane + 64; j < D; j += 64) yv[j] = a.y[yrow * a.ldy + j];
    } else {
      for (int j = lane; j < K; j += 64) tile[j] = a.seg_w[(int64_t)row * a.ld_w + j];
      for (int j = lane; j < DK; j += 64) tile[K + j] = a.seg_mu[(int64_t)row * a.ld_mu + j];
      for (int j = lane; j < DK; j += 64) tile[K + DK + j] = a.seg_sg[(int64_t)row * a.ld_sg + j];
      for (int j = lane; j < D; j += 64) yv[j] = a.y[yrow * a.ldy + j];
    }
  }
  // jitter scale eps = EPS_NOISE * mean(exp(pre)): every wave adds the same
  // partial sums in the same order
  float eps = 0.f;
  if (want_eps) {
    float s = (sp[0] + sp[1]) + (sp[2] + sp[3]);
    // (a split-K head product delivers up to 2048 partials: eight loads in flight per lane --
    // one at a time, the 28 dependent round trips were most of this kernel at minibatch 8192)
    for (int base = 256; base < a.n_sig; base += 512) {
      float q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = base + lane + 64 * u;
        q[u] = i < a.n_sig ? a.sig_partials[i] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += q[u];
    }
    eps = a.eps_noise * (wave_sum(s) / ((float)a.batch * (float)DK));
  }
  __builtin_amdgcn_wave_barrier();

  RowOut ro;
  diag_row_capped<NQCAP>(a, row, active, lane, tile, yv, rk, lpk, dlg, [&] { return eps; }, ro);
  const bool bwd = a.d_out != nullptr;
  if (bwd && active) {
    float* o = a.d_out + (int64_t)row * a.ld_dout;
    for (int j = lane; j < K; j += 64) o[j] = dlg[j];
    for (int j = K + lane; j < Nh; j += 64) o[j] = tile[j];
  }
  // per-wave sums -> one partial per workgroup, fixed order
  const float uds_w = wave_sum(ro.uds);
  if (lane == 0) { wsum[wid] = active ? ro.lse : 0.f; wsum[WPB + wid] = uds_w; }
  __syncthreads();
  if (tid == 0) {
    float sl = 0.f, su = 0.f;
#pragma unroll
    for (int w = 0; w < WPB; ++w) { sl += wsum[w]; su += wsum[WPB + w]; }
    a.block_lse[blockIdx.x] = sl;
    if (a.block_uds) a.block_uds[blockIdx.x] = su;
  }
  if (ro.bad && a.nonfinite) atomicOr(a.nonfinite, 1);
}

// Fit-engine state advance (single writer; every reader of these words runs
// in an earlier or a later kernel, or has published its last dependent value).
__device__ inline void run_finish_hook(const FinishHook& hook) {
  int32_t* st = hook.state;
  if (hook.kind == 1) {           // end of the forward half of update `step`
    // beta^t as running products (double): no pow() on the device
    double* bp = reinterpret_cast<double*>(st + 12);
    const double b1t = bp[0] * hook.beta1, b2t = bp[1] * hook.beta2;
    bp[0] = b1t; bp[1] = b2t;
    reinterpret_cast<float*>(st)[4] = (float)(hook.lr / (1.0 - b1t));
    reinterpret_cast<float*>(st)[5] = (float)(1.0 / sqrt(1.0 - b2t));
    st[0] = st[0] + 1;
  } else {                        // end of a held-out evaluation
    st[1] = st[1] + 1;
  }
  reinterpret_cast<uint64_t*>(st + 8)[1] += 1;   // jitter RNG stream
}

// Finishing kernel: loss = -(sum of block partials) / batch; jitter-scale
// gradient term d pre += (EPS/(B*D*K)) * sum(u * dL/dsigma) * exp(pre) (the
// non-detached mean of mdnn.py:115); column sums of the corrected d_out (the
// head bias gradients).  Grid: x = 64-column groups of d_out, y = row slabs;
// block = 64 columns x 16 row lanes.
constexpr int kFinishLanes = 16;
__global__ __launch_bounds__(64 * kFinishLanes) void mdn_finish_kernel(
    const float* __restrict__ block_lse, const float* __restrict__ block_uds, int nblocks,
    int batch, int pre_begin, int dk, int nh, float eps_noise, const float* __restrict__ pre,
    int64_t ld_pre, float* __restrict__ d_out, int64_t ld_dout, float* __restrict__ colsum,
    int rows_per_slab, float* __restrict__ loss, const int32_t* __restrict__ loss_slot,
    int32_t* __restrict__ nonfinite, FinishHook hook) {
  __shared__ float red[16];
  __shared__ float part[kFinishLanes][64];
  if (blockIdx.x == 0 && blockIdx.y == 0 && loss) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += block_lse[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
      const float l = -s / (float)batch;
      loss[loss_slot ? *loss_slot : 0] = l;
      if (!isfinite(l) && nonfinite) atomicOr(nonfinite, 1);
    }
  }
  if (hook.state && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) run_finish_hook(hook);
  if (!d_out) return;
  float c = 0.f;
  if (eps_noise != 0.f) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) s += block_uds[i];
    s = block_sum(s, red);
    c = eps_noise / ((float)batch * (float)dk) * s;
  }
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_slab;
  const int r1 = min(batch, r0 + rows_per_slab);
  const bool fix = c != 0.f && col >= pre_begin && col < pre_begin + dk;
  float acc = 0.f;
  if (col < nh) {
    for (int row0 = r0 + rl; row0 < r1; row0 += 8 * kFinishLanes) {   // 8 rows in flight
      float dv[8], pv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = min(row0 + kFinishLanes * u, r1 - 1);
        dv[u] = d_out[(int64_t)row * ld_dout + col];
        pv[u] = fix ? pre[(int64_t)row * ld_pre + (col - pre_begin)] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int row = row0 + kFinishLanes * u;
        if (row < r1) {
          float v = dv[u];
          if (fix) {
            v += c * expf(pv[u]);
            d_out[(int64_t)row * ld_dout + col] = v;
          }
          acc += v;
        }
      }
    }
  }
  if (!colsum) return;
  part[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && col < nh) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < kFinishLanes; ++q) t += part[q][cl];
    colsum[(int64_t)blockIdx.y * nh + col] = t;
  }
}

// forward() tuple, mdnn.py:109-119
__global__ __launch_bounds__(256) void mdn_outputs_kernel(
    const float* __restrict__ o, int64_t ld, int batch, int D, int K, int Ls,
    const float* __restrict__ noise, uint64_t seed, uint64_t stream_id, float eps_noise,
    float min_w, const float* __restrict__ sig_partials, int n_sig, float* __restrict__ weights,
    float* __restrict__ mu, float* __restrict__ l_d, float* __restrict__ lower,
    int32_t* __restrict__ nonfinite) {
  __shared__ float red[8];
  const int DK = D * K, Nh = K + 2 * DK + Ls * K;
  float eps = 0.f;
  if (eps_noise != 0.f) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n_sig; i += blockDim.x) s += sig_partials[i];
    s = block_sum(s, red);
    eps = eps_noise * (s / ((float)batch * (float)DK));
  }
  bool bad = false;
  for (int row = blockIdx.x; row < batch; row += gridDim.x) {
    const float* t = o + (int64_t)row * ld;
    for (int j = threadIdx.x; j < Nh; j += blockDim.x) {
      float v;
      if (j < K) {
        float mx = t[0];
        for (int q = 1; q < K; ++q) mx = fmaxf(mx, t[q]);
        float den = 0.f;
        for (int q = 0; q < K; ++q) den += expf(t[q] - mx);
        float csum = 0.f;
        for (int q = 0; q < K; ++q) csum += fminf(fmaxf(expf(t[q] - mx) / den, min_w), 1.f);
        v = fminf(fmaxf(expf(t[j] - mx) / den, min_w), 1.f) / csum;
        weights[(int64_t)row * K + j] = v;
      } else if (j < K + DK) {
        v = t[j];
        mu[(int64_t)row * DK + (j - K)] = v;
      } else if (j < K + 2 * DK) {
        const int e = j - K - DK;
        v = expf(t[j]);
        if (eps != 0.f) {
          const int64_t ne = (int64_t)row * DK + e;  // [B, D, K] flat == d*K+k
          const float u = noise ? noise[ne] : u01(philox4x32_10(seed, stream_id, (uint64_t)ne).v[0]);
          v += u * eps;
        }
        l_d[(int64_t)row * DK + e] = v;
      } else {
        v = t[j];
        lower[(int64_t)row * Ls * K + (j - K - 2 * DK)] = v;
      }
      bad |= !isfinite(v);
    }
  }
  if (bad && nonfinite) atomicOr(nonfinite, 1);
}

// ---------------------------------------------------------------- host side
int colsum_launch(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                  void* workspace, size_t workspace_bytes, hipStream_t st);

constexpr int kWavesPerBlock = 8;     // diag kernel: rows per workgroup
constexpr int kMaxSlabs = 64;         // finish kernel: row slabs for tall batches

struct HeadGeom {
  int D, K, Ls, Nh, R, threads; size_t lds; int blocks;
  bool wave_per_row; int slabs, rows_per_slab;
};

static int head_geom(const bsig_head_dims* d, int64_t batch, HeadGeom* g) {
  BSIG_REQUIRE(d && d->out_dim >= 1 && d->n_comp >= 1, "mdn head: bad dims");
  BSIG_REQUIRE(d->n_comp <= 64, "mdn head: at most 64 components");
  g->D = d->out_dim; g->K = d->n_comp;
  g->Ls = d->full_cov ? d->out_dim * (d->out_dim - 1) / 2 : 0;
  g->Nh = g->K + 2 * g->D * g->K + g->Ls * g->K;
  const int groups = 64 / g->K;
  g->wave_per_row = g->Ls == 0 && ceil_div(g->D, groups) <= kElemsPerLane;
  const size_t wave_lds =
      (32 + (size_t)kWavesPerBlock * (g->Nh + g->D + 3 * g->K)) * sizeof(float);
  if (g->wave_per_row && wave_lds > 60 * 1024) g->wave_per_row = false;
  if (g->wave_per_row) {
    g->R = kWavesPerBlock; g->threads = kWavesPerBlock * 64; g->lds = wave_lds;
  } else {
    const int rmax = 256 / g->K;
    int R = (int)std::min<int64_t>(rmax, std::max<int64_t>(1, ceil_div<int64_t>(batch, 128)));
    for (;; --R) {
      const int threads = (int)round_up(R * g->K, 64);
      const size_t lds = ((size_t)R * (g->Nh + g->D + g->K) + 16 +
                          (g->Ls ? (size_t)2 * g->D * threads : 0)) * sizeof(float);
      if (lds <= 60 * 1024 || R == 1) {
        g->R = R; g->threads = threads; g->lds = lds;
        break;
      }
    }
  }
  if (g->lds > 64 * 1024) {
    set_error("mdn head: %zu B of LDS needed for one workgroup", g->lds);
    return BSIG_EUNSUPPORTED;
  }
  g->blocks = (int)ceil_div<int64_t>(batch, g->R);
  // (128 rows per slab: one pass of 8 rows in flight per thread -- at 512 rows per slab the 80
  // workgroups of a minibatch of 8192 made four dependent passes: 21 us)
  g->slabs = batch > 1024 ? (int)std::min<int64_t>(ceil_div<int64_t>(batch, 128), kMaxSlabs) : 1;
  g->rows_per_slab = (int)ceil_div<int64_t>(batch, g->slabs);
  return BSIG_OK;
}

// workspace floats: [sig partials kSigMax][block_lse nblk][block_uds nblk][colsum slabs x Nh]
static size_t head_ws_floats(const HeadGeom& g) {
  return kSigMax + 2 * (size_t)g.blocks + (size_t)kMaxSlabs * g.Nh;
}

int head_sig_capacity() { return kSigMax; }

int mdn_head_nll_launch(const bsig_head_dims* dims, const float* seg_w, int64_t ld_w,
                        const float* seg_mu, int64_t ld_mu, const float* seg_sg,
                        int64_t ld_sg, const float* seg_lo, int64_t ld_lo, int from_tuple,
                        const float* y, int64_t ldy, const int32_t* y_rows, int64_t batch,
                        int64_t norm_batch, const float* noise, uint64_t seed,
                        uint64_t stream_id, const uint64_t* dyn_rng, float* loss,
                        const int32_t* loss_slot, float* d_out, int64_t ld_dout,
                        float* colsum_out, int32_t* nonfinite, void* workspace,
                        size_t workspace_bytes, hipStream_t st, const HeadDyn* dyn,
                        HeadBiasPartials* bias_partials) {
  HeadGeom g;
  BSIG_TRY(head_geom(dims, batch, &g));
  BSIG_REQUIRE(batch >= 1 && batch < (1 << 30), "mdn head: bad batch");
  BSIG_REQUIRE(workspace && workspace_bytes >= head_ws_floats(g) * sizeof(float),
               "mdn head: workspace too small (%zu < %zu)", workspace_bytes,
               head_ws_floats(g) * sizeof(float));
  BSIG_REQUIRE(!(g.Ls > 0 && !seg_lo), "mdn head: full covariance needs lower entries");
  BSIG_REQUIRE(!(colsum_out && !d_out), "mdn head: bias gradients need d_head_out");
  float* ws = reinterpret_cast<float*>(workspace);
  float* sig_partials = ws;
  float* block_lse = ws + kSigMax;
  float* block_uds = block_lse + g.blocks;
  float* slab_sums = block_uds + g.blocks;
  const int DK = g.D * g.K;
  const bool jitter = !from_tuple && dims->eps_noise != 0.f;
  int n_sig = dyn ? dyn->n_sig_ready : 0;   // partial sums delivered by the producer of head_out
  BSIG_REQUIRE(n_sig >= 0 && n_sig <= kSigMax, "mdn head: bad n_sig_ready");
  if (jitter && n_sig == 0) {
    n_sig = (int)std::min<int64_t>(kSigBlocks, ceil_div<int64_t>(batch * DK, 2048));
    hipLaunchKernelGGL(sigma0_sum_kernel, dim3(n_sig), dim3(256), 0, st, seg_sg, ld_sg,
                       (int)batch, DK, sig_partials);
    BSIG_CHECK_LAUNCH("sigma0_sum");
  }
  HeadArgs a{};
  a.seg_w = seg_w; a.ld_w = ld_w; a.seg_mu = seg_mu; a.ld_mu = ld_mu;
  a.seg_sg = seg_sg; a.ld_sg = ld_sg; a.seg_lo = seg_lo; a.ld_lo = ld_lo;
  a.y = y; a.ldy = ldy; a.y_rows = y_rows;
  a.y_dyn = dyn ? dyn->y_dyn : nullptr; a.y_dyn_stride = dyn ? dyn->y_dyn_stride : 0;
  a.batch = (int)batch; a.inv_norm = 1.0f / (float)norm_batch;
  a.D = g.D; a.K = g.K; a.Ls = g.Ls; a.Nh = g.Nh; a.R = g.R;
  a.from_tuple = from_tuple;
  a.noise = noise; a.seed = seed; a.stream_id = stream_id; a.dyn_rng = dyn_rng;
  a.eps_noise = from_tuple ? 0.f : dims->eps_noise;
  a.min_w = dims->min_weight; a.ll_limit = dims->ll_limit;
  a.sig_partials = sig_partials; a.n_sig = n_sig;
  a.d_out = d_out; a.ld_dout = ld_dout;
  a.block_lse = block_lse; a.block_uds = block_uds; a.nonfinite = nonfinite;
  if (g.wave_per_row) {
    // (rows of at most two sweeps -- D <= 2 * (64 / K): every BASELINE shape -- run the kernel that
    // carries only the two-sweep row body: fewer registers, twice the rows per CU in flight)
    const int nq = ceil_div(g.D, 64 / g.K);
    if (nq <= 2)
      hipLaunchKernelGGL((mdn_nll_diag_wave_kernel<kWavesPerBlock, 2>), dim3(g.blocks), dim3(g.threads), g.lds, st, a);
    else
      hipLaunchKernelGGL((mdn_nll_diag_wave_kernel<kWavesPerBlock, kElemsPerLane>), dim3(g.blocks), dim3(g.threads),
                         g.lds, st, a);
  }
  else if (g.Ls > 0)
    hipLaunchKernelGGL(mdn_nll_kernel<true>, dim3(g.blocks), dim3(g.threads), g.lds, st, a);
  else
    hipLaunchKernelGGL(mdn_nll_kernel<false>, dim3(g.blocks), dim3(g.threads), g.lds, st, a);
  BSIG_CHECK_LAUNCH("mdn_nll");
  // finish: loss, jitter-scale gradient correction, head bias gradients
  const bool correct = jitter && d_out != nullptr;
  const bool sweep = correct || colsum_out != nullptr;
  const dim3 fgrid(sweep ? (unsigned)ceil_div(g.Nh, 64) : 1u, sweep ? (unsigned)g.slabs : 1u);
  float* cs = colsum_out ? (g.slabs > 1 ? slab_sums : colsum_out) : nullptr;
  hipLaunchKernelGGL(mdn_finish_kernel, fgrid, dim3(64 * kFinishLanes), 0, st, block_lse,
                     block_uds, g.blocks, (int)batch, g.K + DK, DK, g.Nh,
                     correct ? dims->eps_noise : 0.f, seg_sg, ld_sg, sweep ? d_out : nullptr,
                     ld_dout, cs, g.rows_per_slab, loss, loss_slot, nonfinite,
                     dyn ? dyn->hook : FinishHook());
  BSIG_CHECK_LAUNCH("mdn_finish");
  if (bias_partials) *bias_partials = HeadBiasPartials{};
  if (colsum_out && g.slabs > 1) {
    if (bias_partials) *bias_partials = HeadBiasPartials{slab_sums, g.slabs, g.Nh};   // the caller adds the slabs up
    else BSIG_TRY(colsum_launch(slab_sums, g.Nh, g.slabs, g.Nh, colsum_out, nullptr, 0, st));
  }
  return BSIG_OK;
}

}  // namespace bsig

using namespace bsig;

extern "C" int64_t bsig_head_width(const bsig_head_dims* d) {
  if (!d) return -1;
  const int64_t ls = d->full_cov ? (int64_t)d->out_dim * (d->out_dim - 1) / 2 : 0;
  return d->n_comp + 2 * (int64_t)d->out_dim * d->n_comp + ls * d->n_comp;
}

extern "C" size_t bsig_head_workspace_bytes(const bsig_head_dims* d, int64_t batch) {
  HeadGeom g;
  if (batch < 1) batch = 1;
  if (head_geom(d, batch, &g) != BSIG_OK) return 0;
  return head_ws_floats(g) * sizeof(float);
}

extern "C" int bsig_mdn_head_outputs(const bsig_head_dims* dims, const float* head_out,
                                     int64_t ld, int64_t batch, const float* noise,
                                     uint64_t seed, uint64_t stream_id, float* weights,
                                     float* mu, float* l_d, float* lower, int32_t* nonfinite,
                                     void* workspace, size_t workspace_bytes,
                                     bsig_stream_t stream) {
  BSIG_REQUIRE(dims && head_out && weights && mu && l_d, "head_outputs: null pointer");
  BSIG_REQUIRE(batch >= 1, "head_outputs: empty batch");
  const int D = dims->out_dim, K = dims->n_comp;
  const int Ls = dims->full_cov ? D * (D - 1) / 2 : 0;
  BSIG_REQUIRE(!(Ls > 0 && !lower), "head_outputs: full covariance needs `lower`");
  BSIG_REQUIRE(ld >= bsig_head_width(dims), "head_outputs: ld too small");
  BSIG_REQUIRE(workspace && workspace_bytes >= kSigMax * sizeof(float),
               "head_outputs: workspace too small");
  float* sig_partials = reinterpret_cast<float*>(workspace);
  hipStream_t st = as_stream(stream);
  int n_sig = 0;
  if (dims->eps_noise != 0.f) {
    n_sig = (int)std::min<int64_t>(kSigBlocks, ceil_div<int64_t>(batch * D * K, 2048));
    hipLaunchKernelGGL(sigma0_sum_kernel, dim3(n_sig), dim3(256), 0, st,
                       head_out + K + D * K, ld, (int)batch, D * K, sig_partials);
    BSIG_CHECK_LAUNCH("sigma0_sum");
  }
  hipLaunchKernelGGL(mdn_outputs_kernel, dim3((int)std::min<int64_t>(batch, 2048)), dim3(256),
                     0, st, head_out, ld, (int)batch, D, K, Ls, noise, seed, stream_id,
                     dims->eps_noise, dims->min_weight, sig_partials, n_sig, weights, mu, l_d,
                     lower, nonfinite);
  BSIG_CHECK_LAUNCH("mdn_outputs");
  return BSIG_OK;
}

extern "C" int bsig_mdn_nll_from_tuple(const bsig_head_dims* dims, const float* weights,
                                       const float* mu, const float* l_d, const float* lower,
                                       const float* y, int64_t ldy, int64_t batch, float* loss,
                                       int32_t* nonfinite, void* workspace,
                                       size_t workspace_bytes, bsig_stream_t stream) {
  BSIG_REQUIRE(dims && weights && mu && l_d && y && loss, "nll_from_tuple: null pointer");
  const int64_t D = dims->out_dim, K = dims->n_comp;
  const int64_t Ls = dims->full_cov ? D * (D - 1) / 2 : 0;
  return mdn_head_nll_launch(dims, weights, K, mu, D * K, l_d, D * K, lower, Ls * K, 1, y,
                             ldy, nullptr, batch, batch, nullptr, 0, 0, nullptr, loss, nullptr,
                             nullptr, 0, nullptr, nonfinite, workspace, workspace_bytes,
                             as_stream(stream), nullptr);
}

extern "C" int bsig_mdn_head_nll(const bsig_head_dims* dims, const float* head_out,
                                 int64_t ld, const float* y, int64_t ldy,
                                 const int32_t* y_rows, int64_t batch, int64_t norm_batch,
                                 const float* noise, uint64_t seed, uint64_t stream_id,
                                 float* loss, float* d_head_out, int32_t* nonfinite,
                                 void* workspace, size_t workspace_bytes,
                                 bsig_stream_t stream) {
  BSIG_REQUIRE(dims && head_out && y, "head_nll: null pointer");
  BSIG_REQUIRE(ld >= bsig_head_width(dims), "head_nll: ld too small");
  BSIG_REQUIRE(norm_batch >= 1, "head_nll: norm_batch must be >= 1");
  const int64_t D = dims->out_dim, K = dims->n_comp;
  return mdn_head_nll_launch(dims, head_out, ld, head_out + K, ld, head_out + K + D * K, ld,
                             dims->full_cov ? head_out + K + 2 * D * K : nullptr, ld, 0, y,
                             ldy, y_rows, batch, norm_batch, noise, seed, stream_id, nullptr,
                             loss, nullptr, d_head_out, ld, nullptr, nonfinite, workspace,
                             workspace_bytes, as_stream(stream), nullptr);
}
