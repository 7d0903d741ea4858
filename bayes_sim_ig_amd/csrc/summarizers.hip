// Trajectory summarizers as HBM-streaming HIP kernels (gfx950).
// Replaces the PyTorch op sequences of bayes_sim_ig/utils/summarizers.py
// (reference file:line cited per kernel).  One workgroup per trajectory;
// all global traffic is lane-contiguous.
#include "common.h"

#include <algorithm>

namespace bsig {

// ------------------------------------------------------------------ K1
// summary_start / summary_waypts: summarizers.py:65-87 after the crop/pad of
// :20-62.  out[n, t*(sd+ad)+c] = c<sd ? s[n,min(t,Ts-1),c] : a[n,min(t,Ta-1),c-sd]
// Algorithmic bytes / trajectory: 2 * 4 * W*(sd+ad).
template <int UNROLL_T>
__global__ __launch_bounds__(256) void summary_start_kernel(
    const float* __restrict__ states, const float* __restrict__ actions,
    float* __restrict__ out, int64_t n, int ts, int ta, int sd, int ad, int w,
    int64_t ld_out) {
  const int width = sd + ad;
  for (int64_t traj = blockIdx.x; traj < n; traj += gridDim.x) {
    const float* s = states + traj * (int64_t)ts * sd;
    const float* a = actions + traj * (int64_t)ta * ad;
    float* o = out + traj * ld_out;
    for (int t0 = 0; t0 < w; t0 += UNROLL_T) {
      for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float v[UNROLL_T];
#pragma unroll
        for (int u = 0; u < UNROLL_T; ++u) {
          const int t = t0 + u;
          if (t < w) {
            v[u] = (c < sd) ? s[(int64_t)min(t, ts - 1) * sd + c]
                            : a[(int64_t)min(t, ta - 1) * ad + (c - sd)];
          }
        }
#pragma unroll
        for (int u = 0; u < UNROLL_T; ++u) {
          const int t = t0 + u;
          if (t < w) o[(int64_t)t * width + c] = v[u];
        }
      }
    }
  }
}

// mean and unbiased std of a feature row (summarizers.py:114-119: torch.mean / torch.std, two
// passes), taken by one wavefront from LDS.  Accumulated in fp64 and rounded once: the value every
// fp32 summation order approximates (the reference's own order is a property of the torch build),
// at 5 elements per lane.  Every lane returns the same value.
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ void row_mean_std(const float* sf, int S, int ln, float& mean, float& sdev) {
  double part = 0.0;
  for (int i = ln; i < S; i += 64) part += (double)sf[i];
  const double m = wave_sum_f64(part) / (double)S;
  part = 0.0;
  for (int i = ln; i < S; i += 64) {
    const double d = (double)sf[i] - m;
    part += d * d;
  }
  mean = (float)m;
  sdev = (S < 2) ? 0.f : (float)sqrt(wave_sum_f64(part) / (double)(S - 1));
}

// ------------------------------------------------------------------ K2
// cross_correlation: summarizers.py:90-122.
//   sf[t*(sd-1)+c] = s[t,c+1]-s[t,c]  (corrdiff, :105-106)  or  s[t,c] (:108)
//   af[t*ad+c]     = a[t,c]
//   out[i*A + j]   = sf[i]*af[j]      (bmm outer product, :112-113)
//   out[S*A]       = mean(sf), out[S*A+1] = unbiased std(sf)   (:114-119)
// Algorithmic bytes / trajectory: 4*(W*(sd+ad) + S*A + 2); write-bound.
// `fac` (optional): the row's FACTORS [sf S | af A | mean | std | 1 | 0..] -- everything the
// summary is made of, 1/35 of its size for Ant; `out` may then be null (no outer product):
// the fit engine's first layer forms sf[i]*af[j] itself (bsig_crosscorr_factors).
__global__ __launch_bounds__(256) void crosscorr_kernel(
    const float* __restrict__ states, const float* __restrict__ actions,
    float* __restrict__ out, int64_t n, int ts, int ta, int sd, int ad, int w,
    int use_diff, int64_t ld_out, int vec4, int32_t* __restrict__ nonfinite,
    float* __restrict__ fac, int64_t ld_fac) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int sfeat = sd - 1;
  const int S = w * sfeat, A = w * ad;
  float* sf = smem;            // [S]
  float* af = smem + S;        // [A]
  const int tid = threadIdx.x, nt = blockDim.x;
  // A workgroup that stops storing while it fetches and reduces its next trajectory leaves the
  // chip's write path idle for that time: with the fetch -> LDS -> mean -> std -> max chain of six
  // workgroup barriers per trajectory the kernel ran at 0.8 of what the same store loop does alone
  // (tools/micro/fill_bench.hip).  So the NEXT trajectory's features are fetched into registers
  // before this one's store loop (their latency hides behind it), and the statistics are taken by
  // every wavefront for itself (same values, same order in each: no barrier) -- two barriers per
  // trajectory.
  constexpr int kPfW = 10;                 // window <= 10 (crosscorr_window)
  const bool pf_ok = sfeat <= nt && ad <= nt && w <= kPfW;   // (else: fetched in place, below)
  float pf_s0[kPfW], pf_s1[kPfW], pf_a[kPfW];
  auto prefetch = [&](int64_t traj) {
    const float* s = states + traj * (int64_t)ts * sd;
    const float* a = actions + traj * (int64_t)ta * ad;
#pragma unroll
    for (int t = 0; t < kPfW; ++t) {
      if (t < w) {
        const float* srow = s + (int64_t)t * sd;
        const float* arow = a + (int64_t)min(t, ta - 1) * ad;   // pad: repeat the last step (:52-58)
        if (tid < sfeat) { pf_s0[t] = srow[tid]; pf_s1[t] = use_diff ? srow[tid + 1] : 0.f; }
        if (tid < ad) pf_a[t] = arow[tid];
      }
    }
  };
  if (pf_ok && (int64_t)blockIdx.x < n) prefetch(blockIdx.x);
  for (int64_t traj = blockIdx.x; traj < n; traj += gridDim.x) {
    const float* s = states + traj * (int64_t)ts * sd;
    const float* a = actions + traj * (int64_t)ta * ad;
    float* o = out ? out + traj * ld_out : nullptr;
    __syncthreads();
    if (pf_ok) {
#pragma unroll
      for (int t = 0; t < kPfW; ++t) {
        if (t < w) {
          if (tid < sfeat) sf[t * sfeat + tid] = use_diff ? pf_s1[t] - pf_s0[t] : pf_s0[t];
          if (tid < ad) af[t * ad + tid] = pf_a[t];
        }
      }
    } else {
      // state features: iterate (t, c) without integer division
      for (int t = 0; t < w; ++t) {
        const float* srow = s + (int64_t)t * sd;
        for (int c = tid; c < sfeat; c += nt)
          sf[t * sfeat + c] = use_diff ? (srow[c + 1] - srow[c]) : srow[c];
        // actions beyond their own length repeat the last step (pad, :52-58)
        const float* arow = a + (int64_t)min(t, ta - 1) * ad;
        for (int c = tid; c < ad; c += nt) af[t * ad + c] = arow[c];
      }
    }
    __syncthreads();
    if (pf_ok && traj + gridDim.x < n) prefetch(traj + gridDim.x);
    // mean and unbiased std (two passes, like torch.std), per wavefront
    const int ln = tid & 63;
    float mean, sdev;
    row_mean_std(sf, S, ln, mean, sdev);
    // isfinite(feats) (summarizers.py:120) without touching every product: all
    // inputs finite and max|sf| * max|af| far from overflow => every product is
    // finite; otherwise fall back to checking each product.
    float ms = 0.f, ma = 0.f;
    bool in_bad = false;
    for (int i = ln; i < S; i += 64) { ms = fmaxf(ms, fabsf(sf[i])); in_bad |= !isfinite(sf[i]); }
    for (int i = ln; i < A; i += 64) { ma = fmaxf(ma, fabsf(af[i])); in_bad |= !isfinite(af[i]); }
    ms = wave_max(ms); ma = wave_max(ma);
    const bool check_each = !(ms < 1e18f && ma < 1e18f);
    if (fac) {
      float* f = fac + traj * ld_fac;
      for (int i = tid; i < S; i += nt) f[i] = sf[i];
      for (int i = tid; i < A; i += nt) f[S + i] = af[i];
      for (int i = S + A + tid; i < ld_fac; i += nt)
        f[i] = i == S + A ? mean : (i == S + A + 1 ? sdev : (i == S + A + 2 ? 1.0f : 0.f));
    }
    if (!o) {   // factors only: |sf[i] af[j]| <= max|sf| max|af|
      if (nonfinite && (in_bad || (tid == 0 && !(isfinite(ms * ma) && isfinite(mean) && isfinite(sdev)))))
        atomicOr(nonfinite, 1);
      continue;
    }
    // streaming outer product
    const int64_t total = (int64_t)S * A;
    bool bad = in_bad;
    if (vec4 && (A & 3) == 0 && A <= 4 * nt) {
      // A thread keeps ONE quad of action features (columns 4 jq .. 4 jq + 3) in registers and walks
      // down the state features: per 16-byte store one LDS broadcast read and four multiplies
      // (the generic loop below spends ~30 instructions per store on index arithmetic and eight
      // LDS reads).  Consecutive threads cover consecutive quads of a row, then the next row: the
      // stores of a wavefront are one contiguous run.
      const int qpr = A >> 2, rstep = nt / qpr;
      const int jq = tid % qpr, r0 = tid / qpr;
      if (r0 < rstep) {
        const float4 a4 = *reinterpret_cast<const float4*>(af + 4 * jq);
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        f32x4_t* orow = reinterpret_cast<f32x4_t*>(o) + jq;
        for (int i = r0; i < S; i += rstep) {
          const float sv = sf[i];
          const f32x4_t pk = {sv * a4.x, sv * a4.y, sv * a4.z, sv * a4.w};
          if (check_each) bad |= !(isfinite(pk.x) && isfinite(pk.y) && isfinite(pk.z) && isfinite(pk.w));
          __builtin_nontemporal_store(pk, orow + (int64_t)i * qpr);
        }
      }
    } else if (vec4) {
      // rows are 16-B aligned (ld_out % 4 == 0): one float4 per lane
      const int64_t nvec = total >> 2;
      const int step_i = (4 * nt) / A, step_j = (4 * nt) % A;
      int64_t e = 4 * (int64_t)tid;
      int i = (int)(e / A), j = (int)(e % A);
      for (int64_t q = tid; q < nvec; q += nt) {
        float4 v;
        int ii = i, jj = j;
        v.x = sf[ii] * af[jj]; if (++jj == A) { jj = 0; ++ii; }
        v.y = sf[ii] * af[jj]; if (++jj == A) { jj = 0; ++ii; }
        v.z = sf[ii] * af[jj]; if (++jj == A) { jj = 0; ++ii; }
        v.w = sf[ii] * af[jj];
        if (check_each) bad |= !(isfinite(v.x) && isfinite(v.y) && isfinite(v.z) && isfinite(v.w));
        typedef float f32x4_t __attribute__((ext_vector_type(4)));
        const f32x4_t pk = {v.x, v.y, v.z, v.w};
        // (one 16-byte streaming store; four 4-byte ones run at the same 4.4 TB/s, cached stores at 4.1)
        __builtin_nontemporal_store(pk, reinterpret_cast<f32x4_t*>(o + 4 * q));
        i += step_i; j += step_j;
        if (j >= A) { j -= A; ++i; }
      }
      for (int64_t e2 = (nvec << 2) + tid; e2 < total; e2 += nt) {
        const float v = sf[e2 / A] * af[e2 % A];
        bad |= !isfinite(v);
        o[e2] = v;
      }
    } else {
      const int step_i = nt / A, step_j = nt % A;
      int i = tid / A, j = tid % A;
      for (int64_t e = tid; e < total; e += nt) {
        const float v = sf[i] * af[j];
        bad |= !isfinite(v);
        __builtin_nontemporal_store(v, o + e);
        i += step_i; j += step_j;
        if (j >= A) { j -= A; ++i; }
      }
    }
    if (tid == 0) {
      o[total] = mean;
      o[total + 1] = sdev;
      bad |= !(isfinite(mean) && isfinite(sdev));
    }
    if (bad && nonfinite) atomicOr(nonfinite, 1);
  }
}

// The materialised summary alone (bsig_crosscorr, vector-width rows, A % 4 == 0): the same
// arithmetic as crosscorr_kernel's quad path in a kernel small enough for 8 workgroups per CU, with
// every wavefront's 1 KB run of stores starting on a 128-byte line.  The kernel is write-bound and
// its rows (4 (S A + 2) bytes, padded to 16) start at arbitrary 16-byte offsets: streaming stores
// that straddle lines ran at 4.0-4.4 TB/s where the same bytes in line-aligned runs take 5.2
// (tools/micro/fill_bench.hip, profiles/r03_fill_bench.txt).  So the thread <-> quad map is rotated
// per trajectory by the row's offset within a line: thread t owns quads shift + t + k * act of the
// row (act = rstep * A/4 threads store per sweep, a multiple of 8 quads = 128 bytes); the up to
// seven quads in front of the first line boundary are stored by threads 0..shift-1 on the side.
__global__ __launch_bounds__(256) void crosscorr_quads_kernel(
    const float* __restrict__ states, const float* __restrict__ actions, float* __restrict__ out,
    int64_t n, int ts, int ta, int sd, int ad, int w, int use_diff, int64_t ld_out, int rstep,
    int32_t* __restrict__ nonfinite) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const int sfeat = sd - 1;
  const int S = w * sfeat, A = w * ad;
  float* sf = smem;            // [S]
  float* af = smem + S;        // [A]
  const int tid = threadIdx.x, ln = tid & 63;
  const int qpr = A >> 2, act = rstep * qpr;
  constexpr int kPfW = 10;
  float pf_s0[kPfW], pf_s1[kPfW], pf_a[kPfW];
  auto prefetch = [&](int64_t traj) {     // (see crosscorr_kernel)
    const float* s = states + traj * (int64_t)ts * sd;
    const float* a = actions + traj * (int64_t)ta * ad;
#pragma unroll
    for (int t = 0; t < kPfW; ++t) {
      if (t < w) {
        const float* srow = s + (int64_t)t * sd;
        const float* arow = a + (int64_t)min(t, ta - 1) * ad;
        if (tid < sfeat) { pf_s0[t] = srow[tid]; pf_s1[t] = use_diff ? srow[tid + 1] : 0.f; }
        if (tid < ad) pf_a[t] = arow[tid];
      }
    }
  };
  if ((int64_t)blockIdx.x < n) prefetch(blockIdx.x);
  for (int64_t traj = blockIdx.x; traj < n; traj += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < kPfW; ++t) {
      if (t < w) {
        if (tid < sfeat) sf[t * sfeat + tid] = use_diff ? pf_s1[t] - pf_s0[t] : pf_s0[t];
        if (tid < ad) af[t * ad + tid] = pf_a[t];
      }
    }
    __syncthreads();
    if (traj + gridDim.x < n) prefetch(traj + gridDim.x);
    float mean, sdev;
    row_mean_std(sf, S, ln, mean, sdev);
    float ms = 0.f, ma = 0.f;
    bool bad = false;
    for (int i = ln; i < S; i += 64) { ms = fmaxf(ms, fabsf(sf[i])); bad |= !isfinite(sf[i]); }
    for (int i = ln; i < A; i += 64) { ma = fmaxf(ma, fabsf(af[i])); bad |= !isfinite(af[i]); }
    ms = wave_max(ms); ma = wave_max(ma);
    const bool check_each = !(ms < 1e18f && ma < 1e18f);   // (see crosscorr_kernel)
    const int64_t base_q = (traj * ld_out) >> 2;
    f32x4_t* o4 = reinterpret_cast<f32x4_t*>(out) + base_q;
    const int shift = (int)((8 - (base_q & 7)) & 7);
    if (tid < shift) {                       // quads in front of the first line boundary
      const int i = tid / qpr, jq = tid - i * qpr;
      const float sv = sf[i];
      const float4 a4 = *reinterpret_cast<const float4*>(af + 4 * jq);
      const f32x4_t pk = {sv * a4.x, sv * a4.y, sv * a4.z, sv * a4.w};
      if (check_each) bad |= !(isfinite(pk.x) && isfinite(pk.y) && isfinite(pk.z) && isfinite(pk.w));
      o4[tid] = pk;
    }
    if (tid < act) {
      const int q0 = shift + tid;
      const int i0 = q0 / qpr, jq = q0 - i0 * qpr;
      const float4 a4 = *reinterpret_cast<const float4*>(af + 4 * jq);
      f32x4_t* orow = o4 + jq;
      for (int i = i0; i < S; i += rstep) {
        const float sv = sf[i];
        const f32x4_t pk = {sv * a4.x, sv * a4.y, sv * a4.z, sv * a4.w};
        if (check_each) bad |= !(isfinite(pk.x) && isfinite(pk.y) && isfinite(pk.z) && isfinite(pk.w));
        __builtin_nontemporal_store(pk, orow + (int64_t)i * qpr);
      }
    }
    if (tid == 0) {
      float* o = out + traj * ld_out;
      o[(int64_t)S * A] = mean;
      o[(int64_t)S * A + 1] = sdev;
      bad |= !(isfinite(mean) && isfinite(sdev));
    }
    if (bad && nonfinite) atomicOr(nonfinite, 1);
  }
}

// Small summaries (Cartpole: 302 floats): one wavefront per trajectory, four
// per workgroup, so a launch still covers the chip with lane-contiguous stores.
__global__ __launch_bounds__(256) void crosscorr_wave_kernel(
    const float* __restrict__ states, const float* __restrict__ actions,
    float* __restrict__ out, int64_t n, int ts, int ta, int sd, int ad, int w,
    int use_diff, int64_t ld_out, int32_t* __restrict__ nonfinite,
    float* __restrict__ fac, int64_t ld_fac) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int sfeat = sd - 1;
  const int S = w * sfeat, A = w * ad;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* sf = smem + wid * (S + A);
  float* af = sf + S;
  for (int64_t base = (int64_t)blockIdx.x * 4; base < n; base += (int64_t)gridDim.x * 4) {
    const int64_t traj = base + wid;
    const bool active = traj < n;
    __syncthreads();
    if (active) {
      const float* s = states + traj * (int64_t)ts * sd;
      const float* a = actions + traj * (int64_t)ta * ad;
      for (int e = lane; e < S; e += 64) {
        const int t = e / sfeat, c = e - t * sfeat;
        const float* srow = s + (int64_t)t * sd;
        sf[e] = use_diff ? (srow[c + 1] - srow[c]) : srow[c];
      }
      for (int e = lane; e < A; e += 64) {
        const int t = e / ad, c = e - t * ad;
        af[e] = a[(int64_t)min(t, ta - 1) * ad + c];
      }
    }
    __syncthreads();
    if (!active) continue;
    float part = 0.f;
    for (int i = lane; i < S; i += 64) part += sf[i];
    const float mean = wave_sum(part) / (float)S;
    part = 0.f;
    for (int i = lane; i < S; i += 64) {
      const float dlt = sf[i] - mean;
      part += dlt * dlt;
    }
    const float ss = wave_sum(part);
    const float sdev = (S < 2) ? 0.f : sqrtf(ss / (float)(S - 1));
    if (fac) {
      float* f = fac + traj * ld_fac;
      for (int i = lane; i < S + A; i += 64) f[i] = sf[i];     // af follows sf in LDS
      for (int i = S + A + lane; i < ld_fac; i += 64)
        f[i] = i == S + A ? mean : (i == S + A + 1 ? sdev : (i == S + A + 2 ? 1.0f : 0.f));
    }
    if (!out) {   // factors only: a non-finite product needs a non-finite or overflowing pair
      float ms = 0.f, ma = 0.f;
      bool in_bad = false;
      for (int i = lane; i < S; i += 64) { ms = fmaxf(ms, fabsf(sf[i])); in_bad |= !isfinite(sf[i]); }
      for (int i = lane; i < A; i += 64) { ma = fmaxf(ma, fabsf(af[i])); in_bad |= !isfinite(af[i]); }
      ms = wave_max(ms); ma = wave_max(ma);
      if (nonfinite && (in_bad || !(isfinite(ms * ma) && isfinite(mean) && isfinite(sdev))))
        atomicOr(nonfinite, 1);
      continue;
    }
    float* o = out + traj * ld_out;
    const int total = S * A;
    bool bad = false;
    const int step_i = 64 / A, step_j = 64 % A;
    int i = lane / A, j = lane % A;
    for (int e = lane; e < total; e += 64) {
      const float v = sf[i] * af[j];
      bad |= !isfinite(v);
      o[e] = v;
      i += step_i; j += step_j;
      if (j >= A) { j -= A; ++i; }
    }
    if (lane == 0) {
      o[total] = mean;
      o[total + 1] = sdev;
      bad |= !(isfinite(mean) && isfinite(sdev));
    }
    if (bad && nonfinite) atomicOr(nonfinite, 1);
  }
}

// ------------------------------------------------------------------ K3
// summary_signatory: summarizers.py:144-168.  Path X_l = [l+1 | s_l | a_l]
// (:152-155), signature levels 1..depth in signatory's layout.  Chen's
// identity per segment, Horner form:
//   S3[i,j,k] += (S2[i,j] + (S1[i] + D[i]/3) * D[j]/2) * D[k]
//   S2[i,j]   += (S1[i] + D[i]/2) * D[j]
//   S1[i]      = X_l[i] - X_0[i]
// One thread per (i,j) pair keeps S2[i,j] and the S3[i,j,:] row in registers
// for the whole path; the result is staged in LDS and stored lane-contiguous (straight from the
// registers -- 16 bytes per lane at a stride of 4 d bytes -- the same stores ran at 2.4 instead of
// 4.0 TB/s: a wavefront's store instruction wants one contiguous run).
// Algorithmic bytes / trajectory: 4*(L*d + d + d^2 + d^3).
template <int DMAX>
__global__ void signature3_kernel(const float* __restrict__ states,
                                  const float* __restrict__ actions,
                                  float* __restrict__ out, int64_t n, int length,
                                  int sd, int ad, int64_t ld_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = 1 + sd + ad;
  const int dp = (d + 3) & ~3;                 // increment rows padded to float4
  float* path = smem;                          // [length * d]
  float* delta = smem + ((length * d + 3) & ~3);   // [(length-1) * dp], zero padded
  // level-3 terms are staged at the same offset modulo 4 floats as they have in the output row
  // (o + d + d*d): 16-byte chunks are then aligned on both sides
  const int mis = (d + d * d) & 3;
  float* stage = delta + (length - 1) * dp + mis;    // [d*d*d]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int npairs = d * d;
  const int i = tid / d, j = tid % d;
  const bool active = tid < npairs;
  // the next trajectory's samples are fetched into registers before this one's store phase
  // (one element per thread: their latency hides behind it instead of opening every trajectory)
  const bool pf_ok = length * sd <= nt && length * ad <= nt;
  float pf_s = 0.f, pf_a = 0.f;
  auto prefetch = [&](int64_t traj) {
    if (tid < length * sd) pf_s = states[traj * (int64_t)length * sd + tid];
    if (tid < length * ad) pf_a = actions[traj * (int64_t)length * ad + tid];
  };
  if (pf_ok && (int64_t)blockIdx.x < n) prefetch(blockIdx.x);
  for (int64_t traj = blockIdx.x; traj < n; traj += gridDim.x) {
    const float* s = states + traj * (int64_t)length * sd;
    const float* a = actions + traj * (int64_t)length * ad;
    float* o = out + traj * ld_out;
    __syncthreads();
    if (pf_ok) {
      if (tid < length * sd) { const int l = tid / sd, c = tid - l * sd; path[l * d + 1 + c] = pf_s; }
      if (tid < length * ad) { const int l = tid / ad, c = tid - l * ad; path[l * d + 1 + sd + c] = pf_a; }
    } else {
      for (int e = tid; e < length * sd; e += nt) {
        const int l = e / sd, c = e - l * sd;
        path[l * d + 1 + c] = s[e];
      }
      for (int e = tid; e < length * ad; e += nt) {
        const int l = e / ad, c = e - l * ad;
        path[l * d + 1 + sd + c] = a[e];
      }
    }
    for (int l = tid; l < length; l += nt) path[l * d] = (float)(l + 1);
    __syncthreads();
    if (pf_ok && traj + gridDim.x < n) prefetch(traj + gridDim.x);
    for (int e = tid; e < (length - 1) * dp; e += nt) {
      const int l = e / dp, c = e - l * dp;
      delta[e] = c < d ? path[(l + 1) * d + c] - path[l * d + c] : 0.f;
    }
    __syncthreads();
    float s2 = 0.f;
    float s3[DMAX];
#pragma unroll
    for (int k = 0; k < DMAX; ++k) s3[k] = 0.f;
    if (active) {
      const float x0i = path[i];
      for (int l = 0; l + 1 < length; ++l) {
        const float* dl = delta + l * dp;
        const float di = dl[i], dj = dl[j];
        const float s1i = path[l * d + i] - x0i;
        const float coef = s2 + (s1i + di * (1.0f / 3.0f)) * dj * 0.5f;
#pragma unroll
        for (int k4 = 0; k4 < DMAX / 4; ++k4) {        // broadcast ds_read_b128
          if (4 * k4 < d) {
            const float4 q = *reinterpret_cast<const float4*>(dl + 4 * k4);
            s3[4 * k4 + 0] = fmaf(coef, q.x, s3[4 * k4 + 0]);
            s3[4 * k4 + 1] = fmaf(coef, q.y, s3[4 * k4 + 1]);
            s3[4 * k4 + 2] = fmaf(coef, q.z, s3[4 * k4 + 2]);
            s3[4 * k4 + 3] = fmaf(coef, q.w, s3[4 * k4 + 3]);
          }
        }
        s2 = fmaf(s1i + di * 0.5f, dj, s2);
      }
      if ((d & 1) == 0) {   // 8-byte aligned row of the stage: ds_write_b64
#pragma unroll
        for (int k = 0; k < DMAX; k += 2)
          if (k < d)
            *reinterpret_cast<float2*>(stage + tid * d + k) = make_float2(s3[k], s3[k + 1]);
      } else {
#pragma unroll
        for (int k = 0; k < DMAX; ++k)
          if (k < d) stage[tid * d + k] = s3[k];
      }
      o[d + tid] = s2;                                   // level 2
    }
    for (int c = tid; c < d; c += nt)                    // level 1
      o[c] = path[(length - 1) * d + c] - path[c];
    __syncthreads();
    float* o3 = o + d + npairs;
    const int n3 = npairs * d;
    if ((ld_out & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
      // head up to the first 128-byte LINE of the output row (a wavefront's 64 quads are then whole
      // lines: streaming stores that straddle lines run at 4.0-4.4 TB/s, line-aligned runs at 5.2,
      // tools/micro/fill_bench.hip), aligned quads, tail.  (head = (4 - mis) mod 4: the stage has
      // the row's 16-byte phase)
      const int head = min((int)((32 - ((reinterpret_cast<uintptr_t>(o3) >> 2) & 31)) & 31), n3);
      const int nq = (n3 - head) >> 2;
      if (tid < head) __builtin_nontemporal_store(stage[tid], o3 + tid);
      typedef float f32x4_t __attribute__((ext_vector_type(4)));
      const f32x4_t* sq = reinterpret_cast<const f32x4_t*>(stage + head);
      f32x4_t* oq = reinterpret_cast<f32x4_t*>(o3 + head);
      for (int q = tid; q < nq; q += nt) __builtin_nontemporal_store(sq[q], oq + q);
      for (int e = head + 4 * nq + tid; e < n3; e += nt) __builtin_nontemporal_store(stage[e], o3 + e);
    } else {
      for (int e = tid; e < n3; e += nt) __builtin_nontemporal_store(stage[e], o3 + e);
    }
  }
}

// depth <= 2 for wider paths: S2[i,j] = sum_l (X_l[i]-X_0[i] + D_l[i]/2) D_l[j]
__global__ __launch_bounds__(256) void signature12_kernel(
    const float* __restrict__ states, const float* __restrict__ actions,
    float* __restrict__ out, int64_t n, int length, int sd, int ad, int depth,
    int64_t ld_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int d = 1 + sd + ad;
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int64_t traj = blockIdx.x; traj < n; traj += gridDim.x) {
    const float* s = states + traj * (int64_t)length * sd;
    const float* a = actions + traj * (int64_t)length * ad;
    float* o = out + traj * ld_out;
    if (depth == 1) {  // last - first row (time channel: L-1)
      for (int c = tid; c < d; c += nt) {
        float v;
        if (c == 0) v = (float)(length - 1);
        else if (c <= sd) v = s[(int64_t)(length - 1) * sd + c - 1] - s[c - 1];
        else v = a[(int64_t)(length - 1) * ad + c - 1 - sd] - a[c - 1 - sd];
        o[c] = v;
      }
      continue;
    }
    float* path = smem;  // [length * d]
    __syncthreads();
    for (int l = 0; l < length; ++l) {
      float* row = path + l * d;
      if (tid == 0) row[0] = (float)(l + 1);
      for (int c = tid; c < sd; c += nt) row[1 + c] = s[(int64_t)l * sd + c];
      for (int c = tid; c < ad; c += nt) row[1 + sd + c] = a[(int64_t)l * ad + c];
    }
    __syncthreads();
    for (int c = tid; c < d; c += nt) o[c] = path[(length - 1) * d + c] - path[c];
    const int n2 = d * d;
    const int step_i = nt / d, step_j = nt % d;
    int i = tid / d, j = tid % d;
    for (int e = tid; e < n2; e += nt) {
      const float x0i = path[i];
      float acc = 0.f;
      for (int l = 0; l + 1 < length; ++l) {
        const float* p0 = path + l * d;
        const float* p1 = p0 + d;
        acc = fmaf((p0[i] - x0i) + (p1[i] - p0[i]) * 0.5f, p1[j] - p0[j], acc);
      }
      o[d + e] = acc;
      i += step_i; j += step_j;
      if (j >= d) { j -= d; ++i; }
    }
  }
}

static int grid_for(int64_t n) {
  // one workgroup per trajectory, capped; the kernels grid-stride
  const int64_t cap = 256 * 64;
  return (int)(n < cap ? n : cap);
}

static int ref_signature_depth(int64_t d) {  // summarizers.py:133-141
  for (int depth = 3; depth >= 0; --depth) {
    int64_t p = 1;
    for (int q = 0; q < depth; ++q) p *= d;
    if (p <= 110 * 110) return depth;
  }
  return 1;
}

}  // namespace bsig

using namespace bsig;

extern "C" int64_t bsig_summary_dim(int kind, int traj_len, int sd, int ad, int depth) {
  if (kind == 0) return 10 * (int64_t)(sd + ad);
  if (kind == 1 || kind == 2) {
    int w = sd > 50 ? 5 : 10;
    if (traj_len <= w) w = traj_len;
    return (int64_t)w * (sd - 1) * w * ad + 2;
  }
  if (kind == 3) {
    const int64_t d = 1 + sd + ad;
    if (depth <= 0) depth = ref_signature_depth(d);
    int64_t tot = 0, p = 1;
    for (int q = 1; q <= depth; ++q) { p *= d; tot += p; }
    return tot;
  }
  return -1;
}

extern "C" int bsig_summary_start(const float* states, const float* actions, float* out,
                                  int64_t n, int t_states, int t_actions, int sd, int ad,
                                  int max_t, int64_t ld_out, bsig_stream_t stream) {
  bsig::Range roctx_range("bsig_summary_start");
  if (n == 0) return BSIG_OK;
  BSIG_REQUIRE(states && actions && out, "summary_start: null pointer");
  BSIG_REQUIRE(n >= 0 && t_states >= 1 && t_actions >= 1 && sd >= 1 && ad >= 1 && max_t >= 1,
               "summary_start: bad dims n=%lld ts=%d ta=%d sd=%d ad=%d max_t=%d",
               (long long)n, t_states, t_actions, sd, ad, max_t);
  BSIG_REQUIRE(ld_out >= (int64_t)max_t * (sd + ad), "summary_start: ld_out too small");
  if (n == 0) return BSIG_OK;
  const int width = sd + ad;
  const int threads = width >= 192 ? 256 : (width >= 96 ? 128 : 64);
  hipLaunchKernelGGL((summary_start_kernel<10>), dim3(grid_for(n)), dim3(threads), 0,
                     as_stream(stream), states, actions, out, n, t_states, t_actions, sd,
                     ad, max_t, ld_out);
  BSIG_CHECK_LAUNCH("summary_start");
  return BSIG_OK;
}

namespace bsig {
static int crosscorr_window(int t_states, int sd) {
  int w = sd > 50 ? 5 : 10;                       // summarizers.py:96-98
  if (t_states <= w) w = t_states;                // :99 (only crop when longer)
  return w;
}

// the materialised summary (`out`), its factors (`fac`), or both
static int crosscorr_launch(const float* states, const float* actions, float* out, int64_t ld_out,
                            float* fac, int64_t ld_fac, int64_t n, int t_states, int t_actions,
                            int sd, int ad, int use_state_diff, int32_t* nonfinite,
                            hipStream_t st) {
  if (n == 0) return BSIG_OK;
  BSIG_REQUIRE(states && actions && (out || fac), "crosscorr: null pointer");
  BSIG_REQUIRE(n >= 0 && sd >= 2 && ad >= 1, "crosscorr: bad dims sd=%d ad=%d", sd, ad);
  BSIG_REQUIRE(t_states > 1, "crosscorr: traj_len must be > 1 (summarizers.py:94)");
  BSIG_REQUIRE(t_actions >= 1, "crosscorr: no actions");
  const int w = crosscorr_window(t_states, sd);
  const int64_t S = (int64_t)w * (sd - 1), A = (int64_t)w * ad;
  BSIG_REQUIRE(!out || ld_out >= S * A + 2, "crosscorr: ld_out too small");
  BSIG_REQUIRE(!fac || ld_fac >= S + A + 3, "crosscorr: ld_factors too small (need S + A + 3 = %lld)",
               (long long)(S + A + 3));
  const size_t lds = (size_t)(S + A + 8) * sizeof(float);
  if (lds > 150 * 1024) {
    set_error("crosscorr: %zu B of LDS needed", lds);
    return BSIG_EUNSUPPORTED;
  }
  if (S * A <= 2048 && (S + A) * 4 * sizeof(float) <= 32 * 1024) {
    const int64_t blocks = ceil_div<int64_t>(n, 4);
    hipLaunchKernelGGL(crosscorr_wave_kernel, dim3((int)std::min<int64_t>(blocks, 65536)),
                       dim3(256), (size_t)(S + A) * 4 * sizeof(float), st, states,
                       actions, out, n, t_states, t_actions, sd, ad, w, use_state_diff, ld_out,
                       nonfinite, fac, ld_fac);
    BSIG_CHECK_LAUNCH("crosscorr_wave");
    return BSIG_OK;
  }
  const int vec4 = out && (ld_out % 4 == 0) && aligned(out, 16);
  // the materialised summary alone, quads of action features: the lean streaming kernel
  if (out && !fac && vec4 && (A & 3) == 0 && sd - 1 <= 256 && ad <= 256 && w <= 10 &&
      getenv("BSIG_CC_GENERAL") == nullptr) {
    const int qpr = (int)(A >> 2);
    int m8 = 8; for (int g = qpr; (g & 1) == 0 && m8 > 1; g >>= 1) m8 >>= 1;   // 8 / gcd(qpr, 8)
    const int rstep = (256 / qpr) / m8 * m8;
    if (rstep >= 1 && S * qpr >= 512) {
      hipLaunchKernelGGL(crosscorr_quads_kernel, dim3(grid_for(n)), dim3(256), lds, st, states, actions,
                         out, n, t_states, t_actions, sd, ad, w, use_state_diff, ld_out, rstep, nonfinite);
      BSIG_CHECK_LAUNCH("crosscorr_quads");
      return BSIG_OK;
    }
  }
  hipLaunchKernelGGL(crosscorr_kernel, dim3(grid_for(n)), dim3(256), lds, st,
                     states, actions, out, n, t_states, t_actions, sd, ad, w,
                     use_state_diff, ld_out, vec4, nonfinite, fac, ld_fac);
  BSIG_CHECK_LAUNCH("crosscorr");
  return BSIG_OK;
}

// out[r, i*A + j] = sf[i] * af[j], out[r, S*A] = mean, out[r, S*A + 1] = std  from factor rows
__global__ __launch_bounds__(256) void crosscorr_expand_kernel(const float* __restrict__ fac,
                                                               int64_t ld_fac, int64_t n, int S,
                                                               int A, float* __restrict__ out,
                                                               int64_t ld_out) {
  const int total = S * A;
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    const float* f = fac + r * ld_fac;
    float* o = out + r * ld_out;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total + 2; e += gridDim.x * blockDim.x) {
      const int i = e / A, j = e - i * A;
      o[e] = e < total ? f[i] * f[S + j] : f[S + A + (e - total)];
    }
  }
}
}  // namespace bsig

extern "C" int bsig_crosscorr(const float* states, const float* actions, float* out,
                              int64_t n, int t_states, int t_actions, int sd, int ad,
                              int use_state_diff, int64_t ld_out, int32_t* nonfinite,
                              bsig_stream_t stream) {
  bsig::Range roctx_range("bsig_crosscorr");
  if (n == 0) return BSIG_OK;
  BSIG_REQUIRE(out, "crosscorr: null pointer");
  return crosscorr_launch(states, actions, out, ld_out, nullptr, 0, n, t_states, t_actions, sd, ad,
                          use_state_diff, nonfinite, as_stream(stream));
}

extern "C" int bsig_crosscorr_factor_dims(int traj_len, int sd, int ad, int32_t* s_out,
                                          int32_t* a_out) {
  BSIG_REQUIRE(traj_len > 1 && sd >= 2 && ad >= 1 && s_out && a_out, "crosscorr_factor_dims: bad args");
  const int w = crosscorr_window(traj_len, sd);
  *s_out = w * (sd - 1);
  *a_out = w * ad;
  return BSIG_OK;
}

extern "C" int bsig_crosscorr_factors(const float* states, const float* actions, float* factors,
                                      int64_t n, int t_states, int t_actions, int sd, int ad,
                                      int use_state_diff, int64_t ld_factors, int32_t* nonfinite,
                                      bsig_stream_t stream) {
  bsig::Range roctx_range("bsig_crosscorr_factors");
  if (n == 0) return BSIG_OK;
  BSIG_REQUIRE(factors, "crosscorr_factors: null pointer");
  return crosscorr_launch(states, actions, nullptr, 0, factors, ld_factors, n, t_states, t_actions,
                          sd, ad, use_state_diff, nonfinite, as_stream(stream));
}

extern "C" int bsig_crosscorr_expand(const float* factors, int64_t ld_factors, int64_t n, int s_dim,
                                     int a_dim, float* out, int64_t ld_out, bsig_stream_t stream) {
  if (n == 0) return BSIG_OK;
  BSIG_REQUIRE(factors && out && n >= 0 && s_dim >= 1 && a_dim >= 1, "crosscorr_expand: bad args");
  BSIG_REQUIRE(ld_factors >= s_dim + a_dim + 3 && ld_out >= (int64_t)s_dim * a_dim + 2,
               "crosscorr_expand: leading dims too small");
  const int total = s_dim * a_dim + 2;
  const dim3 grid(std::min(ceil_div(total, 256), 64), (unsigned)std::min<int64_t>(n, 16384));
  hipLaunchKernelGGL(crosscorr_expand_kernel, grid, dim3(256), 0, as_stream(stream), factors,
                     ld_factors, n, s_dim, a_dim, out, ld_out);
  BSIG_CHECK_LAUNCH("crosscorr_expand");
  return BSIG_OK;
}

extern "C" int bsig_signature(const float* states, const float* actions, float* out,
                              int64_t n, int length, int sd, int ad, int depth,
                              int64_t ld_out, bsig_stream_t stream) {
  bsig::Range roctx_range("bsig_signature");
  if (n == 0) return BSIG_OK;
  BSIG_REQUIRE(states && actions && out, "signature: null pointer");
  BSIG_REQUIRE(n >= 0 && length >= 2 && sd >= 1 && ad >= 1,
               "signature: bad dims length=%d sd=%d ad=%d", length, sd, ad);
  const int d = 1 + sd + ad;
  if (depth <= 0) depth = ref_signature_depth(d);
  BSIG_REQUIRE(depth >= 1 && depth <= 3, "signature: depth %d not in 1..3", depth);
  BSIG_REQUIRE(ld_out >= bsig_summary_dim(3, length, sd, ad, depth),
               "signature: ld_out too small");
  if (n == 0) return BSIG_OK;
  if (depth == 3) {
    if (d > 32) {
      set_error("signature: depth 3 needs path dim <= 32 (got %d)", d);
      return BSIG_EUNSUPPORTED;
    }
    const size_t lds = ((size_t)((length * d + 3) & ~3) + (size_t)(length - 1) * ((d + 3) & ~3) +
                        (size_t)d * d * d + 4) * sizeof(float);
    if (lds > 150 * 1024) {
      set_error("signature: %zu B of LDS needed", lds);
      return BSIG_EUNSUPPORTED;
    }
    const int threads = (int)round_up<int64_t>((int64_t)d * d, 64);
    if (d <= 8)
      hipLaunchKernelGGL((signature3_kernel<8>), dim3(grid_for(n)), dim3(threads), lds,
                         as_stream(stream), states, actions, out, n, length, sd, ad, ld_out);
    else if (d <= 16)
      hipLaunchKernelGGL((signature3_kernel<16>), dim3(grid_for(n)), dim3(threads), lds,
                         as_stream(stream), states, actions, out, n, length, sd, ad, ld_out);
    else if (d <= 24)
      hipLaunchKernelGGL((signature3_kernel<24>), dim3(grid_for(n)), dim3(threads), lds,
                         as_stream(stream), states, actions, out, n, length, sd, ad, ld_out);
    else
      hipLaunchKernelGGL((signature3_kernel<32>), dim3(grid_for(n)), dim3(threads), lds,
                         as_stream(stream), states, actions, out, n, length, sd, ad, ld_out);
  } else {
    size_t lds = 0;
    if (depth == 2) {
      lds = (size_t)length * d * sizeof(float);
      if (lds > 150 * 1024) {
        set_error("signature: %zu B of LDS needed", lds);
        return BSIG_EUNSUPPORTED;
      }
    }
    hipLaunchKernelGGL(signature12_kernel, dim3(grid_for(n)), dim3(256), lds,
                       as_stream(stream), states, actions, out, n, length, sd, ad, depth,
                       ld_out);
  }
  BSIG_CHECK_LAUNCH("signature");
  return BSIG_OK;
}
