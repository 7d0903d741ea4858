// Persistent update kernel of the two-layer MDNN whose FIRST LAYER DOES NOT FIT THE CHIP:
// the cross-correlation summaries of cfg/anymal.yaml:103-109 (I = 56 402) and
// cfg/shadow_hand_more.yaml:73-81 (I = 105 002, a 13.4 M-parameter first layer: 161 MB of
// weights + Adam moments) -- summarizers.py:112-119 into mdnn.py:71,108, updated by
// mdnn.py:219-233.  The row-owner and small-weight workgroups are those of the resident
// kernel (persist_mdnn_device.h); the tile workgroups here STREAM W1 and its moments:
//
//  * tile workgroup g of T (= the CUs the owners and the small-weight workgroups leave) walks
//    the 64-column chunks [g*C/T, (g+1)*C/T) of W1, all 128 rows.  The minibatch never
//    exists as a [B, I] tile: its FACTOR rows (sf | af | mean std, 1.3-4.6 KB per row
//    instead of 226-420 KB) sit in LDS -- row-major for the forward product's A operand,
//    transposed for the weight gradient's B operand -- and both MFMA operand streams form
//    x[b, i*A + j] = sf[b, i] * af[b, j] on the fly: the single fp32 multiply the summarizer
//    itself does.
//  * ONE pass over its chunks per update, taking Adam step t and the forward product of
//    minibatch t+1 from the same read of W1 (the minibatch ids are known ahead):
//        W, m, v (chunk) -> registers (fp32 MFMA accumulator layout)
//        dW  = dz1_t^T X_t[:, chunk]          (MFMA 32x32x2; dz1_t in registers)
//        Adam -> W', m', v' -> memory;  W' -> LDS
//        P_{t+1} += X_{t+1}[:, chunk] W'^T    (MFMA 32x32x2; accumulators live over the pass)
//    W1 crosses HBM 6 times per update (W, m, v in and out) instead of 7, the two products
//    run in the shadow of that stream, and no [B, I] summary, staging copy or gather exists.
//  * the T partial products [B, 128] are summed by the tile workgroups themselves (each owns
//    B*128/T consecutive elements: T*51 KB read once in total instead of once per row owner)
//    and reach the owners as one slab, bias included.
//
// Per update: pass -> slabs -> sum -> owners (h1 .. NLL .. dz1) -> pass.  Every sum is taken
// in a fixed order: runs are bitwise reproducible.  Data-parallel ranks (one update per
// launch): forward pass, owners, then a backward pass that writes dW to the flat gradient
// buffer; the Adam step follows the caller's all-reduce as a flat kernel.
#include "persist_mdnn_device.h"

namespace bsig {

constexpr int kSC = 64;               // columns of W1 per chunk
constexpr int kSPitch = kSC + 4;      // LDS pitch of the chunk's weights (forward B operand)
constexpr int kSZRows = 56;           // dz1 rows per staging pass (two passes: 7 + 6 groups of 8)
constexpr int kSZPitch = kMH + 4;
constexpr int kSRedGroups = 32;       // slab groups of the cross-workgroup sum
constexpr int kSTP = 108;             // LDS pitch of a transposed factor column (104 rows + 4)

// k / a for 0 <= k < 2^24 without an integer division
__device__ __forceinline__ int div_small(int k, int a, float ra) {
  int i = (int)((float)k * ra);
  if (i * a > k) --i;
  if ((i + 1) * a <= k) ++i;
  return i;
}

template <bool DP>
__device__ __forceinline__ void mdnn_stream_tile_workgroup(const MdnnArgs& p, float* smem) {
  const int PF = p.s_pf, NIP = p.s_nip, FR = p.FR, B = p.B;
  float* Fr = smem;                          // [FR][PF] factor rows of the minibatch of the forward product
  float* Ft = Fr + FR * PF;                  // [NIP + A + 8][kSTP] factor columns of the minibatch of dW
  float* Wc = Ft + (NIP + p.xA + 8) * kSTP;  // [128][kSPitch] this chunk's weights; also the
                                             // dz1 staging [56][132] and the partial sums [32][16][4]
  float* red = Wc + kMH * kSPitch;           // [64]
  float* b1s = red + 64;                     // [128] b1 as of the forward product being summed
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int g = blockIdx.x, T = p.G1;
  const int c_lo = (int)((int64_t)g * p.s_chunks / T), c_hi = (int)((int64_t)(g + 1) * p.s_chunks / T);
  const int S = p.xS, A = p.xA, SA = S * A, I = p.I;
  const float rA = __builtin_amdgcn_rcpf((float)A);
  const int i_lo = (c_lo * kSC) / A;
  int32_t* flagp = p.state + 2;
  const int step0 = p.state[0];
  double b1t = reinterpret_cast<const double*>(p.state + 12)[0];
  double b2t = reinterpret_cast<const double*>(p.state + 12)[1];
  float a0 = 0.f, a1 = 0.f;
  const AdamK ak{1.0f - (float)p.beta1, (float)p.beta2, 1.0f - (float)p.beta2, p.adam_eps};
  const bool fresh = !DP && step0 == 0;      // fresh optimizer (mdnn.py:203): moments start at zero
  // roles: weight-gradient tile (nb, kt) = 32 rows x 32 columns of the chunk;
  //        forward tiles (mt; 2 nbh, 2 nbh + 1) = 32 minibatch rows x 2 x 32 hidden units
  const int nb = w & 3, kt = w >> 2, mt = w & 3, nbh = w >> 2;

  // b1 (128 values): lanes 0-31 of wavefronts 0-3 own b1[32 w + l31] in registers; every tile
  // workgroup takes the same Adam steps (same values, same order), workgroup 0 writes back
  const bool bias_lane = w < 4 && lane < 32;
  float bw = 0.f, bm = 0.f, bv = 0.f;
  if (bias_lane) {
    const int64_t off = p.b1_off + 32 * w + lane;
    bw = p.params[off];
    if (!fresh && !DP) { bm = p.m1[off]; bv = p.m2[off]; }
    b1s[32 * w + lane] = bw;
  }

  // ---- factor rows of the minibatch starting at id-table row `row0`: slots [0, NIP) hold
  //      sf[i_lo ..] (i == S: the "1" beside mean / std), then af | mean | std | zeros.
  //      Row-major into Fr (transposed == false) or column-major into Ft
  auto load_factors = [&](bool transposed, int64_t row0) {
    const int per_row = NIP + A + 8;
    for (int b = w; b < (transposed ? kSTP : FR); b += kMT / 64) {
      const float* src = b < B ? p.x + (int64_t)p.ids[row0 + b] * p.ldx : nullptr;
      for (int c = lane; c < per_row; c += 64) {
        float v = 0.f;
        if (src) {
          if (c < NIP) {
            const int i = i_lo + c;
            v = i < S ? src[i] : (i == S ? 1.0f : 0.f);
          } else {
            const int j = c - NIP;
            v = j < A + 2 ? src[S + j] : 0.f;
          }
        }
        if (transposed) Ft[c * kSTP + b] = v;
        else Fr[b * PF + c] = v;
      }
    }
  };

  // ---- the chunk's tile in the accumulator layout ------------------------------------------
  // (buffer addressing: one lane offset per chunk, the 16 row offsets of the accumulator layout
  // are wavefront-uniform -- no per-element 64-bit addresses kept in registers)
  float Wv[16], Mv[16], Vv[16];
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(p.params + p.w1_off, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rM = __builtin_amdgcn_make_buffer_rsrc(p.m1 + p.w1_off, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(p.m2 + p.w1_off, 0, 0x7fffffff, 0x00020000);
  const int lane_off = ((nb * 32 + 4 * h) * I + kt * 32 + l31) * 4;     // bytes; + chunk column, + row(i)
  auto bld = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
  };
  auto bst = [](__amdgpu_buffer_rsrc_t r, int voff, int soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
  };
  auto load_chunk = [&](int c, bool moments) {
    const bool ok = c * kSC + kt * 32 + l31 < I;
    const int voff = lane_off + c * kSC * 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int soff = acc_row0(i) * I * 4;
      Wv[i] = 0.f; Mv[i] = 0.f; Vv[i] = 0.f;
      if (ok) {
        Wv[i] = bld(rW, voff, soff);
        if (moments) { Mv[i] = bld(rM, voff, soff); Vv[i] = bld(rV, voff, soff); }
      }
    }
  };

  f32x16 facc[2];
  // forward product of one chunk: rows mt, hidden blocks 2 nbh / 2 nbh + 1, weights in Wc
  auto forward_chunk = [&](int c) {
    const float* frow = Fr + min(mt * 32 + l31, FR - 1) * PF;
    const float* wrow = Wc + (2 * nbh * 32 + l31) * kSPitch + 4 * h;
#pragma unroll 2
    for (int kk = 0; kk < kSC; kk += 8) {
      const int k4 = c * kSC + kk + 4 * h;
      int si, ai;
      if (k4 >= SA) { si = S - i_lo; ai = A + min(k4 - SA, 4); }
      else { const int i = div_small(k4, A, rA); si = i - i_lo; ai = k4 - i * A; }
      const float sfv = frow[si];
      const float4 af4 = *reinterpret_cast<const float4*>(frow + NIP + ai);
      const float ax = sfv * af4.x, ay = sfv * af4.y, az = sfv * af4.z, aw = sfv * af4.w;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(wrow + q * 32 * kSPitch + kk);
        facc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, b4.x, facc[q], 0, 0, 0);
        facc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, b4.y, facc[q], 0, 0, 0);
        facc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(az, b4.z, facc[q], 0, 0, 0);
        facc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, b4.w, facc[q], 0, 0, 0);
      }
    }
  };
  auto chunk_to_lds = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) Wc[(nb * 32 + acc_row(i, h)) * kSPitch + kt * 32 + l31] = Wv[i];
  };

  // ---- partial products of this workgroup -> slab; the sum over the workgroups ---------------
  auto publish_and_sum = [&](unsigned epoch) {
    {
      float* dst = fresh_ptr(p.slabs + ((int64_t)g * B + mt * 32 + 4 * h) * kMH + 2 * nbh * 32 + l31);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = mt * 32 + acc_row(i, h);
          if (row < B) xwg_store(dst + acc_row0(i) * kMH + q * 32, facc[q][i]);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) flag_raise(p.flag_fwd, g, epoch);
    if (w == 0) flags_wait(p.flag_fwd, T, epoch, lane, flagp);
    __syncthreads();
    // my share of the B*32 quads of the [B, 128] block; 32 groups of slabs per quad, partial sums
    // combined in group order
    const int nq = B * (kMH / 4);
    const int q_lo = (int)((int64_t)g * nq / T), q_hi = (int)((int64_t)(g + 1) * nq / T);
    const __amdgpu_buffer_rsrc_t sr = xwg_buffer(p.slabs);
    const int zs = B * kMH;
    float* part = Wc;
    for (int qb = q_lo; qb < q_hi; qb += 16) {
      const int qi = tid & 15, sg = tid >> 4;
      const int q = min(qb + qi, q_hi - 1);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int z = sg; z < T; z += kSRedGroups * 8) {
        f32x4 ld[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ld[u] = xwg_load4(sr, min(z + u * kSRedGroups, T - 1) * zs + q * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (z + u * kSRedGroups < T) v += ld[u];
      }
      *reinterpret_cast<f32x4*>(part + (sg * 16 + qi) * 4) = v;
      __syncthreads();
      if (tid < 16 && qb + tid < q_hi) {
        f32x4 s = *reinterpret_cast<const f32x4*>(part + tid * 4);
        for (int u = 1; u < kSRedGroups; ++u) s += *reinterpret_cast<const f32x4*>(part + (u * 16 + tid) * 4);
        const int n = ((qb + tid) * 4) & (kMH - 1);
        xwg_store4(xwg_buffer(p.hpre), (qb + tid) * 4, s.x + b1s[n], s.y + b1s[n + 1], s.z + b1s[n + 2],
                   s.w + b1s[n + 3]);
      }
      __syncthreads();
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) flag_raise(p.flag_red, g, epoch);
  };

  // ---- prologue: the forward product of the launch's first minibatch --------------------------
  if (p.n_updates > 0) {
    load_factors(false, (int64_t)step0 * B);
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) facc[q][i] = 0.f;
    if (c_lo < c_hi) load_chunk(c_lo, false);
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
      chunk_to_lds();
      if (c + 1 < c_hi) load_chunk(c + 1, false);
      __syncthreads();
      forward_chunk(c);
      __syncthreads();
    }
    publish_and_sum((unsigned)step0 + 1u);
  }

  for (int t = 0; t < p.n_updates; ++t) {
    const int step = step0 + t;
    const unsigned epoch = (unsigned)step + 1u;
    const bool has_next = !DP && t + 1 < p.n_updates;
    if (run_aborted(flagp, red, tid)) break;
    BSIG_MSTAMP(0);
    // ---- while the owners work: this minibatch's factor columns (dW), the next minibatch's
    //      factor rows (forward), the first chunk, Adam scalars
    load_factors(true, (int64_t)step * B);
    if (has_next) load_factors(false, (int64_t)(step + 1) * B);
    if (!DP && c_lo < c_hi) load_chunk(c_lo, !(fresh && t == 0));
    b1t *= p.beta1; b2t *= p.beta2;
    a0 = (float)(p.lr / (1.0 - b1t));
    a1 = (float)(1.0 / sqrt(1.0 - b2t));
    BSIG_MSTAMP(1);
    if (w == 0) flags_wait(p.flag_own, p.n_owner, epoch, lane, flagp);
    __syncthreads();
    BSIG_MSTAMP(2);
    // ---- dz1 [B, 128] -> the A operand registers of this wavefront's 32 hidden units -------
    float za[52];
    {
      const __amdgpu_buffer_rsrc_t zr = xwg_buffer(p.dz1);
      float* Zs = Wc;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int b0 = half * kSZRows, nrow = half == 0 ? kSZRows : FR - kSZRows;
        for (int base = 0; base < nrow * 32; base += kMT * 4) {
          f32x4 q[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = base + u * kMT + tid;
            const int b = b0 + (idx >> 5);
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            q[u] = (idx < nrow * 32 && b < B) ? xwg_load4(zr, b * kMH + (idx & 31) * 4) : zero;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = base + u * kMT + tid;
            if (idx < nrow * 32) *reinterpret_cast<f32x4*>(Zs + (idx >> 5) * kSZPitch + (idx & 31) * 4) = q[u];
          }
        }
        __syncthreads();
#pragma unroll
        for (int gq = 0; gq < 13; ++gq) {
          if ((gq < 7) == (half == 0)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int b = 8 * gq + 4 * h + e;
              za[4 * gq + e] = b < FR ? Zs[(b - b0) * kSZPitch + nb * 32 + l31] : 0.f;
            }
          }
        }
        __syncthreads();
      }
    }
    // ---- b1: column sums of dz1 (fixed order: lane half 0 then 1), Adam -------------------------
    {
      float gs = 0.f;
#pragma unroll
      for (int q = 0; q < 52; ++q) gs += za[q];
      gs += __shfl_xor(gs, 32, 64);
      if (bias_lane) {
        if (DP) { if (g == 0) p.grads[p.b1_off + 32 * w + lane] = gs; }
        else bw = adam_bias(gs, bm, bv, bw, a0, a1, ak);
      }
    }
    __syncthreads();                         // (b1s readers of the previous sum are done)
    if (bias_lane) b1s[32 * w + lane] = bw;
    BSIG_MSTAMP(3);

    // ---- the pass --------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int i = 0; i < 16; ++i) facc[q][i] = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
      const int k = c * kSC + kt * 32 + l31;
      // dW tile = dz1^T X_t[:, chunk]: lane <-> column k = i*A + j, its two factors down the rows
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      {
        int si, ai;
        if (k >= SA) { si = S - i_lo; ai = A + min(k - SA, 2); }
        else { const int i = div_small(k, A, rA); si = i - i_lo; ai = k - i * A; }
        const float* sfp = Ft + si * kSTP + 4 * h;
        const float* afp = Ft + (NIP + ai) * kSTP + 4 * h;
#pragma unroll
        for (int gq = 0; gq < 13; ++gq) {
          if (8 * gq < FR) {
            const float4 s4 = *reinterpret_cast<const float4*>(sfp + 8 * gq);
            const float4 f4 = *reinterpret_cast<const float4*>(afp + 8 * gq);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq + 0], s4.x * f4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq + 1], s4.y * f4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq + 2], s4.z * f4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq + 3], s4.w * f4.w, acc, 0, 0, 0);
          }
        }
      }
      const int voff = lane_off + c * kSC * 4;
      if (DP) {
        if (k < I) {
          const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc(p.grads + p.w1_off, 0, 0x7fffffff, 0x00020000);
#pragma unroll
          for (int i = 0; i < 16; ++i) bst(rG, voff, acc_row0(i) * I * 4, acc[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) Wv[i] = adam_weight(acc[i], Mv[i], Vv[i], Wv[i], a0, a1, ak);
        if (k < I) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int soff = acc_row0(i) * I * 4;
            bst(rW, voff, soff, Wv[i]); bst(rM, voff, soff, Mv[i]); bst(rV, voff, soff, Vv[i]);
          }
        }
        if (has_next) chunk_to_lds();
        if (c + 1 < c_hi) load_chunk(c + 1, !(fresh && t == 0));
        if (has_next) {
          __syncthreads();
          forward_chunk(c);
          __syncthreads();
        }
      }
    }
    BSIG_MSTAMP(4);
    if (has_next) publish_and_sum(epoch + 1u);
    BSIG_MSTAMP(5);
  }

  // ---- write b1 back, advance the engine state -------------------------------------------------
  if (!DP && g == 0 && bias_lane) {
    const int64_t off = p.b1_off + 32 * w + lane;
    p.params[off] = bw; p.m1[off] = bm; p.m2[off] = bv;
  }
  if (g == 0 && tid == 0 && p.n_updates > 0) {
    int32_t* st = p.state;
    reinterpret_cast<double*>(st + 12)[0] = b1t;
    reinterpret_cast<double*>(st + 12)[1] = b2t;
    reinterpret_cast<float*>(st)[4] = a0;
    reinterpret_cast<float*>(st)[5] = a1;
    reinterpret_cast<uint64_t*>(st + 8)[1] += (uint64_t)p.n_updates;   // one jitter stream per update
    st[0] = step0 + p.n_updates;
  }
}

// DP: data-parallel rank; WIDE / FULL: as in the resident kernel
template <bool DP, bool WIDE, bool FULL>
__global__ __launch_bounds__(kMT) void mdnn_stream_updates_kernel(MdnnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int wg = blockIdx.x;
  if (wg < p.G1) mdnn_stream_tile_workgroup<DP>(p, smem);
#ifndef BSIG_STREAM_TILE_ONLY   // (resource usage of the tile body alone)
  else if (wg < p.G1 + p.n_owner) mdnn_owner_workgroup<DP, WIDE, FULL>(p, smem);
  else mdnn_small_workgroup<DP, WIDE>(p, smem);
#endif
}

// ---------------------------------------------------------------- host side
// LDS floats of a tile workgroup; false when the factor rows of two minibatches do not fit
bool mdnn_stream_tile_geom(int FR, int chunks_per_wg, int S, int A, int* nip, int* pf, size_t* lds_bytes) {
  if (S < 1 || A < 4 || A % 4 != 0) return false;
  const int cols = chunks_per_wg * kSC;
  const int ni = (cols - 1) / A + 2;            // distinct i = k / A over `cols` columns (i = S: the tail)
  *nip = (int)round_up(ni, 4);
  int pitch = *nip + A + 8;
  while (pitch % 8 != 4) pitch += 4;            // 16-byte reads down the rows: conflict-free
  *pf = pitch;
  const size_t floats = (size_t)FR * pitch + (size_t)(*nip + A + 8) * kSTP + (size_t)kMH * kSPitch + 64 + kMH;
  *lds_bytes = floats * sizeof(float);
  return *lds_bytes <= (size_t)kMLdsLimit && (size_t)kSZRows * kSZPitch <= (size_t)kMH * kSPitch &&
         FR - kSZRows <= kSZRows && FR <= 104;
}

int mdnn_stream_launch(const MdnnArgs& p, bool dp, bool wide, bool full, int grid, size_t lds, hipStream_t st) {
  static bool attr_set_dev[64] = {};
  int dev = 0;
  BSIG_HIP(hipGetDevice(&dev));
  bool& attr_set = attr_set_dev[dev & 63];
  if (!attr_set) {
    const void* kernels[8] = {
#define BSIG_K(a, b, c) reinterpret_cast<const void*>(mdnn_stream_updates_kernel<a, b, c>)
        BSIG_K(false, false, false), BSIG_K(true, false, false), BSIG_K(false, true, false), BSIG_K(true, true, false),
        BSIG_K(false, false, true),  BSIG_K(true, false, true),  BSIG_K(false, true, true),  BSIG_K(true, true, true)};
#undef BSIG_K
    for (const void* k : kernels)
      BSIG_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kMLdsLimit));
    attr_set = true;
  }
#define BSIG_L(a, b, c) hipLaunchKernelGGL((mdnn_stream_updates_kernel<a, b, c>), dim3(grid), dim3(kMT), lds, st, p)
  if (dp) {
    if (wide && full) BSIG_L(true, true, true); else if (wide) BSIG_L(true, true, false);
    else if (full) BSIG_L(true, false, true); else BSIG_L(true, false, false);
  } else {
    if (wide && full) BSIG_L(false, true, true); else if (wide) BSIG_L(false, true, false);
    else if (full) BSIG_L(false, false, true); else BSIG_L(false, false, false);
  }
#undef BSIG_L
  BSIG_CHECK_LAUNCH("mdnn_stream_updates");
  return BSIG_OK;
}

}  // namespace bsig
