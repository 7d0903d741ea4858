// Persistent update kernel: a run of consecutive SGD updates of a linear
// mixture-density head on precomputed features in ONE launch
// (MDRFF.run_training's inner loop, mdnn.py:219-233, with the RFF projection
// hoisted: forward mdnn.py:108-119, NLL :127-178, its backward, Adam :203/:229).
//
// At the reference's minibatch of 100 rows an update is a chain of five tiny
// dependent kernels (head GEMM, split-K reduce, NLL, finish, dW+Adam), each
// bounded by launch + first-load latency, and the weights and both Adam
// moments stream from HBM every update.  Here the head matrix W [Nh, F] is
// tiled over the chip once: workgroup (nb, ks) owns the 32 x 256 tile
// W[32nb.., 256ks..] and keeps it, with its Adam moments, in REGISTERS (in the
// 32x32 MFMA accumulator layout) for the whole run.  Per update:
//   1. every workgroup loads its [B, 256] slice of the minibatch features into
//      LDS, multiplies it with its W tile (fp32 MFMA 32x32x2, operands from
//      LDS) and writes the [B, 32] partial product (a split-K slab) through to
//      memory, then raises its flag;
//   2. the first n_owner workgroups each own R minibatch rows: they wait for
//      every flag, sum the k-slices of their rows, and run the row-wise
//      NLL forward/backward (diag_row, one wavefront per row); the three
//      batch-wide sums (jitter scale, its gradient term, the loss) cross
//      workgroups as {tag, value} granules; d_out rows are written through
//      and the owner raises its flag;
//   3. every workgroup waits for the owners, loads its [B, 32] block of d_out
//      (transposed into LDS), forms dW = d_out^T F on the MFMA units and
//      applies Adam to its register tile; the k-slice-0 workgroups also own
//      the 32 biases of their block.
// The write-through / cache-bypassing accesses (agent-scope relaxed atomics)
// are what makes data cross the per-XCD L2s inside a launch without fences
// (tools/micro/grid_barrier_bench.hip: an agent-scope fence costs ~0.13 us per
// workgroup here, the flag hop ~3 us).  All polls are bounded; a time-out
// raises bit 1 of the state block's nonfinite word.
#include "persist.h"

#include <algorithm>

#include "head_device.h"
#include "persist_device.h"

namespace bsig {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPT = 512;            // threads per workgroup (8 wavefronts)
constexpr int kPC = 256;            // feature columns per workgroup
constexpr int kPitch = kPC + 4;     // LDS row pitch of the feature / weight tiles
constexpr int kNB = 32;             // head rows per workgroup
constexpr int kPbufPitch = 33;
constexpr int kLdsLimit = 160 * 1024;

struct PersistArgs {
  int B, FR, Fdim, Nh, NhP, D, K;
  int n_blocks, k_slices, G, n_owner, R;
  int n_updates, x_floats;
  const float* feats; int64_t ld_feats; const int32_t* feat_ids;
  const float* y; int64_t ldy; const int32_t* ids;
  float* params; float* m1; float* m2; int64_t w_off, b_off;
  int32_t* state; float* train_loss;
  double lr, beta1, beta2;
  float adam_eps, eps_noise, min_w, ll_limit, inv_norm;
  float* slabs; float* d_out; float* e_out; unsigned* flag_fwd; unsigned long long* gran;
  // data-parallel ranks (one update per launch, the caller all-reduces `grads`
  // between launches): dW / bias gradients go to `grads` instead of Adam, and the
  // Adam step of the PREVIOUS update (on the reduced gradients) is taken while the
  // tile is loaded (adam_pending)
  float* grads; int adam_pending;
  int quad_ok;                       // weight rows are 16-byte aligned quads (w_off, Fdim multiples of 4)
  // held-out evaluations inside the launch (mdnn.py:235-242; do_eval): after update `it`
  // with it % eval_every == 0, or after the last of the call's n_total updates.  The tile
  // workgroups form the held-out rows' products while they wait for the row owners of the
  // NEXT update (their LDS still holds the evaluated weights), the owners evaluate their
  // rows after they have published that update's rows.
  int do_eval, eval_every, n_total, n_test, eval_passes;
  int64_t eval_row0;                 // first held-out row in `feats`
  const float* y_test; int64_t ldy_test;
  float* test_loss;                  // [n_evals]
  float* eval_slabs;                 // [2][eval_passes][k_slices][B][NhP]
  unsigned* flag_eval;               // [G]      evaluation number + 1
  unsigned long long* gran_eval;     // [2][kXwgMax] {tag, value}: sum exp(pre), sum logsumexp
  long long* prof;   // diagnostics: [G][kProfUpdates][16] wall-clock stamps, or null
};

// The 32-column tile of summed partial products a tile workgroup holds in X (pitch kPbufPitch) to
// its slab rows: every lane stores 16 bytes (8 lanes per 128-byte row segment), all wavefronts.
__device__ __forceinline__ void slab_tile_store(const float* X, float* dst, int rows, int ld, int tid) {
  const __amdgpu_buffer_rsrc_t sr = xwg_buffer(dst);
  for (int idx = tid; idx < rows * (kNB / 4); idx += kPT) {
    const int row = idx >> 3, c4 = (idx & 7) * 4;
    const float* x = X + row * kPbufPitch + c4;
    xwg_store4(sr, row * ld + c4, x[0], x[1], x[2], x[3]);
  }
}

// A row owner's sum over the k-slices of its rows' partial products: slabs [k_slices][B][ld], rows
// r0.., `nrows` of them -> out[r * out_pitch + col] (col < n_cols), slices added in order, 16 loads
// in flight per lane.  One dword per lane: measured faster here than 16-byte loads on a quarter of
// the lanes (1.4 us against 1.6 us for 16 slices of 260 columns) -- the loads are latency-bound and
// more wavefronts issue them.  `fn(col, v)` sees every finished element.
template <typename F>
__device__ __forceinline__ void slab_rows_sum(const float* slabs, int k_slices, int B, int ld, int nrows,
                                              int n_cols, float* out, int out_pitch, int tid, F&& fn) {
  const int64_t zs = (int64_t)B * ld;
  for (int idx = tid; idx < nrows * n_cols; idx += kPT) {
    const int r = idx / n_cols, col = idx - r * n_cols;
    const float* src = slabs + (int64_t)r * ld + col;
    float v = 0.f;
    for (int z = 0; z < k_slices; z += 16) {
      float q[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) q[u] = xwg_load(src + (int64_t)min(z + u, k_slices - 1) * zs);
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (z + u < k_slices) v += q[u];
    }
    out[r * out_pitch + col] = v;
    fn(col, v);
  }
}

// number of evaluation points it % every == 0 strictly before update s (the evaluation after
// the last update of a call is not one of them)
__device__ __forceinline__ int evals_before(int s, int every) { return s == 0 ? 0 : (s - 1) / every + 1; }

constexpr int kProfUpdates = 8;
#define BSIG_STAMP(k)                                                              \
  do {                                                                             \
    if (p.prof && threadIdx.x == 0 && t < kProfUpdates)                                    \
      p.prof[((int64_t)wg * kProfUpdates + t) * 16 + (k)] = wall_clock64();        \
  } while (0)

// one [<=104, 256] feature tile = 13 float4 per thread, kept in named registers
// between the prefetch and the LDS write (an indexed array lands in scratch)
#define BSIG_PF_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#define BSIG_PF_DECL(u) float4 pf##u;
#define BSIG_PF_LOAD(u)                                                                   \
  {                                                                                       \
    const int idx = min((u) * kPT + tid, nvec - 1);                                       \
    const int64_t r = pf_row0 + (idx >> 6);                                               \
    const int64_t fr = p.feat_ids ? (int64_t)p.feat_ids[r] : r;                           \
    const int64_t col = min((int64_t)k0 + (idx & 63) * 4, p.ld_feats - 4);                \
    pf##u = *reinterpret_cast<const float4*>(p.feats + fr * p.ld_feats + col);            \
  }
#define BSIG_PF_ZERO(u) pf##u = make_float4(0.f, 0.f, 0.f, 0.f);
// (columns >= F, the tail of the last k-slice when F is not a multiple of 256, enter as zeros)
#define BSIG_PF_STORE(u)                                                                  \
  {                                                                                       \
    const int idx = (u) * kPT + tid;                                                      \
    if (idx < nvec) {                                                                     \
      const int col = k0 + (idx & 63) * 4;                                                \
      float4 v = pf##u;                                                                   \
      if (col + 3 >= p.Fdim) {                                                            \
        v.x = col + 0 < p.Fdim ? v.x : 0.f; v.y = col + 1 < p.Fdim ? v.y : 0.f;           \
        v.z = col + 2 < p.Fdim ? v.z : 0.f; v.w = col + 3 < p.Fdim ? v.w : 0.f;           \
      }                                                                                   \
      *reinterpret_cast<float4*>(Fl + (idx >> 6) * kPitch + (idx & 63) * 4) = v;          \
    }                                                                                     \
  }

// ---- tile workgroups, evaluation number eidx: held-out rows x this tile's weights (the A
//      operand straight from memory: the minibatch tile in LDS is still needed for dW) ->
//      evaluation slabs, flag.  Deliberately NOT inlined: it runs a few times per call, and
//      its registers must not weigh on the update loop.
__device__ __forceinline__ void tile_eval(const PersistArgs& p, const float* Wl, float* X,
                                                    const float* biasl, int eidx) {
  // (laundered: nothing below may be computed ahead of the update loop and kept live in it)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wg = blockIdx.x, ks = wg % p.k_slices, nb = wg / p.k_slices;
  const int n0 = nb * kNB, k0 = ks * kPC;
  const int B = p.B, NhP = p.NhP;
  const int mt = w & 3, kh = w >> 2;
  for (int pass = 0; pass < p.eval_passes; ++pass) {
    const int rows = min(B, p.n_test - pass * B);
    if (rows <= 0) break;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int64_t r = p.eval_row0 + pass * B + min(mt * 32 + l31, rows - 1);
    const float* src = p.feats + r * p.ld_feats;
    const float* bp = Wl + l31 * kPitch + kh * 128 + 4 * h;
    // two halves of the k range: 8 x 16 bytes of the A operand in registers at a time
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      float4 areg[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int64_t col = min((int64_t)k0 + kh * 128 + 64 * half + 8 * q + 4 * h, p.ld_feats - 4);
        areg[q] = *reinterpret_cast<const float4*>(src + col);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4 b4 = *reinterpret_cast<const float4*>(bp + 64 * half + 8 * q);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[q].x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[q].y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[q].z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[q].w, b4.w, acc, 0, 0, 0);
      }
    }
    __syncthreads();                       // X free (previous pass read)
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) X[(mt * 32 + acc_row(i, h)) * kPbufPitch + l31] = acc[i];
    }
    __syncthreads();
    if (kh == 0) {
      const float bias = ks == 0 ? biasl[l31] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float* x = X + (mt * 32 + acc_row(i, h)) * kPbufPitch + l31;
        *x = acc[i] + *x + bias;
      }
    }
    __syncthreads();
    slab_tile_store(X, p.eval_slabs + ((((int64_t)(eidx & 1) * p.eval_passes + pass) * p.k_slices + ks) * B) * NhP + n0,
                    rows, NhP, tid);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0)
    flag_raise(p.flag_eval, wg, (unsigned)eidx + 1u);
}

// ---- tile workgroups: forward partial products, dW, Adam ----------------------
template <bool DP>
__device__ __forceinline__ void tile_workgroup(const PersistArgs& p, float* smem) {
  if (p.prof && p.n_updates > 0 && threadIdx.x == 0) p.prof[((int64_t)blockIdx.x * kProfUpdates) * 16 + 14] = wall_clock64();
  float* Fl = smem;                          // [FR][kPitch] minibatch features (this k-slice)
  float* Wl = Fl + p.FR * kPitch;            // [32][kPitch] weight tile (authoritative copy)
  float* X = Wl + kNB * kPitch;              // scratch: forward k-halves | d_out^T
  float* red = X + p.x_floats;               // [64]
  float* biasl = red + 64;                   // [3][32] bias, exp_avg, exp_avg_sq (k-slice 0)
  float* bpart = biasl + 96;                 // [16][32] partial column sums of d_out (k-slice 0)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wg = blockIdx.x, ks = wg % p.k_slices, nb = wg / p.k_slices;
  const int n0 = nb * kNB, k0 = ks * kPC;
  const int B = p.B, Nh = p.Nh, NhP = p.NhP;
  int32_t* flagp = p.state + 2;
  const int step0 = p.state[0];
  double b1t = reinterpret_cast<const double*>(p.state + 12)[0];
  double b2t = reinterpret_cast<const double*>(p.state + 12)[1];
  float a0 = 0.f, a1 = 0.f;
  const AdamK ak{1.0f - (float)p.beta1, (float)p.beta2, 1.0f - (float)p.beta2, p.adam_eps};

  // resident Adam moments in the dW accumulator layout: element i of lane
  // (h, l31) of wave w  <->  W[n0 + acc_row(i, h)][k0 + 32w + l31]; W itself
  // lives in LDS (it is the B operand of the forward product)
  float Mr[16], Vr[16];
  const int kcol = 32 * w + l31;
  const bool col_ok = k0 + kcol < p.Fdim;
  constexpr bool dp = DP;
  const bool pend = DP && p.adam_pending != 0;
  // first launch of a run_training call: a fresh optimizer (mdnn.py:203) -- the moments start
  // at zero in the registers, nobody has to clear (or read) the 2 x 4.3 MB in memory
  const bool fresh = !DP && step0 == 0;
  // data-parallel: Adam scalars of the update whose reduced gradients are pending
  const float pa0 = pend ? reinterpret_cast<const float*>(p.state)[4] : 0.f;
  const float pa1 = pend ? reinterpret_cast<const float*>(p.state)[5] : 0.f;
  // (three passes: every load is issued before the first store -- the stores of a
  // data-parallel launch's pending Adam step may alias the loads for the compiler, and
  // interleaved they serialise into 16 dependent round trips: 10 us instead of 3)
  if (DP && p.quad_ok) {
    // A data-parallel launch only passes through the tile here (pending Adam step, then the weights
    // into LDS; its dW goes to the gradient buffer), and Adam is elementwise: the tile is taken as
    // rows of 16-byte quads -- one 1 KB row of W / exp_avg / exp_avg_sq / gradients per wavefront
    // instruction, all 16 loads of a thread in flight -- instead of 64 dword loads per lane in the
    // accumulator layout.
    f32x4 Wq[4], Mq[4], Vq[4], Gq4[4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + q * 8 + w;
      Wq[q] = zero; Mq[q] = zero; Vq[q] = zero; Gq4[q] = zero;
      if (n < Nh && k0 + 4 * lane < p.Fdim) {
        const int64_t off = p.w_off + (int64_t)n * p.Fdim + k0 + 4 * lane;
        Wq[q] = *reinterpret_cast<const f32x4*>(p.params + off);
        if (pend) {
          Mq[q] = *reinterpret_cast<const f32x4*>(p.m1 + off);
          Vq[q] = *reinterpret_cast<const f32x4*>(p.m2 + off);
          Gq4[q] = *reinterpret_cast<const f32x4*>(p.grads + off);
        }
      }
    }
    if (pend) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float m = Mq[q][j], v = Vq[q][j];
          Wq[q][j] = adam_weight(Gq4[q][j], m, v, Wq[q][j], pa0, pa1, ak);
          Mq[q][j] = m; Vq[q][j] = v;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + q * 8 + w;
        if (n < Nh && k0 + 4 * lane < p.Fdim) {
          const int64_t off = p.w_off + (int64_t)n * p.Fdim + k0 + 4 * lane;
          *reinterpret_cast<f32x4*>(p.params + off) = Wq[q];
          *reinterpret_cast<f32x4*>(p.m1 + off) = Mq[q];
          *reinterpret_cast<f32x4*>(p.m2 + off) = Vq[q];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(Wl + (q * 8 + w) * kPitch + 4 * lane) = Wq[q];
#pragma unroll
    for (int i = 0; i < 16; ++i) { Mr[i] = 0.f; Vr[i] = 0.f; }
  } else {
    float Wv[16], Gq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + acc_row(i, h);
      Wv[i] = 0.f; Gq[i] = 0.f; Mr[i] = 0.f; Vr[i] = 0.f;
      if (n < Nh && col_ok) {
        const int64_t off = p.w_off + (int64_t)n * p.Fdim + k0 + kcol;
        Wv[i] = p.params[off];
        if (!fresh) { Mr[i] = p.m1[off]; Vr[i] = p.m2[off]; }
        if (pend) Gq[i] = p.grads[off];
      }
    }
    if (pend) {
      // (a data-parallel launch changes the tile only here: written back at once,
      // the stores drain under the forward product)
#pragma unroll
      for (int i = 0; i < 16; ++i) Wv[i] = adam_weight(Gq[i], Mr[i], Vr[i], Wv[i], pa0, pa1, ak);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + acc_row(i, h);
        if (n < Nh && col_ok) {
          const int64_t off = p.w_off + (int64_t)n * p.Fdim + k0 + kcol;
          p.params[off] = Wv[i]; p.m1[off] = Mr[i]; p.m2[off] = Vr[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Wl[acc_row(i, h) * kPitch + kcol] = Wv[i];
  }
  if (ks == 0 && tid < kNB) {
    const int n = n0 + tid;
    float bw = n < Nh ? p.params[p.b_off + n] : 0.f;
    float bm = n < Nh && !fresh ? p.m1[p.b_off + n] : 0.f;
    float bv = n < Nh && !fresh ? p.m2[p.b_off + n] : 0.f;
    if (pend && n < Nh) {
      bw = adam_bias(p.grads[p.b_off + n], bm, bv, bw, pa0, pa1, ak);
      p.params[p.b_off + n] = bw; p.m1[p.b_off + n] = bm; p.m2[p.b_off + n] = bv;
    }
    biasl[tid] = bw; biasl[32 + tid] = bm; biasl[64 + tid] = bv;
  }
  for (int idx = tid; idx < (p.FR - B) * kPitch; idx += kPT) Fl[B * kPitch + idx] = 0.f;
  const int DOP = p.FR + 4;
  const int nvec = B * (kPC / 4);
  // The bias step of an update (k-slice 0) is finished off the critical path: by
  // the otherwise idle lanes of wave 7 during the next forward product, before
  // the bias is read again (or after the last update).
  bool bias_pending = false;
  auto bias_step = [&](int n) {
    float g = 0.f;
#pragma unroll
    for (int q = 0; q < kPT / 32; ++q) g += bpart[q * 32 + n];
    float bm = biasl[32 + n], bv = biasl[64 + n];
    biasl[n] = adam_bias(g, bm, bv, biasl[n], a0, a1, ak);
    biasl[32 + n] = bm;
    biasl[64 + n] = bv;
  };

  // the sticky time-out bit of an EARLIER launch of this call (a data-parallel rank makes one launch per
  // update): sampled at entry, so that such a launch leaves at once instead of running its forward
  // product into the bounded polls of owners that have already left
  if (tid == 0)
    red[63] = (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) ? 1.f : 0.f;
  // feature tile of the first update (later ones are fetched during the waits)
  BSIG_PF_LIST(BSIG_PF_DECL)
  if (p.n_updates > 0) {
    const int64_t pf_row0 = (int64_t)step0 * B;
    BSIG_PF_LIST(BSIG_PF_LOAD)
  } else {
    BSIG_PF_LIST(BSIG_PF_ZERO)
  }
  __syncthreads();

  for (int t = 0; t < p.n_updates; ++t) {
    // lane-derived indices are laundered once per update so that the address
    // arithmetic built on them is recomputed, not kept live across the waits
    int tid_l = tid, l31_l = l31, h_l = h, kcol_l = kcol;
    asm volatile("" : "+v"(tid_l), "+v"(l31_l), "+v"(h_l), "+v"(kcol_l));
    const int step = step0 + t;
    const unsigned epoch = (unsigned)step + 1u;
    if (red[63] != 0.f) break;   // time-out bit as sampled during the previous update's wait
    BSIG_STAMP(0);
    // ---- 1. feature tile -> LDS ---------------------------------------------
    BSIG_PF_LIST(BSIG_PF_STORE)
    __syncthreads();
    BSIG_STAMP(1);

    // ---- 2. partial forward: P[b, n] = sum_{k in slice} F[b, k] W[n, k] -------
    {
      const int mt = w & 3, kh = w >> 2;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = Fl + (mt * 32 + l31_l) * kPitch + kh * 128 + 4 * h_l;
      const float* bp = Wl + l31_l * kPitch + kh * 128 + 4 * h_l;
#pragma unroll 4
      for (int kk = 0; kk < 128; kk += 8) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + kk);
        const float4 b4 = *reinterpret_cast<const float4*>(bp + kk);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
      }
      BSIG_STAMP(2);
      if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) X[(mt * 32 + acc_row(i, h_l)) * kPbufPitch + l31_l] = acc[i];
      }
      if (bias_pending && w == kPT / 64 - 1 && lane < kNB) bias_step(lane);   // a0, a1 of that update
      bias_pending = false;
      __syncthreads();
      if (kh == 0) {
        const float bias = ks == 0 ? biasl[l31_l] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* x = X + (mt * 32 + acc_row(i, h_l)) * kPbufPitch + l31_l;
          *x = acc[i] + *x + bias;
        }
      }
      __syncthreads();
      slab_tile_store(X, p.slabs + (int64_t)ks * B * NhP + n0, B, NhP, tid_l);
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid_l == 0)
        flag_raise(p.flag_fwd, wg, epoch);
      BSIG_STAMP(3);
    }

    // ---- while the row owners work: the evaluation due after the previous update
    //      (this tile still holds those weights), next feature tile, Adam scalars ------
    if (__builtin_expect(p.do_eval && step > 0 && (step - 1) % p.eval_every == 0, 0))
      tile_eval(p, Wl, X, biasl, evals_before(step, p.eval_every) - 1);
    // the time-out bit (set by any bounded poll on the chip), sampled off the critical path:
    // tested at the top of the next update
    if (tid_l == 0)
      red[63] = (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) ? 1.f : 0.f;
    if (t + 1 < p.n_updates) {
      const int64_t pf_row0 = (int64_t)(step + 1) * B;
      BSIG_PF_LIST(BSIG_PF_LOAD)
    }
    b1t *= p.beta1; b2t *= p.beta2;      // beta^t as running products (double)
    a0 = (float)(p.lr / (1.0 - b1t));
    a1 = (float)(1.0 / sqrt(1.0 - b2t));

    // ---- 3. dW = d_out^T F on this tile, Adam ----------------------------------
    // the owners' sum(u * dL/dsigma) granules double as their "d_out rows are
    // out" flags; the jitter-scale gradient term
    //   d pre += (EPS/(B*D*K)) * sum(u*dL/dsigma) * exp(pre)
    // is applied here, to this block's columns, while the tile is loaded
    if (w == 0) {
      const float su = granule_gather(p.gran + kGranArr, p.n_owner, epoch * 4u + 2u, lane, flagp);
      if (lane == 0) red[62] = p.eps_noise != 0.f ? p.eps_noise / ((float)B * (float)(p.D * p.K)) * su : 0.f;
    }
    __syncthreads();
    BSIG_STAMP(10);
    {
      const float c = red[62];
      const int sg_lo = p.K + p.D * p.K, sg_hi = p.K + 2 * p.D * p.K;
      const bool any_e = c != 0.f && n0 + kNB > sg_lo && n0 < sg_hi;     // workgroup-uniform
      const __amdgpu_buffer_rsrc_t dr = xwg_buffer(p.d_out + n0), er = xwg_buffer(p.e_out + n0);
      for (int base = 0; base < p.FR * (kNB / 4); base += kPT * 4) {
        f32x4 q[4], e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * kPT + tid_l;
          const int b = idx >> 3, c4 = (idx & 7) * 4;
          const bool ok = idx < p.FR * (kNB / 4) && b < B;
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          q[u] = ok ? xwg_load4(dr, b * NhP + c4) : zero;
          e[u] = ok && any_e ? xwg_load4(er, b * NhP + c4) : zero;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * kPT + tid_l;
          const int b = idx >> 3, c4 = (idx & 7) * 4;
          if (idx < p.FR * (kNB / 4)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int n = n0 + c4 + j;
              const float ev = (n >= sg_lo && n < sg_hi) ? e[u][j] : 0.f;   // (other columns of e_out hold no data)
              X[(c4 + j) * DOP + b] = q[u][j] + c * ev;
            }
          }
        }
      }
    }
    __syncthreads();
    BSIG_STAMP(11);
    if (ks == 0) {                         // biases of this block: column sums of d_out
      const int n = tid_l & 31, part = tid_l >> 5;
      float g = 0.f;
      for (int b = part; b < B; b += kPT / 32) g += X[n * DOP + b];
      bpart[part * 32 + n] = g;
    }
    {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = X + l31_l * DOP + 4 * h_l;
      const float* bp = Fl + (4 * h_l) * kPitch + kcol_l;
      for (int bb = 0; bb < p.FR; bb += 8) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + bb);
        const float f0 = bp[(bb + 0) * kPitch], f1 = bp[(bb + 1) * kPitch];
        const float f2 = bp[(bb + 2) * kPitch], f3 = bp[(bb + 3) * kPitch];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, f0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, f1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, f2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, f3, acc, 0, 0, 0);
      }
      if (dp) {
        // this rank's share of the gradient: summed over the ranks by the caller
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = n0 + acc_row(i, h_l);
          if (n < Nh && k0 + kcol_l < p.Fdim) p.grads[p.w_off + (int64_t)n * p.Fdim + k0 + kcol_l] = acc[i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* wp = Wl + acc_row(i, h_l) * kPitch + kcol_l;
          *wp = adam_weight(acc[i], Mr[i], Vr[i], *wp, a0, a1, ak);
        }
      }
    }
    bias_pending = ks == 0 && !dp;
    __syncthreads();
    if (dp && ks == 0 && tid_l < kNB && n0 + tid_l < Nh) {
      float g = 0.f;
#pragma unroll
      for (int q = 0; q < kPT / 32; ++q) g += bpart[q * 32 + tid_l];
      p.grads[p.b_off + n0 + tid_l] = g;
    }
    BSIG_STAMP(12);
  }

  // ---- the evaluation after the last update of the call ----------------------------
  // (a data-parallel rank: in the launch that only takes the pending Adam step of that update)
  if (p.do_eval && step0 + p.n_updates == p.n_total && (!dp || p.n_updates == 0)) {
    if (bias_pending && tid < kNB) bias_step(tid);
    bias_pending = false;
    __syncthreads();
    tile_eval(p, Wl, X, biasl, evals_before(p.n_total - 1, p.eval_every));
  }
  // ---- write the tile back, advance the engine state -------------------------
  const bool dirty = !dp;
  if (dirty) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + acc_row(i, h);
      if (n < Nh && col_ok) {
        const int64_t off = p.w_off + (int64_t)n * p.Fdim + k0 + kcol;
        p.params[off] = Wl[acc_row(i, h) * kPitch + kcol]; p.m1[off] = Mr[i]; p.m2[off] = Vr[i];
      }
    }
  }
  if (bias_pending && tid < kNB) bias_step(tid);
  if (dirty && ks == 0 && tid < kNB && n0 + tid < Nh) {
    p.params[p.b_off + n0 + tid] = biasl[tid];
    p.m1[p.b_off + n0 + tid] = biasl[32 + tid];
    p.m2[p.b_off + n0 + tid] = biasl[64 + tid];
  }
  if (p.prof && p.n_updates > 0 && tid == 0) {   // (diagnostics) the tile is written back
    __builtin_amdgcn_s_waitcnt(0);
    p.prof[((int64_t)wg * kProfUpdates) * 16 + 13] = wall_clock64();
  }
  if (wg == 0 && tid == 0 && p.n_updates > 0) {
    // (an aborted run leaves the counters of the planned run: the call fails anyway)
    int32_t* st = p.state;
    reinterpret_cast<double*>(st + 12)[0] = b1t;
    reinterpret_cast<double*>(st + 12)[1] = b2t;
    reinterpret_cast<float*>(st)[4] = a0;
    reinterpret_cast<float*>(st)[5] = a1;
    // (one jitter stream per update and per evaluation, in program order)
    int n_ev = 0;
    if (p.do_eval) {
      n_ev = evals_before(step0 + p.n_updates, p.eval_every) - evals_before(step0, p.eval_every);
      if (!dp && step0 + p.n_updates == p.n_total && (p.n_total - 1) % p.eval_every != 0) ++n_ev;
    }
    reinterpret_cast<uint64_t*>(st + 8)[1] += (uint64_t)(p.n_updates + n_ev);
    st[0] = step0 + p.n_updates;
  }
}

// ---- row-owner workgroups: reduce the k-slices, NLL forward / backward ---------
__device__ __forceinline__ void owner_workgroup(const PersistArgs& p, float* smem) {
  float* X = smem;                           // [R][per_wave] rows
  float* red = smem + p.x_floats;            // [64]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wg = blockIdx.x, o = wg - p.G;
  const int B = p.B, Nh = p.Nh, NhP = p.NhP, D = p.D, K = p.K, DK = D * K;
  int32_t* flagp = p.state + 2;
  const int step0 = p.state[0];
  const uint64_t rng_seed = reinterpret_cast<const uint64_t*>(p.state + 8)[0];
  const uint64_t rng_ctr0 = reinterpret_cast<const uint64_t*>(p.state + 8)[1];
  HeadArgs a{};
  a.D = D; a.K = K; a.Nh = Nh; a.batch = B; a.from_tuple = 0;
  a.min_w = p.min_w; a.ll_limit = p.ll_limit; a.inv_norm = p.inv_norm;
  a.eps_noise = p.eps_noise; a.seed = rng_seed; a.d_out = p.d_out;
  const float norm = (float)B * (float)DK;
  const int per_wave = Nh + D + 3 * K;
  const int r0 = o * p.R;
  float* tile = X + w * per_wave;
  float* yv = tile + Nh;
  float* rk = yv + D;
  float* lpk = rk + K;
  float* dlg = lpk + K;
  const int row = r0 + w;
  const bool owner_wave = w < p.R;
  const bool active = owner_wave && row < B;

  // ---- held-out evaluation number eidx (jitter stream `stream`): slot s = pass * R + r
  //      <-> held-out row pass * B + r0 + r, one wavefront per slot, forward only ---------
  auto owner_eval = [&](int eidx, uint64_t stream) {
    const unsigned etag = (unsigned)eidx + 1u;
    const int nslots = p.eval_passes * p.R;
    if (w == 0) flags_wait(p.flag_eval, p.G, etag, lane, flagp);
    __syncthreads();
    float eacc = 0.f;
    for (int pass = 0; pass < p.eval_passes; ++pass) {
      const int nr = min(min(p.R, B - r0), p.n_test - pass * B - r0);
      // (slots of rows past the held-out set are not read by anyone)
      if (nr > 0)
        slab_rows_sum(p.eval_slabs + ((((int64_t)(eidx & 1) * p.eval_passes + pass) * p.k_slices) * B + r0) * NhP,
                      p.k_slices, B, NhP, nr, Nh, X + pass * p.R * per_wave, per_wave, tid,
                      [&](int col, float v) { if (col >= K + DK && col < K + 2 * DK) eacc += expf(v); });
      const int nz = max(nr, 0);          // slots past the held-out set: zeros (their rows are masked)
      for (int idx = tid; idx < (p.R - nz) * Nh; idx += kPT)
        X[(pass * p.R + nz + idx / Nh) * per_wave + idx % Nh] = 0.f;
    }
    eacc = wave_sum_dpp(eacc);
    if (lane == 0) red[w] = eacc;
    __syncthreads();
    if (tid == 0) {
      float sx = 0.f;
      for (int q = 0; q < kPT / 64; ++q) sx += red[q];
      granule_publish(p.gran_eval, o, etag, sx);
    }
    const int pass = w / p.R, r = w - pass * p.R;
    const int erow = pass * B + r0 + r;
    const bool act = w < nslots && r0 + r < B && erow < p.n_test;
    if (act)
      for (int j = lane; j < D; j += 64) yv[j] = p.y_test[(int64_t)erow * p.ldy_test + j];
    RowOut ro;
    ro.lse = 0.f; ro.uds = 0.f; ro.bad = false;
#pragma unroll
    for (int q = 0; q < kElemsPerLane; ++q) ro.esg0[q] = 0.f;
    if (w < nslots) {
      HeadArgs ae = a;
      ae.d_out = nullptr;                    // forward only
      ae.batch = p.n_test;
      ae.stream_id = stream;
      diag_row(ae, erow, act, lane, tile, yv, rk, lpk, dlg,
               [&] {
                 return p.eps_noise != 0.f
                            ? p.eps_noise * (granule_gather(p.gran_eval, p.n_owner, etag, lane, flagp) /
                                             ((float)p.n_test * (float)DK))
                            : 0.f;
               },
               ro);
    }
    if (lane == 0) red[16 + w] = act ? ro.lse : 0.f;
    __syncthreads();
    if (tid == 0) {
      float sl = 0.f;
      for (int q = 0; q < kPT / 64; ++q) sl += red[16 + q];
      granule_publish(p.gran_eval + kGranArr, o, etag, sl);
    }
    if (o == 0 && w == 0) {
      const float sum = granule_gather(p.gran_eval + kGranArr, p.n_owner, etag, lane, flagp);
      if (lane == 0) {
        const float l = -sum / (float)p.n_test;
        p.test_loss[p.state[1]] = l;
        p.state[1] = p.state[1] + 1;
        if (!isfinite(l)) atomicOr(flagp, 1);
      }
    }
    if (ro.bad) atomicOr(flagp, 1);
    __syncthreads();
  };
  const int ev0 = p.do_eval ? evals_before(step0, p.eval_every) : 0;

  for (int t = 0; t < p.n_updates; ++t) {
    const int step = step0 + t;
    const unsigned epoch = (unsigned)step + 1u;
    const uint32_t tag = epoch * 4u;
    if (run_aborted(flagp, red, tid)) break;
    BSIG_STAMP(0);
    if (active) {      // target row (independent of the forward product)
      const int64_t yrow = p.ids[(int64_t)step * B + row];
      for (int j = lane; j < D; j += 64) yv[j] = p.y[yrow * p.ldy + j];
    }
    if (w == 0) flags_wait(p.flag_fwd, p.G, epoch, lane, flagp);
    __syncthreads();
    BSIG_STAMP(4);
    float eacc = 0.f;
    slab_rows_sum(p.slabs + (int64_t)r0 * NhP, p.k_slices, B, NhP, min(p.R, B - r0), Nh, X, per_wave, tid,
                  [&](int col, float v) { if (col >= K + DK && col < K + 2 * DK) eacc += expf(v); });
    eacc = wave_sum_dpp(eacc);
    if (lane == 0) red[w] = eacc;
    __syncthreads();
    if (tid == 0) {
      float sx = 0.f;
      for (int q = 0; q < kPT / 64; ++q) sx += red[q];
      granule_publish(p.gran, o, tag + 1, sx);
    }
    BSIG_STAMP(5);
    RowOut ro;
    ro.lse = 0.f; ro.uds = 0.f; ro.bad = false;
#pragma unroll
    for (int q = 0; q < kElemsPerLane; ++q) ro.esg0[q] = 0.f;
    float uds_w = 0.f;
    if (owner_wave) {
      // one jitter stream per update and per evaluation, in program order
      a.stream_id = rng_ctr0 + (uint64_t)t +
                    (uint64_t)(p.do_eval ? evals_before(step, p.eval_every) - ev0 : 0);
      diag_row(a, row, active, lane, tile, yv, rk, lpk, dlg,
               [&] {
                 BSIG_STAMP(6);
                 return p.eps_noise != 0.f
                            ? p.eps_noise * (granule_gather(p.gran, p.n_owner, tag + 1, lane, flagp) / norm)
                            : 0.f;
               },
               ro);
      uds_w = wave_sum_dpp(ro.uds);
      if (lane == 0) { red[16 + w] = active ? ro.lse : 0.f; red[32 + w] = uds_w; }
      BSIG_STAMP(7);
      if (active) {
        // d_out row without the jitter-scale term, and exp(pre) of the row for the
        // tile workgroups to add it (lane's elements are columns lane + q*TPR)
        for (int j = lane; j < K; j += 64) tile[j] = dlg[j];
        __builtin_amdgcn_wave_barrier();
        float* dst = p.d_out + (int64_t)row * NhP;
        for (int j = lane; j < Nh; j += 64) xwg_store(dst + j, tile[j]);
        if (p.eps_noise != 0.f) {
          const int TPR = (64 / K) * K;
          float* est = p.e_out + (int64_t)row * NhP + K + DK;
#pragma unroll
          for (int q = 0; q < kElemsPerLane; ++q)
            if (lane < TPR && lane + q * TPR < DK) xwg_store(est + lane + q * TPR, ro.esg0[q]);
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      float sl = 0.f, su = 0.f;
      for (int q = 0; q < p.R; ++q) { sl += red[16 + q]; su += red[32 + q]; }
      granule_publish(p.gran + kGranArr, o, tag + 2, su);
      granule_publish(loss_granules(p.gran, epoch), o, tag + 3, sl);
    }
    BSIG_STAMP(9);
    if (o == 0 && w == 0) {
      const float s = granule_gather(loss_granules(p.gran, epoch), p.n_owner, tag + 3, lane, flagp);
      if (lane == 0) {
        const float l = -s / (float)B;
        p.train_loss[step] = l;
        if (!isfinite(l)) atomicOr(flagp, 1);
      }
    }
    if (ro.bad) atomicOr(flagp, 1);
    // the evaluation due after the previous update: the tile workgroups formed its
    // products while this update's rows were being finished
    if (__builtin_expect(p.do_eval && step > 0 && (step - 1) % p.eval_every == 0, 0)) {
      __syncthreads();
      owner_eval(evals_before(step, p.eval_every) - 1,
                 rng_ctr0 + (uint64_t)t + (uint64_t)(evals_before(step, p.eval_every) - ev0) - 1u);
    }
  }
  if (p.do_eval && step0 + p.n_updates == p.n_total && (p.grads == nullptr || p.n_updates == 0) &&
      !run_aborted(flagp, red, tid))
    owner_eval(evals_before(p.n_total - 1, p.eval_every),
               rng_ctr0 + (uint64_t)p.n_updates + (uint64_t)(evals_before(p.n_total - 1, p.eval_every) - ev0));
}

// DP: data-parallel rank (gradients out, pending Adam step in; see PersistArgs)
template <bool DP>
__global__ __launch_bounds__(kPT) void linear_head_updates_v1_kernel(PersistArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifndef BSIG_HOST_SAN_BUILD   // (see fit_persistent_mdnn.hip)
  if ((int)blockIdx.x < p.G) tile_workgroup<DP>(p, smem);
  else owner_workgroup(p, smem);
#endif
}

// ---------------------------------------------------------------- host side
struct PersistGeom {
  int FR, Nh, NhP, n_blocks, k_slices, G, n_owner, R, x_floats;
  int eval_passes;              // 0: evaluations stay outside the launches
  size_t lds;
  size_t slab_floats, dout_floats, eval_floats;
};

static bool persist_geom(const PersistShape& s, PersistGeom* g) {
  if (s.batch < 1 || s.feat_dim < 4 || s.feat_dim % 4 != 0 || s.out_dim < 1 ||
      s.n_comp < 1 || s.n_comp > 64)
    return false;
  const int groups = 64 / s.n_comp;
  if (ceil_div(s.out_dim, groups) > kElemsPerLane) return false;   // diag_row's register cache
  g->FR = (int)round_up(s.batch, 8);
  if (g->FR > 128) return false;
  g->Nh = s.n_comp + 2 * s.out_dim * s.n_comp;
  g->n_blocks = ceil_div(g->Nh, kNB);
  g->NhP = g->n_blocks * kNB;
  g->k_slices = ceil_div(s.feat_dim, kPC);
  g->G = g->n_blocks * g->k_slices;          // tile workgroups
  if (g->G > kXwgMax - 8 || g->k_slices > 32) return false;
  // row owners: further workgroups, every workgroup of the launch on its own CU
  g->R = ceil_div(s.batch, std::min(kXwgMax - g->G, s.batch));
  if (g->R > kPT / 64) return false;
  g->n_owner = ceil_div(s.batch, g->R);
  const int per_wave = g->Nh + s.out_dim + 3 * s.n_comp;
  g->x_floats = (int)round_up(std::max(std::max(128 * kPbufPitch, kNB * (g->FR + 4)),
                                       (kPT / 64) * per_wave), 4);
  g->lds = ((size_t)g->FR * kPitch + (size_t)kNB * kPitch + g->x_floats + 64 + 96 + (kPT / 32) * 32) * sizeof(float);
  // the forward reads feature rows up to 127 (results of rows >= batch are dropped):
  // small minibatches get an allocation that covers those reads
  g->lds = std::max(g->lds, (size_t)128 * kPitch * sizeof(float));
  if (g->lds > (size_t)kLdsLimit) return false;
  g->slab_floats = (size_t)g->k_slices * s.batch * g->NhP;
  g->dout_floats = (size_t)s.batch * g->NhP;
  // in-launch evaluations: one wavefront of an owner workgroup per held-out row slot
  g->eval_passes = s.max_test > 0 ? ceil_div(s.max_test, s.batch) : 0;
  if (g->eval_passes * g->R > kPT / 64) g->eval_passes = 0;
  g->eval_floats = (size_t)2 * g->eval_passes * g->slab_floats;
  return true;
}

// Every workgroup of the launch must be resident at once (they wait for each
// other): one per CU because of the LDS footprint, so the device needs that many
// CUs and must grant the LDS.
static bool device_can_host(const PersistGeom& g) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  if (prop.multiProcessorCount < g.G + g.n_owner || (size_t)prop.maxSharedMemoryPerMultiProcessor < g.lds)
    return false;
  // ... and the runtime's own occupancy answer for this kernel at this LDS size must admit a
  // workgroup per CU (registers, LDS, wave slots): the workgroups wait for each other, a grid
  // the device cannot hold at once never finishes
  int per_cu = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_head_updates_v1_kernel<false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, linear_head_updates_v1_kernel<false>, kPT, g.lds) != hipSuccess)
    return false;
  return per_cu >= 1;
}

// (diagnostics / tests) occupy `blocks` CUs for `ms` milliseconds with workgroups that hold
// `lds_bytes` of LDS each: a persistent launch behind it does not get all its workgroups resident
__global__ void spin_kernel(long long ticks, int* sink) {
  extern __shared__ float spin_lds[];
  spin_lds[threadIdx.x] = 1.f;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
  if (spin_lds[threadIdx.x] == 2.f) sink[0] = 1;
}
int debug_spin(int blocks, size_t lds_bytes, int ms, hipStream_t st) {
  BSIG_REQUIRE(blocks >= 1 && ms >= 0 && lds_bytes <= (size_t)kLdsLimit, "debug_spin: bad args");
  BSIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit));
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), std::max<size_t>(lds_bytes, 1024), st,
                     (long long)ms * 100000LL /* 100 MHz wall clock */, (int*)nullptr);
  BSIG_CHECK_LAUNCH("debug_spin");
  return BSIG_OK;
}

bool persist_v1_supported(const PersistShape& s) {
  PersistGeom g;
  return persist_geom(s, &g) && device_can_host(g);
}
bool persist_v1_eval_supported(const PersistShape& s) {
  PersistGeom g;
  return persist_geom(s, &g) && device_can_host(g) && g.eval_passes > 0;
}

static size_t data_bytes(const PersistGeom& g) {
  return round_up<size_t>((g.slab_floats + 2 * g.dout_floats + g.eval_floats) * sizeof(float), 256);
}
static size_t sync_bytes() { return 2 * kFlagArr * sizeof(unsigned) + 6 * kGranArr * 8; }

size_t persist_v1_workspace_bytes(const PersistShape& s) {
  PersistGeom g;
  if (!persist_geom(s, &g)) return 0;
  return data_bytes(g) + sync_bytes();
}

int persist_v1_reset_regions(const PersistShape& s, void* workspace, size_t workspace_bytes,
                          ZeroRegion* regions) {
  PersistGeom g;
  BSIG_REQUIRE(persist_geom(s, &g), "persistent updates: shape not covered");
  BSIG_REQUIRE(workspace && workspace_bytes >= persist_v1_workspace_bytes(s),
               "persistent updates: workspace too small");
  char* base = reinterpret_cast<char*>(workspace);
  const size_t slab_bytes = g.slab_floats * sizeof(float);
  // d_out / exp(pre) rows (their padding columns stay zero), flags and granules
  regions[0] = ZeroRegion{base + slab_bytes, 2 * g.dout_floats * sizeof(float)};
  regions[1] = ZeroRegion{base + data_bytes(g), sync_bytes()};
  return BSIG_OK;
}

static long long* g_prof = nullptr;
void persist_set_profile_buffer(void* buf) { g_prof = reinterpret_cast<long long*>(buf); }
void* persist_profile_buffer() { return g_prof; }

int persist_v1_run(const PersistShape& s, const PersistBuffers& b, const PersistHyper& hy, int n,
                hipStream_t st) {
  PersistGeom g;
  BSIG_REQUIRE(persist_geom(s, &g), "persistent updates: shape not covered");
  BSIG_REQUIRE(b.feats && b.y && b.ids && b.params && b.exp_avg && b.exp_avg_sq && b.state &&
                   b.train_loss && b.workspace, "persistent updates: null buffer");
  BSIG_REQUIRE(b.workspace_bytes >= persist_v1_workspace_bytes(s),
               "persistent updates: workspace too small");
  BSIG_REQUIRE(b.ld_feats % 4 == 0 && aligned(b.feats, 16) && b.ld_feats >= s.feat_dim,
               "persistent updates: features must be 16-byte aligned rows");
  BSIG_REQUIRE(!(b.adam_pending && !b.grads), "persistent updates: pending Adam step without gradients");
  BSIG_REQUIRE(!(b.grads && n > 1), "persistent updates: data-parallel launches take one update");
  if (n <= 0 && !b.adam_pending && !b.do_eval) return BSIG_OK;
  // the > 64 KB dynamic-LDS attribute is per device (the plan's device is the current one:
  // the Python mirror enters the model's device around every call)
  static bool attr_set_dev[64] = {};
  int attr_dev = 0;
  BSIG_HIP(hipGetDevice(&attr_dev));
  bool& attr_set = attr_set_dev[attr_dev & 63];
  if (!attr_set) {
    BSIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_head_updates_v1_kernel<false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit));
    BSIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_head_updates_v1_kernel<true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit));
    attr_set = true;
  }
  PersistArgs p{};
  p.B = s.batch; p.FR = g.FR; p.Fdim = s.feat_dim; p.Nh = g.Nh; p.NhP = g.NhP;
  p.D = s.out_dim; p.K = s.n_comp;
  p.n_blocks = g.n_blocks; p.k_slices = g.k_slices; p.G = g.G; p.n_owner = g.n_owner; p.R = g.R;
  p.n_updates = std::max(n, 0); p.x_floats = g.x_floats;
  p.grads = b.grads; p.adam_pending = b.adam_pending;
  p.feats = b.feats; p.ld_feats = b.ld_feats; p.feat_ids = b.feat_ids; p.y = b.y; p.ldy = b.ldy; p.ids = b.ids;
  p.params = b.params; p.m1 = b.exp_avg; p.m2 = b.exp_avg_sq; p.w_off = b.w_off; p.b_off = b.b_off;
  p.quad_ok = (p.w_off % 4 == 0 && p.Fdim % 4 == 0 &&
               ((reinterpret_cast<uintptr_t>(b.params) | reinterpret_cast<uintptr_t>(b.exp_avg) |
                 reinterpret_cast<uintptr_t>(b.exp_avg_sq) | reinterpret_cast<uintptr_t>(b.grads)) & 15) == 0) ? 1 : 0;
  p.state = b.state; p.train_loss = b.train_loss;
  p.lr = hy.lr; p.beta1 = hy.beta1; p.beta2 = hy.beta2;
  p.adam_eps = hy.adam_eps; p.eps_noise = hy.eps_noise; p.min_w = hy.min_weight;
  p.ll_limit = hy.ll_limit; p.inv_norm = 1.0f / (float)hy.norm_batch;
  char* base = reinterpret_cast<char*>(b.workspace);
  p.slabs = reinterpret_cast<float*>(base);
  p.d_out = p.slabs + g.slab_floats;
  p.e_out = p.d_out + g.dout_floats;
  p.eval_slabs = p.e_out + g.dout_floats;
  char* sync = base + data_bytes(g);
  p.flag_fwd = reinterpret_cast<unsigned*>(sync);
  p.flag_eval = p.flag_fwd + kFlagArr;
  p.gran = reinterpret_cast<unsigned long long*>(sync + 2 * kFlagArr * sizeof(unsigned));
  p.gran_eval = p.gran + 3 * kGranArr;
  if (b.do_eval) {
    BSIG_REQUIRE(g.eval_passes > 0 && b.n_test >= 1 && b.n_test <= g.eval_passes * s.batch &&
                     b.y_test && b.test_loss && b.eval_every >= 1 && b.n_total >= 1,
                 "persistent updates: in-launch evaluation not covered");
    p.do_eval = 1; p.eval_every = b.eval_every; p.n_total = b.n_total; p.n_test = b.n_test;
    p.eval_passes = ceil_div(b.n_test, s.batch); p.eval_row0 = b.eval_row0;
    p.y_test = b.y_test; p.ldy_test = b.ldy_test; p.test_loss = b.test_loss;
  }
  p.prof = g_prof;
  // (a pending-Adam-only launch needs the tile workgroups only)
  if (b.grads)
    hipLaunchKernelGGL(linear_head_updates_v1_kernel<true>, dim3(n > 0 || b.do_eval ? g.G + g.n_owner : g.G),
                       dim3(kPT), g.lds, st, p);
  else
    hipLaunchKernelGGL(linear_head_updates_v1_kernel<false>, dim3(g.G + g.n_owner), dim3(kPT), g.lds,
                       st, p);
  BSIG_CHECK_LAUNCH("linear_head_updates");
  return BSIG_OK;
}

}  // namespace bsig
