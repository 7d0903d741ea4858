// Device side of the persistent update kernels of the two-layer MDNN (fit_persistent_mdnn.hip:
// W1 tiles resident on the chip; fit_persistent_mdnn_stream.hip: W1 streamed): the argument
// block, the tile / small-weight / row-owner workgroup bodies.  Reference: mdnn.py:89-125
// (forward), :127-178 (NLL), :203/:219-233 (Adam, update loop).
#pragma once
#include "persist_mdnn.h"

#include <algorithm>

#include "persist.h"
#include "persist_device.h"

namespace bsig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMT = 512;            // threads per workgroup (8 wavefronts)
constexpr int kMC = 256;            // first-layer input columns per tile workgroup
constexpr int kMPitch = kMC + 4;    // LDS row pitch of the summary / weight tiles
constexpr int kMNB = 32;            // weight rows per tile / small-weight workgroup
constexpr int kMH = 128;            // hidden width (both layers)
constexpr int kMHP = kMH + 4;       // LDS pitch of a 128-wide activation row
constexpr int kMR = 4;              // minibatch rows per owner workgroup (power of two <= 8; 2 where the chip has the CUs)
constexpr int kStreamMR = 8;        // ... of a plan with a streamed first layer
constexpr int kMPbuf = 33;
constexpr int kMLdsLimit = 160 * 1024;
constexpr int kFastRowsMaxInput = 4096;   // (mdnn_owner_workgroup: fast rows)

struct MdnnArgs {
  int B, FR, I, Nh, Nh16, NhP, D, K;
  int k_slices, G1, n_owner, n_small;
  int n_updates, x_floats;
  const float* x; int64_t ldx; const int32_t* ids;
  int x_fac, xS, xA;   // x rows are cross-correlation factors [sf S | af A | mean | std | 1]
                       // (x_test rows are always materialised summaries)
  const float* y; int64_t ldy;
  float* params; float* m1; float* m2;
  int64_t w1_off, b1_off, w2_off, b2_off, wh_off, bh_off;
  int32_t* state; float* train_loss;
  double lr, beta1, beta2;
  float adam_eps, eps_noise, min_w, ll_limit, inv_norm;
  // exchange area (written through, read around the L2)
  float* slabs;   // [k_slices][B][128] first-layer partial products
  float* dz1;     // [B][128]   owners -> tiles
  float* h1;      // [B][128]   owners -> small-weight workgroups (W2 blocks)
  float* h2;      // [B][128]   ... (head blocks)
  float* dz2;     // [B][128]   ... (W2 blocks)
  float* d_out;   // [B][NhP]   ... (head blocks)
  // W2 again, in the order the owners' MFMA operands want it (one coalesced 8-byte load
  // per lane and instruction): written by the W2 small-weight workgroups next to `params`
  // Two copies by the parity of the update that reads them: the copy of update t stays
  // intact while update t's Adam step writes the copy of t+1 (an evaluation at the end of
  // update t still reads W2 as of t).
  float* w2f_pack;   // [2][8 waves][8 tt][2][64 lanes][2]: W2[16w + c16][16tt + 4g + 2half + e]
  float* w2b_pack;   // [2][8 waves][8 tt][2][64 lanes][2]: W2[16tt + 4g + 2half + e][16w + c16]
  unsigned* flag_fwd; unsigned* flag_own; unsigned* flag_small; unsigned* flag_pack;
  // WIDE heads (Nh > 272: the head matrix no longer fits the row owners' LDS next to their
  // activations -- the reference YAMLs' 10 components with D > 13, cfg/ant.yaml:69-70): the
  // head-block workgroups, which hold 32 rows of the head matrix anyway, form the head outputs
  // and their share of d_out Wh for ALL minibatch rows; per update owners -> (h2) -> head blocks
  // -> (o_wide) -> owners -> (d_out) -> head blocks -> (dz2_part) -> owners
  int wide, n_hb;
  float* o_wide;     // [n_hb][B][32]   raw head outputs, bias included (block-major like the slabs)
  float* dz2_part;   // [n_hb][B][128]  partial d_out Wh of each head block
  unsigned* flag_h2; unsigned* flag_o; unsigned* flag_dout; unsigned* flag_dz2;
  unsigned launch_tag;   // flag_pack value of THIS launch: the small weights (and the W2 packs)
                         // as of its start are out, one flag per small-weight workgroup
  unsigned long long* gran;
  // data-parallel ranks (one update per launch, the caller all-reduces `grads` between
  // launches): weight / bias gradients go to `grads` (flat layout) instead of into Adam,
  // and the Adam step of the PREVIOUS update (on the reduced gradients) is taken by the
  // weights' owners while they load them (adam_pending)
  float* grads; int adam_pending;
  // ... or RESIDENT across the exchange (DP = false instantiations, xr_ready != null; persist.h): the
  // G1 tile and the n_small small-weight workgroups write their gradients through to `grads`, count
  // themselves in xr_count (a word of the sync region), the last one raises *xr_ready, all poll
  // *xr_done and take their Adam step from the reduced `grads`
  unsigned* xr_ready; const unsigned* xr_done; unsigned* xr_count; unsigned xr_base;
  int pair_ok;                       // rows of W1 are 8-byte aligned pairs (w1_off, I even)
  int fast_rows;                     // (A/B runs: BSIG_MDNN_FAST_ROWS=0 keeps the shape-generic row)
  // held-out evaluations inside the launch (mdnn.py:235-242; do_eval), as in
  // fit_persistent.hip: the tile workgroups form the held-out rows' first-layer products
  // while they wait for the row owners of the NEXT update (their LDS still holds the
  // evaluated weights); the owners run layers 2.. and the forward NLL of their held-out
  // rows after they have published that update's rows.
  int do_eval, eval_every, n_total, n_test, eval_passes;
  const float* x_test; int64_t ldx_test;
  const float* y_test; int64_t ldy_test;
  float* test_loss;                  // [n_evals]
  float* eval_slabs;                 // [2][eval_passes][k_slices][B][128]
  float* eval_out;                   // [n_test][NhP] head outputs of the held-out rows (staging)
  unsigned* flag_eval;               // [G1]  evaluation number + 1
  unsigned long long* gran_eval;     // [2][kXwgMax] {tag, value}: sum exp(pre), sum logsumexp
  long long* prof;   // diagnostics: [256][kMProfUpdates][16] wall-clock stamps, or null
  // where the row owners take the first-layer pre-activations from: o_k_slices slabs [B][128]
  // behind G1 flags.  Resident W1 tiles: the tiles' split-K slabs (slabs, k_slices, flag_fwd);
  // streamed W1: ONE slab, already summed by the tile workgroups (hpre, 1, flag_red)
  const float* o_slabs; int o_k_slices; unsigned* o_flags;
  // ---- streamed first layer (fit_persistent_mdnn_stream.hip): W1 does not fit the chip
  //      (cfg/anymal.yaml I = 56402, cfg/shadow_hand_more.yaml I = 105002); tile workgroup g
  //      owns hidden units [32 (g % 4), +32) and walks the 256-column chunks
  //      [q*s_chunks/G, (q+1)*s_chunks/G), q = g / 4, G = G1 / 4
  int stream, s_chunks;
  int s_nip, s_pf;     // factor rows in LDS: sf slots [0, s_nip), af | mean std 0.. at s_nip, pitch s_pf
  float* hpre;         // [B][128] layer-1 pre-activations summed over the tile workgroups (+ b1)
  unsigned* flag_red;  // [G1] the quads of hpre this workgroup sums are out
  unsigned* flag_evp; unsigned* flag_evr;   // [G1] evaluation passes of a streamed plan: slab out / summed
  const float* xe; int64_t ldxe;   // streamed plan: the held-out pairs' factor rows
  // wide heads, evaluation passes: h2 rows of the held-out pairs [B][128] out (owners), their head
  // outputs [n_hb][B][32] out (head blocks)  (last: the offsets of the fields above are the ones the
  // update loop's register allocation was tuned with)
  float* h2e; float* oe; unsigned* flag_h2e; unsigned* flag_oe;
};

constexpr int kMProfUpdates = 8;
#define BSIG_MSTAMP(k)                                                              \
  do {                                                                              \
    if (p.prof && threadIdx.x == 0 && t < kMProfUpdates)                            \
      p.prof[((int64_t)blockIdx.x * kMProfUpdates + t) * 16 + (k)] = wall_clock64(); \
  } while (0)

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global
// load and store of the wavefront (s_waitcnt vmcnt(0)): in the streamed pass that
// would expose the latency of the next chunk's loads at every barrier, and in the owners the round
// trip of each write-through store (h1, h2, d_out, dz2 leave for memory while the next product runs;
// the s_waitcnt(0) in front of the owners' flag is what publishes them).
// (lds_barrier: persist_device.h)

// LDS hand-over inside one wavefront (its own stores before its own loads of other lanes' data)
__device__ __forceinline__ void lds_wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// First-layer inputs from cross-correlation FACTOR rows (summarizers.py:106-119; layout in
// bsig.h): x[i*A + j] = sf[i] * af[j] -- the one fp32 multiply the summarizer itself would
// do --, x[S*A] = mean * 1, x[S*A + 1] = std * 1; columns beyond that come out as 1 * 1 and
// are masked by the callers.  FacCols: where the two factors of the four consecutive inputs
// x[col .. col+3] sit inside a factor row (the same for every row).
struct FacCols { int oi[4], oj[4]; };
__device__ __forceinline__ FacCols fac_cols(int col, int S, int A) {
  const int SA = S * A, one = S + A + 2;
  // col / A without an integer division: float estimate, two corrections (col < 2^24)
  int i = (int)((float)col * __builtin_amdgcn_rcpf((float)A));
  if (i * A > col) --i;
  if ((i + 1) * A <= col) ++i;
  int j = col - i * A;
  FacCols c;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int k = col + e;
    c.oi[e] = k < SA ? i : (k == SA ? S + A : (k == SA + 1 ? S + A + 1 : one));
    c.oj[e] = k < SA ? S + j : one;
    if (++j == A) { j = 0; ++i; }
  }
  return c;
}
__device__ __forceinline__ float4 fac_load4(const float* __restrict__ rowp, const FacCols& c) {
  return make_float4(rowp[c.oi[0]] * rowp[c.oj[0]], rowp[c.oi[1]] * rowp[c.oj[1]],
                     rowp[c.oi[2]] * rowp[c.oj[2]], rowp[c.oi[3]] * rowp[c.oj[3]]);
}

// one [<=104, 256] summary tile = 13 float4 per thread in named registers.  Wavefront w holds -- and later
// writes to LDS -- the tile's columns 32w .. 32w+31 (lane: row 8u + (lane >> 3), the quad at column
// 32w + 4 (lane & 7)): the same 32 columns whose dW1 block it forms, so it can put the NEXT minibatch's
// columns into LDS as soon as its own dW1 product has read the old ones, without waiting for the other
// wavefronts (the tile used to go in at the top of the next update, 0.9 us in front of its forward MFMAs:
// 104 KB through ds_write_b128 at ~80 B/clk, then a barrier).
#define BSIG_MPF_LIST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12)
#define BSIG_MPF_DECL(u) float4 pf##u;
#define BSIG_MPF_COL() (32 * (tid >> 6) + 4 * (tid & 7))
#define BSIG_MPF_LOAD(u)                                                                    \
  {                                                                                         \
    const int prow = min(8 * (u) + ((tid & 63) >> 3), B - 1);                               \
    const int64_t fr = (int64_t)p.ids[pf_row0 + prow];                                      \
    if constexpr (FAC) {                                                                    \
      pf##u = fac_load4(p.x + fr * p.ldx, fcols);                                           \
    } else {                                                                                \
      const int64_t col = min((int64_t)k0 + BSIG_MPF_COL(), p.ldx - 4);                     \
      pf##u = *reinterpret_cast<const float4*>(p.x + fr * p.ldx + col);                     \
    }                                                                                       \
  }
#define BSIG_MPF_ZERO(u) pf##u = make_float4(0.f, 0.f, 0.f, 0.f);
// columns >= I (row padding, the tail of the last k-slice) enter as zeros
#define BSIG_MPF_STORE(u)                                                                   \
  {                                                                                         \
    const int prow = 8 * (u) + ((tid & 63) >> 3);                                           \
    if (prow < B) {                                                                         \
      const int col = k0 + BSIG_MPF_COL();                                                  \
      float4 v = pf##u;                                                                     \
      v.x = col + 0 < p.I ? v.x : 0.f; v.y = col + 1 < p.I ? v.y : 0.f;                     \
      v.z = col + 2 < p.I ? v.z : 0.f; v.w = col + 3 < p.I ? v.w : 0.f;                     \
      *reinterpret_cast<float4*>(Fl + prow * kMPitch + BSIG_MPF_COL()) = v;                 \
    }                                                                                       \
  }

// number of evaluation points it % every == 0 strictly before update s (the evaluation after
// the last update of a call is not one of them)
__device__ __forceinline__ int mdnn_evals_before(int s, int every) { return s == 0 ? 0 : (s - 1) / every + 1; }

// ---- tile workgroups, evaluation number eidx: held-out summaries x this tile's weights (the
//      A operand straight from memory: the minibatch tile in LDS is still needed for dW1)
//      -> evaluation slabs, flag
__device__ __forceinline__ void mdnn_tile_eval(const MdnnArgs& p, const float* Wl, float* X,
                                               const float* biasl, int eidx) {
  // (laundered: nothing below may be computed ahead of the update loop and kept live in it)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wg = blockIdx.x, ks = wg % p.k_slices, nb = wg / p.k_slices;
  const int n0 = nb * kMNB, k0 = ks * kMC;
  const int B = p.B;
  const int mt = w & 3, kh = w >> 2;
  for (int pass = 0; pass < p.eval_passes; ++pass) {
    const int rows = min(B, p.n_test - pass * B);
    if (rows <= 0) break;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* src = p.x_test + (int64_t)(pass * B + min(mt * 32 + l31, rows - 1)) * p.ldx_test;
    const float* bp = Wl + l31 * kMPitch + kh * 128 + 4 * h;
    // two halves of the k range: 8 x 16 bytes of the A operand in registers at a time
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      float4 areg[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int col = k0 + kh * 128 + 64 * half + 8 * q + 4 * h;
        float4 v = *reinterpret_cast<const float4*>(src + min((int64_t)col, p.ldx_test - 4));
        v.x = col + 0 < p.I ? v.x : 0.f; v.y = col + 1 < p.I ? v.y : 0.f;
        v.z = col + 2 < p.I ? v.z : 0.f; v.w = col + 3 < p.I ? v.w : 0.f;
        areg[q] = v;
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
      for (int i = 0; i < 16; ++i) X[(mt * 32 + acc_row(i, h)) * kMPbuf + l31] = acc[i];
    }
    __syncthreads();
    if (kh == 0) {     // (rows of 16-byte quads, as the update's slab)
      const float bias = ks == 0 ? biasl[l31] : 0.f;
      const __amdgpu_buffer_rsrc_t sr = xwg_buffer(p.eval_slabs + (((int64_t)(eidx & 1) * p.eval_passes + pass) * p.k_slices + ks) * B * kMH + n0);
      const int a = l31 & 3, c4 = l31 & ~3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = acc[4 * q + e] + X[(mt * 32 + acc_row(4 * q + e, h)) * kMPbuf + l31] + bias;
        const f32x4 t = quad_transpose4(v[0], v[1], v[2], v[3], a);
        const int row = mt * 32 + 8 * q + 4 * h + a;
        if (row < rows) xwg_store4(sr, row * kMH + c4, t[0], t[1], t[2], t[3]);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0)
    flag_raise(p.flag_eval, wg, (unsigned)eidx + 1u);
}

// ---- tile workgroups: first-layer partial products, dW1, Adam -------------------
template <bool DP, bool FAC>
__device__ __forceinline__ void mdnn_tile_workgroup(const MdnnArgs& p, float* smem) {
  float* Fl = smem;                          // [FR][kMPitch] minibatch summaries (this k-slice)
  float* Wl = Fl + p.FR * kMPitch;           // [32][kMPitch] weight tile (authoritative copy)
  float* X = Wl + kMNB * kMPitch;            // scratch: forward k-halves | dz1^T
  float* red = X + p.x_floats;               // [64]
  float* biasl = red + 64;                   // [3][32] b1, exp_avg, exp_avg_sq (k-slice 0)
  float* bpart = biasl + 96;                 // [16][32] partial column sums of dz1 (k-slice 0)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int wg = blockIdx.x, ks = wg % p.k_slices, nb = wg / p.k_slices;
  const int n0 = nb * kMNB, k0 = ks * kMC;
  const int B = p.B;
  int32_t* flagp = p.state + 2;
  const int step0 = p.state[0];
  double b1t = reinterpret_cast<const double*>(p.state + 12)[0];
  double b2t = reinterpret_cast<const double*>(p.state + 12)[1];
  float a0 = 0.f, a1 = 0.f;
  const AdamK ak{1.0f - (float)p.beta1, (float)p.beta2, 1.0f - (float)p.beta2, p.adam_eps};

  float Mr[16], Vr[16];
  const int kcol = 32 * w + l31;
  const bool col_ok = k0 + kcol < p.I;
  const bool pend = DP && p.adam_pending != 0;
  const bool xr = !DP && p.xr_ready != nullptr;      // resident across the gradient exchange (MdnnArgs)
  // first launch of a run_training call: a fresh optimizer (mdnn.py:203) -- the moments start
  // at zero in the registers, nobody has to clear (or read) them in memory
  const bool fresh = !DP && step0 == 0;
  const float pa0 = pend ? reinterpret_cast<const float*>(p.state)[4] : 0.f;
  const float pa1 = pend ? reinterpret_cast<const float*>(p.state)[5] : 0.f;
  // (three passes: every load is issued before the first store of a data-parallel launch's
  // pending Adam step -- interleaved, the possibly aliasing stores serialise the loads)
  if (DP && p.pair_ok) {
    // A data-parallel launch only passes through the tile here (pending Adam step, weights into
    // LDS), and Adam is elementwise: the tile is taken as rows of 8-byte pairs (the rows of W1 are
    // I floats apart -- 11 802 for the Ant summaries: 8-byte aligned, not 16), 512 contiguous bytes
    // per wavefront instruction and all 32 loads of a thread in flight, instead of 64 dword loads
    // per lane in the accumulator layout.
    float2 Wq[8], Mq[8], Vq[8], Gq2[8];
    const int c2 = (tid & 127) * 2, rq = tid >> 7;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t off = p.w1_off + (int64_t)(n0 + q * 4 + rq) * p.I + k0 + c2;
      Wq[q] = Mq[q] = Vq[q] = Gq2[q] = make_float2(0.f, 0.f);
      if (k0 + c2 < p.I) {
        Wq[q] = *reinterpret_cast<const float2*>(p.params + off);
        if (pend) {
          Mq[q] = *reinterpret_cast<const float2*>(p.m1 + off);
          Vq[q] = *reinterpret_cast<const float2*>(p.m2 + off);
          Gq2[q] = *reinterpret_cast<const float2*>(p.grads + off);
        }
      }
    }
    if (pend) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        Wq[q].x = adam_weight(Gq2[q].x, Mq[q].x, Vq[q].x, Wq[q].x, pa0, pa1, ak);
        Wq[q].y = adam_weight(Gq2[q].y, Mq[q].y, Vq[q].y, Wq[q].y, pa0, pa1, ak);
      }
      if (k0 + c2 < p.I) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int64_t off = p.w1_off + (int64_t)(n0 + q * 4 + rq) * p.I + k0 + c2;
          *reinterpret_cast<float2*>(p.params + off) = Wq[q];
          *reinterpret_cast<float2*>(p.m1 + off) = Mq[q];
          *reinterpret_cast<float2*>(p.m2 + off) = Vq[q];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<float2*>(Wl + (q * 4 + rq) * kMPitch + c2) = Wq[q];
#pragma unroll
    for (int i = 0; i < 16; ++i) { Mr[i] = 0.f; Vr[i] = 0.f; }
  } else {
    float Wv[16], Gq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + acc_row(i, h);
      Wv[i] = 0.f; Gq[i] = 0.f; Mr[i] = 0.f; Vr[i] = 0.f;
      if (col_ok) {
        const int64_t off = p.w1_off + (int64_t)n * p.I + k0 + kcol;
        Wv[i] = p.params[off];
        if (!fresh) { Mr[i] = p.m1[off]; Vr[i] = p.m2[off]; }
        if (pend) Gq[i] = p.grads[off];
      }
    }
    if (pend) {   // written back at once: a data-parallel launch changes the tile only here
#pragma unroll
      for (int i = 0; i < 16; ++i) Wv[i] = adam_weight(Gq[i], Mr[i], Vr[i], Wv[i], pa0, pa1, ak);
      if (col_ok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int64_t off = p.w1_off + (int64_t)(n0 + acc_row(i, h)) * p.I + k0 + kcol;
          p.params[off] = Wv[i]; p.m1[off] = Mr[i]; p.m2[off] = Vr[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Wl[acc_row(i, h) * kMPitch + kcol] = Wv[i];
  }
  if (ks == 0 && tid < kMNB) {
    const int64_t off = p.b1_off + n0 + tid;
    float bw = p.params[off], bm = fresh ? 0.f : p.m1[off], bv = fresh ? 0.f : p.m2[off];
    if (pend) {
      bw = adam_bias(p.grads[off], bm, bv, bw, pa0, pa1, ak);
      p.params[off] = bw; p.m1[off] = bm; p.m2[off] = bv;
    }
    biasl[tid] = bw; biasl[32 + tid] = bm; biasl[64 + tid] = bv;
  }
  for (int idx = tid; idx < (p.FR - B) * kMPitch; idx += kMT) Fl[B * kMPitch + idx] = 0.f;
  const int DOP = p.FR + 4;
  const int nvec = B * (kMC / 4);
  bool bias_pending = false;
  auto bias_step = [&](int n) {
    float g = 0.f;
#pragma unroll
    for (int q = 0; q < kMT / 32; ++q) g += bpart[q * 32 + n];
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
  BSIG_MPF_LIST(BSIG_MPF_DECL)
  if (p.n_updates > 0) {
    const int64_t pf_row0 = (int64_t)step0 * B;
    FacCols fcols{};
    if constexpr (FAC) fcols = fac_cols(k0 + BSIG_MPF_COL(), p.xS, p.xA);
    BSIG_MPF_LIST(BSIG_MPF_LOAD)
    BSIG_MPF_LIST(BSIG_MPF_STORE)          // the first update's tile (later ones: behind the dW1 product)
  } else {
    BSIG_MPF_LIST(BSIG_MPF_ZERO)
  }
  __syncthreads();

  for (int t = 0; t < p.n_updates; ++t) {
    int tid_l = tid, l31_l = l31, h_l = h, kcol_l = kcol;
    asm volatile("" : "+v"(tid_l), "+v"(l31_l), "+v"(h_l), "+v"(kcol_l));
    const int step = step0 + t;
    const unsigned epoch = (unsigned)step + 1u;
    if (red[63] != 0.f) break;   // time-out bit as sampled during the previous update's wait
    BSIG_MSTAMP(0);
    // ---- 1. (the summary tile is in LDS: written behind the previous update's dW1 product) ----
    BSIG_MSTAMP(1);

    // ---- 2. partial forward: P[b, n] = sum_{k in slice} X[b, k] W1[n, k] ---------
    {
      const int mt = w & 3, kh = w >> 2;
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = Fl + (mt * 32 + l31_l) * kMPitch + kh * 128 + 4 * h_l;
      const float* bp = Wl + l31_l * kMPitch + kh * 128 + 4 * h_l;
#pragma unroll 4
      for (int kk = 0; kk < 128; kk += 8) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + kk);
        const float4 b4 = *reinterpret_cast<const float4*>(bp + kk);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
      }
      BSIG_MSTAMP(2);
      if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) X[(mt * 32 + acc_row(i, h_l)) * kMPbuf + l31_l] = acc[i];
      }
      if (bias_pending && w == kMT / 64 - 1 && lane < kMNB) bias_step(lane);
      bias_pending = false;
      __syncthreads();
      // The two k-halves meet (same operation order as ever: (first half + second half) + bias), and
      // the block goes out as rows of 16-byte quads: an accumulator lane holds four ROWS of one column
      // -- stored from there the slab was 16 dword stores per lane, and a cross-workgroup payload costs
      // by the number of memory transactions, not by its bytes (tools/micro/handoff_bench.hip) -- a
      // 4 x 4 transpose inside each lane quad (quad_transpose4) turns them into four columns of a row.
      if (kh == 0) {
        const float bias = ks == 0 ? biasl[l31_l] : 0.f;
        const __amdgpu_buffer_rsrc_t sr = xwg_buffer(p.slabs + (int64_t)ks * B * kMH + n0);
        const int a = l31_l & 3, c4 = l31_l & ~3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = acc[4 * q + e] + X[(mt * 32 + acc_row(4 * q + e, h_l)) * kMPbuf + l31_l] + bias;
          const f32x4 t = quad_transpose4(v[0], v[1], v[2], v[3], a);
          const int row = mt * 32 + 8 * q + 4 * h_l + a;
          if (row < B) xwg_store4(sr, row * kMH + c4, t[0], t[1], t[2], t[3]);
        }
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid_l == 0)
        flag_raise(p.flag_fwd, wg, epoch);
      BSIG_MSTAMP(3);
    }

    // ---- while the owners work: the evaluation due after the previous update (this tile
    //      still holds those weights), next summary tile, Adam scalars -------------------
    if (__builtin_expect(p.do_eval && step > 0 && (step - 1) % p.eval_every == 0, 0))
      mdnn_tile_eval(p, Wl, X, biasl, mdnn_evals_before(step, p.eval_every) - 1);
    // the time-out bit (set by any bounded poll on the chip), sampled off the critical path:
    // tested at the top of the next update
    if (tid_l == 0)
      red[63] = (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) ? 1.f : 0.f;
    if (t + 1 < p.n_updates) {
      const int64_t pf_row0 = (int64_t)(step + 1) * B;
      FacCols fcols{};
      if constexpr (FAC) fcols = fac_cols(k0 + BSIG_MPF_COL(), p.xS, p.xA);
      BSIG_MPF_LIST(BSIG_MPF_LOAD)
    }
    b1t *= p.beta1; b2t *= p.beta2;
    a0 = (float)(p.lr / (1.0 - b1t));
    a1 = (float)(1.0 / sqrt(1.0 - b2t));

    if constexpr (FAC) {   // (diagnostics) the factor products of the next tile are in registers
      if (p.prof) { asm volatile("" :: "v"(pf0.x), "v"(pf12.w)); BSIG_MSTAMP(8); }
    }
    // ---- 3. dW1 = dz1^T X on this tile, Adam ---------------------------------------
    // (quiet waiting, head_device.h: one word until the first owner is through, then the rest)
    if (w == 0) {
      flag_wait_one(p.flag_own, 0, epoch, flagp);
      flags_wait(p.flag_own, p.n_owner, epoch, lane, flagp);
    }
    __syncthreads();
    BSIG_MSTAMP(10);
    {
      // [FR, 32] block of dz1, transposed into X: 16 bytes per lane, 8 lanes per row
      const __amdgpu_buffer_rsrc_t zr = xwg_buffer(p.dz1 + n0);
      for (int base = 0; base < p.FR * (kMNB / 4); base += kMT * 2) {
        f32x4 q[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int idx = base + u * kMT + tid_l;
          const int b = idx >> 3;
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          q[u] = (idx < p.FR * (kMNB / 4) && b < B) ? xwg_load4(zr, b * kMH + (idx & 7) * 4) : zero;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int idx = base + u * kMT + tid_l;
          if (idx < p.FR * (kMNB / 4)) {
            float* x = X + ((idx & 7) * 4) * DOP + (idx >> 3);
            x[0] = q[u].x; x[DOP] = q[u].y; x[2 * DOP] = q[u].z; x[3 * DOP] = q[u].w;
          }
        }
      }
    }
    __syncthreads();
    BSIG_MSTAMP(11);
    if (ks == 0) {                         // b1 of this block: column sums of dz1
      const int n = tid_l & 31, part = tid_l >> 5;
      float g = 0.f;
      for (int b = part; b < B; b += kMT / 32) g += X[n * DOP + b];
      bpart[part * 32 + n] = g;
    }
    {
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = X + l31_l * DOP + 4 * h_l;
      const float* bp = Fl + (4 * h_l) * kMPitch + kcol_l;
      for (int bb = 0; bb < p.FR; bb += 8) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + bb);
        const float f0 = bp[(bb + 0) * kMPitch], f1 = bp[(bb + 1) * kMPitch];
        const float f2 = bp[(bb + 2) * kMPitch], f3 = bp[(bb + 3) * kMPitch];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, f0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, f1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, f2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, f3, acc, 0, 0, 0);
      }
      if (DP) {
        // this rank's share of the gradient: summed over the ranks by the caller
        if (k0 + kcol_l < p.I) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            p.grads[p.w1_off + (int64_t)(n0 + acc_row(i, h_l)) * p.I + k0 + kcol_l] = acc[i];
        }
      } else if (xr) {
        // resident across the exchange: written through (the exchange stream's all-reduce reads memory)
        if (k0 + kcol_l < p.I) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            xwg_store(p.grads + p.w1_off + (int64_t)(n0 + acc_row(i, h_l)) * p.I + k0 + kcol_l, acc[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* wp = Wl + acc_row(i, h_l) * kMPitch + kcol_l;
          *wp = adam_weight(acc[i], Mr[i], Vr[i], *wp, a0, a1, ak);
        }
      }
    }
    // this wavefront's 32 columns of the NEXT minibatch tile (requested in the wait window above)
    if (t + 1 < p.n_updates) { BSIG_MPF_LIST(BSIG_MPF_STORE) }
    bias_pending = ks == 0 && !DP && !xr;
    __syncthreads();
    if (DP && ks == 0 && tid_l < kMNB) {
      float g = 0.f;
#pragma unroll
      for (int q = 0; q < kMT / 32; ++q) g += bpart[q * 32 + tid_l];
      p.grads[p.b1_off + n0 + tid_l] = g;
    }
    if (!DP && xr) {
      if (ks == 0 && tid_l < kMNB) {
        float g = 0.f;
#pragma unroll
        for (int q = 0; q < kMT / 32; ++q) g += bpart[q * 32 + tid_l];
        xwg_store(p.grads + p.b1_off + n0 + tid_l, g);
      }
      // ---- the exchange: gradients out (acknowledged), count, signal, wait, reduced gradients in ----
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      BSIG_MSTAMP(13);
      if (tid_l == 0)
        xr_hand_off(p.xr_count, p.xr_ready, p.xr_done, p.xr_base, (unsigned)t + 1u, (unsigned)(p.G1 + p.n_small), flagp);
      __syncthreads();
      BSIG_MSTAMP(14);
      if (k0 + kcol_l < p.I) {
        float gq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
          gq[i] = xwg_load(p.grads + p.w1_off + (int64_t)(n0 + acc_row(i, h_l)) * p.I + k0 + kcol_l);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* wp = Wl + acc_row(i, h_l) * kMPitch + kcol_l;
          *wp = adam_weight(gq[i], Mr[i], Vr[i], *wp, a0, a1, ak);
        }
      }
      if (ks == 0 && tid_l < kMNB) {
        const float g = xwg_load(p.grads + p.b1_off + n0 + tid_l);
        float bm = biasl[32 + tid_l], bv = biasl[64 + tid_l];
        biasl[tid_l] = adam_bias(g, bm, bv, biasl[tid_l], a0, a1, ak);
        biasl[32 + tid_l] = bm;
        biasl[64 + tid_l] = bv;
      }
      __syncthreads();
    }
    BSIG_MSTAMP(12);
  }

  // ---- the evaluation after the last update of the call ------------------------------
  // (a data-parallel rank: in the launch that only takes the pending Adam step of that update)
  if (p.do_eval && step0 + p.n_updates == p.n_total && (!DP || p.n_updates == 0)) {
    if (bias_pending && tid < kMNB) bias_step(tid);
    bias_pending = false;
    __syncthreads();
    mdnn_tile_eval(p, Wl, X, biasl, mdnn_evals_before(p.n_total - 1, p.eval_every));
  }
  // ---- write the tile back, advance the engine state ---------------------------
  if (col_ok && !DP) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + acc_row(i, h);
      const int64_t off = p.w1_off + (int64_t)n * p.I + k0 + kcol;
      p.params[off] = Wl[acc_row(i, h) * kMPitch + kcol]; p.m1[off] = Mr[i]; p.m2[off] = Vr[i];
    }
  }
  if (bias_pending && tid < kMNB) bias_step(tid);
  if (!DP && ks == 0 && tid < kMNB) {
    p.params[p.b1_off + n0 + tid] = biasl[tid];
    p.m1[p.b1_off + n0 + tid] = biasl[32 + tid];
    p.m2[p.b1_off + n0 + tid] = biasl[64 + tid];
  }
  // a resident launch that gave up must not leave the exchange stream waiting for gradients that will
  // never come: every wait of the call passes
  if (xr && wg == 0 && tid == 0 && (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2))
    __hip_atomic_store(p.xr_ready, p.xr_base + (unsigned)p.n_updates, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (wg == 0 && tid == 0 && p.n_updates > 0) {
    int32_t* st = p.state;
    reinterpret_cast<double*>(st + 12)[0] = b1t;
    reinterpret_cast<double*>(st + 12)[1] = b2t;
    reinterpret_cast<float*>(st)[4] = a0;
    reinterpret_cast<float*>(st)[5] = a1;
    // (one jitter stream per update and per evaluation, in program order)
    int n_ev = 0;
    if (p.do_eval) {
      n_ev = mdnn_evals_before(step0 + p.n_updates, p.eval_every) - mdnn_evals_before(step0, p.eval_every);
      if (!DP && step0 + p.n_updates == p.n_total && (p.n_total - 1) % p.eval_every != 0) ++n_ev;
    }
    reinterpret_cast<uint64_t*>(st + 8)[1] += (uint64_t)(p.n_updates + n_ev);
    st[0] = step0 + p.n_updates;
  }
}

// ---- wide heads, held-out evaluation number eidx, a head-block workgroup's part: head outputs of
//      the pass's rows for its 32 columns, from the h2 rows the owners publish (h2e) to
//      oe [n_hb][B][32]; the weights are those of the operand copy Wb / bsh.  Deliberately NOT
//      inlined: it runs six times per call, and its registers (and the scalar arguments it keeps)
//      must not weigh on the update loops of the kernel -- inlined, the tile workgroups' loop spilled.
__device__ __attribute__((noinline)) void mdnn_serve_eval(float* Hs, const float* Wb, float* Xo, const float* bsh,
                                                          const float* h2e, float* oe, unsigned* flag_h2e,
                                                          unsigned* flag_oe, int32_t* flagp, int eval_passes,
                                                          int n_test, int B, int n_owner, int hb, int eidx) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  for (int gp = 0; gp < eval_passes; ++gp) {
    if (n_test - gp * B <= 0) break;
    const unsigned wtag = (unsigned)eidx * 16u + (unsigned)gp + 1u;
    if (w == 0) flags_wait(flag_h2e, n_owner, wtag, lane, flagp);
    __syncthreads();
    {
      const __amdgpu_buffer_rsrc_t hr = xwg_buffer(h2e);
      for (int base = 0; base < B * (kMH / 4); base += kMT * 4) {
        f32x4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = min(base + u * kMT + tid, B * (kMH / 4) - 1);
          q[u] = xwg_load4(hr, (idx >> 5) * kMH + (idx & 31) * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int idx = base + u * kMT + tid;
          if (idx < B * (kMH / 4))
            *reinterpret_cast<f32x4*>(Hs + (idx >> 5) * kMHP + (idx & 31) * 4) = q[u];
        }
      }
    }
    __syncthreads();
    const int mt = w & 3, kh = w >> 2;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float* ap = Hs + (mt * 32 + l31) * kMHP + kh * 64 + 4 * h;
    const float* bp = Wb + l31 * kMHP + kh * 64 + 4 * h;
#pragma unroll 4
    for (int kk = 0; kk < 64; kk += 8) {
      const float4 a4 = *reinterpret_cast<const float4*>(ap + kk);
      const float4 b4 = *reinterpret_cast<const float4*>(bp + kk);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
    }
    if (kh == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) Xo[(mt * 32 + acc_row(i, h)) * kMPbuf + l31] = acc[i];
    }
    __syncthreads();
    if (kh == 0) {
      const float bias = bsh[l31];
      const __amdgpu_buffer_rsrc_t orr = xwg_buffer(oe + (int64_t)hb * B * kMNB);
      const int a = l31 & 3, c4 = l31 & ~3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = mt * 32 + acc_row(4 * q + e, h);
          v[e] = acc[4 * q + e] + Xo[row * kMPbuf + l31] + bias;
        }
        const f32x4 t = quad_transpose4(v[0], v[1], v[2], v[3], a);
        const int row = mt * 32 + 8 * q + 4 * h + a;
        if (row < B) xwg_store4(orr, row * kMNB + c4, t[0], t[1], t[2], t[3]);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) flag_raise(flag_oe, hb, wtag);
  }
}

// ---- small-weight workgroups: 32 rows of W2 or of the head matrix ----------------
template <bool DP, bool WIDE>
__device__ __forceinline__ void mdnn_small_workgroup(const MdnnArgs& p, float* smem) {
  float* Hs = smem;                          // [FR][kMHP] input activations of the layer
  float* X = Hs + p.FR * kMHP;               // [32][FR + 4] output gradients, transposed
  float* red = X + kMNB * (p.FR + 4);        // [64]
  float* Wb = red + 64;                      // wide heads: [32][kMHP] this block's weights, operand order
  float* Xo = Wb + kMNB * kMHP;              // ... [128][33] k-half exchange of the head-output product
  float* bsh = Xo + 128 * kMPbuf;            // ... [32] this block's biases
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int sb = blockIdx.x - p.G1 - p.n_owner;
  const bool is_w2 = sb < kMH / kMNB;
  const int n0 = (is_w2 ? sb : sb - kMH / kMNB) * kMNB;
  const int nrows = is_w2 ? kMH : p.Nh;
  const int64_t w_off = is_w2 ? p.w2_off : p.wh_off, b_off = is_w2 ? p.b2_off : p.bh_off;
  const float* dsrc = is_w2 ? p.dz2 : p.d_out;
  const int dpitch = is_w2 ? kMH : p.NhP;
  const float* hsrc = is_w2 ? p.h1 : p.h2;
  const int B = p.B, DOP = p.FR + 4;
  const bool whead = WIDE && !is_w2;         // forms head outputs / d_out Wh for all rows (MdnnArgs)
  const int hb = sb - kMH / kMNB;
  int32_t* flagp = p.state + 2;
  const int step0 = p.state[0];
  double b1t = reinterpret_cast<const double*>(p.state + 12)[0];
  double b2t = reinterpret_cast<const double*>(p.state + 12)[1];
  const AdamK ak{1.0f - (float)p.beta1, (float)p.beta2, 1.0f - (float)p.beta2, p.adam_eps};

  const bool pend = DP && p.adam_pending != 0;
  const bool xr = !DP && p.xr_ready != nullptr;      // resident across the gradient exchange (MdnnArgs)
  const bool fresh = !DP && step0 == 0;      // fresh optimizer: the moments start at zero
  const float pa0 = pend ? reinterpret_cast<const float*>(p.state)[4] : 0.f;
  const float pa1 = pend ? reinterpret_cast<const float*>(p.state)[5] : 0.f;
  // waves 0-3: element i of lane (h, l31) of wave w <-> W[n0 + acc_row(i, h)][32w + l31]
  float Wr[16], Mr[16], Vr[16];
  const int kcol = 32 * (w & 3) + l31;
  {
    float Gq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + acc_row(i, h);
      Wr[i] = 0.f; Mr[i] = 0.f; Vr[i] = 0.f; Gq[i] = 0.f;
      if (w < 4 && n < nrows) {
        const int64_t off = w_off + (int64_t)n * kMH + kcol;
        Wr[i] = p.params[off];
        if (!fresh) { Mr[i] = p.m1[off]; Vr[i] = p.m2[off]; }
        if (pend) Gq[i] = p.grads[off];
      }
    }
    if (pend) {   // (all loads before the first store: see the tile workgroups)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + acc_row(i, h);
        if (w < 4 && n < nrows) {
          const int64_t off = w_off + (int64_t)n * kMH + kcol;
          Wr[i] = adam_weight(Gq[i], Mr[i], Vr[i], Wr[i], pa0, pa1, ak);
          xwg_store(p.params + off, Wr[i]); p.m1[off] = Mr[i]; p.m2[off] = Vr[i];
        }
      }
    }
  }
  // the owners' operand-order copies of W2 (see MdnnArgs)
  auto publish_w2 = [&](int i, int h, int kcol, int for_step) {
    const int n = n0 + acc_row(i, h);
    const int lf = ((kcol >> 2) & 3) * 16 + (n & 15), lb = ((n >> 2) & 3) * 16 + (kcol & 15);
    const int par = (for_step & 1) * (kMH * kMH);
    xwg_store(p.w2f_pack + par + ((((n >> 4) * 8 + (kcol >> 4)) * 2 + ((kcol >> 1) & 1)) * 64 + lf) * 2 + (kcol & 1), Wr[i]);
    xwg_store(p.w2b_pack + par + ((((kcol >> 4) * 8 + (n >> 4)) * 2 + ((n >> 1) & 1)) * 64 + lb) * 2 + (n & 1), Wr[i]);
  };
  if (is_w2 && w < 4) {
#pragma unroll
    for (int i = 0; i < 16; ++i) publish_w2(i, h, kcol, step0);
  }
  // wave 4, lanes 0-31: the 32 biases
  const bool bias_lane = w == 4 && lane < kMNB && n0 + lane < nrows;
  float bw = 0.f, bm = 0.f, bv = 0.f;
  if (bias_lane) {
    const int64_t off = b_off + n0 + lane;
    bw = p.params[off];
    if (!fresh) { bm = p.m1[off]; bv = p.m2[off]; }
    if (pend) {
      bw = adam_bias(p.grads[off], bm, bv, bw, pa0, pa1, ak);
      xwg_store(p.params + off, bw); p.m1[off] = bm; p.m2[off] = bv;
    }
  }
#define BSIG_REFRESH_OPERAND_COPY()                                                        \
  if constexpr (WIDE) {                                                                     \
    if (whead && w < 4) {                                                                   \
      _Pragma("unroll") for (int i = 0; i < 16; ++i) Wb[acc_row(i, h) * kMHP + kcol] = Wr[i]; \
    } else if (whead && w == 4 && lane < kMNB) {                                            \
      bsh[lane] = bw;                                                                       \
    }                                                                                       \
  }
  BSIG_REFRESH_OPERAND_COPY()
  // the weights as of the start of this launch are out (owners wait for every block)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0)
    flag_raise(p.flag_pack, sb, p.launch_tag);
  for (int idx = tid; idx < (p.FR - B) * kMHP; idx += kMT) Hs[B * kMHP + idx] = 0.f;
  __syncthreads();

  // (wide heads: the head outputs of an evaluation pass -- mdnn_serve_eval, out of line)
  auto serve_eval = [&](int eidx) {
    if constexpr (WIDE) {
      if (whead)
        mdnn_serve_eval(Hs, Wb, Xo, bsh, p.h2e, p.oe, p.flag_h2e, p.flag_oe, flagp, p.eval_passes, p.n_test,
                        B, p.n_owner, hb, eidx);
    }
  };

  for (int t = 0; t < p.n_updates; ++t) {
    const int step = step0 + t;
    const unsigned epoch = (unsigned)step + 1u;
    if (run_aborted(flagp, red, tid)) break;
    b1t *= p.beta1; b2t *= p.beta2;
    const float a0 = (float)(p.lr / (1.0 - b1t));
    const float a1 = (float)(1.0 / sqrt(1.0 - b2t));
    BSIG_MSTAMP(0);
#define BSIG_LOAD_ACTIVATIONS() /* [B, 128], 16-byte loads */                                  \
    {                                                                                           \
      const __amdgpu_buffer_rsrc_t hr = xwg_buffer(hsrc);                                       \
      for (int base = 0; base < B * (kMH / 4); base += kMT * 4) {                               \
        f32x4 q[4];                                                                             \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
          const int idx = min(base + u * kMT + tid, B * (kMH / 4) - 1);                         \
          q[u] = xwg_load4(hr, (idx >> 5) * kMH + (idx & 31) * 4);                              \
        }                                                                                       \
        _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                         \
          const int idx = base + u * kMT + tid;                                                 \
          if (idx < B * (kMH / 4))                                                              \
            *reinterpret_cast<f32x4*>(Hs + (idx >> 5) * kMHP + (idx & 31) * 4) = q[u];          \
        }                                                                                       \
      }                                                                                         \
    }
#define BSIG_LOAD_GRADIENT_BLOCK() /* this block's gradient columns [B, 32], transposed */      \
    {                                                                                           \
      const __amdgpu_buffer_rsrc_t gr = xwg_buffer(dsrc);                                       \
      for (int base = 0; base < p.FR * (kMNB / 4); base += kMT * 2) {                           \
        f32x4 q[2];                                                                             \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                         \
          const int idx = base + u * kMT + tid;                                                 \
          const int b = idx >> 3, n = n0 + (idx & 7) * 4;                                       \
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};                                              \
          q[u] = (idx < p.FR * (kMNB / 4) && b < B && n < nrows) ? xwg_load4(gr, b * dpitch + n) : zero; \
        }                                                                                       \
        _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                         \
          const int idx = base + u * kMT + tid;                                                 \
          if (idx < p.FR * (kMNB / 4)) {                                                        \
            const int n = n0 + (idx & 7) * 4;                                                   \
            float* x = X + ((idx & 7) * 4) * DOP + (idx >> 3);                                  \
            x[0] = n < nrows ? q[u].x : 0.f; x[DOP] = n + 1 < nrows ? q[u].y : 0.f;            \
            x[2 * DOP] = n + 2 < nrows ? q[u].z : 0.f; x[3 * DOP] = n + 3 < nrows ? q[u].w : 0.f; \
          }                                                                                     \
        }                                                                                       \
      }                                                                                         \
    }
    if (WIDE && whead) {
      // ---- wide heads: head outputs of ALL minibatch rows for this block's 32 columns ------
      if (w == 0) flags_wait(p.flag_h2, p.n_owner, epoch, lane, flagp);
      __syncthreads();
      BSIG_MSTAMP(7);
      BSIG_LOAD_ACTIVATIONS()
      __syncthreads();
      BSIG_MSTAMP(8);
      {
        const int mt = w & 3, kh = w >> 2;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* ap = Hs + (mt * 32 + l31) * kMHP + kh * 64 + 4 * h;
        const float* bp = Wb + l31 * kMHP + kh * 64 + 4 * h;
#pragma unroll 4
        for (int kk = 0; kk < 64; kk += 8) {
          const float4 a4 = *reinterpret_cast<const float4*>(ap + kk);
          const float4 b4 = *reinterpret_cast<const float4*>(bp + kk);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        BSIG_MSTAMP(1);
        if (kh == 1) {
#pragma unroll
          for (int i = 0; i < 16; ++i) Xo[(mt * 32 + acc_row(i, h)) * kMPbuf + l31] = acc[i];
        }
        __syncthreads();
        BSIG_MSTAMP(2);
        if (kh == 0) {
          const float bias = bsh[l31];
          // block-major: [n_hb][B][32]; four rows of a column per lane -> four columns of a row
          // (quad_transpose4): 4 stores of 16 bytes per lane instead of 16 of 4
          const __amdgpu_buffer_rsrc_t orr = xwg_buffer(p.o_wide + (int64_t)hb * B * kMNB);
          const int a = l31 & 3, c4 = l31 & ~3;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int row = mt * 32 + acc_row(4 * q + e, h);
              v[e] = acc[4 * q + e] + Xo[row * kMPbuf + l31] + bias;
            }
            const f32x4 t = quad_transpose4(v[0], v[1], v[2], v[3], a);
            const int row = mt * 32 + 8 * q + 4 * h + a;
            if (row < B) xwg_store4(orr, row * kMNB + c4, t[0], t[1], t[2], t[3]);
          }
        }
        BSIG_MSTAMP(11);
        __builtin_amdgcn_s_waitcnt(0);
        BSIG_MSTAMP(3);
        __syncthreads();
        if (tid == 0) flag_raise(p.flag_o, hb, epoch);
        BSIG_MSTAMP(9);
      }
      // ---- ... and this block's share of d_out Wh (with the weights of this update's forward)
      if (w == 0) flags_wait(p.flag_dout, p.n_owner, epoch, lane, flagp);
      __syncthreads();
      BSIG_MSTAMP(4);
      BSIG_LOAD_GRADIENT_BLOCK()
      __syncthreads();
      {
        const int mt = w & 3, ih = w >> 2;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const int col = (2 * ih + jt) * 32 + l31;
          f32x16 acc;
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
          for (int g8 = 0; g8 < 4; ++g8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int k = 8 * g8 + 4 * h + e;
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(X[k * DOP + mt * 32 + l31], Wb[k * kMHP + col],
                                                         acc, 0, 0, 0);
            }
          }
          const __amdgpu_buffer_rsrc_t zr = xwg_buffer(p.dz2_part + (int64_t)hb * B * kMH);
          const int a = l31 & 3, c4 = col & ~3;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 t = quad_transpose4(acc[4 * q + 0], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3], a);
            const int row = mt * 32 + 8 * q + 4 * h + a;
            if (row < B) xwg_store4(zr, row * kMH + c4, t[0], t[1], t[2], t[3]);
          }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) flag_raise(p.flag_dz2, hb, epoch);
        BSIG_MSTAMP(10);
      }
    } else {
      if (w == 0) flags_wait(p.flag_own, p.n_owner, epoch, lane, flagp);
      __syncthreads();
      BSIG_MSTAMP(4);
      BSIG_LOAD_ACTIVATIONS()
      BSIG_LOAD_GRADIENT_BLOCK()
    }
    __syncthreads();
    BSIG_MSTAMP(5);
    if (w < 4) {
      // (laundered: the 48 store addresses below are recomputed per update, not kept live)
      int h_l = h, kcol_l = kcol;
      asm volatile("" : "+v"(h_l), "+v"(kcol_l));
      f32x16 acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const float* ap = X + l31 * DOP + 4 * h;
      const float* bp = Hs + (4 * h) * kMHP + kcol;
      for (int bb = 0; bb < p.FR; bb += 8) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + bb);
        const float f0 = bp[(bb + 0) * kMHP], f1 = bp[(bb + 1) * kMHP];
        const float f2 = bp[(bb + 2) * kMHP], f3 = bp[(bb + 3) * kMHP];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, f0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, f1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, f2, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, f3, acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + acc_row(i, h_l);
        if (DP) {
          if (n < nrows) p.grads[w_off + (int64_t)n * kMH + kcol_l] = acc[i];
        } else if (xr) {
          if (n < nrows) xwg_store(p.grads + w_off + (int64_t)n * kMH + kcol_l, acc[i]);
        } else {
          Wr[i] = adam_weight(acc[i], Mr[i], Vr[i], Wr[i], a0, a1, ak);
          if (n < nrows) xwg_store(p.params + w_off + (int64_t)n * kMH + kcol_l, Wr[i]);
          if (is_w2) publish_w2(i, h_l, kcol_l, step + 1);
        }
      }
    } else if (bias_lane) {
      float g = 0.f;
      for (int b = 0; b < B; ++b) g += X[lane * DOP + b];
      if (DP) {
        p.grads[b_off + n0 + lane] = g;
      } else if (xr) {
        xwg_store(p.grads + b_off + n0 + lane, g);
      } else {
        bw = adam_bias(g, bm, bv, bw, a0, a1, ak);
        xwg_store(p.params + b_off + n0 + lane, bw);
      }
    }
    if (!DP && xr) {
      // ---- the exchange (see the tile workgroups), then the Adam step on the reduced gradients ----
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0)
        xr_hand_off(p.xr_count, p.xr_ready, p.xr_done, p.xr_base, (unsigned)t + 1u, (unsigned)(p.G1 + p.n_small), flagp);
      __syncthreads();
      if (w < 4) {
        int h_l = h, kcol_l = kcol;
        asm volatile("" : "+v"(h_l), "+v"(kcol_l));
        float gq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = n0 + acc_row(i, h_l);
          gq[i] = n < nrows ? xwg_load(p.grads + w_off + (int64_t)n * kMH + kcol_l) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int n = n0 + acc_row(i, h_l);
          Wr[i] = adam_weight(gq[i], Mr[i], Vr[i], Wr[i], a0, a1, ak);
          if (n < nrows) xwg_store(p.params + w_off + (int64_t)n * kMH + kcol_l, Wr[i]);
          if (is_w2) publish_w2(i, h_l, kcol_l, step + 1);
        }
      } else if (bias_lane) {
        bw = adam_bias(xwg_load(p.grads + b_off + n0 + lane), bm, bv, bw, a0, a1, ak);
        xwg_store(p.params + b_off + n0 + lane, bw);
      }
    }
    // wide heads: the evaluation due after the PREVIOUS update asks for head outputs now -- the owners
    // run it behind this update's rows -- and wants that update's weights: the operand copy, before
    // it is refreshed with this update's step
    if constexpr (WIDE && !DP) {
      if (__builtin_expect(p.do_eval && step > 0 && (step - 1) % p.eval_every == 0, 0)) {
        __syncthreads();
        serve_eval(mdnn_evals_before(step, p.eval_every) - 1);
      }
    }
    if (!DP) {
      BSIG_REFRESH_OPERAND_COPY()   // (all reads of the old copy are behind the barrier above)
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0)
        flag_raise(p.flag_small, sb, epoch);
    }
    BSIG_MSTAMP(6);
  }
  // the evaluation after the last update of the call (the operand copy holds that update's weights)
  if constexpr (WIDE && !DP) {
    if (p.do_eval && step0 + p.n_updates == p.n_total && !run_aborted(flagp, red, tid))
      serve_eval(mdnn_evals_before(p.n_total - 1, p.eval_every));
  }
  if (!DP) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + acc_row(i, h);
      if (w < 4 && n < nrows) {
        const int64_t off = w_off + (int64_t)n * kMH + kcol;
        p.m1[off] = Mr[i]; p.m2[off] = Vr[i];
      }
    }
    if (bias_lane) { p.m1[b_off + n0 + lane] = bm; p.m2[b_off + n0 + lane] = bv; }
  }
}

// ---- row-owner workgroups: layers 2.., NLL forward / backward ---------------------
// FULL: full covariance (one lane per component runs the triangular solves: full_row)
// A row owner's sum over the k-slices of its rows' layer-1 partial products: slabs [k_slices][B][128];
// item = (row, column quad), kSumSub threads share an item's slices (16-byte loads, all in flight),
// their partial sums meet in a fixed order through `part` ([kSumSub - 1][kSumItems] quads of LDS).
// Returns the item's sum to the threads tid < kSumItems.  Ends on a workgroup barrier.
constexpr int kSumFlight = 12;
constexpr int sum_items(int mr) { return mr * (kMH / 4); }
// (owners of fewer than four rows keep the four-way split of the k-slices -- and with it the bits of
// every sum -- on the first sum_items * 4 threads; the bytes a CU has to pull are what the rows cost)
constexpr int sum_sub(int mr) { return mr >= 4 ? kMT / sum_items(mr) : 4; }
constexpr int kSumItems = sum_items(kMR), kSumSub = sum_sub(kMR);   // (host-side LDS sizing of the default)
template <int MR>
__device__ __forceinline__ f32x4 slab_quads_sum(const float* slabs, int k_slices, int zs, int row, int tid,
                                                float* part) {
  constexpr int SUB = sum_sub(MR);
  const int item = tid & ((MR * (kMH / 4)) - 1), sub = tid / (MR * (kMH / 4));
  const int c4 = (item & 31) * 4;
  const __amdgpu_buffer_rsrc_t sr = xwg_buffer(slabs);
  if (k_slices == 1) {     // (streamed first layer: already summed; `part` is not used -- nor allocated)
    f32x4 one = {0.f, 0.f, 0.f, 0.f};
    if (sub == 0) one = xwg_load4(sr, row * kMH + c4);
    __syncthreads();
    return one;
  }
  const int per = ceil_div(k_slices, SUB);
  const int z_lo = sub * per, z_hi = sub < SUB ? min(z_lo + per, k_slices) : z_lo;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  for (int z = z_lo; z < z_hi; z += kSumFlight) {
    f32x4 q[kSumFlight];
#pragma unroll
    for (int u = 0; u < kSumFlight; ++u) q[u] = xwg_load4(sr, min(z + u, z_hi - 1) * zs + row * kMH + c4);
#pragma unroll
    for (int u = 0; u < kSumFlight; ++u)
      if (z + u < z_hi) v += q[u];
  }
  if (sub > 0 && sub < SUB) *reinterpret_cast<f32x4*>(part + ((sub - 1) * (MR * (kMH / 4)) + item) * 4) = v;
  __syncthreads();
  if (sub == 0) {
#pragma unroll
    for (int q = 1; q < SUB; ++q) v += *reinterpret_cast<const f32x4*>(part + ((q - 1) * (MR * (kMH / 4)) + item) * 4);
  }
  return v;
}

// ---- wide heads, an owner's part of an evaluation pass: flag its h2 rows (stored, acknowledged),
//      wait for the head blocks' outputs, fetch its MR rows of them into Os; returns this thread's
//      share of sum(exp(pre)).  Out of line for the same reason as mdnn_serve_eval.
template <int MR>
__device__ __attribute__((noinline)) float mdnn_owner_eval_heads(unsigned* flag_h2e, unsigned* flag_oe, const float* oe,
                                                                 int32_t* flagp, float* Os, int po, int o, int n_hb,
                                                                 int B, int r0, int Nh, int Nh16, int K, int DK,
                                                                 int n_test, int gp, int eidx) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned wtag = (unsigned)eidx * 16u + (unsigned)gp + 1u;
  float eacc = 0.f;
  if (tid == 0) flag_raise(flag_h2e, o, wtag);
  if (w == 0) flags_wait(flag_oe, n_hb, wtag, lane, flagp);
  __syncthreads();
  const int per_row = n_hb * (kMNB / 4), n_items = MR * per_row;
  const __amdgpu_buffer_rsrc_t orr = xwg_buffer(oe);
  for (int base = 0; base < n_items; base += 4 * kMT) {
    f32x4 q[4];
    int rr[4], jj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(base + u * kMT + tid, n_items - 1);
      const int r = idx / per_row, rem = idx - r * per_row;
      rr[u] = r; jj[u] = (rem >> 3) * kMNB + (rem & 7) * 4;
      q[u] = xwg_load4(orr, ((rem >> 3) * B + min(r0 + r, B - 1)) * kMNB + (rem & 7) * 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (base + u * kMT + tid < n_items) {
        const bool rok = r0 + rr[u] < B && gp * B + r0 + rr[u] < n_test;
        const float v4[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int j = jj[u] + e;
          const float v = rok && j < Nh ? v4[e] : 0.f;
          if (j < Nh16) Os[rr[u] * po + j] = v;
          if (rok && j >= K + DK && j < K + 2 * DK) eacc += expf(v);
        }
      }
    }
  }
  return eacc;
}

// Head outputs h2 Wh^T of an owner with one or two rows, on the vector ALU.  A 16x16x4 fp32 MFMA is a
// chain of four fused multiply-adds in ascending k (= lane group) order -- tools/micro/mfma_order_probe.hip:
// 256 of 256 outputs bit-equal to fmaf chains, also over 32 chained instructions -- so the product of the
// MFMA path below (instruction (tt, j) contracts k = 16 tt + 4 g + j over g = 0..3) is the fmaf chain
// over (tt, j, g) in that order, whoever computes it.  With MR rows of the 16 an MFMA forms, the matrix
// unit is 1/16 .. 1/8 used and a wavefront runs 32 dependent instructions per 16 columns (17 column
// blocks for the ShadowHand head: three rounds on 8 wavefronts, 1.5 us); one lane per COLUMN runs the 128
// fmas of its column once (32 conflict-free 16-byte reads of its row of Wh, h2 broadcast): 0.4 us.
template <int MR>
__device__ __forceinline__ void heads_fma_chain(const float* H2s, const float* Whs, int n, float (&acc)[MR]) {
  const float* bp = Whs + n * kMH;
  const int sw = 4 * (n & 15);
#pragma unroll
  for (int r = 0; r < MR; ++r) acc[r] = 0.f;
#pragma unroll
  for (int tt = 0; tt < 8; ++tt) {
    f32x4 b[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) b[g] = *reinterpret_cast<const f32x4*>(bp + ((16 * tt + 4 * g) ^ sw));
#pragma unroll
    for (int r = 0; r < MR; ++r) {
      f32x4 a[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) a[g] = *reinterpret_cast<const f32x4*>(H2s + r * kMHP + 16 * tt + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[r] = __builtin_fmaf(a[g][j], b[g][j], acc[r]);
    }
  }
}

// The row-wise phase of an owner on the lean two-wavefront row (head_device.h: diag_row_fast_core): the
// lane's elements from the head-output row image `tile` (Os), the target from `yv`; writes the gradients of
// ITS elements (without the jitter-scale term) back into the image, the logit gradients from wavefront 0.
// ro: lse, this wavefront's lanes' partial sums of u dL/dsigma, esg0[i] = exp(pre) of sweep hw + 2 i.
// `xch`: the pair's exchange, [KP][8] floats (+ 4 KP 8: the log-determinants).
template <int KP, int NQH, typename Eps>
__device__ __forceinline__ void mdnn_fast_row(const MdnnArgs& p, int K, int D, bool active, int lane, int hw,
                                              float* tile, const float* yv, float* xch, const float* eu_pre,
                                              Eps& ge, RowOut& ro) {
  constexpr int GR = 64 / KP;
  const int DK = D * K;
  const int k = lane & (KP - 1), d0 = lane / KP;
  const bool kok = k < K;
  bool valid[NQH];
  float muv[NQH], yd[NQH], eu[NQH], ev[NQH], dmu[NQH], dpre[NQH];
  int ecol[NQH];
#pragma unroll
  for (int i = 0; i < NQH; ++i) {
    const int d = d0 + (hw + 2 * i) * GR;
    valid[i] = active && kok && d < D;
    ecol[i] = min(d, D - 1) * K + min(k, K - 1);
    ev[i] = tile[K + DK + ecol[i]];
    muv[i] = tile[K + ecol[i]];
    yd[i] = yv[min(d, D - 1)];
    eu[i] = eu_pre[i];
  }
  const float lg_own = tile[min(k, K - 1)];
  ge.issue();
#pragma unroll
  for (int i = 0; i < NQH; ++i) ev[i] = valid[i] ? expf(ev[i]) : 0.f;
  float dlogit = 0.f, lse = 0.f, uds = 0.f;
  bool bad = false;
  float* xq = xch + k * 8;
  diag_row_fast_core<KP, NQH>(K, D, active, kok, valid, muv, yd, eu, ev, lg_own, p.min_w, p.ll_limit, p.inv_norm,
                              xq, xq + 4 * KP * 8, hw, lane, ge, [] { lds_barrier(); }, dmu, dpre, dlogit, lse, uds, bad);
#pragma unroll
  for (int i = 0; i < NQH; ++i) {
    ro.esg0[i] = ev[i];
    if (valid[i]) { tile[K + ecol[i]] = dmu[i]; tile[K + DK + ecol[i]] = dpre[i]; }
  }
  if (active && hw == 0 && lane < KP && kok) tile[k] = dlogit;
  __builtin_amdgcn_wave_barrier();
  ro.lse = lse; ro.uds = uds; ro.bad = bad;
}

template <bool DP, bool WIDE, bool FULL, int MR = kMR>
__device__ __forceinline__ void mdnn_owner_workgroup(const MdnnArgs& p, float* smem) {
  const int Nh = p.Nh, Nh16 = p.Nh16, D = p.D, K = p.K, DK = D * K, B = p.B;
  const int po = Nh16 + 4;                   // pitch of a head-output row
  const int per_wave = D + 3 * K + (FULL ? 3 * DK : 0);
  // rows of a 16-row MFMA result this owner keeps (lanes g == 0 hold result rows 0..3: with fewer than
  // four rows per owner the others repeat its rows -- `rowA` wraps -- and belong to nobody here)
  constexpr int RQ = MR < 4 ? MR : 4;
  // FAST (round 6): the row-wise phase on two wavefronts per row with the lean row of head_device.h
  // (diag_row_fast_core: component count padded to 4 / 8 / 16 lanes, DPP reductions) -- diagonal
  // covariance, at most four rows per owner (eight wavefronts), K <= 16, at most 8 sweeps; the pairs'
  // exchange lives where the k-slice sums' partial quads were (free since h1).  fk = KP * 8 + sweeps per
  // wavefront, 0: the shape-generic row (BSIG_MDNN_FAST_ROWS=0).
  constexpr bool PAIR = !FULL && MR <= 4;
  int fk = 0;
  if constexpr (PAIR) {
    // (first layers wider than kFastRowsMaxInput inputs keep the shape-generic row: they amplify an ulp of the
    // row arithmetic to ~1e-4 of held-out NLL within a chunk -- cfg3 I = 11802, cfg4b I = 11154: on their
    // ill-conditioned seeds ANY change of summation order moves the result by more than the north-star
    // tolerance, the reference's own fp32 path included (profiles/r06_NOTES.md) -- and the agreement of
    // round 5's row with the reference on six seeds each is what their parity tests pinned)
    if (K <= 16 && p.fast_rows && (p.I <= kFastRowsMaxInput || p.fast_rows == 2)) {
      const int KPr = K <= 4 ? 4 : (K <= 8 ? 8 : 16), nq = (D + 64 / KPr - 1) / (64 / KPr);
      if (nq <= 8) fk = KPr * 8 + (nq <= 2 ? 1 : (nq <= 4 ? 2 : 4));
    }
  }
  float* Whs = smem;                         // [Nh16][128], element (n, i) at n*128 + (i ^ 4(n & 15))
  float* H1s = Whs + (WIDE ? 0 : Nh16 * kMH);   // [MR][kMHP]  (wide heads: no head matrix here)
  float* H2s = H1s + MR * kMHP;             // [MR][kMHP]  h2, later dz2 in place
  float* Os = H2s + MR * kMHP;              // [MR][po]    head outputs, later d_out in place
  float* b2s = Os + MR * po;                // [128]
  float* bhs = b2s + kMH;                    // [Nh16]
  float* wsc = bhs + Nh16;                   // [MR][D + 3K] per-row scratch of diag_row
  float* red = wsc + MR * per_wave;         // [64]
  float* part4 = red + 64;                   // [((kMT / (MR * (kMH / 4))) - 1) * (MR * (kMH / 4)) * 4]  partial k-slice sums
  const int tid_0 = threadIdx.x, w_0 = __builtin_amdgcn_readfirstlane(tid_0 >> 6);
  const int c16_0 = tid_0 & 15, g_0 = (tid_0 & 63) >> 4;
  const int o = blockIdx.x - p.G1;
  const int r0 = o * MR;
  int32_t* flagp = p.state + 2;
  const int step0 = p.state[0];
  const uint64_t rng_seed = reinterpret_cast<const uint64_t*>(p.state + 8)[0];
  const uint64_t rng_ctr0 = reinterpret_cast<const uint64_t*>(p.state + 8)[1];
  HeadArgs a{};
  a.D = D; a.K = K; a.Nh = Nh; a.batch = B; a.from_tuple = 0;
  a.Ls = FULL ? D * (D - 1) / 2 : 0;
  a.min_w = p.min_w; a.ll_limit = p.ll_limit; a.inv_norm = p.inv_norm;
  a.eps_noise = p.eps_noise; a.seed = rng_seed; a.d_out = p.d_out;
  const float norm = (float)B * (float)DK;
  // (wave w <-> minibatch row r0 + w in the row-wise phases)
  const float* Wh = p.params + p.wh_off;
  const int64_t zs = (int64_t)B * kMH;
  // W2 is the B operand of two products, one column block of 16 per wave, straight
  // from registers: w2f[4t + j] = W2[16w + c16][16t + 4g + j] (forward, fetched under the
  // k-slice sum), w2b[4t + j] = W2[16t + 4g + j][16w + c16] (backward, fetched under the
  // wait for the other owners' rows)


  // ---- held-out evaluation number eidx (jitter stream `stream`), forward only: per pass,
  //      this owner's 4 rows of the pass go through layers 2.. (the weights in LDS / the
  //      W2 pack are the evaluated ones); head outputs are parked in memory until the
  //      batch-wide sum of exp(pre) is known, then one wavefront per row takes the NLL
  auto owner_eval = [&](int eidx, uint64_t stream, int step, bool refresh) {
    int c16 = c16_0, g = g_0, tid = tid_0;
    const int w = w_0;
    asm volatile("" : "+v"(c16), "+v"(g), "+v"(tid));
    const int lane = tid & 63, rowA = c16 & (MR - 1);
    const unsigned etag = (unsigned)eidx + 1u;
    float* tile = Os + (w & (MR - 1)) * po;
    float* yv = wsc + (w & (MR - 1)) * per_wave;
    float* rk = yv + D;
    float* lpk = rk + K;
    float* dlg = lpk + K;
    if (refresh) {
      // after the last update of the call: the head matrix and the biases in LDS are those
      // of that update -- fetch what its Adam step published
      // (a data-parallel rank has them behind flag_pack, waited for at the top)
      if (!DP && w == 0) flags_wait(p.flag_small, p.n_small, (unsigned)step, lane, flagp);
      __syncthreads();
      if constexpr (!WIDE) {
        const __amdgpu_buffer_rsrc_t whr = xwg_buffer(Wh);
        for (int idx = tid; idx < Nh16 * (kMH / 4); idx += kMT) {
          const int n = idx >> 5, c4 = (idx & 31) * 4;
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(Whs + n * kMH + (c4 ^ (4 * (n & 15)))) = n < Nh ? xwg_load4(whr, n * kMH + c4) : zero;
        }
      }
      if (tid < kMH) b2s[tid] = xwg_load(p.params + p.b2_off + tid);
      if constexpr (!WIDE) {
        for (int j = tid; j < Nh16; j += kMT) bhs[j] = j < Nh ? xwg_load(p.params + p.bh_off + j) : 0.f;
      }
    }
    if (w == 0) flags_wait(p.flag_eval, p.G1, etag, lane, flagp);
    __syncthreads();
    float w2f[32];
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const float* src = p.w2f_pack + (step & 1) * (kMH * kMH) + (((w * 8 + tt) * 2) * 64 + lane) * 2;
      const float2 lo = xwg_load2(src), hi = xwg_load2(src + 128);
      w2f[4 * tt + 0] = lo.x; w2f[4 * tt + 1] = lo.y; w2f[4 * tt + 2] = hi.x; w2f[4 * tt + 3] = hi.y;
    }
    float eacc = 0.f;
    for (int gp = 0; gp < p.eval_passes; ++gp) {
      // (streamed W1: the tile workgroups have summed the pass already -- one slab)
      const int ek = p.stream ? 1 : p.k_slices;
      const float* slabs = p.eval_slabs + (((int64_t)(eidx & 1) * p.eval_passes + gp) * ek) * zs;
      {
        const f32x4 v = slab_quads_sum<MR>(slabs, ek, (int)zs, min(r0 + (tid >> 5 & (MR - 1)), B - 1), tid, part4);
        if (tid < (MR * (kMH / 4))) {
          const int r = tid >> 5, c4 = (tid & 31) * 4;
          const bool ok = r0 + r < B && gp * B + r0 + r < p.n_test;
          float* hl = H1s + r * kMHP + c4;
          hl[0] = ok ? tanhf(v.x) : 0.f; hl[1] = ok ? tanhf(v.y) : 0.f;
          hl[2] = ok ? tanhf(v.z) : 0.f; hl[3] = ok ? tanhf(v.w) : 0.f;
        }
      }
      __syncthreads();
      {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* ap = H1s + rowA * kMHP + 4 * g;
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * tt);
          acc = mfma16(a4.x, w2f[4 * tt + 0], acc);
          acc = mfma16(a4.y, w2f[4 * tt + 1], acc);
          acc = mfma16(a4.z, w2f[4 * tt + 2], acc);
          acc = mfma16(a4.w, w2f[4 * tt + 3], acc);
        }
        if (4 * g < MR) {
          const int n = 16 * w + c16;
          const float bias = b2s[n];
          if constexpr (WIDE) {
#pragma unroll
            for (int r = 0; r < RQ; ++r) {
              const float v = tanhf(acc[r] + bias);
              H2s[(4 * g + r) * kMHP + n] = v;
              if (r0 + 4 * g + r < B) xwg_store(p.h2e + (int64_t)(r0 + 4 * g + r) * kMH + n, v);
            }
          } else {
#pragma unroll
            for (int r = 0; r < RQ; ++r) H2s[(4 * g + r) * kMHP + n] = tanhf(acc[r] + bias);
          }
        }
      }
      if constexpr (WIDE) __builtin_amdgcn_s_waitcnt(0);   // h2 rows are out before the flag
      __syncthreads();
      if constexpr (WIDE) {
        // wide heads: the head-block workgroups form the pass's head outputs (with the weights of the
        // evaluated update: they serve this before they refresh their operand copies)
        eacc += mdnn_owner_eval_heads<MR>(p.flag_h2e, p.flag_oe, p.oe, flagp, Os, po, o, p.n_hb, B, r0, Nh, Nh16,
                                          K, DK, p.n_test, gp, eidx);
      }
      if constexpr (!WIDE && MR <= 2) {     // (one lane per column: heads_fma_chain)
        for (int n = tid; n < Nh16; n += kMT) {
          float hacc[MR];
          heads_fma_chain<MR>(H2s, Whs, n, hacc);
          const float bias = bhs[n];
#pragma unroll
          for (int r = 0; r < MR; ++r) {
            const float v = hacc[r] + bias;
            Os[r * po + n] = v;
            if (r0 + r < B && gp * B + r0 + r < p.n_test && n >= K + DK && n < K + 2 * DK) eacc += expf(v);
          }
        }
      }
      for (int cb = w; !WIDE && MR > 2 && cb * 16 < Nh16; cb += 8) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int n = 16 * cb + c16;
        const float* ap = H2s + rowA * kMHP + 4 * g;
        const float* bp = Whs + n * kMH;
        const int sw = 4 * c16;
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
          const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * tt);
          const float4 b4 = *reinterpret_cast<const float4*>(bp + ((16 * tt + 4 * g) ^ sw));
          acc = mfma16(a4.x, b4.x, acc);
          acc = mfma16(a4.y, b4.y, acc);
          acc = mfma16(a4.z, b4.z, acc);
          acc = mfma16(a4.w, b4.w, acc);
        }
        if (4 * g < MR) {
          const float bias = bhs[n];
#pragma unroll
          for (int r = 0; r < RQ; ++r) {
            const int rr = 4 * g + r;
            const float v = acc[r] + bias;
            Os[rr * po + n] = v;
            if (r0 + rr < B && gp * B + r0 + rr < p.n_test && n >= K + DK && n < K + 2 * DK) eacc += expf(v);
          }
        }
      }
      __syncthreads();
      for (int idx = tid; idx < MR * Nh; idx += kMT) {
        const int r = idx / Nh, j = idx - r * Nh;
        const int erow = gp * B + r0 + r;
        if (r0 + r < B && erow < p.n_test) xwg_store(p.eval_out + (int64_t)erow * p.NhP + j, Os[r * po + j]);
      }
      __syncthreads();
    }
    eacc = wave_sum_dpp(eacc);
    if (lane == 0) red[w] = eacc;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
      float sx = 0.f;
      for (int q = 0; q < kMT / 64; ++q) sx += red[q];
      granule_publish(p.gran_eval, o, etag, sx);
    }
    HeadArgs ae = a;
    ae.d_out = nullptr;                      // forward only
    ae.batch = p.n_test;
    ae.stream_id = stream;
    float lse_acc = 0.f;
    bool bad = false;
    for (int gp = 0; gp < p.eval_passes; ++gp) {
      for (int idx = tid; idx < MR * Nh; idx += kMT) {
        const int r = idx / Nh, j = idx - r * Nh;
        const int erow = gp * B + r0 + r;
        Os[r * po + j] = (r0 + r < B && erow < p.n_test) ? xwg_load(p.eval_out + (int64_t)erow * p.NhP + j) : 0.f;
      }
      const int erow = gp * B + r0 + w;
      const bool act = w < MR && r0 + w < B && erow < p.n_test;
      if (act)
        for (int j = lane; j < D; j += 64) yv[j] = p.y_test[(int64_t)erow * p.ldy_test + j];
      __syncthreads();
      if (w < MR) {
        RowOut ro;
        ro.lse = 0.f; ro.uds = 0.f; ro.bad = false;
#pragma unroll
        for (int q = 0; q < kElemsPerLane; ++q) ro.esg0[q] = 0.f;
        auto eval_eps = [&] {
          return p.eps_noise != 0.f
                     ? p.eps_noise * (granule_gather(p.gran_eval, p.n_owner, etag, lane, flagp) /
                                      ((float)p.n_test * (float)DK))
                     : 0.f;
        };
        if constexpr (FULL) full_row(ae, erow, act, lane, tile, yv, rk, dlg, dlg + K, eval_eps, ro);
        else diag_row(ae, erow, act, lane, tile, yv, rk, lpk, dlg, eval_eps, ro);
        if (act) lse_acc += ro.lse;
        bad |= ro.bad;
      }
      __syncthreads();
    }
    if (lane == 0) red[16 + w] = lse_acc;
    __syncthreads();
    if (tid == 0) {
      float sl = 0.f;
      for (int q = 0; q < MR; ++q) sl += red[16 + q];
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
    if (bad) atomicOr(flagp, 1);
    __syncthreads();
  };
  const int ev0 = p.do_eval ? mdnn_evals_before(step0, p.eval_every) : 0;
  const RowGeom rg0 = row_geom(D, K, tid_0 & 63);   // (integer divisions by run-time values: once per launch)

  if (w_0 == 0) flags_wait(p.flag_pack, p.n_small, p.launch_tag, tid_0 & 63, flagp);
  __syncthreads();
  for (int t = 0; t < p.n_updates; ++t) {
    // lane-derived indices are laundered once per update so that the address
    // arithmetic built on them is recomputed, not kept live across the update loop
    int c16 = c16_0, g = g_0, w = w_0, tid = tid_0;
    asm volatile("" : "+v"(c16), "+v"(g), "+v"(tid));
    asm volatile("" : "+s"(w));
    const int lane = tid & 63, rowA = c16 & (MR - 1);
    // (fast rows: wavefronts 2r, 2r + 1 run row r0 + r in the row-wise phase; else wavefront w runs row r0 + w.
    // The target row is loaded by wavefront r < MR either way: `yrow_w`.)
    const bool fast = PAIR && fk != 0;
    const int rw = fast ? w >> 1 : w, hw = fast ? w & 1 : 0;
    const bool row_wave = w < (fast ? 2 * MR : MR);
    const int row = r0 + rw;
    const bool active = row_wave && row < B;
    float* tile = Os + (rw & (MR - 1)) * po;
    float* yv = wsc + (rw & (MR - 1)) * per_wave;
    float* rk = yv + D;
    float* lpk = rk + K;
    float* dlg = lpk + K;
    const int step = step0 + t;
    const unsigned epoch = (unsigned)step + 1u;
    const uint32_t tag = epoch * 4u;
    if (run_aborted(flagp, red, tid)) break;
    BSIG_MSTAMP(0);
    // ---- weights of this update (written by the small-weight workgroups) ---------
    // (first update of a launch, and the only one of a data-parallel launch: flag_pack above)
    // (wide heads: the owners only need W2 / b2 -- the first four small-weight workgroups)
    if (w < MR && r0 + w < B) {      // target row: two dependent loads, issued before the wait instead of after the weights
      const int64_t yrow = p.ids[(int64_t)step * B + r0 + w];
      float* yw = wsc + w * per_wave;
      for (int j = lane; j < D; j += 64) yw[j] = p.y[yrow * p.ldy + j];
    }
    // one jitter stream per update and per evaluation, in program order; the row's draws do not depend
    // on the forward product: taken here, in the wait
    a.stream_id = rng_ctr0 + (uint64_t)t +
                  (uint64_t)(p.do_eval ? mdnn_evals_before(step, p.eval_every) - ev0 : 0);
    float eu_pre[kElemsPerLane];
    if constexpr (!FULL) {
      if (fast) {
        // the draw of element (d, k) is diag_row_noise's (the generic row's lane (d % G) * K + k, sweep d / G);
        // eu_pre[i]: sweep hw + 2 i of this wavefront in the padded lane geometry
        const int KPr = fk >> 3, GRr = 64 / KPr, kk = lane & (KPr - 1), dd0 = lane / KPr, G = rg0.groups;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int d = dd0 + (hw + 2 * i) * GRr;
          eu_pre[i] = 0.f;
          if (active && kk < K && d < D && a.eps_noise != 0.f && i < (fk & 7)) {
            const int q = d / G, gl = (d - q * G) * K + kk;
            const Philox4 ph = philox4x32_10(a.seed, a.stream_id, ((uint64_t)row * 64 + gl) * 2 + (q >> 2));
            const int oi = q & 3;
            eu_pre[i] = u01(oi == 0 ? ph.v[0] : (oi == 1 ? ph.v[1] : (oi == 2 ? ph.v[2] : ph.v[3])));
          }
        }
      } else if (w < MR) {
        diag_row_noise(a, rg0.groups, rg0.k, rg0.d0, row, active, lane, eu_pre);
      }
    }
    if (!DP && t > 0 && w == 0)
      flags_wait(p.flag_small, WIDE ? kMH / kMNB : p.n_small, epoch - 1u, lane, flagp);
    __syncthreads();
    BSIG_MSTAMP(4);
    // The head matrix refresh sits on the owners' critical path (a CU pulls ~40 GB/s through
    // cache-bypassing loads, tools/micro/fanout_bench.hip: 73 KB of a 144-row matrix = 2 us), and the
    // first-layer flags have usually been up for a while when it ends: wavefront 0 polls them while
    // wavefronts 1..7 fetch (kWhFlight 16-byte loads in flight per thread).
    // (this wavefront's W2 operand registers ride in the same window)
    float w2f[32];
#define BSIG_LOAD_W2F()                                                                        \
    _Pragma("unroll") for (int tt = 0; tt < 8; ++tt) {                                          \
      const float* src = p.w2f_pack + (step & 1) * (kMH * kMH) + (((w * 8 + tt) * 2) * 64 + lane) * 2; \
      const float2 lo = xwg_load2(src), hi = xwg_load2(src + 128);                              \
      w2f[4 * tt + 0] = lo.x; w2f[4 * tt + 1] = lo.y; w2f[4 * tt + 2] = hi.x; w2f[4 * tt + 3] = hi.y; \
    }
    if (w == 0) {
      // (no quiet wait here: the other wavefronts are pulling the head matrix through this CU's
      // memory pipe, a sentinel poll would queue behind it and cost a second round trip -- measured:
      // flags seen 2.8 us after the last one went up instead of 0.9)
      flags_wait(p.o_flags, p.G1, epoch, lane, flagp);
      BSIG_LOAD_W2F()
    } else {
      BSIG_LOAD_W2F()
      constexpr int kWhFlight = 11, kFetch = kMT - 64;
      const int ft = tid - 64;
      const __amdgpu_buffer_rsrc_t whr = xwg_buffer(Wh);
      for (int base = 0; base < (WIDE ? 0 : Nh16 * (kMH / 4)); base += kFetch * kWhFlight) {
        f32x4 q[kWhFlight];
#pragma unroll
        for (int u = 0; u < kWhFlight; ++u) {
          const int idx = base + u * kFetch + ft;
          const int n = idx >> 5;
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          q[u] = n < Nh ? xwg_load4(whr, n * kMH + (idx & 31) * 4) : zero;
        }
#pragma unroll
        for (int u = 0; u < kWhFlight; ++u) {
          const int idx = base + u * kFetch + ft;
          const int n = idx >> 5, c4 = (idx & 31) * 4;
          if (n < Nh16) *reinterpret_cast<f32x4*>(Whs + n * kMH + (c4 ^ (4 * (n & 15)))) = q[u];
        }
      }
      if (ft < kMH) b2s[ft] = xwg_load(p.params + p.b2_off + ft);
      for (int j = ft; j < Nh16; j += kFetch) bhs[j] = j < Nh ? xwg_load(p.params + p.bh_off + j) : 0.f;
    }
    // ---- h1 = tanh(sum of the k-slices) (b1 rides on slice 0) -----------------------
    BSIG_MSTAMP(5);
    lds_barrier();                             // (the W2 operand registers stay in flight)
    BSIG_MSTAMP(6);
    {
      // (row, column pair) items; kSub threads share an item's k-slices, their partial
      // sums are combined in a fixed order through LDS
      const f32x4 v = slab_quads_sum<MR>(p.o_slabs, p.o_k_slices, (int)zs, min(r0 + (tid >> 5 & (MR - 1)), B - 1), tid, part4);
      if (tid < (MR * (kMH / 4))) {
        const int r = tid >> 5, c4 = (tid & 31) * 4;
        const bool ok = r0 + r < B;
        const float h0 = ok ? tanhf(v.x) : 0.f, h1v = ok ? tanhf(v.y) : 0.f;
        const float h2v = ok ? tanhf(v.z) : 0.f, h3 = ok ? tanhf(v.w) : 0.f;
        float* hl = H1s + r * kMHP + c4;
        hl[0] = h0; hl[1] = h1v; hl[2] = h2v; hl[3] = h3;
        if (ok) xwg_store4(xwg_buffer(p.h1), (r0 + r) * kMH + c4, h0, h1v, h2v, h3);
      }
    }
    lds_barrier();                             // (h1 rows still on their way out)
    BSIG_MSTAMP(7);
    // ---- h2 = tanh(h1 W2^T + b2): wave w -> columns 16w .. 16w+15 -------------------
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* ap = H1s + rowA * kMHP + 4 * g;
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * tt);
        acc = mfma16(a4.x, w2f[4 * tt + 0], acc);
        acc = mfma16(a4.y, w2f[4 * tt + 1], acc);
        acc = mfma16(a4.z, w2f[4 * tt + 2], acc);
        acc = mfma16(a4.w, w2f[4 * tt + 3], acc);
      }
      if (4 * g < MR) {
        const int n = 16 * w + c16;
        const float bias = b2s[n];
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
          const int rr = 4 * g + r;
          const float v = tanhf(acc[r] + bias);
          H2s[rr * kMHP + n] = v;
          if (r0 + rr < B) xwg_store(p.h2 + (int64_t)(r0 + rr) * kMH + n, v);
        }
      }
    }
    if constexpr (WIDE) __builtin_amdgcn_s_waitcnt(0);   // h2 rows are out before the flag
    lds_barrier();
    BSIG_MSTAMP(8);
    float eacc = 0.f;
    if constexpr (WIDE) {
      // ---- wide heads: the head-block workgroups form h2 Wh^T + bh for all rows ------------
      if (tid == 0) flag_raise(p.flag_h2, o, epoch);
      if (w == 0) flags_wait(p.flag_o, p.n_hb, epoch, lane, flagp);
      __syncthreads();
      {
        // the owner's MR rows of every head block [n_hb][B][32] as 16-byte quads, four per thread in
        // flight (a conditional 4-byte load per element came back one round trip at a time: ten
        // dependent ~0.8 us trips with 8 rows per owner)
        const int per_row = p.n_hb * (kMNB / 4), n_items = MR * per_row;
        const __amdgpu_buffer_rsrc_t orr = xwg_buffer(p.o_wide);
        for (int base = 0; base < n_items; base += 4 * kMT) {
          f32x4 q[4];
          int rr[4], jj[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int idx = min(base + u * kMT + tid, n_items - 1);
            const int r = idx / per_row, rem = idx - r * per_row;
            rr[u] = r; jj[u] = (rem >> 3) * kMNB + (rem & 7) * 4;
            q[u] = xwg_load4(orr, ((rem >> 3) * B + min(r0 + r, B - 1)) * kMNB + (rem & 7) * 4);
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (base + u * kMT + tid < n_items) {
              const bool rok = r0 + rr[u] < B;
              const float v4[4] = {q[u].x, q[u].y, q[u].z, q[u].w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int j = jj[u] + e;
                const float v = rok && j < Nh ? v4[e] : 0.f;
                if (j < Nh16) Os[rr[u] * po + j] = v;
                if (rok && j >= K + DK && j < K + 2 * DK) eacc += expf(v);
              }
            }
          }
        }
      }
    }
    // ---- head outputs = h2 Wh^T + bh: one lane per column (one or two rows per owner), else
    //      column blocks w, w+8, w+16 on the MFMA units ------------------------------------
    if constexpr (!WIDE && MR <= 2) {
      for (int n = tid; n < Nh16; n += kMT) {
        float hacc[MR];
        heads_fma_chain<MR>(H2s, Whs, n, hacc);
        const float bias = bhs[n];
#pragma unroll
        for (int r = 0; r < MR; ++r) {
          const float v = hacc[r] + bias;
          Os[r * po + n] = v;
          if (r0 + r < B && n >= K + DK && n < K + 2 * DK) eacc += expf(v);
        }
      }
    }
    for (int cb = w; !WIDE && MR > 2 && cb * 16 < Nh16; cb += 8) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int n = 16 * cb + c16;
      const float* ap = H2s + rowA * kMHP + 4 * g;
      const float* bp = Whs + n * kMH;
      const int sw = 4 * c16;                      // = 4 * (n & 15)
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * tt);
        const float4 b4 = *reinterpret_cast<const float4*>(bp + ((16 * tt + 4 * g) ^ sw));
        acc = mfma16(a4.x, b4.x, acc);
        acc = mfma16(a4.y, b4.y, acc);
        acc = mfma16(a4.z, b4.z, acc);
        acc = mfma16(a4.w, b4.w, acc);
      }
      if (4 * g < MR) {
        const float bias = bhs[n];
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
          const int rr = 4 * g + r;
          const float v = acc[r] + bias;
          Os[rr * po + n] = v;
          if (r0 + rr < B && n >= K + DK && n < K + 2 * DK) eacc += expf(v);
        }
      }
    }
    eacc = wave_sum_dpp(eacc);
    if (lane == 0) red[w] = eacc;
    lds_barrier();
    if (tid == 0) {
      float sx = 0.f;
      for (int q = 0; q < kMT / 64; ++q) sx += red[q];
      granule_publish(p.gran, o, tag + 1, sx);
    }
    BSIG_MSTAMP(9);
    // ---- row-wise NLL forward / backward (one wavefront per row) -----------------------
    RowOut ro;
    ro.lse = 0.f; ro.uds = 0.f; ro.bad = false;
#pragma unroll
    for (int q = 0; q < kElemsPerLane; ++q) ro.esg0[q] = 0.f;
    if constexpr (FULL) {
      auto row_eps = [&] {
        return p.eps_noise != 0.f
                   ? p.eps_noise * (granule_gather(p.gran, p.n_owner, tag + 1, lane, flagp) / norm)
                   : 0.f;
      };
      full_row(a, row, active, lane, tile, yv, rk, dlg, dlg + K, row_eps, ro);
    } else {
      // (lane geometry from before the loop, jitter draws from the wait, the jitter-scale gather issued
      // ahead of the arithmetic that does not need it: head_device.h)
      GranuleEps ge{p.gran, p.n_owner, tag + 1, lane, flagp, p.eps_noise, norm, {0ull, 0ull, 0ull, 0ull}};
      if (fast) {
        if (row_wave) {
          switch (fk) {
            case 4 * 8 + 1: mdnn_fast_row<4, 1>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 4 * 8, eu_pre, ge, ro); break;
            case 4 * 8 + 2: mdnn_fast_row<4, 2>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 4 * 8, eu_pre, ge, ro); break;
            case 4 * 8 + 4: mdnn_fast_row<4, 4>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 4 * 8, eu_pre, ge, ro); break;
            case 8 * 8 + 1: mdnn_fast_row<8, 1>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 8 * 8, eu_pre, ge, ro); break;
            case 8 * 8 + 2: mdnn_fast_row<8, 2>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 8 * 8, eu_pre, ge, ro); break;
            case 8 * 8 + 4: mdnn_fast_row<8, 4>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 8 * 8, eu_pre, ge, ro); break;
            case 16 * 8 + 1: mdnn_fast_row<16, 1>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 16 * 8, eu_pre, ge, ro); break;
            case 16 * 8 + 2: mdnn_fast_row<16, 2>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 16 * 8, eu_pre, ge, ro); break;
            default: mdnn_fast_row<16, 4>(p, K, D, active, lane, hw, tile, yv, part4 + (rw & (MR - 1)) * 16 * 8, eu_pre, ge, ro); break;
          }
        } else {
          lds_barrier();                     // (the row wavefronts' exchange of partial sums)
        }
      } else {
        diag_row_impl(a, rg0, row, active, lane, tile, yv, rk, lpk, dlg, ge, ro, eu_pre);
      }
    }
    {
      const float uds_w = wave_sum_dpp(ro.uds);
      if (lane == 0) { red[16 + w] = active && hw == 0 ? ro.lse : 0.f; red[32 + w] = uds_w; }
    }
    lds_barrier();
    if (tid == 0) {
      float sl = 0.f, su = 0.f;
      for (int q = 0; q < (fast ? 2 * MR : MR); ++q) { sl += red[16 + q]; su += red[32 + q]; }
      granule_publish(p.gran + kGranArr, o, tag + 2, su);
      granule_publish(loss_granules(p.gran, epoch), o, tag + 3, sl);
    }
    BSIG_MSTAMP(13);
    float w2b[32];
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
      const float* src = p.w2b_pack + (step & 1) * (kMH * kMH) + (((w * 8 + tt) * 2) * 64 + lane) * 2;
      const float2 lo = xwg_load2(src), hi = xwg_load2(src + 128);
      w2b[4 * tt + 0] = lo.x; w2b[4 * tt + 1] = lo.y; w2b[4 * tt + 2] = hi.x; w2b[4 * tt + 3] = hi.y;
    }
    // the jitter-scale gradient term d pre += (EPS/(B*D*K)) * sum(u*dL/dsigma) * exp(pre)
    // needs the sum over the whole minibatch
    {
      float c = 0.f;
      if (p.eps_noise != 0.f)
        c = p.eps_noise / norm * granule_gather(p.gran + kGranArr, p.n_owner, tag + 2, lane, flagp);
      if (active) {
        if constexpr (FULL) {
          // exp(pre) of the row: third block of full_row's scratch, [d][k] like the head outputs
          const float* sg0 = dlg + K + 2 * DK;
          if (c != 0.f)
            for (int j = lane; j < DK; j += 64) tile[K + DK + j] += c * sg0[j];
        } else {
          if (fast) {
            // (each wavefront of the pair: the elements of ITS sweeps -- the ones it wrote -- and their
            // d_out columns; the logit gradients are already in the row image, wavefront 0 stores them)
            const int KPr = fk >> 3, GRr = 64 / KPr, kk = lane & (KPr - 1), dd0 = lane / KPr;
            float* dst = p.d_out + (int64_t)row * p.NhP;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int d = dd0 + (hw + 2 * i) * GRr;
              if (i < (fk & 7) && kk < K && d < D) {
                const int e = d * K + kk;
                const float dp = tile[K + DK + e] + c * ro.esg0[i];
                tile[K + DK + e] = dp;
                xwg_store(dst + K + DK + e, dp);
                xwg_store(dst + K + e, tile[K + e]);
              }
            }
            if (hw == 0 && lane < K) xwg_store(dst + lane, tile[lane]);
          } else {
          const int groups = 64 / K, TPR = groups * K;
          const int k = lane % K, d0 = lane / K;
          if (c != 0.f && lane < TPR) {
#pragma unroll
            for (int q = 0; q < kElemsPerLane; ++q) {
              const int d = d0 + q * groups;
              if (d < D) tile[K + DK + d * K + k] += c * ro.esg0[q];
            }
          }
          }
        }
        if (!fast) {
          __builtin_amdgcn_wave_barrier();
          for (int j = lane; j < K; j += 64) tile[j] = dlg[j];
          __builtin_amdgcn_wave_barrier();
          float* dst = p.d_out + (int64_t)row * p.NhP;
          for (int j = lane; j < Nh; j += 64) xwg_store(dst + j, tile[j]);
        }
      }
    }
    if (ro.bad) atomicOr(flagp, 1);
    if constexpr (WIDE) __builtin_amdgcn_s_waitcnt(0);   // d_out rows are out before the flag
    lds_barrier();
    BSIG_MSTAMP(14);
    if constexpr (WIDE) {
      // ---- wide heads: dz2 = (sum of the head blocks' shares of d_out Wh) * (1 - h2^2) -------
      if (tid == 0) flag_raise(p.flag_dout, o, epoch);
      if (w == 0) flags_wait(p.flag_dz2, p.n_hb, epoch, lane, flagp);
      __syncthreads();
      if (tid < MR * 64) {
        const int r = tid >> 6, c2 = (tid & 63) * 2;
        const float* src = p.dz2_part + (int64_t)min(r0 + r, B - 1) * kMH + c2;
        float vx = 0.f, vy = 0.f;
        for (int z = 0; z < p.n_hb; z += 8) {
          float2 q[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) q[u] = xwg_load2(src + (int64_t)min(z + u, p.n_hb - 1) * zs);
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (z + u < p.n_hb) { vx += q[u].x; vy += q[u].y; }
        }
        const float hx = H2s[r * kMHP + c2], hy = H2s[r * kMHP + c2 + 1];
        vx *= 1.0f - hx * hx; vy *= 1.0f - hy * hy;
        H2s[r * kMHP + c2] = vx; H2s[r * kMHP + c2 + 1] = vy;
        if (r0 + r < B) {
          xwg_store(p.dz2 + (int64_t)(r0 + r) * kMH + c2, vx);
          xwg_store(p.dz2 + (int64_t)(r0 + r) * kMH + c2 + 1, vy);
        }
      }
    }
    // ---- dz2 = (d_out Wh) * (1 - h2^2): wave w -> columns 16w .. 16w+15 ----------------
    if constexpr (!WIDE) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int i = 16 * w + c16;
      const float* ap = Os + rowA * po + 4 * g;
      for (int tt = 0; tt * 16 < Nh16; ++tt) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * tt);
        const float* bp = Whs + (16 * tt + 4 * g) * kMH;
        // rows 16tt + 4g + j: (row & 15) = 4g + j
        const float f0 = bp[0 * kMH + (i ^ (4 * (4 * g + 0)))];
        const float f1 = bp[1 * kMH + (i ^ (4 * (4 * g + 1)))];
        const float f2 = bp[2 * kMH + (i ^ (4 * (4 * g + 2)))];
        const float f3 = bp[3 * kMH + (i ^ (4 * (4 * g + 3)))];
        acc = mfma16(a4.x, f0, acc);
        acc = mfma16(a4.y, f1, acc);
        acc = mfma16(a4.z, f2, acc);
        acc = mfma16(a4.w, f3, acc);
      }
      if (4 * g < MR) {
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
          const int rr = 4 * g + r;
          const float hv = H2s[rr * kMHP + i];
          const float v = acc[r] * (1.0f - hv * hv);
          H2s[rr * kMHP + i] = v;            // same lane read h2 just above
          if (r0 + rr < B) xwg_store(p.dz2 + (int64_t)(r0 + rr) * kMH + i, v);
        }
      }
    }
    lds_barrier();
    // ---- dz1 = (dz2 W2) * (1 - h1^2) -------------------------------------------------------
    {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      const int i = 16 * w + c16;
      const float* ap = H2s + rowA * kMHP + 4 * g;
#pragma unroll
      for (int tt = 0; tt < 8; ++tt) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + 16 * tt);
        acc = mfma16(a4.x, w2b[4 * tt + 0], acc);
        acc = mfma16(a4.y, w2b[4 * tt + 1], acc);
        acc = mfma16(a4.z, w2b[4 * tt + 2], acc);
        acc = mfma16(a4.w, w2b[4 * tt + 3], acc);
      }
      if (4 * g < MR) {
#pragma unroll
        for (int r = 0; r < RQ; ++r) {
          const int rr = 4 * g + r;
          const float hv = H1s[rr * kMHP + i];
          if (r0 + rr < B) xwg_store(p.dz1 + (int64_t)(r0 + rr) * kMH + i, acc[r] * (1.0f - hv * hv));
        }
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0)
      flag_raise(p.flag_own, o, epoch);
    BSIG_MSTAMP(15);
    if (o == 0 && w == 0) {
      const float s = granule_gather(loss_granules(p.gran, epoch), p.n_owner, tag + 3, lane, flagp);
      if (lane == 0) {
        const float l = -s / (float)B;
        p.train_loss[step] = l;
        if (!isfinite(l)) atomicOr(flagp, 1);
      }
    }
    // the evaluation due after the previous update: the tile workgroups formed its
    // first-layer products while this update's rows were being finished
    if (__builtin_expect(p.do_eval && step > 0 && (step - 1) % p.eval_every == 0, 0)) {
      __syncthreads();
      owner_eval(mdnn_evals_before(step, p.eval_every) - 1,
                 rng_ctr0 + (uint64_t)t + (uint64_t)(mdnn_evals_before(step, p.eval_every) - ev0) - 1u, step, false);
    }
  }
  if (p.do_eval && step0 + p.n_updates == p.n_total && (!DP || p.n_updates == 0) &&
      !run_aborted(flagp, red, tid_0))
    owner_eval(mdnn_evals_before(p.n_total - 1, p.eval_every),
               rng_ctr0 + (uint64_t)p.n_updates +
                   (uint64_t)(mdnn_evals_before(p.n_total - 1, p.eval_every) - ev0), p.n_total, true);
}

// fit_persistent_mdnn_stream.hip
bool mdnn_stream_tile_geom(int FR, int chunks_per_wg, int S, int A, int* nip, int* pf, size_t* lds_bytes);
int mdnn_stream_chunks(int input_dim);
int mdnn_stream_launch(const MdnnArgs& p, bool dp, bool wide, bool full, int grid, size_t lds, hipStream_t st);

}  // namespace bsig
