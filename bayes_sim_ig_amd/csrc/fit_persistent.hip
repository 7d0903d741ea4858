// Persistent update kernel: a run of consecutive SGD updates of a linear
// mixture-density head on precomputed features in ONE launch
// (MDRFF.run_training's inner loop, mdnn.py:219-233, with the RFF projection
// hoisted: forward mdnn.py:108-119, NLL :127-178, its backward, Adam :203/:229).
//
// Round 4: UNIFIED workgroups.  The head matrix W [Nh, F] is tiled over the WHOLE chip:
// workgroup (nb, ks) owns the 16 x KS tile W[16nb.., KS*ks..] (NT = 1; 32 rows with NT = 2
// for heads whose 16-row tiles would need more than 256 workgroups), keeps it in LDS and
// both Adam moments in registers for the whole run.  The host's cost model (u_geom_try) picks the
// tiling and where the minibatch rows live: cfg5 (260 x 4096) runs 9 x 22 = 198 tile workgroups
// of 32 x 192 + 50 row owners of two rows each on CUs of their own (owner_only_workgroup), cfg2
// (270 x 1024) 187 tiles of 16 x 96 + 50 owners -- instead of the 9 x 16 = 144 tiles of 32 x 256
// (+ 100 row-owner workgroups) of fit_persistent_v1.hip.  Where the chip has no CUs to spare the
// first `n_owner` tile workgroups ALSO own minibatch rows (the row work then sits in what used to
// be their wait; BSIG_PERSIST_MIXED=1 forces that layout).  Per update:
//   1. forward: the minibatch tile [B, KS] arrives in REGISTERS (fetched during the
//      previous update's wait, already in the MFMA operand layout: wavefront w holds
//      rows 16w..16w+15), so the product P^T = W_tile F^T (fp32 MFMA 16x16x4, W from
//      LDS) starts at once; the same registers go to LDS transposed (F^T, the dW
//      operand) under the MFMAs.  A lane's accumulator is 4 adjacent head columns of
//      one row: it goes straight to the split-K slab (16-byte write-through stores),
//      no LDS staging, one barrier, flag;
//   2. row owners (workgroups 0..n_owner): wait for every flag, sum the k-slices of their
//      row, row-wise NLL forward / backward (diag_row, one wavefront per row); the three
//      batch-wide sums (jitter scale, its gradient term, the loss) cross workgroups as
//      {tag, value} granules; d_out rows written through;
//   3. every tile workgroup waits for the owners' granules, loads its [B, 16] block of
//      d_out transposed into LDS (jitter-scale gradient term applied on the way), forms
//      dW = d_out^T F (MFMA 16x16x4, both operands 16-byte LDS reads along the minibatch
//      axis) and applies Adam to its tile; the k-slice-0 workgroups also own the biases.
// Held-out evaluations (mdnn.py:235-242) run inside the launch: the tile part (held-out
// rows x the evaluated weights, still in LDS) in the wait of the update after the
// evaluation point, the row part one update later by workgroups that own no minibatch
// row (they idle in that wait) -- three evaluation slab buffers make that safe.
// Cross-workgroup data: write-through stores / cache-bypassing loads + flags, as in v1
// (persist_device.h); all polls are bounded (bit 1 of the state block's flag word).
#include "persist.h"

#include <algorithm>
#include <cstdlib>

#include "head_device.h"
#include "persist_device.h"

namespace bsig {

constexpr int kUT = 512;             // threads per workgroup (8 wavefronts)
constexpr int kUSteps = 18;          // 16-column steps of a k-slice a lane can hold (18 x 16 B registers)
constexpr int kUGroup = 6;           // k-slices are 1, 2 or 3 groups of 6 steps (96 / 192 / 288 columns): the
                                     // unrolled step loops branch once per group, not once per step
constexpr int kUMT = 7;              // 16-row m-tiles of the minibatch (<= 112 rows): fixed, so that the
                                     // dW product's loop over the minibatch has no run-time bound
constexpr int kULds = 160 * 1024;
constexpr int kUProf = 8;            // profiled updates per launch
// Pitch of the F^T / d_out^T rows in LDS (floats): minibatches of up to 112 rows.  = 4 (mod 8):
// conflict-free 16-byte reads down 8 rows and 4-byte writes across rows.  A compile-time constant:
// the 72 transposed writes of a lane per update are then immediate offsets off one address --
// with a run-time pitch the compiler keeps 72 loop-invariant addresses in registers (or scratch).
constexpr int kUFP = 116;

struct UArgs {
  int B, Bp, MT;                     // minibatch rows, padded to 16, 16-row m-tiles
  int Fdim, KS, ksteps, KB, WP;      // features, k-slice width, its 16-column steps / blocks, pitch of W rows
  int Nh, NhP, D, K;
  int n_blocks, k_slices, G, T, n_owner, R;
  int n_updates, xs_floats;
  int fast_rows;                     // (A/B runs: BSIG_PERSIST_FAST_ROWS=0 keeps the shape-generic row)
  const float* feats; int64_t ld_feats; const int32_t* feat_ids;
  const float* y; int64_t ldy; const int32_t* ids;
  float* params; float* m1; float* m2; int64_t w_off, b_off;
  int32_t* state; float* train_loss;
  double lr, beta1, beta2;
  float adam_eps, eps_noise, min_w, ll_limit, inv_norm;
  float* slabs; float* d_out; float* e_out; unsigned* flag_fwd; unsigned long long* gran;
  unsigned long long* gran_rep;      // [2][8][kGranArr] replicated granules (gran_rep())
  // data-parallel ranks (one update per launch; see PersistBuffers)
  float* grads; int adam_pending; int quad_ok;
  // data-parallel rank RESIDENT across the exchange (XR instantiation; see PersistBuffers)
  unsigned* xr_ready; const unsigned* xr_done; unsigned* xr_count; unsigned xr_base;   // (xr_count: a word of the sync region)
  // held-out evaluations inside the launch
  int do_eval, eval_every, n_total, n_test, eval_passes, NE, RE;
  int64_t eval_row0;
  const float* y_test; int64_t ldy_test;
  float* test_loss;
  float* eval_slabs;                 // [3][eval_passes][k_slices][B][NhP]
  unsigned* flag_eval;               // [G]   evaluation number + 1
  unsigned long long* gran_eval;     // [2][kGranArr]
  long long* prof;                   // diagnostics: [T][kUProf][16] wall-clock stamps, or null
  int prof_t0;                       // ... of updates prof_t0 .. prof_t0 + kUProf of the launch (BSIG_PROF_T0)
};

__device__ __forceinline__ f32x4 umfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Workgroup id -> roles.  Workgroup ids go round the 8 XCDs (id mod 8 is the XCD, measured).  The
// G tiles are dealt to the XCDs in contiguous runs -- tile (nb, ks) is number ks * n_blocks + nb,
// so the workgroups of one k-slice, which all read the same feature columns, share an XCD's L2
// -- and every XCD keeps its share of the T - G workgroups WITHOUT a tile.  Those come first in
// the owner order (a workgroup that owns minibatch rows and no tile has its vector-memory pipe
// to itself: the 77-115 KB minibatch tile a tile workgroup fetches per update costs a CU ~5 us
// of it), the tile workgroups follow in id order.
struct URole { int tile, owner; };        // tile: < G or -1; owner: rank 0 .. T-1 (rows if < n_owner)
__device__ __forceinline__ URole wg_role(int wg, int T, int G) {
  const int per = T >> 3, rem = T & 7, gq = G >> 3, gr = G & 7;
  const int x = wg & 7, j = wg >> 3;
  const int tiles_x = gq + (x < gr ? 1 : 0);
  URole r;
  if (j < tiles_x) {
    r.tile = x * gq + min(x, gr) + j;
    int rank = 0;                          // tile workgroups with a smaller id
    for (int xx = 0; xx < 8; ++xx) {
      const int cnt = wg > xx ? (wg - xx + 7) >> 3 : 0;
      rank += min(cnt, gq + (xx < gr ? 1 : 0));
    }
    r.owner = (T - G) + rank;
  } else {
    r.tile = -1;
    r.owner = x * (per - gq) + min(x, rem) - min(x, gr) + (j - tiles_x);
  }
  return r;
}

// Position of minibatch row b along the rows of F^T / d_out^T in LDS: inside its block of 16, row
// 4c + g sits at 4g + c.  The dW product reads 16 bytes at 16S + 4g of both operands, and MFMA c of
// step S then contracts the four ADJACENT rows 16S + 4c .. + 3: the padding behind the last
// minibatch row (100 rows: 12 of the last 16) costs no MFMAs.
__device__ __forceinline__ int u_rowpos(int b) { return (b & ~15) + ((b & 3) << 2) + ((b >> 2) & 3); }

__device__ __forceinline__ int u_evals_before(int s, int every) { return s == 0 ? 0 : (s - 1) / every + 1; }

#define BSIG_USTAMP(k)                                                           \
  do {                                                                           \
    if (p.prof && threadIdx.x == 0 && t >= p.prof_t0 && t < p.prof_t0 + kUProf)  \
      p.prof[((int64_t)wg * kUProf + t - p.prof_t0) * 16 + (k)] = wall_clock64(); \
  } while (0)

// which row the owners of this launch run: workgroup-uniform, the same for every launch of a shape.
// 0: the shape-generic row; else KP * 8 + NQH (u_own_update_fast<KP, NQH>).
__device__ __forceinline__ int u_fast_kind(const UArgs& p) {
  const int K = p.K;
  if (K > 16 || p.fast_rows == 0) return 0;
  if (p.T - p.G < p.n_owner) return 0;                    // (owners that also hold a tile run the shape-generic row)
  const int KP = K <= 4 ? 4 : (K <= 8 ? 8 : 16);
  const int nq = (p.D + 64 / KP - 1) / (64 / KP);         // sweeps of a row
  if (p.R > 4 || nq > 8 || p.R * ((p.Nh + 3) >> 2) > kUT) return 0;
  const int nqh = nq <= 2 ? 1 : (nq <= 4 ? 2 : 4);        // sweeps per wavefront
  return KP * 8 + nqh;
}

// ---- replicated granules (round 6) -------------------------------------------------------------------
// A granule that EVERY consumer of the chip reads -- an owner's sum of exp(pre) (read by the row wavefronts of
// all owners), its sum of u * dL/dsigma (read by all ~200 tile workgroups) -- sat in ONE 64-byte line: ~200
// pollers per line and round, and the quiet stage of all of them on the SAME word (owner 0's granule, the
// last tile's flag).  A line serves on the order of 100 accesses per us (MI355X_MICROARCH.md: a word
// saturates at ~88 dequeues / us; the arrival-counter experiment of profiles/r06_NOTES.md): the release of the
// tile workgroups took 0.9-1.3 us behind the last owner.  Now the producer writes EIGHT copies (lanes 0-7 of
// its wavefront 0, one store each), a consumer reads the copy of its XCD (workgroup id mod 8: ~25 readers
// per line), and the quiet stages wait on different words (tile workgroup i on owner i mod n_owner, owner i
// on tile i * G / n_owner).
// rep(arr, x): copy x of replicated array arr (0: sum exp(pre), 1: sum u dL/dsigma)
__device__ __forceinline__ unsigned long long* gran_rep(const UArgs& p, int arr, int x) {
  return p.gran_rep + (int64_t)(arr * 8 + x) * kGranArr;
}
// (all of lanes 0-7 of one wavefront come here with the same value)
__device__ __forceinline__ void granule_publish8(const UArgs& p, int arr, int slot, uint32_t tag, float v, int lane) {
  if (lane < 8) granule_publish(gran_rep(p, arr, lane), slot, tag, v);
}

// A row owner's sum over the k-slices: out[r * out_pitch + col] = sum_z slabs[z * zs + rowoff(r) + col]
// for r < nrows, col < n_cols (float offsets; `slabs` workgroup-uniform), slices added in order, up
// to 16 loads in flight per lane -- buffer loads: one address register per lane, the slice offset
// is scalar.  `fn(col, v)` sees every sum.
template <typename RowOff, typename F>
__device__ __forceinline__ void u_rows_sum(const float* slabs, RowOff&& rowoff, int k_slices, int zs, int nrows,
                                           int n_cols, float* out, int out_pitch, int tid, F&& fn) {
  const __amdgpu_buffer_rsrc_t sr = xwg_buffer(slabs);
  for (int idx = tid; idx < nrows * n_cols; idx += kUT) {
    const int r = idx / n_cols, col = idx - r * n_cols;
    const int voff = (rowoff(r) + col) * 4;
    float v = 0.f;
    for (int z = 0; z < k_slices; z += 16) {
      float q[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        q[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sr, voff, min(z + u, k_slices - 1) * zs * 4, kXwgPolicy));
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (z + u < k_slices) v += q[u];
    }
    out[r * out_pitch + col] = v;
    fn(col, v);
  }
}

// The same sum in ONE round trip for an owner without a tile (its LDS is free): thread (quad column,
// slice group sg) adds the slices sg, sg + SG, ... of 4 adjacent columns (16-byte loads, <= 8 in
// flight), the SG partial sums meet in LDS (`part`, 4 * kUT floats) and are added in group order.
// A fixed partition: bitwise reproducible.  nrows * ld / 4 <= kUT quads (else: u_rows_sum).
template <typename RowOff, typename F>
__device__ __forceinline__ void u_rows_sum4(const float* slabs, RowOff&& rowoff, int k_slices, int zs, int nrows,
                                            int n_cols, int ld, float* out, int out_pitch, float* part, int tid,
                                            F&& fn) {
  const __amdgpu_buffer_rsrc_t sr = xwg_buffer(slabs);
  const int ncq = (n_cols + 3) >> 2, Q = nrows * ncq;      // (only the quads that hold head outputs: 260 of 288 columns)
  const int SG = max(1, min(min(k_slices, kUT / Q), 8));
  const int qc = tid % Q, sg = tid / Q;
  if (sg < SG) {
    const int r = qc / ncq, c4 = (qc - r * ncq) * 4;
    const int voff = rowoff(r) + c4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int z0 = sg; z0 < k_slices; z0 += 8 * SG) {
      f32x4 q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = xwg_load4(sr, voff + min(z0 + u * SG, k_slices - 1) * zs);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (z0 + u * SG < k_slices) v += q[u];
    }
    *reinterpret_cast<f32x4*>(part + 4 * (sg * Q + qc)) = v;
  }
  __syncthreads();
  if (tid < Q) {
    const int r = tid / ncq, c4 = (tid - r * ncq) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + 4 * tid);
    for (int g = 1; g < SG; ++g) v += *reinterpret_cast<const f32x4*>(part + 4 * (g * Q + tid));
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c4 + j < n_cols) {
        out[r * out_pitch + c4 + j] = v[j];
        fn(c4 + j, v[j]);
      }
  }
}

// ---- tile part of held-out evaluation eidx: held-out rows x this tile's weights (the B operand
//      straight from memory, six 16-column steps at a time) -> evaluation slab buffer eidx % 3, flag
template <int NT>
__device__ __forceinline__ void u_tile_eval(const UArgs& p, const float* Wl, const float* biasl, int slot,
                                            int ks, int n0, int k0, int eidx) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));    // (nothing below may be computed ahead of the update loop and kept live in it)
  const int lane = tid & 63, w = tid >> 6, c16 = lane & 15, g4 = lane >> 4;
  const int B = p.B, NhP = p.NhP;
  for (int pass = 0; pass < p.eval_passes; ++pass) {
    const int rows = min(B, p.n_test - pass * B);
    if (rows <= 0) break;
    if (w < p.MT) {
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        acc[nt] = ks == 0 ? *reinterpret_cast<const f32x4*>(biasl + 16 * nt + 4 * g4) : zero;
      }
      const int64_t r = p.eval_row0 + pass * B + min(16 * w + c16, rows - 1);
      const float* src = p.feats + r * p.ld_feats;
      const float* ap = Wl + c16 * p.WP + 4 * g4;
#pragma unroll 1
      for (int s0 = 0; s0 < p.ksteps; s0 += kUGroup) {
        f32x4 breg[kUGroup];
#pragma unroll
        for (int q = 0; q < kUGroup; ++q) {
          const int col = k0 + 16 * (s0 + q) + 4 * g4;
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + min((int64_t)col, p.ld_feats - 4));
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          breg[q] = col < p.Fdim ? v : zero;
        }
#pragma unroll
        for (int q = 0; q < kUGroup; ++q) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + nt * 16 * p.WP + 16 * (s0 + q));
            acc[nt] = umfma(a4[0], breg[q][0], acc[nt]);
            acc[nt] = umfma(a4[1], breg[q][1], acc[nt]);
            acc[nt] = umfma(a4[2], breg[q][2], acc[nt]);
            acc[nt] = umfma(a4[3], breg[q][3], acc[nt]);
          }
        }
      }
      if (16 * w + c16 < rows) {
        const __amdgpu_buffer_rsrc_t sr = xwg_buffer(
            p.eval_slabs + ((((int64_t)(eidx % 3) * p.eval_passes + pass) * p.k_slices + ks) * B) * NhP + n0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          xwg_store4(sr, (16 * w + c16) * NhP + 16 * nt + 4 * g4, acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
      }
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (tid == 0) flag_raise(p.flag_eval, slot, (unsigned)eidx + 1u);
}

// ---- row part of held-out evaluation eidx (jitter stream `stream`) by evaluation owner `eo`:
//      held-out rows eo, eo + NE, ... (RE of them, one wavefront each), forward only
__device__ __forceinline__ void u_owner_eval(const UArgs& p, float* XS, float* red, int eo, int eidx,
                                             uint64_t stream, const HeadArgs& a) {
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63, w = tid >> 6;
  const int B = p.B, Nh = p.Nh, NhP = p.NhP, D = p.D, K = p.K, DK = D * K;
  const int per_wave = Nh + D + 3 * K;
  int32_t* flagp = p.state + 2;
  const unsigned etag = (unsigned)eidx + 1u;
  if (w == 0) flags_wait(p.flag_eval, p.G, etag, lane, flagp);
  __syncthreads();
  const int nr = eo < p.n_test ? min(p.RE, (p.n_test - eo + p.NE - 1) / p.NE) : 0;   // this owner's rows
  const float* ebase = p.eval_slabs + (int64_t)(eidx % 3) * p.eval_passes * p.k_slices * B * NhP;
  float eacc = 0.f;
  u_rows_sum(ebase,
             [&](int r) {
               const int erow = eo + p.NE * r, pass = erow / B, rb = erow - pass * B;
               return (pass * p.k_slices * B + rb) * NhP;
             },
             p.k_slices, B * NhP, nr, Nh, XS, per_wave, tid,
             [&](int col, float v) { if (col >= K + DK && col < K + 2 * DK) eacc += expf(v); });
  eacc = wave_sum_dpp(eacc);
  if (lane == 0) red[w] = eacc;
  __syncthreads();
  if (tid == 0) {
    float sx = 0.f;
    for (int q = 0; q < kUT / 64; ++q) sx += red[q];
    granule_publish(p.gran_eval, eo, etag, sx);
  }
  const int erow = eo + p.NE * w;
  const bool act = w < nr;
  float* tile = XS + w * per_wave;
  float* yv = tile + Nh;
  float* rk = yv + D;
  float* lpk = rk + K;
  float* dlg = lpk + K;
  if (act)
    for (int j = lane; j < D; j += 64) yv[j] = p.y_test[(int64_t)erow * p.ldy_test + j];
  RowOut ro;
  ro.lse = 0.f; ro.uds = 0.f; ro.bad = false;
#pragma unroll
  for (int q = 0; q < kElemsPerLane; ++q) ro.esg0[q] = 0.f;
  if (w < p.RE) {
    HeadArgs ae = a;
    ae.d_out = nullptr;                    // forward only
    ae.batch = p.n_test;
    ae.stream_id = stream;
    diag_row(ae, erow, act, lane, tile, yv, rk, lpk, dlg,
             [&] {
               return p.eps_noise != 0.f
                          ? p.eps_noise * (granule_gather(p.gran_eval, p.NE, etag, lane, flagp) /
                                           ((float)p.n_test * (float)DK))
                          : 0.f;
             },
             ro);
  }
  if (lane == 0) red[16 + w] = act ? ro.lse : 0.f;
  __syncthreads();
  if (tid == 0) {
    float sl = 0.f;
    for (int q = 0; q < kUT / 64; ++q) sl += red[16 + q];
    granule_publish(p.gran_eval + kGranArr, eo, etag, sl);
  }
  if (eo == 0 && w == 0) {
    const float sum = granule_gather(p.gran_eval + kGranArr, p.NE, etag, lane, flagp);
    if (lane == 0) {
      const float l = -sum / (float)p.n_test;
      p.test_loss[p.state[1]] = l;
      p.state[1] = p.state[1] + 1;
      if (!isfinite(l)) atomicOr(flagp, 1);
    }
  }
  if (ro.bad) atomicOr(flagp, 1);
  __syncthreads();
}

// ---- the row-owner role ---------------------------------------------------------------------
// (functions of their own with a context struct, not code inside the tile workgroups' loop: a
// workgroup WITHOUT a tile runs owner_only_workgroup, where none of the tile registers -- 72 of the
// next minibatch tile, the Adam moments -- exist, and the row arithmetic has the register file to
// itself; inside the common function it spilled)
struct UOwn {
  int own, r0, per_wave;          // owner rank, first row, floats of a row's block in LDS
  bool has_tile;                  // (a workgroup without a tile sums the k-slices through its free F^T region)
  float *XS, *red, *part;         // LDS: row blocks, [64] scratch, partial k-slice sums
  HeadArgs a; RowGeom rg;
  uint64_t rng_ctr0; int step0, ev0; float norm; int32_t* flagp;
};
__device__ __forceinline__ void u_own_init(const UArgs& p, UOwn& o, int own, bool has_tile, float* XS,
                                           float* red, float* part, int lane0) {
  o.own = own; o.r0 = own * p.R; o.per_wave = p.Nh + p.D + 3 * p.K; o.has_tile = has_tile;
  o.XS = XS; o.red = red; o.part = part;
  o.a = HeadArgs{};
  o.a.D = p.D; o.a.K = p.K; o.a.Nh = p.Nh; o.a.batch = p.B; o.a.from_tuple = 0;
  o.a.min_w = p.min_w; o.a.ll_limit = p.ll_limit; o.a.inv_norm = p.inv_norm;
  o.a.eps_noise = p.eps_noise; o.a.seed = reinterpret_cast<const uint64_t*>(p.state + 8)[0]; o.a.d_out = p.d_out;
  o.rg = row_geom(p.D, p.K, lane0);       // (integer divisions by run-time values: once per launch)
  o.rng_ctr0 = reinterpret_cast<const uint64_t*>(p.state + 8)[1];
  o.step0 = p.state[0];
  o.ev0 = p.do_eval ? u_evals_before(o.step0, p.eval_every) : 0;
  o.norm = (float)p.B * (float)(p.D * p.K);
  o.flagp = p.state + 2;
}
// jitter stream of evaluation e (one stream per update and per evaluation, in program order: the
// per-phase path's numbering); `last`: the evaluation after the call's last update
__device__ __forceinline__ uint64_t u_eval_stream(const UArgs& p, const UOwn& o, int e, bool last) {
  const int at = last ? p.n_total : e * p.eval_every + 1;     // the update it precedes
  return o.rng_ctr0 + (uint64_t)(at - o.step0) + (uint64_t)(e - o.ev0);
}

// Update t of the launch for the rows r0 .. r0 + R of this owner (every wavefront of the workgroup
// comes through here; wavefront w < R runs row r0 + w): k-slice sum, NLL forward / backward.
__device__ __forceinline__ void u_own_update(const UArgs& p, UOwn& o, int t, int w, int lane0, int wg) {
  const int B = p.B, Nh = p.Nh, NhP = p.NhP, D = p.D, K = p.K, DK = D * K;
  const int step = o.step0 + t;
  const unsigned epoch = (unsigned)step + 1u;
  const uint32_t tag = epoch * 4u;
  const int own = o.own, r0 = o.r0, per_wave = o.per_wave;
  const bool has_tile = o.has_tile;
  float* XS = o.XS; float* red = o.red; float* Ft = o.part;
  int32_t* flagp = o.flagp;
  HeadArgs& a = o.a;
  const RowGeom& rg = o.rg;
  const uint64_t rng_ctr0 = o.rng_ctr0;
  const int ev0 = o.ev0;
  const float norm = o.norm;
  const int tid = threadIdx.x;
  float* tile = XS + w * per_wave;             // (only waves < R touch theirs)
  float* yv = tile + Nh;
  float* rk = yv + D;
  float* lpk = rk + K;
  float* dlg = lpk + K;
  const int row = r0 + w;
  const bool owner_wave = w < p.R;
  const bool active = owner_wave && row < B;
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int tid_l = 64 * w + lane;
    if (active) {      // target row (independent of the forward product)
      const int64_t yrow = p.ids[(int64_t)step * B + row];
      for (int j = lane; j < D; j += 64) yv[j] = p.y[yrow * p.ldy + j];
    }
    // the row's jitter draws do not depend on the forward product: drawn in the wait
    float eu_pre[kElemsPerLane];
    a.stream_id = rng_ctr0 + (uint64_t)t + (uint64_t)(p.do_eval ? u_evals_before(step, p.eval_every) - ev0 : 0);
    if (owner_wave) diag_row_noise(a, rg.groups, rg.k, rg.d0, row, active, lane, eu_pre);
    if (w == 0) {
      flag_wait_one(p.flag_fwd, (int)(((int64_t)own * p.G) / p.n_owner), epoch, flagp);
      flags_wait(p.flag_fwd, p.G, epoch, lane, flagp);
    }
    __syncthreads();
    BSIG_USTAMP(4);
    float eacc = 0.f;
    if (!has_tile && p.R * NhP <= 4 * kUT)      // (the F^T region of a workgroup without a tile is free)
      u_rows_sum4(p.slabs, [&](int r) { return (r0 + r) * NhP; }, p.k_slices, B * NhP, min(p.R, B - r0), Nh,
                  NhP, XS, per_wave, Ft, tid_l,
                  [&](int col, float v) { if (col >= K + DK && col < K + 2 * DK) eacc += expf(v); });
    else
      u_rows_sum(p.slabs, [&](int r) { return (r0 + r) * NhP; }, p.k_slices, B * NhP,
                 min(p.R, B - r0), Nh, XS, per_wave, tid_l,
                 [&](int col, float v) { if (col >= K + DK && col < K + 2 * DK) eacc += expf(v); });
    eacc = wave_sum_dpp(eacc);
    if (lane == 0) red[w] = eacc;
    __syncthreads();
    if (tid_l < 8) {
      float sx = 0.f;
      for (int q = 0; q < kUT / 64; ++q) sx += red[q];
      granule_publish8(p, 0, own, tag + 1, sx, tid_l);
    }
    BSIG_USTAMP(5);
    RowOut ro;
    ro.lse = 0.f; ro.uds = 0.f; ro.bad = false;
#pragma unroll
    for (int q = 0; q < kElemsPerLane; ++q) ro.esg0[q] = 0.f;
    if (owner_wave) {
      GranuleEps ge{gran_rep(p, 0, wg & 7), p.n_owner, tag + 1, lane, flagp, p.eps_noise, norm, {0ull, 0ull, 0ull, 0ull}};
      diag_row_impl(a, rg, row, active, lane, tile, yv, rk, lpk, dlg, ge, ro, eu_pre);
#ifdef BSIG_ROW_PROF
      if (p.prof && tid == 0 && t >= p.prof_t0 && t < p.prof_t0 + kUProf)
        for (int i = 0; i < 9; ++i)
          p.prof[(int64_t)256 * kUProf * 16 + ((int64_t)wg * kUProf + t - p.prof_t0) * 16 + i] = ro.ts[i];
#endif
      const float uds_w = wave_sum_dpp(ro.uds);
      if (lane == 0) { red[16 + w] = active ? ro.lse : 0.f; red[32 + w] = uds_w; }
      BSIG_USTAMP(7);
      if (active) {
        // d_out row without the jitter-scale term, and exp(pre) of the row for the tile
        // workgroups to add it (lane's elements are columns lane + q*TPR)
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
    if (tid_l < 8) {
      float sl = 0.f, su = 0.f;
      for (int q = 0; q < p.R; ++q) { sl += red[16 + q]; su += red[32 + q]; }
      granule_publish8(p, 1, own, tag + 2, su, tid_l);
      if (tid_l == 0) granule_publish(loss_granules(p.gran, epoch), own, tag + 3, sl);
    }
    BSIG_USTAMP(9);
    if (own == 0 && w == 0) {
      const float s = granule_gather(loss_granules(p.gran, epoch), p.n_owner, tag + 3, lane, flagp);
      if (lane == 0) {
        const float l = -s / (float)B;
        p.train_loss[step] = l;
        if (!isfinite(l)) atomicOr(flagp, 1);
      }
    }
    if (ro.bad) atomicOr(flagp, 1);
}


// ---- fast row owners (round 6) -----------------------------------------------------------------
// An owner's row used to be ONE wavefront running the shape-generic diag_row: ~1800 instruction slots at
// one vector instruction per 4 cycles -- the row was bound by its instruction COUNT (ISA: run-time loops
// over the sweeps, index divisions by K, ~40 LDS round trips, a reduction tree of selects, 4 300
// v_readlane_b32 of spilled scalars in the owners' loop; profiles/r06_NOTES.md).  For a workgroup WITHOUT a
// tile and K in {4, 8, 16} components (the lanes of a component's dimensions then sit at a stride of K
// inside the 16-lane DPP rows, and K adjacent lanes hold the K components) the row is written again:
//  * TWO wavefronts per row (wavefront 2r + h takes the sweeps h, h + 2, ... of row r: cfg5 one sweep
//    each), on different SIMDs -- the workgroup has eight wavefronts and at most four rows;
//  * the lane's elements come straight from the partial k-slice sums in LDS (no combine pass, no row
//    image, no second barrier), its target value and jitter draws are fetched during the wait;
//  * sums over the dimensions of a component: DPP row rotations inside the 16-lane rows, then ONE LDS
//    exchange of 8 partial sums per component (4 DPP rows x 2 wavefronts), read back by EVERY lane;
//    softmax, logsumexp and the logit gradients by DPP butterflies over the K adjacent lanes, computed
//    redundantly by every lane -- no broadcast step, no second exchange;
//  * d_out / exp(pre): the four lanes of a quad hold four adjacent columns: gathered by quad_perm, one
//    16-byte write-through store per quad instead of five 4-byte stores per lane.
// Same formulas, same IEEE divisions, same Philox draws per (row, lane, sweep) as diag_row_body; the
// summation orders differ (tests: persistent == per-phase kernels / oracle at their tolerances; every
// variant of this kernel -- resident, data-parallel, resident across the exchange -- shares this code).
// Update t of the launch for the rows of this owner, every wavefront of the workgroup comes through here
// (wavefronts 2r, 2r + 1 run row r0 + r; the others only fetch k-slices and keep the barriers).
#ifdef BSIG_ROW_PROF
#define BSIG_FSTAMP(i) do { if (p.prof && threadIdx.x == 0 && t >= p.prof_t0 && t < p.prof_t0 + kUProf) \
    p.prof[(int64_t)256 * kUProf * 16 + ((int64_t)wg * kUProf + t - p.prof_t0) * 16 + (i)] = wall_clock64(); } while (0)
#else
#define BSIG_FSTAMP(i)
#endif
// KP: the component count padded to a power of two (4, 8, 16): lane = d-slot * KP + k, the lanes k >= K of
// a group idle (K = 5 -> 8, K = 10 -> 16: the reference YAMLs' 10 components) -- they enter the butterflies
// with the neutral element.  K == KP (`exact`: K a multiple of 4): the quad stores and the one-wavefront
// exp(pre) sum; else 4-byte stores and the sum through the wavefronts' partial sums.
template <int KP, int NQH>
__device__ __forceinline__ void u_own_update_fast(const UArgs& p, UOwn& o, int t, int w, int lane0, int wg) {
#pragma clang fp contract(off)
  constexpr int GR = 64 / KP;                  // dimensions per sweep
  const int K = p.K;
  const bool exact = K == KP;
  const int B = p.B, Nh = p.Nh, NhP = p.NhP, D = p.D, DK = D * K, R = p.R;
  const int step = o.step0 + t;
  const unsigned epoch = (unsigned)step + 1u;
  const uint32_t tag = epoch * 4u;
  const int own = o.own, r0 = o.r0;
  float* red = o.red; float* part = o.part; float* XS = o.XS;
  int32_t* flagp = o.flagp;
  int lane = lane0;
  asm volatile("" : "+v"(lane));
  const int tid_l = 64 * w + lane;
  const int rr = w >> 1, h = w & 1;
  const int row = r0 + rr;
  const bool row_wave = rr < R;                // (wavefront-uniform)
  const bool active = row_wave && row < B;
  const int k = lane & (KP - 1), d0 = lane / KP;
  const bool kok = k < K;
  const bool jitter = p.eps_noise != 0.f;

  // ---- in the wait: target values, jitter draws, the elements' columns ------------------------------
  float yd[NQH], eu[NQH];
  bool valid[NQH];
  int ecol[NQH];                               // element (d, k) of sweep h + 2 i: d * K + k
#pragma unroll
  for (int i = 0; i < NQH; ++i) {
    const int d = d0 + (h + 2 * i) * GR;
    valid[i] = active && kok && d < D;
    ecol[i] = min(d, D - 1) * K + min(k, K - 1);
    yd[i] = 0.f; eu[i] = 0.f;
  }
  if (active) {
    const int64_t yrow = p.ids[(int64_t)step * B + row];
#pragma unroll
    for (int i = 0; i < NQH; ++i)
      if (valid[i]) yd[i] = p.y[yrow * p.ldy + d0 + (h + 2 * i) * GR];
    if (jitter) {
      // the draw of element (d, k) is diag_row_noise's: the generic row's lane of the element is
      // (d % G) * K + k with G = 64 / K dimension slots, its sweep q = d / G; counter (row * 64 + lane) * 2
      // + (q >> 2), output q & 3.  (exact, NQH <= 2: the lanes and sweeps coincide -- one call)
      const uint64_t sid = o.rng_ctr0 + (uint64_t)t + (uint64_t)(p.do_eval ? u_evals_before(step, p.eval_every) - o.ev0 : 0);
      if (exact && NQH <= 2) {
        const Philox4 ph = philox4x32_10(o.a.seed, sid, ((uint64_t)row * 64 + lane) * 2);
#pragma unroll
        for (int i = 0; i < NQH; ++i)
          if (valid[i]) eu[i] = u01(h ? ph.v[(2 * i + 1) & 3] : ph.v[(2 * i) & 3]);
      } else {
        const int G = 64 / K;
#pragma unroll
        for (int i = 0; i < NQH; ++i) {
          if (valid[i]) {
            const int d = d0 + (h + 2 * i) * GR, q = d / G, gl = (d - q * G) * K + k;
            const Philox4 ph = philox4x32_10(o.a.seed, sid, ((uint64_t)row * 64 + gl) * 2 + (q >> 2));
            const int oi = q & 3;
            eu[i] = u01(oi == 0 ? ph.v[0] : (oi == 1 ? ph.v[1] : (oi == 2 ? ph.v[2] : ph.v[3])));
          }
        }
      }
    }
  }
  if (w == 0) {
    flag_wait_one(p.flag_fwd, (int)(((int64_t)own * p.G) / p.n_owner), epoch, flagp);
    flags_wait(p.flag_fwd, p.G, epoch, lane, flagp);
  }
  __syncthreads();
  BSIG_USTAMP(4);

  // ---- partial sums over the k-slices -> LDS (u_rows_sum4's first half) -----------------------------
  const int nrows = min(R, B - r0);
  const int ncq = (Nh + 3) >> 2, Q = nrows * ncq;
  const int SG = max(1, min(min(p.k_slices, kUT / Q), 8));
  {
    const __amdgpu_buffer_rsrc_t sr = xwg_buffer(p.slabs);
    const int qc = tid_l % Q, sg = tid_l / Q;
    if (sg < SG) {
      const int r = qc / ncq, c4 = (qc - r * ncq) * 4;
      const int voff = (r0 + r) * NhP + c4, zs = B * NhP;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int z0 = sg; z0 < p.k_slices; z0 += 8 * SG) {
        f32x4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = xwg_load4(sr, voff + min(z0 + u * SG, p.k_slices - 1) * zs);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (z0 + u * SG < p.k_slices) v += q[u];
      }
      *reinterpret_cast<f32x4*>(part + 4 * (sg * Q + qc)) = v;
    }
  }
  BSIG_FSTAMP(0);
  __syncthreads();
  BSIG_FSTAMP(1);

  // ---- the lane's elements from the partial sums: pre first (its exp goes to the other owners) --------
  // value(col) of row rr = sum over g < SG of part[4 * (g * Q + rr * ncq) + col], groups added in order
  const float* pb = part + 4 * (active ? rr : 0) * ncq;
  const int gs = 4 * Q;
  auto slice_sum = [&](int col) {
    const float* b = pb + col;
    const float v0 = b[0], v1 = b[gs * min(1, SG - 1)], v2 = b[gs * min(2, SG - 1)], v3 = b[gs * min(3, SG - 1)];
    float sacc = v0;
    if (SG > 1) sacc += v1;
    if (SG > 2) sacc += v2;
    if (SG > 3) sacc += v3;
    if (SG > 4) {
      const float v4 = b[gs * 4], v5 = b[gs * min(5, SG - 1)], v6 = b[gs * min(6, SG - 1)], v7 = b[gs * min(7, SG - 1)];
      sacc += v4;
      if (SG > 5) sacc += v5;
      if (SG > 6) sacc += v6;
      if (SG > 7) sacc += v7;
    }
    return sacc;
  };
  // The workgroup's sum of exp(pre) is what the OTHER owners wait for: when the pre-activation quads of
  // its rows fit one wavefront (cfg5: 2 rows x 32 quads) ONE wavefront adds their partial sums itself
  // (16-byte LDS reads), takes the exponentials and publishes the granule -- no second barrier, no
  // exchange of wavefront sums between the k-slice sums and the granule.
  const int pq = DK >> 2, npq = nrows * pq;
  const bool one_wave_exp = exact && npq <= 64;
  if (one_wave_exp && w == kUT / 64 - 1) {      // (the last wavefront: no row of its own unless the workgroup has four)
    const int r = lane / pq, c4 = K + DK + 4 * (lane - r * pq);
    const float* b = part + 4 * min(r, nrows - 1) * ncq + c4;
    f32x4 v = *reinterpret_cast<const f32x4*>(b);
    for (int g = 1; g < SG; ++g) v += *reinterpret_cast<const f32x4*>(b + gs * g);
    float es = lane < npq ? (expf(v[0]) + expf(v[1])) + (expf(v[2]) + expf(v[3])) : 0.f;
    es = wave_sum_dpp(es);
    granule_publish8(p, 0, own, tag + 1, es, lane);
  }
  float ev[NQH], muv[NQH], pre[NQH];
#pragma unroll
  for (int i = 0; i < NQH; ++i) pre[i] = slice_sum(K + DK + ecol[i]);
#pragma unroll
  for (int i = 0; i < NQH; ++i) muv[i] = slice_sum(K + ecol[i]);
  const float lg_own = slice_sum(min(k, K - 1));
#pragma unroll
  for (int i = 0; i < NQH; ++i) ev[i] = valid[i] ? expf(pre[i]) : 0.f;
  if (jitter && row_wave) {
    // exp(pre) of the row for the tile workgroups (they add the jitter-scale gradient term): known now,
    // stored now -- in the wait for the other owners' sums, not behind the backward pass
    const __amdgpu_buffer_rsrc_t er = xwg_buffer(p.e_out + (int64_t)(active ? row : 0) * NhP);
#pragma unroll
    for (int i = 0; i < NQH; ++i) {
      const f32x4 qe = quad_gather(ev[i]);
      if (exact) {
        if (valid[i] && (lane & 3) == 0) xwg_store4(er, K + DK + ecol[i], qe[0], qe[1], qe[2], qe[3]);
      } else if (valid[i]) {
        xwg_store(p.e_out + (int64_t)row * NhP + K + DK + ecol[i], ev[i]);
      }
    }
  }
  if (!one_wave_exp) {
    float esum = 0.f;
#pragma unroll
    for (int i = 0; i < NQH; ++i) esum += ev[i];
    esum = wave_sum_dpp(esum);
    if (lane == 0) red[w] = esum;
    __syncthreads();
    if (tid_l < 8) {
      float sx = 0.f;
      for (int q = 0; q < 2 * R; ++q) sx += red[q];
      granule_publish8(p, 0, own, tag + 1, sx, tid_l);
    }
  }
  BSIG_USTAMP(5);
  GranuleEps ge{gran_rep(p, 0, wg & 7), p.n_owner, tag + 1, lane, flagp, p.eps_noise, o.norm, {0ull, 0ull, 0ull, 0ull}};
  float uds = 0.f, lse = 0.f;
  bool bad = false;
  if (row_wave) {
    ge.issue();      // (the other owners' sums: in flight over the mixture weights)
    float dmu[NQH], dpre[NQH], dlogit;
    float* xq = XS + ((active ? rr : 0) * KP + k) * 8;      // the pair's exchange: [row][k][8] quad | + 4 KP 8: logdet
    diag_row_fast_core<KP, NQH>(K, D, active, kok, valid, muv, yd, eu, ev, lg_own, p.min_w, p.ll_limit, p.inv_norm,
                                xq, xq + 4 * KP * 8, h, lane, ge, [] { lds_barrier(); }, dmu, dpre, dlogit, lse, uds, bad);
  BSIG_FSTAMP(6);
    BSIG_USTAMP(7);
    // ---- d_out row (without the jitter-scale term) and exp(pre), 16 bytes per quad -------------------
    const __amdgpu_buffer_rsrc_t dr = xwg_buffer(p.d_out + (int64_t)(active ? row : 0) * NhP);
    float* drow = p.d_out + (int64_t)(active ? row : 0) * NhP;
#pragma unroll
    for (int i = 0; i < NQH; ++i) {
      const f32x4 qm = quad_gather(dmu[i]), qp = quad_gather(dpre[i]);
      if (exact) {
        if (valid[i] && (lane & 3) == 0) {
          xwg_store4(dr, K + ecol[i], qm[0], qm[1], qm[2], qm[3]);
          xwg_store4(dr, K + DK + ecol[i], qp[0], qp[1], qp[2], qp[3]);
        }
      } else if (valid[i]) {
        xwg_store(drow + K + ecol[i], dmu[i]);
        xwg_store(drow + K + DK + ecol[i], dpre[i]);
      }
    }
    {
      const f32x4 ql = quad_gather(dlogit);
      if (exact) {
        if (active && h == 0 && lane < K && (lane & 3) == 0) xwg_store4(dr, lane, ql[0], ql[1], ql[2], ql[3]);
      } else if (active && h == 0 && lane < KP && kok) {
        xwg_store(drow + k, dlogit);
      }
    }
    // Every row wavefront publishes for itself: its stores acknowledged, then ITS sum of u * dL/dsigma
    // (slot own * 2R + wavefront; the tile workgroups add the granules of all row wavefronts in slot
    // order) and, from the first wavefront of a row, the row's logsumexp -- no workgroup barrier and
    // no LDS exchange between the last store and the granules the tile workgroups wait for.
    uds = wave_sum_dpp(uds);
    BSIG_FSTAMP(7);
    __builtin_amdgcn_s_waitcnt(0);
    BSIG_FSTAMP(8);
    granule_publish8(p, 1, own * 2 * R + w, tag + 2, uds, lane);
    if (h == 0 && lane == 0) granule_publish(loss_granules(p.gran, epoch), own * R + rr, tag + 3, active ? lse : 0.f);
  } else {
    lds_barrier();                                // (the row wavefronts' exchange of partial sums)
  }
  BSIG_USTAMP(9);
  if (own == 0 && w == 0) {
    const float s = granule_gather(loss_granules(p.gran, epoch), p.n_owner * R, tag + 3, lane, flagp);
    if (lane == 0) {
      const float l = -s / (float)B;
      p.train_loss[step] = l;
      if (!isfinite(l)) atomicOr(flagp, 1);
    }
  }
  if (bad) atomicOr(flagp, 1);
}

// The minibatch tile of update `step`, straight into the forward product's B-operand registers:
// wavefront w < MT holds rows 16w .. 16w+15, lane (c16, g4) the four columns 16S + 4 g4 .. +3 of
// step S (64 contiguous bytes per row and instruction).
#define BSIG_U_ROWID(step_)                                                                  \
  if (has_tile && w < p.MT) {                                                                \
    const int64_t r = (int64_t)(step_) * B + min(16 * w + c16_l, B - 1);                     \
    pf_row = p.feat_ids ? (int64_t)p.feat_ids[r] : r;                                        \
  }
#define BSIG_U_PREFETCH()                                                                    \
  if (has_tile && w < p.MT) {                                                                \
    const float* src = p.feats + pf_row * p.ld_feats + k0 + 4 * g4_l;                        \
    if (k0 + p.KS <= p.ld_feats) {         /* the whole k-slice inside the rows: immediate offsets */ \
      _Pragma("unroll") for (int G6 = 0; G6 < kUSteps / kUGroup; ++G6)                       \
        if (kUGroup * G6 < p.ksteps) {                                                       \
          _Pragma("unroll") for (int S = kUGroup * G6; S < kUGroup * (G6 + 1); ++S)          \
            Freg[S] = *reinterpret_cast<const f32x4*>(src + 16 * S);                         \
        }                                                                                    \
    } else {                                                                                 \
      const int cmax = (int)p.ld_feats - 4 - k0 - 4 * g4_l;                                  \
      _Pragma("unroll") for (int G6 = 0; G6 < kUSteps / kUGroup; ++G6)                       \
        if (kUGroup * G6 < p.ksteps) {                                                       \
          _Pragma("unroll") for (int S = kUGroup * G6; S < kUGroup * (G6 + 1); ++S)          \
            Freg[S] = *reinterpret_cast<const f32x4*>(src + min(16 * S, cmax));              \
        }                                                                                    \
    }                                                                                        \
  }

// DP: data-parallel rank (gradients out, pending Adam step in).  NT: 16-row blocks per tile.
// XR (with DP = false): a data-parallel rank that stays RESIDENT across the gradient exchange -- the
// single-rank kernel (W in LDS, moments in registers, a whole call per launch) whose dW goes to the flat
// gradient buffer, written through; the tile workgroups count themselves, the last one raises the word
// a second stream waits on (hipStreamWaitValue32 -> all-reduce -> hipStreamWriteValue32), every tile
// workgroup polls the word that stream writes, takes its tile of the REDUCED gradients back around
// the caches and takes the Adam step.  Same arithmetic per element as the other two instantiations:
// a 1-rank group reproduces the single-rank run bit for bit.
template <bool DP, int NT, bool XR = false>
__device__ __forceinline__ void unified_workgroup(const UArgs& p, float* smem, const URole& role) {
  static_assert(!(DP && XR), "XR is the resident kernel with the exchange inside, not the per-update launch");
  constexpr int NBW = 16 * NT;                 // head rows of the tile
  constexpr int MAXBLK = NT == 1 ? 3 : 2;      // 16-column blocks per wavefront in the dW / Adam phase
  float* Ft = smem;                            // [KS][FP]   minibatch features of this k-slice, transposed
  float* Wl = Ft + p.KS * kUFP;               // [NBW][WP]  weight tile (authoritative copy)
  float* XS = Wl + NBW * p.WP;                 // scratch: d_out^T [NBW][FP] | owner rows | evaluation rows
  float* red = XS + p.xs_floats;               // [64]
  float* biasl = red + 64;                     // [NBW] biases of the block (k-slice 0)
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);      // (wavefront-uniform: scalar tests)
  const int c16 = lane0 & 15, g4 = lane0 >> 4;
  const int wg = blockIdx.x;
  const int slot = role.tile, own = role.owner;
  constexpr bool has_tile = true;           // (workgroups without a tile: owner_only_workgroup)
  const bool has_row = own < p.n_owner;
  const int ks = slot / p.n_blocks, nb = slot - ks * p.n_blocks;
  const int n0 = nb * NBW, k0 = ks * p.KS;
  const int B = p.B, Nh = p.Nh, NhP = p.NhP, D = p.D, K = p.K, DK = D * K;
  constexpr int FP = kUFP;
  const int WP = p.WP;
  int32_t* flagp = p.state + 2;
  const int step0 = p.state[0];
  if (p.prof && p.n_updates > 0 && tid == 0) p.prof[((int64_t)wg * kUProf) * 16 + 14] = wall_clock64();
  double b1t = reinterpret_cast<const double*>(p.state + 12)[0];
  double b2t = reinterpret_cast<const double*>(p.state + 12)[1];
  float a0 = 0.f, a1 = 0.f;
  const AdamK ak{1.0f - (float)p.beta1, (float)p.beta2, 1.0f - (float)p.beta2, p.adam_eps};
  const bool pend = DP && p.adam_pending != 0;
  // first launch of a run_training call: a fresh optimizer (mdnn.py:203) -- the moments start at
  // zero in the registers, nobody has to clear (or read) them in memory
  const bool fresh = !DP && step0 == 0;
  const float pa0 = pend ? reinterpret_cast<const float*>(p.state)[4] : 0.f;
  const float pa1 = pend ? reinterpret_cast<const float*>(p.state)[5] : 0.f;

  // ---- the tile: W -> LDS, Adam moments -> registers in the dW accumulator layout: element v of
  //      block (jj, nt) of lane (c16, g4) of wave w  <->  W[n0 + 16nt + 4g4 + v][k0 + 16(w + 8jj) + c16]
  float Mr[MAXBLK][NT][4], Vr[MAXBLK][NT][4];
#pragma unroll
  for (int jj = 0; jj < MAXBLK; ++jj)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int v = 0; v < 4; ++v) { Mr[jj][nt][v] = 0.f; Vr[jj][nt][v] = 0.f; }
  float bw[NT], bm[NT], bv[NT];                // biases of the block and their moments: 16-row block nt on wave 8 - NT + nt, lanes < 16
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) { bw[nt] = 0.f; bm[nt] = 0.f; bv[nt] = 0.f; }
  if (has_tile) {
    if (DP && p.quad_ok) {
      // A data-parallel launch only passes through the tile here (pending Adam step, weights into
      // LDS; its dW goes to the gradient buffer) and Adam is elementwise: rows of 16-byte quads,
      // every load of a thread issued before the first store.
      const int qpr = p.KS >> 2, total = NBW * qpr;
      for (int base = 0; base < total; base += 3 * kUT) {
        f32x4 Wq[3], Mq[3], Vq[3], Gq[3];
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int idx = base + q * kUT + tid, r = idx / qpr, c = (idx - r * qpr) * 4;
          Wq[q] = zero; Mq[q] = zero; Vq[q] = zero; Gq[q] = zero;
          if (idx < total && n0 + r < Nh && k0 + c < p.Fdim) {
            const int64_t off = p.w_off + (int64_t)(n0 + r) * p.Fdim + k0 + c;
            Wq[q] = *reinterpret_cast<const f32x4*>(p.params + off);
            if (pend) {
              Mq[q] = *reinterpret_cast<const f32x4*>(p.m1 + off);
              Vq[q] = *reinterpret_cast<const f32x4*>(p.m2 + off);
              Gq[q] = *reinterpret_cast<const f32x4*>(p.grads + off);
            }
          }
        }
        if (pend) {
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              float m = Mq[q][j], v = Vq[q][j];
              Wq[q][j] = adam_weight(Gq[q][j], m, v, Wq[q][j], pa0, pa1, ak);
              Mq[q][j] = m; Vq[q][j] = v;
            }
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const int idx = base + q * kUT + tid, r = idx / qpr, c = (idx - r * qpr) * 4;
            if (idx < total && n0 + r < Nh && k0 + c < p.Fdim) {
              const int64_t off = p.w_off + (int64_t)(n0 + r) * p.Fdim + k0 + c;
              *reinterpret_cast<f32x4*>(p.params + off) = Wq[q];
              *reinterpret_cast<f32x4*>(p.m1 + off) = Mq[q];
              *reinterpret_cast<f32x4*>(p.m2 + off) = Vq[q];
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const int idx = base + q * kUT + tid, r = idx / qpr, c = (idx - r * qpr) * 4;
          if (idx < total) *reinterpret_cast<f32x4*>(Wl + r * WP + c) = Wq[q];
        }
      }
    } else {
      // (three passes: every load is issued before the first store)
      float Wv[MAXBLK][NT][4], Gv[MAXBLK][NT][4];
#pragma unroll
      for (int jj = 0; jj < MAXBLK; ++jj)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int n = n0 + 16 * nt + 4 * g4 + v, kc = 16 * (w + 8 * jj) + c16;
            Wv[jj][nt][v] = 0.f; Gv[jj][nt][v] = 0.f;
            if (w + 8 * jj < p.KB && n < Nh && k0 + kc < p.Fdim) {
              const int64_t off = p.w_off + (int64_t)n * p.Fdim + k0 + kc;
              Wv[jj][nt][v] = p.params[off];
              if (!fresh) { Mr[jj][nt][v] = p.m1[off]; Vr[jj][nt][v] = p.m2[off]; }
              if (pend) Gv[jj][nt][v] = p.grads[off];
            }
          }
      if (pend) {
#pragma unroll
        for (int jj = 0; jj < MAXBLK; ++jj)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 4; ++v)
              Wv[jj][nt][v] = adam_weight(Gv[jj][nt][v], Mr[jj][nt][v], Vr[jj][nt][v], Wv[jj][nt][v], pa0, pa1, ak);
#pragma unroll
        for (int jj = 0; jj < MAXBLK; ++jj)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int n = n0 + 16 * nt + 4 * g4 + v, kc = 16 * (w + 8 * jj) + c16;
              if (w + 8 * jj < p.KB && n < Nh && k0 + kc < p.Fdim) {
                const int64_t off = p.w_off + (int64_t)n * p.Fdim + k0 + kc;
                p.params[off] = Wv[jj][nt][v]; p.m1[off] = Mr[jj][nt][v]; p.m2[off] = Vr[jj][nt][v];
              }
            }
      }
#pragma unroll
      for (int jj = 0; jj < MAXBLK; ++jj)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int v = 0; v < 4; ++v)
            if (w + 8 * jj < p.KB) Wl[(16 * nt + 4 * g4 + v) * WP + 16 * (w + 8 * jj) + c16] = Wv[jj][nt][v];
    }
    if (ks == 0 && w >= 8 - NT && g4 == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (w != 8 - NT + nt) continue;
        const int n = n0 + 16 * nt + c16;
        if (n < Nh) {
          bw[nt] = p.params[p.b_off + n];
          if (!fresh) { bm[nt] = p.m1[p.b_off + n]; bv[nt] = p.m2[p.b_off + n]; }
          if (pend) {
            bw[nt] = adam_bias(p.grads[p.b_off + n], bm[nt], bv[nt], bw[nt], pa0, pa1, ak);
            p.params[p.b_off + n] = bw[nt]; p.m1[p.b_off + n] = bm[nt]; p.m2[p.b_off + n] = bv[nt];
          }
        }
        biasl[16 * nt + c16] = bw[nt];
      }
    }
  }
  // the sticky time-out bit of an EARLIER launch of this call (a data-parallel rank makes one launch per
  // update): sampled at entry, so that such a launch leaves at once instead of running its forward
  // product into the bounded polls of owners that have already left
  if (tid == 0)
    red[63] = (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) ? 1.f : 0.f;
  if (has_tile && p.MT < kUMT)      // F^T columns of the m-tiles no wavefront writes (minibatches of <= 96 rows)
    for (int idx = tid; idx < p.KS * (kUFP / 4); idx += kUT) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(Ft + 4 * idx) = zero;
    }

  // ---- row-owner state (a tile workgroup that also owns minibatch rows: heads with no CUs to spare)
  UOwn o;
  u_own_init(p, o, own, true, XS, red, Ft, lane0);
  // evaluation owner: the workgroups without a minibatch row come first
  const int eo = (own - p.n_owner + p.T) % p.T;
  const bool has_erow = p.do_eval && eo < p.NE;
  int pending_eval = -1;                       // evaluation whose tile part is out and whose row part is due

  // feature tile of the first update (later ones are fetched during the waits)
  f32x4 Freg[kUSteps];
  {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int S = 0; S < kUSteps; ++S) Freg[S] = zero;
  }
  int64_t pf_row = 0;                         // feature row of this lane's minibatch row, next update
  {
    const int c16_l = c16, g4_l = g4;
    if (p.n_updates > 0) { BSIG_U_ROWID(step0) BSIG_U_PREFETCH() }
  }
  __syncthreads();

  for (int t = 0; t < p.n_updates; ++t) {
    // Lane-derived indices are laundered at the start of every phase: the address arithmetic and
    // the predicates built on them are loop-invariant, and hoisted out of the update loop they do
    // not fit the register file (the row owner's code alone left 90 spilled registers that way).
    int tid_l = tid, c16_l = c16, g4_l = g4;
    asm volatile("" : "+v"(tid_l), "+v"(c16_l), "+v"(g4_l));
    const int step = step0 + t;
    const unsigned epoch = (unsigned)step + 1u;
    const uint32_t tag = epoch * 4u;
    if (red[63] != 0.f) break;     // time-out bit as sampled during the previous update's wait
    BSIG_USTAMP(0);
    {
      // ---- 1. forward: P^T[n, b] = sum_k W[n, k] F[b, k] on this k-slice ------------------
      if (w < p.MT) {
        const bool rok = 16 * w + c16_l < B;
        if (16 * (w + 1) > B || k0 + p.KS > p.Fdim) {      // padding rows / the tail of the last k-slice
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int S = 0; S < kUSteps; ++S)
            if (!(rok && k0 + 16 * S + 4 * g4_l < p.Fdim)) Freg[S] = zero;
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
          acc[nt] = ks == 0 ? *reinterpret_cast<const f32x4*>(biasl + 16 * nt + 4 * g4_l) : zero;
        }
        const float* ap = Wl + c16_l * WP + 4 * g4_l;
        float* ftp = Ft + (4 * g4_l) * FP + 16 * w + u_rowpos(c16_l);
#pragma unroll
        for (int G6 = 0; G6 < kUSteps / kUGroup; ++G6) {
          if (kUGroup * G6 < p.ksteps) {
#pragma unroll
            for (int S = kUGroup * G6; S < kUGroup * (G6 + 1); ++S) {
              const f32x4 b4 = Freg[S];
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(ap + nt * 16 * WP + 16 * S);
                acc[nt] = umfma(a4[0], b4[0], acc[nt]);
                acc[nt] = umfma(a4[1], b4[1], acc[nt]);
                acc[nt] = umfma(a4[2], b4[2], acc[nt]);
                acc[nt] = umfma(a4[3], b4[3], acc[nt]);
              }
              // the same registers, transposed, are the dW phase's B operand
              ftp[(16 * S + 0) * FP] = b4[0];
              ftp[(16 * S + 1) * FP] = b4[1];
              ftp[(16 * S + 2) * FP] = b4[2];
              ftp[(16 * S + 3) * FP] = b4[3];
            }
          }
        }
        BSIG_USTAMP(1);
        if (rok) {
          const __amdgpu_buffer_rsrc_t sr = xwg_buffer(p.slabs + (int64_t)ks * B * NhP + n0);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            xwg_store4(sr, (16 * w + c16_l) * NhP + 16 * nt + 4 * g4_l, acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
        }
      }
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid_l == 0) flag_raise(p.flag_fwd, slot, epoch);
      BSIG_USTAMP(3);
      // the next minibatch's row ids (the tile itself is requested further down)
      if (t + 1 < p.n_updates) { BSIG_U_ROWID(step + 1) }
    }
    // the time-out bit (set by any bounded poll on the chip), sampled off the critical path
    if (tid_l == 0)
      red[63] = (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2) ? 1.f : 0.f;
    b1t *= p.beta1; b2t *= p.beta2;        // beta^t as running products (double)
    a0 = (float)(p.lr / (1.0 - b1t));
    a1 = (float)(1.0 / sqrt(1.0 - b2t));

    // ---- 2. rows of this workgroup, if it owns any ------------------------------------------------
    if (has_row) u_own_update(p, o, t, w, lane0, wg);
    // ---- held-out evaluations in the wait: the row part of the evaluation whose products went out
    //      during the previous update, the tile part of the one due after the previous update
    if (__builtin_expect(pending_eval >= 0, 0)) {
      __syncthreads();
      if (has_erow) u_owner_eval(p, XS, red, eo, pending_eval, u_eval_stream(p, o, pending_eval, false), o.a);
      pending_eval = -1;
    }
    if (__builtin_expect(p.do_eval && step > 0 && (step - 1) % p.eval_every == 0, 0)) {
      const int e = u_evals_before(step, p.eval_every) - 1;
      u_tile_eval<NT>(p, Wl, biasl, slot, ks, n0, k0, e);
      pending_eval = e;
    }

    // ---- 3. dW = d_out^T F on this tile, Adam --------------------------------------------------
    {
      int tid_l = tid, c16_l = c16, g4_l = g4;
      asm volatile("" : "+v"(tid_l), "+v"(c16_l), "+v"(g4_l));
      const int lane = tid_l & 63;
      // The next minibatch tile, 126 KB per CU through the vector-memory pipe: requested HERE, in
      // front of the wait for the owners -- idle time for a workgroup without a row (it comes
      // straight from its forward flag) and for a row owner (the others are still finishing).  Not
      // earlier: the 18 x 16-byte registers would be live through the row owner's work.  Not later:
      // requested behind the d_out^T block it took 2.6 us out of the dW phase.
      if (t + 1 < p.n_updates) { BSIG_U_PREFETCH() }
      BSIG_USTAMP(2);
      // the owners' sum(u * dL/dsigma) granules double as their "d_out rows are out" flags; the
      // jitter-scale gradient term  d pre += (EPS/(B*D*K)) * sum(u*dL/dsigma) * exp(pre)
      // is applied here, to this block's columns, while the block is loaded
      if (w == 0) {
        // (fast row owners publish one granule per row wavefront, the shape-generic ones one per owner)
        unsigned long long* sug = gran_rep(p, 1, wg & 7);
        const int n_su = (p.T - p.G >= p.n_owner && u_fast_kind(p) != 0) ? p.n_owner * 2 * p.R : p.n_owner;
        granule_wait_one(sug, slot % n_su, tag + 2, flagp);
        const float su = granule_gather(sug, n_su, tag + 2, lane, flagp);
        if (lane == 0) red[62] = p.eps_noise != 0.f ? p.eps_noise / ((float)B * (float)DK) * su : 0.f;
      }
      __syncthreads();
      BSIG_USTAMP(10);
      {
        const float c = red[62];
        const int sg_lo = K + DK, sg_hi = K + 2 * DK;
        const bool any_e = c != 0.f && n0 + NBW > sg_lo && n0 < sg_hi;     // workgroup-uniform
        const __amdgpu_buffer_rsrc_t dr = xwg_buffer(p.d_out + n0), er = xwg_buffer(p.e_out + n0);
        constexpr int QN = NBW / 4;                // quads per row of the block
        constexpr int total = 16 * kUMT * QN;     // (rows beyond B: zeros)
        for (int base = 0; base < total; base += 2 * kUT) {
          f32x4 q[2], e[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int idx = base + u * kUT + tid_l;
            const int b = idx / QN, c4 = (idx - b * QN) * 4;
            const bool ok = idx < total && b < B;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            q[u] = ok ? xwg_load4(dr, b * NhP + c4) : zero;
            e[u] = ok && any_e ? xwg_load4(er, b * NhP + c4) : zero;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int idx = base + u * kUT + tid_l;
            const int b = idx / QN, c4 = (idx - b * QN) * 4;
            if (idx < total) {
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int n = n0 + c4 + j;
                const float ev = (n >= sg_lo && n < sg_hi) ? e[u][j] : 0.f;   // (other columns of e_out hold no data)
                XS[(c4 + j) * FP + u_rowpos(b)] = q[u][j] + c * ev;
              }
            }
          }
        }
      }
      __syncthreads();
      BSIG_USTAMP(11);
      const int n4 = (B + 3) >> 2;
      // (XR) gradient tile <-> p.grads: byte offset of this lane's 16 bytes in block 0 of tile 0
      // (blocks [0, xr_kb) of this k-slice hold columns < Fdim; in the last of them, when Fdim is not a
      // multiple of 16, only the lanes with xr_col + 128 jj < xr_cols -- the others get an offset beyond
      // the end of the buffer: stores dropped, loads zero)
      [[maybe_unused]] const int xr_cols = p.Fdim - k0, xr_kb = min(p.KB, (xr_cols + 15) >> 4), xr_nt = 64 * p.Fdim;
      [[maybe_unused]] const int xr_col = 16 * w + (c16_l & ~3);
      [[maybe_unused]] const int xr_lane = 4 * ((n0 + 4 * g4_l + (c16_l & 3)) * p.Fdim + k0 + 16 * w + (c16_l & ~3));
      [[maybe_unused]] __amdgpu_buffer_rsrc_t xr_rsrc;
      if constexpr (XR) xr_rsrc = xwg_buffer_n(p.grads + p.w_off, 4 * Nh * p.Fdim);
      const float* ap = XS + c16_l * FP + 4 * g4_l;
#pragma unroll
      for (int jj = 0; jj < MAXBLK; ++jj) {
        const int j = w + 8 * jj;
        if (j < p.KB) {            // (wavefront-uniform; no run-time bound inside)
          f32x4 acc[NT];
          // the block's weights: requested in front of the products, not one by one between the Adam steps
          // (eight LDS round trips in a row, each waited for: the compiler cannot tell the addresses apart)
          [[maybe_unused]] float wv[NT][4];
          if constexpr (!DP && !XR) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int v = 0; v < 4; ++v) wv[nt][v] = Wl[(16 * nt + 4 * g4_l + v) * WP + 16 * j + c16_l];
          }
          const float* bp = Ft + (16 * j + c16_l) * FP + 4 * g4_l;
          // Groups of 4 minibatch rows: n4 = ceil(B / 4) MFMAs per accumulator would do; the reference's
          // minibatch of 100 rows (n4 = 25: 6 steps and one MFMA) has its own straight-line variant, any
          // other size runs all 7 steps (the padding rows are zeros) -- a test per step costs more than
          // the three MFMAs it saves (every step its own basic block: 4.0 -> 4.6 us for the phase).
          // The NT accumulators advance TOGETHER, step by step: one F^T quad and NT d_out^T quads feed
          // 4 NT MFMAs, the quads of the next step are in flight meanwhile (accumulator after accumulator
          // the seven F^T quads had to stay in registers across the first chain and the scheduler, short of
          // registers, waited for every pair of reads right in front of its four MFMAs).  Each accumulator
          // still sums its steps in the same order.
          const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[nt] = zero;
          f32x4 qa[2][NT], qb[2];
          qb[0] = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) qa[0][nt] = *reinterpret_cast<const f32x4*>(ap + nt * 16 * FP);
#pragma unroll
          for (int S = 0; S < kUMT; ++S) {
            if (S + 1 < kUMT) {
              qb[(S + 1) & 1] = *reinterpret_cast<const f32x4*>(bp + 16 * (S + 1));
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                qa[(S + 1) & 1][nt] = *reinterpret_cast<const f32x4*>(ap + nt * 16 * FP + 16 * (S + 1));
            }
            __builtin_amdgcn_sched_barrier(0);      // (pinned: the reads of step S + 1 in front of the MFMAs of step S)
            const f32x4 b = qb[S & 1];
            if (S < kUMT - 1) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = umfma(qa[S & 1][nt][0], b[0], acc[nt]);
                acc[nt] = umfma(qa[S & 1][nt][1], b[1], acc[nt]);
                acc[nt] = umfma(qa[S & 1][nt][2], b[2], acc[nt]);
                acc[nt] = umfma(qa[S & 1][nt][3], b[3], acc[nt]);
              }
            } else {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) acc[nt] = umfma(qa[S & 1][nt][0], b[0], acc[nt]);
              if (n4 != 4 * (kUMT - 1) + 1) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                  acc[nt] = umfma(qa[S & 1][nt][1], b[1], acc[nt]);
                  acc[nt] = umfma(qa[S & 1][nt][2], b[2], acc[nt]);
                  acc[nt] = umfma(qa[S & 1][nt][3], b[3], acc[nt]);
                }
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          if (DP) {
            // this rank's share of the gradient: summed over the ranks by the caller
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                const int n = n0 + 16 * nt + 4 * g4_l + v, kc = k0 + 16 * j + c16_l;
                if (n < Nh && kc < p.Fdim) p.grads[p.w_off + (int64_t)n * p.Fdim + kc] = acc[nt][v];
              }
          } else if constexpr (XR) {
            // ... written through (the exchange stream's all-reduce reads it from memory), 16 bytes per
            // lane: the accumulator of a lane is four ROWS of one column, a 4 x 4 transpose inside the
            // lane quad makes that four columns of a row -- a cross-workgroup payload costs by its
            // memory transactions (dword stores: 58 us per update with the exchange instead of 34).
            // ONE lane offset for all blocks (block jj and tile nt: constants on top; rows beyond Nh fall
            // off the end of the buffer -- the bounds check sees the VGPR offset only, so nothing rides
            // in the scalar offset).
            if (j < xr_kb) {
#pragma unroll
              for (int nt = 0; nt < NT; ++nt) {
                const f32x4 q = quad_transpose4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3], c16_l & 3);
                xwg_store4s(xr_rsrc, xr_col + 128 * jj < xr_cols ? xr_lane + jj * 512 + nt * xr_nt : -1, 0, q);
              }
            }
          } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int v = 0; v < 4; v += 2)
                adam_weight2(acc[nt][v], acc[nt][v + 1], Mr[jj][nt][v], Mr[jj][nt][v + 1], Vr[jj][nt][v], Vr[jj][nt][v + 1],
                             wv[nt][v], wv[nt][v + 1], a0, a1, ak);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int v = 0; v < 4; ++v) Wl[(16 * nt + 4 * g4_l + v) * WP + 16 * j + c16_l] = wv[nt][v];
          }
        }
      }
      if (ks == 0 && w >= 8 - NT) {
        // biases of this block: sums of the d_out^T rows (lane (c16, g4) takes a quarter of the
        // minibatch, the quarters combine by butterfly shuffles), Adam on lanes < 16; 16-row block nt
        // on wavefront 8 - NT + nt.  BEHIND the wavefront's block of the product (wavefronts 4 .. 7 have
        // one block where 0 .. 3 have two: in front of it, all on wavefront 7, the sums put the k-slice-0
        // workgroups 0.5 us behind the others, and the owners wait for the last forward flag of the
        // chip), and with the seven reads of a sum in flight together (the scheduler, short of
        // registers, had serialized them: 14 LDS round trips).
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if (w != 8 - NT + nt) continue;
          const float* xp = XS + (16 * nt + c16_l) * FP + 4 * g4_l;
          f32x4 x4[kUMT];
#pragma unroll
          for (int S = 0; S < kUMT; ++S) x4[S] = *reinterpret_cast<const f32x4*>(xp + 16 * S);
          __builtin_amdgcn_sched_barrier(0);
          float g = 0.f;
#pragma unroll
          for (int S = 0; S < kUMT; ++S) g += (x4[S][0] + x4[S][1]) + (x4[S][2] + x4[S][3]);
          g += __shfl_xor(g, 16, 64);
          g += __shfl_xor(g, 32, 64);
          if (g4_l == 0) {
            const int n = n0 + 16 * nt + c16_l;
            if (DP) {
              if (n < Nh) p.grads[p.b_off + n] = g;
            } else if constexpr (XR) {
              if (n < Nh) xwg_store(p.grads + p.b_off + n, g);      // (Adam behind the exchange, below)
            } else {
              bw[nt] = adam_bias(g, bm[nt], bv[nt], bw[nt], a0, a1, ak);
              biasl[16 * nt + c16_l] = bw[nt];
            }
          }
        }
      }
      if constexpr (XR) {
        // ---- the exchange: gradients out (acknowledged), count, signal, wait, reduced gradients in ----
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        BSIG_USTAMP(13);
        if (tid_l == 0)      // (updates of THIS call whose gradients are out: the launch is the whole call)
          xr_hand_off(p.xr_count, p.xr_ready, p.xr_done, p.xr_base, (unsigned)t + 1u, (unsigned)p.G, flagp);
        // (a run that gave up finishes this update on whatever the buffer holds and leaves at the top of
        // the next one, the call fails with the sticky time-out error; a `break` HERE keeps every
        // loop-carried register alive on one more path: 300 bytes of scratch per lane, 58 us per update)
        (void)run_aborted(flagp, red, tid);
        BSIG_USTAMP(14);
        if (ks == 0 && w >= 8 - NT && g4_l == 0) {
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if (w != 8 - NT + nt) continue;
            const int n = n0 + 16 * nt + c16_l;
            const float g = n < Nh ? xwg_load(p.grads + p.b_off + n) : 0.f;
            bw[nt] = adam_bias(g, bm[nt], bv[nt], bw[nt], a0, a1, ak);
            biasl[16 * nt + c16_l] = bw[nt];
          }
        }
#pragma unroll
        for (int jj = 0; jj < MAXBLK; ++jj) {
          const int j = w + 8 * jj;
          if (j < p.KB) {
            float gr[NT][4];
            f32x4 gq[NT];
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)      // (rows beyond Nh: zeros; every lane takes part in the transposes)
              gq[nt] = j < xr_kb ? xwg_load4s(xr_rsrc, xr_col + 128 * jj < xr_cols ? xr_lane + jj * 512 + nt * xr_nt : -1, 0) : zero;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const f32x4 c = quad_transpose4(gq[nt][0], gq[nt][1], gq[nt][2], gq[nt][3], c16_l & 3);
#pragma unroll
              for (int v = 0; v < 4; ++v) gr[nt][v] = c[v];
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
              for (int v = 0; v < 4; ++v) {
                float* wp = Wl + (16 * nt + 4 * g4_l + v) * WP + 16 * j + c16_l;
                *wp = adam_weight(gr[nt][v], Mr[jj][nt][v], Vr[jj][nt][v], *wp, a0, a1, ak);
              }
          }
        }
      }
      BSIG_USTAMP(8);
      lds_barrier();                 // (the next minibatch tile stays in flight)
      BSIG_USTAMP(12);
    }
  }
  if constexpr (XR) {
    // a launch that gave up (a bounded poll timed out, here or in any workgroup) must not leave the
    // exchange stream waiting for gradients that will never come: every wait of the call passes
    if (slot == 0 && tid == 0 &&
        (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 2))
      __hip_atomic_store(p.xr_ready, p.xr_base + (unsigned)p.n_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }

  // ---- evaluations still owed at the end of the launch ------------------------------------------
  if (p.do_eval && !run_aborted(flagp, red, tid)) {
    if (pending_eval >= 0) {
      if (has_erow) u_owner_eval(p, XS, red, eo, pending_eval, u_eval_stream(p, o, pending_eval, false), o.a);
      pending_eval = -1;
    }
    // the evaluation after the last update of the call (a data-parallel rank: in the launch that
    // only takes the pending Adam step of that update)
    if (step0 + p.n_updates == p.n_total && (!DP || p.n_updates == 0)) {
      const int e = u_evals_before(p.n_total - 1, p.eval_every);
      __syncthreads();
      u_tile_eval<NT>(p, Wl, biasl, slot, ks, n0, k0, e);
      if (has_erow) u_owner_eval(p, XS, red, eo, e, u_eval_stream(p, o, e, true), o.a);
    }
  }
  // ---- write the tile back, advance the engine state -----------------------------------------
  if (!DP && has_tile) {
#pragma unroll
    for (int jj = 0; jj < MAXBLK; ++jj)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int n = n0 + 16 * nt + 4 * g4 + v, kc = 16 * (w + 8 * jj) + c16;
          if (w + 8 * jj < p.KB && n < Nh && k0 + kc < p.Fdim) {
            const int64_t off = p.w_off + (int64_t)n * p.Fdim + k0 + kc;
            p.params[off] = Wl[(16 * nt + 4 * g4 + v) * WP + kc];
            p.m1[off] = Mr[jj][nt][v]; p.m2[off] = Vr[jj][nt][v];
          }
        }
    if (ks == 0 && w >= 8 - NT && g4 == 0) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (w != 8 - NT + nt) continue;
        const int n = n0 + 16 * nt + c16;
        if (n < Nh) { p.params[p.b_off + n] = bw[nt]; p.m1[p.b_off + n] = bm[nt]; p.m2[p.b_off + n] = bv[nt]; }
      }
    }
  }
  if (p.prof && p.n_updates > 0 && tid == 0) {   // (diagnostics) the tile is written back
    __builtin_amdgcn_s_waitcnt(0);
    p.prof[((int64_t)wg * kUProf) * 16 + 13] = wall_clock64();
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
      n_ev = u_evals_before(step0 + p.n_updates, p.eval_every) - u_evals_before(step0, p.eval_every);
      if (!DP && step0 + p.n_updates == p.n_total && (p.n_total - 1) % p.eval_every != 0) ++n_ev;
    }
    reinterpret_cast<uint64_t*>(st + 8)[1] += (uint64_t)(p.n_updates + n_ev);
    st[0] = step0 + p.n_updates;
  }
}

// ---- a workgroup without a tile: row owner (and evaluation owner) only -------------------------
__device__ __forceinline__ void owner_only_workgroup(const UArgs& p, float* smem, const URole& role, int NBW, bool dp) {
  float* XS = smem + p.KS * kUFP + NBW * p.WP;      // (the layout of the tile workgroups)
  float* red = XS + p.xs_floats;
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  int32_t* flagp = p.state + 2;
  UOwn o;
  u_own_init(p, o, role.owner, false, XS, red, smem, lane0);
  const bool has_row = role.owner < p.n_owner;
  const int eo = (role.owner - p.n_owner + p.T) % p.T;
  const bool has_erow = p.do_eval && eo < p.NE;
  const int fk = u_fast_kind(p);                    // (one loop, the row picked per update: the evaluation code once)
  int pending_eval = -1;
  for (int t = 0; t < p.n_updates; ++t) {
    const int step = o.step0 + t;
    // (the time-out bit, sampled behind a barrier: every wavefront must take the same way out.  The
    // round trip sits in this workgroup's wait for the forward product.)
    if (run_aborted(flagp, red, tid)) break;
    BSIG_USTAMP(0);
    if (has_row) {
      switch (fk) {
        case 4 * 8 + 1: u_own_update_fast<4, 1>(p, o, t, w, lane0, wg); break;
        case 4 * 8 + 2: u_own_update_fast<4, 2>(p, o, t, w, lane0, wg); break;
        case 4 * 8 + 4: u_own_update_fast<4, 4>(p, o, t, w, lane0, wg); break;
        case 8 * 8 + 1: u_own_update_fast<8, 1>(p, o, t, w, lane0, wg); break;
        case 8 * 8 + 2: u_own_update_fast<8, 2>(p, o, t, w, lane0, wg); break;
        case 8 * 8 + 4: u_own_update_fast<8, 4>(p, o, t, w, lane0, wg); break;
        case 16 * 8 + 1: u_own_update_fast<16, 1>(p, o, t, w, lane0, wg); break;
        case 16 * 8 + 2: u_own_update_fast<16, 2>(p, o, t, w, lane0, wg); break;
        case 16 * 8 + 4: u_own_update_fast<16, 4>(p, o, t, w, lane0, wg); break;
        default: u_own_update(p, o, t, w, lane0, wg); break;
      }
    }
    if (__builtin_expect(pending_eval >= 0, 0)) {
      __syncthreads();
      if (has_erow) u_owner_eval(p, XS, red, eo, pending_eval, u_eval_stream(p, o, pending_eval, false), o.a);
      pending_eval = -1;
    }
    if (__builtin_expect(p.do_eval && step > 0 && (step - 1) % p.eval_every == 0, 0))
      pending_eval = u_evals_before(step, p.eval_every) - 1;
    __syncthreads();
  }
  if (p.do_eval && !run_aborted(flagp, red, tid)) {
    if (pending_eval >= 0 && has_erow)
      u_owner_eval(p, XS, red, eo, pending_eval, u_eval_stream(p, o, pending_eval, false), o.a);
    if (o.step0 + p.n_updates == p.n_total && (!dp || p.n_updates == 0)) {
      const int e = u_evals_before(p.n_total - 1, p.eval_every);
      __syncthreads();
      if (has_erow) u_owner_eval(p, XS, red, eo, e, u_eval_stream(p, o, e, true), o.a);
    }
  }
}

template <bool DP, int NT, bool XR = false>
__global__ __launch_bounds__(kUT) void linear_head_updates_kernel(UArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifndef BSIG_HOST_SAN_BUILD   // (see fit_persistent_mdnn.hip)
  const URole role = wg_role(blockIdx.x, p.T, p.G);
  if (role.tile < 0) owner_only_workgroup(p, smem, role, 16 * NT, DP);
  else unified_workgroup<DP, NT, XR>(p, smem, role);
#endif
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
  BSIG_REQUIRE(blocks >= 1 && ms >= 0 && lds_bytes <= (size_t)kULds, "debug_spin: bad args");
  BSIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kULds));
  hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), std::max<size_t>(lds_bytes, 1024), st,
                     (long long)ms * 100000LL /* 100 MHz wall clock */, (int*)nullptr);
  BSIG_CHECK_LAUNCH("debug_spin");
  return BSIG_OK;
}

static long long* g_prof = nullptr;
void persist_set_profile_buffer(void* buf) { g_prof = reinterpret_cast<long long*>(buf); }
void* persist_profile_buffer() { return g_prof; }

// ---------------------------------------------------------------- host side
struct UGeom {
  int NT, Bp, MT, FP, KS, ksteps, KB, WP, Nh, NhP, n_blocks, k_slices, G, T, n_owner, R;
  int per_wave, xs_floats;
  int eval_passes, NE, RE;      // eval_passes 0: evaluations stay outside the launches
  size_t lds, slab_floats, dout_floats, eval_floats;
};

// One candidate tiling (NT 16-row blocks per tile, k-slices of KS columns) -> geometry and a
// cost estimate (us per update, relative).
static bool u_geom_try(const PersistShape& s, int NT, int KS, UGeom* g, double* cost) {
  const int NBW = 16 * NT, max_blk = NT == 1 ? 3 : 2;
  g->Bp = (int)round_up(s.batch, 16);
  g->MT = g->Bp / 16;
  g->FP = kUFP;
  g->Nh = s.n_comp + 2 * s.out_dim * s.n_comp;
  g->per_wave = g->Nh + s.out_dim + 3 * s.n_comp;
  g->NT = NT;
  g->n_blocks = ceil_div(g->Nh, NBW);
  g->NhP = g->n_blocks * NBW;
  g->KS = KS;
  g->ksteps = g->KB = KS / 16;
  if (g->ksteps > kUSteps || g->KB > 8 * max_blk) return false;
  g->WP = KS + 4;                     // = 4 (mod 16): conflict-free 16-byte reads down 16 rows
  g->k_slices = ceil_div(s.feat_dim, KS);
  g->G = g->n_blocks * g->k_slices;
  if (g->G > kXwgMax) return false;
  // Row owners.  Workgroups without a tile first: one row each if the chip has that many CUs to
  // spare, else two; else the owners are (also) tile workgroups, one row each.
  const int spare = kXwgMax - g->G;
  bool mixed = false;
  static const bool force_mixed = [] { const char* e = getenv("BSIG_PERSIST_MIXED"); return e && e[0] == '1'; }();   // (A/B runs)
  if (!force_mixed && spare >= s.batch) { g->R = 1; g->T = g->G + s.batch; }
  else if (!force_mixed && spare >= ceil_div(s.batch, 2)) { g->R = 2; g->T = g->G + ceil_div(s.batch, 2); }
  else {
    mixed = true;
    g->T = std::max(g->G, std::min(s.batch, kXwgMax));
    g->R = ceil_div(s.batch, std::min(g->T, s.batch));
    if (g->R > kUT / 64) return false;
  }
  g->n_owner = ceil_div(s.batch, g->R);
  // in-launch evaluations: one wavefront per held-out row; the workgroups without a minibatch
  // row take them if that costs at most twice the rows per workgroup
  g->eval_passes = s.max_test > 0 ? ceil_div(s.max_test, s.batch) : 0;
  g->NE = 1; g->RE = 1;
  if (g->eval_passes > 0) {
    const int idle = g->T - g->n_owner;
    const int re_all = ceil_div(s.max_test, g->T);
    const int re_idle = idle > 0 ? ceil_div(s.max_test, idle) : 1 << 20;
    if (re_idle <= kUT / 64 && re_idle <= 2 * re_all) { g->NE = idle; g->RE = re_idle; }
    else { g->NE = g->T; g->RE = re_all; }
    if (g->RE > kUT / 64) { g->eval_passes = 0; g->NE = 1; g->RE = 1; }
    else g->NE = std::min(g->NE, ceil_div(s.max_test, g->RE));   // (no owner without a row)
  }
  auto lds_of = [&] {
    g->xs_floats = (int)round_up(std::max(NBW * g->FP, std::max(g->R, g->RE) * g->per_wave), 4);
    g->lds = ((size_t)g->KS * g->FP + (size_t)NBW * g->WP + g->xs_floats + 64 + NBW) * sizeof(float);
  };
  lds_of();
  if (g->lds > (size_t)kULds && g->eval_passes > 0 && g->RE > g->R) {
    // (the evaluation rows may be what does not fit: retry without them)
    g->eval_passes = 0; g->NE = 1; g->RE = 1;
    lds_of();
  }
  if (g->lds > (size_t)kULds) return false;
  g->slab_floats = (size_t)g->k_slices * s.batch * g->NhP;
  g->dout_floats = (size_t)s.batch * g->NhP;
  g->eval_floats = (size_t)3 * g->eval_passes * g->slab_floats;
  // cost: the two matrix phases (two wavefronts share a SIMD's pipe in the forward product, the
  // dW blocks are dealt to 4 SIMDs; 32 cycles per 16x16x4 at ~2.2 GHz), the minibatch tile's way
  // through a CU's vector-memory pipe (~20 KB/us) where it cannot hide in a wait (owners that also
  // hold a tile), the owners' k-slice sum
  const double mfma_us = 32.0 * (2.0 * 4 * g->ksteps * NT + 4.0 * kUMT * NT * ceil_div(g->KB, 4)) / 2200.0;
  const double tile_us = mixed ? (double)s.batch * KS * 4 / 20000.0 : 0.0;
  const double sum_us = 0.035 * g->k_slices * g->R;
  *cost = mfma_us + tile_us + sum_us;
  return true;
}

static bool u_geom(const PersistShape& s, UGeom* g) {
  if (s.batch < 1 || s.batch > 16 * kUMT || s.feat_dim < 4 || s.feat_dim % 4 != 0 || s.out_dim < 1 ||
      s.n_comp < 1 || s.n_comp > 64)
    return false;
  const int groups = 64 / s.n_comp;
  if (ceil_div(s.out_dim, groups) > kElemsPerLane) return false;   // diag_row's register cache
  bool found = false;
  double best = 0.0;
  // (A/B runs: BSIG_PERSIST_NT / BSIG_PERSIST_KS pin the tiling)
  static const int force_nt = [] { const char* e = getenv("BSIG_PERSIST_NT"); return e ? atoi(e) : 0; }();
  static const int force_ks = [] { const char* e = getenv("BSIG_PERSIST_KS"); return e ? atoi(e) : 0; }();
  for (int NT = 1; NT <= 2; ++NT)
    for (int groups6 = 1; groups6 <= kUSteps / kUGroup; ++groups6) {   // k-slices of 96 / 192 / 288 columns
      UGeom c;
      double cost = 0.0;
      if ((force_nt && NT != force_nt) || (force_ks && 16 * kUGroup * groups6 != force_ks)) continue;
      if (!u_geom_try(s, NT, 16 * kUGroup * groups6, &c, &cost)) continue;
      if (!found || cost < best) { *g = c; best = cost; found = true; }
    }
  return found;
}

int persist_geometry(const PersistShape& s, int32_t* out) {
  UGeom g;
  for (int i = 0; i < 16; ++i) out[i] = 0;
  if (!u_geom(s, &g)) return 0;
  const int32_t v[13] = {g.NT, g.KS, g.n_blocks, g.k_slices, g.G, g.T, g.n_owner, g.R, g.NE, g.RE,
                         g.eval_passes, (int32_t)g.lds, std::max(0, g.n_owner - (g.T - g.G))};
  for (int i = 0; i < 13; ++i) out[i] = v[i];
  return 1;
}

template <int NT>
static bool u_can_host(const UGeom& g) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  if (prop.multiProcessorCount < g.T || (size_t)prop.maxSharedMemoryPerMultiProcessor < g.lds) return false;
  // every workgroup of the launch must be resident at once (they wait for each other): the
  // runtime's own occupancy answer for this kernel at this LDS size must admit a workgroup per CU
  int per_cu = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(linear_head_updates_kernel<false, NT>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, kULds) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, linear_head_updates_kernel<false, NT>, kUT, g.lds) != hipSuccess)
    return false;
  return per_cu >= 1;
}
static bool u_device_can_host(const UGeom& g) { return g.NT == 1 ? u_can_host<1>(g) : u_can_host<2>(g); }

// (asked on every launch: the answers -- a device query each -- are kept per shape and device)
int persist_variant(const PersistShape& s) {
  struct Key { int dev, batch, feat_dim, out_dim, n_comp, max_test, variant; };
  static thread_local Key cache[8];
  static thread_local int n_cache = 0, next = 0;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) dev = -1;
  for (int i = 0; i < n_cache; ++i) {
    const Key& k = cache[i];
    if (k.dev == dev && k.batch == s.batch && k.feat_dim == s.feat_dim && k.out_dim == s.out_dim &&
        k.n_comp == s.n_comp && k.max_test == s.max_test)
      return k.variant;
  }
  UGeom g;
  int v = 0;
  // (2: this file's unified workgroups; 0: the per-phase kernels.  1 was fit_persistent_v1.hip -- the
  // round-1..3 decomposition, no shipped configuration reached it, retired in round 6)
  if (u_geom(s, &g) && u_device_can_host(g)) v = 2;
  cache[next] = Key{dev, s.batch, s.feat_dim, s.out_dim, s.n_comp, s.max_test, v};
  next = (next + 1) % 8;
  n_cache = std::min(n_cache + 1, 8);
  return v;
}

bool persist_supported(const PersistShape& s) { return persist_variant(s) != 0; }
bool persist_eval_supported(const PersistShape& s) {
  const int v = persist_variant(s);
  UGeom g;
  return v == 2 && u_geom(s, &g) && g.eval_passes > 0;
}

static size_t u_data_bytes(const UGeom& g) {
  return round_up<size_t>((g.slab_floats + 2 * g.dout_floats + g.eval_floats) * sizeof(float), 256);
}
// flags, granules and (last 256 bytes) the word the tile workgroups of a resident rank count themselves in
static size_t u_sync_bytes() { return 2 * kFlagArr * sizeof(unsigned) + (6 + 16) * kGranArr * 8 + 256; }

size_t persist_workspace_bytes(const PersistShape& s) {
  const int v = persist_variant(s);
  UGeom g;
  if (v != 2 || !u_geom(s, &g)) return 0;
  return u_data_bytes(g) + u_sync_bytes();
}

int persist_reset_regions(const PersistShape& s, void* workspace, size_t workspace_bytes,
                          ZeroRegion* regions) {
  UGeom g;
  BSIG_REQUIRE(u_geom(s, &g), "persistent updates: shape not covered");
  BSIG_REQUIRE(workspace && workspace_bytes >= persist_workspace_bytes(s),
               "persistent updates: workspace too small");
  char* base = reinterpret_cast<char*>(workspace);
  const size_t slab_bytes = g.slab_floats * sizeof(float);
  // d_out / exp(pre) rows (their padding columns stay zero), flags and granules
  regions[0] = ZeroRegion{base + slab_bytes, 2 * g.dout_floats * sizeof(float)};
  regions[1] = ZeroRegion{base + u_data_bytes(g), u_sync_bytes()};
  return BSIG_OK;
}

template <int NT>
static int u_launch(const UGeom& g, const UArgs& p, bool dp, int n, bool do_eval, hipStream_t st) {
  // the > 64 KB dynamic-LDS attribute is per device (the plan's device is the current one:
  // the Python mirror enters the model's device around every call)
  static bool attr_set_dev[64] = {};
  int attr_dev = 0;
  BSIG_HIP(hipGetDevice(&attr_dev));
  bool& attr_set = attr_set_dev[attr_dev & 63];
  if (!attr_set) {
    BSIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_head_updates_kernel<false, NT>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kULds));
    BSIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_head_updates_kernel<true, NT>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kULds));
    BSIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(linear_head_updates_kernel<false, NT, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, kULds));
    attr_set = true;
  }
  if (p.xr_ready)
    hipLaunchKernelGGL((linear_head_updates_kernel<false, NT, true>), dim3(g.T), dim3(kUT), g.lds, st, p);
  else if (dp)
    hipLaunchKernelGGL((linear_head_updates_kernel<true, NT>), dim3(g.T), dim3(kUT), g.lds, st, p);
  else
    hipLaunchKernelGGL((linear_head_updates_kernel<false, NT>), dim3(g.T), dim3(kUT), g.lds, st, p);
  BSIG_CHECK_LAUNCH("linear_head_updates");
  return BSIG_OK;
}

int persist_run(const PersistShape& s, const PersistBuffers& b, const PersistHyper& hy, int n,
                hipStream_t st) {
  UGeom g;
  BSIG_REQUIRE(u_geom(s, &g), "persistent updates: shape not covered");
  BSIG_REQUIRE(b.feats && b.y && b.ids && b.params && b.exp_avg && b.exp_avg_sq && b.state &&
                   b.train_loss && b.workspace, "persistent updates: null buffer");
  BSIG_REQUIRE(b.workspace_bytes >= persist_workspace_bytes(s),
               "persistent updates: workspace too small");
  BSIG_REQUIRE(b.ld_feats % 4 == 0 && aligned(b.feats, 16) && b.ld_feats >= s.feat_dim,
               "persistent updates: features must be 16-byte aligned rows");
  BSIG_REQUIRE(!(b.adam_pending && !b.grads), "persistent updates: pending Adam step without gradients");
  BSIG_REQUIRE(!(b.grads && n > 1 && !b.xr_ready), "persistent updates: data-parallel launches take one update");
  BSIG_REQUIRE(!(b.xr_ready && !(b.grads && b.xr_done && !b.adam_pending && b.n_total == n)),
               "persistent updates: a resident data-parallel launch takes the whole call");
  BSIG_REQUIRE(!(b.xr_ready && !(b.w_off % 4 == 0 && s.feat_dim % 4 == 0 && aligned(b.grads, 16) &&
                                 (int64_t)g.Nh * s.feat_dim < ((int64_t)1 << 29))),
               "persistent updates: a resident data-parallel launch moves 16-byte gradient quads");
  if (n <= 0 && !b.adam_pending && !b.do_eval) return BSIG_OK;
  UArgs p{};
  p.B = s.batch; p.Bp = g.Bp; p.MT = g.MT;
  p.Fdim = s.feat_dim; p.KS = g.KS; p.ksteps = g.ksteps; p.KB = g.KB; p.WP = g.WP;
  p.Nh = g.Nh; p.NhP = g.NhP; p.D = s.out_dim; p.K = s.n_comp;
  p.n_blocks = g.n_blocks; p.k_slices = g.k_slices; p.G = g.G; p.T = g.T; p.n_owner = g.n_owner; p.R = g.R;
  p.n_updates = std::max(n, 0); p.xs_floats = g.xs_floats;
  {
    const char* e = getenv("BSIG_PERSIST_FAST_ROWS");     // (read per launch -- one launch per call: A/B runs, tests)
    p.fast_rows = (e && e[0] == '0') ? 0 : 1;
  }
  p.grads = b.grads; p.adam_pending = b.adam_pending;
  p.xr_ready = b.xr_ready; p.xr_done = b.xr_done; p.xr_base = b.xr_base;
  p.xr_count = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(b.workspace) + u_data_bytes(g) + u_sync_bytes() - 256);
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
  char* sync = base + u_data_bytes(g);
  p.flag_fwd = reinterpret_cast<unsigned*>(sync);
  p.flag_eval = p.flag_fwd + kFlagArr;
  p.gran = reinterpret_cast<unsigned long long*>(sync + 2 * kFlagArr * sizeof(unsigned));
  p.gran_eval = p.gran + 3 * kGranArr;
  p.gran_rep = p.gran + 6 * kGranArr;
  p.NE = g.NE; p.RE = g.RE;
  if (b.do_eval) {
    BSIG_REQUIRE(g.eval_passes > 0 && b.n_test >= 1 && b.n_test <= g.eval_passes * s.batch &&
                     b.n_test <= g.NE * g.RE && b.y_test && b.test_loss && b.eval_every >= 1 && b.n_total >= 1,
                 "persistent updates: in-launch evaluation not covered");
    p.do_eval = 1; p.eval_every = b.eval_every; p.n_total = b.n_total; p.n_test = b.n_test;
    p.eval_passes = ceil_div(b.n_test, s.batch); p.eval_row0 = b.eval_row0;
    p.y_test = b.y_test; p.ldy_test = b.ldy_test; p.test_loss = b.test_loss;
    // (the slab buffers are laid out for this call's passes: [3][eval_passes][k_slices][B][NhP])
  }
  p.prof = reinterpret_cast<long long*>(persist_profile_buffer());
  if (p.prof) {
    const char* t0 = getenv("BSIG_PROF_T0");
    p.prof_t0 = t0 ? atoi(t0) : 0;
  }
  if (b.xr_ready && !p.n_total) p.n_total = n;      // (the release of a launch that gave up: every wait passes)
  return g.NT == 1 ? u_launch<1>(g, p, b.grads != nullptr, n, b.do_eval != 0, st)
                   : u_launch<2>(g, p, b.grads != nullptr, n, b.do_eval != 0, st);
}

}  // namespace bsig
