// Persistent update kernel of the two-layer MDNN whose FIRST LAYER DOES NOT FIT THE CHIP:
// the cross-correlation summaries of cfg/anymal.yaml:103-109 (I = 56 402) and
// cfg/shadow_hand_more.yaml:73-81 (I = 105 002, a 13.4 M-parameter first layer: 161 MB of
// weights + Adam moments) -- summarizers.py:112-119 into mdnn.py:71,108, updated by
// mdnn.py:219-233.  The row-owner and small-weight workgroups are those of the resident
// kernel (persist_mdnn_device.h); the tile workgroups here STREAM W1 and its moments:
//
//  * tile workgroup g of T (= the CUs the owners and the small-weight workgroups leave, a multiple
//    of 4) owns 32 hidden units (g mod 4) and walks the 256-column chunks [q*C/G, (q+1)*C/G), q = g / 4,
//    G = T / 4: 32 rows x 1 KB per chunk and array, 16-byte accesses.  The minibatch never
//    exists as a [B, I] tile: its FACTOR rows (sf | af | mean std, 1.3-4.6 KB per row
//    instead of 226-420 KB) sit in LDS -- row-major for the forward product's A operand,
//    transposed for the weight gradient's B operand -- and both MFMA operand streams form
//    x[b, i*A + j] = sf[b, i] * af[b, j] on the fly: the single fp32 multiply the summarizer
//    itself does.
//  * ONE pass over its chunks per update, taking Adam step t and the forward product of
//    minibatch t+1 from the same read of W1 (the minibatch ids are known ahead):
//        W, m, v (chunk) -> registers (rows of 16-byte quads)
//        dW  = dz1_t^T X_t[:, chunk]          (MFMA 32x32x2; dz1_t in registers) -> LDS
//        Adam -> W', m', v' -> memory;  W' -> LDS
//        P_{t+1} += X_{t+1}[:, chunk] W'^T    (MFMA 32x32x2; accumulators live over the pass)
//    W1 crosses HBM 6 times per update (W, m, v in and out) instead of 7, the two products
//    run in the shadow of that stream, and no [B, I] summary, staging copy or gather exists.
//  * the G partial products [B, 32] of a block of hidden units are summed by that block's tile
//    workgroups themselves (each owns B*32/G consecutive elements) and reach the owners as one
//    slab [B, 128], bias included.
//
// Per update: pass -> slabs -> sum -> owners (h1 .. NLL .. dz1) -> pass.  Every sum is taken
// in a fixed order: runs are bitwise reproducible.  Data-parallel ranks (one update per
// launch): forward pass, owners, then a backward pass that writes dW to the flat gradient
// buffer; the Adam step follows the caller's all-reduce as a flat kernel.
#include "persist_mdnn_device.h"

namespace bsig {

constexpr int kSC = 256;              // columns of W1 per chunk (1 KB of a row: what the HBM likes)
constexpr int kSPitch = kSC + 4;      // LDS pitch of the chunk's 32 weight rows (forward B operand)
constexpr int kSTP = 108;             // LDS pitch of a transposed factor column (104 rows + 4)
constexpr int kSBP = 36;              // LDS pitch of a [rows][32] block (dz1 staging, slab staging)
constexpr int kSRedGroups = 32;       // slab groups of the cross-workgroup sum
typedef f32x4 __attribute__((aligned(4))) f32x4u;   // rows of W1 are only 8-byte aligned (I = S*A + 2)

// k / a for 0 <= k < 2^24 without an integer division
__device__ __forceinline__ int div_small(int k, int a, float ra) {
  int i = (int)((float)k * ra);
  if (i * a > k) --i;
  if ((i + 1) * a <= k) ++i;
  return i;
}

// (diagnostics) per-wavefront stamps of tile workgroup 3 in update 4, second chunk: row 254 of the
// profile buffer (no workgroup 254 exists on a 256-CU chip with owners and small-weight workgroups)
#define BSIG_WSTAMP(k)                                                                      \
  do {                                                                                      \
    if (p.prof && g == 3 && t == 4 && c == c_lo + 1 && (threadIdx.x & 63) == 0)             \
      p.prof[((int64_t)254 * kMProfUpdates + (threadIdx.x >> 6)) * 16 + (k)] = wall_clock64(); \
  } while (0)

// (diagnostics) the shader clock beside the 100 MHz wall clock: the engine frequency during the pass
#define BSIG_WCLK(k)                                                                        \
  do {                                                                                      \
    if (p.prof && g == 3 && t == 4 && c == c_lo + 1 && (threadIdx.x & 63) == 0)             \
      p.prof[((int64_t)254 * kMProfUpdates + (threadIdx.x >> 6)) * 16 + (k)] = (int64_t)__builtin_readcyclecounter(); \
  } while (0)

template <bool DP>
__device__ __forceinline__ void mdnn_stream_tile_workgroup(const MdnnArgs& p, float* smem) {
  const int PF = p.s_pf, NIP = p.s_nip, FR = p.FR, B = p.B;
  float* Fr = smem;                          // [FR][PF] factor rows of the minibatch of the forward product
  float* Ft = Fr + FR * PF;                  // [NIP + A + 8][kSTP] factor columns of the minibatch of dW;
                                             // after the pass: the [B][kSBP] block of partial products
  float* Wc = Ft + (NIP + p.xA + 8) * kSTP;  // [32][kSPitch] this chunk: dW, then the new weights; also
                                             // the dz1 block [FR][kSBP], the k-half exchange, partial sums
  float* red = Wc + kMNB * kSPitch;          // [64]
  float* b1s = red + 64;                     // [32] b1 of this block as of the forward product being summed
  int* Tb = reinterpret_cast<int*>(b1s + kMNB);   // [chunks of this workgroup][2 k-halves][2 lane halves][16 steps][2]
  // (lane-derived indices are laundered once per update -- below -- so that the address arithmetic
  // built on them is recomputed where it is used instead of being kept live, and spilled, across
  // the update loop: a spilled address comes back as scratch_load + s_waitcnt vmcnt(0) in front of
  // its load, which serialises every batch of loads -- 17 us for the 39 loads of the factor rows)
  int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int h = lane >> 5, l31 = lane & 31;
  // workgroup g: hidden units [32 nb, 32 nb + 32), column chunks [c_lo, c_hi) of 256
  const int g = blockIdx.x, T = p.G1, G = T >> 2;
  const int nb = g & 3, gq = g >> 2;
  const int c_lo = (int)((int64_t)gq * p.s_chunks / G), c_hi = (int)((int64_t)(gq + 1) * p.s_chunks / G);
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
  // roles of wavefront w: weight-gradient tile = columns [32 w, 32 w + 32) of the chunk;
  // forward tile = minibatch rows [32 mt, 32 mt + 32) over the k-half kh of the chunk
  int mt = w & 3, kh = w >> 2;
  // quad layout of the chunk for loads / Adam / stores: a WAVEFRONT owns the 32 x 32 block it
  // computes the gradient of (columns 32 w ..): lane <-> rows (lane >> 3) + 8u, columns cl .. cl + 3 --
  // gradient and weights change hands inside the wavefront, no workgroup barrier in between
  int rl = lane >> 3, cl = w * 32 + (lane & 7) * 4;
  auto relaunder = [&]() {
    asm volatile("" : "+v"(tid));
    lane = tid & 63; w = __builtin_amdgcn_readfirstlane(tid >> 6);
    h = lane >> 5; l31 = lane & 31; mt = w & 3; kh = w >> 2;
    rl = lane >> 3; cl = w * 32 + (lane & 7) * 4;
  };

  // b1 of this block: lanes 0-31 of wavefront 0; every workgroup of the block takes the same Adam
  // steps (same values, same order), the first one (gq == 0) writes back
  const bool bias_lane = tid < kMNB;
  float bw = 0.f, bm = 0.f, bv = 0.f;
  if (bias_lane) {
    const int64_t off = p.b1_off + nb * kMNB + tid;
    bw = p.params[off];
    if (!fresh && !DP) { bm = p.m1[off]; bv = p.m2[off]; }
    b1s[tid] = bw;
  }

  // ---- factor rows of the minibatch starting at id-table row `row0` -> Fr (row-major): slots
  //      [0, NIP) hold sf[i_lo ..] (i == S: the "1" beside mean / std), then af | mean | std | zeros.
  //      All loads of a wavefront's 13 rows are in flight together.
  // (direct: rows row0 .. row0 + rows - 1 of x themselves -- the held-out pairs of an evaluation pass --
  // instead of the rows the id table names)
  auto load_factor_rows = [&](int64_t row0, bool direct = false, int rows = 0) {
    const int nrow = direct ? rows : B;
    const int per_row = NIP + A + 8;
    constexpr int kRows = 13, kCols = 3;                    // rows per wavefront, 64-column groups
    float v[kRows][kCols];
    int fr[kRows];
#pragma unroll
    for (int r = 0; r < kRows; ++r)
      fr[r] = direct ? (int)(row0 + min(w + 8 * r, nrow - 1)) : p.ids[row0 + min(w + 8 * r, B - 1)];
#pragma unroll
    for (int r = 0; r < kRows; ++r) asm volatile("" : "+v"(fr[r]));
    // (unconditional loads from clamped addresses, the selects afterwards: a load under a branch
    // whose other side writes the same register is waited for on the spot)
    int ofs[kCols];
    bool live[kCols];
#pragma unroll
    for (int u = 0; u < kCols; ++u) {
      const int c = lane + 64 * u;
      const int i = i_lo + c, j = c - NIP;
      // i == S: the "1" the summarizer stores behind mean and std (index S + A + 2)
      ofs[u] = c < NIP ? (i < S ? i : S + A + 2) : S + min(j, A + 1);
      live[u] = c < NIP ? i <= S : j < A + 2;
    }
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const float* src = direct ? p.xe + (int64_t)fr[r] * p.ldxe : p.x + (int64_t)fr[r] * p.ldx;
#pragma unroll
      for (int u = 0; u < kCols; ++u) v[r][u] = src[ofs[u]];
    }
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const int b = w + 8 * r;
#pragma unroll
      for (int u = 0; u < kCols; ++u) {
        const int c = lane + 64 * u;
        if (b < FR && c < per_row) Fr[b * PF + c] = (b < nrow && live[u]) ? v[r][u] : 0.f;
      }
    }
  };
  // Fr (row-major, minibatch t) -> Ft (column-major: the B operand of dW reads 4 rows per load)
  auto transpose_factors = [&]() {
    const int per_row = NIP + A + 8;
    for (int c = w; c < per_row; c += kMT / 64) {
      for (int b = lane; b < kSTP; b += 64) Ft[c * kSTP + b] = b < FR ? Fr[b * PF + c] : 0.f;
    }
  };

  // ---- the chunk's 32 x 256 block in quads (16-byte loads and stores, 8 rows x 128 B per wavefront
  //      instruction, the eight wavefronts side by side on 1 KB of each row)
  float W4[4][4], M4[4][4], V4[4][4];
  auto chunk_ptr = [&](float* base, int c, int u) {
    return base + p.w1_off + (int64_t)(nb * kMNB + rl + 8 * u) * I + (int64_t)c * kSC + cl;
  };
  // valid columns of my quad in chunk c: 4, 2 (I = S*A + 2 ends in the middle of a quad) or 0
  auto quad_valid = [&](int c) { return min(max(I - (c * kSC + cl), 0), 4); };
  auto ld4 = [&](const float* q, int nv, float (&r)[4]) {
    r[0] = r[1] = r[2] = r[3] = 0.f;
    if (nv == 4) { const f32x4 t4 = *reinterpret_cast<const f32x4u*>(q); r[0] = t4.x; r[1] = t4.y; r[2] = t4.z; r[3] = t4.w; }
    else if (nv >= 2) { const float2 t2 = *reinterpret_cast<const float2*>(q); r[0] = t2.x; r[1] = t2.y; }
  };
  auto st4 = [&](float* q, int nv, const float (&r)[4]) {
    if (nv == 4) { const f32x4 t4 = {r[0], r[1], r[2], r[3]}; *reinterpret_cast<f32x4u*>(q) = t4; }
    else if (nv >= 2) *reinterpret_cast<float2*>(q) = make_float2(r[0], r[1]);
  };
  auto load_chunk = [&](int c, bool moments) {
    const int nv = quad_valid(c);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ld4(chunk_ptr(p.params, c, u), nv, W4[u]);
      ld4(chunk_ptr(p.m1, c, u), moments ? nv : 0, M4[u]);
      ld4(chunk_ptr(p.m2, c, u), moments ? nv : 0, V4[u]);
    }
  };
  auto store_quad_full = [&](int c, int u) {
    st4(chunk_ptr(p.params, c, u), 4, W4[u]);
    st4(chunk_ptr(p.m1, c, u), 4, M4[u]);
    st4(chunk_ptr(p.m2, c, u), 4, V4[u]);
  };
  auto load_quad_full = [&](int c, int u) {
    ld4(chunk_ptr(p.params, c, u), 4, W4[u]);
    ld4(chunk_ptr(p.m1, c, u), 4, M4[u]);
    ld4(chunk_ptr(p.m2, c, u), 4, V4[u]);
  };
  // (every chunk but the matrix's last one is 256 whole columns: one uniform test instead of a
  // per-lane width on each of the 24 memory instructions of a chunk)
  auto store_quad = [&](int c, int u) {
    if (__builtin_expect((c + 1) * kSC <= I, 1)) {
      st4(chunk_ptr(p.params, c, u), 4, W4[u]);
      st4(chunk_ptr(p.m1, c, u), 4, M4[u]);
      st4(chunk_ptr(p.m2, c, u), 4, V4[u]);
    } else {
      const int nv = quad_valid(c);
      st4(chunk_ptr(p.params, c, u), nv, W4[u]);
      st4(chunk_ptr(p.m1, c, u), nv, M4[u]);
      st4(chunk_ptr(p.m2, c, u), nv, V4[u]);
    }
  };
  auto load_quad = [&](int c, int u, bool moments) {
    if (__builtin_expect((c + 1) * kSC <= I && moments, 1)) {
      ld4(chunk_ptr(p.params, c, u), 4, W4[u]);
      ld4(chunk_ptr(p.m1, c, u), 4, M4[u]);
      ld4(chunk_ptr(p.m2, c, u), 4, V4[u]);
    } else {
      const int nv = quad_valid(c);
      ld4(chunk_ptr(p.params, c, u), nv, W4[u]);
      ld4(chunk_ptr(p.m1, c, u), moments ? nv : 0, M4[u]);
      ld4(chunk_ptr(p.m2, c, u), moments ? nv : 0, V4[u]);
    }
  };
  auto store_chunk = [&](int c) {
    const int nv = quad_valid(c);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      st4(chunk_ptr(p.params, c, u), nv, W4[u]);
      st4(chunk_ptr(p.m1, c, u), nv, M4[u]);
      st4(chunk_ptr(p.m2, c, u), nv, V4[u]);
    }
  };
  auto weights_to_lds = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f32x4 t4 = {W4[u][0], W4[u][1], W4[u][2], W4[u][3]};
      *reinterpret_cast<f32x4*>(Wc + (rl + 8 * u) * kSPitch + cl) = t4;
    }
  };

  // ---- forward product of one chunk (weights in Wc): rows mt, columns [128 kh + kk_lo, 128 kh + kk_hi).
  //      Operands of step kk are fetched from LDS while the MFMAs of step kk - 8 run (the two
  //      wavefronts of a SIMD execute this in lock-step behind the barrier).
  f32x16 facc;
  struct FwdOps { float sfv; float4 af4, b4; };
  // position of a quad of columns k4 = i*A + ai (i <= S; i == S: the mean / std tail, ai = k4 - S*A)
  struct FwdPos { int i, ai; };
  auto forward_pos = [&](int k4) {
    FwdPos q;
    if (k4 >= SA) { q.i = S; q.ai = k4 - SA; }
    else { q.i = div_small(k4, A, rA); q.ai = k4 - q.i * A; }
    return q;
  };
  // c_st >= 0: quads q_from.. of that chunk's block (in W4 / M4 / V4) are stored behind steps 1..4;
  // c_ld >= 0: quads q_from.. of the next chunk's block are loaded behind steps 6..9
  //
  // On a SIMD the VALU instructions of its two wavefronts do NOT hide behind their MFMAs here
  // (measured: the phase costs MFMAs + VALU; 18 address instructions per step were 20 % of it), so a
  // step is 4 products + 2 address additions: where the factors of a step's columns sit in a factor
  // row (the (i, j) walk with its wrap at j = A and the mean / std tail) is the same in every update
  // -- byte offsets per (chunk, k-half, lane half, step) are tabulated in LDS once per launch.
  auto forward_chunk = [&](int c, int kk_lo, int kk_hi, int c_st, int q_from, int c_ld, bool moments) {
    const char* frow = reinterpret_cast<const char*>(Fr + min(mt * 32 + l31, FR - 1) * PF);
    const float* wrow = Wc + l31 * kSPitch + kh * (kSC / 2) + 4 * h;
    const int2* tb = reinterpret_cast<const int2*>(Tb + (((c - c_lo) * 2 + kh) * 2 + h) * 32);
    auto fetch = [&](int2 o, int kk, FwdOps& d) {
      d.sfv = *reinterpret_cast<const float*>(frow + o.x);
      d.af4 = *reinterpret_cast<const float4*>(frow + o.y);
      d.b4 = *reinterpret_cast<const float4*>(wrow + kk);
    };
    FwdOps cur;
    fetch(tb[kk_lo / 8], kk_lo, cur);
    int2 o_nxt = tb[kk_lo / 8 + 1];
#pragma unroll
    for (int kk = kk_lo; kk < kk_hi; kk += 8) {
      const float a0_ = cur.sfv * cur.af4.x, a1_ = cur.sfv * cur.af4.y;
      const float a2_ = cur.sfv * cur.af4.z, a3_ = cur.sfv * cur.af4.w;
      const float4 b = cur.b4;
      if (kk + 8 < kk_hi) {
        fetch(o_nxt, kk + 8, cur);
        if (kk + 16 < kk_hi) o_nxt = tb[kk / 8 + 2];
      }
      facc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_, b.x, facc, 0, 0, 0);
      facc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_, b.y, facc, 0, 0, 0);
      facc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2_, b.z, facc, 0, 0, 0);
      facc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3_, b.w, facc, 0, 0, 0);
      const int step = kk / 8;
      if (c_st >= 0 && step >= 1 && step <= 4 && step - 1 >= q_from) store_quad(c_st, step - 1);
      if (c_ld >= 0 && step >= 6 && step <= 9 && step - 6 >= q_from) load_quad(c_ld, step - 6, moments);
      // issue order of the step: the first product, MFMA 1, then -- in the 64 cycles the dependent
      // MFMA 2 waits anyway -- the other products, the next step's addresses and its LDS reads
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  constexpr int kHalf = kSC / 4;             // a wavefront's 128 columns in two halves of 64

  // ---- partial products of this workgroup -> slab; the sum over the workgroups of the block ----
  auto stamp_sum = [&](unsigned epoch, int slot) {
    if (p.prof && tid == 0 && epoch >= (unsigned)step0 + 2u && epoch - (unsigned)step0 - 2u < kMProfUpdates)
      p.prof[((int64_t)blockIdx.x * kMProfUpdates + (epoch - step0 - 2u)) * 16 + slot] = wall_clock64();
  };
  // (pub / done: the flag arrays of the hand-off -- the update's or an evaluation pass's --, dst: where
  // the summed [B][128] slab goes)
  auto publish_and_sum = [&](unsigned epoch, unsigned* pub, unsigned* done, float* dst, bool stamps) {
    {
      // the two k-halves meet in LDS; the [B][32] block goes out as 16-byte write-through stores
      float* X = Wc;                                       // [4][32][33]
      float* stage = Ft;                                   // [B][kSBP]
      if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) X[(mt * 32 + acc_row(i, h)) * kMPbuf + l31] = facc[i];
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = mt * 32 + acc_row(i, h);
          if (row < B) stage[row * kSBP + l31] = facc[i] + X[row * kMPbuf + l31];
        }
      }
      __syncthreads();
      const __amdgpu_buffer_rsrc_t dr = xwg_buffer(p.slabs + (int64_t)g * B * kMNB);
      for (int idx = tid; idx < B * (kMNB / 4); idx += kMT) {
        const float4 v = *reinterpret_cast<const float4*>(stage + (idx >> 3) * kSBP + (idx & 7) * 4);
        xwg_store4(dr, idx * 4, v.x, v.y, v.z, v.w);
      }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) flag_raise(pub, g, epoch);
    if (stamps) stamp_sum(epoch, 14);
    if (w == 0) flags_wait(pub, T, epoch, lane, flagp);
    __syncthreads();
    if (stamps) stamp_sum(epoch, 15);
    // my share of the B*8 quads of the block's [B, 32] columns; 32 groups of slabs per quad, partial
    // sums combined in group order
    const int nq = B * (kMNB / 4);
    const int q_lo = (int)((int64_t)gq * nq / G), q_hi = (int)((int64_t)(gq + 1) * nq / G);
    const __amdgpu_buffer_rsrc_t sr = xwg_buffer(p.slabs);
    const int zs = 4 * B * kMNB;                           // from one workgroup of the block to the next
    float* part = Wc;
    for (int qb = q_lo; qb < q_hi; qb += 16) {
      const int qi = tid & 15, sg = tid >> 4;
      const int q = min(qb + qi, q_hi - 1);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      for (int z = sg; z < G; z += kSRedGroups * 2) {
        f32x4 ld[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
          ld[u] = xwg_load4(sr, min(z + u * kSRedGroups, G - 1) * zs + nb * B * kMNB + q * 4);
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (z + u * kSRedGroups < G) v += ld[u];
      }
      *reinterpret_cast<f32x4*>(part + (sg * 16 + qi) * 4) = v;
      __syncthreads();
      if (tid < 16 && qb + tid < q_hi) {
        f32x4 s = *reinterpret_cast<const f32x4*>(part + tid * 4);
        for (int u = 1; u < kSRedGroups; ++u) s += *reinterpret_cast<const f32x4*>(part + (u * 16 + tid) * 4);
        const int qd = qb + tid, b = qd >> 3, n = (qd & 7) * 4;
        xwg_store4(xwg_buffer(dst), b * kMH + nb * kMNB + n, s.x + b1s[n], s.y + b1s[n + 1],
                   s.z + b1s[n + 2], s.w + b1s[n + 3]);
      }
      __syncthreads();
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) flag_raise(done, g, epoch);
  };

  // ---- where a forward step finds its factors (see forward_chunk) ---------------------------------
  for (int idx = tid; idx < (c_hi - c_lo) * 64; idx += kMT) {
    const int st = idx & 15, hh = (idx >> 4) & 1, khh = (idx >> 5) & 1, ci = idx >> 6;
    const FwdPos q = forward_pos((c_lo + ci) * kSC + khh * (kSC / 2) + st * 8 + 4 * hh);
    Tb[idx * 2 + 0] = (q.i - i_lo) * 4;
    Tb[idx * 2 + 1] = (NIP + (q.i >= S ? A + min(q.ai, 4) : q.ai)) * 4;
  }
  __syncthreads();

  // ---- held-out evaluation number eidx (mdnn.py:235-242), the tile workgroups' part: a forward-only
  //      pass per 100 held-out pairs over the weights in memory (those of the update just taken: the
  //      pass runs while the row owners work on the next update's rows, the window in which the tile
  //      workgroups otherwise wait), from the held-out pairs' factor rows; the summed slabs go to
  //      eval_slabs[eidx & 1][pass], the owners evaluate from there (owner_eval).  The update's slabs
  //      are reused: `slabs_free_at` is the flag_red tag of their last readers (0: none left).
  //      Clobbers Fr, Ft, Wc, the chunk registers and the forward accumulators.
  auto eval_all = [&](int eidx, unsigned slabs_free_at) {
    for (int pass = 0; pass < p.eval_passes; ++pass) {
      const int rows = min(B, p.n_test - pass * B);
      if (rows <= 0) break;
      const unsigned tag = (unsigned)eidx * 16u + (unsigned)pass + 1u;
      if (w == 0) {
        if (pass == 0) flags_wait(p.flag_red, T, slabs_free_at, lane, flagp);
        else flags_wait(p.flag_evr, T, tag - 1u, lane, flagp);
      }
      __syncthreads();
      load_factor_rows((int64_t)pass * B, true, rows);
#pragma unroll
      for (int i = 0; i < 16; ++i) facc[i] = 0.f;
      if (c_lo < c_hi) load_chunk(c_lo, false);
      __syncthreads();
      for (int c = c_lo; c < c_hi; ++c) {
        weights_to_lds();
        lds_barrier();
        forward_chunk(c, 0, 2 * kHalf, -1, 0, c + 1 < c_hi ? c + 1 : -1, false);
        lds_barrier();
      }
      publish_and_sum(tag, p.flag_evp, p.flag_evr,
                      p.eval_slabs + ((int64_t)(eidx & 1) * p.eval_passes + pass) * B * kMH, false);
    }
    if (tid == 0) flag_raise(p.flag_eval, g, (unsigned)eidx + 1u);
    // (nobody overwrites its slab -- the next update's product -- before every workgroup has read it)
    const int last = min(p.eval_passes, (p.n_test + B - 1) / B) - 1;
    if (w == 0 && last >= 0) flags_wait(p.flag_evr, T, (unsigned)eidx * 16u + (unsigned)last + 1u, lane, flagp);
    __syncthreads();
  };

  // ---- prologue: the forward product of the launch's first minibatch --------------------------
  if (p.n_updates > 0) {
    load_factor_rows((int64_t)step0 * B);
#pragma unroll
    for (int i = 0; i < 16; ++i) facc[i] = 0.f;
    if (c_lo < c_hi) load_chunk(c_lo, false);
    __syncthreads();
    for (int c = c_lo; c < c_hi; ++c) {
      weights_to_lds();
      lds_barrier();
      forward_chunk(c, 0, 2 * kHalf, -1, 0, c + 1 < c_hi ? c + 1 : -1, false);
      lds_barrier();
    }
    publish_and_sum((unsigned)step0 + 1u, p.flag_fwd, p.flag_red, p.hpre, true);
  }

  // (one more trip than updates when the call's last evaluation is due: the evaluation passes have
  // ONE call site -- inlined twice they cost the chunk loop registers: spills inside the pass)
  const bool eval_last = !DP && p.do_eval && step0 + p.n_updates == p.n_total;
  for (int t = 0; t < p.n_updates + (eval_last ? 1 : 0); ++t) {
    const int step = step0 + t;
    const unsigned epoch = (unsigned)step + 1u;
    const bool tail = t == p.n_updates;                    // only the evaluation after the last update
    const bool has_next = !DP && t + 1 < p.n_updates;
    const bool moments = !(fresh && t == 0);
    relaunder();
    if (run_aborted(flagp, red, tid)) break;
    // the evaluation due after the previous update, then this minibatch's factor rows again
    if (!DP && __builtin_expect(p.do_eval && step > 0 && ((step - 1) % p.eval_every == 0 || tail), 0)) {
      eval_all(tail ? mdnn_evals_before(p.n_total - 1, p.eval_every) : mdnn_evals_before(step, p.eval_every) - 1,
               tail ? 0u : epoch);
      if (tail) break;
      relaunder();
      load_factor_rows((int64_t)step * B);
      __syncthreads();
    }
    BSIG_MSTAMP(0);
    // ---- while the owners work: the first chunk (the long fetch first), this minibatch's factor
    //      columns (dW), the next minibatch's factor rows (forward), Adam scalars
    if (!DP && c_lo < c_hi) load_chunk(c_lo, moments);
    transpose_factors();
    __syncthreads();
    BSIG_MSTAMP(1);
    if (has_next) load_factor_rows((int64_t)(step + 1) * B);
    BSIG_MSTAMP(6);
    b1t *= p.beta1; b2t *= p.beta2;
    a0 = (float)(p.lr / (1.0 - b1t));
    a1 = (float)(1.0 / sqrt(1.0 - b2t));
    if (w == 0) flags_wait(p.flag_own, p.n_owner, epoch, lane, flagp);
    __syncthreads();
    BSIG_MSTAMP(2);
    // ---- dz1[:, block] -> the A operand registers (the same in every wavefront) ---------------
    float za[52];
    {
      const __amdgpu_buffer_rsrc_t zr = xwg_buffer(p.dz1);
      float* Zs = Wc;                                      // [FR][kSBP]
      f32x4 q[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = u * kMT + tid;
        q[u] = xwg_load4(zr, min(idx >> 3, B - 1) * kMH + nb * kMNB + (idx & 7) * 4);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int idx = u * kMT + tid, b = idx >> 3;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        if (b < FR) *reinterpret_cast<f32x4*>(Zs + b * kSBP + (idx & 7) * 4) = b < B ? q[u] : zero;
      }
      __syncthreads();
#pragma unroll
      for (int gq8 = 0; gq8 < 13; ++gq8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int b = 8 * gq8 + 4 * h + e;
          za[4 * gq8 + e] = b < FR ? Zs[b * kSBP + l31] : 0.f;
        }
      }
      __syncthreads();
    }
    // ---- b1: column sums of dz1 (fixed order: lane half 0 then 1), Adam -------------------------
    {
      float gs = 0.f;
#pragma unroll
      for (int q = 0; q < 52; ++q) gs += za[q];
      gs += __shfl_xor(gs, 32, 64);
      if (bias_lane) {
        if (DP) { if (gq == 0) p.grads[p.b1_off + nb * kMNB + tid] = gs; }
        else bw = adam_bias(gs, bm, bv, bw, a0, a1, ak);
        b1s[tid] = bw;                                     // (the previous sum's readers are long done)
      }
    }
    BSIG_MSTAMP(3);

    // ---- the pass --------------------------------------------------------------------------
    // dW tile of chunk c = dz1^T X_t[:, 32 columns]: lane <-> column k = i*A + j, its two factors
    // down the minibatch rows; 13 groups of 4 MFMAs
    const float* sfp = Ft;
    const float* afp = Ft;
    auto dw_setup = [&](int c) {
      const int k = c * kSC + w * 32 + l31;
      int si, ai;
      if (k >= SA) { si = S - i_lo; ai = A + min(k - SA, 2); }
      else { const int i = div_small(k, A, rA); si = i - i_lo; ai = k - i * A; }
      sfp = Ft + si * kSTP + 4 * h;
      afp = Ft + (NIP + ai) * kSTP + 4 * h;
    };
    f32x16 acc;
    // (operands of group gq8 + 1 are fetched while the MFMAs of group gq8 run; the scheduling
    // barrier keeps the compiler from hoisting all 26 fetches of a tile to its top -- 104 registers)
    float4 dw_s, dw_f;
    auto dw_fetch = [&](int gq8) {
      dw_s = *reinterpret_cast<const float4*>(sfp + 8 * gq8);
      dw_f = *reinterpret_cast<const float4*>(afp + 8 * gq8);
    };
    // (always the 13 groups of FR <= 104 rows: rows past the minibatch are zeros in za and in Ft; a
    // branch per group on FR cost more -- its masks spilled out of the SGPRs -- than the MFMAs it saved)
    // A dependent MFMA waits ~64 cycles at the issue port for its predecessor, and nothing else of
    // the SIMD issues meanwhile -- except what sits between the two in the SAME instruction stream.
    // So the group's issue order is: first product, MFMA 1, then (in the wait of MFMA 2) the other
    // products and the next group's LDS reads.
    auto dw_group = [&](int gq8) {
      const float x0 = dw_s.x * dw_f.x, x1 = dw_s.y * dw_f.y, x2 = dw_s.z * dw_f.z, x3 = dw_s.w * dw_f.w;
      if (gq8 + 1 < 13) dw_fetch(gq8 + 1);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq8 + 0], x0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq8 + 1], x1, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq8 + 2], x2, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq8 + 3], x3, acc, 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // VALU
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_barrier(0);
    };
    // The same group carrying the Adam step of quad u of chunk c (whole chunks c, c + 1): one weight
    // (10 VALU instructions, two of them quarter rate: ~64 cycles) in the wait of each MFMA, the quad's
    // stores and the reloads behind the last one.  Pinned by scheduling barriers: left to the
    // scheduler (sched_group_barrier) the spread order exceeds the register budget and is reverted.
    auto dw_group_adam = [&](int gq8, int c, int u) {
      const float x0 = dw_s.x * dw_f.x, x1 = dw_s.y * dw_f.y, x2 = dw_s.z * dw_f.z, x3 = dw_s.w * dw_f.w;
      if (gq8 + 1 < 13) dw_fetch(gq8 + 1);
      float* gp = Wc + (rl + 8 * u) * kSPitch + cl;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gp);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq8 + 0], x0, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      W4[u][0] = adam_weight(g4.x, M4[u][0], V4[u][0], W4[u][0], a0, a1, ak);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq8 + 1], x1, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      W4[u][1] = adam_weight(g4.y, M4[u][1], V4[u][1], W4[u][1], a0, a1, ak);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq8 + 2], x2, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      W4[u][2] = adam_weight(g4.z, M4[u][2], V4[u][2], W4[u][2], a0, a1, ak);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(za[4 * gq8 + 3], x3, acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      W4[u][3] = adam_weight(g4.w, M4[u][3], V4[u][3], W4[u][3], a0, a1, ak);
      const f32x4 t4 = {W4[u][0], W4[u][1], W4[u][2], W4[u][3]};
      *reinterpret_cast<f32x4*>(gp) = t4;                  // (same thread, same spot)
      store_quad_full(c, u);
      load_quad_full(c + 1, u);
      __builtin_amdgcn_sched_barrier(0);
    };
    auto dw_to_lds = [&]() {
#pragma unroll
      for (int i = 0; i < 16; ++i) Wc[acc_row(i, h) * kSPitch + w * 32 + l31] = acc[i];
    };
    // Adam on quad u of the chunk in W4 / M4 / V4: its gradient sits in Wc, the new weights replace it
    auto adam_quad = [&](int u) {
      float* gp = Wc + (rl + 8 * u) * kSPitch + cl;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(gp);
      W4[u][0] = adam_weight(g4.x, M4[u][0], V4[u][0], W4[u][0], a0, a1, ak);
      W4[u][1] = adam_weight(g4.y, M4[u][1], V4[u][1], W4[u][1], a0, a1, ak);
      W4[u][2] = adam_weight(g4.z, M4[u][2], V4[u][2], W4[u][2], a0, a1, ak);
      W4[u][3] = adam_weight(g4.w, M4[u][3], V4[u][3], W4[u][3], a0, a1, ak);
      const f32x4 t4 = {W4[u][0], W4[u][1], W4[u][2], W4[u][3]};
      *reinterpret_cast<f32x4*>(gp) = t4;                  // (same thread, same spot)
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) facc[i] = 0.f;
    if (DP) {
      // this rank's share of the gradient, summed over the ranks by the caller
      for (int c = c_lo; c < c_hi; ++c) {
        const int k = c * kSC + w * 32 + l31;
        dw_setup(c);
        dw_fetch(0);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int gq8 = 0; gq8 < 13; ++gq8) dw_group(gq8);
        if (k < I) {
#pragma unroll
          for (int i = 0; i < 16; ++i)
            p.grads[p.w1_off + (int64_t)(nb * kMNB + acc_row(i, h)) * I + k] = acc[i];
        }
      }
    } else if (c_lo < c_hi) {
      // The Adam arithmetic of chunk c (VALU) runs in the shadow of the MFMAs of chunk c + 1's
      // gradient: chunk c's gradient is already in LDS when its Adam step starts.
      dw_setup(c_lo);
      dw_fetch(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
      for (int gq8 = 0; gq8 < 13; ++gq8) dw_group(gq8);
      dw_to_lds();
      lds_wave_sync();
      for (int c = c_lo; c < c_hi; ++c) {
        const bool more = c + 1 < c_hi;
        relaunder();                                       // (nothing lane-derived stays live across chunks)
        if (c == c_lo + 1) BSIG_MSTAMP(8);
        BSIG_WSTAMP(0);
        if (more) {
          dw_setup(c + 1);
          dw_fetch(0);
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[i] = 0.f;
          // (chunks c and c + 1 whole, moments live: no branch inside a group -- one scheduling region)
          if (__builtin_expect((c + 2) * kSC <= I && moments, 1)) {
#pragma unroll
            for (int gq8 = 0; gq8 < 13; ++gq8) {
              if (gq8 % 3 == 1) dw_group_adam(gq8, c, gq8 / 3);   // groups 1, 4, 7, 10
              else dw_group(gq8);
            }
          } else {
#pragma unroll
            for (int gq8 = 0; gq8 < 13; ++gq8) {
              dw_group(gq8);
              if (gq8 % 3 == 1) {
                adam_quad(gq8 / 3);
                store_quad(c, gq8 / 3);
                load_quad(c + 1, gq8 / 3, moments);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u) adam_quad(u);
          if (!has_next) store_chunk(c);
        }
        if (c == c_lo + 1) { asm volatile("" :: "v"(acc[0])); BSIG_MSTAMP(9); }
        asm volatile("" :: "v"(acc[0])); BSIG_WSTAMP(1);
        // (the LAST chunk of a pass that is followed by a forward sum keeps its block in registers
        // until the sum is out: these stores would otherwise sit in front of the slab stores and
        // their flag on the chip's write path)
        const bool defer = has_next && !more;
        if (c == c_lo + 1) BSIG_MSTAMP(10);
        if (has_next) {
          lds_barrier();                                   // every block of the chunk holds its new weights
          if (c == c_lo + 1) BSIG_MSTAMP(11);
          BSIG_WSTAMP(2); BSIG_WCLK(10);
          // the block's 12 stores and the next chunk's 12 loads go out between the quarters of the
          // forward product: their issue (1 KB per instruction through the CU's 64 B/clk path) runs
          // in the shadow of the MFMAs instead of in front of a barrier
          forward_chunk(c, 0, 2 * kHalf, -1, 4, -1, moments);
          asm volatile("" :: "v"(facc[0])); BSIG_WSTAMP(7); BSIG_WCLK(11);
          if (c == c_lo + 1) { asm volatile("" :: "v"(facc[0])); BSIG_MSTAMP(12); }
          lds_barrier();                                   // the chunk's weights are read: Wc is free
          BSIG_WSTAMP(8);
        } else if (more) {
          load_chunk(c + 1, moments);
        }
        if (more) {
          dw_to_lds();
          lds_wave_sync();
        }
        BSIG_WSTAMP(9);
        if (c == c_lo + 1) BSIG_MSTAMP(13);
        if (c == c_lo) BSIG_MSTAMP(7);
      }
    }
    BSIG_MSTAMP(4);
    if (has_next) {
      publish_and_sum(epoch + 1u, p.flag_fwd, p.flag_red, p.hpre, true);
      if (c_lo < c_hi) store_chunk(c_hi - 1);              // the deferred block of the last chunk
    }
    BSIG_MSTAMP(5);
  }

  // ---- write b1 back, advance the engine state -------------------------------------------------
  if (!DP && gq == 0 && bias_lane) {
    const int64_t off = p.b1_off + nb * kMNB + tid;
    p.params[off] = bw; p.m1[off] = bm; p.m2[off] = bv;
  }
  if (g == 0 && tid == 0 && p.n_updates > 0) {
    int32_t* st = p.state;
    reinterpret_cast<double*>(st + 12)[0] = b1t;
    reinterpret_cast<double*>(st + 12)[1] = b2t;
    reinterpret_cast<float*>(st)[4] = a0;
    reinterpret_cast<float*>(st)[5] = a1;
    // (one jitter stream per update and per evaluation, in program order)
    int n_ev = 0;
    if (!DP && p.do_eval) {
      n_ev = mdnn_evals_before(step0 + p.n_updates, p.eval_every) - mdnn_evals_before(step0, p.eval_every);
      if (step0 + p.n_updates == p.n_total && (p.n_total - 1) % p.eval_every != 0) ++n_ev;
    }
    reinterpret_cast<uint64_t*>(st + 8)[1] += (uint64_t)(p.n_updates + n_ev);
    st[0] = step0 + p.n_updates;
  }
}

// DP: data-parallel rank; WIDE / FULL: as in the resident kernel
template <bool DP, bool WIDE, bool FULL>
__global__ __launch_bounds__(kMT) void mdnn_stream_updates_kernel(MdnnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
#ifndef BSIG_HOST_SAN_BUILD
  const int wg = blockIdx.x;
  if (wg < p.G1) mdnn_stream_tile_workgroup<DP>(p, smem);
#ifndef BSIG_STREAM_TILE_ONLY   // (resource usage of the tile body alone)
  else if (wg < p.G1 + p.n_owner) mdnn_owner_workgroup<DP, WIDE, FULL, kStreamMR>(p, smem);
  else mdnn_small_workgroup<DP, WIDE>(p, smem);
#endif
#endif
}

// ---------------------------------------------------------------- host side
// LDS floats of a tile workgroup; false when the factor rows of two minibatches do not fit
bool mdnn_stream_tile_geom(int FR, int chunks_per_wg, int S, int A, int* nip, int* pf, size_t* lds_bytes) {
  if (S < 1 || A < 8 || A % 4 != 0) return false;
  const int cols = chunks_per_wg * kSC;
  const int ni = (cols - 1) / A + 2;            // distinct i = k / A over `cols` columns (i = S: the tail)
  *nip = (int)round_up(ni, 4);
  int pitch = *nip + A + 8;
  while (pitch % 8 != 4) pitch += 4;            // 16-byte reads down the rows: conflict-free
  *pf = pitch;
  const size_t floats = (size_t)FR * pitch + (size_t)(*nip + A + 8) * kSTP + (size_t)kMNB * kSPitch + 64 + kMNB +
                        (size_t)chunks_per_wg * 128;   // (+ the forward steps' offset table)
  *lds_bytes = floats * sizeof(float);
  // (the chunk region also stages dz1 [FR][36], the k-half exchange [128][33] and the partial
  // sums [32][16][4]; the factor columns the [B][36] block of partial products; a factor row is
  // loaded as at most three 64-column groups)
  return *lds_bytes <= (size_t)kMLdsLimit && FR <= 104 && *nip + A + 8 <= 192 &&
         (size_t)kMNB * kSPitch >= (size_t)128 * kMPbuf && (size_t)kMNB * kSPitch >= (size_t)FR * kSBP &&
         (size_t)(*nip + A + 8) * kSTP >= (size_t)FR * kSBP;
}

// chunks of 256 columns; the tile workgroups come in blocks of four (one per 32 hidden units)
int mdnn_stream_chunks(int input_dim) { return ceil_div(input_dim, kSC); }

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
  // Every workgroup of the launch must be resident at once: the runtime's occupancy answer for THE
  // instantiation that is launched, at the LDS size it is launched with (known only now: it depends
  // on the factor dimensions), must admit a workgroup per CU.  Asked once per (device, variant, size).
  {
    struct Seen { int dev, variant; size_t lds; bool ok; };
    static thread_local Seen seen[8];
    static thread_local int n_seen = 0;
    const int variant = (dp ? 1 : 0) | (wide ? 2 : 0) | (full ? 4 : 0);
    bool known = false, ok = false;
    for (int i = 0; i < n_seen; ++i)
      if (seen[i].dev == dev && seen[i].variant == variant && seen[i].lds == lds) { known = true; ok = seen[i].ok; }
    if (!known) {
      int per_cu = 0;
      hipError_t e = hipErrorUnknown;
#define BSIG_O(a, b, c) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mdnn_stream_updates_kernel<a, b, c>, kMT, lds)
      switch (variant) {
        case 0: BSIG_O(false, false, false); break; case 1: BSIG_O(true, false, false); break;
        case 2: BSIG_O(false, true, false); break;  case 3: BSIG_O(true, true, false); break;
        case 4: BSIG_O(false, false, true); break;  case 5: BSIG_O(true, false, true); break;
        case 6: BSIG_O(false, true, true); break;   default: BSIG_O(true, true, true); break;
      }
#undef BSIG_O
      hipDeviceProp_t prop;
      ok = e == hipSuccess && per_cu >= 1 && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
           (size_t)prop.maxSharedMemoryPerMultiProcessor >= lds && prop.multiProcessorCount >= grid;
      seen[n_seen % 8] = Seen{dev, variant, lds, ok};
      n_seen = std::min(n_seen + 1, 8);
    }
    if (!ok) {
      set_error("mdnn_stream_updates: the device cannot hold the launch's %d workgroups (%zu bytes of LDS each) at once",
                grid, lds);
      return BSIG_EUNSUPPORTED;
    }
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
