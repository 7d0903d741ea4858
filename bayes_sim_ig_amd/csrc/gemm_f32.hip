// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// fma chain at the 157 TFLOP/s vector rate; gfx950 has no xf32/TF32 path) with
// fused epilogues.  Serves every dense contraction of the estimator:
//   RFF projection  x @ (freqs/sigma)^T -> a*[cos|sin]     rff.py:128-132
//   nn.Linear forward / backward of trunk and heads        mdnn.py:68-86,108-119
// C[m,n] = epi( sum_k A(m,k) * B(n,k) ); each operand is either k-contiguous
// (rows of a [rows, K] matrix, optionally gathered by an index vector — the
// minibatch gather of mdnn.py:222 fused into the loader) or k-major
// (the contraction index is the slow dimension: the transposed operands of
// the backward products dW = dY^T X and dX = dY W).
//
// Tiling: BK = 32; block = WM x WN waves, each wave owns TM x TN 32x32 MFMA
// tiles.  LDS images: k-contiguous operands as [rows][BK+4] (ds_read_b128 of
// 4 consecutive k per lane, conflict-free at pitch 36), k-major operands as
// [BK][rows] (ds_read_b32, 32 consecutive rows per half-wave).  Lane l of an
// MFMA supplies row (l & 31) and k-slot (l >> 5); within an 8-wide k group the
// half h = l>>5 takes k = 4h..4h+3, one per MFMA step — the k order is a
// permutation of 0..7, identical for A and B.
// Split-K (gridDim.z) writes fp32 partial slabs and a second kernel reduces
// them in a fixed order and applies the epilogue: bitwise reproducible.
#include <cstdlib>
#include "gemm_kernel.h"
#include "gemm_lean.h"
#include "gemm_wide.h"

namespace bsig {

__global__ __launch_bounds__(256) void gemm_reduce_kernel(GemmParams p) {
  const int64_t total = (int64_t)p.m * p.n;
  const int64_t slab = p.partial_ld ? p.partial_slab : total;
  float exp_acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(e / p.n), col = (int)(e % p.n);
    const int64_t pe = p.partial_ld ? (int64_t)row * p.partial_ld + col : e;   // (pitched slabs: gemm_wide.h)
    float v = 0.f;
    int z = 0;
    for (; z + 16 <= p.splits; z += 16) {   // 16 slab loads in flight, summed in slab order
      float q[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) q[u] = p.partial[(int64_t)(z + u) * slab + pe];
#pragma unroll
      for (int u = 0; u < 16; ++u) v += q[u];
    }
    if (z + 8 <= p.splits) {
      float q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = p.partial[(int64_t)(z + u) * slab + pe];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += q[u];
      z += 8;
    }
    for (; z < p.splits; ++z) v += p.partial[(int64_t)z * slab + pe];
    epilogue_store(p, row, col, v);
    if (p.expsum && p.epilogue == BSIG_EPI_BIAS && col >= p.expsum_col0 &&
        col < p.expsum_col0 + p.expsum_ncols)
      exp_acc += expf(v + p.bias[col]);
  }
  if (p.expsum) {
    __shared__ float red[8];
    const float s = block_sum(exp_acc, red);
    if (threadIdx.x == 0) p.expsum[blockIdx.x] = s;
  }
}

// The same reduction for a fused Adam step (EPI_ADAM: dW slabs -> one optimizer step on the weights),
// four adjacent columns per thread: 16-byte loads of the slabs and of the optimizer state, all in
// flight before the first dependent instruction.  Same arithmetic per element as epilogue_store's
// EPI_ADAM branch (slab order, then the identical expressions): bit-identical to the scalar kernel.
template <bool ADAM>   // (false: BSIG_EPI_NONE, the plain sum -- a data-parallel rank's gradients)
__global__ __launch_bounds__(256) void gemm_reduce_adam4_kernel(GemmParams p) {
  const int n4 = p.n >> 2;
  const int64_t total4 = (int64_t)p.m * n4;
  const int64_t slab = p.partial_ld ? p.partial_slab : (int64_t)p.m * p.n;
  float ss = 0.f, ib = 0.f;
  if constexpr (ADAM) { ss = p.adam_dyn[0]; ib = p.adam_dyn[1]; }
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total4;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(e / n4), col = (int)(e % n4) * 4;
    const int64_t pe = p.partial_ld ? (int64_t)row * p.partial_ld + col : (int64_t)row * p.n + col;
    const int64_t ce = (int64_t)row * p.ldc + col;
    F4 g{{0.f, 0.f, 0.f, 0.f}};
    F4 pm{{0.f, 0.f, 0.f, 0.f}}, pv = pm, pp = pm;
    if constexpr (ADAM) { pm = ld4(p.adam_m + ce); pv = ld4(p.adam_v + ce); pp = ld4(p.c + ce); }
    int z = 0;
    for (; z + 4 <= p.splits; z += 4) {
      F4 q[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) q[u] = ld4(p.partial + (int64_t)(z + u) * slab + pe);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) g.v[c] += q[u].v[c];
    }
    for (; z < p.splits; ++z) {
      const F4 q = ld4(p.partial + (int64_t)z * slab + pe);
#pragma unroll
      for (int c = 0; c < 4; ++c) g.v[c] += q.v[c];
    }
    if constexpr (!ADAM) { st4(p.c + ce, g); continue; }
    if (p.grad_out) st4(p.grad_out + ce, g);
    F4 m1, v1, p1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      m1.v[c] = pm.v[c] + (g.v[c] - pm.v[c]) * (1.0f - p.beta1);
      v1.v[c] = pv.v[c] * p.beta2 + (1.0f - p.beta2) * g.v[c] * g.v[c];
      p1.v[c] = pp.v[c] - ss * (m1.v[c] / (sqrtf(v1.v[c]) * ib + p.adam_eps));
    }
    st4(p.adam_m + ce, m1); st4(p.adam_v + ce, v1); st4(p.c + ce, p1);
    if (col == 0 && p.bias_p) {
      // (partial sums in index order, 16 loads in flight: one lane of a wavefront does this while the
      // others wait -- a load per iteration made it 64 dependent round trips)
      float bg = 0.f;
      for (int q0 = 0; q0 < max(p.bias_g_n, 1); q0 += 16) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          t[u] = p.bias_g[(int64_t)min(q0 + u, max(p.bias_g_n, 1) - 1) * p.bias_g_stride + row];
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (q0 + u < max(p.bias_g_n, 1)) bg += t[u];
      }
      const float bm = p.bias_m[row] + (bg - p.bias_m[row]) * (1.0f - p.beta1);
      const float bv = p.bias_v[row] * p.beta2 + (1.0f - p.beta2) * bg * bg;
      p.bias_m[row] = bm;
      p.bias_v[row] = bv;
      p.bias_p[row] = p.bias_p[row] - ss * (bm / (sqrtf(bv) * ib + p.adam_eps));
    }
  }
}

// (the reduce of a split product: the 16-byte kernel where it applies)
static int launch_reduce(const GemmParams& p, hipStream_t st, int* n_expsum) {
  const int64_t total = (int64_t)p.m * p.n;
  auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  const bool quads = (p.n & 3) == 0 && (p.ldc & 3) == 0 && (p.partial_ld & 3) == 0 && (p.partial_slab & 3) == 0 &&
                     al(p.partial) && al(p.c) && getenv("BSIG_GEMM_NO_ADAM4") == nullptr;
  const bool adam4 = quads && p.epilogue == EPI_ADAM && al(p.adam_m) && al(p.adam_v) && (!p.grad_out || al(p.grad_out));
  BSIG_REQUIRE(!(p.bias_g_n > 1 && !adam4), "gemm: partial bias sums need the 16-byte reduce + Adam kernel");
  if (adam4 || (quads && p.epilogue == BSIG_EPI_NONE && total >= (1 << 18))) {
    const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>(total / 4, 256), 2048);
    if (adam4) hipLaunchKernelGGL(gemm_reduce_adam4_kernel<true>, dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(gemm_reduce_adam4_kernel<false>, dim3(blocks), dim3(256), 0, st, p);
    BSIG_CHECK_LAUNCH("gemm_reduce_adam4");
    return BSIG_OK;
  }
  const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>(total, 256), 2048);
  hipLaunchKernelGGL(gemm_reduce_kernel, dim3(blocks), dim3(256), 0, st, p);
  BSIG_CHECK_LAUNCH("gemm_reduce");
  if (n_expsum && p.expsum) *n_expsum = blocks;
  return BSIG_OK;
}

// Debug instantiation (BSIG_DEBUG_F64_ACC_MIN_K=<k>): a product whose contraction is at least <k>
// long is formed by one thread per output element with the products and their sum in fp64, rounded
// to fp32 once -- the value every fp32 summation order of the same operands approximates.  Same
// operand addressing (gathers, k-major operands, device-resolved row offsets), same epilogues and
// side outputs as the MFMA kernel; 100x slower.  SURVEY.md section 7 ("hard parts") asks for it to
// extend the horizon of the ill-conditioned wide first layers (cfg/anymal.yaml: 56 402 terms per
// output, cfg/shadow_hand_more.yaml: 105 002): with it the per-phase path's chunk stays with the
// fp64 chunk where fp32 accumulation orders leave each other
// (tests/test_gpu_fit.py::test_wide_chunk_with_fp64_first_layer_sums_stays_with_fp64).
__global__ __launch_bounds__(256) void gemm_f64acc_kernel(GemmParams p) {
  const int64_t total = (int64_t)p.m * p.n;
  const int64_t dstep = p.dyn ? (int64_t)(p.dyn[0] + p.dyn_delta) : 0;
  const int64_t a_off = dstep * p.a_dyn_stride + p.a_dyn_base, b_off = dstep * p.b_dyn_stride + p.b_dyn_base;
  float exp_acc = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(e / p.n), col = (int)(e % p.n);
    // k-contiguous operand: the gathered dimension is its row; k-major: its contraction index
    const int64_t ar = p.a_kmajor ? 0 : (p.a_rows ? p.a_rows[row + a_off] : row + a_off);
    const int64_t br = p.b_kmajor ? 0 : (p.b_rows ? p.b_rows[col + b_off] : col + b_off);
    double acc = 0.0;
    for (int kk = 0; kk < p.k; ++kk) {
      const float av = p.a_kmajor ? p.a[(p.a_rows ? (int64_t)p.a_rows[kk + a_off] : kk + a_off) * p.lda + row]
                                  : p.a[ar * p.lda + kk];
      const float bv = p.b_kmajor ? p.b[(p.b_rows ? (int64_t)p.b_rows[kk + b_off] : kk + b_off) * p.ldb + col]
                                  : p.b[br * p.ldb + kk];
      acc += (double)av * (double)bv;
    }
    const float v = (float)acc;
    epilogue_store(p, row, col, v);
    if (p.expsum && p.epilogue == BSIG_EPI_BIAS && col >= p.expsum_col0 &&
        col < p.expsum_col0 + p.expsum_ncols)
      exp_acc += expf(v + p.bias[col]);
  }
  if (p.expsum) {
    __shared__ float red[8];
    const float s = block_sum(exp_acc, red);
    if (threadIdx.x == 0) p.expsum[blockIdx.x] = s;
  }
}

enum { TILE_64 = 0, TILE_128 = 1, TILE_128x32 = 2, TILE_128x64 = 3, TILE_128x96 = 4,
       TILE_96x128 = 5, TILE_128x288 = 6, TILE_288x128 = 7, N_TILES = 8 };
static const int kTileM[N_TILES] = {64, 128, 128, 128, 128, 96, 128, 288};
static const int kTileN[N_TILES] = {64, 128, 32, 64, 96, 128, 288, 128};
struct GemmPlan { int tile; int splits; int k_chunk; };

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// Tile / split-K choice.  Large problems: 128x128 tiles, no split.  Minibatch
// sized M (<= 128 rows): 128x32 tiles so the big operand (weights / RFF
// coefficients) is streamed exactly once, split-K until ~4 workgroups per CU.
// Large-minibatch shapes (the scaled-batch fit: thousands of rows per update, or a
// contraction over thousands of rows): pick the tile with the least padded area -- a
// head of 260 outputs on 128-wide tiles wastes a third of every block -- and split K
// until there are ~3 workgroups per CU: one 256-thread workgroup per CU hides neither
// the global nor the LDS latency (8192 x 260 x 4096 ran at 34 TFLOP/s on 192 unsplit
// 128 x 128 tiles).
static bool plan_gemm_large(int64_t m, int64_t n, int64_t k, size_t ws_bytes, GemmPlan* pl) {
  if (!((m >= 4096 || k >= 4096) && m * n >= ((int64_t)1 << 20) && k >= 1024 && m > 128)) return false;
  static const int cand[] = {TILE_128, TILE_128x96, TILE_96x128, TILE_128x64, TILE_64, TILE_128x288, TILE_288x128};
  // per-flop cost of the tile shape (a whole-width tile streams the long operand once)
  static const double cost[] = {1.00, 1.04, 1.04, 1.15, 1.40, 0.98, 0.98};
  // (measured on the scaled-batch fit, tools/scaled_tile_ab.py / profiles/r03_scaled_tile_ab.txt: the
  // whole-width tiles run the forward product at 0.52 and dW at 0.51-0.58 of the fp32 MFMA peak, no
  // better than the 96-wide ones -- one wavefront per SIMD does not hide the operand latency -- so
  // they stay opt-in: BSIG_GEMM_WIDE_TILE=1)
  // (and they are only in a library built with BSIG_BUILD_WIDE_TILES=1 ./build.sh: each of the two
  // takes 3.5 minutes to compile, longer than the rest of the library together)
#ifdef BSIG_WITH_WIDE_TILES
  const bool wide_ok = env_int("BSIG_GEMM_WIDE_TILE", 0) != 0;
#else
  const bool wide_ok = false;
#endif
  double best = 0.0;
  int best_t = -1;
  int64_t best_tiles = 0;
  for (int i = 0; i < 7; ++i) {
    const int t = cand[i];
    if (!wide_ok && (t == TILE_128x288 || t == TILE_288x128)) continue;
    const int64_t tm = ceil_div<int64_t>(m, kTileM[t]), tn = ceil_div<int64_t>(n, kTileN[t]);
    const double padded = (double)(tm * kTileM[t]) * (double)(tn * kTileN[t]) * cost[i];
    if (best_t < 0 || padded < best) { best = padded; best_t = t; best_tiles = tm * tn; }
  }
  int64_t splits = 1;
  if (best_t == TILE_128x288 || best_t == TILE_288x128) {
    // 60 KB of LDS and 9 accumulators per wave: at most two workgroups per CU -- one round of 256..512
    splits = std::max<int64_t>(env_int("BSIG_GEMM_WIDE_WGS", 256) / best_tiles, 1);
    splits = std::min<int64_t>(splits, std::max<int64_t>(k / (8 * BK), 1));
  } else if (best_tiles < 640) {
    splits = ceil_div<int64_t>(768, best_tiles);
    splits = std::min<int64_t>(splits, std::max<int64_t>(k / (8 * BK), 1));
    splits = std::min<int64_t>(splits, 32);
  }
  const int64_t max_by_ws = (int64_t)(ws_bytes / (sizeof(float) * (size_t)(m * n)));
  splits = std::max<int64_t>(std::min(splits, max_by_ws), 1);
  const int64_t chunk = round_up<int64_t>(ceil_div<int64_t>(k, splits), BK);
  pl->tile = best_t;
  pl->splits = (int)ceil_div<int64_t>(k, chunk);
  pl->k_chunk = (int)chunk;
  return true;
}

// lean_ok: gemm_lean_kernel will run this product (both operands move as aligned 16-byte items)
static GemmPlan plan_gemm(int64_t m, int64_t n, int64_t k, size_t ws_bytes, bool lean_ok) {
  GemmPlan pl;
  if (env_int("BSIG_GEMM_TILE", -1) < 0 && env_int("BSIG_GEMM_SPLITS", 0) <= 0 &&
      env_int("BSIG_GEMM_NO_LARGE_PLAN", 0) == 0 && plan_gemm_large(m, n, k, ws_bytes, &pl))
    return pl;
  const int64_t t128 = ceil_div<int64_t>(m, 128) * ceil_div<int64_t>(n, 128);
  int64_t tiles, target, min_splits = 1;
  const int64_t t12864 = ceil_div<int64_t>(m, 128) * ceil_div<int64_t>(n, 64);
  if (m >= 256 && n >= 128 && t128 >= 192) {
    pl.tile = TILE_128; tiles = t128; target = 256;
  } else if (m >= 256 && n >= 128 && t12864 >= 160 && t12864 <= 512) {
    // mid-sized (a chunk's RFF projection: 1000 x 2048 x 2310 -- training and held-out rows in one
    // launch).  With the lean loop (two LDS images: four 64 x 64 workgroups per CU) the unsplit 64 x 64
    // tiling is the fastest of the sweep (profiles/r05_rff_chunk_sweep.txt: 1000 rows = 16 x 32 = 512
    // workgroups, two per CU, 90 us against 97 for 128 x 64 tiles in two K halves; 800 rows 95 = 95).
    // (round 2-4, gemm_mfma_kernel: 128x64 tiles in two k-halves, 101-104 us against 118-135.)
    // That measurement holds for the LEAN kernel only: operands that are not 16-byte aligned still run
    // gemm_mfma_kernel, where unsplit 64 x 64 tiles measured 118-135 us against 101-104 for the old plan
    // -- they keep 128 x 64 tiles in two K halves (round-5 advisor finding).
    if (lean_ok) { pl.tile = TILE_64; tiles = ceil_div<int64_t>(m, 64) * ceil_div<int64_t>(n, 64); target = 0; }
    else { pl.tile = TILE_128x64; tiles = t12864; target = 0; min_splits = 2; }
  } else if (m <= 128 && n >= 64) {
    pl.tile = TILE_128x32; tiles = ceil_div<int64_t>(n, 32); target = 512;
  } else {
    pl.tile = TILE_64; tiles = ceil_div<int64_t>(m, 64) * ceil_div<int64_t>(n, 64); target = 512;
  }
  const int forced_tile = env_int("BSIG_GEMM_TILE", -1);
  if (forced_tile >= 0) {
    pl.tile = forced_tile;
    const int bm = kTileM[forced_tile % N_TILES], bn = kTileN[forced_tile % N_TILES];
    tiles = ceil_div<int64_t>(m, bm) * ceil_div<int64_t>(n, bn);
  }
  int64_t splits = 1;
  if (tiles < target && tiles < 256) {   // fewer workgroups than CUs: split K
    splits = ceil_div<int64_t>(target, tiles);
    const int64_t max_by_k = k / (4 * BK) > 0 ? k / (4 * BK) : 1;   // >= PD K steps per block
    if (splits > max_by_k) splits = max_by_k;
    if (splits > 64) splits = 64;
  }
  if (forced_tile < 0 && splits < min_splits && k / (4 * BK) >= min_splits) splits = min_splits;
  const int forced = env_int("BSIG_GEMM_SPLITS", 0);
  if (forced > 0) splits = forced;
  const int64_t max_by_ws = (int64_t)(ws_bytes / (sizeof(float) * (size_t)(m * n)));
  if (splits > max_by_ws) splits = max_by_ws;
  if (splits < 1) splits = 1;
  int64_t chunk = round_up<int64_t>(ceil_div<int64_t>(k, splits), BK);
  splits = ceil_div<int64_t>(k, chunk);
  pl.splits = (int)splits;
  pl.k_chunk = (int)chunk;
  return pl;
}

static int pick_vec(const float* ptr, int64_t ld) {
  return (ld % 4 == 0 && aligned(ptr, 16)) ? 4 : 1;
}
// k-contiguous operand with unaligned rows (ld % 4 != 0: the cross-correlation widths S*A + 2):
// quads at any 4-byte address, the row's last one shifted into place
static bool unaligned_quads(const float* ptr, int64_t ld, int kmajor, int k) {
  return !kmajor && pick_vec(ptr, ld) == 1 && ld >= 4 && k >= 4 && getenv("BSIG_GEMM_NO_UNALIGNED") == nullptr;
}

bool gemm_wide_gradient_applies(int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, const float* a,
                                const float* b, bool gathered) {
  return env_int("BSIG_GEMM_WIDE", 1) && gathered && k >= 2048 && k % BK == 0 && gemm_wide_covers((int)m) &&
         lda == ceil_div<int64_t>(m, 16) * 16 && n % 64 == 0 && pick_vec(a, lda) == 4 && pick_vec(b, ldb) == 4;
}

int gemm_run(GemmParams p, void* workspace, size_t workspace_bytes, hipStream_t st,
             int* n_expsum) {
  BSIG_REQUIRE(p.a && p.b && p.c, "gemm: null pointer");
  BSIG_REQUIRE(p.m >= 0 && p.n >= 0 && p.k >= 0, "gemm: bad dims");
  const int e = p.epilogue;
  BSIG_REQUIRE((e >= BSIG_EPI_NONE && e <= BSIG_EPI_MUL_DACT) || e == EPI_ADAM,
               "gemm: bad epilogue");
  BSIG_REQUIRE(!((e == BSIG_EPI_BIAS || e == BSIG_EPI_BIAS_ACT || e == BSIG_EPI_COS_OFF) &&
                 !p.bias), "gemm: epilogue needs bias");
  BSIG_REQUIRE(!(e == BSIG_EPI_MUL_DACT && !p.aux), "gemm: epilogue needs aux");
  BSIG_REQUIRE(!(e == EPI_ADAM && !(p.adam_m && p.adam_v && p.adam_dyn)),
               "gemm: Adam epilogue needs its state");
  BSIG_REQUIRE(p.ldc >= (e == BSIG_EPI_COS_SIN ? 2 * (int64_t)p.n : (int64_t)p.n),
               "gemm: ldc too small");
  if (n_expsum) *n_expsum = 0;
  BSIG_REQUIRE(!(p.expsum && e != BSIG_EPI_BIAS), "gemm: expsum needs the bias epilogue");
  if (p.m == 0 || p.n == 0) return BSIG_OK;
  {
    const char* f64_env = getenv("BSIG_DEBUG_F64_ACC_MIN_K");     // (read per call: tests switch it)
    const int f64_min_k = f64_env ? atoi(f64_env) : 0;
    if (f64_min_k > 0 && p.k >= f64_min_k) {
      const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>((int64_t)p.m * p.n, 256), 2048);
      hipLaunchKernelGGL(gemm_f64acc_kernel, dim3(blocks), dim3(256), 0, st, p);
      BSIG_CHECK_LAUNCH("gemm_f64acc");
      if (n_expsum && p.expsum) *n_expsum = blocks;
      return BSIG_OK;
    }
  }
  // Large-minibatch products of a few-hundred-wide head (the scaled-batch update): whole-width tiles
  // of 16x16x4 MFMAs (gemm_wide.h) -- forward O = X[ids] W^T and the gradient dW = dO^T X[ids] with
  // dO at a pitch of ceil16(Nh) floats.  Split K so that the launch is one round of ~256 workgroups;
  // the slabs are reduced (and the epilogue applied) by gemm_reduce_kernel as for every split product.
  if (env_int("BSIG_GEMM_WIDE", 1) && workspace && p.k >= 1024 && p.k % BK == 0 && !p.a_unal && !p.b_unal) {
    WideParams wp;
    wp.dyn = p.dyn; wp.dyn_delta = p.dyn_delta; wp.k = p.k;
    int rc = BSIG_EUNSUPPORTED;
    auto split = [&](int64_t tiles, int64_t slab_floats) {
      // one workgroup per CU at a time (123 KB of LDS): the launch takes ceil(tiles * s / 256) rounds
      // of 1 / s of a tile's K loop each, plus a slab's worth of traffic per slice -- e.g. 313 row
      // tiles (a 20000-row evaluation): unsplit 2 rounds of the whole K loop, in 4 slices 5 rounds of
      // a quarter
      const int64_t smax = std::min<int64_t>(std::min<int64_t>(8, std::max<int64_t>(p.k / (8 * BK), 1)),
                                             (int64_t)(workspace_bytes / (sizeof(float) * (size_t)slab_floats)));
      if (smax < 1) return false;
      int64_t s = 1;
      double best = 1e30;
      for (int64_t c = 1; c <= smax; ++c) {
        const double cost = (double)ceil_div<int64_t>(tiles * c, 256) / (double)c + 0.02 * (double)c;
        if (cost < best - 1e-9) { best = cost; s = c; }
      }
      const int64_t chunk = round_up<int64_t>(ceil_div<int64_t>(p.k, s), BK);
      wp.k_chunk = (int)chunk; wp.splits = (int)ceil_div<int64_t>(p.k, chunk);
      wp.slab = slab_floats;
      return true;
    };
    if (!p.a_kmajor && !p.b_kmajor && !p.b_rows && p.m >= 2048 && gemm_wide_covers(p.n) &&
        pick_vec(p.a, p.lda) == 4 && pick_vec(p.b, p.ldb) == 4 && !p.b_dyn_stride && !p.b_dyn_base &&
        (p.epilogue == BSIG_EPI_NONE || p.epilogue == BSIG_EPI_BIAS)) {
      const int wide = ceil_div(p.n, 16) * 16;
      if (split(ceil_div<int64_t>(p.m, 64), (int64_t)p.m * wide)) {
        wp.wide = p.b; wp.ld_wide = p.ldb; wp.n_wide = p.n;
        wp.x = p.a; wp.ldx = p.lda; wp.ids = p.a_rows; wp.n_narrow = p.m;
        wp.dyn_stride = p.a_dyn_stride; wp.dyn_base = p.a_dyn_base;
        wp.out = reinterpret_cast<float*>(workspace); wp.ld_out = wide;
        // (one ticket per 64-row tile: a product of more row tiles than the caller's ticket words -- a head
        // whose width is already a multiple of 16 keeps ldc == wide at ANY row count -- must not combine
        // in the launch: the tickets beyond the buffer are somebody else's memory)
        const bool combine = p.combine_tickets && ceil_div(p.m, 64) <= p.combine_capacity &&
                             p.epilogue == BSIG_EPI_BIAS && p.ldc >= wide &&
                             (p.ldc & 3) == 0 && aligned(p.c, 16) && wp.slab < ((int64_t)1 << 31) &&
                             env_int("BSIG_GEMM_NO_COMBINE", 0) == 0;
        if (combine) {
          wp.tickets = p.combine_tickets; wp.bias = p.bias; wp.final_out = p.c; wp.ld_final = p.ldc;
          wp.expsum = p.expsum; wp.expsum_col0 = p.expsum_col0; wp.expsum_ncols = p.expsum_ncols;
        }
        rc = gemm_wide_forward(wp, st);
        if (rc == BSIG_OK && combine) {
          if (n_expsum && p.expsum) *n_expsum = ceil_div(p.m, 64);
          return BSIG_OK;
        }
        if (rc == BSIG_OK) { p.partial_ld = wide; p.partial_slab = wp.slab; }
      }
    } else if (p.a_kmajor && p.b_kmajor && !p.a_rows && !p.a_dyn_stride && !p.a_dyn_base &&
               gemm_wide_gradient_applies(p.m, p.n, p.k, p.lda, p.ldb, p.a, p.b, p.b_rows != nullptr)) {
      if (split(p.n / 64, (int64_t)p.m * p.n)) {
        wp.wide = p.a; wp.ld_wide = p.lda; wp.n_wide = p.m;
        wp.x = p.b; wp.ldx = p.ldb; wp.ids = p.b_rows; wp.n_narrow = p.n;
        wp.dyn_stride = p.b_dyn_stride; wp.dyn_base = p.b_dyn_base;
        wp.out = reinterpret_cast<float*>(workspace); wp.ld_out = p.n;
        rc = gemm_wide_gradient(wp, st);
      }
    }
    if (rc == BSIG_OK) {
      p.splits = wp.splits; p.k_chunk = wp.k_chunk;
      p.partial = reinterpret_cast<float*>(workspace);
      return launch_reduce(p, st, n_expsum);
    }
    if (rc != BSIG_EUNSUPPORTED) return rc;
  }
  BSIG_REQUIRE(p.bias_g_n <= 1, "gemm: partial bias sums outside the whole-width gradient path");
  // a device-resolved row offset changes the alignment of nothing: offsets are whole rows
  {
    static const int xcd = [] { const char* e = getenv("BSIG_GEMM_XCD"); return e ? atoi(e) : 1; }();
    p.xcd_swz = xcd;
  }
  p.a_unal = unaligned_quads(p.a, p.lda, p.a_kmajor, p.k);
  p.b_unal = unaligned_quads(p.b, p.ldb, p.b_kmajor, p.k);
  const int avec = p.a_unal ? 4 : pick_vec(p.a, p.lda), bvec = p.b_unal ? 4 : pick_vec(p.b, p.ldb);
  // the lean main loop (gemm_lean.h: bit-identical, no vector-ALU work in the K loop) wherever
  // both operands move as aligned 16-byte items
  const int lean = env_int("BSIG_GEMM_LEAN", 1);   // (read per call: tests compare the two kernels)
  const bool lean_ok = avec == 4 && bvec == 4 && !p.a_unal && !p.b_unal;
  // (the plan does not depend on BSIG_GEMM_LEAN: the two kernels are compared on the SAME plan)
  const GemmPlan pl = plan_gemm(p.m, p.n, p.k, workspace ? workspace_bytes : 0, lean_ok);
  p.splits = pl.splits; p.k_chunk = pl.k_chunk;
  p.partial = reinterpret_cast<float*>(workspace);
  int rc = BSIG_EUNSUPPORTED;
  if (lean && lean_ok) {
    const bool akm = p.a_kmajor != 0, bkm = p.b_kmajor != 0;
    switch (pl.tile) {
      case TILE_64: rc = launch_lean_64(p, akm, bkm, st); break;
      case TILE_128: rc = launch_lean_128(p, akm, bkm, st); break;
      case TILE_128x32: rc = launch_lean_128x32(p, akm, bkm, st); break;
      case TILE_128x64: rc = launch_lean_128x64(p, akm, bkm, st); break;
      case TILE_128x96: rc = launch_lean_128x96(p, akm, bkm, st); break;
      case TILE_96x128: rc = launch_lean_96x128(p, akm, bkm, st); break;
      default: break;
    }
  }
  if (rc != BSIG_EUNSUPPORTED) {
  } else
  if (pl.tile == TILE_128)
    rc = launch_tile_128(p, p.a_kmajor != 0, p.b_kmajor != 0, avec, bvec, st);
  else if (pl.tile == TILE_128x32)
    rc = launch_tile_128x32(p, p.a_kmajor != 0, p.b_kmajor != 0, avec, bvec, st);
  else if (pl.tile == TILE_128x64)
    rc = launch_tile_128x64(p, p.a_kmajor != 0, p.b_kmajor != 0, avec, bvec, st);
  else if (pl.tile == TILE_128x96)
    rc = launch_tile_128x96(p, p.a_kmajor != 0, p.b_kmajor != 0, avec, bvec, st);
  else if (pl.tile == TILE_96x128)
    rc = launch_tile_96x128(p, p.a_kmajor != 0, p.b_kmajor != 0, avec, bvec, st);
#ifdef BSIG_WITH_WIDE_TILES
  else if (pl.tile == TILE_128x288)
    rc = launch_tile_128x288(p, p.a_kmajor != 0, p.b_kmajor != 0, avec, bvec, st);
  else if (pl.tile == TILE_288x128)
    rc = launch_tile_288x128(p, p.a_kmajor != 0, p.b_kmajor != 0, avec, bvec, st);
#endif
  else
    rc = launch_tile_64(p, p.a_kmajor != 0, p.b_kmajor != 0, avec, bvec, st);
  if (rc != BSIG_OK) return rc;
  BSIG_CHECK_LAUNCH("gemm_mfma");
  if (p.splits > 1) {
    BSIG_TRY(launch_reduce(p, st, n_expsum));
  } else if (n_expsum && p.expsum) {
    const int bm = kTileM[pl.tile], bn = kTileN[pl.tile];
    *n_expsum = ceil_div(p.m, bm) * ceil_div(p.n, bn);
  }
  return BSIG_OK;
}

int gemm_f32(const float* a, int64_t lda, int a_kmajor, const int32_t* a_rows,
             const float* b, int64_t ldb, int b_kmajor, const int32_t* b_rows, float* c,
             int64_t ldc, int64_t m, int64_t n, int64_t k, int epilogue, int act,
             const float* bias, const float* aux, int64_t ldaux, float alpha,
             void* workspace, size_t workspace_bytes, hipStream_t st) {
  BSIG_REQUIRE(m >= 0 && n >= 0 && k >= 0 && m < (1 << 30) && n < (1 << 30) && k < (1 << 30),
               "gemm: bad dims");
  BSIG_REQUIRE(epilogue >= BSIG_EPI_NONE && epilogue <= BSIG_EPI_MUL_DACT, "gemm: bad epilogue");
  GemmParams p;
  p.a = a; p.lda = lda; p.a_rows = a_rows; p.a_kmajor = a_kmajor;
  p.b = b; p.ldb = ldb; p.b_rows = b_rows; p.b_kmajor = b_kmajor;
  p.c = c; p.ldc = ldc;
  p.m = (int)m; p.n = (int)n; p.k = (int)k;
  p.epilogue = epilogue; p.act = act; p.bias = bias; p.aux = aux; p.ldaux = ldaux;
  p.alpha = alpha;
  return gemm_run(p, workspace, workspace_bytes, st);
}

__global__ __launch_bounds__(256) void rff_coeff_kernel(const float* __restrict__ freqs,
                                                        const float* __restrict__ sigma,
                                                        float* __restrict__ coeff,
                                                        int64_t m_feat, int64_t in_dim,
                                                        int64_t ld) {
  const int64_t total = m_feat * ld;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / ld, c = e % ld;
    coeff[e] = c < in_dim ? freqs[r * in_dim + c] / sigma[c] : 0.f;
  }
}

}  // namespace bsig

using namespace bsig;

extern "C" size_t bsig_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k) {
  size_t need = (size_t)64 * (size_t)m * (size_t)n * sizeof(float) <= ((size_t)256 << 20)
                    ? (size_t)64 * (size_t)m * (size_t)n * sizeof(float)
                    : plan_gemm(m, n, k, (size_t)1 << 40, false).splits * (size_t)m * (size_t)n * sizeof(float);
  // the whole-width products (gemm_wide.h): up to 8 K slices of slabs at the padded head width
  if (m >= 2048 && gemm_wide_covers((int)n))
    need = std::max(need, (size_t)8 * (size_t)m * (size_t)round_up<int64_t>(n, 16) * sizeof(float));
  if (k >= 2048 && gemm_wide_covers((int)m))
    need = std::max(need, (size_t)8 * (size_t)m * (size_t)n * sizeof(float));
  return need;
}

extern "C" int bsig_gemm_f32(const float* a, int64_t lda, int a_kmajor,
                             const int32_t* a_rows, const float* b, int64_t ldb,
                             int b_kmajor, const int32_t* b_rows, float* c, int64_t ldc,
                             int64_t m, int64_t n, int64_t k, int epilogue, int act,
                             const float* bias, const float* aux, int64_t ldaux, float alpha,
                             void* workspace, size_t workspace_bytes, bsig_stream_t stream) {
  return gemm_f32(a, lda, a_kmajor, a_rows, b, ldb, b_kmajor, b_rows, c, ldc, m, n, k,
                  epilogue, act, bias, aux, ldaux, alpha, workspace, workspace_bytes,
                  as_stream(stream));
}

extern "C" int bsig_rff_coeff(const float* freqs, const float* sigma, float* coeff,
                              int64_t m_feat, int64_t in_dim, int64_t ld_coeff,
                              bsig_stream_t stream) {
  BSIG_REQUIRE(freqs && sigma && coeff && ld_coeff >= in_dim, "rff_coeff: bad args");
  const int64_t total = m_feat * ld_coeff;
  if (total == 0) return BSIG_OK;
  const int blocks = (int)std::min<int64_t>(ceil_div<int64_t>(total, 256), 4096);
  hipLaunchKernelGGL(rff_coeff_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), freqs,
                     sigma, coeff, m_feat, in_dim, ld_coeff);
  BSIG_CHECK_LAUNCH("rff_coeff");
  return BSIG_OK;
}

extern "C" int bsig_rff_project(const float* x, int64_t ldx, const int32_t* x_rows,
                                const float* coeff, int64_t ld_coeff, const float* offset,
                                float* feats, int64_t ld_feats, int64_t batch,
                                int64_t in_dim, int64_t m_feat, float a, int cos_only,
                                void* workspace, size_t workspace_bytes,
                                bsig_stream_t stream) {
  bsig::Range roctx_range("bsig_rff_project");
  BSIG_REQUIRE(!(cos_only && !offset), "rff_project: cos-only features need an offset");
  return gemm_f32(x, ldx, 0, x_rows, coeff, ld_coeff, 0, nullptr, feats, ld_feats, batch,
                  m_feat, in_dim, cos_only ? BSIG_EPI_COS_OFF : BSIG_EPI_COS_SIN, 0, offset,
                  nullptr, 0, a, workspace, workspace_bytes, as_stream(stream));
}
