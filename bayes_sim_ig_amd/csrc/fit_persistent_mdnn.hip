// Persistent update kernel for the MDNN estimator with the reference's default
// trunk (hidden_layers [128, 128], tanh, diagonal covariance): a run of
// consecutive SGD updates (MDNN.run_training's inner loop, mdnn.py:219-233:
// forward :89-125, NLL :127-178, backward, Adam :203/:229) in ONE launch.
//
// As separate kernels an update is ~15 launches of 9-18 us each (two trunk GEMMs
// with split-K reduces, head GEMM, NLL, finish, three dX GEMMs, three dW+Adam
// GEMMs, bias column sums).  Here three kinds of workgroups (512 threads, one per
// CU) keep the model resident and hand each other activations through
// write-through stores and flags (see fit_persistent.hip for the mechanism):
//
//  * tile workgroups (4 x ceil(I/256)): own the 32 x 256 tile W1[32nb.., 256ks..]
//    of the first layer (weights in LDS, both Adam moments in registers, MFMA
//    accumulator layout).  Per update: gather their [B, 256] slice of the
//    minibatch summaries into LDS, partial product X_slice W1_tile^T (fp32 MFMA
//    32x32x2) -> split-K slab; later dW1_tile = dz1^T X_slice and Adam.
//  * row-owner workgroups (ceil(B/4)): own 4 minibatch rows.  Sum the k-slices
//    (+b1, tanh) -> h1; h2 = tanh(h1 W2^T + b2); head outputs; the row-wise
//    mixture NLL forward/backward (diag_row, one wavefront per row); dz2 = (d_out
//    Wh) * (1 - h2^2); dz1 = (dz2 W2) * (1 - h1^2).  The four small products run
//    on the MFMA units (fp32 16x16x4) with W2 in registers (both operand layouts)
//    and the head matrix in LDS (XOR-swizzled rows, conflict-free for both
//    directions), refreshed once per update.
//  * small-weight workgroups (4 + ceil(Nh/32)): own 32 rows of W2 or of the head
//    matrix with their Adam moments in registers: dW = d^T h over the minibatch
//    (MFMA 32x32x2), Adam, rows written through for the owners' next refresh.
//
// Per update: tiles -> (slabs) -> owners -> (dz1 | h1, h2, dz2, d_out) -> tiles |
// small-weight workgroups -> (W2, Wh) -> owners of the next update.  Every sum
// across workgroups is taken in a fixed order: runs are bitwise reproducible.



#include "persist_mdnn_device.h"

namespace bsig {

// DP: data-parallel rank (gradients out, pending Adam step in; see MdnnArgs)
// FAC: the summary rows arrive as cross-correlation factor rows
// WIDE: heads wider than the owners' LDS: head outputs formed by the head-block workgroups
// MR: minibatch rows per owner workgroup
template <bool DP, bool FAC, bool WIDE, bool FULL, int MR = kMR>
__global__ __launch_bounds__(kMT) void mdnn_updates_kernel(MdnnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // (BSIG_HOST_SAN_BUILD: tools/build_host_san.sh checks the HOST side under the sanitizers and
  // launches nothing -- the device bodies, minutes of compile time, are left out of that build)
#ifndef BSIG_HOST_SAN_BUILD
  const int wg = blockIdx.x;
  if (wg < p.G1) mdnn_tile_workgroup<DP, FAC>(p, smem);
  else if (wg < p.G1 + p.n_owner) mdnn_owner_workgroup<DP, WIDE, FULL, MR>(p, smem);
  else mdnn_small_workgroup<DP, WIDE>(p, smem);
#endif
}

// ---------------------------------------------------------------- host side
struct MdnnGeom {
  int FR, Nh, Nh16, NhP, k_slices, G1, n_owner, n_small, x_floats;
  int wide;                     // head outputs formed by the head-block workgroups (see MdnnArgs)
  int stream, s_chunks;         // W1 streamed by G1 tile workgroups (fit_persistent_mdnn_stream.hip)
  int mr;                       // minibatch rows per owner workgroup
  int eval_passes;              // 0: evaluations stay outside the launches
  size_t lds;
  size_t slab_floats, act_floats, dout_floats, eval_floats, eval_slab_floats;
};

static bool mdnn_geom(const PersistMdnnShape& s, MdnnGeom* g) {
  if (s.batch < 1 || s.input_dim < 1 || s.h1 != kMH || s.h2 != kMH ||
      s.activation != BSIG_ACT_TANH || s.out_dim < 1 || s.n_comp < 1 || s.n_comp > 64)
    return false;
  const int groups = 64 / s.n_comp;
  const bool full = s.full_cov != 0 && s.out_dim >= 2;
  // (full covariance: one lane per component runs D sequential solves -- small theta only)
  if (full ? s.out_dim > 16 : ceil_div(s.out_dim, groups) > kElemsPerLane) return false;
  const int ls = full ? s.out_dim * (s.out_dim - 1) / 2 : 0;
  g->FR = (int)round_up(s.batch, 8);
  if (g->FR > 104) return false;               // 13 float4 of the summary tile per thread
  g->Nh = s.n_comp * (1 + 2 * s.out_dim + ls);
  g->Nh16 = (int)round_up(g->Nh, 16);
  g->NhP = (int)round_up(g->Nh, kMNB);
  g->k_slices = ceil_div(s.input_dim, kMC);
  g->G1 = (kMH / kMNB) * g->k_slices;
  g->n_small = kMH / kMNB + g->NhP / kMNB;
  // Rows per owner: an owner pulls its rows of every k-slice's slab through ONE CU's vector-memory
  // pipe (~40 GB/s of cache-bypassing loads: 47 slices x 4 rows x 512 B = 96 KB = 2.4 us for the Ant
  // summaries), and the per-row loops of its chain shrink with its rows -- one row each where the chip has
  // the CUs for B owners, else two, else four (diagonal covariance; same sums in the same order:
  // slab_quads_sum).  BSIG_MDNN_MR=1|2|4: A/B runs, tests.
  g->mr = kMR;
  {
    static const int force_mr = [] { const char* e = getenv("BSIG_MDNN_MR"); return e ? atoi(e) : 0; }();
    for (int mr = 1; mr <= 2 && !s.full_cov && force_mr != 4; ++mr)
      if ((force_mr == 0 || force_mr == mr) && g->G1 + ceil_div(s.batch, mr) + g->n_small <= kXwgMax) { g->mr = mr; break; }
  }
  g->n_owner = ceil_div(s.batch, g->mr);
  g->stream = 0; g->s_chunks = 0;
  const char* no_stream = getenv("BSIG_NO_STREAMED_W1");
  if (g->G1 > kXwgMax || g->G1 + g->n_owner + g->n_small > kXwgMax) {
    // W1 (+ its Adam moments) does not fit the chip's LDS and registers: the tile workgroups
    // stream it (cross-correlation factor rows only; fit_persistent_mdnn_stream.hip)
    if (no_stream && no_stream[0] == '1') return false;
    g->stream = 1;
    // (the owners read ONE summed slab here: 8 rows each cost them nothing and free 12 CUs)
    g->mr = kStreamMR;
    g->n_owner = ceil_div(s.batch, g->mr);
    g->G1 = (kXwgMax - g->n_owner - g->n_small) / 4 * 4;     // blocks of four (32 hidden units each)
    g->s_chunks = mdnn_stream_chunks(s.input_dim);
    if (s.input_dim % 2 != 0) return false;                  // rows of W1 as 8-byte aligned pairs
    if (g->G1 < 64) return false;
  }
  g->x_floats = (int)round_up(std::max(128 * kMPbuf, kMNB * (g->FR + 4)), 4);
  const size_t tile_lds = ((size_t)g->FR * kMPitch + (size_t)kMNB * kMPitch + g->x_floats + 64 + 96 +
                           (kMT / 32) * 32) * sizeof(float);
  size_t owner_lds = ((size_t)g->Nh16 * kMH + 2 * g->mr * kMHP + (size_t)g->mr * (g->Nh16 + 4) + kMH +
                      g->Nh16 + (size_t)g->mr * (s.out_dim + 3 * s.n_comp + (full ? 3 * s.out_dim * s.n_comp : 0)) +
                      64 + (g->stream ? 0 : (kSumSub - 1) * kSumItems * 4)) * sizeof(float);
  size_t small_lds = ((size_t)g->FR * kMHP + (size_t)kMNB * (g->FR + 4) + 64) * sizeof(float);
  g->wide = owner_lds > (size_t)kMLdsLimit ? 1 : 0;
  const char* force_wide = getenv("BSIG_MDNN_WIDE_HEADS");     // tests: the wide path on small heads
  if (force_wide && force_wide[0] == '1') g->wide = 1;
  if (g->wide) {
    owner_lds -= (size_t)g->Nh16 * kMH * sizeof(float);         // no head matrix in the owners
    // + the block's weights in operand order [32][kMHP] and the k-half exchange [128][33]
    small_lds += ((size_t)kMNB * kMHP + 128 * kMPbuf + kMNB) * sizeof(float);
  }
  // the forward of the tile workgroups reads summary rows up to 127
  g->lds = std::max(std::max(tile_lds, (size_t)128 * kMPitch * sizeof(float)),
                    std::max(owner_lds, small_lds));
  if (g->stream) g->lds = std::max(owner_lds, small_lds);   // + the tile workgroups' (S, A known at bind)
  if (g->lds > (size_t)kMLdsLimit) return false;
  g->slab_floats = g->stream ? (size_t)g->G1 * s.batch * kMNB : (size_t)g->k_slices * s.batch * kMH;
  g->act_floats = (size_t)s.batch * kMH;
  g->dout_floats = (size_t)s.batch * g->NhP;
  // in-launch evaluations: slabs of both parities and the parked head outputs
  g->eval_passes = s.max_test > 0 ? ceil_div(s.max_test, s.batch) : 0;
  if (g->eval_passes > 8) g->eval_passes = 0;
  if (g->wide) {   // (A/B switch: a wide-head plan's evaluations as graphs between its launches)
    const char* e = getenv("BSIG_NO_WIDE_EVAL");
    if (e && e[0] == '1') g->eval_passes = 0;
  }
  if (g->stream) {   // (A/B switch: a streamed plan's evaluations as graphs between its launches)
    const char* e = getenv("BSIG_NO_STREAM_EVAL");
    if (e && e[0] == '1') g->eval_passes = 0;
  }
  // (streamed W1: an evaluation pass leaves ONE summed slab [B][128], like an update)
  g->eval_slab_floats = g->stream ? g->act_floats : g->slab_floats;
  g->eval_floats = (size_t)2 * g->eval_passes * g->eval_slab_floats +
                   (size_t)g->eval_passes * s.batch * g->NhP +
                   (g->wide && g->eval_passes > 0 ? g->act_floats + g->dout_floats : 0);   // h2e, oe
  return true;
}

static bool mdnn_device_can_host(const MdnnGeom& g) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
  if (prop.multiProcessorCount < g.G1 + g.n_owner + g.n_small || (size_t)prop.maxSharedMemoryPerMultiProcessor < g.lds)
    return false;
  // the runtime's own occupancy answer (registers, LDS, wave slots) must admit a workgroup per CU
  int per_cu = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(mdnn_updates_kernel<false, false, false, false>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, kMLdsLimit) != hipSuccess ||
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mdnn_updates_kernel<false, false, false, false>, kMT,
                                                   std::min(g.lds, (size_t)kMLdsLimit)) != hipSuccess)
    return false;
  return per_cu >= 1;
}

int persist_mdnn_geometry(const PersistMdnnShape& s, int32_t* out) {
  MdnnGeom g;
  for (int i = 0; i < 16; ++i) out[i] = 0;
  if (!mdnn_geom(s, &g)) return 0;
  const int32_t v[10] = {g.k_slices, g.G1, g.n_owner, g.mr, g.n_small, g.wide, g.stream, g.eval_passes,
                         (int32_t)g.lds, g.Nh};
  for (int i = 0; i < 10; ++i) out[i] = v[i];
  return 1;
}

bool persist_mdnn_supported(const PersistMdnnShape& s) {
  MdnnGeom g;
  return mdnn_geom(s, &g) && mdnn_device_can_host(g);
}
int persist_mdnn_streams(const PersistMdnnShape& s) {
  MdnnGeom g;
  return mdnn_geom(s, &g) && mdnn_device_can_host(g) && g.stream ? 1 : 0;
}
// the LDS of a streamed plan's tile workgroups for S x A cross-correlation factors
static bool mdnn_stream_fits(const MdnnGeom& g, int S, int A, int* nip, int* pf, size_t* lds) {
  return mdnn_stream_tile_geom(g.FR, ceil_div(g.s_chunks, g.G1 / 4), S, A, nip, pf, lds);
}
bool persist_mdnn_accepts_factors(const PersistMdnnShape& s, int S, int A) {
  MdnnGeom g;
  if (!mdnn_geom(s, &g) || !mdnn_device_can_host(g)) return false;
  if ((int64_t)S * A + 2 != s.input_dim || S < 1 || A < 1) return false;
  if (!g.stream) return true;
  int nip, pf; size_t lds;
  return mdnn_stream_fits(g, S, A, &nip, &pf, &lds);
}
bool persist_mdnn_eval_supported(const PersistMdnnShape& s) {
  MdnnGeom g;
  return mdnn_geom(s, &g) && mdnn_device_can_host(g) && g.eval_passes > 0;
}
// ... by a data-parallel rank (one launch per update): narrow heads on a resident first layer only
bool persist_mdnn_dp_eval_supported(const PersistMdnnShape& s) {
  MdnnGeom g;
  return mdnn_geom(s, &g) && mdnn_device_can_host(g) && g.eval_passes > 0 && !g.wide && !g.stream;
}

constexpr size_t kPackFloats = (size_t)2 * kMH * kMH;   // both parities
static size_t mdnn_wide_floats(const MdnnGeom& g) {
  return g.wide ? g.dout_floats + (size_t)(g.NhP / kMNB) * g.act_floats : 0;
}
static size_t mdnn_data_bytes(const MdnnGeom& g) {
  return round_up<size_t>((g.slab_floats + 4 * g.act_floats + g.dout_floats + 2 * kPackFloats +
                           g.eval_floats + mdnn_wide_floats(g) + (g.stream ? g.act_floats : 0)) * sizeof(float), 256);
}
// flags, granules and (last 256 bytes) the word the workgroups of a resident rank count themselves in
static size_t mdnn_sync_bytes() { return 14 * kFlagArr * sizeof(unsigned) + 6 * kGranArr * 8 + 256; }

size_t persist_mdnn_workspace_bytes(const PersistMdnnShape& s) {
  MdnnGeom g;
  if (!mdnn_geom(s, &g)) return 0;
  return mdnn_data_bytes(g) + mdnn_sync_bytes();
}

int persist_mdnn_reset_regions(const PersistMdnnShape& s, void* workspace, size_t workspace_bytes,
                               ZeroRegion* regions) {
  MdnnGeom g;
  BSIG_REQUIRE(mdnn_geom(s, &g), "persistent MDNN updates: shape not covered");
  BSIG_REQUIRE(workspace && workspace_bytes >= persist_mdnn_workspace_bytes(s),
               "persistent MDNN updates: workspace too small");
  char* base = reinterpret_cast<char*>(workspace);
  const size_t slab_bytes = g.slab_floats * sizeof(float);
  // activations / gradients (padding columns of d_out stay zero), flags and granules
  regions[0] = ZeroRegion{base + slab_bytes, (4 * g.act_floats + g.dout_floats) * sizeof(float)};
  regions[1] = ZeroRegion{base + mdnn_data_bytes(g), mdnn_sync_bytes()};
  return BSIG_OK;
}

int persist_mdnn_run(const PersistMdnnShape& s, const PersistMdnnBuffers& b,
                     const PersistHyper& hy, int n, hipStream_t st) {
  MdnnGeom g;
  BSIG_REQUIRE(mdnn_geom(s, &g), "persistent MDNN updates: shape not covered");
  BSIG_REQUIRE(b.x && b.y && b.ids && b.params && b.exp_avg && b.exp_avg_sq && b.state &&
                   b.train_loss && b.workspace, "persistent MDNN updates: null buffer");
  BSIG_REQUIRE(b.workspace_bytes >= persist_mdnn_workspace_bytes(s),
               "persistent MDNN updates: workspace too small");
  const bool fac = b.x_kind == BSIG_X_CROSSCORR_FACTORS;
  BSIG_REQUIRE(b.x_kind == BSIG_X_ROWS || fac, "persistent MDNN updates: unknown x_kind");
  BSIG_REQUIRE(!fac || (b.x_s >= 1 && b.x_a >= 1 && (int64_t)b.x_s * b.x_a + 2 == s.input_dim &&
                        b.ldx >= b.x_s + b.x_a + 3),
               "persistent MDNN updates: factor rows do not match the first layer (S=%d A=%d I=%d)",
               b.x_s, b.x_a, s.input_dim);
  BSIG_REQUIRE(fac || (b.ldx % 4 == 0 && b.ldx >= s.input_dim && aligned(b.x, 16)),
               "persistent MDNN updates: summaries must be 16-byte aligned rows");
  BSIG_REQUIRE(b.w2_off % 2 == 0 && b.wh_off % 2 == 0 && aligned(b.params, 16),
               "persistent MDNN updates: weight blocks must be 8-byte aligned");
  BSIG_REQUIRE(!(b.adam_pending && !b.grads), "persistent MDNN updates: pending Adam step without gradients");
  BSIG_REQUIRE(!(b.grads && n > 1 && !b.xr_ready), "persistent MDNN updates: data-parallel launches take one update");
  BSIG_REQUIRE(!(b.xr_ready && !(b.grads && b.xr_done && !b.adam_pending && !g.stream && b.do_eval && b.n_total == n && n >= 1)),
               "persistent MDNN updates: a resident data-parallel launch takes the whole call, evaluations inside");
  if (n <= 0 && !b.adam_pending && !b.do_eval) return BSIG_OK;
  int s_nip = 0, s_pf = 0;
  size_t s_lds = 0;
  if (g.stream) {
    BSIG_REQUIRE(fac, "persistent MDNN updates: a streamed first layer takes cross-correlation factor rows");
    BSIG_REQUIRE(mdnn_stream_fits(g, b.x_s, b.x_a, &s_nip, &s_pf, &s_lds),
                 "persistent MDNN updates: factor rows S=%d A=%d not covered by the streamed first layer",
                 b.x_s, b.x_a);
    BSIG_REQUIRE(!b.adam_pending && n >= 1 && !(b.do_eval && b.grads),
                 "persistent MDNN updates: a streamed first layer takes its Adam step outside");
  }
  // the > 64 KB dynamic-LDS attribute is per device (the plan's device is the current one:
  // the Python mirror enters the model's device around every call)
  static bool attr_set_dev[64] = {};
  int attr_dev = 0;
  BSIG_HIP(hipGetDevice(&attr_dev));
  bool& attr_set = attr_set_dev[attr_dev & 63];
  if (!attr_set) {
    const void* kernels[16] = {
#define BSIG_K(a, b, c, d) reinterpret_cast<const void*>(mdnn_updates_kernel<a, b, c, d>)
        BSIG_K(false, false, false, false), BSIG_K(true, false, false, false), BSIG_K(false, true, false, false),
        BSIG_K(true, true, false, false),   BSIG_K(false, false, true, false), BSIG_K(true, false, true, false),
        BSIG_K(false, true, true, false),   BSIG_K(true, true, true, false),   BSIG_K(false, false, false, true),
        BSIG_K(true, false, false, true),   BSIG_K(false, true, false, true),  BSIG_K(true, true, false, true),
        BSIG_K(false, false, true, true),   BSIG_K(true, false, true, true),   BSIG_K(false, true, true, true),
        BSIG_K(true, true, true, true)};
#undef BSIG_K
    for (const void* k : kernels)
      BSIG_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, kMLdsLimit));
    attr_set = true;
  }
  MdnnArgs p{};
  p.B = s.batch; p.FR = g.FR; p.I = s.input_dim; p.Nh = g.Nh; p.Nh16 = g.Nh16; p.NhP = g.NhP;
  p.D = s.out_dim; p.K = s.n_comp;
  p.k_slices = g.k_slices; p.G1 = g.G1; p.n_owner = g.n_owner; p.n_small = g.n_small;
  p.n_updates = std::max(n, 0); p.x_floats = g.x_floats;
  p.grads = b.grads; p.adam_pending = b.adam_pending;
  p.xr_ready = b.xr_ready; p.xr_done = b.xr_done; p.xr_base = b.xr_base;
  p.x = b.x; p.ldx = b.ldx; p.ids = b.ids; p.y = b.y; p.ldy = b.ldy;
  p.x_fac = fac ? 1 : 0; p.xS = b.x_s; p.xA = b.x_a;
  p.params = b.params; p.m1 = b.exp_avg; p.m2 = b.exp_avg_sq;
  p.w1_off = b.w1_off; p.b1_off = b.b1_off; p.w2_off = b.w2_off; p.b2_off = b.b2_off;
  p.pair_ok = (p.w1_off % 2 == 0 && s.input_dim % 2 == 0 &&
               ((reinterpret_cast<uintptr_t>(b.params) | reinterpret_cast<uintptr_t>(b.exp_avg) |
                 reinterpret_cast<uintptr_t>(b.exp_avg_sq) | reinterpret_cast<uintptr_t>(b.grads)) & 7) == 0) ? 1 : 0;
  p.wh_off = b.wh_off; p.bh_off = b.bh_off;
  {
    const char* e = getenv("BSIG_MDNN_FAST_ROWS");     // (read per launch -- one launch per call: A/B runs, tests)
    p.fast_rows = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1);     // (2: also for wide first layers)
  }
  p.state = b.state; p.train_loss = b.train_loss;
  p.lr = hy.lr; p.beta1 = hy.beta1; p.beta2 = hy.beta2;
  p.adam_eps = hy.adam_eps; p.eps_noise = hy.eps_noise; p.min_w = hy.min_weight;
  p.ll_limit = hy.ll_limit; p.inv_norm = 1.0f / (float)hy.norm_batch;
  char* base = reinterpret_cast<char*>(b.workspace);
  p.slabs = reinterpret_cast<float*>(base);
  p.dz1 = p.slabs + g.slab_floats;
  p.h1 = p.dz1 + g.act_floats;
  p.h2 = p.h1 + g.act_floats;
  p.dz2 = p.h2 + g.act_floats;
  p.d_out = p.dz2 + g.act_floats;
  p.w2f_pack = p.d_out + g.dout_floats;
  p.w2b_pack = p.w2f_pack + kPackFloats;
  p.eval_slabs = p.w2b_pack + kPackFloats;
  p.eval_out = p.eval_slabs + (size_t)2 * g.eval_passes * g.eval_slab_floats;
  p.wide = g.wide; p.n_hb = g.NhP / kMNB;
  p.h2e = p.eval_out + (size_t)g.eval_passes * s.batch * g.NhP;     // (wide heads with evaluations only)
  p.oe = p.h2e + (g.wide && g.eval_passes > 0 ? g.act_floats : 0);
  p.o_wide = p.oe + (g.wide && g.eval_passes > 0 ? g.dout_floats : 0);
  p.dz2_part = p.o_wide + g.dout_floats;
  p.hpre = g.wide ? p.dz2_part + (size_t)(g.NhP / kMNB) * g.act_floats : p.o_wide;   // (o_wide / dz2_part: wide plans only)
  p.stream = g.stream; p.s_chunks = g.s_chunks; p.s_nip = s_nip; p.s_pf = s_pf;
  char* sync = base + mdnn_data_bytes(g);
  p.flag_fwd = reinterpret_cast<unsigned*>(sync);
  p.flag_own = p.flag_fwd + kFlagArr;
  p.flag_small = p.flag_own + kFlagArr;
  p.flag_pack = p.flag_small + kFlagArr;
  p.flag_eval = p.flag_pack + kFlagArr;
  p.flag_h2 = p.flag_eval + kFlagArr;
  p.flag_o = p.flag_h2 + kFlagArr;
  p.flag_dout = p.flag_o + kFlagArr;
  p.flag_dz2 = p.flag_dout + kFlagArr;
  static unsigned launch_tag = 0;
  p.launch_tag = ++launch_tag;
  p.flag_red = p.flag_dz2 + kFlagArr;
  p.o_slabs = g.stream ? p.hpre : p.slabs; p.o_k_slices = g.stream ? 1 : g.k_slices;
  p.o_flags = g.stream ? p.flag_red : p.flag_fwd;
  p.flag_evp = p.flag_red + kFlagArr;     // streamed W1, evaluation passes: slab out / quads summed
  p.flag_evr = p.flag_evp + kFlagArr;
  p.flag_h2e = p.flag_evr + kFlagArr;
  p.flag_oe = p.flag_h2e + kFlagArr;
  p.gran = reinterpret_cast<unsigned long long*>(sync + 14 * kFlagArr * sizeof(unsigned));
  p.xr_count = reinterpret_cast<unsigned*>(sync + mdnn_sync_bytes() - 256);
  p.gran_eval = p.gran + 3 * kGranArr;
  if (b.do_eval) {
    BSIG_REQUIRE(g.eval_passes > 0 && b.n_test >= 1 && b.n_test <= g.eval_passes * s.batch &&
                     b.y_test && b.test_loss && b.eval_every >= 1 && b.n_total >= 1 &&
                     (g.stream || (b.x_test && b.ldx_test % 4 == 0 && b.ldx_test >= s.input_dim &&
                                   aligned(b.x_test, 16))),
                 "persistent MDNN updates: in-launch evaluation not covered");
    // (streamed W1: the held-out pairs' FACTOR rows, which lie behind the training rows in the
    // bound block -- bsig_fit_buffers.x_kind)
    BSIG_REQUIRE(!g.stream || (b.x_test_fac && b.ldx_test_fac >= b.x_s + b.x_a + 3),
                 "persistent MDNN updates: a streamed first layer evaluates from the held-out pairs' factor rows");
    p.xe = b.x_test_fac; p.ldxe = b.ldx_test_fac;
    p.do_eval = 1; p.eval_every = b.eval_every; p.n_total = b.n_total; p.n_test = b.n_test;
    p.eval_passes = ceil_div(b.n_test, s.batch);
    p.x_test = b.x_test; p.ldx_test = b.ldx_test; p.y_test = b.y_test; p.ldy_test = b.ldy_test;
    p.test_loss = b.test_loss;
  }
  p.prof = reinterpret_cast<long long*>(persist_profile_buffer());
  const dim3 grid(g.G1 + g.n_owner + g.n_small);
#define BSIG_MDNN_LAUNCH(DP_, FAC_, WIDE_, FULL_, MR_) \
  hipLaunchKernelGGL((mdnn_updates_kernel<DP_, FAC_, WIDE_, FULL_, MR_>), grid, dim3(kMT), g.lds, st, p)
#define BSIG_MDNN_LAUNCH_WF(DP_, FAC_)                                      \
  do {                                                                       \
    if (g.wide && full) BSIG_MDNN_LAUNCH(DP_, FAC_, true, true, kMR);        \
    else if (g.wide && g.mr == 1) BSIG_MDNN_LAUNCH(DP_, FAC_, true, false, 1);   \
    else if (g.wide && g.mr == 2) BSIG_MDNN_LAUNCH(DP_, FAC_, true, false, 2);   \
    else if (g.wide) BSIG_MDNN_LAUNCH(DP_, FAC_, true, false, kMR);          \
    else if (full) BSIG_MDNN_LAUNCH(DP_, FAC_, false, true, kMR);            \
    else if (g.mr == 1) BSIG_MDNN_LAUNCH(DP_, FAC_, false, false, 1);        \
    else if (g.mr == 2) BSIG_MDNN_LAUNCH(DP_, FAC_, false, false, 2);        \
    else BSIG_MDNN_LAUNCH(DP_, FAC_, false, false, kMR);                     \
  } while (0)
  const bool dp = b.grads != nullptr && !b.xr_ready, full = s.full_cov != 0;      // (resident: the single-rank instantiations)
  if (g.stream) return mdnn_stream_launch(p, dp, g.wide != 0, full, (int)grid.x, std::max(g.lds, s_lds), st);
  if (dp && fac) BSIG_MDNN_LAUNCH_WF(true, true);
  else if (dp) BSIG_MDNN_LAUNCH_WF(true, false);
  else if (fac) BSIG_MDNN_LAUNCH_WF(false, true);
  else BSIG_MDNN_LAUNCH_WF(false, false);
#undef BSIG_MDNN_LAUNCH_WF
#undef BSIG_MDNN_LAUNCH
  BSIG_CHECK_LAUNCH("mdnn_updates");
  return BSIG_OK;
}

}  // namespace bsig
