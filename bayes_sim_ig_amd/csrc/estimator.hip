// MDNN / MDRFF estimator passes and the fit engine.
//   head forward   : mdnn.py:108-119 (trunk + heads) / mdrff.py:28-30 (RFF first)
//   loss + grad    : mdnn.py:229-233 (forward, mdn_loss_fn, backward)
//   fit engine     : mdnn.py:228-242 — the whole update (minibatch gather,
//                    forward, NLL, backward, Adam) is captured once in a HIP
//                    graph and replayed per step with no host synchronisation;
//                    per-step values (minibatch ids, Adam bias corrections,
//                    loss slot, RNG stream) live in a small device state block
//                    advanced by the graph's first kernel.
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

namespace bsig {

// implemented in the other translation units
int gemm_f32(const float* a, int64_t lda, int a_kmajor, const int32_t* a_rows,
             const float* b, int64_t ldb, int b_kmajor, const int32_t* b_rows, float* c,
             int64_t ldc, int64_t m, int64_t n, int64_t k, int epilogue, int act,
             const float* bias, const float* aux, int64_t ldaux, float alpha,
             void* workspace, size_t workspace_bytes, hipStream_t st);
int mdn_head_nll_launch(const bsig_head_dims* dims, const float* seg_w, int64_t ld_w,
                        const float* seg_mu, int64_t ld_mu, const float* seg_sg,
                        int64_t ld_sg, const float* seg_lo, int64_t ld_lo, int from_tuple,
                        const float* y, int64_t ldy, const int32_t* y_rows, int64_t batch,
                        int64_t norm_batch, const float* noise, uint64_t seed,
                        uint64_t stream_id, const uint64_t* dyn_rng, float* loss,
                        const int32_t* loss_slot, float* d_out, int64_t ld_dout,
                        int32_t* nonfinite, void* workspace, size_t workspace_bytes,
                        hipStream_t st);
int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr,
                float beta1, float beta2, float eps, int64_t t, const float* dyn,
                hipStream_t st);
int colsum_launch(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                  void* workspace, size_t workspace_bytes, hipStream_t st);

constexpr int64_t kAlign = 4;  // floats: every tensor starts 16-B aligned

struct Layout {
  int n_layers;                        // trunk layers
  int64_t in_dim[BSIG_MAX_HIDDEN + 1]; // input width of trunk layer l / of the heads
  int64_t w_off[BSIG_MAX_HIDDEN], b_off[BSIG_MAX_HIDDEN];
  int64_t feat_dim;                    // F: width of the heads' input
  int64_t nh;                          // Nh
  int64_t head_w_off, head_b_off;
  int64_t total;
};

static int make_layout(const bsig_mdn_cfg* c, Layout* L) {
  BSIG_REQUIRE(c, "cfg is null");
  BSIG_REQUIRE(c->input_dim >= 1, "cfg: input_dim must be >= 1");
  BSIG_REQUIRE(c->n_hidden >= 0 && c->n_hidden <= BSIG_MAX_HIDDEN, "cfg: bad n_hidden");
  BSIG_REQUIRE(!(c->rff_feats > 0 && c->n_hidden > 0), "cfg: MDRFF has no trunk (mdrff.py:18)");
  BSIG_REQUIRE(c->rff_feats >= 0 && (c->rff_cos_only || c->rff_feats % 2 == 0),
               "cfg: n_feat must be even (rff.py:105)");
  L->n_layers = c->n_hidden;
  int64_t off = 0, width = c->rff_feats > 0 ? c->rff_feats : c->input_dim;
  for (int l = 0; l < c->n_hidden; ++l) {
    BSIG_REQUIRE(c->hidden[l] >= 1, "cfg: hidden layer width must be >= 1");
    L->in_dim[l] = width;
    L->w_off[l] = off; off = round_up<int64_t>(off + (int64_t)c->hidden[l] * width, kAlign);
    L->b_off[l] = off; off = round_up<int64_t>(off + c->hidden[l], kAlign);
    width = c->hidden[l];
  }
  L->in_dim[c->n_hidden] = width;
  L->feat_dim = width;
  L->nh = bsig_head_width(&c->head);
  BSIG_REQUIRE(L->nh >= 1, "cfg: bad head dims");
  L->head_w_off = off; off = round_up<int64_t>(off + L->nh * width, kAlign);
  L->head_b_off = off; off = round_up<int64_t>(off + L->nh, kAlign);
  L->total = off;
  return BSIG_OK;
}

// ---- workspace carve ------------------------------------------------------
struct Scratch {
  float* feat;                      // [B, F] RFF features (MDRFF)
  float* h[BSIG_MAX_HIDDEN];        // trunk activations [B, hidden_l]
  float* dz[2];                     // ping-pong [B, max hidden]
  float* o;                         // [B, Nh] raw head outputs
  float* d_o;                       // [B, Nh]
  float* head_ws; size_t head_ws_bytes;
  float* gemm_ws; size_t gemm_ws_bytes;
  float* colsum_ws; size_t colsum_ws_bytes;
  size_t total_bytes;
};

static size_t gemm_ws_need(const bsig_mdn_cfg* c, const Layout& L, int64_t B) {
  size_t need = 0;
  auto upd = [&](int64_t m, int64_t n, int64_t k) {
    need = std::max(need, bsig_gemm_workspace_bytes(m, n, k));
  };
  if (c->rff_feats > 0) upd(B, c->rff_cos_only ? c->rff_feats : c->rff_feats / 2, c->input_dim);
  for (int l = 0; l < L.n_layers; ++l) {
    upd(B, c->hidden[l], L.in_dim[l]);       // forward
    upd(c->hidden[l], L.in_dim[l], B);       // dW
    if (l > 0) upd(B, L.in_dim[l], c->hidden[l]);  // dX
  }
  upd(B, L.nh, L.feat_dim);
  upd(L.nh, L.feat_dim, B);
  if (L.n_layers > 0) upd(B, L.feat_dim, L.nh);
  return need;
}

static void carve(const bsig_mdn_cfg* c, const Layout& L, int64_t B, void* base, Scratch* s) {
  size_t off = 0;
  auto take = [&](size_t floats) {
    float* p = base ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off) : nullptr;
    off += round_up<size_t>(floats * sizeof(float), 256);
    return p;
  };
  s->feat = c->rff_feats > 0 ? take((size_t)B * c->rff_feats) : nullptr;
  int64_t hmax = 0;
  for (int l = 0; l < L.n_layers; ++l) {
    s->h[l] = take((size_t)B * c->hidden[l]);
    hmax = std::max<int64_t>(hmax, c->hidden[l]);
  }
  s->dz[0] = hmax ? take((size_t)B * hmax) : nullptr;
  s->dz[1] = hmax ? take((size_t)B * hmax) : nullptr;
  s->o = take((size_t)B * L.nh);
  s->d_o = take((size_t)B * L.nh);
  s->head_ws_bytes = bsig_head_workspace_bytes(&c->head, B);
  s->head_ws = take(s->head_ws_bytes / sizeof(float) + 1);
  s->gemm_ws_bytes = gemm_ws_need(c, L, B);
  s->gemm_ws = take(s->gemm_ws_bytes / sizeof(float) + 1);
  s->colsum_ws_bytes = (size_t)64 * std::max<int64_t>(L.nh, hmax) * sizeof(float);
  s->colsum_ws = take(s->colsum_ws_bytes / sizeof(float));
  s->total_bytes = off;
}

// ---- passes ---------------------------------------------------------------
struct Inputs {
  const float* x; int64_t ldx; const int32_t* rows;
  const float* rff_coeff; int64_t ld_coeff; const float* rff_offset;
};

// trunk (or RFF) + heads -> s.o ; leaves activations in s.h / s.feat
static int forward_pass(const bsig_mdn_cfg* c, const Layout& L, const float* params,
                        const Inputs& in, int64_t B, const Scratch& s, float* o, int64_t ldo,
                        hipStream_t st) {
  const float* feat = in.x; int64_t ldf = in.ldx; const int32_t* frows = in.rows;
  if (c->rff_feats > 0) {
    BSIG_REQUIRE(in.rff_coeff, "MDRFF needs rff_coeff");
    const int64_t mf = c->rff_cos_only ? c->rff_feats : c->rff_feats / 2;
    BSIG_REQUIRE(!(c->rff_cos_only && !in.rff_offset), "cos-only RFF needs an offset");
    BSIG_TRY(gemm_f32(in.x, in.ldx, 0, in.rows, in.rff_coeff, in.ld_coeff, 0, nullptr, s.feat,
                      c->rff_feats, B, mf, c->input_dim,
                      c->rff_cos_only ? BSIG_EPI_COS_OFF : BSIG_EPI_COS_SIN, 0, in.rff_offset,
                      nullptr, 0, c->rff_scale, s.gemm_ws, s.gemm_ws_bytes, st));
    feat = s.feat; ldf = c->rff_feats; frows = nullptr;
  }
  for (int l = 0; l < L.n_layers; ++l) {
    BSIG_TRY(gemm_f32(feat, ldf, 0, frows, params + L.w_off[l], L.in_dim[l], 0, nullptr, s.h[l],
                      c->hidden[l], B, c->hidden[l], L.in_dim[l], BSIG_EPI_BIAS_ACT,
                      c->activation, params + L.b_off[l], nullptr, 0, 1.f, s.gemm_ws,
                      s.gemm_ws_bytes, st));
    feat = s.h[l]; ldf = c->hidden[l]; frows = nullptr;
  }
  BSIG_TRY(gemm_f32(feat, ldf, 0, frows, params + L.head_w_off, L.feat_dim, 0, nullptr, o, ldo,
                    B, L.nh, L.feat_dim, BSIG_EPI_BIAS, 0, params + L.head_b_off, nullptr, 0,
                    1.f, s.gemm_ws, s.gemm_ws_bytes, st));
  return BSIG_OK;
}

// backward from s.d_o into the flat gradient buffer
static int backward_pass(const bsig_mdn_cfg* c, const Layout& L, const float* params,
                         const Inputs& in, int64_t B, const Scratch& s, float* grads,
                         hipStream_t st) {
  // input of the heads
  const float* feat; int64_t ldf; const int32_t* frows = nullptr;
  if (L.n_layers > 0) { feat = s.h[L.n_layers - 1]; ldf = c->hidden[L.n_layers - 1]; }
  else if (c->rff_feats > 0) { feat = s.feat; ldf = c->rff_feats; }
  else { feat = in.x; ldf = in.ldx; frows = in.rows; }
  // dW_heads[Nh, F] = dO^T feat ; db = colsum(dO)
  BSIG_TRY(gemm_f32(s.d_o, L.nh, 1, nullptr, feat, ldf, 1, frows, grads + L.head_w_off,
                    L.feat_dim, L.nh, L.feat_dim, B, BSIG_EPI_NONE, 0, nullptr, nullptr, 0, 1.f,
                    s.gemm_ws, s.gemm_ws_bytes, st));
  BSIG_TRY(colsum_launch(s.d_o, L.nh, B, L.nh, grads + L.head_b_off, s.colsum_ws,
                         s.colsum_ws_bytes, st));
  if (L.n_layers == 0) return BSIG_OK;
  // dz_L = (dO W_heads) * act'(h_L)
  int cur = 0;
  BSIG_TRY(gemm_f32(s.d_o, L.nh, 0, nullptr, params + L.head_w_off, L.feat_dim, 1, nullptr,
                    s.dz[cur], L.feat_dim, B, L.feat_dim, L.nh, BSIG_EPI_MUL_DACT,
                    c->activation, nullptr, feat, ldf, 1.f, s.gemm_ws, s.gemm_ws_bytes, st));
  for (int l = L.n_layers - 1; l >= 0; --l) {
    const int64_t hw = c->hidden[l];
    const float* xin; int64_t ldin; const int32_t* rin = nullptr;
    if (l > 0) { xin = s.h[l - 1]; ldin = c->hidden[l - 1]; }
    else { xin = in.x; ldin = in.ldx; rin = in.rows; }
    BSIG_TRY(gemm_f32(s.dz[cur], hw, 1, nullptr, xin, ldin, 1, rin, grads + L.w_off[l],
                      L.in_dim[l], hw, L.in_dim[l], B, BSIG_EPI_NONE, 0, nullptr, nullptr, 0,
                      1.f, s.gemm_ws, s.gemm_ws_bytes, st));
    BSIG_TRY(colsum_launch(s.dz[cur], hw, B, hw, grads + L.b_off[l], s.colsum_ws,
                           s.colsum_ws_bytes, st));
    if (l > 0) {
      BSIG_TRY(gemm_f32(s.dz[cur], hw, 0, nullptr, params + L.w_off[l], L.in_dim[l], 1, nullptr,
                        s.dz[cur ^ 1], L.in_dim[l], B, L.in_dim[l], hw, BSIG_EPI_MUL_DACT,
                        c->activation, nullptr, xin, ldin, 1.f, s.gemm_ws, s.gemm_ws_bytes, st));
      cur ^= 1;
    }
  }
  return BSIG_OK;
}

static int head_nll(const bsig_mdn_cfg* c, const Layout& L, const Scratch& s, const float* y,
                    int64_t ldy, const int32_t* y_rows, int64_t B, int64_t norm_batch,
                    const float* noise, uint64_t seed, uint64_t stream_id,
                    const uint64_t* dyn_rng, float* loss, const int32_t* loss_slot, bool bwd,
                    int32_t* nonfinite, hipStream_t st) {
  const int64_t D = c->head.out_dim, K = c->head.n_comp;
  return mdn_head_nll_launch(&c->head, s.o, L.nh, s.o + K, L.nh, s.o + K + D * K, L.nh,
                             c->head.full_cov ? s.o + K + 2 * D * K : nullptr, L.nh, 0, y, ldy,
                             y_rows, B, norm_batch, noise, seed, stream_id, dyn_rng, loss,
                             loss_slot, bwd ? s.d_o : nullptr, L.nh, nonfinite, s.head_ws,
                             s.head_ws_bytes, st);
}

// ---- fit engine -----------------------------------------------------------
// device state block (int32 words)
enum { ST_STEP = 0, ST_EVAL = 1, ST_NONFINITE = 2, ST_CUR_STEP = 3, ST_ADAM0 = 4, ST_ADAM1 = 5,
       ST_CUR_EVAL = 6, ST_RNG = 8 /* 4 words: seed, counter (uint64 x2) */, ST_WORDS = 16 };

__global__ void fit_begin_kernel(int32_t* state, uint64_t seed) {
  if (threadIdx.x < ST_WORDS) state[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) reinterpret_cast<uint64_t*>(state + ST_RNG)[0] = seed;
}

// First kernel of every update: publish this step's values, advance counters,
// copy the minibatch ids (mdnn.py:219-222: ids drawn on the host in the
// reference's numpy-RNG order, uploaded once per chunk as a table).
__global__ __launch_bounds__(256) void step_begin_kernel(int32_t* __restrict__ state,
                                                         const int32_t* __restrict__ ids_table,
                                                         int32_t* __restrict__ cur_ids,
                                                         int batch, double beta1, double beta2,
                                                         double lr) {
  const int step = state[ST_STEP];
  for (int i = threadIdx.x; i < batch; i += blockDim.x)
    cur_ids[i] = ids_table[(int64_t)step * batch + i];
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (double)(step + 1);
    const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
    reinterpret_cast<float*>(state)[ST_ADAM0] = (float)(lr / bc1);
    reinterpret_cast<float*>(state)[ST_ADAM1] = (float)(1.0 / sqrt(bc2));
    state[ST_CUR_STEP] = step;
    state[ST_STEP] = step + 1;
    reinterpret_cast<uint64_t*>(state + ST_RNG)[1] += 1;
  }
}

__global__ void eval_begin_kernel(int32_t* state) {
  if (threadIdx.x == 0) {
    const int e = state[ST_EVAL];
    state[ST_CUR_EVAL] = e;
    state[ST_EVAL] = e + 1;
    reinterpret_cast<uint64_t*>(state + ST_RNG)[1] += 1;
  }
}

}  // namespace bsig

using namespace bsig;

struct bsig_fit_plan {
  bsig_mdn_cfg cfg;
  Layout L;
  int64_t batch, max_test;
  bsig_fit_buffers buf;
  bool bound;
  int64_t norm_batch;
  size_t train_ws_bytes, test_ws_bytes;
  bool use_graph;
  hipStream_t cap_stream;
  hipGraphExec_t g_step, g_grad, g_apply, g_eval;
};

namespace bsig {

static size_t plan_ws_bytes(const bsig_fit_plan* p) {
  // [cur_ids batch][train scratch][eval scratch]
  return round_up<size_t>((size_t)p->batch * sizeof(int32_t), 256) + p->train_ws_bytes +
         p->test_ws_bytes;
}

static void plan_scratch(const bsig_fit_plan* p, int32_t** cur_ids, Scratch* tr, Scratch* te) {
  char* base = reinterpret_cast<char*>(p->buf.workspace);
  *cur_ids = reinterpret_cast<int32_t*>(base);
  base += round_up<size_t>((size_t)p->batch * sizeof(int32_t), 256);
  carve(&p->cfg, p->L, p->batch, base, tr);
  base += p->train_ws_bytes;
  carve(&p->cfg, p->L, std::max<int64_t>(p->max_test, 1), base, te);
}

static int enqueue_grad(bsig_fit_plan* p, hipStream_t st) {
  int32_t* cur_ids; Scratch tr, te;
  plan_scratch(p, &cur_ids, &tr, &te);
  const bsig_fit_buffers& b = p->buf;
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(256), 0, st, b.state, b.ids_table,
                     cur_ids, (int)p->batch, (double)p->cfg.beta1, (double)p->cfg.beta2,
                     (double)p->cfg.lr);
  BSIG_CHECK_LAUNCH("step_begin");
  Inputs in{b.x_train, b.ldx_train, cur_ids, b.rff_coeff, b.ld_coeff, b.rff_offset};
  BSIG_TRY(forward_pass(&p->cfg, p->L, b.params, in, p->batch, tr, tr.o, p->L.nh, st));
  BSIG_TRY(head_nll(&p->cfg, p->L, tr, b.y_train, b.ldy_train, cur_ids, p->batch,
                    p->norm_batch, nullptr, 0, 0,
                    reinterpret_cast<const uint64_t*>(b.state + ST_RNG), b.train_loss,
                    b.state + ST_CUR_STEP, true, b.state + ST_NONFINITE, st));
  BSIG_TRY(backward_pass(&p->cfg, p->L, b.params, in, p->batch, tr, b.grads, st));
  return BSIG_OK;
}

static int enqueue_apply(bsig_fit_plan* p, hipStream_t st) {
  const bsig_fit_buffers& b = p->buf;
  return adam_launch(b.params, b.grads, b.exp_avg, b.exp_avg_sq, p->L.total, p->cfg.lr,
                     p->cfg.beta1, p->cfg.beta2, p->cfg.adam_eps, 1,
                     reinterpret_cast<const float*>(b.state) + ST_ADAM0, st);
}

static int enqueue_eval(bsig_fit_plan* p, hipStream_t st) {
  int32_t* cur_ids; Scratch tr, te;
  plan_scratch(p, &cur_ids, &tr, &te);
  const bsig_fit_buffers& b = p->buf;
  hipLaunchKernelGGL(eval_begin_kernel, dim3(1), dim3(64), 0, st, b.state);
  BSIG_CHECK_LAUNCH("eval_begin");
  if (b.n_test <= 0) return BSIG_OK;
  Inputs in{b.x_test, b.ldx_test, nullptr, b.rff_coeff, b.ld_coeff, b.rff_offset};
  BSIG_TRY(forward_pass(&p->cfg, p->L, b.params, in, b.n_test, te, te.o, p->L.nh, st));
  BSIG_TRY(head_nll(&p->cfg, p->L, te, b.y_test, b.ldy_test, nullptr, b.n_test, b.n_test,
                    nullptr, 0, 0, reinterpret_cast<const uint64_t*>(b.state + ST_RNG),
                    b.test_loss, b.state + ST_CUR_EVAL, false, b.state + ST_NONFINITE, st));
  return BSIG_OK;
}

static void drop_graphs(bsig_fit_plan* p) {
  hipGraphExec_t* gs[4] = {&p->g_step, &p->g_grad, &p->g_apply, &p->g_eval};
  for (auto g : gs)
    if (*g) { (void)hipGraphExecDestroy(*g); *g = nullptr; }
}

template <typename F>
static int capture(bsig_fit_plan* p, hipGraphExec_t* out, F&& body) {
  hipGraph_t graph = nullptr;
  BSIG_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed));
  const int rc = body(p->cap_stream);
  const hipError_t e = hipStreamEndCapture(p->cap_stream, &graph);
  if (rc != BSIG_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); return BSIG_ELAUNCH; }
  const hipError_t e2 = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e2 != hipSuccess) { set_error("hipGraphInstantiate: %s", hipGetErrorString(e2)); return BSIG_ELAUNCH; }
  return BSIG_OK;
}

static int ensure_graphs(bsig_fit_plan* p) {
  if (!p->use_graph || p->g_step) return BSIG_OK;
  BSIG_TRY(capture(p, &p->g_grad, [&](hipStream_t s) { return enqueue_grad(p, s); }));
  BSIG_TRY(capture(p, &p->g_apply, [&](hipStream_t s) { return enqueue_apply(p, s); }));
  BSIG_TRY(capture(p, &p->g_eval, [&](hipStream_t s) { return enqueue_eval(p, s); }));
  BSIG_TRY(capture(p, &p->g_step, [&](hipStream_t s) {
    BSIG_TRY(enqueue_grad(p, s));
    return enqueue_apply(p, s);
  }));
  return BSIG_OK;
}

}  // namespace bsig

extern "C" int64_t bsig_mdn_param_count(const bsig_mdn_cfg* cfg) {
  Layout L;
  if (make_layout(cfg, &L) != BSIG_OK) return -1;
  return L.total;
}

extern "C" int bsig_mdn_param_offsets(const bsig_mdn_cfg* cfg, int64_t* offsets, int n_offsets) {
  Layout L;
  BSIG_TRY(make_layout(cfg, &L));
  const int need = 2 * (L.n_layers + 4);
  BSIG_REQUIRE(offsets && n_offsets >= need, "param_offsets: need %d slots", need);
  int q = 0;
  for (int l = 0; l < L.n_layers; ++l) { offsets[q++] = L.w_off[l]; offsets[q++] = L.b_off[l]; }
  const int64_t D = cfg->head.out_dim, K = cfg->head.n_comp, F = L.feat_dim;
  const int64_t row0[4] = {0, K, K + D * K, K + 2 * D * K};
  for (int hsel = 0; hsel < 4; ++hsel) {
    offsets[q++] = L.head_w_off + row0[hsel] * F;
    offsets[q++] = L.head_b_off + row0[hsel];
  }
  return BSIG_OK;
}

extern "C" size_t bsig_mdn_workspace_bytes(const bsig_mdn_cfg* cfg, int64_t max_batch) {
  Layout L;
  if (make_layout(cfg, &L) != BSIG_OK) return 0;
  Scratch s;
  carve(cfg, L, std::max<int64_t>(max_batch, 1), nullptr, &s);
  return s.total_bytes;
}

extern "C" int bsig_mdn_head_forward(const bsig_mdn_cfg* cfg, const float* params,
                                     const float* rff_coeff, int64_t ld_coeff,
                                     const float* rff_offset, const float* x, int64_t ldx,
                                     const int32_t* x_rows, int64_t batch, float* head_out,
                                     int64_t ld_head, void* workspace, size_t workspace_bytes,
                                     bsig_stream_t stream) {
  Layout L;
  BSIG_TRY(make_layout(cfg, &L));
  BSIG_REQUIRE(params && x && head_out && batch >= 1, "head_forward: bad args");
  BSIG_REQUIRE(ldx >= cfg->input_dim && ld_head >= L.nh, "head_forward: leading dims too small");
  Scratch s;
  carve(cfg, L, batch, nullptr, &s);
  BSIG_REQUIRE(workspace && workspace_bytes >= s.total_bytes,
               "head_forward: workspace %zu < %zu", workspace_bytes, s.total_bytes);
  carve(cfg, L, batch, workspace, &s);
  Inputs in{x, ldx, x_rows, rff_coeff, ld_coeff, rff_offset};
  return forward_pass(cfg, L, params, in, batch, s, head_out, ld_head, as_stream(stream));
}

extern "C" int bsig_mdn_loss_grad(const bsig_mdn_cfg* cfg, const float* params,
                                  const float* rff_coeff, int64_t ld_coeff,
                                  const float* rff_offset, const float* x, int64_t ldx,
                                  const float* y, int64_t ldy, const int32_t* rows,
                                  int64_t batch, int64_t norm_batch, const float* noise,
                                  uint64_t seed, uint64_t stream_id, float* grads, float* loss,
                                  int32_t* nonfinite, void* workspace, size_t workspace_bytes,
                                  bsig_stream_t stream) {
  Layout L;
  BSIG_TRY(make_layout(cfg, &L));
  BSIG_REQUIRE(params && x && y && grads && batch >= 1 && norm_batch >= 1,
               "loss_grad: bad args");
  BSIG_REQUIRE(ldx >= cfg->input_dim && ldy >= cfg->head.out_dim, "loss_grad: leading dims");
  Scratch s;
  carve(cfg, L, batch, nullptr, &s);
  BSIG_REQUIRE(workspace && workspace_bytes >= s.total_bytes, "loss_grad: workspace %zu < %zu",
               workspace_bytes, s.total_bytes);
  carve(cfg, L, batch, workspace, &s);
  hipStream_t st = as_stream(stream);
  Inputs in{x, ldx, rows, rff_coeff, ld_coeff, rff_offset};
  BSIG_TRY(forward_pass(cfg, L, params, in, batch, s, s.o, L.nh, st));
  BSIG_TRY(head_nll(cfg, L, s, y, ldy, rows, batch, norm_batch, noise, seed, stream_id, nullptr,
                    loss, nullptr, true, nonfinite, st));
  return backward_pass(cfg, L, params, in, batch, s, grads, st);
}

extern "C" int bsig_fit_create(const bsig_mdn_cfg* cfg, int64_t batch, int64_t max_test_rows,
                               bsig_fit_plan** plan) {
  BSIG_REQUIRE(cfg && plan && batch >= 1 && max_test_rows >= 0, "fit_create: bad args");
  bsig_fit_plan* p = new (std::nothrow) bsig_fit_plan();
  BSIG_REQUIRE(p, "fit_create: out of memory");
  std::memset(p, 0, sizeof(*p));
  p->cfg = *cfg;
  const int rc = make_layout(cfg, &p->L);
  if (rc != BSIG_OK) { delete p; return rc; }
  p->batch = batch; p->max_test = max_test_rows; p->norm_batch = batch;
  Scratch s;
  carve(cfg, p->L, batch, nullptr, &s); p->train_ws_bytes = s.total_bytes;
  carve(cfg, p->L, std::max<int64_t>(max_test_rows, 1), nullptr, &s);
  p->test_ws_bytes = s.total_bytes;
  *plan = p;
  return BSIG_OK;
}

extern "C" void bsig_fit_destroy(bsig_fit_plan* p) {
  if (!p) return;
  drop_graphs(p);
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  delete p;
}

extern "C" size_t bsig_fit_workspace_bytes(const bsig_fit_plan* p) {
  return p ? plan_ws_bytes(p) : 0;
}

extern "C" int bsig_fit_bind(bsig_fit_plan* p, const bsig_fit_buffers* b, int use_graph) {
  BSIG_REQUIRE(p && b, "fit_bind: null");
  BSIG_REQUIRE(b->params && b->grads && b->exp_avg && b->exp_avg_sq && b->state &&
               b->workspace && b->x_train && b->y_train && b->ids_table && b->train_loss &&
               b->test_loss, "fit_bind: null buffer");
  BSIG_REQUIRE(b->workspace_bytes >= plan_ws_bytes(p), "fit_bind: workspace %zu < %zu",
               b->workspace_bytes, plan_ws_bytes(p));
  BSIG_REQUIRE(b->n_train >= 1 && b->n_test >= 0 && b->n_test <= std::max<int64_t>(p->max_test, 0),
               "fit_bind: n_train=%lld n_test=%lld (max %lld)", (long long)b->n_train,
               (long long)b->n_test, (long long)p->max_test);
  BSIG_REQUIRE(b->ldx_train >= p->cfg.input_dim && b->ldy_train >= p->cfg.head.out_dim,
               "fit_bind: leading dims too small");
  const bool same = p->bound && std::memcmp(&p->buf, b, sizeof(*b)) == 0 &&
                    p->use_graph == (use_graph != 0);
  if (!same) {
    drop_graphs(p);
    p->buf = *b;
    p->bound = true;
    p->use_graph = use_graph != 0;
  }
  if (p->use_graph && !p->cap_stream)
    BSIG_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
  return BSIG_OK;
}

extern "C" int bsig_fit_begin(bsig_fit_plan* p, uint64_t seed, int64_t norm_batch,
                              bsig_stream_t stream) {
  BSIG_REQUIRE(p && p->bound, "fit_begin: plan not bound");
  BSIG_REQUIRE(norm_batch >= 1, "fit_begin: norm_batch must be >= 1");
  if (norm_batch != p->norm_batch) { drop_graphs(p); p->norm_batch = norm_batch; }
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(fit_begin_kernel, dim3(1), dim3(64), 0, st, p->buf.state, seed);
  BSIG_CHECK_LAUNCH("fit_begin");
  // fresh optimizer state for every run_training call (mdnn.py:203)
  BSIG_HIP(hipMemsetAsync(p->buf.exp_avg, 0, (size_t)p->L.total * sizeof(float), st));
  BSIG_HIP(hipMemsetAsync(p->buf.exp_avg_sq, 0, (size_t)p->L.total * sizeof(float), st));
  return ensure_graphs(p);
}

extern "C" int bsig_fit_grad(bsig_fit_plan* p, bsig_stream_t stream) {
  BSIG_REQUIRE(p && p->bound, "fit_grad: plan not bound");
  if (p->use_graph) { BSIG_TRY(ensure_graphs(p)); BSIG_HIP(hipGraphLaunch(p->g_grad, as_stream(stream))); return BSIG_OK; }
  return enqueue_grad(p, as_stream(stream));
}

extern "C" int bsig_fit_apply(bsig_fit_plan* p, bsig_stream_t stream) {
  BSIG_REQUIRE(p && p->bound, "fit_apply: plan not bound");
  if (p->use_graph) { BSIG_TRY(ensure_graphs(p)); BSIG_HIP(hipGraphLaunch(p->g_apply, as_stream(stream))); return BSIG_OK; }
  return enqueue_apply(p, as_stream(stream));
}

extern "C" int bsig_fit_eval(bsig_fit_plan* p, bsig_stream_t stream) {
  BSIG_REQUIRE(p && p->bound, "fit_eval: plan not bound");
  if (p->use_graph) { BSIG_TRY(ensure_graphs(p)); BSIG_HIP(hipGraphLaunch(p->g_eval, as_stream(stream))); return BSIG_OK; }
  return enqueue_eval(p, as_stream(stream));
}

extern "C" int bsig_fit_run(bsig_fit_plan* p, int64_t n_updates, bsig_stream_t stream) {
  BSIG_REQUIRE(p && p->bound, "fit_run: plan not bound");
  BSIG_REQUIRE(n_updates >= 0, "fit_run: n_updates < 0");
  hipStream_t st = as_stream(stream);
  if (p->use_graph) BSIG_TRY(ensure_graphs(p));
  const int64_t every = std::max<int64_t>(n_updates / 5, 1);   // mdnn.py:235
  for (int64_t it = 0; it < n_updates; ++it) {
    if (p->use_graph) {
      BSIG_HIP(hipGraphLaunch(p->g_step, st));
    } else {
      BSIG_TRY(enqueue_grad(p, st));
      BSIG_TRY(enqueue_apply(p, st));
    }
    if (it % every == 0 || it + 1 == n_updates) {
      if (p->use_graph) BSIG_HIP(hipGraphLaunch(p->g_eval, st));
      else BSIG_TRY(enqueue_eval(p, st));
    }
  }
  return BSIG_OK;
}
