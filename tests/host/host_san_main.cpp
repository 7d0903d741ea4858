// Host-side logic of libbsig_hip under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build:
// every source compiled --cuda-host-only, tools/build_host_san.sh): argument checks, parameter
// layouts, GEMM planners, persistent-kernel geometry, plan binding, the external-exchange
// communicator.  No kernel is launched (there may be no GPU); every call must return its
// documented code and leave a message in bsig_last_error() on failure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/bsig.h"

static int g_fail = 0;
#define EXPECT(cond)                                                              \
  do {                                                                            \
    if (!(cond)) { std::printf("FAIL %s:%d: %s (last error: %s)\n", __FILE__, __LINE__, #cond, bsig_last_error()); ++g_fail; } \
  } while (0)

static bsig_mdn_cfg make_cfg(int input_dim, int n_hidden, int h0, int h1, int rff, int d, int k, int full) {
  bsig_mdn_cfg c;
  std::memset(&c, 0, sizeof(c));
  c.input_dim = input_dim; c.n_hidden = n_hidden; c.hidden[0] = h0; c.hidden[1] = h1;
  c.activation = BSIG_ACT_TANH; c.rff_feats = rff; c.rff_scale = 0.1f;
  c.head.out_dim = d; c.head.n_comp = k; c.head.full_cov = full;
  c.head.eps_noise = 1e-5f; c.head.min_weight = 1e-5f; c.head.ll_limit = 1e5f;
  c.lr = 1e-3f; c.beta1 = 0.9f; c.beta2 = 0.999f; c.adam_eps = 1e-8f;
  return c;
}

static int sum_exchange(void* ctx, int op, float* buf, int64_t n, int root, bsig_stream_t) {
  int* calls = static_cast<int*>(ctx);
  ++*calls;
  if (op == BSIG_EXCHANGE_SUM) for (int64_t i = 0; i < n; ++i) buf[i] *= 2.f;   // "two ranks with equal data"
  (void)root;
  return 0;
}
static int failing_exchange(void*, int, float*, int64_t, int, bsig_stream_t) { return 7; }

int main() {
  EXPECT(bsig_version() >= 100);
  EXPECT(bsig_abi_info(1) == sizeof(bsig_head_dims) && bsig_abi_info(2) == sizeof(bsig_mdn_cfg) &&
         bsig_abi_info(3) == sizeof(bsig_fit_buffers) && bsig_abi_info(99) == 0);

  // ---- parameter layout
  bsig_mdn_cfg mdnn = make_cfg(11802, 2, 128, 128, 0, 17, 5, 0);
  const int64_t total = bsig_mdn_param_count(&mdnn);
  EXPECT(total > (int64_t)11802 * 128);
  int64_t offs[16];
  EXPECT(bsig_mdn_param_offsets(&mdnn, offs, 16) == BSIG_OK && offs[0] == 0);
  EXPECT(bsig_mdn_param_offsets(&mdnn, offs, 3) == BSIG_EINVAL && std::strlen(bsig_last_error()) > 0);
  bsig_mdn_cfg bad = mdnn; bad.n_hidden = BSIG_MAX_HIDDEN + 1;
  EXPECT(bsig_mdn_param_count(&bad) == -1);
  bad = mdnn; bad.rff_feats = 512;                          // MDRFF has no trunk (mdrff.py:18)
  EXPECT(bsig_mdn_param_count(&bad) == -1);
  bad = make_cfg(100, 0, 0, 0, 511, 3, 4, 0);               // n_feat must be even (rff.py:105)
  EXPECT(bsig_mdn_param_count(&bad) == -1);
  EXPECT(bsig_mdn_param_count(nullptr) == -1);
  EXPECT(bsig_mdn_workspace_bytes(&mdnn, 100) > 0 && bsig_mdn_workspace_bytes(&bad, 100) == 0);
  EXPECT(bsig_head_width(&mdnn.head) == 5 * (1 + 2 * 17));
  bsig_head_dims fullh = mdnn.head; fullh.full_cov = 1; fullh.out_dim = 4; fullh.n_comp = 3;
  EXPECT(bsig_head_width(&fullh) == 3 * (1 + 2 * 4 + 6));

  // ---- GEMM planner over a spread of shapes (minibatch-sized, large, skinny, split-K)
  const int64_t shapes[][3] = {{100, 128, 11802}, {128, 11802, 100}, {8192, 260, 4096}, {260, 4096, 8192},
                               {100000, 2048, 2310}, {1, 1, 1}, {100, 650, 128}, {200, 128, 105002},
                               {128, 105002, 100}, {7, 3, 5}, {104, 270, 4096}};
  for (const auto& sh : shapes) EXPECT(bsig_gemm_workspace_bytes(sh[0], sh[1], sh[2]) < ((size_t)1 << 34));

  // ---- summary widths (summarizers.py:75-79, 106-119, 133-168)
  EXPECT(bsig_summary_dim(0, 11, 211, 20, 0) == 10 * 231);

  // ---- fit plans: creation, sizes, binding checks (no launches)
  struct Case { bsig_mdn_cfg cfg; int64_t batch, n_updates; };
  const Case cases[] = {{make_cfg(2310, 0, 0, 0, 4096, 32, 4, 0), 100, 100},
                        {make_cfg(11802, 2, 128, 128, 0, 17, 5, 0), 100, 100},
                        {make_cfg(105002, 2, 128, 128, 0, 32, 10, 0), 100, 100},   // streamed first layer
                        {make_cfg(56402, 2, 128, 128, 0, 13, 10, 0), 100, 100},
                        {make_cfg(40, 2, 24, 24, 0, 2, 10, 1), 7, 3},
                        {make_cfg(1, 2, 128, 128, 0, 2, 3, 1), 100, 500}};
  for (const Case& cs : cases) {
    bsig_fit_plan* plan = nullptr;
    EXPECT(bsig_fit_create_sized(&cs.cfg, cs.batch, 800, 200, cs.n_updates, &plan) == BSIG_OK && plan);
    if (!plan) continue;
    const size_t ws = bsig_fit_workspace_bytes(plan);
    EXPECT(ws > 0);
    (void)bsig_fit_is_persistent(plan);
    (void)bsig_fit_accepts_factors(plan);
    (void)bsig_fit_accepts_factor_rows(plan, 1050, 100);
    (void)bsig_fit_takes_features(plan, 800);
    bsig_fit_buffers fb;
    std::memset(&fb, 0, sizeof(fb));
    EXPECT(bsig_fit_bind(plan, &fb, 0) == BSIG_EINVAL);                     // null buffers
    EXPECT(bsig_fit_bind(plan, nullptr, 0) == BSIG_EINVAL);
    // plausible (host) addresses: binding only records and checks them
    std::vector<float> dummy(64);
    float* q = dummy.data();
    fb.params = fb.grads = fb.exp_avg = fb.exp_avg_sq = q;
    fb.x_train = fb.y_train = fb.x_test = fb.y_test = q;
    fb.rff_coeff = q; fb.ld_coeff = cs.cfg.input_dim;
    fb.ids_table = reinterpret_cast<const int32_t*>(q);
    fb.train_loss = fb.test_loss = q; fb.state = reinterpret_cast<int32_t*>(q);
    fb.workspace = q; fb.workspace_bytes = ws / 2;                           // too small
    fb.ldx_train = fb.ldx_test = (cs.cfg.input_dim + 3) / 4 * 4; fb.ldy_train = fb.ldy_test = 36;
    fb.n_train = 800; fb.n_test = 200;
    EXPECT(bsig_fit_bind(plan, &fb, 0) == BSIG_EINVAL);
    fb.workspace_bytes = ws;
    fb.n_test = 201;                                                         // more than the plan's 200
    EXPECT(bsig_fit_bind(plan, &fb, 0) == BSIG_EINVAL);
    fb.n_test = 200;
    fb.ldx_train = 3;                                                        // rows narrower than input_dim
    if (cs.cfg.input_dim > 3) EXPECT(bsig_fit_bind(plan, &fb, 0) == BSIG_EINVAL);
    fb.ldx_train = (cs.cfg.input_dim + 3) / 4 * 4;
    EXPECT(bsig_fit_bind(plan, &fb, 0) == BSIG_OK);
    fb.x_kind = 7;
    EXPECT(bsig_fit_bind(plan, &fb, 0) == BSIG_EINVAL);
    fb.x_kind = BSIG_X_CROSSCORR_FACTORS; fb.x_s = 3; fb.x_a = 5;            // 3*5+2 != input_dim
    EXPECT(bsig_fit_bind(plan, &fb, 0) != BSIG_OK);
    EXPECT(bsig_fit_begin(plan, 1, 0, nullptr) == BSIG_EINVAL);              // norm_batch >= 1
    EXPECT(bsig_fit_run(plan, cs.n_updates + 1, nullptr) != BSIG_OK);
    EXPECT(bsig_fit_updates(plan, -1, nullptr) == BSIG_EINVAL);
    EXPECT(bsig_fit_grad(plan, nullptr) == BSIG_EINVAL);                     // not bound with SPLIT_ADAM
    bsig_fit_destroy(plan);
  }
  bsig_fit_plan* none = nullptr;
  EXPECT(bsig_fit_create_sized(nullptr, 100, 0, 0, 1, &none) == BSIG_EINVAL);
  EXPECT(bsig_fit_create(&mdnn, 0, 0, 1, &none) == BSIG_EINVAL);
  bsig_fit_destroy(nullptr);
  EXPECT(bsig_fit_workspace_bytes(nullptr) == 0 && bsig_fit_is_persistent(nullptr) == 0);

  // ---- communicator behind a caller-supplied exchange (bsig_comm_init_external)
  bsig_comm* comm = nullptr;
  int calls = 0;
  EXPECT(bsig_comm_init_external(2, 0, sum_exchange, &calls, &comm) == BSIG_OK && comm);
  EXPECT(bsig_comm_world(comm) == 2 && bsig_comm_rank(comm) == 0);
  std::vector<float> g(1000, 1.5f);
  EXPECT(bsig_comm_allreduce(comm, g.data(), (int64_t)g.size(), nullptr) == BSIG_OK && g[999] == 3.0f && calls == 1);
  EXPECT(bsig_comm_allreduce(comm, g.data(), 0, nullptr) == BSIG_OK && calls == 1);   // empty: no exchange
  EXPECT(bsig_comm_broadcast(comm, g.data(), 10, 1, nullptr) == BSIG_OK && calls == 2);
  EXPECT(bsig_comm_broadcast(comm, g.data(), 10, 2, nullptr) == BSIG_EINVAL);         // root outside the group
  EXPECT(bsig_comm_allreduce(comm, nullptr, 10, nullptr) == BSIG_EINVAL);
  EXPECT(bsig_comm_allreduce(nullptr, g.data(), 10, nullptr) == BSIG_EINVAL);
  bsig_comm_destroy(comm);
  comm = nullptr;
  EXPECT(bsig_comm_init_external(2, 0, failing_exchange, nullptr, &comm) == BSIG_OK);
  EXPECT(bsig_comm_allreduce(comm, g.data(), 4, nullptr) == BSIG_ELAUNCH && std::strstr(bsig_last_error(), "7"));
  bsig_comm_destroy(comm);
  EXPECT(bsig_comm_init_external(2, 2, sum_exchange, nullptr, &comm) == BSIG_EINVAL);   // rank outside the group
  EXPECT(bsig_comm_init_external(1, 0, nullptr, nullptr, &comm) == BSIG_EINVAL);
  EXPECT(bsig_comm_init(nullptr, 1, 0, 0, &comm) == BSIG_EINVAL);
  bsig_comm_destroy(nullptr);
  EXPECT(bsig_comm_world(nullptr) == 0 && bsig_comm_rank(nullptr) == -1);

  std::printf(g_fail ? "host sanitizer test: %d failure(s)\n" : "host sanitizer test: ok\n", g_fail);
  return g_fail ? 1 : 0;
}
