/*
 * bsig.h — C ABI of libbsig_hip.so: the MI355X (gfx950) implementation of the
 * BayesSim posterior-estimator training path of NVlabs/bayes-sim-ig.
 *
 * The reference has no FFI for this path: the boundary is Python duck typing
 * inside bayes_sim_ig/bayes_sim.py (it builds MDNN / MDRFF and a summary_*
 * function by name, bayes_sim.py:56,82, and calls run_training / predict_MoGs,
 * bayes_sim.py:108-113,134).  Each entry point below replaces the PyTorch op
 * sequence at the cited reference lines; the Python mirror in
 * bayes_sim_ig_amd/ binds them with ctypes (INTEGRATION.md shows the stub a
 * reference maintainer would add).
 *
 * Conventions
 *   - all pointers are DEVICE pointers to fp32 / int32 unless stated;
 *   - `ld*` are leading dimensions in ELEMENTS (row pitch), >= the row width;
 *   - every call is asynchronous on `stream` (a hipStream_t), never allocates,
 *     never synchronises unless stated, and returns BSIG_OK or a negative
 *     code; `bsig_last_error()` gives the thread-local message;
 *   - row-major everywhere.
 */
#ifndef BSIG_H
#define BSIG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* bsig_stream_t; /* hipStream_t */

#define BSIG_OK 0
#define BSIG_EINVAL (-1)       /* bad argument (the reference would assert) */
#define BSIG_ELAUNCH (-2)      /* HIP launch / runtime failure */
#define BSIG_EUNSUPPORTED (-3) /* shape outside what the kernels cover */
#define BSIG_ENONFINITE (-4)   /* device-side isfinite flag was raised */

const char* bsig_last_error(void);
int bsig_version(void);
/* What the library was compiled against, so that a binding can compare it with its own view of
 * this header before the first call: which = 0 -> FNV-1a (64-bit) hash of this file's text as the
 * build saw it; 1, 2, 3 -> sizeof(bsig_head_dims), sizeof(bsig_mdn_cfg), sizeof(bsig_fit_buffers);
 * 4 -> offsetof(bsig_fit_buffers, x_kind); anything else 0.  (No reference counterpart: the
 * reference has no native boundary.) */
uint64_t bsig_abi_info(int which);
/* Number of devices visible to the library, or a negative code. */
int bsig_device_count(void);

/* ------------------------------------------------------------------ */
/* Trajectory summarizers  (bayes_sim_ig/utils/summarizers.py)         */
/* states [n, t_states, sd], actions [n, t_actions, ad], contiguous.   */
/* out [n, ld_out]; only the first F columns of each row are written.  */
/* ------------------------------------------------------------------ */

/* Width F of the summary (no device work). kind: 0 start/waypts(max_t=10),
 * 1 corr, 2 corrdiff, 3 signature with depth (0 = reference rule). */
int64_t bsig_summary_dim(int kind, int traj_len, int sd, int ad, int depth);

/* summary_start / summary_waypts, summarizers.py:65-87 (+ the crop/pad of
 * :20-62; every trajectory repeats its own last step when shorter). */
int bsig_summary_start(const float* states, const float* actions, float* out,
                       int64_t n, int t_states, int t_actions, int sd, int ad,
                       int max_t, int64_t ld_out, bsig_stream_t stream);

/* cross_correlation, summarizers.py:90-122 (summary_corr: use_state_diff=0,
 * summary_corrdiff: 1).  `nonfinite` (int32, may be NULL) is OR-ed with 1 if
 * any feature is non-finite (the reference asserts, :120). */
int bsig_crosscorr(const float* states, const float* actions, float* out,
                   int64_t n, int t_states, int t_actions, int sd, int ad,
                   int use_state_diff, int64_t ld_out, int32_t* nonfinite,
                   bsig_stream_t stream);

/* The cross-correlation summary in FACTORED form (no summary
 * materialisation).  summarizers.py:106-119 builds out[i*A + j] = sf[i] * af[j]
 * (S = W(sd-1) state features, A = W*ad action features; 47 KB per Ant row, 420 KB per
 * ShadowHand row) plus mean(sf), std(sf).  A factor row holds what that is made of:
 *   [ sf[0..S) | af[0..A) | mean | std | 1.0 | 0... ]   (ld_factors >= S + A + 3)
 * and the fit engine's first layer forms the products itself, one fp32 multiply each,
 * exactly as the summarizer would (bsig_fit_buffers.x_kind = BSIG_X_CROSSCORR_FACTORS).
 * bsig_crosscorr_expand materialises the summary rows from factor rows. */
int bsig_crosscorr_factor_dims(int traj_len, int sd, int ad, int32_t* s_out, int32_t* a_out);
int bsig_crosscorr_factors(const float* states, const float* actions, float* factors,
                           int64_t n, int t_states, int t_actions, int sd, int ad,
                           int use_state_diff, int64_t ld_factors, int32_t* nonfinite,
                           bsig_stream_t stream);
int bsig_crosscorr_expand(const float* factors, int64_t ld_factors, int64_t n, int s_dim,
                          int a_dim, float* out, int64_t ld_out, bsig_stream_t stream);

/* summary_signatory, summarizers.py:144-168: truncated signature (levels
 * 1..depth, signatory layout) of the time-augmented path [l+1 | s_l | a_l].
 * depth 1..3; depth 3 needs 1+sd+ad <= 32, depth 2 needs <= 160. */
int bsig_signature(const float* states, const float* actions, float* out,
                   int64_t n, int length, int sd, int ad, int depth,
                   int64_t ld_out, bsig_stream_t stream);

/* ------------------------------------------------------------------ */
/* fp32 MFMA GEMM with fused epilogues (v_mfma_f32_32x32x2_f32)        */
/* C[m,n] = epi( sum_k A(m,k) * B(n,k) )                               */
/*   a_kmajor = 0: A(m,k) = A[arow(m)*lda + k]   (k contiguous)        */
/*   a_kmajor = 1: A(m,k) = A[arow(k)*lda + m]   (m contiguous)        */
/*   arow(i) = a_rows ? a_rows[i] : i   (row gather, mdnn.py:222)      */
/*   likewise for B.                                                   */
/* ------------------------------------------------------------------ */
enum {
  BSIG_EPI_NONE = 0,
  BSIG_EPI_BIAS = 1,      /* + bias[n]                                  */
  BSIG_EPI_BIAS_ACT = 2,  /* act(acc + bias[n])   nn.Linear + activation */
  BSIG_EPI_COS_SIN = 3,   /* C[m,n]=alpha*cos(acc), C[m,N+n]=alpha*sin(acc)
                             (rff.py:128-132); ldc >= 2N                 */
  BSIG_EPI_COS_OFF = 4,   /* alpha*cos(acc + bias[n])   (rff.py:122-126) */
  BSIG_EPI_MUL_DACT = 5   /* acc * act'(aux[m*ldaux+n]) with aux = act output */
};
enum { BSIG_ACT_TANH = 0, BSIG_ACT_RELU = 1, BSIG_ACT_LEAKY_RELU = 2,
       BSIG_ACT_SIGMOID = 3, BSIG_ACT_IDENTITY = 4 };

size_t bsig_gemm_workspace_bytes(int64_t m, int64_t n, int64_t k);

int bsig_gemm_f32(const float* a, int64_t lda, int a_kmajor, const int32_t* a_rows,
                  const float* b, int64_t ldb, int b_kmajor, const int32_t* b_rows,
                  float* c, int64_t ldc, int64_t m, int64_t n, int64_t k,
                  int epilogue, int act, const float* bias, const float* aux,
                  int64_t ldaux, float alpha, void* workspace, size_t workspace_bytes,
                  bsig_stream_t stream);

/* RFF projection, rff.py:128-132 / :122-126: feats = a*[cos|sin](x coeff^T)
 * with coeff = freqs / sigma [m_feat, in_dim] precomputed by bsig_rff_coeff.
 * x_rows gathers minibatch rows (may be NULL). cos_only uses `offset[m_feat]`. */
int bsig_rff_coeff(const float* freqs, const float* sigma, float* coeff,
                   int64_t m_feat, int64_t in_dim, int64_t ld_coeff,
                   bsig_stream_t stream);
int bsig_rff_project(const float* x, int64_t ldx, const int32_t* x_rows,
                     const float* coeff, int64_t ld_coeff, const float* offset,
                     float* feats, int64_t ld_feats, int64_t batch, int64_t in_dim,
                     int64_t m_feat, float a, int cos_only, void* workspace,
                     size_t workspace_bytes, bsig_stream_t stream);

/* ------------------------------------------------------------------ */
/* Mixture-density head  (bayes_sim_ig/models/mdnn.py:108-178)         */
/* head_out [B, ld] = [logits K | mu D*K | pre_diag D*K | lower Ls*K], */
/* index d*K+k inside each block (mdnn.py:112-119).                    */
/* ------------------------------------------------------------------ */
typedef struct bsig_head_dims {
  int32_t out_dim;   /* D */
  int32_t n_comp;    /* K */
  int32_t full_cov;  /* 1: Ls = D(D-1)/2 strict-lower entries per component */
  float eps_noise;   /* MDNN.EPS_NOISE  (mdnn.py:24)  */
  float min_weight;  /* MDNN.MIN_WEIGHT (mdnn.py:23)  */
  float ll_limit;    /* MDNN.LL_LIMIT   (mdnn.py:22)  */
} bsig_head_dims;

int64_t bsig_head_width(const bsig_head_dims* dims);          /* Nh */
size_t bsig_head_workspace_bytes(const bsig_head_dims* dims, int64_t batch);

/* forward() tuple from raw head outputs, mdnn.py:109-119.  `noise` [B,D,K]
 * is the injected rand_like draw (NULL: Philox noise from seed/stream_id). */
int bsig_mdn_head_outputs(const bsig_head_dims* dims, const float* head_out,
                          int64_t ld, int64_t batch, const float* noise,
                          uint64_t seed, uint64_t stream_id, float* weights,
                          float* mu, float* l_d, float* lower, int32_t* nonfinite,
                          void* workspace, size_t workspace_bytes,
                          bsig_stream_t stream);

/* mdn_loss_fn(weights, mu, L_d, L, y), mdnn.py:127-178 -> loss[0]. */
int bsig_mdn_nll_from_tuple(const bsig_head_dims* dims, const float* weights,
                            const float* mu, const float* l_d, const float* lower,
                            const float* y, int64_t ldy, int64_t batch, float* loss,
                            int32_t* nonfinite, void* workspace,
                            size_t workspace_bytes, bsig_stream_t stream);

/* Fused forward()+mdn_loss_fn (+ autograd backward when d_head_out != NULL)
 * on raw head outputs: loss[0] = mean NLL over `batch` rows; d_head_out
 * [B, ld] = d(sum_b nll_b / norm_batch)/d head_out (norm_batch = batch for
 * one rank, the global batch under data parallelism).  y rows may be gathered
 * by y_rows. */
int bsig_mdn_head_nll(const bsig_head_dims* dims, const float* head_out, int64_t ld,
                      const float* y, int64_t ldy, const int32_t* y_rows,
                      int64_t batch, int64_t norm_batch, const float* noise,
                      uint64_t seed, uint64_t stream_id, float* loss,
                      float* d_head_out, int32_t* nonfinite, void* workspace,
                      size_t workspace_bytes, bsig_stream_t stream);

/* ------------------------------------------------------------------ */
/* Flat-buffer helpers                                                 */
/* ------------------------------------------------------------------ */
/* torch.optim.Adam step (mdnn.py:203,234), t = 1-based step number. */
int bsig_adam_flat(float* params, const float* grads, float* exp_avg,
                   float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                   float eps, int64_t t, bsig_stream_t stream);
/* out[j] = sum_i x[i*ld + j]  (bias gradients). */
int bsig_colsum(const float* x, int64_t ld, int64_t rows, int64_t cols, float* out,
                void* workspace, size_t workspace_bytes, bsig_stream_t stream);
/* normalize_samples, mdnn.py:245-248: out = (theta - lows) / (highs - lows). */
int bsig_normalize_rows(const float* theta, int64_t ld_in, const float* lows,
                        const float* highs, float* out, int64_t ld_out, int64_t rows,
                        int64_t cols, bsig_stream_t stream);
/* dst[i, :cols] = src[rows ? rows[i] : i, :cols]  (strided copy / gather). */
int bsig_copy_rows(const float* src, int64_t ld_src, const int32_t* rows, float* dst,
                   int64_t ld_dst, int64_t n_rows, int64_t cols, bsig_stream_t stream);

/* ------------------------------------------------------------------ */
/* Estimator + fit engine (MDNN / MDRFF construct, forward, run_training) */
/* ------------------------------------------------------------------ */
#define BSIG_MAX_HIDDEN 8

typedef struct bsig_mdn_cfg {
  int32_t input_dim;               /* width of a summary row (I)            */
  int32_t n_hidden;                /* trunk layers; 0 for MDRFF             */
  int32_t hidden[BSIG_MAX_HIDDEN]; /* mdnn.py:68-75                         */
  int32_t activation;              /* BSIG_ACT_*                            */
  int32_t rff_feats;               /* 0: MDNN; else n_feat (mdrff.py:14-24) */
  int32_t rff_cos_only;
  float rff_scale;                 /* a = sqrt(2/n_feat)  (rff.py:107)      */
  bsig_head_dims head;
  float lr, beta1, beta2, adam_eps; /* torch.optim.Adam defaults, mdnn.py:203 */
} bsig_mdn_cfg;

/* Flat parameter layout, state_dict order: net.fcon{l}.weight, .bias ...,
 * then the head block [pi | mu | Diag.0 | Lower] weights (contiguous rows:
 * one [Nh, F] matrix), then the head biases [Nh].  offsets[2*i], [2*i+1] =
 * start of weight/bias i (trunk layers first, then pi, mu, Diag.0, Lower). */
int64_t bsig_mdn_param_count(const bsig_mdn_cfg* cfg);
int bsig_mdn_param_offsets(const bsig_mdn_cfg* cfg, int64_t* offsets, int n_offsets);
size_t bsig_mdn_workspace_bytes(const bsig_mdn_cfg* cfg, int64_t max_batch);

/* head_out[B, Nh] = heads(trunk(x)) (or heads(rff(x))): mdnn.py:108-119
 * without the softmax/exp (apply bsig_mdn_head_outputs / _head_nll). */
int bsig_mdn_head_forward(const bsig_mdn_cfg* cfg, const float* params,
                          const float* rff_coeff, int64_t ld_coeff,
                          const float* rff_offset, const float* x, int64_t ldx,
                          const int32_t* x_rows, int64_t batch, float* head_out,
                          int64_t ld_head, void* workspace, size_t workspace_bytes,
                          bsig_stream_t stream);

/* One forward + NLL + backward over a minibatch: flat `grads` (same layout
 * as params; fully overwritten) and loss[0].  mdnn.py:229-233. */
int bsig_mdn_loss_grad(const bsig_mdn_cfg* cfg, const float* params,
                       const float* rff_coeff, int64_t ld_coeff,
                       const float* rff_offset, const float* x, int64_t ldx,
                       const float* y, int64_t ldy, const int32_t* rows,
                       int64_t batch, int64_t norm_batch, const float* noise,
                       uint64_t seed, uint64_t stream_id, float* grads, float* loss,
                       int32_t* nonfinite, void* workspace, size_t workspace_bytes,
                       bsig_stream_t stream);

/* Fit engine: MDNN.run_training's loop (mdnn.py:228-242) for one chunk,
 * replayed from HIP graphs with no host synchronisation.  All buffers are
 * caller-owned device memory and must stay valid while the plan is bound. */
typedef struct bsig_fit_buffers {
  float* params; float* grads; float* exp_avg; float* exp_avg_sq; /* [P]   */
  const float* rff_coeff; int64_t ld_coeff; const float* rff_offset;
  const float* x_train; int64_t ldx_train; int64_t n_train;  /* summaries  */
  const float* y_train; int64_t ldy_train;          /* normalised theta    */
  const float* x_test; int64_t ldx_test; int64_t n_test;
  const float* y_test; int64_t ldy_test;
  const int32_t* ids_table;      /* [n_updates, batch] minibatch row ids   */
  float* train_loss;             /* [n_updates]  loss of every update      */
  float* test_loss;              /* [n_evals]                              */
  int32_t* state;                /* [16] int32 engine state (reset by begin); word 2 = non-finite flag */
  void* workspace; size_t workspace_bytes;
  /* what the x_train rows hold: the summary itself (BSIG_X_ROWS), or the factor rows of a
   * cross-correlation summary (bsig_crosscorr_factors; cfg.input_dim = x_s*x_a + 2,
   * ldx_train >= x_s + x_a + 3) -- accepted by plans for which bsig_fit_accepts_factors()
   * != 0.  The training rows are the ones the updates gather over and over (12.5 visits per
   * row and call); the held-out x_test rows, read once per evaluation, are always summary
   * rows (bsig_crosscorr_expand of their factor rows). */
  int32_t x_kind; int32_t x_s; int32_t x_a;
  /* optional, with x_kind == BSIG_X_CROSSCORR_FACTORS: the factor rows of the n_test held-out
   * pairs (same layout as the x_train rows).  A plan whose first layer is streamed
   * (bsig_fit_accepts_factor_rows) then takes its evaluations inside the launch of the run,
   * from these rows; without them (NULL) it runs them between its launches on x_test. */
  const float* x_test_factors; int64_t ldx_test_factors;
} bsig_fit_buffers;
#define BSIG_X_ROWS 0
#define BSIG_X_CROSSCORR_FACTORS 1

typedef struct bsig_fit_plan bsig_fit_plan;

/* n_updates: updates per run (sizes the hoisted-RFF feature block of MDRFF:
 * the projections of all n_updates*batch minibatch rows and of the held-out
 * rows of every evaluation are computed by bsig_fit_begin in one large GEMM). */
int bsig_fit_create(const bsig_mdn_cfg* cfg, int64_t batch, int64_t max_test_rows,
                    int64_t n_updates, bsig_fit_plan** plan);
/* The same with a bound on the training rows the plan will be bound to (0 = none):
 * MDRFF plans then size the hoisted feature block for min(n_updates*batch,
 * max_train_rows) rows -- a call that visits every row many times (large minibatches
 * over many epochs) keeps ONE feature row per distinct training row instead of
 * projecting inside every update. */
int bsig_fit_create_sized(const bsig_mdn_cfg* cfg, int64_t batch, int64_t max_train_rows,
                          int64_t max_test_rows, int64_t n_updates, bsig_fit_plan** plan);
/* The same with per-plan options.  BSIG_PLAN_NO_PERSISTENT: this plan's updates run on the per-phase
 * kernels whatever its shape (a model that met a persistent-launch time-out on a shared GPU stays on
 * them) -- an argument, not the process-wide BSIG_NO_PERSISTENT environment switch, which other
 * threads' plans would see. */
#define BSIG_PLAN_NO_PERSISTENT 1
int bsig_fit_create_ex(const bsig_mdn_cfg* cfg, int64_t batch, int64_t max_train_rows,
                       int64_t max_test_rows, int64_t n_updates, int plan_flags, bsig_fit_plan** plan);
void bsig_fit_destroy(bsig_fit_plan* plan);
size_t bsig_fit_workspace_bytes(const bsig_fit_plan* plan);
/* Bind buffers (re-captures graphs only if something changed).  flags:
 * BSIG_FIT_GRAPH replays the update from HIP graphs; BSIG_FIT_SPLIT_ADAM keeps
 * gradient and optimizer as separate phases (bsig_fit_grad / bsig_fit_apply,
 * needed for the data-parallel gradient exchange) instead of fusing Adam into
 * the weight-gradient GEMM epilogues. */
#define BSIG_FIT_GRAPH 1
#define BSIG_FIT_SPLIT_ADAM 2
int bsig_fit_bind(bsig_fit_plan* plan, const bsig_fit_buffers* buffers, int flags);
/* MDRFF: hand the RFF features of this call's rows over instead of having bsig_fit_begin
 * project them (rff.py:128-132 is a pure function of the row): `feats` [rows, ld_feats],
 * rows = n_train + n_test in the order of x_train / x_test.  A caller that fits many chunks
 * (BayesSim.fit) projects all their rows in one large GEMM.  Valid for the next
 * bsig_fit_begin only. */
int bsig_fit_set_features(bsig_fit_plan* plan, const float* feats, int64_t ld_feats,
                          int64_t rows, bsig_stream_t stream);
/* != 0: bound to n_train training rows the plan keeps one feature row per row (MDRFF whose
 * call visits its rows more than once): after bsig_fit_set_features nothing reads x_train /
 * x_test -- the caller need not stage the summaries. */
int bsig_fit_takes_features(const bsig_fit_plan* plan, int64_t n_train);
/* Reset step counter / Adam state (fresh optimizer per call, mdnn.py:203). */
int bsig_fit_begin(bsig_fit_plan* plan, uint64_t seed, int64_t norm_batch,
                   bsig_stream_t stream);
/* n_updates SGD updates with a held-out evaluation every max(n_updates/5,1)
 * updates and after the last (mdnn.py:235-242). */
int bsig_fit_run(bsig_fit_plan* plan, int64_t n_updates, bsig_stream_t stream);
/* n_updates consecutive SGD updates (mdnn.py:219-233) without evaluation: ONE
 * launch of the persistent update kernel when the plan is covered by it (linear
 * heads on cached RFF features, diagonal covariance, single rank), else
 * n_updates replays of the update graph.  bsig_fit_run = runs of these between
 * the held-out evaluations. */
int bsig_fit_updates(bsig_fit_plan* plan, int64_t n_updates, bsig_stream_t stream);
/* The call's logs for ONE read-back: out[2*E + 1] (device) = train_loss at the E
 * logging points of mdnn.py:235-242 | test_loss[E] | the state block's flag word
 * (bit 0 non-finite, bit 1 poll time-out) as a float. */
int bsig_fit_pack_logs(bsig_fit_plan* plan, int64_t n_updates, float* out,
                       bsig_stream_t stream);
/* Data-parallel pieces (BSIG_FIT_SPLIT_ADAM; mdnn.py:229-233 with the exchange
 * the reference does not have between loss.backward() and optimizer.step()):
 * bsig_fit_grad = forward + NLL + backward of one minibatch into `grads`
 * (all-reduce `grads` outside), bsig_fit_apply = the Adam step on `grads`.
 * Plans covered by the persistent update kernel run an update as ONE launch:
 * bsig_fit_apply then only marks the step as pending and the next bsig_fit_grad
 * takes it while it loads its weight tiles; bsig_fit_eval takes a pending step
 * first, bsig_fit_flush takes it at once (call it before reading `params`). */
int bsig_fit_grad(bsig_fit_plan* plan, bsig_stream_t stream);
int bsig_fit_apply(bsig_fit_plan* plan, bsig_stream_t stream);
int bsig_fit_flush(bsig_fit_plan* plan, bsig_stream_t stream);
/* != 0: the plan's training rows may be cross-correlation factor rows (all its updates run
 * in the persistent kernel of the two-layer MDNN, whose first-layer tiles form the products). */
int bsig_fit_accepts_factors(const bsig_fit_plan* plan);
/* ... for S x A factor rows (x_kind BSIG_X_CROSSCORR_FACTORS with x_s = S, x_a = A).  A plan whose
 * first layer does not fit the chip's LDS and registers (cfg/anymal.yaml:103-109, I = 56402;
 * cfg/shadow_hand_more.yaml:73-81, I = 105002) STREAMS it through the tile workgroups of the
 * persistent kernel and takes factor rows only (A % 4 == 0, the factor rows of two minibatches in
 * one workgroup's LDS); bsig_fit_accepts_factors answers 0 for such a plan, this call decides.
 * Summary rows of that width run the per-phase kernels.  (summarizers.py:112-119 into
 * mdnn.py:71,108.) */
int bsig_fit_accepts_factor_rows(const bsig_fit_plan* plan, int s_dim, int a_dim);
/* 1 if a call bound with these factor rows (S, A) and bind flags evaluates its held-out pairs from
 * their FACTOR rows inside the launch (streamed first layer, single rank): nothing then reads
 * held-out summary rows -- bsig_fit_buffers.x_test may be NULL, and the caller need not expand the
 * held-out fifth of the chunk (84 MB per 1000-pair chunk of cfg/shadow_hand_more.yaml). */
int bsig_fit_evaluates_from_factors(const bsig_fit_plan* plan, int s_dim, int a_dim, int bind_flags);
/* 1: the plan's updates run in the persistent kernel for linear heads on cached
 * features (MDRFF), 2: in the one for the two-layer MDNN trunk (as bound), 0: as
 * per-phase kernels (diagnostics / tests). */
int bsig_fit_is_persistent(const bsig_fit_plan* plan);
int bsig_fit_eval(bsig_fit_plan* plan, bsig_stream_t stream);

/* ------------------------------------------------------------------ */
/* Data-parallel exchange (no counterpart in the reference, which is   */
/* single-device; the site is between loss.backward() and              */
/* optimizer.step(), mdnn.py:233-234): every rank holds a replica and  */
/* a shard of the pairs; the flat fp32 gradient buffer is summed over  */
/* the ranks once per update, RCCL over xGMI, on the fit's stream.     */
/* ------------------------------------------------------------------ */
typedef struct bsig_comm bsig_comm;
#define BSIG_COMM_ID_BYTES 128
/* Rank 0: fill `id_out` (HOST, BSIG_COMM_ID_BYTES) with a fresh rendezvous id
 * (ncclGetUniqueId); the caller hands the bytes to every rank by its own means. */
int bsig_comm_unique_id(void* id_out);
/* Collective over all `world` ranks: join the communicator `unique_id` names as
 * `rank`, on HIP device `device` (ncclCommInitRank).  BSIG_EUNSUPPORTED when no
 * RCCL can be loaded. */
int bsig_comm_init(const void* unique_id, int world, int rank, int device,
                   bsig_comm** comm);
/* A communicator over a caller-supplied exchange instead of RCCL (another
 * transport, or tests of the multi-rank path on one GPU, where RCCL refuses two
 * ranks per device).  `exchange(ctx, op, buf, n, root, stream)` must leave the
 * sum over ranks (op BSIG_EXCHANGE_SUM) / root's values (BSIG_EXCHANGE_BROADCAST)
 * in the DEVICE buffer `buf[n]`, ordered after the work already enqueued on
 * `stream` and before anything enqueued later; it returns 0 on success. */
enum { BSIG_EXCHANGE_SUM = 0, BSIG_EXCHANGE_BROADCAST = 1 };
typedef int (*bsig_exchange_fn)(void* ctx, int op, float* buf, int64_t n, int root,
                                bsig_stream_t stream);
int bsig_comm_init_external(int world, int rank, bsig_exchange_fn exchange, void* ctx,
                            bsig_comm** comm);
/* 1: RCCL (its collectives can be captured into a HIP graph), 2: a caller-supplied exchange, 0: null */
int bsig_comm_transport(const bsig_comm* comm);
int bsig_comm_world(const bsig_comm* comm);
int bsig_comm_rank(const bsig_comm* comm);
/* (diagnostics) bsig_fit_run_dp calls this communicator ran with the rank RESIDENT across the exchange
 * (one launch per call, the all-reduces on a second stream: INTEGRATION.md, BSIG_DP_RESIDENT) */
int64_t bsig_comm_resident_calls(const bsig_comm* comm);
/* Per communicator, over the BSIG_DP_RESIDENT policy: 0 never resident (what the Python mirror sets
 * after a resident launch timed out: one launch per update from then on), 1 always where covered,
 * -1 back to the policy. */
void bsig_comm_set_resident(bsig_comm* comm, int mode);
int bsig_comm_resident_mode(const bsig_comm* comm);
/* buf[n] (device, fp32) <- sum over ranks, in place, asynchronous on `stream`. */
int bsig_comm_allreduce(bsig_comm* comm, float* buf, int64_t n, bsig_stream_t stream);
/* buf[n] <- rank `root`'s buf (replica initialisation). */
int bsig_comm_broadcast(bsig_comm* comm, float* buf, int64_t n, int root,
                        bsig_stream_t stream);
void bsig_comm_destroy(bsig_comm* comm);

/* run_training's loop for a data-parallel rank (plan bound with
 * BSIG_FIT_SPLIT_ADAM, after bsig_fit_begin with norm_batch = batch * world):
 * per update bsig_fit_grad -> bsig_comm_allreduce(grads) -> bsig_fit_apply, the
 * held-out evaluations at the reference's logging points (mdnn.py:235-242),
 * bsig_fit_flush at the end -- all enqueued from here, no host work in between.
 * `reduced_logs` (device, n_updates + n_evals + 3 floats, may be NULL) receives,
 * summed over the ranks: train_loss[n_updates] | test_loss[e] * n_test |
 * n_test | #ranks with the non-finite flag | #ranks with a poll time-out.
 * A rank whose plan runs in a chip-resident persistent kernel with its evaluations inside the
 * launch can make ONE launch for the whole call instead and stay resident across the exchange:
 * the all-reduces then run on a stream of the communicator's own (RCCL transport; BSIG_DP_RESIDENT,
 * default: 1-rank groups only -- INTEGRATION.md; same results bit for bit). */
int bsig_fit_run_dp(bsig_fit_plan* plan, bsig_comm* comm, int64_t n_updates,
                    float* reduced_logs, bsig_stream_t stream);
/* With BSIG_DP_GRAPH=1, bsig_fit_run_dp captures a steady-state update of a rank whose updates run
 * in the persistent kernel of the linear heads (launch + ncclAllReduce of the gradients) into ONE
 * HIP graph and replays it.  Returns 1: captured, 0: not tried yet, -1: the capture failed (the
 * runtime's message in msg; the direct calls are used), -2: not applicable (msg says why). */
int bsig_fit_dp_graph_status(const bsig_fit_plan* plan, char* msg, size_t msg_bytes);

/* Diagnostics: device buffer of [256][8][16] int64 wall-clock stamps filled by the
 * persistent update kernel (first 8 updates of every later launch); NULL = off.
 * tools/persist_prof.py prints the phase breakdown. */
void bsig_debug_persist_profile(void* device_buffer);
/* Tests: occupy `blocks` CUs for `ms` milliseconds with workgroups that hold `lds_bytes` of LDS
 * each, so that a persistent update launch behind it cannot get all its workgroups resident
 * (its bounded polls then time out: bit 1 of the state block's flag word; the Python mirror
 * restores the call's start state and repeats it on the per-phase kernels). */
int bsig_debug_spin(int blocks, size_t lds_bytes, int ms, bsig_stream_t stream);
/* Diagnostics / tests: how the persistent update kernel of the linear heads (MDRFF) would tile a
 * head of n_comp components over out_dim parameters on feat_dim features at this minibatch size --
 * host arithmetic only, no device is asked.  out[16] = { NT (16-row blocks per tile), k-slice
 * width, head blocks, k-slices, tile workgroups G, workgroups of the launch T, row owners, rows per
 * owner, evaluation owners, rows per evaluation owner, evaluation passes (0: evaluations outside
 * the launches), LDS bytes per workgroup, owners that also hold a tile, the kernel a launch of this
 * shape would run on THIS process's device (2: this tiling, 1: the round-1..3 kernel
 * fit_persistent_v1.hip -- the tiling above then does not describe it --, 0: per-phase kernels or
 * no device), 0, 0 }.  Returns 1 if the shape is covered, 0 if not. */
int bsig_debug_persist_geometry(int batch, int feat_dim, int out_dim, int n_comp, int max_test,
                                int32_t* out);
/* ... and how the persistent update kernel of the two-layer MDNN (trunk [128, 128], tanh; the reference's
 * default, models/mdnn.py:68-75) would lay out its workgroups for input_dim summary columns:
 * out[16] = { first-layer k-slices of 256 columns, first-layer tile workgroups, row-owner workgroups,
 * minibatch rows per owner (1, 2 or 4; 8 with a streamed first layer), small-weight / head-block
 * workgroups, wide heads (head outputs formed by the head-block workgroups), first layer streamed,
 * evaluation passes (0: evaluations outside the launches), LDS bytes per workgroup (a streamed plan:
 * without its tile workgroups, whose need depends on the factor dimensions), head width Nh, 0... }.
 * Returns 1 if the shape is covered, 0 if not. */
int bsig_debug_persist_mdnn_geometry(int batch, int input_dim, int out_dim, int n_comp, int full_cov,
                                     int max_test, int32_t* out);

/* Tests: the property the MDNN kernel's fma-chain head outputs rest on -- a 16x16x4 fp32 MFMA adds its four
 * products as a chain of fused multiply-adds in ascending k.  D = A [16][k] B [k][16] (device pointers, k a
 * multiple of 4) by a chain of MFMAs and by fmaf chains; *mismatches (device, caller-zeroed) += the number
 * of the 256 outputs whose bits differ. */
int bsig_debug_mfma_vs_fma(const float* a, const float* b, int k, int32_t* mismatches, bsig_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BSIG_H */
