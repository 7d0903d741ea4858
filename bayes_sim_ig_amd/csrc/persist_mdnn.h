// Persistent update kernel of the fit engine for the MDNN estimator with the
// reference's default trunk (two hidden layers of 128 tanh units, diagonal
// covariance; fit_persistent_mdnn.hip): a run of consecutive Adam updates
// (mdnn.py:219-233) in ONE launch.
#pragma once
#include "common.h"

namespace bsig {

struct PersistMdnnShape {
  int batch, input_dim, h1, h2, activation, out_dim, n_comp, full_cov;
  int max_test = 0;   // held-out rows the plan may evaluate inside the launches (0: none)
};

struct PersistMdnnBuffers {
  const float* x; int64_t ldx;            // summary rows (training split of the chunk), or
  int x_kind = 0, x_s = 0, x_a = 0;       // ... cross-correlation factor rows (bsig.h: the tile
                                          // workgroups form x[i*A + j] = sf[i] * af[j] themselves)
  const int32_t* ids;                     // [n_updates*batch] minibatch row ids (rows of x and y)
  const float* y; int64_t ldy;            // normalised targets
  float* params; float* exp_avg; float* exp_avg_sq;   // flat buffers
  int64_t w1_off, b1_off, w2_off, b2_off, wh_off, bh_off;
  int32_t* state;                         // the fit engine's 16-word state block
  float* train_loss;                      // [n_updates]
  void* workspace; size_t workspace_bytes;
  // data-parallel ranks (null / 0 otherwise): the gradients of the update go to `grads`
  // (flat layout) for the caller's all-reduce instead of into Adam; with `adam_pending` the
  // launch first takes the Adam step of the previous update from the (reduced) `grads`.
  // n = 0 with adam_pending: that step only.
  float* grads = nullptr; int adam_pending = 0;
  // ... or RESIDENT across the exchange, as persist.h: PersistBuffers::xr_* (plans whose first layer is
  // resident on the chip and whose evaluations run inside the launch; ONE launch for the whole call)
  unsigned* xr_ready = nullptr; const unsigned* xr_done = nullptr; unsigned xr_base = 0;
  // held-out evaluations inside the launch (persist_mdnn_eval_supported): after update `it`
  // of the call with it % eval_every == 0 and after the last of its n_total updates
  // (mdnn.py:235-242); evaluation k writes test_loss[state[1]] and advances state[1]
  int do_eval = 0; int eval_every = 1; int n_total = 0; int n_test = 0;
  // streamed first layer: the held-out pairs' factor rows (in-launch evaluation)
  const float* x_test_fac = nullptr; int64_t ldx_test_fac = 0;
  const float* x_test = nullptr; int64_t ldx_test = 0;
  const float* y_test = nullptr; int64_t ldy_test = 0;
  float* test_loss = nullptr;
};

struct PersistHyper;   // persist.h

bool persist_mdnn_supported(const PersistMdnnShape& s);
// (diagnostics) the decomposition fit_persistent_mdnn.hip plans for a shape, bsig.h: bsig_debug_persist_mdnn_geometry
int persist_mdnn_geometry(const PersistMdnnShape& s, int32_t* out);
// ... with the first layer STREAMED by the tile workgroups (it does not fit the chip): such a plan
// takes cross-correlation factor rows only; the Adam step of a data-parallel rank runs outside the
// launches (flat kernel after the all-reduce); a single rank's held-out evaluations run INSIDE its
// one launch, from the held-out pairs' factor rows (bsig_fit_evaluates_from_factors)
int persist_mdnn_streams(const PersistMdnnShape& s);
// ... and S x A cross-correlation factor rows are covered (bsig.h: x_kind)
bool persist_mdnn_accepts_factors(const PersistMdnnShape& s, int S, int A);
// ... and the held-out evaluations of up to s.max_test rows can run inside the launches
bool persist_mdnn_eval_supported(const PersistMdnnShape& s);
bool persist_mdnn_dp_eval_supported(const PersistMdnnShape& s);
size_t persist_mdnn_workspace_bytes(const PersistMdnnShape& s);
struct ZeroRegion;   // persist.h
int persist_mdnn_reset_regions(const PersistMdnnShape& s, void* workspace, size_t workspace_bytes,
                               ZeroRegion* regions);
// n consecutive updates starting at the state block's step counter; advances the
// counter, the jitter RNG stream and the Adam bias-correction powers
int persist_mdnn_run(const PersistMdnnShape& s, const PersistMdnnBuffers& b,
                     const PersistHyper& h, int n, hipStream_t st);

}  // namespace bsig
