// Persistent update kernel of the fit engine (fit_persistent.hip): a run of
// consecutive Adam updates of a linear mixture-density head on precomputed
// features (MDRFF with the hoisted RFF projection) in ONE launch.
#pragma once
#include "common.h"

namespace bsig {

struct PersistShape {
  int batch, feat_dim, out_dim, n_comp;
  int max_test = 0;   // held-out rows the plan may evaluate inside the launches (0: none)
};

struct PersistBuffers {
  const float* feats; int64_t ld_feats;   // feature rows
  const int32_t* feat_ids;                // minibatch row i of update `step` = feats[feat_ids[step*batch+i]]
                                          // (null: feats[step*batch + i])
  const float* y; int64_t ldy;            // targets, gathered through ids
  const int32_t* ids;                     // [n_updates*batch] minibatch row ids
  float* params; float* exp_avg; float* exp_avg_sq;   // flat buffers
  int64_t w_off, b_off;                   // head weights [Nh, feat_dim] / bias [Nh] inside them
  int32_t* state;                         // the fit engine's 16-word state block
  float* train_loss;                      // [n_updates]
  void* workspace; size_t workspace_bytes;
  // data-parallel ranks (null / 0 otherwise): the gradients of the update go to
  // `grads` (flat layout) for the caller's all-reduce instead of into Adam; with
  // `adam_pending` the launch first takes the Adam step of the previous update
  // from the (reduced) `grads`.  n = 0 with adam_pending: that step only.
  float* grads = nullptr; int adam_pending = 0;
  // ... or RESIDENT across the exchange (fit_persistent.hip; persist_mdnn.h has the same three fields): ONE
  // launch for the whole call; after update u (1-based) the kernel writes its gradients through to `grads`,
  // raises *xr_ready to xr_base + u (a word a stream can wait on) and polls *xr_done until the caller's
  // exchange stream has written the same number behind its all-reduce.  The two words only ever grow
  // (xr_base: the updates of earlier calls): nothing has to reset them between calls, so no stream
  // operation of one call has to be ordered against the exchange of another.
  unsigned* xr_ready = nullptr; const unsigned* xr_done = nullptr; unsigned xr_base = 0;
  // held-out evaluations inside the launch (persist_eval_supported): after update `it` of the
  // call with it % eval_every == 0 and after the last of its n_total updates (mdnn.py:235-242);
  // evaluation k writes test_loss[state[1]] and advances state[1]
  int do_eval = 0; int eval_every = 1; int n_total = 0;
  int64_t eval_row0 = 0; int n_test = 0;   // held-out rows: feats[eval_row0 .. +n_test)
  const float* y_test = nullptr; int64_t ldy_test = 0;
  float* test_loss = nullptr;
};

struct PersistHyper {
  double lr, beta1, beta2; float adam_eps, eps_noise, min_weight, ll_limit;
  int64_t norm_batch;
};

// true when the shape is covered (diagonal covariance, feat_dim % 256 == 0, ...)
bool persist_supported(const PersistShape& s);
// ... and the held-out evaluations of up to s.max_test rows can run inside the launches
bool persist_eval_supported(const PersistShape& s);
size_t persist_workspace_bytes(const PersistShape& s);
// what must be zero at the start of a fit call (cross-workgroup flags, granules, the padding of
// the d_out rows): two regions of the workspace, cleared by the engine's begin kernel
struct ZeroRegion { void* ptr; size_t bytes; };
int persist_reset_regions(const PersistShape& s, void* workspace, size_t workspace_bytes,
                          ZeroRegion* regions);
// n consecutive updates starting at the state block's step counter; advances
// the counter, the jitter RNG stream and the Adam bias-correction powers
int persist_run(const PersistShape& s, const PersistBuffers& b, const PersistHyper& h, int n,
                hipStream_t st);

// 2: unified workgroups (fit_persistent.hip), 0: shape not covered (1 was fit_persistent_v1.hip, retired in round 6)
int persist_variant(const PersistShape& s);
// (diagnostics) the tiling fit_persistent.hip plans for a shape, bsig.h: bsig_debug_persist_geometry
int persist_geometry(const PersistShape& s, int32_t* out);

// diagnostics: [256][8][16] int64 wall-clock stamps of the first 8 updates of every
// following launch (null: off)
void persist_set_profile_buffer(void* buf);
// (tests) occupy `blocks` CUs (workgroups holding lds_bytes of LDS) for `ms` milliseconds
int debug_spin(int blocks, size_t lds_bytes, int ms, hipStream_t st);
void* persist_profile_buffer();

}  // namespace bsig
