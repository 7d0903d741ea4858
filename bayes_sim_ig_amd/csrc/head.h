// Internal interface of the mixture-density head (mdn_head.hip).
#pragma once
#include "common.h"

namespace bsig {

// Optional fit-engine hook executed by the finishing kernel's block (0,0):
// kind 1 ends the forward half of an update (publishes the Adam bias
// corrections of that update, advances the step counter), kind 2 ends a
// held-out evaluation; both advance the jitter RNG stream.
struct FinishHook {
  int32_t* state = nullptr; int kind = 0; double lr = 0, beta1 = 0, beta2 = 0;
};
// Where the finishing kernel left the head bias gradients when the caller takes them as PARTIAL
// column sums (one row of Nh sums per 128-row slab of the minibatch) instead of having a second
// kernel add the slabs up: the reduce + Adam kernel of the head's weight gradient adds them itself.
struct HeadBiasPartials { const float* sums = nullptr; int n = 0; int64_t stride = 0; };

struct HeadDyn {
  const int32_t* y_dyn = nullptr; int64_t y_dyn_stride = 0;  // y_rows offset = y_dyn[0]*stride
  int n_sig_ready = 0;   // exp(pre) partial sums already sit at workspace[0..n) (GEMM side output)
  FinishHook hook;
};

int mdn_head_nll_launch(const bsig_head_dims* dims, const float* seg_w, int64_t ld_w,
                        const float* seg_mu, int64_t ld_mu, const float* seg_sg,
                        int64_t ld_sg, const float* seg_lo, int64_t ld_lo, int from_tuple,
                        const float* y, int64_t ldy, const int32_t* y_rows, int64_t batch,
                        int64_t norm_batch, const float* noise, uint64_t seed,
                        uint64_t stream_id, const uint64_t* dyn_rng, float* loss,
                        const int32_t* loss_slot, float* d_out, int64_t ld_dout,
                        float* colsum_out, int32_t* nonfinite, void* workspace,
                        size_t workspace_bytes, hipStream_t st, const HeadDyn* dyn,
                        HeadBiasPartials* bias_partials = nullptr);

int head_sig_capacity();   // max partial sums the head workspace can take

}  // namespace bsig
