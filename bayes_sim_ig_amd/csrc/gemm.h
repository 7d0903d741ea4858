// Internal description of one GEMM launch (see gemm_f32.hip).
#pragma once
#include "common.h"

namespace bsig {

// extra epilogue, internal to the fit engine: the accumulator is the gradient
// of c; c / adam_m / adam_v are updated in place with one Adam step
// (torch.optim.Adam defaults, mdnn.py:203,234) — dW GEMM and optimizer fused.
constexpr int EPI_ADAM = 100;

struct GemmParams {
  const float* a = nullptr; int64_t lda = 0; const int32_t* a_rows = nullptr; int a_kmajor = 0;
  const float* b = nullptr; int64_t ldb = 0; const int32_t* b_rows = nullptr; int b_kmajor = 0;
  float* c = nullptr; int64_t ldc = 0;
  int m = 0, n = 0, k = 0, k_chunk = 0, splits = 1;
  int epilogue = BSIG_EPI_NONE, act = 0;
  const float* bias = nullptr; const float* aux = nullptr; int64_t ldaux = 0; float alpha = 1.f;
  float* partial = nullptr;
  int64_t partial_ld = 0, partial_slab = 0;   // split-K slabs with a row pitch (0: dense [m, n])
  // whole-width forward product (gemm_wide.h) with BSIG_EPI_BIAS: combine the K slices inside the
  // launch instead of in gemm_reduce_kernel.  >= ceil(m / 64) zeroed int32 (left zeroed); c must
  // have a pitch of at least ceil16(n) floats (every column of the padded width is stored).
  int32_t* combine_tickets = nullptr;
  int combine_capacity = 0;            // words behind combine_tickets: the combine needs ceil(m / 64) of them
  // Row offsets resolved on the device (graph replay): logical row i of the
  // gathered / offset operand reads source index i + (dyn[0]+dyn_delta)*stride + base
  // (A: its M rows, or its contraction rows when k-major; same for B).
  const int32_t* dyn = nullptr; int dyn_delta = 0;
  int64_t a_dyn_stride = 0, a_dyn_base = 0, b_dyn_stride = 0, b_dyn_base = 0;
  // EPI_ADAM
  float* adam_m = nullptr; float* adam_v = nullptr; const float* adam_dyn = nullptr;
  float beta1 = 0.9f, beta2 = 0.999f, adam_eps = 1e-8f;
  // Side output of BSIG_EPI_BIAS: per-workgroup partial sums of exp(c[m, n]) over
  // the columns [expsum_col0, expsum_col0 + expsum_ncols) — the head GEMM hands
  // sum(exp(pre_diag)) (jitter scale, mdnn.py:115) to the NLL kernel for free.
  float* expsum = nullptr; int expsum_col0 = 0, expsum_ncols = 0;
  float* grad_out = nullptr;   // optional: also store the raw gradient
  // bias of the same layer, updated by the threads that own column 0:
  // bias_p[row] with gradient bias_g[row] (column sums computed earlier)
  float* bias_p = nullptr; float* bias_m = nullptr; float* bias_v = nullptr;
  const float* bias_g = nullptr;
  // ... or as bias_g_n partial sums per row, bias_g_stride floats apart (added up in index order)
  int bias_g_n = 0; int64_t bias_g_stride = 0;
  // set by gemm_run: a k-contiguous operand whose rows are not 16-byte aligned is still fetched
  // in 16-byte quads (TileLoader::fetch)
  int a_unal = 0, b_unal = 0;
  // set by gemm_run / the kernel: XCD-aware workgroup -> tile remap, the K split of this workgroup
  int xcd_swz = 0, bid_z = 0;
};

// the head widths the whole-width kernels of gemm_wide.h are instantiated for
bool gemm_wide_covers(int n_wide);

// will dW = dO^T X[ids] (both operands k-major, dO at a pitch of lda) run as the whole-width gradient
// kernel + the 16-byte reduce + Adam kernel?  (the one path that takes partial bias sums, bias_g_n > 1)
bool gemm_wide_gradient_applies(int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, const float* a,
                                const float* b, bool gathered);

// n_expsum (optional) receives the number of expsum partials written.
int gemm_run(GemmParams p, void* workspace, size_t workspace_bytes, hipStream_t st,
             int* n_expsum = nullptr);


}  // namespace bsig
