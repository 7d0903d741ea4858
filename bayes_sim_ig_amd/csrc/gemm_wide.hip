// Launchers of gemm_wide_kernel (gemm_wide.h): the whole-head-width products of a large-minibatch
// update.  One instantiation per number of 16-wide head tiles.
#include "gemm_wide.h"

namespace bsig {

template <int NWT, bool KMAJ>
static int launch_wide(const WideParams& p, hipStream_t st) {
  constexpr size_t lds = wide_lds_bytes<NWT>(KMAJ);
  // (the attribute belongs to the function ON A DEVICE: once per device of this process)
  static bool attr_set[64] = {};
  int dev = 0;
  BSIG_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    BSIG_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wide_kernel<NWT, KMAJ>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const dim3 grid(ceil_div(p.n_narrow, 64), p.splits);
  hipLaunchKernelGGL((gemm_wide_kernel<NWT, KMAJ>), grid, dim3(512), lds, st, p);
  BSIG_CHECK_LAUNCH("gemm_wide");
  return BSIG_OK;
}

bool gemm_wide_covers(int n_wide) {
  const int t = ceil_div(n_wide, 16);
  return t == 17;
}

template <bool KMAJ>
static int dispatch(const WideParams& p, hipStream_t st) {
  switch (ceil_div(p.n_wide, 16)) {
    case 17: return launch_wide<17, KMAJ>(p, st);
    default: return BSIG_EUNSUPPORTED;
  }
}

int gemm_wide_forward(const WideParams& p, hipStream_t st) {
  if (!gemm_wide_covers(p.n_wide) || p.k % BK || p.k_chunk % BK || (p.ldx & 3) || (p.ld_wide & 3) ||
      !aligned(p.x, 16) || !aligned(p.wide, 16) || !aligned(p.out, 16) || (p.ld_out & 3))
    return BSIG_EUNSUPPORTED;
  return dispatch<false>(p, st);
}

int gemm_wide_gradient(const WideParams& p, hipStream_t st) {
  if (!gemm_wide_covers(p.n_wide) || !p.ids || p.k % BK || p.k_chunk % BK || (p.ldx & 3) || p.n_narrow % 64 ||
      p.ld_wide != ceil_div(p.n_wide, 16) * 16 || !aligned(p.x, 16) || !aligned(p.wide, 16) ||
      !aligned(p.out, 16) || (p.ld_out & 3))
    return BSIG_EUNSUPPORTED;
  return dispatch<true>(p, st);
}

}  // namespace bsig
