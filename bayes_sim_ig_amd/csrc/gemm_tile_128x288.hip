// The 128 x 288 tile of the fp32 MFMA GEMM (gemm_kernel.h; four waves, each 32 rows x 288
// columns = nine 32 x 32 accumulators): the WHOLE ShadowHand head (Nh = 260) in one column
// block, so that the [8192, 4096] feature operand of a scaled-batch update streams once instead
// of once per 96-wide column block.  Its 16 operand-layout variants.
#include "gemm_kernel.h"

namespace bsig {

int launch_tile_128x288(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st) {
  return launch_tile<4, 1, 1, 9>(p, akm, bkm, avec, bvec, st);
}

}  // namespace bsig
