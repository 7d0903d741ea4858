// The 96 x 128 tile of the fp32 MFMA GEMM (gemm_kernel.h; four waves, each 96 rows x 32
// columns): the weight-gradient product of a head with Nh = 260 rows (dW = dO^T F,
// M = 260 = 3 x 96 - 28).  Its 16 operand-layout variants.
#include "gemm_kernel.h"

namespace bsig {

int launch_tile_96x128(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st) {
  return launch_tile<1, 4, 3, 1>(p, akm, bkm, avec, bvec, st);
}

}  // namespace bsig
