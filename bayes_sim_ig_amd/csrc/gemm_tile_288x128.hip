// The 288 x 128 tile of the fp32 MFMA GEMM (gemm_kernel.h; four waves, each 288 rows x 32
// columns): the weight gradient of the whole ShadowHand head, dW [260, 4096] = dO^T F, with the
// [8192, 260] gradient operand read once per column block.  Its 16 operand-layout variants.
#include "gemm_kernel.h"

namespace bsig {

int launch_tile_288x128(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st) {
  return launch_tile<1, 4, 9, 1>(p, akm, bkm, avec, bvec, st);
}

}  // namespace bsig
