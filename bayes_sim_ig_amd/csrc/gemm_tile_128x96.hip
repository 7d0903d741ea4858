// The 128 x 96 tile of the fp32 MFMA GEMM (gemm_kernel.h; four waves, each 32 rows x 96
// columns): outputs whose width is a little over a multiple of 96 / under 288 -- the
// ShadowHand head, Nh = 260 = 3 x 96 - 28, wastes 10 % of the tile area here against
// 32 % on 128-wide tiles.  Its 16 operand-layout variants.
#include "gemm_kernel.h"

namespace bsig {

int launch_tile_128x96(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st) {
  return launch_tile<4, 1, 1, 3>(p, akm, bkm, avec, bvec, st);
}

}  // namespace bsig
