// The 128 x 32 tile of the fp32 MFMA GEMM (gemm_kernel.h): its 16 operand-layout variants.
#include "gemm_kernel.h"

namespace bsig {

int launch_tile_128x32(const GemmParams& p, bool akm, bool bkm, int avec, int bvec, hipStream_t st) {
  return launch_tile<4, 1, 1, 1>(p, akm, bkm, avec, bvec, st);
}

}  // namespace bsig
