// gemm_lean_kernel instantiations for one tile shape (see gemm_lean.h); a translation unit each
// so that they compile in parallel.
#include "gemm_lean.h"

namespace bsig {

int launch_lean_128x64(const GemmParams& p, bool akm, bool bkm, hipStream_t st) {
  return launch_lean<2,2,2,1>(p, akm, bkm, st);
}

}  // namespace bsig
