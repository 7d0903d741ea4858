// Thread-local error string + version for libbsig_hip.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/bsig.h"

namespace bsig {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace bsig

extern "C" const char* bsig_last_error(void) { return bsig::g_err; }
extern "C" int bsig_version(void) { return 100; }
extern "C" int bsig_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return BSIG_ELAUNCH;
  return n;
}
