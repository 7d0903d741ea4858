// Thread-local error string + version for libbsig_hip.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

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

// ---- roctx ranges (the reference has no tracing; rocprofv3 --marker-trace
//      shows these around the summarizer / projection / update-launch / all-reduce entry points).
//      Bound at run time: rocprofiler-sdk's roctx when a profiler carries it, else roctracer's;
//      without either the ranges are no-ops.
#include <dlfcn.h>
#include <mutex>
namespace bsig {
namespace {
typedef int (*roctx_push_fn)(const char*);
typedef int (*roctx_pop_fn)(void);
roctx_push_fn g_push = nullptr;
roctx_pop_fn g_pop = nullptr;
std::once_flag g_roctx_once;
void load_roctx() {
  const char* off = getenv("BSIG_NO_ROCTX");
  if (off && off[0] == '1') return;
  const char* names[] = {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"};
  for (const char* n : names) {
    void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (!h) continue;
    g_push = reinterpret_cast<roctx_push_fn>(dlsym(h, "roctxRangePushA"));
    g_pop = reinterpret_cast<roctx_pop_fn>(dlsym(h, "roctxRangePop"));
    if (g_push && g_pop) return;
    g_push = nullptr; g_pop = nullptr;
  }
}
}  // namespace
void range_push(const char* name) {
  std::call_once(g_roctx_once, load_roctx);
  if (g_push) (void)g_push(name);
}
void range_pop() {
  if (g_pop) (void)g_pop();
}
}  // namespace bsig

extern "C" const char* bsig_last_error(void) { return bsig::g_err; }
extern "C" int bsig_version(void) { return 106; }

// What the library was compiled against, for a binding to compare with its own view
// (bayes_sim_ig_amd/_lib.py: ctypes mirrors of the structs; a stale object built against an
// older include/bsig.h would otherwise fail silently).  which: 0 -> FNV-1a hash of the header
// text the build saw (build.sh passes it), 1..3 -> sizeof bsig_head_dims / bsig_mdn_cfg /
// bsig_fit_buffers, 4 -> offsetof(bsig_fit_buffers, x_kind).
#ifndef BSIG_HEADER_HASH
#define BSIG_HEADER_HASH 0ull
#endif
extern "C" uint64_t bsig_abi_info(int which) {
  switch (which) {
    case 0: return (uint64_t)BSIG_HEADER_HASH;
    case 1: return sizeof(bsig_head_dims);
    case 2: return sizeof(bsig_mdn_cfg);
    case 3: return sizeof(bsig_fit_buffers);
    case 4: return offsetof(bsig_fit_buffers, x_kind);
    default: return 0;
  }
}
extern "C" int bsig_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return BSIG_ELAUNCH;
  return n;
}
