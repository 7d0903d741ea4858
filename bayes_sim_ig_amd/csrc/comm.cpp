// The one exchange of the data-parallel fit: a sum all-reduce of the
// flat fp32 gradient buffer across the ranks of one node, RCCL over xGMI, enqueued on
// the fit's stream.  The reference has no counterpart (single device); the call site is
// between loss.backward() and optimizer.step() of mdnn.py:233-234.
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a process that already carries an
// RCCL (PyTorch-ROCm bundles one next to its HIP runtime) keeps using that one — the
// collective must run on the HIP runtime that owns the caller's streams — and the
// library stays loadable on hosts without RCCL (single-GPU use never touches this file).
#include <dlfcn.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// No RCCL headers on this host: the six entry points this file binds with dlsym, declared as
// RCCL's public API has them (the library itself is still found, or not, at run time).
#include <hip/hip_runtime_api.h>
#include <cstddef>
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclFloat32 = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
ncclResult_t ncclBroadcast(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
const char* ncclGetErrorString(ncclResult_t);
}
#endif

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

#include <hip/hip_runtime.h>

#include "comm.h"
#include "common.h"

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclBroadcast) broadcast = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;
char g_load_error[256] = "";

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) {
    snprintf(g_load_error, sizeof(g_load_error), "dlopen(librccl.so.1): %s", dlerror());
    return;
  }
#define BSIG_SYM(field, name)                                                         \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.handle, name)); \
  if (!g_rccl.field) {                                                                \
    snprintf(g_load_error, sizeof(g_load_error), "librccl: no symbol %s", name);      \
    return;                                                                           \
  }
  BSIG_SYM(get_unique_id, "ncclGetUniqueId")
  BSIG_SYM(comm_init_rank, "ncclCommInitRank")
  BSIG_SYM(comm_destroy, "ncclCommDestroy")
  BSIG_SYM(all_reduce, "ncclAllReduce")
  BSIG_SYM(broadcast, "ncclBroadcast")
  BSIG_SYM(error_string, "ncclGetErrorString")
#undef BSIG_SYM
  g_rccl.ok = true;
}

int need_rccl() {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.ok) {
    bsig::set_error("RCCL is not available: %s", g_load_error);
    return BSIG_EUNSUPPORTED;
  }
  return BSIG_OK;
}

#define BSIG_NCCL(call)                                                          \
  do {                                                                           \
    ncclResult_t r__ = (call);                                                   \
    if (r__ != ncclSuccess) {                                                    \
      ::bsig::set_error("%s: %s", #call, g_rccl.error_string(r__));              \
      return BSIG_ELAUNCH;                                                       \
    }                                                                            \
  } while (0)

}  // namespace

struct bsig_comm {
  ncclComm_t nccl = nullptr;       // RCCL communicator, or
  bsig_exchange_fn external = nullptr;   // a caller-supplied exchange (tests, other transports)
  void* external_ctx = nullptr;
  int world = 1, rank = 0, device = -1;
  bsig::CommXr xr;                 // resident-exchange resources (comm_xr), created on first use
  bool xr_made = false;
  int resident_mode = -1;          // bsig_comm_set_resident: -1 policy (BSIG_DP_RESIDENT, default off), 0 never, 1 always
  bool channels_capped = false;    // world > 1: NCCL_MAX_NCHANNELS was capped before the communicator came up
  int xr_last_slot = -1;           // ev_end slot of the last resident call not yet waited for by the host
  int group_usable = -1;           // comm_xr_group_usable: -1 not asked yet
};

namespace bsig {

namespace {
// one thread: raise the word, time the answer (100 MHz wall clock; gives up after 2 ms)
__global__ void xr_probe_kernel(unsigned* ready, const unsigned* done, unsigned target, long long* out) {
  const long long t0 = wall_clock64();
  __hip_atomic_store(ready, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  while ((int)(__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - target) < 0) {
    if (wall_clock64() - t0 > 200000) break;
    __builtin_amdgcn_s_sleep(8);
  }
  out[0] = wall_clock64() - t0;
}
}  // namespace

// The hand-off of one update, as run_dp_resident enqueues it, with xr_probe_kernel in the place of the
// resident launch (and a packet behind it in the launch stream, as the fit has).
static int xr_probe(CommXr& x, hipStream_t st, hipStream_t cand, double* us) {
  const unsigned target = ++x.base;
  x.probe_out[0] = -1;
  BSIG_HIP(hipEventRecord(x.ev_begin[0], st));
  BSIG_HIP(hipStreamWaitEvent(cand, x.ev_begin[0], 0));
  hipLaunchKernelGGL(xr_probe_kernel, dim3(1), dim3(1), 0, st, x.ready, x.done, target, x.probe_out);
  BSIG_CHECK_LAUNCH("xr_probe");
  BSIG_HIP(hipEventRecord(x.ev_end[0], st));
  BSIG_HIP(hipStreamWaitValue32(cand, x.ready, target, hipStreamWaitValueGte, 0xFFFFFFFFu));
  BSIG_HIP(hipStreamWriteValue32(cand, x.done, target, 0));
  BSIG_HIP(hipStreamSynchronize(st));
  BSIG_HIP(hipStreamSynchronize(cand));
  *us = (double)x.probe_out[0] / 100.0;
  return BSIG_OK;
}

int comm_xr(bsig_comm* c, hipStream_t launch_stream, CommXr* out) {
  BSIG_REQUIRE(c && out, "comm_xr: null");
  if (c->external) return BSIG_EUNSUPPORTED;
  if (!c->xr_made) {
    CommXr x;
    // Streams of the HIGHEST priority: the runtime multiplexes the streams of one priority over a few
    // hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and an exchange stream that lands on the
    // queue of the fit's stream sits BEHIND the resident kernel it is meant to answer.  Priorities
    // have queues of their own (four more).
    int lo = 0, hi = 0;
    BSIG_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
    for (int i = 0; i < CommXr::kCand; ++i)
      BSIG_HIP(hipStreamCreateWithPriority(&x.cand[i], hipStreamNonBlocking, hi));
    // (hipStreamWaitValue32 of this runtime is a spinning kernel for either kind of memory)
    BSIG_HIP(hipExtMallocWithFlags(reinterpret_cast<void**>(&x.ready), 8, hipMallocSignalMemory));
    BSIG_HIP(hipMalloc(reinterpret_cast<void**>(&x.done), 512));
    BSIG_HIP(hipHostMalloc(reinterpret_cast<void**>(&x.probe_out), 64, 0));
    BSIG_HIP(hipMemset(x.ready, 0, 8));
    BSIG_HIP(hipMemset(x.done, 0, 512));
    BSIG_HIP(hipDeviceSynchronize());
    for (int i = 0; i < CommXr::kRing; ++i) {
      BSIG_HIP(hipEventCreateWithFlags(&x.ev_begin[i], hipEventDisableTiming));
      BSIG_HIP(hipEventCreateWithFlags(&x.ev_end[i], hipEventDisableTiming));
    }
    c->xr = x;
    c->xr_made = true;
  }
  CommXr& x = c->xr;
  if (!x.probed || x.probed_for != launch_stream) {
    BSIG_HIP(hipStreamSynchronize(launch_stream));
    if (x.stream) BSIG_HIP(hipStreamSynchronize(x.stream));
    int best = 0;
    for (int i = 0; i < CommXr::kCand; ++i) {
      double a = 0, b = 0;
      BSIG_TRY(xr_probe(x, launch_stream, x.cand[i], &a));      // (the first one also loads the runtime's kernels)
      BSIG_TRY(xr_probe(x, launch_stream, x.cand[i], &a));
      BSIG_TRY(xr_probe(x, launch_stream, x.cand[i], &b));
      x.probe_us[i] = a > b ? a : b;
      if (x.probe_us[i] < x.probe_us[best]) best = i;
    }
    x.stream = x.cand[best];
    // (tests: BSIG_DP_XR_PROBE_OK_US=0 makes every candidate unusable -- the launch-per-update fallback)
    const char* ok_us = getenv("BSIG_DP_XR_PROBE_OK_US");
    x.usable = x.probe_us[best] <= (ok_us ? atof(ok_us) : kProbeOkUs);
    x.probed = true; x.probed_for = launch_stream;
    if (getenv("BSIG_DP_XR_TRACE"))
      fprintf(stderr, "comm_xr: probes %.1f %.1f %.1f %.1f us -> candidate %d (%s)\n", x.probe_us[0], x.probe_us[1],
              x.probe_us[2], x.probe_us[3], best, x.usable ? "usable" : "NOT usable");
  }
  *out = x;
  return BSIG_OK;
}

int comm_xr_drain(bsig_comm* c) {
  if (!c || !c->xr_made || c->xr_last_slot < 0) return BSIG_OK;
  BSIG_HIP(hipEventSynchronize(c->xr.ev_end[c->xr_last_slot]));
  c->xr_last_slot = -1;
  return BSIG_OK;
}

bool comm_resident_allowed(const bsig_comm* c) { return c && (c->world == 1 || c->channels_capped); }

int comm_xr_group_usable(bsig_comm* c, hipStream_t st, bool* usable) {
  BSIG_REQUIRE(c && c->xr_made && usable, "comm_xr_group_usable: no resident-exchange state");
  if (c->world == 1) { *usable = c->xr.usable; return BSIG_OK; }
  if (c->group_usable < 0) {
    // (the second 256 bytes of `done`: nothing else uses them)
    float* word = reinterpret_cast<float*>(reinterpret_cast<char*>(c->xr.done) + 256);
    float v = c->xr.usable ? 1.0f : 0.0f;
    BSIG_HIP(hipMemcpyAsync(word, &v, sizeof(v), hipMemcpyHostToDevice, st));
    BSIG_HIP(hipStreamSynchronize(st));
    BSIG_TRY(bsig_comm_allreduce(c, word, 1, reinterpret_cast<bsig_stream_t>(st)));
    BSIG_HIP(hipMemcpyAsync(&v, word, sizeof(v), hipMemcpyDeviceToHost, st));
    BSIG_HIP(hipStreamSynchronize(st));
    c->group_usable = v > (float)c->world - 0.5f ? 1 : 0;
  }
  *usable = c->group_usable == 1;
  return BSIG_OK;
}

int comm_xr_advance(bsig_comm* c, unsigned n) {
  BSIG_REQUIRE(c && c->xr_made, "comm_xr_advance: no resident-exchange state");
  c->xr_last_slot = (int)(c->xr.calls % CommXr::kRing);
  ++c->xr.calls;
  c->xr.base += n;
  if (c->xr.base > (1u << 30)) {      // (once per ~10^9 updates) start over, with nothing in flight
    BSIG_HIP(hipDeviceSynchronize());
    BSIG_HIP(hipMemset(c->xr.ready, 0, 8));
    BSIG_HIP(hipMemset(c->xr.done, 0, 512));
    BSIG_HIP(hipDeviceSynchronize());
    c->xr.base = 0;
  }
  return BSIG_OK;
}

}  // namespace bsig

static_assert(sizeof(ncclUniqueId) == BSIG_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");

extern "C" int bsig_comm_unique_id(void* id_out) {
  BSIG_REQUIRE(id_out, "comm_unique_id: null");
  BSIG_TRY(need_rccl());
  ncclUniqueId id;
  BSIG_NCCL(g_rccl.get_unique_id(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return BSIG_OK;
}

extern "C" int bsig_comm_init(const void* unique_id, int world, int rank, int device,
                              bsig_comm** comm) {
  BSIG_REQUIRE(comm, "comm_init: null handle");
  *comm = nullptr;
  BSIG_REQUIRE(unique_id, "comm_init: null unique id");
  BSIG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of world %d", rank,
               world);
  int n_dev = 0;
  BSIG_HIP(hipGetDeviceCount(&n_dev));
  BSIG_REQUIRE(device >= 0 && device < n_dev, "comm_init: device %d of %d", device, n_dev);
  BSIG_TRY(need_rccl());
  bsig_comm* c = new (std::nothrow) bsig_comm();
  BSIG_REQUIRE(c, "comm_init: out of memory");
  c->world = world; c->rank = rank; c->device = device;
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  // A rank that stays resident across the exchange (BSIG_DP_RESIDENT=1 with peers: opt-in, never run on
  // more than one GPU so far) leaves RCCL the 8 CUs its 248 workgroups do not hold: every channel must
  // be resident on every rank at once, or a ring waits for a workgroup that cannot be scheduled.
  {
    const char* r = getenv("BSIG_DP_RESIDENT");
    if (world > 1 && r && r[0] == '1') {
      setenv("NCCL_MAX_NCHANNELS", "8", 0);
      const char* have = getenv("NCCL_MAX_NCHANNELS");      // (a caller's own, smaller cap is as good)
      c->channels_capped = have && atoi(have) >= 1 && atoi(have) <= 8;
    }
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  hipError_t e = hipSetDevice(device);
  ncclResult_t r = e == hipSuccess ? g_rccl.comm_init_rank(&c->nccl, world, id, rank)
                                   : ncclUnhandledCudaError;
  (void)hipSetDevice(prev);
  if (r != ncclSuccess) {
    bsig::set_error("ncclCommInitRank(world %d, rank %d, device %d): %s", world, rank, device,
                    e == hipSuccess ? g_rccl.error_string(r) : hipGetErrorString(e));
    delete c;
    return BSIG_ELAUNCH;
  }
  *comm = c;
  return BSIG_OK;
}

extern "C" int bsig_comm_init_external(int world, int rank, bsig_exchange_fn exchange, void* ctx,
                                       bsig_comm** comm) {
  BSIG_REQUIRE(comm, "comm_init_external: null handle");
  *comm = nullptr;
  BSIG_REQUIRE(exchange, "comm_init_external: null exchange function");
  BSIG_REQUIRE(world >= 1 && rank >= 0 && rank < world, "comm_init_external: rank %d of world %d",
               rank, world);
  bsig_comm* c = new (std::nothrow) bsig_comm();
  BSIG_REQUIRE(c, "comm_init_external: out of memory");
  c->world = world; c->rank = rank; c->external = exchange; c->external_ctx = ctx;
  *comm = c;
  return BSIG_OK;
}

extern "C" int bsig_comm_transport(const bsig_comm* c) { return !c ? 0 : (c->external ? 2 : 1); }
extern "C" int bsig_comm_world(const bsig_comm* c) { return c ? c->world : 0; }
extern "C" int bsig_comm_rank(const bsig_comm* c) { return c ? c->rank : -1; }

extern "C" int bsig_comm_allreduce(bsig_comm* c, float* buf, int64_t n, bsig_stream_t stream) {
  bsig::Range roctx_range("bsig_comm_allreduce");
  BSIG_REQUIRE(c && buf && n >= 0, "comm_allreduce: bad args");
  if (n == 0) return BSIG_OK;
  if (c->external) {
    const int rc = c->external(c->external_ctx, BSIG_EXCHANGE_SUM, buf, n, 0, stream);
    if (rc != 0) { bsig::set_error("comm_allreduce: external exchange failed (%d)", rc); return BSIG_ELAUNCH; }
    return BSIG_OK;
  }
  BSIG_NCCL(g_rccl.all_reduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, c->nccl,
                              bsig::as_stream(stream)));
  return BSIG_OK;
}

extern "C" int bsig_comm_broadcast(bsig_comm* c, float* buf, int64_t n, int root,
                                   bsig_stream_t stream) {
  BSIG_REQUIRE(c && buf && n >= 0 && root >= 0 && root < c->world, "comm_broadcast: bad args");
  if (n == 0) return BSIG_OK;
  if (c->external) {
    const int rc = c->external(c->external_ctx, BSIG_EXCHANGE_BROADCAST, buf, n, root, stream);
    if (rc != 0) { bsig::set_error("comm_broadcast: external exchange failed (%d)", rc); return BSIG_ELAUNCH; }
    return BSIG_OK;
  }
  BSIG_NCCL(g_rccl.broadcast(buf, buf, (size_t)n, ncclFloat32, root, c->nccl,
                             bsig::as_stream(stream)));
  return BSIG_OK;
}

extern "C" void bsig_comm_set_resident(bsig_comm* c, int mode) { if (c) c->resident_mode = mode < 0 ? -1 : (mode ? 1 : 0); }
extern "C" int bsig_comm_resident_mode(const bsig_comm* c) { return c ? c->resident_mode : -1; }
extern "C" int64_t bsig_comm_resident_calls(const bsig_comm* c) { return c && c->xr_made ? (int64_t)c->xr.calls : 0; }

extern "C" void bsig_comm_destroy(bsig_comm* c) {
  if (!c) return;
  if (c->xr_made) {
    for (int i = 0; i < bsig::CommXr::kCand; ++i) (void)hipStreamSynchronize(c->xr.cand[i]);
    for (int i = 0; i < bsig::CommXr::kRing; ++i) { (void)hipEventDestroy(c->xr.ev_begin[i]); (void)hipEventDestroy(c->xr.ev_end[i]); }
    (void)hipFree(c->xr.ready); (void)hipFree(c->xr.done); (void)hipHostFree(c->xr.probe_out);
    for (int i = 0; i < bsig::CommXr::kCand; ++i) (void)hipStreamDestroy(c->xr.cand[i]);
  }
  if (c->nccl && g_rccl.ok) (void)g_rccl.comm_destroy(c->nccl);
  delete c;
}
