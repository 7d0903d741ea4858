// Internal: what a data-parallel rank that stays RESIDENT across the gradient exchange needs from its
// communicator (comm.cpp owns it): a second stream for the exchange and the three words of the
// hand-off (persist.h: PersistBuffers::xr_*).
#pragma once
#include "common.h"

struct bsig_comm;

namespace bsig {

struct CommXr {
  hipStream_t stream = nullptr;     // the exchange runs here: wait(ready) -> all-reduce -> write(done), per update
  unsigned* ready = nullptr;        // signal memory (hipStreamWaitValue32 can wait on it); the kernel raises it
  unsigned* done = nullptr;         // device memory; the stream writes it, the kernel polls it
  unsigned base = 0;                // both words hold base + (updates of the call in flight): they only grow
  static constexpr int kRing = 4;
  hipEvent_t ev_begin[kRing] = {}, ev_end[kRing] = {};   // per call, round robin
  long long calls = 0;              // resident calls enqueued so far
};

// Created on first use.  BSIG_EUNSUPPORTED for a communicator whose exchange is a caller-supplied
// function (it runs on the host: nothing a stream could wait for).
int comm_xr(bsig_comm* c, CommXr* out);
// ... one more call of n updates enqueued
int comm_xr_advance(bsig_comm* c, unsigned n);

}  // namespace bsig
