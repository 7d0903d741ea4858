// Internal: what a data-parallel rank that stays RESIDENT across the gradient exchange needs from its
// communicator (comm.cpp owns it): a second stream for the exchange, chosen by a probe, and the two
// words of the hand-off (persist.h: PersistBuffers::xr_*; the word the workgroups count themselves in
// belongs to the plan's workspace).
#pragma once
#include "common.h"

struct bsig_comm;

namespace bsig {

struct CommXr {
  static constexpr int kCand = 4, kRing = 4;
  hipStream_t stream = nullptr;     // the exchange runs here: wait(ready) -> all-reduce -> write(done), per update
  unsigned* ready = nullptr;        // the kernel raises it; hipStreamWaitValue32 waits on it
  unsigned* done = nullptr;         // device memory; the stream writes it, the kernel polls it
  unsigned base = 0;                // both words hold base + (updates of the call in flight): they only grow
  hipEvent_t ev_begin[kRing] = {}, ev_end[kRing] = {};   // per call, round robin
  long long calls = 0;              // resident calls enqueued so far
  // The exchange stream is CHOSEN: kCand streams of the highest priority (four hardware queues of
  // their own), each probed against the stream the launches go to -- a one-thread kernel there raises
  // the word and times the answer.  A hardware queue that shares a pipe of the command processor with
  // the launch stream's queue is not served while the resident kernel runs (1 of the 4, which one
  // depends on the order the process created its queues in): measured > 2 ms against 10-20 us.
  hipStream_t cand[kCand] = {};
  hipStream_t probed_for = nullptr; bool probed = false;
  double probe_us[kCand] = {};      // the slower of two probes, per candidate
  bool usable = false;              // the chosen stream answered within kProbeOkUs
  long long* probe_out = nullptr;   // pinned host word the probe kernel reports through
};
constexpr double kProbeOkUs = 100.0;

// Created on first use; (re)probed when `launch_stream` is not the one the choice was made for.
// BSIG_EUNSUPPORTED for a communicator whose exchange is a caller-supplied function (it runs on the
// host: nothing a stream could wait for).
int comm_xr(bsig_comm* c, hipStream_t launch_stream, CommXr* out);
// ... one more call of n updates enqueued
int comm_xr_advance(bsig_comm* c, unsigned n);

}  // namespace bsig
