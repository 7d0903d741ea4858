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
// Host-side wait for the exchange stream's part of the LAST resident call (no-op when there is none, or
// it has been waited for): before anything that is not a resident call writes, reduces or consumes the
// gradient buffer on the fit's stream -- a resident launch that gave up releases its call's
// hipStreamWaitValue32s at once, and the stale all-reduces behind them would otherwise run beside the
// launch-per-update path's on the same communicator (round-5 advisor finding).  A HOST wait, not a
// stream wait: profiles/r05_NOTES.md item 5.
int comm_xr_drain(bsig_comm* c);
// May this communicator's ranks stay resident across the exchange?  A 1-rank group: yes.  With peers only
// when RCCL's channels were capped at init (BSIG_DP_RESIDENT=1 in the environment of bsig_comm_init:
// comm.cpp) -- an uncapped ring waits for workgroups the resident launch leaves no CU for.
bool comm_resident_allowed(const bsig_comm* c);
// The probe's answer for the GROUP: usable only if every rank's exchange stream is (ranks that disagreed
// would finish their Adam steps in different kernels).  One 4-byte all-reduce per communicator, cached.
int comm_xr_group_usable(bsig_comm* c, hipStream_t st, bool* usable);

}  // namespace bsig
