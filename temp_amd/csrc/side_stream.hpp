// A side stream per caller stream for work of ONE library call that may run beside the rest of the call (fork event -> side
// stream -> join event).  The events are ordinary stream dependencies: inside a HIP-graph capture the side work becomes a parallel
// branch of the graph.  Shared by rgcn_kernels.hip (the relation-weight gradient beside d/dh) and gru_kernels.hip (d_x beside the
// GRU weight gradients).
#pragma once
#include <atomic>
#include <mutex>
#include "common.hpp"

namespace temp {

// ---- side stream for the weight-gradient edge kernel ---------------------------------------------------------------------------
// In a layer's backward the relation-weight gradient (k_rgcn_dw + its fix-up: an L2-gather kernel of small blocks, ~110 us at the
// S-gdelt shape) depends only on dz, like the d/dh aggregation, the self-loop product and the loop-weight gradient -- kernels
// bound by LDS or by the matrix pipe.  It is launched on a per-device side stream between a fork event and a join event, so it
// fills the CUs' spare wave slots under those kernels instead of queueing behind them.  Same kernels, same results.  The events
// are ordinary stream dependencies: inside a HIP-graph capture they become a parallel branch of the graph.  The stream and the two
// events are created once per device, on first use (never inside a capture: every captured step is preceded by warm-up runs);
// nothing is synchronised.  temp_set_option(TEMP_OPT_OVERLAP, 0): off.
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  hipStream_t owner = nullptr;                   // caller stream this entry last served (an entry is re-used by the same stream)
  int dev = -1;
  bool ok = false, tried = false;
  std::atomic_flag busy = ATOMIC_FLAG_INIT;      // held for the duration of ONE backward call
};
// A small pool per process: an entry (side stream + its two events) serves one backward call at a time.  Two host threads that
// run backward passes concurrently (different caller streams, or even the same one) never share events; when every entry is
// busy the call simply runs the weight gradient in-stream.
#define SIDE_POOL 16
inline SideStream* side_acquire(hipStream_t st) {
  static SideStream pool[SIDE_POOL];
  static std::mutex mu;                          // guards the scan AND creation: an entry's state fields are read and written under it
  if (!option(TEMP_OPT_OVERLAP)) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return nullptr;
  std::lock_guard<std::mutex> lock(mu);          // (a handful of backward calls per step: the lock costs nothing next to a launch)
  // first choice: the entry this (device, stream) used before -- a captured graph then sees the same side stream on every capture
  for (int pass = 0; pass < 2; ++pass) {
    for (int i = 0; i < SIDE_POOL; ++i) {
      SideStream& p = pool[i];
      const bool mine = p.tried && p.ok && p.dev == dev && p.owner == st;
      const bool fresh = !p.tried;
      const bool any = p.tried && p.ok && p.dev == dev;
      if (!(pass == 0 ? mine : (fresh || any))) continue;
      if (p.busy.test_and_set(std::memory_order_acquire)) continue;      // (released without the lock, by the call that holds it)
      if (!p.tried) {
        p.tried = true;
        p.dev = dev;
        p.ok = hipStreamCreateWithFlags(&p.s, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&p.join, hipEventDisableTiming) == hipSuccess;
        if (!p.ok) (void)hipGetLastError();
      }
      if (p.ok && p.dev == dev) { p.owner = st; return &p; }
      p.busy.clear(std::memory_order_release);
    }
  }
  return nullptr;
}
// One backward call's use of a side stream: whatever path leaves the call, a branch that was forked is joined back into the
// caller's stream (an un-joined branch would invalidate a HIP-graph capture and leave d_weight in flight behind the return)
// and the entry is released.
struct SideScope {
  SideStream* ss;
  hipStream_t st;
  bool forked = false;                           // the side stream waits on the caller's: it must be joined
  bool join_recorded = false;                    // ss->join was recorded AFTER this call's work (waiting on it otherwise = a stale event)
  SideScope(hipStream_t stream) : ss(side_acquire(stream)), st(stream) {}
  SideScope(const SideScope&) = delete;
  SideScope& operator=(const SideScope&) = delete;
  // Join the forked branch back into the caller's stream.  If recording the join event failed the branch cannot be joined by an
  // event; outside a capture the side stream is drained on the host instead (slow, correct), and the call reports the failure.
  int join() {
    if (!ss || !forked) return TEMP_OK;
    forked = false;
    if (!join_recorded) {
      (void)hipStreamSynchronize(ss->s);
      return TEMP_E_LAUNCH;
    }
    return hipStreamWaitEvent(st, ss->join, 0) == hipSuccess ? TEMP_OK : TEMP_E_LAUNCH;
  }
  ~SideScope() {
    if (!ss) return;
    (void)join();
    ss->busy.clear(std::memory_order_release);
  }
};

// fork: the side stream waits for everything the caller's stream holds so far -> false: stay in-stream
inline bool side_fork(SideScope& sc) {
  if (!sc.ss) return false;
  if (hipEventRecord(sc.ss->fork, sc.st) != hipSuccess || hipStreamWaitEvent(sc.ss->s, sc.ss->fork, 0) != hipSuccess) { (void)hipGetLastError(); return false; }
  sc.forked = true;
  return true;
}
// the side branch's work is issued: record its end (join() then makes the caller's stream wait for it)
inline bool side_done(SideScope& sc) {
  if (hipEventRecord(sc.ss->join, sc.ss->s) != hipSuccess) return false;
  sc.join_recorded = true;
  return true;
}

}  // namespace temp
