// Small pieces of device-side plumbing the entry points share: pinned staging
// rings for row tables, the argument ring the CPU writes through the BAR, the
// HIP-event stamps of timed launches.  None of them is thread-safe by itself:
// every instance belongs to one replay handle (used under its mutex) or is the
// process-wide ring behind g_ring_mu.
#pragma once

#include "abi_common.h"

namespace emb_abi {

// Row tables travel host -> device through a small ring of pinned slots.  The
// kernel reads the device copy; a slot is reused only after the launch that
// read it has finished (event per slot).
class TableRing {
 public:
  struct Lease {
    int slot;
    uint8_t* host;
    uint8_t* device;
  };

  ~TableRing() { release_all(); }

  Lease acquire(size_t bytes, hipStream_t stream) {
    if (bytes > cap_) regrow(bytes, stream);
    if (events_.empty()) regrow(cap_ ? cap_ : 4096, stream);
    const int slot = next_;
    next_ = (next_ + 1) % kSlots;
    if (busy_[slot]) {
      HIP_OK(hipEventSynchronize(events_[slot]));
      busy_[slot] = false;
    }
    return {slot, host_ + slot * cap_, dev_ + slot * cap_};
  }

  void upload(const Lease& l, size_t bytes, hipStream_t stream) {
    HIP_OK(hipMemcpyAsync(l.device, l.host, bytes, hipMemcpyHostToDevice, stream));
  }

  void retire(const Lease& l, hipStream_t stream) {
    HIP_OK(hipEventRecord(events_[l.slot], stream));
    busy_[l.slot] = true;
  }

 private:
  static constexpr int kSlots = 32;

  void regrow(size_t bytes, hipStream_t stream) {
    size_t cap = 4096;
    while (cap < bytes) cap *= 2;
    for (int s = 0; s < static_cast<int>(events_.size()); ++s)
      if (busy_[s]) HIP_OK(hipEventSynchronize(events_[s]));
    (void)stream;
    if (host_) HIP_OK(hipHostFree(host_));
    if (dev_) HIP_OK(hipFree(dev_));
    host_ = dev_ = nullptr;
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&host_), cap * kSlots, hipHostMallocDefault));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dev_), cap * kSlots));
    if (events_.empty()) {
      events_.resize(kSlots);
      for (auto& e : events_) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    busy_.assign(kSlots, false);
    cap_ = cap;
  }

  void release_all() {
    for (auto& e : events_) (void)hipEventDestroy(e);
    if (host_) (void)hipHostFree(host_);
    if (dev_) (void)hipFree(dev_);
  }

  size_t cap_ = 0;
  uint8_t* host_ = nullptr;
  uint8_t* dev_ = nullptr;
  std::vector<hipEvent_t> events_;
  std::vector<bool> busy_;
  int next_ = 0;
};

// Argument blocks the CPU writes STRAIGHT into device memory (large-BAR
// systems: all of VRAM is mapped into the host's address space; this is how the
// HIP runtime itself places kernel arguments in device memory).  Fine-grained
// memory, so the GPU reads what the host wrote without an L2 copy in between;
// write-combined stores + a store fence, ordered before the doorbell write of
// the launch that follows.  3.7 KB take ~0.8 us — no upload, no writer kernel.
// A slot is reused only after the launch that read it has finished.
class ArgRing {
 public:
  ~ArgRing() {
    for (auto& e : events_) (void)hipEventDestroy(e);
    if (dev_) (void)hipFree(dev_);
  }
  // False when this device cannot do it (no large BAR, allocation refused, or
  // EMB_ARGS_BAR=0): the caller falls back to the writer kernel.
  bool usable() {
    if (state_ == 0) {
      state_ = -1;
      const char* knob = emb::knob("EMB_ARGS_BAR");
      int dev = 0, large = 0;
      if (!(knob && knob[0] == '0') && hipGetDevice(&dev) == hipSuccess &&
          hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && large &&
          hipExtMallocWithFlags(reinterpret_cast<void**>(&dev_), kSlots * kSlotBytes,
                                hipDeviceMallocFinegrained) == hipSuccess) {
        events_.resize(kSlots / kGroup);
        bool ok = true;
        for (auto& e : events_) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        busy_.assign(kSlots / kGroup, false);
        if (ok) state_ = 1;
      }
      (void)hipGetLastError();
    }
    return state_ == 1;
  }
  // Copies `bytes` (<= 4 KiB) of arguments into the next slot; returns its
  // device address.  Call retire() after the launch that reads it.  Slots are
  // guarded in groups of kGroup with one event per group (an event record costs
  // the host ~3.7 us: once per group, not once per launch): a group is entered
  // again only after the event recorded behind its last launch is done.
  void* put(const void* args, size_t bytes, hipStream_t stream) {
    uint8_t* dst = take(stream);
    std::memcpy(dst, args, bytes);
    __builtin_ia32_sfence();
    return dst;
  }
  // The next slot itself (write-combined device memory: fill it front to back,
  // never read it), for callers that build their block in place; follow with
  // publish() before the launch and retire() after it.
  uint8_t* take(hipStream_t stream) {
    if (next_ % kGroup != 0 && stream != group_stream_) {
      // Another stream takes over in the middle of a group (actor / learner):
      // close the group on the stream that filled it so far, start a new one.
      HIP_OK(hipEventRecord(events_[next_ / kGroup], group_stream_));
      busy_[next_ / kGroup] = true;
      next_ = (next_ / kGroup + 1) * kGroup % kSlots;
    }
    slot_ = next_;
    next_ = (next_ + 1) % kSlots;
    const int group = slot_ / kGroup;
    if (slot_ % kGroup == 0) {
      if (busy_[group]) {
        HIP_OK(hipEventSynchronize(events_[group]));
        busy_[group] = false;
      }
      group_stream_ = stream;
    }
    return dev_ + static_cast<size_t>(slot_) * kSlotBytes;
  }
  static void publish() { __builtin_ia32_sfence(); }
  static constexpr size_t kSlotBytes = 4096;
  // After the launch that reads the slot `put` returned (same stream).
  void retire(hipStream_t stream) {
    if (slot_ % kGroup == kGroup - 1) {
      HIP_OK(hipEventRecord(events_[slot_ / kGroup], stream));
      busy_[slot_ / kGroup] = true;
    }
  }

 private:
  // 512 slots in groups of 64: one event record (~3.7 us of host time) per 64
  // launches; a group is entered again 448 launches after it was closed.
  static constexpr int kSlots = 512, kGroup = 64;
  hipStream_t group_stream_ = nullptr;
  int state_ = 0;          // 0 unknown, 1 usable, -1 not
  uint8_t* dev_ = nullptr;
  std::vector<hipEvent_t> events_;
  std::vector<bool> busy_;
  int next_ = 0, slot_ = 0;
};

// HIP-event pairs around gather launches (bench.py's roofline leg).
class LaunchTimer {
 public:
  ~LaunchTimer() {
    for (auto& p : pairs_) {
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
  }
  bool enabled = false;
  bool discard = false;   // stamps only, never read: a small ring of pairs reused in turn
  // Stamp one launch in `every` (a stamped launch costs the host a few
  // microseconds more than a plain one: sampling keeps the timed loop close to
  // the un-instrumented one).
  int every = 1;
  uint64_t tick = 0;
  bool due() { return enabled && (tick++ % static_cast<uint64_t>(every) == 0); }
  // Next (start, stop) pair for hipExtLaunchKernelGGL, or nulls when disabled.
  // Pairs come from a fixed pool created when stamping is switched on (event
  // creation is far too slow to happen inside a timed region); when the pool is
  // used up the stamps are read — all but the newest few belong to launches
  // that finished long ago — and the pool starts over.
  void next(hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!enabled) return;
    if (pairs_.empty()) reserve(discard ? 256 : kPool);
    if (used_ == pairs_.size()) {
      if (discard) used_ = 0;
      else collect();
    }
    *start = pairs_[used_].first;
    *stop = pairs_[used_].second;
    ++used_;
  }
  void reserve(size_t n) {
    while (pairs_.size() < n) {
      hipEvent_t a, b;
      HIP_OK(hipEventCreate(&a));
      HIP_OK(hipEventCreate(&b));
      pairs_.emplace_back(a, b);
    }
  }
  static constexpr size_t kPool = 2048;
  void collect() {
    for (size_t i = 0; i < used_; ++i) {
      HIP_OK(hipEventSynchronize(pairs_[i].second));
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, pairs_[i].first, pairs_[i].second));
      total_ms_ += ms;
      ++launches_;
    }
    used_ = 0;
  }
  void read(int64_t* launches, double* ms, bool reset) {
    collect();
    *launches = launches_;
    *ms = total_ms_;
    if (reset) {
      launches_ = 0;
      total_ms_ = 0;
    }
  }

 private:
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs_;
  size_t used_ = 0;
  int64_t launches_ = 0;
  double total_ms_ = 0;
};

// HIP_FORCE_DEV_KERNARG=0: the runtime leaves kernel arguments in host memory.
inline bool host_kernargs() {
  static const bool value = [] {
    const char* e = std::getenv("HIP_FORCE_DEV_KERNARG");
    return e && e[0] == '0';
  }();
  return value;
}

// While a replay's gather launches are being timed (bench.py's roofline leg),
// its scatter launches carry a completion stamp too (stop event only, from a
// small ring, never read).  Why: a dispatch WITHOUT a completion signal leaves
// its end-of-kernel cache release to the window of the dispatch that follows,
// so a stamped gather behind an unstamped insert reads ~2 us long
// (tools/gather_lab.hip "pipeline study": plain copy 10.2 us alone, 12.2 us
// behind an unstamped tiny kernel, 10.3 us behind one with a stop stamp; wall
// time per pair is the same).  rocprofv3 gives every dispatch a signal, so this
// is also what makes the in-process number agree with the profiler's.
inline bool stamp_predecessors() { return true; }

inline std::mutex g_ring_mu;
inline TableRing& global_ring() {
  static TableRing* ring = new TableRing();  // leaked on purpose: HIP may be gone at exit
  return *ring;
}

inline size_t emb_timer_pool() { return 2048; }

}  // namespace emb_abi
