// Cross-stream ordering of pool accesses (actor / learner split).  INVARIANTS:
// the caller holds the replay's mutex; before(kind, stream) is called ahead of
// every launch that touches the pool and after(kind, stream) behind it; nothing
// is recorded at issue time, a stream that must wait records an event on the
// OTHER stream then (it covers everything queued there so far).
#pragma once

#include "abi_common.h"

namespace emb_abi {

// Pool accesses from several HIP streams (actor / learner split: inserts on one
// stream, sample + write-back on another) ordered with as few events as the
// hazards need -- an event record costs the host ~4 us, a stepping loop inserts
// every ~12 us:
//   read  (sample gather)            after every earlier WRITE on another stream;
//   write to live rows (update,      after every earlier write AND read on another
//     scatter_rows)                    stream;
//   write to fresh rows (add: rows   after the other streams' work only when a chunk
//     of the workers' open chunks)     slot has been opened since THIS stream's last look --
//                                      rows of an open chunk belong to no item, so no
//                                      gather reads them and no write-back targets
//                                      them, unless the slot was recycled.
// Nothing is recorded when work is issued: counters only.  The stream that has to
// wait records an event on the OTHER stream at that moment (it covers everything
// queued there so far) and waits for it.  The caller holds the replay's mutex.
struct StreamOrder {
  enum { kRead = 0, kWriteLive = 1, kWriteFresh = 2 };
  static constexpr int kMax = 6;
  struct Entry {
    hipStream_t stream = nullptr;
    uint64_t writes = 0, reads = 0;          // issued so far
    hipEvent_t event = nullptr;
    uint64_t cover_w = 0, cover_r = 0;       // what the event's last record covers
  };
  Entry e[kMax];
  uint64_t seen_w[kMax][kMax] = {}, seen_r[kMax][kMax] = {};    // [waiter][other]
  int n = 0;
  int64_t opens_seen[kMax] = {-1, -1, -1, -1, -1, -1};           // per inserting stream

  ~StreamOrder() {
    for (int i = 0; i < n; ++i)
      if (e[i].event) (void)hipEventDestroy(e[i].event);
  }
  int entry(hipStream_t stream) {
    for (int i = 0; i < n; ++i)
      if (e[i].stream == stream) return i;
    if (n == kMax) {
      // More streams than the table holds (not a stepping loop any more): drain
      // the device and start over.
      HIP_OK(hipDeviceSynchronize());
      for (int i = 0; i < n; ++i) {
        e[i].stream = nullptr;
        e[i].writes = e[i].reads = e[i].cover_w = e[i].cover_r = 0;
      }
      std::memset(seen_w, 0, sizeof(seen_w));
      std::memset(seen_r, 0, sizeof(seen_r));
      std::fill(opens_seen, opens_seen + kMax, int64_t{-1});
      n = 0;
    }
    e[n].stream = stream;
    if (!e[n].event) HIP_OK(hipEventCreateWithFlags(&e[n].event, hipEventDisableTiming));
    return n++;
  }
  void wait_for(int i, int j) {
    Entry& other = e[j];
    if (other.cover_w < other.writes || other.cover_r < other.reads) {
      HIP_OK(hipEventRecord(other.event, other.stream));
      other.cover_w = other.writes;
      other.cover_r = other.reads;
    }
    HIP_OK(hipStreamWaitEvent(e[i].stream, other.event, 0));
    seen_w[i][j] = other.cover_w;
    seen_r[i][j] = other.cover_r;
  }
  void before(int kind, hipStream_t stream, int64_t chunks_opened) {
    const int i = entry(stream);
    bool reads_too = kind == kWriteLive;
    if (kind == kWriteFresh) {
      if (chunks_opened == opens_seen[i]) return;
      opens_seen[i] = chunks_opened;
      reads_too = true;
    }
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      if (e[j].writes > seen_w[i][j] || (reads_too && e[j].reads > seen_r[i][j])) wait_for(i, j);
    }
  }
  void after(int kind, hipStream_t stream) {
    Entry& mine = e[entry(stream)];
    if (kind == kRead) ++mine.reads;
    else ++mine.writes;
  }
};

}  // namespace emb_abi
