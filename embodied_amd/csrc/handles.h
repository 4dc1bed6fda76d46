// The opaque handle types of include/embodied_hip.h.
#pragma once

#include "abi_common.h"
#include "defer_gate.h"
#include "device_rings.h"
#include "kernels.h"
#include "np_random.h"
#include "replay_index.h"
#include "selectors.h"
#include "stream_order.h"

using namespace emb_abi;

struct emb_rng {
  std::mutex mu;
  emb::NpRandom impl;
  explicit emb_rng(const std::vector<uint32_t>& w) : impl(w) {}
};

struct emb_tree {
  std::mutex mu;
  emb::SampleTree impl;
  emb_tree(int b, uint64_t s) : impl(b, s) {}
};

struct emb_selector {
  std::shared_ptr<std::mutex> mu = std::make_shared<std::mutex>();
  std::shared_ptr<DeferGate> gate = make_gate();     // see DeferGate: drained by every operation
  std::shared_ptr<emb::Selector> impl;
};

struct KeyInfo {
  std::string name;
  int64_t rowbytes;
  uint8_t* pool;
};

struct emb_replay {
  std::mutex mu;
  std::unique_ptr<emb::ReplayIndex> index;
  std::shared_ptr<emb::Selector> selector;
  // The selector handle's own lock (emb_selector_* take it): replay operations
  // hold it too, so direct calls on the handle cannot interleave with them.
  std::shared_ptr<std::mutex> selector_mu = std::make_shared<std::mutex>();
  std::shared_ptr<DeferGate> gate;                 // the selector handle's, or this replay's own
  bool may_defer = false;                          // selector is a native Uniform / Prioritized
  int64_t deferred_adds = 0;                       // publishes whose bookkeeping went to the helper
  std::vector<int32_t> defer_rows;                 // the helper's outputs (checked, not used)
  std::vector<emb::StepId> defer_ids;
  std::vector<KeyInfo> keys;
  int key_stepid = -1, key_is_first = -1, key_is_last = -1;
  TableRing ring;
  LaunchTimer timer, timer_other, timer_update;   // gathers, unread predecessor stamps, write-backs
  std::string timed_kernel[2];                    // the kernel the last stamped gather / write-back ran
  bool timing_update = false;                     // set by emb_replay_update around its launches
  std::vector<int32_t> rows, spans;
  std::vector<std::pair<int32_t, int32_t>> runs;   // update: [first, last) pool rows per run
  std::vector<uint32_t> stamp;                     // update: last-writer-wins marks per pool row
  uint32_t stamp_epoch = 0;
  std::vector<emb::StepId> ids;
  // Actor and learner on different HIP streams (StreamOrder below).
  bool multistream = false;
  StreamOrder order;
  // Early insert (emb_replay_obs_stack_insert): the keys of the next add for
  // `workers` that are already in their pool rows, and where they came from.
  struct Prewritten {
    uint64_t token = 0;              // 0 = nothing outstanding
    std::vector<int64_t> workers;
    std::vector<int32_t> rows;
    std::vector<const void*> src;    // per replay key: the buffer it was copied from (null = not written)
    hipStream_t stream = nullptr;
  } pre;
  uint64_t pre_serial = 0, peek_mark = 0;
  // Carried publish (emb_replay_carry_publish): the one small masked key a
  // publish had left, not launched yet -- it rides in the next early-insert
  // launch on the same stream, or is settled by a publish_one launch before
  // anything else touches the pool.
  struct Carried {
    bool active = false;
    const void* src = nullptr;
    uint8_t* pool = nullptr;
    const uint8_t* flags = nullptr;  // the is_last POOL: a step's flag is read at its own row
    int64_t rowbytes = 0, n = 0;
    int dtype = 0;
    std::vector<int32_t> rows;       // the carried step's pool rows (host copy)
    std::vector<int32_t> sorted;     // the same, sorted (does a sampled window end on one of them?)
    hipStream_t stream = nullptr;
  } carry;
  bool carry_publish = false;
  int64_t carried_total = 0, carried_inline = 0;
  // The helper thread's job reads ITS OWN copies of the workers and rows (the
  // next early insert rewrites `pre` while the job may still run, see below).
  std::vector<int64_t> job_workers;
  std::vector<int32_t> job_rows;
  // Rows and step ids of the NEXT step, known when a publish hands its
  // bookkeeping to the helper thread and no worker fills its chunk's last row:
  // every cursor moves on by one (same chunk, index + 1).  The next early insert
  // of the same workers takes them from here WITHOUT waiting for the helper --
  // it then touches neither the index nor the selector -- provided nothing else
  // has been called on this handle in between (`epoch`).  The publish behind it
  // drains as ever and finds the cursors where the prediction put them.
  struct Predicted {
    bool valid = false;
    uint64_t epoch = 0;
    std::vector<int64_t> workers;
    std::vector<int32_t> rows;
    std::vector<emb::StepId> ids;
  } predict;
  uint64_t epoch = 0;               // operations on this handle so far (REP_OP)
  int64_t predicted_inserts = 0;
  int32_t* dev_rows = nullptr;       // the prewrite launch's rows, for the publish launch
  size_t dev_rows_cap = 0;

  ~emb_replay() {
    if (gate)
      while (gate->state.load(std::memory_order_acquire) != 0) sched_yield();
    if (dev_rows) (void)hipFree(dev_rows);
  }

  // While the helper thread may be at the index (from the post of a job to the
  // next drain: the publish's own launch, an early insert on predicted rows) the
  // count of opened chunks is not read: the one seen before the post stands.  A
  // launch in that window writes rows of chunks that were open before the job; a
  // chunk the job opens is first written after the next drain.
  bool index_posted = false;
  int64_t opens_known = -1;
  void order_before(int kind, hipStream_t stream, bool index_busy = false) {
    if (!multistream) return;
    if (!index_busy && !index_posted) opens_known = index->chunks_opened();
    order.before(kind, stream, opens_known);
  }
  void order_after(int kind, hipStream_t stream) {
    if (multistream) order.after(kind, stream);
  }

  // One argument ring per stream (a ring guards its slots with events on the
  // stream that filled them: two streams taking turns on one ring would close a
  // group -- an event record, ~4 us of host time -- at every change of hands).
  ArgRing& args_for(hipStream_t stream) {
    for (auto& entry : arg_rings)
      if (entry.first == stream) return *entry.second;
    if (arg_rings.size() < 4) {
      arg_rings.emplace_back(stream, std::make_unique<ArgRing>());
      return *arg_rings.back().second;
    }
    return *arg_rings.back().second;      // more streams than rings: the last one is shared
  }
  std::vector<std::pair<hipStream_t, std::unique_ptr<ArgRing>>> arg_rings;
};
