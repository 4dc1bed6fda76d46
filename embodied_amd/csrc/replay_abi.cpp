// emb_replay_*: the integer index (host) and the payload launches (HIP) of the
// Replay.  Every entry point runs under REP_OP: the replay's mutex, the selector
// handle's mutex, the helper thread drained (defer_gate.h states the rules).
#include "handles.h"

extern "C" {

// ------------------------------------------------------------------- replay --

int32_t emb_replay_create(const emb_replay_config_t* cfg, emb_selector_t* selector, uint64_t seed,
                          emb_replay_t** out) {
  return guarded([&] {
    need(cfg && out, "replay_create: bad arguments");
    emb::ReplayConfig c;
    c.length = cfg->length;
    c.capacity = cfg->capacity;
    c.chunksize = cfg->chunksize;
    c.n_slots = cfg->n_slots;
    c.online = cfg->online != 0;
    c.uid_hi = cfg->uid_hi;
    c.owners = cfg->owners > 0 ? cfg->owners : 1;
    c.workers_per_owner = cfg->workers_per_owner;
    auto rep = std::make_unique<emb_replay>();
    rep->selector = selector ? selector->impl : std::make_shared<emb::Uniform>(seed);
    if (selector) rep->selector_mu = selector->mu;
    rep->gate = selector ? selector->gate : make_gate();
    // (a callback selector runs Python, a mixture may hold one: their work stays
    // on the calling thread)
    rep->may_defer = dynamic_cast<emb::Uniform*>(rep->selector.get()) != nullptr ||
                     dynamic_cast<emb::Prioritized*>(rep->selector.get()) != nullptr;
    rep->index = std::make_unique<emb::ReplayIndex>(c, rep->selector);
    *out = rep.release();
  });
}

int32_t emb_replay_destroy(emb_replay_t* rep) {
  delete rep;
  return EMB_OK;
}

static void settle_carry(emb_replay* rep, bool index_busy = false);

#define REP_OP(...)                                   \
  return guarded([&] {                                \
    need(rep, "replay handle is null");               \
    std::lock_guard<std::mutex> lock(rep->mu);        \
    std::lock_guard<std::mutex> sel_lock(*rep->selector_mu); \
    rep->gate->drain();                               \
    rep->index_posted = false;                        \
    ++rep->epoch;                                     \
    __VA_ARGS__;                                      \
  })

int32_t emb_replay_set_keys(emb_replay_t* rep, int32_t n_keys, const char* const* names,
                            const int64_t* rowbytes, void* const* pools) {
  REP_OP({
    need(n_keys >= 1 && n_keys <= 1024 && names && rowbytes, "set_keys: bad arguments");
    rep->keys.clear();
    rep->key_stepid = rep->key_is_first = rep->key_is_last = -1;
    for (int k = 0; k < n_keys; ++k) {
      need(names[k] && rowbytes[k] > 0, "set_keys: bad key");
      KeyInfo info{names[k], rowbytes[k], pools ? static_cast<uint8_t*>(pools[k]) : nullptr};
      if (info.name == "stepid") {
        need(rowbytes[k] == EMB_STEPID_BYTES, "set_keys: stepid must be 20 bytes");
        rep->key_stepid = k;
      } else if (info.name == "is_first" && rowbytes[k] == 1) {
        rep->key_is_first = k;
      } else if (info.name == "is_last" && rowbytes[k] == 1) {
        rep->key_is_last = k;
      }
      rep->keys.push_back(info);
    }
  });
}

int32_t emb_replay_grow(emb_replay_t* rep, int64_t n_slots, void* const* pools) {
  REP_OP({
    settle_carry(rep);            // (callers that move the pool settle BEFORE they copy it: emb_replay_settle)
    rep->index->grow(n_slots);
    if (pools)
      for (size_t k = 0; k < rep->keys.size(); ++k) rep->keys[k].pool = static_cast<uint8_t*>(pools[k]);
  });
}

static void add_index_locked(emb_replay* rep, int64_t n, const int64_t* workers, int32_t* rows,
                             emb::StepId* ids) {
  if (!rep->index->fits(workers, n)) throw emb::PoolFull();
  for (int64_t i = 0; i < n; ++i)
    rows[i] = static_cast<int32_t>(rep->index->add(workers[i], &ids[i]));
}

static void sample_index_locked(emb_replay* rep, int64_t batch, int32_t mode, int32_t* rows,
                                uint8_t* online, std::vector<int32_t>* spans = nullptr,
                                uint8_t* first_ids = nullptr, int64_t* workers = nullptr) {
  need(mode >= EMB_MODE_TRAIN && mode <= EMB_MODE_EVAL, "sample: bad mode");
  const int64_t L = rep->index->config().length;
  bool spans_ok = spans != nullptr;
  if (spans) spans->resize(3 * batch);
  for (int64_t b = 0; b < batch; ++b) {
    bool from_online = false;
    const auto pos = rep->index->draw(mode == EMB_MODE_TRAIN, &from_online);
    if (!rep->index->rows(pos, L, rows + b * L))
      throw std::logic_error("replay: sampled window vanished");
    if (spans_ok) spans_ok = rep->index->two_spans(pos, L, spans->data() + 3 * b);
    if (online) online[b] = from_online ? 1 : 0;
    if (workers) workers[b] = rep->index->worker_of(pos);
    if (first_ids) {
      const emb::StepId sid = rep->index->make_stepid(pos.first, pos.second);
      std::memcpy(first_ids + b * EMB_STEPID_BYTES, sid.b, EMB_STEPID_BYTES);
    }
  }
  if (spans && !spans_ok) spans->clear();
}

int32_t emb_replay_add_index(emb_replay_t* rep, int64_t n, const int64_t* workers,
                             int32_t* rows_out, uint8_t* stepids_out, int32_t* new_chunks_out) {
  REP_OP({
    need(n >= 0 && workers && rows_out, "add_index: bad arguments");
    rep->ids.resize(n);
    const int64_t before = rep->index->recycled_opens();
    add_index_locked(rep, n, workers, rows_out, rep->ids.data());
    if (stepids_out) std::memcpy(stepids_out, rep->ids.data(), n * EMB_STEPID_BYTES);
    if (new_chunks_out) *new_chunks_out = static_cast<int32_t>(rep->index->recycled_opens() - before);
  });
}

int32_t emb_replay_sample_index(emb_replay_t* rep, int64_t batch, int32_t mode, int32_t* rows_out,
                                uint8_t* online_out, int64_t* workers_out) {
  REP_OP({
    need(batch >= 0 && rows_out, "sample_index: bad arguments");
    sample_index_locked(rep, batch, mode, rows_out, online_out, nullptr, nullptr, workers_out);
  });
}

int32_t emb_replay_resolve(emb_replay_t* rep, int64_t n, const uint8_t* stepids, int64_t count,
                           int32_t* rows_out, uint8_t* found_out) {
  REP_OP({
    need(n >= 0 && stepids && count >= 0 && rows_out, "resolve: bad arguments");
    for (int64_t i = 0; i < n; ++i) {
      emb::ReplayIndex::Pos pos;
      bool ok = rep->index->parse_stepid(stepids + i * EMB_STEPID_BYTES, &pos);
      if (ok) ok = rep->index->rows(pos, count, rows_out + i * count);
      else for (int64_t j = 0; j < count; ++j) rows_out[i * count + j] = -1;
      if (found_out) found_out[i] = ok ? 1 : 0;
    }
  });
}

int32_t emb_replay_prioritize(emb_replay_t* rep, const uint8_t* stepids, const double* prios, int64_t n) {
  REP_OP({
    need(stepids && prios && n >= 0, "prioritize: bad arguments");
    if (!rep->selector->can_prioritize())
      throw std::invalid_argument("replay: selector has no prioritize()");  // AttributeError in replay.py:137
    rep->selector->prioritize(reinterpret_cast<const emb::StepId*>(stepids), prios, n);
  });
}

int32_t emb_replay_len(emb_replay_t* rep, int64_t* items) { REP_OP(need(items, "replay_len: null output"); *items = rep->index->size()); }
int32_t emb_replay_online_pending(emb_replay_t* rep, int64_t* n) { REP_OP(need(n, "replay_online_pending: null output"); *n = rep->index->online_pending()); }
int32_t emb_replay_sampler_len(emb_replay_t* rep, int64_t* n) { REP_OP(need(n, "replay_sampler_len: null output"); *n = rep->selector->size()); }
int32_t emb_replay_free_slots(emb_replay_t* rep, int64_t* n) { REP_OP(need(n, "replay_free_slots: null output"); *n = rep->index->free_slots()); }
int32_t emb_replay_stats(emb_replay_t* rep, int64_t out[6], int32_t reset) { REP_OP(need(out, "replay_stats: null output"); rep->index->stats(out, reset != 0)); }

// Launch one gather/scatter.  Small tables travel inside the kernel arguments;
// larger ones through the pinned ring (one async upload).
static void run_move(emb_replay* rep, emb::MovePlan& plan, const int32_t* rows, int64_t n_rows,
                     const emb::StepId* ids, int stepid_plan_slot, bool gather, hipStream_t stream,
                     const std::vector<int32_t>* spans = nullptr) {
  plan.n_rows = static_cast<int32_t>(n_rows);
  plan.rows_host = rows;
  plan.args_in_host_memory = host_kernargs();
  plan.cu_limit = g_cu_streams.cus_of(stream);
  if (spans && !spans->empty()) {
    plan.spans_host = spans->data();
    plan.n_seq = static_cast<int32_t>(spans->size() / 3);
  }
  if (ids) {
    plan.inline_key = stepid_plan_slot;
    plan.inline_bytes = reinterpret_cast<const uint8_t*>(ids);
  }
  TableRing::Lease lease{-1, nullptr, nullptr};
  bool fits = emb::plan_fits_inline(plan);          // spans (+ step ids)
  if (!fits && plan.spans_host) {
    plan.spans_host = nullptr;
    plan.n_seq = 0;
    fits = emb::plan_fits_inline(plan);             // plain rows (+ step ids)
  }
  if (!fits) {
    // Ring slot layout: int32 rows[n_rows] | (16-aligned) step ids.
    plan.rows_host = nullptr;
    plan.inline_key = -1;
    plan.inline_bytes = nullptr;
    const size_t rows_bytes = static_cast<size_t>(n_rows) * sizeof(int32_t);
    const size_t ids_off = (rows_bytes + 15) & ~size_t(15);
    const size_t total = ids ? ids_off + static_cast<size_t>(n_rows) * EMB_STEPID_BYTES : rows_bytes;
    lease = rep->ring.acquire(total, stream);
    std::memcpy(lease.host, rows, rows_bytes);
    if (ids) {
      std::memcpy(lease.host + ids_off, ids, static_cast<size_t>(n_rows) * EMB_STEPID_BYTES);
      plan.key[stepid_plan_slot].batch = lease.device + ids_off;
    }
    rep->ring.upload(lease, total, stream);
    plan.rows = reinterpret_cast<const int32_t*>(lease.device);
  }
  HostLap hp;
  emb::MoveLaunch launch;
  HIP_OK(emb::prepare_move(plan, &launch, gather));
  hp.lap(gather ? 11 : 16, gather ? "gather: prepare_move" : "scatter: prepare_move");
  // Processes that keep kernel arguments in host memory (HIP_FORCE_DEV_KERNARG=0,
  // cheaper launches) pay PCIe latency on every wave's argument reads: for big
  // moves hand the kernel a device copy of its arguments instead.
  // Stamped launches: sample gathers, and (when asked for) the write-backs of
  // emb_replay_update, each with its own counter.
  LaunchTimer& own = gather ? rep->timer : rep->timer_update;
  const bool stamp_this = (gather || rep->timing_update) && own.due();
  TableRing::Lease args_lease{-1, nullptr, nullptr};
  const void* device_args = nullptr;
  bool args_in_bar = false;
  if (host_kernargs()) {
    int64_t bytes = 0;
    for (int k = 0; k < plan.n_keys; ++k) bytes += plan.key[k].rowbytes * n_rows;
    // Smallest move (bytes) that gets a device copy of its arguments: 4 MB for
    // writes (a plain insert of 64 Atari steps, 1.8 MB, is cheaper on the host
    // without the copy) and 1 MB for gathers: a B = 1 or 2 sample of 65 x 28 KB
    // steps (1.8 / 3.7 MB) takes 8.8 us with its 4 KB of arguments in host memory
    // and 4.9 us with the device copy, for 0.4 us of host time.
    const int64_t least = gather ? (int64_t{1} << 20) : (int64_t{4} << 20);
    if (bytes >= least) {
      hipEvent_t none = nullptr, done = nullptr;
      if (stamp_this && stamp_predecessors()) {
        rep->timer_other.enabled = rep->timer_other.discard = true;
        rep->timer_other.next(&none, &done);
      }
      ArgRing& arg_ring = rep->args_for(stream);
      if (arg_ring.usable()) {
        // The CPU writes the block into device memory through the BAR.
        device_args = arg_ring.put(launch.args, emb::move_args_bytes(), stream);
        args_in_bar = true;
        // A timed gather wants a predecessor that carries a completion stamp
        // (see stamp_predecessors): a one-lane marker kernel, only then.
        if (done) HIP_OK(emb::launch_marker(stream, done));
      } else {
        // No large BAR: a one-workgroup kernel writes the block (an H2D copy in
        // front of the mover costs more on both sides); while gathers are timed
        // it carries the completion stamp.
        args_lease = rep->ring.acquire(emb::move_args_bytes(), stream);
        HIP_OK(emb::launch_args_writer(launch, args_lease.device, stream, done));
        device_args = args_lease.device;
      }
    }
  }
  hp.lap(gather ? 12 : 17, gather ? "gather: args -> device (+marker)" : "scatter: args -> device");
  const int access = gather ? StreamOrder::kRead
                            : plan.fresh_rows ? StreamOrder::kWriteFresh : StreamOrder::kWriteLive;
  rep->order_before(access, stream);
  hipEvent_t start = nullptr, stop = nullptr;
  if (stamp_this) {
    own.next(&start, &stop);
    rep->timed_kernel[gather ? 0 : 1] = emb::move_kernel_name(launch, gather, device_args != nullptr);
  } else if (!gather && rep->timer.enabled && stamp_predecessors() && !host_kernargs()) {
    // (With host-resident kernel arguments every big gather already follows its
    // stamped argument-writer launch, and a stamp on each insert would cost
    // ~10 % of the step rate there; with device-resident arguments it is free.)
    rep->timer_other.enabled = rep->timer_other.discard = true;
    rep->timer_other.next(&start, &stop);
    start = nullptr;                 // completion stamp only
  }
  HIP_OK(emb::launch_move(launch, gather, device_args, stream, start, stop));
  rep->order_after(access, stream);
  hp.lap(gather ? (stamp_this ? 14 : 13) : 18,
         gather ? (stamp_this ? "gather: launch (stamped)" : "gather: launch") : "scatter: launch");
  if (args_in_bar) rep->args_for(stream).retire(stream);
  if (args_lease.slot >= 0) rep->ring.retire(args_lease, stream);
  if (lease.slot >= 0) rep->ring.retire(lease, stream);
  hp.lap(gather ? 15 : 19, gather ? "gather: retire" : "scatter: retire");
}

// Any number of keys: launches of at most kMaxKeys keys each (the kernel
// argument block is 4 KiB).
struct KeyList {
  std::vector<emb::KeyDesc> key;
  int key_is_first = -1, key_is_last = -1, key_stepid = -1;
  int32_t seq_len = 1;
  int32_t group = 0;            // gather: destination groups (MovePlan::group)
  int64_t group_stride = 0;
  bool fresh_rows = false;      // scatter: MovePlan::fresh_rows (an insert)
  // Masked insert: per key a DType code (-1 = plain copy) and the buffer that
  // also receives the masked value; mask_flags = is_last of the rows.
  std::vector<int8_t> mask_dtype;
  std::vector<uint8_t*> mask_out;
  const uint8_t* mask_flags = nullptr;
  // Gather: steps of a sequence that key k receives (MovePlan::key_len; 0 = all).
  std::vector<int32_t> key_len;
  KeyList() {           // one allocation each instead of a doubling series per call
    key.reserve(16);
    mask_dtype.reserve(16);
    mask_out.reserve(16);
    key_len.reserve(16);
  }
  void push(uint8_t* pool, const void* batch, int64_t rowbytes, int32_t len = 0) {
    key.push_back({pool, const_cast<uint8_t*>(static_cast<const uint8_t*>(batch)), rowbytes});
    mask_dtype.push_back(-1);
    mask_out.push_back(nullptr);
    key_len.push_back(len);
  }
};

static void run_move_all(emb_replay* rep, KeyList& list, const int32_t* rows, int64_t n_rows,
                         const emb::StepId* ids, bool gather, hipStream_t stream,
                         const std::vector<int32_t>* spans = nullptr) {
  const uint8_t* first_pool = list.key_is_first >= 0 ? list.key[list.key_is_first].pool : nullptr;
  const int total = static_cast<int>(list.key.size());
  for (int lo = 0; lo < total; lo += emb::kMaxKeys) {
    const int hi = std::min(total, lo + emb::kMaxKeys);
    emb::MovePlan plan;
    plan.seq_len = list.seq_len;
    plan.group = list.group;
    plan.group_stride = list.group_stride;
    plan.fresh_rows = list.fresh_rows;
    plan.is_first_pool = first_pool;
    for (int k = lo; k < hi; ++k) {
      if (list.mask_flags && list.mask_dtype[k] >= 0) {
        plan.mask_bits |= 1u << plan.n_keys;
        plan.mask_dtype[plan.n_keys] = list.mask_dtype[k];
        plan.mask_out[plan.n_keys] = list.mask_out[k];
        plan.mask_flags = list.mask_flags;
      }
      plan.key_len[plan.n_keys] = list.key_len[k];
      plan.key[plan.n_keys++] = list.key[k];
    }
    if (list.key_is_first >= lo && list.key_is_first < hi) plan.key_is_first = list.key_is_first - lo;
    if (list.key_is_last >= lo && list.key_is_last < hi) plan.key_is_last = list.key_is_last - lo;
    const int sid = (ids && list.key_stepid >= lo && list.key_stepid < hi) ? list.key_stepid - lo : -1;
    run_move(rep, plan, rows, n_rows, sid >= 0 ? ids : nullptr, sid, gather, stream, spans);
  }
}

// A completion stamp for a pool write while gathers are being timed with
// device-resident kernel arguments (see stamp_predecessors), else null.
static hipEvent_t write_stamp(emb_replay* rep) {
  hipEvent_t none = nullptr, stop = nullptr;
  if (rep->timer.enabled && stamp_predecessors() && !host_kernargs()) {
    rep->timer_other.enabled = rep->timer_other.discard = true;
    rep->timer_other.next(&none, &stop);
  }
  return stop;
}

// A carried publish that cannot ride in an early-insert launch (something else
// touches the pool first): the publish_one launch it replaced, now.  The rows
// are still in dev_rows -- only the next early insert overwrites them, and that
// one takes the carry along itself.
static void settle_carry(emb_replay* rep, bool index_busy) {
  emb_replay::Carried& c = rep->carry;
  if (!c.active) return;
  c.active = false;
  rep->order_before(StreamOrder::kWriteFresh, c.stream, index_busy);
  HIP_OK(emb::launch_publish_one(c.src, c.pool, nullptr, rep->dev_rows, c.flags, c.n, c.rowbytes, c.dtype,
                                 c.stream, write_stamp(rep), /*flags_by_row=*/true));
  rep->order_after(StreamOrder::kWriteFresh, c.stream);
}

// The helper thread's job (DeferGate): the index bookkeeping of a publish whose
// rows were fixed by the early insert.  `pre.workers` / `pre.rows` are not
// written again before the next replay operation, which drains the gate first.
static void deferred_add(void* ctx) {
  emb_replay* rep = static_cast<emb_replay*>(ctx);
  const int64_t n = static_cast<int64_t>(rep->job_workers.size());
  rep->defer_rows.resize(n);
  rep->defer_ids.resize(n);
  add_index_locked(rep, n, rep->job_workers.data(), rep->defer_rows.data(), rep->defer_ids.data());
  if (!std::equal(rep->defer_rows.begin(), rep->defer_rows.end(), rep->job_rows.begin()))
    throw std::logic_error("replay: a deferred add left the rows of its early insert");
}

static void add_locked(emb_replay* rep, int64_t n, const int64_t* workers, const void* const* src,
                       int32_t n_masked, const int32_t* masked_keys, const int32_t* masked_dtypes,
                       void* const* masked_out, const void* is_last, hipStream_t stream,
                       uint64_t token = 0) {
  need(n >= 0 && workers && src, "add: bad arguments");
  need(!rep->keys.empty(), "add: call emb_replay_set_keys first");
  need(n_masked == 0 || (masked_keys && masked_dtypes && is_last), "add: bad mask arguments");
  if (n == 0) return;
  settle_carry(rep);            // (in a stepping loop the early insert in between has taken it along)
  for (size_t k = 0; k < rep->keys.size(); ++k) {
    need(rep->keys[k].pool, "add: key has no pool");
    need(static_cast<int>(k) == rep->key_stepid || src[k], "add: null source buffer");
  }
  for (int32_t j = 0; j < n_masked; ++j) {
    need(masked_keys[j] >= 0 && masked_keys[j] < static_cast<int32_t>(rep->keys.size()) &&
             masked_keys[j] != rep->key_stepid, "add: masked key id out of range");
    need(masked_dtypes[j] >= 0 && masked_dtypes[j] <= emb::kBool, "add: bad masked dtype");
  }
  if (token != 0 && rep->pre.token == token)
    for (int32_t j = 0; j < n_masked; ++j)     // a masked key must not have been written unmasked
      need(!rep->pre.src[masked_keys[j]] || rep->pre.src[masked_keys[j]] != src[masked_keys[j]],
           "add: a masked key was part of the early insert");
  HostLap hp;
  // Keys that emb_replay_obs_stack_insert already wrote: same token, same
  // workers, same stream, and the rows this add is given are the peeked ones.
  emb_replay::Prewritten& pre = rep->pre;
  bool early = token != 0 && pre.token == token && pre.stream == stream &&
               static_cast<int64_t>(pre.workers.size()) == n &&
               std::equal(workers, workers + n, pre.workers.begin());
  const int32_t* rows = nullptr;
  bool deferred = false;
  if (early && rep->may_defer && defer_index() && rep->index->config().owners == 1 &&
      rep->gate->allowed()) {
    // The rows an add hands out are the cursors peek reads: if they still are
    // what the early insert saw, the bookkeeping (which the launch below does not
    // need -- rows and step ids are in device memory already) runs on the helper
    // thread while this thread launches and goes back to the interpreter.
    const uint64_t mark = ++rep->peek_mark;
    const int64_t chunksize = rep->index->config().chunksize;
    int64_t rotations = 0;               // workers that fill their chunk's last row: one new slot each
    bool same = true;
    emb_replay::Predicted& next = rep->predict;
    next.valid = false;
    next.ids.resize(n);
    for (int64_t i = 0; i < n && same; ++i) {
      int64_t row = 0;
      same = rep->index->peek(workers[i], mark, &row, &next.ids[i]) && row == pre.rows[i];
      rotations += (row % chunksize) + 1 >= chunksize;
    }
    // (PoolFull must be raised before anything changes: only a batch that cannot
    // run out of slots goes to the helper)
    if (same && rotations <= rep->index->free_slots()) {
      rep->job_workers = pre.workers;
      rep->job_rows = pre.rows;
      if (rep->multistream) rep->opens_known = rep->index->chunks_opened();
      rep->index_posted = true;
      rep->gate->post(&deferred_add, rep);
      rep->deferred_adds += 1;
      rows = pre.rows.data();
      deferred = true;
      if (rotations == 0 && predict_rows()) {
        // where the next step of these workers goes: one row on, same chunk
        next.workers = pre.workers;
        next.rows.resize(n);
        for (int64_t i = 0; i < n; ++i) {
          next.rows[i] = pre.rows[i] + 1;
          uint8_t* be = next.ids[i].b + 16;            // 4-byte big-endian row-in-chunk
          const uint32_t index = (uint32_t{be[0]} << 24 | uint32_t{be[1]} << 16 | uint32_t{be[2]} << 8 | be[3]) + 1;
          be[0] = static_cast<uint8_t>(index >> 24);
          be[1] = static_cast<uint8_t>(index >> 16);
          be[2] = static_cast<uint8_t>(index >> 8);
          be[3] = static_cast<uint8_t>(index);
        }
        next.epoch = rep->epoch;
        next.valid = true;
      }
      hp.lap(22, "add: peek check + post");
    }
  }
  if (!deferred) {
    rep->rows.resize(n);
    rep->ids.resize(n);
    add_index_locked(rep, n, workers, rep->rows.data(), rep->ids.data());   // PoolFull: nothing changed yet
    hp.lap(0, "add: index bookkeeping");
    early = early && std::equal(rep->rows.begin(), rep->rows.end(), pre.rows.begin());
    rows = rep->rows.data();
  }
  pre.token = 0;        // any add consumes an outstanding early insert
  KeyList list;
  for (size_t k = 0; k < rep->keys.size(); ++k) {
    if (early && pre.src[k] && (static_cast<int>(k) == rep->key_stepid || pre.src[k] == src[k]))
      continue;                                         // already in its pool rows
    if (static_cast<int>(k) == rep->key_stepid) {
      list.key_stepid = static_cast<int>(list.key.size());
      list.push(rep->keys[k].pool, nullptr, rep->keys[k].rowbytes);
      continue;
    }
    list.push(rep->keys[k].pool, src[k], rep->keys[k].rowbytes);
    for (int32_t j = 0; j < n_masked; ++j) {
      if (masked_keys[j] != static_cast<int32_t>(k)) continue;
      list.mask_dtype.back() = static_cast<int8_t>(masked_dtypes[j]);
      list.mask_out.back() = masked_out ? static_cast<uint8_t*>(masked_out[j]) : nullptr;
      list.mask_flags = static_cast<const uint8_t*>(is_last);
    }
  }
  if (list.key.empty()) return;
  hp.lap(1, "add: early check + key list");
  if (early && list.key.size() == 1 && list.key_stepid < 0 &&
      list.key[0].rowbytes * n <= (int64_t{1} << 20)) {
    // All that is left is one small key (the action): the rows are in device
    // memory since the early insert, the launch needs 56 bytes of arguments.
    const bool masked = list.mask_flags && list.mask_dtype[0] >= 0;
    // (the flags the mask uses are this step's is_last, and the early insert has
    // put exactly that buffer into the is_last pool rows of this step)
    const bool flags_stored = rep->key_is_last >= 0 && rep->keys[rep->key_is_last].pool &&
                              pre.src[rep->key_is_last] == static_cast<const void*>(list.mask_flags);
    if (rep->carry_publish && masked && !list.mask_out[0] && n <= INT32_MAX && flags_stored &&
        emb::carry_supported(list.key[0].rowbytes, list.mask_dtype[0])) {
      // Nobody wants the masked values back: no launch now.  The source is read
      // by the next launch on this replay (the caller's contract,
      // emb_replay_carry_publish); the flags are read from the replay's own
      // is_last rows of this step, so the env may reuse its flag buffer at once.
      emb_replay::Carried& c = rep->carry;
      c.active = true;
      c.src = list.key[0].batch;
      c.pool = list.key[0].pool;
      c.flags = rep->keys[rep->key_is_last].pool;
      c.rowbytes = list.key[0].rowbytes;
      c.n = n;
      c.dtype = list.mask_dtype[0];
      c.rows.assign(rows, rows + n);
      c.sorted = c.rows;
      std::sort(c.sorted.begin(), c.sorted.end());
      c.stream = stream;
      rep->carried_total += 1;
      hp.lap(2, "add: publish_one launch");
      return;
    }
    rep->order_before(StreamOrder::kWriteFresh, stream);
    HIP_OK(emb::launch_publish_one(list.key[0].batch, list.key[0].pool, masked ? list.mask_out[0] : nullptr,
                                   rep->dev_rows, masked ? list.mask_flags : nullptr, n,
                                   list.key[0].rowbytes, masked ? list.mask_dtype[0] : emb::kU8, stream,
                                   write_stamp(rep)));
    rep->order_after(StreamOrder::kWriteFresh, stream);
    hp.lap(2, "add: publish_one launch");
    return;
  }
  // (a deferred add never has the step ids in the list: the early insert wrote them)
  list.fresh_rows = true;
  run_move_all(rep, list, rows, n, list.key_stepid >= 0 ? rep->ids.data() : nullptr, false, stream);
  hp.lap(3, "add: mover launch (run_move)");
}

int32_t emb_replay_add(emb_replay_t* rep, int64_t n, const int64_t* workers, const void* const* src,
                       void* stream) {
  REP_OP(add_locked(rep, n, workers, src, 0, nullptr, nullptr, nullptr, nullptr,
                    static_cast<hipStream_t>(stream)));
}

int32_t emb_replay_add_masked(emb_replay_t* rep, int64_t n, const int64_t* workers,
                              const void* const* src, int32_t n_masked, const int32_t* masked_keys,
                              const int32_t* masked_dtypes, void* const* masked_out,
                              const void* is_last, void* stream) {
  REP_OP(add_locked(rep, n, workers, src, n_masked, masked_keys, masked_dtypes, masked_out, is_last,
                    static_cast<hipStream_t>(stream)));
}

int32_t emb_replay_publish(emb_replay_t* rep, int64_t n, const int64_t* workers,
                           const void* const* src, int32_t n_masked, const int32_t* masked_keys,
                           const int32_t* masked_dtypes, void* const* masked_out, const void* is_last,
                           uint64_t token, void* stream) {
  REP_OP(add_locked(rep, n, workers, src, n_masked, masked_keys, masked_dtypes, masked_out, is_last,
                    static_cast<hipStream_t>(stream), token));
}

int32_t emb_replay_obs_stack_insert(emb_replay_t* rep, int64_t n, const int64_t* workers,
                                    int32_t frame_key, const void* frames, const emb_obs_spec_t* spec,
                                    void* dst, const void* const* src, void* stream,
                                    uint64_t* token_out) {
  return guarded([&] {
    need(rep, "replay handle is null");
    std::lock_guard<std::mutex> lock(rep->mu);
    std::lock_guard<std::mutex> sel_lock(*rep->selector_mu);
    // Rows predicted by the publish before this call (emb_replay::Predicted):
    // nothing below touches the index or the selector then, so the helper
    // thread's job may still be running.
    emb_replay::Predicted& known = rep->predict;
    const bool predicted = known.valid && known.epoch == rep->epoch && workers && n > 0 &&
                           static_cast<int64_t>(known.workers.size()) == n &&
                           std::equal(workers, workers + n, known.workers.begin());
    known.valid = false;
    if (!predicted) {
      rep->gate->drain();
      rep->index_posted = false;
    }
    ++rep->epoch;
    need(n >= 0 && workers && frames && spec && dst && src && token_out, "obs_stack_insert: bad arguments");
    need(spec->pixels > 0 && spec->channels > 0, "obs_stack_insert: bad frame shape");
    need(spec->layout == EMB_LAYOUT_SAME || spec->layout == EMB_LAYOUT_CHANNELS_FIRST,
         "obs_stack_insert: bad layout");
    *token_out = 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n == 0) return;
    HostLap hp;
    emb_replay::Prewritten& pre = rep->pre;
    pre.token = 0;
    const int n_keys = static_cast<int>(rep->keys.size());
    emb::PrewritePlan plan;
    plan.frames = static_cast<const uint8_t*>(frames);
    plan.dst = dst;
    plan.pixels = spec->pixels;
    plan.channels = spec->channels;
    plan.layout = spec->layout;
    plan.out_dtype = spec->out_dtype;
    plan.scale = spec->scale;
    plan.offset = spec->offset;
    plan.n = static_cast<int32_t>(n);
    bool early = frame_key >= 0 && frame_key < n_keys && frame_key != rep->key_stepid &&
                 rep->keys[frame_key].pool && src[frame_key] == frames &&
                 rep->keys[frame_key].rowbytes == spec->pixels * spec->channels && n <= INT32_MAX;
    if (early && predicted) {
      rep->rows = known.rows;
      rep->ids = known.ids;
      rep->predicted_inserts += 1;
    } else if (early) {
      if (predicted) rep->gate->drain();       // (not reached: `early` only depends on the arguments)
      rep->rows.resize(n);
      rep->ids.resize(n);
      const uint64_t mark = ++rep->peek_mark;
      for (int64_t i = 0; i < n && early; ++i) {
        int64_t row = 0;
        early = rep->index->peek(workers[i], mark, &row, &rep->ids[i]);
        rep->rows[i] = static_cast<int32_t>(row);
      }
    }
    hp.lap(4, "early insert: peek");
    if (early) {
      plan.frame_pool = rep->keys[frame_key].pool;
      pre.src.assign(n_keys, nullptr);
      pre.src[frame_key] = frames;
      for (int k = 0; k < n_keys; ++k) {
        if (k == frame_key) continue;
        if (k == rep->key_stepid) {
          plan.stepid_pool = rep->keys[k].pool;
          pre.src[k] = rep->keys[k].pool;       // any non-null mark: step ids have no source buffer
          continue;
        }
        // Narrow observation keys the caller listed ride along; wide ones and
        // everything not listed (actions, agent outputs) wait for the publish.
        if (!src[k] || !rep->keys[k].pool || rep->keys[k].rowbytes > 256 ||
            plan.n_narrow >= emb::kPreNarrow)
          continue;
        plan.narrow[plan.n_narrow++] = {static_cast<const uint8_t*>(src[k]), rep->keys[k].pool,
                                        rep->keys[k].rowbytes};
        pre.src[k] = src[k];
      }
      early = emb::prewrite_supported(plan);
    }
    emb_replay::Carried& carried = rep->carry;
    if (carried.active && !(early && carried.stream == s && carried.n == n)) settle_carry(rep, predicted);
    if (!early) {
      HIP_OK(emb::launch_obs_stack(static_cast<const uint8_t*>(frames), nullptr, dst, n, spec->pixels,
                                   spec->channels, spec->layout, spec->out_dtype, spec->scale,
                                   spec->offset, s));
      return;
    }
    if (rep->dev_rows_cap < static_cast<size_t>(n)) {
      if (rep->dev_rows) HIP_OK(hipFree(rep->dev_rows));      // (synchronises: nothing still reads it)
      rep->dev_rows = nullptr;
      rep->dev_rows_cap = 0;
      size_t cap = 256;
      while (cap < static_cast<size_t>(n)) cap *= 2;
      HIP_OK(hipMalloc(reinterpret_cast<void**>(&rep->dev_rows), cap * sizeof(int32_t)));
      rep->dev_rows_cap = cap;
    }
    plan.rows_out = rep->dev_rows;
    if (carried.active) {            // the previous step's action rides in this launch
      plan.carry_src = static_cast<const uint8_t*>(carried.src);
      plan.carry_pool = carried.pool;
      plan.carry_flags = carried.flags;
      plan.carry_rowbytes = carried.rowbytes;
      plan.carry_dtype = carried.dtype;
      plan.carry_rows = carried.rows.data();
    }
    hp.lap(5, "early insert: key plan");
    // The per-env table goes to device memory: written by the CPU through the
    // BAR when it fits a slot of the argument ring, else staged and copied.
    const size_t bytes = emb::prewrite_table_bytes(n);
    const uint8_t* ids = reinterpret_cast<const uint8_t*>(rep->ids.data());
    TableRing::Lease lease{-1, nullptr, nullptr};
    bool in_bar = false;
    ArgRing& arg_ring = rep->args_for(s);
    if (bytes <= ArgRing::kSlotBytes && arg_ring.usable()) {
      uint8_t* slot = arg_ring.take(s);
      emb::prewrite_fill_table(slot, plan, rep->rows.data(), ids);
      ArgRing::publish();
      plan.table_dev = slot;
      in_bar = true;
    } else {
      lease = rep->ring.acquire(bytes, s);
      emb::prewrite_fill_table(lease.host, plan, rep->rows.data(), ids);
      rep->ring.upload(lease, bytes, s);
      plan.table_dev = lease.device;
    }
    hp.lap(6, "early insert: table -> device");
    rep->order_before(StreamOrder::kWriteFresh, s, predicted);
    HIP_OK(emb::launch_obs_stack_insert(plan, s, write_stamp(rep)));
    rep->order_after(StreamOrder::kWriteFresh, s);
    if (carried.active) {
      carried.active = false;
      rep->carried_inline += 1;
    }
    hp.lap(7, "early insert: launch");
    if (in_bar) arg_ring.retire(s);
    if (lease.slot >= 0) rep->ring.retire(lease, s);
    pre.workers.assign(workers, workers + n);
    pre.rows = rep->rows;
    pre.stream = s;
    pre.token = ++rep->pre_serial;
    *token_out = pre.token;
    hp.lap(8, "early insert: retire + record");
  });
}

static void sample_locked(emb_replay* rep, int64_t batch, int32_t mode, void* const* dst,
                          int32_t group, int64_t group_stride, uint8_t* online_out,
                          uint8_t* first_stepids_out, hipStream_t stream,
                          const int32_t* key_len = nullptr) {
  need(batch >= 0 && dst, "sample: bad arguments");
  need(!rep->keys.empty(), "sample: call emb_replay_set_keys first");
  need(group >= 0 && group_stride >= 0 && (group == 0 || group_stride % 16 == 0),
       "sample: bad destination groups");
  if (batch == 0) return;
  const int64_t L = rep->index->config().length;
  KeyList list;
  for (size_t k = 0; k < rep->keys.size(); ++k) {
    need(dst[k] && rep->keys[k].pool, "sample: null buffer");
    if (static_cast<int>(k) == rep->key_is_first) list.key_is_first = static_cast<int>(list.key.size());
    if (static_cast<int>(k) == rep->key_is_last) list.key_is_last = static_cast<int>(list.key.size());
    need(!key_len || (key_len[k] >= 0 && key_len[k] <= L), "sample: a key's head is longer than the sequence");
    list.push(rep->keys[k].pool, dst[k], rep->keys[k].rowbytes, key_len ? key_len[k] : 0);
  }
  list.seq_len = static_cast<int32_t>(L);
  list.group = group;
  list.group_stride = group_stride;
  HostLap hp;
  rep->rows.resize(batch * L);
  sample_index_locked(rep, batch, mode, rep->rows.data(), online_out, &rep->spans, first_stepids_out);
  if (rep->carry.active) {
    // A carried publish holds the NEWEST step of every worker stream: a sampled
    // window reads one of its rows only as its own last row.  Settle the carry
    // (a launch) only then; otherwise it stays for the next early insert.
    const auto& newest = rep->carry.sorted;
    bool hit = false;
    for (int64_t b = 0; b < batch && !hit; ++b)
      hit = std::binary_search(newest.begin(), newest.end(), rep->rows[b * L + L - 1]);
    if (hit) settle_carry(rep);
  }
  hp.lap(9, "sample: index draws + spans");
  run_move_all(rep, list, rep->rows.data(), batch * L, nullptr, true, stream, &rep->spans);
  hp.lap(10, "sample: run_move (all of it)");
}

int32_t emb_replay_sample(emb_replay_t* rep, int64_t batch, int32_t mode, void* const* dst,
                          uint8_t* online_out, uint8_t* first_stepids_out, void* stream) {
  REP_OP(sample_locked(rep, batch, mode, dst, 0, 0, online_out, first_stepids_out,
                       static_cast<hipStream_t>(stream)));
}

int32_t emb_replay_sample_grouped(emb_replay_t* rep, int64_t batch, int32_t mode, void* const* dst,
                                  int32_t group, int64_t group_stride, uint8_t* online_out,
                                  uint8_t* first_stepids_out, void* stream) {
  REP_OP(sample_locked(rep, batch, mode, dst, group, group_stride, online_out, first_stepids_out,
                       static_cast<hipStream_t>(stream)));
}

int32_t emb_replay_sample_heads(emb_replay_t* rep, int64_t batch, int32_t mode, void* const* dst,
                                const int32_t* key_len, uint8_t* online_out,
                                uint8_t* first_stepids_out, void* stream) {
  REP_OP(sample_locked(rep, batch, mode, dst, 0, 0, online_out, first_stepids_out,
                       static_cast<hipStream_t>(stream), key_len));
}

static KeyList list_subset(emb_replay* rep, int32_t n_keys, const int32_t* key_ids,
                           const void* const* bufs) {
  KeyList list;
  need(n_keys >= 1 && key_ids && bufs, "bad key subset");
  for (int j = 0; j < n_keys; ++j) {
    need(key_ids[j] >= 0 && key_ids[j] < static_cast<int>(rep->keys.size()), "key id out of range");
    const KeyInfo& info = rep->keys[key_ids[j]];
    need(bufs[j] && info.pool, "null buffer");
    list.push(info.pool, bufs[j], info.rowbytes);
  }
  return list;
}

int32_t emb_replay_update(emb_replay_t* rep, int64_t B, int64_t T, const uint8_t* stepids,
                          int32_t n_keys, const int32_t* key_ids, const void* const* src,
                          void* stream) {
  REP_OP({
    need(B >= 0 && T >= 1 && stepids, "update: bad arguments");
    settle_carry(rep);
    if (B == 0) return;
    KeyList list = list_subset(rep, n_keys, key_ids, src);
    list.seq_len = static_cast<int32_t>(T);
    rep->rows.resize(B * T);
    rep->spans.resize(3 * B);
    bool compact = true;    // every window resolved into at most two runs of pool rows
    for (int64_t i = 0; i < B; ++i) {
      emb::ReplayIndex::Pos pos;
      if (rep->index->parse_stepid(stepids + i * EMB_STEPID_BYTES, &pos) &&
          rep->index->rows(pos, T, rep->rows.data() + i * T)) {
        compact = compact && rep->index->two_spans(pos, T, rep->spans.data() + 3 * i);
      } else {
        for (int64_t j = 0; j < T; ++j) rep->rows[i * T + j] = -1;
        compact = false;
      }
    }
    // The reference applies batch rows one after another (replay.py:139-149),
    // so when sampled windows overlap the LAST writer of a step wins.  One
    // launch has no order.  Usual case: no two windows share a pool row (checked
    // on the sorted runs) and the windows travel as spans in the kernel
    // arguments.  Otherwise drop all but the last occurrence of every pool row.
    if (compact) {
      auto& runs = rep->runs;
      runs.clear();
      for (int64_t i = 0; i < B; ++i) {
        const int32_t* sp = rep->spans.data() + 3 * i;
        runs.emplace_back(sp[0], sp[0] + sp[1]);
        if (sp[1] < T) runs.emplace_back(sp[2], sp[2] + static_cast<int32_t>(T) - sp[1]);
      }
      std::sort(runs.begin(), runs.end());
      for (size_t i = 1; i < runs.size() && compact; ++i) compact = runs[i].first >= runs[i - 1].second;
    }
    if (!compact) {
      rep->spans.clear();
      const size_t pool_rows = static_cast<size_t>(rep->index->config().n_slots * rep->index->config().chunksize);
      if (rep->stamp.size() < pool_rows) rep->stamp.resize(pool_rows, 0);
      if (++rep->stamp_epoch == 0) {       // wrapped: start over
        std::fill(rep->stamp.begin(), rep->stamp.end(), 0u);
        rep->stamp_epoch = 1;
      }
      for (int64_t i = B * T - 1; i >= 0; --i) {
        const int32_t row = rep->rows[i];
        if (row < 0) continue;
        if (rep->stamp[row] == rep->stamp_epoch) rep->rows[i] = -1;
        else rep->stamp[row] = rep->stamp_epoch;
      }
    }
    rep->timing_update = rep->timer_update.enabled;
    try {
      run_move_all(rep, list, rep->rows.data(), B * T, nullptr, false, static_cast<hipStream_t>(stream),
                   &rep->spans);
    } catch (...) {
      rep->timing_update = false;
      throw;
    }
    rep->timing_update = false;
  });
}

int32_t emb_replay_gather_rows(emb_replay_t* rep, const int32_t* rows, int64_t n_rows,
                               int64_t seq_len, void* const* dst, void* stream) {
  REP_OP({
    need(rows && n_rows >= 0 && dst && seq_len >= 1, "gather_rows: bad arguments");
    // (whole sequences: the is_last annotation of step t reads row t + 1 of its sequence)
    need(n_rows % seq_len == 0, "gather_rows: n_rows is not a multiple of seq_len");
    settle_carry(rep);
    if (n_rows == 0) return;
    KeyList list;
    for (size_t k = 0; k < rep->keys.size(); ++k) {
      if (!dst[k]) continue;                       // key not wanted in this gather
      need(rep->keys[k].pool, "gather_rows: key has no pool");
      if (static_cast<int>(k) == rep->key_is_first) list.key_is_first = static_cast<int>(list.key.size());
      if (static_cast<int>(k) == rep->key_is_last) list.key_is_last = static_cast<int>(list.key.size());
      list.push(rep->keys[k].pool, dst[k], rep->keys[k].rowbytes);
    }
    list.seq_len = static_cast<int32_t>(seq_len);
    if (list.key.empty()) return;
    // Sequences that are at most two contiguous runs of pool rows (windows that
    // cross one chunk boundary) travel as {row0, count0, row1} in the kernel
    // arguments instead of a row table in device memory.
    rep->spans.clear();
    if (n_rows % seq_len == 0) {
      const int64_t n_seq = n_rows / seq_len;
      rep->spans.resize(3 * n_seq);
      bool ok = true;
      for (int64_t q = 0; q < n_seq && ok; ++q) {
        const int32_t* r = rows + q * seq_len;
        int64_t cut = seq_len;
        for (int64_t j = 1; j < seq_len; ++j)
          if (r[j] != r[j - 1] + 1) { cut = j; break; }
        for (int64_t j = cut + 1; j < seq_len && ok; ++j) ok = r[j] == r[j - 1] + 1;
        ok = ok && r[0] >= 0 && (cut == seq_len || r[cut] >= 0);
        rep->spans[3 * q] = r[0];
        rep->spans[3 * q + 1] = static_cast<int32_t>(cut);
        rep->spans[3 * q + 2] = cut < seq_len ? r[cut] : 0;
      }
      if (!ok) rep->spans.clear();
    }
    run_move_all(rep, list, rows, n_rows, nullptr, true, static_cast<hipStream_t>(stream),
                 &rep->spans);
  });
}

int32_t emb_replay_scatter_rows(emb_replay_t* rep, const int32_t* rows, int64_t n_rows,
                                int32_t n_keys, const int32_t* key_ids, const void* const* src,
                                void* stream) {
  REP_OP({
    need(rows && n_rows >= 0, "scatter_rows: bad arguments");
    settle_carry(rep);
    if (n_rows == 0) return;
    KeyList list = list_subset(rep, n_keys, key_ids, src);
    run_move_all(rep, list, rows, n_rows, nullptr, false, static_cast<hipStream_t>(stream));
  });
}

int32_t emb_replay_profile(emb_replay_t* rep, int32_t enable) {
  REP_OP({
    rep->timer.enabled = enable != 0;
    rep->timer.every = enable > 1 ? enable : 1;      // enable = n > 1: stamp every n-th gather
    // ... starting with the n-th: the first launch after (re)starting the counter
    // is often the first one on an idle GPU, the worst sample there is
    rep->timer.tick = static_cast<uint64_t>(rep->timer.every - 1);
    rep->timer_update.enabled = rep->timer.enabled;  // write-backs of emb_replay_update alike
    rep->timer_update.every = rep->timer.every;
    rep->timer_update.tick = rep->timer.tick;
    if (enable) {                       // create the stamp pools now, not inside a timed region
      rep->timer.reserve(emb_timer_pool());
      rep->timer_update.reserve(emb_timer_pool());
      rep->timer_other.discard = true;
      rep->timer_other.reserve(256);
    }
  });
}

int32_t emb_replay_profile_report(emb_replay_t* rep, int32_t which, int64_t* launches, double* total_ms,
                                  int32_t reset, char* kernel_out, int32_t kernel_cap) {
  REP_OP({
    need(launches && total_ms && which >= 0 && which <= 3, "profile_report: bad arguments");
    static const std::string helper_name = "index bookkeeping on the helper thread";
    static const std::string carried_name = "publishes carried into the next early-insert launch (total_ms: of how many carried)";
    if (which == 2) {                    // not a kernel: publishes deferred to the helper thread
      *launches = rep->deferred_adds;
      *total_ms = static_cast<double>(rep->predicted_inserts);   // early inserts that did not wait for it
      if (reset) rep->deferred_adds = rep->predicted_inserts = 0;
    } else if (which == 3) {             // not a kernel either: carried publishes that rode along / all of them
      *launches = rep->carried_inline;
      *total_ms = static_cast<double>(rep->carried_total);
      if (reset) rep->carried_inline = rep->carried_total = 0;
    } else {
      (which == 0 ? rep->timer : rep->timer_update).read(launches, total_ms, reset != 0);
    }
    if (kernel_out && kernel_cap > 0) {
      const std::string& name = which == 2 ? helper_name : which == 3 ? carried_name : rep->timed_kernel[which];
      const size_t n = std::min<size_t>(name.size(), static_cast<size_t>(kernel_cap) - 1);
      std::memcpy(kernel_out, name.data(), n);
      kernel_out[n] = 0;
    }
  });
}

int32_t emb_replay_multistream(emb_replay_t* rep, int32_t enable) {
  REP_OP({
    // Pool accesses issued before the switch were not counted: let them finish
    // (once, when a second stream first appears).
    settle_carry(rep);
    if (enable && !rep->multistream) HIP_OK(hipDeviceSynchronize());
    rep->multistream = enable != 0;
  });
}

int32_t emb_replay_profile_read(emb_replay_t* rep, int64_t* launches, double* total_ms, int32_t reset) {
  REP_OP({
    need(launches && total_ms, "profile_read: bad arguments");
    rep->timer.read(launches, total_ms, reset != 0);
  });
}

int32_t emb_replay_complete_all(emb_replay_t* rep) { REP_OP(settle_carry(rep); rep->index->complete_all()); }

int32_t emb_replay_carry_publish(emb_replay_t* rep, int32_t enable) {
  REP_OP({
    if (!enable) settle_carry(rep);
    rep->carry_publish = enable != 0;
  });
}

int32_t emb_replay_settle(emb_replay_t* rep) { REP_OP(settle_carry(rep)); }
int32_t emb_replay_open_chunks(emb_replay_t* rep, int64_t* n) { REP_OP(need(n, "open_chunks: null output"); *n = rep->index->open_chunks()); }
int32_t emb_replay_reserve_uids(emb_replay_t* rep, uint64_t serial) { REP_OP(rep->index->reserve_uids(serial)); }

int32_t emb_replay_chunks(emb_replay_t* rep, int64_t cap, uint64_t* uid, uint64_t* succ,
                          int64_t* fill, int64_t* slot, int64_t* time_ms, int64_t* n) {
  REP_OP({
    need(n, "chunks: n is null");
    settle_carry(rep);
    int64_t i = 0;
    for (const auto& kv : rep->index->chunks()) {
      if (i < cap) {
        if (uid) uid[i] = kv.second.uid;
        if (succ) succ[i] = kv.second.succ;
        if (fill) fill[i] = kv.second.fill;
        if (slot) slot[i] = kv.second.slot;
        if (time_ms) time_ms[i] = kv.second.time_ms;
      }
      ++i;
    }
    *n = i;
  });
}

int32_t emb_replay_load_chunk(emb_replay_t* rep, uint64_t uid, uint64_t succ, int64_t fill,
                              int64_t time_ms, int64_t* slot) {
  REP_OP({
    need(slot, "load_chunk: slot is null");
    *slot = rep->index->load_chunk(uid, succ, fill, time_ms);
  });
}

int32_t emb_replay_load_items(emb_replay_t* rep, uint64_t uid, int64_t amount) {
  REP_OP(rep->index->load_items(uid, amount));
}

}  // extern "C"
