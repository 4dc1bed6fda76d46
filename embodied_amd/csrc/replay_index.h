// Host-side integer bookkeeping of the replay buffer: which device row every
// step lives in, which windows are sampleable items, FIFO eviction, chunk
// reference counts and the online queue.  Mirrors embodied/core/replay.py
// (add :77-118, _sample :151-169, _insert :171-179, _remove :181-191,
// _getseq :193-214, _complete :362-370) and chunk.py's fixed-size chunks, with
// one change of representation: a chunk's payload is not a numpy dict but one
// slot of a device pool laid out (n_slots, chunksize, rowbytes) per key, so a
// step is addressed by the global row  slot * chunksize + index.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <stdexcept>
#include <unordered_map>
#include <utility>
#include <vector>

#include "selectors.h"

namespace emb {

struct PoolFull : std::runtime_error {
  PoolFull() : std::runtime_error("replay chunk pool is full") {}
};

struct ReplayConfig {
  int64_t length = 1;
  int64_t capacity = 0;   // items; 0 = unbounded
  int64_t chunksize = 1024;
  int64_t n_slots = 0;    // chunk slots in the device pool
  bool online = false;
  uint64_t uid_hi = 0;    // high 64 bits of every chunk uid (replica id)
  // Sharded pools: worker w belongs to owner w / workers_per_owner and its chunks
  // only ever use that owner's contiguous range of n_slots / owners slots, so a
  // rank that holds one owner's range can address its rows without the others'.
  int64_t owners = 1;
  int64_t workers_per_owner = 0;
};

class ReplayIndex {
 public:
  struct Chunk {
    uint64_t uid = 0;
    uint64_t succ = 0;
    int64_t fill = 0;
    int64_t refs = 0;
    int64_t slot = -1;
    int64_t time_ms = 0;   // creation time (file naming / load order)
    int64_t worker = 0;    // the stream this chunk belongs to
  };
  struct Span { uint64_t uid; int64_t slot; int64_t index; int64_t count; };
  using Pos = std::pair<uint64_t, int64_t>;  // (chunk uid, row in chunk)

  ReplayIndex(const ReplayConfig& cfg, std::shared_ptr<Selector> selector)
      : cfg_(cfg), selector_(std::move(selector)) {
    if (cfg_.length < 1 || cfg_.chunksize < 1 || cfg_.n_slots < 1 || cfg_.capacity < 0)
      throw std::invalid_argument("replay: bad length/chunksize/n_slots/capacity");
    if (cfg_.n_slots * cfg_.chunksize > INT32_MAX)
      throw std::invalid_argument("replay: more than 2^31 rows in the pool");
    if (cfg_.owners < 1 || cfg_.n_slots % cfg_.owners != 0 ||
        (cfg_.owners > 1 && cfg_.workers_per_owner < 1))
      throw std::invalid_argument("replay: bad owners / workers_per_owner / n_slots");
    free_.resize(cfg_.owners);
    const int64_t per = cfg_.n_slots / cfg_.owners;
    for (int64_t s = 0; s < cfg_.n_slots; ++s) free_[s / per].push_back(s);
    used_.assign(static_cast<size_t>(cfg_.n_slots), 0);
  }

  const ReplayConfig& config() const { return cfg_; }
  Selector& selector() { return *selector_; }
  int64_t size() const { return next_item_ - first_item_; }
  // Free slots of the tightest owner (what bounds the next insert).
  int64_t free_slots() const {
    size_t least = free_[0].size();
    for (const auto& f : free_) least = f.size() < least ? f.size() : least;
    return static_cast<int64_t>(least);
  }
  int64_t owner_of(int64_t worker) const {
    if (cfg_.owners == 1 || worker < 0) return 0;
    const int64_t o = worker / cfg_.workers_per_owner;
    return o < cfg_.owners ? o : cfg_.owners - 1;
  }
  int64_t worker_of(const Pos& pos) const { return chunks_.at(pos.first).worker; }
  int64_t next_item() const { return next_item_; }
  // Chunks opened so far: a caller that batches payload writes flushes them
  // whenever this moves, so a recycled slot never sees two writers in one launch.
  int64_t chunks_opened() const { return static_cast<int64_t>(next_uid_) - 1 + loaded_; }
  // Of those, the ones that took a slot an earlier chunk had held: only such an
  // opening can make a caller's batched payload writes meet on a pool row (rows
  // still waiting for the old chunk, rows of the new one); a slot that was never
  // handed out has no rows waiting.
  int64_t recycled_opens() const { return recycled_opens_; }

  void grow(int64_t n_slots) {
    if (n_slots < cfg_.n_slots) throw std::invalid_argument("replay: pool cannot shrink");
    if (cfg_.owners != 1) throw std::invalid_argument("replay: a sharded pool cannot grow");
    if (n_slots * cfg_.chunksize > INT32_MAX)
      throw std::invalid_argument("replay: more than 2^31 rows in the pool");
    for (int64_t s = cfg_.n_slots; s < n_slots; ++s) free_[0].push_back(s);
    used_.resize(static_cast<size_t>(n_slots), 0);
    cfg_.n_slots = n_slots;
  }

  // Would `add` for these workers run out of slots?  Checked before any state
  // changes so the caller can grow the pool and retry.
  bool fits(const int64_t* workers, int64_t n) const {
    if (2 * n <= free_slots()) return true;   // a row opens at most two chunks
    std::vector<int64_t> need(cfg_.owners, 0);
    std::unordered_map<int64_t, int64_t> seen;  // worker -> simulated index
    for (int64_t i = 0; i < n; ++i) {
      const int64_t o = owner_of(workers[i]);
      auto s = seen.find(workers[i]);
      int64_t index;
      if (s != seen.end()) {
        index = s->second;
      } else {
        const Worker* w = find_worker(workers[i]);
        if (!w) { ++need[o]; index = 0; }
        else index = w->cursor.second;
      }
      ++index;
      if (index >= cfg_.chunksize) { ++need[o]; index = 0; }
      seen[workers[i]] = index;
    }
    for (int64_t o = 0; o < cfg_.owners; ++o)
      if (need[o] > static_cast<int64_t>(free_[o].size())) return false;
    return true;
  }

  // replay.py:77-118.  Returns the device row the step's payload goes to.
  int64_t add(int64_t worker, StepId* stepid) {
    Worker* w = find_worker(worker);
    if (!w) {
      Chunk& c = new_chunk(1, worker);
      w = &make_worker(worker);
      w->cursor = Pos(c.uid, 0);
    }
    const uint64_t uid = w->cursor.first;
    int64_t index = w->cursor.second;
    if (!w->open || w->open->uid != uid) w->open = &chunks_.at(uid);   // node addresses are stable
    Chunk& chunk = *w->open;
    if (chunk.fill != index) throw std::logic_error("replay: chunk cursor out of sync");
    *stepid = make_stepid(uid, index);
    const int64_t row = chunk.slot * cfg_.chunksize + index;
    chunk.fill += 1;
    // The steps that are not yet the start of an item are the last `length - 1`
    // of the stream: consecutive rows, so their oldest one (a position and its
    // chunk) and a count say it all.
    if (w->pending == 0) {
      w->oldest = Pos(uid, index);
      w->oldest_chunk = &chunk;
    }
    w->pending += 1;
    chunk.refs += 1;
    index += 1;
    if (index < cfg_.chunksize) w->cursor.second = index;
    else rotate(chunk, *w);
    if (w->pending >= cfg_.length) {
      metrics_[kInserts] += 1;
      const Pos start = w->oldest;
      w->pending -= 1;
      if (w->pending > 0) {
        // the stream's next step: the next row of the same chunk, or row 0 of
        // its successor (a chunk may have been closed early by a checkpoint)
        if (start.second + 1 < w->oldest_chunk->fill) {
          w->oldest.second = start.second + 1;
        } else {
          w->oldest_chunk = &chunks_.at(w->oldest_chunk->succ);      // node addresses are stable
          w->oldest = Pos(w->oldest_chunk->uid, 0);
        }
      }
      insert_item(start, w, stepid);
      if (cfg_.online && w->steps_seen % cfg_.length == 0) fresh_.push_back(start);
    }
    if (cfg_.online) w->steps_seen += 1;
    return row;
  }

  // The pool row and step id the NEXT add(worker) will use, without changing
  // anything: a worker's next step always goes to its open chunk's cursor (add
  // rotates eagerly, right after it fills a chunk's last row).  False for a
  // worker that has no chunk yet (its first add creates one) or that was
  // already peeked under the same `mark` (listed twice in one batch).
  bool peek(int64_t worker, uint64_t mark, int64_t* row, StepId* stepid) {
    Worker* w = find_worker(worker);
    if (!w || w->peek_mark == mark) return false;
    w->peek_mark = mark;
    const uint64_t uid = w->cursor.first;
    if (!w->open || w->open->uid != uid) w->open = &chunks_.at(uid);
    *row = w->open->slot * cfg_.chunksize + w->cursor.second;
    *stepid = make_stepid(uid, w->cursor.second);
    return true;
  }

  // Windows waiting in the online queue (replay.py:114-118): a train-mode
  // sample serves these before it asks the selector.
  int64_t online_pending() const { return static_cast<int64_t>(fresh_.size()); }

  // replay.py:151-169: one sequence start.  Stale online entries (first chunk
  // evicted) are dropped and the draw repeated, as the KeyError retry does.
  Pos draw(int mode_train, bool* from_online) {
    if (mode_train) metrics_[kSamples] += 1;
    for (;;) {
      Pos pos;
      if (cfg_.online && !fresh_.empty() && mode_train) {
        pos = fresh_.front();
        fresh_.pop_front();
        *from_online = true;
      } else {
        pos = item_at(selector_->sample());
        *from_online = false;
      }
      if (chunks_.count(pos.first)) return pos;
    }
  }

  // replay.py:193-214.  False if the first chunk is gone.
  bool spans(const Pos& pos, int64_t count, std::vector<Span>* out) const {
    out->clear();
    auto it = chunks_.find(pos.first);
    if (it == chunks_.end()) return false;
    const Chunk* chunk = &it->second;
    const int64_t have = chunk->fill - pos.second;
    if (have < 0) return false;
    if (have >= count) {
      out->push_back({chunk->uid, chunk->slot, pos.second, count});
      return true;
    }
    out->push_back({chunk->uid, chunk->slot, pos.second, have});
    int64_t left = count - have;
    while (left > 0) {
      auto nx = chunks_.find(chunk->succ);
      if (nx == chunks_.end()) return false;
      chunk = &nx->second;
      const int64_t used = left < chunk->fill ? left : chunk->fill;
      if (used == 0) return false;
      out->push_back({chunk->uid, chunk->slot, 0, used});
      left -= used;
    }
    return true;
  }

  // Device rows of `count` consecutive steps from `pos`; false if evicted (rows
  // are then filled with -1, which the scatter kernel skips).
  bool rows(const Pos& pos, int64_t count, int32_t* out) const {
    if (!spans(pos, count, &scratch_)) {
      for (int64_t i = 0; i < count; ++i) out[i] = -1;
      return false;
    }
    int64_t at = 0;
    for (const Span& s : scratch_)
      for (int64_t i = 0; i < s.count; ++i)
        out[at++] = static_cast<int32_t>(s.slot * cfg_.chunksize + s.index + i);
    return true;
  }

  // {row0, count0, row1} if the window crosses at most one chunk boundary.
  bool two_spans(const Pos& pos, int64_t count, int32_t out[3]) const {
    if (!spans(pos, count, &scratch_) || scratch_.size() > 2) return false;
    out[0] = static_cast<int32_t>(scratch_[0].slot * cfg_.chunksize + scratch_[0].index);
    out[1] = static_cast<int32_t>(scratch_[0].count);
    out[2] = scratch_.size() > 1 ? static_cast<int32_t>(scratch_[1].slot * cfg_.chunksize) : 0;
    return true;
  }

  // 16-byte big-endian chunk uid | 4-byte big-endian row (replay.py:90-91).
  StepId make_stepid(uint64_t uid, int64_t index) const {
    StepId s;
    const uint64_t hi = __builtin_bswap64(cfg_.uid_hi), lo = __builtin_bswap64(uid);
    const uint32_t row = __builtin_bswap32(static_cast<uint32_t>(index));
    std::memcpy(s.b, &hi, 8);
    std::memcpy(s.b + 8, &lo, 8);
    std::memcpy(s.b + 16, &row, 4);
    return s;
  }

  // replay.py:141-144.  False when the id was not issued by this replay.
  bool parse_stepid(const uint8_t* b, Pos* pos) const {
    uint64_t hi = 0, lo = 0;
    uint32_t idx = 0;
    for (int i = 0; i < 8; ++i) hi = (hi << 8) | b[i];
    for (int i = 0; i < 8; ++i) lo = (lo << 8) | b[8 + i];
    for (int i = 0; i < 4; ++i) idx = (idx << 8) | b[16 + i];
    *pos = Pos(lo, static_cast<int64_t>(idx));
    return hi == cfg_.uid_hi;
  }

  void count_updates(int64_t n) { metrics_[kUpdates] += n; }

  // replay.py:58-74 (ram_gb is the caller's: it owns the device pool).
  void stats(int64_t out[6], bool reset) {
    out[0] = size();
    out[1] = static_cast<int64_t>(chunks_.size());
    out[2] = static_cast<int64_t>(workers_.size());
    out[3] = metrics_[kInserts];
    out[4] = metrics_[kSamples];
    out[5] = metrics_[kUpdates];
    if (reset) metrics_[0] = metrics_[1] = metrics_[2] = 0;
  }

  const std::unordered_map<uint64_t, Chunk>& chunks() const { return chunks_; }

  // Checkpoint support (replay.py:295-359): close every worker's open chunk.
  // All or nothing: every rotation opens a successor chunk, so the free slots
  // are counted per owner BEFORE any worker is touched (PoolFull leaves the
  // index unchanged; the caller grows the pool and retries).
  void complete_all() {
    std::vector<int64_t> need(cfg_.owners, 0);
    for (auto& kv : workers_)
      if (chunks_.at(kv.second->cursor.first).fill > 0) ++need[owner_of(kv.first)];
    for (int64_t o = 0; o < cfg_.owners; ++o)
      if (need[o] > static_cast<int64_t>(free_[o].size())) throw PoolFull();
    for (auto& kv : workers_) {
      Worker& w = *kv.second;
      Chunk& chunk = chunks_.at(w.cursor.first);
      if (chunk.fill > 0) rotate(chunk, w);
    }
  }
  // Open (non-empty) chunks complete_all would close = slots it needs.
  int64_t open_chunks() const {
    int64_t n = 0;
    for (auto& kv : workers_)
      if (chunks_.at(kv.second->cursor.first).fill > 0) ++n;
    return n;
  }
  // Chunk serials below `serial` are taken (files already on disk).
  void reserve_uids(uint64_t serial) {
    if (serial > next_uid_) next_uid_ = serial;
  }

  // Re-create a saved chunk (replay.py:347-359): returns its slot.
  int64_t load_chunk(uint64_t uid, uint64_t succ, int64_t fill, int64_t time_ms = 0) {
    if (cfg_.owners != 1)
      throw std::invalid_argument("replay: a sharded pool has no checkpoint path (load_chunk)");
    if (chunks_.count(uid)) throw std::runtime_error("replay: chunk already loaded");
    if (free_[0].empty()) throw PoolFull();
    Chunk c;
    c.uid = uid;
    c.succ = succ;
    c.fill = fill;
    c.refs = 0;
    c.time_ms = time_ms;
    c.slot = free_[0].front();
    free_[0].pop_front();
    if (used_[c.slot]) ++recycled_opens_;
    used_[c.slot] = 1;
    chunks_[uid] = c;
    ++loaded_;
    if (uid >= next_uid_) next_uid_ = uid + 1;
    return c.slot;
  }

  void load_items(uint64_t uid, int64_t amount) {
    Chunk& c = chunks_.at(uid);
    c.refs += amount;
    auto nx = chunks_.find(c.succ);
    if (nx != chunks_.end()) nx->second.refs += 1;
    for (int64_t i = 0; i < amount; ++i) insert_item(Pos(uid, i));
  }

 private:
  enum { kInserts = 0, kSamples = 1, kUpdates = 2 };

  Chunk& new_chunk(int64_t refs, int64_t worker) {
    auto& free = free_[owner_of(worker)];
    if (free.empty()) throw PoolFull();
    Chunk c;
    c.uid = next_uid_++;
    c.refs = refs;
    c.time_ms = std::chrono::duration_cast<std::chrono::milliseconds>(
        std::chrono::system_clock::now().time_since_epoch()).count();
    c.slot = free.front();
    free.pop_front();
    if (used_[c.slot]) ++recycled_opens_;
    used_[c.slot] = 1;
    c.worker = worker;
    return chunks_[c.uid] = c;
  }

  struct Worker {
    Pos cursor;                 // (open chunk uid, next row)
    Chunk* open = nullptr;      // cached node of the open chunk
    Pos oldest;                 // the oldest step that is not yet the start of an item ...
    Chunk* oldest_chunk = nullptr;   // ... its chunk ...
    int64_t pending = 0;        // ... and how many such steps there are (they are consecutive)
    int64_t steps_seen = 0;     // online mode
    uint64_t peek_mark = 0;     // last peek() batch that listed this worker
    int64_t last_item = -1;     // key of the newest item of this stream
  };

  // Worker ids are usually 0..N-1: a flat table in front of the general map.
  Worker* find_worker(int64_t id) const {
    if (id >= 0 && id < static_cast<int64_t>(dense_.size())) return dense_[id];
    auto it = workers_.find(id);
    return it == workers_.end() ? nullptr : it->second;
  }
  Worker& make_worker(int64_t id) {
    // (records of workers made one after another sit next to each other: a
    // vectorised step touches a few KB of them, not one allocation per worker)
    worker_pool_.emplace_back();
    Worker* slot = workers_[id] = &worker_pool_.back();
    if (id >= 0 && id < 65536) {
      if (id >= static_cast<int64_t>(dense_.size())) dense_.resize(id + 1, nullptr);
      dense_[id] = slot;
    }
    return *slot;
  }

  // itemids are handed out consecutively and evicted oldest-first, so the live
  // ones are always the contiguous range [first_item_, next_item_).
  const Pos& item_at(int64_t key) const {
    if (key < first_item_ || key >= next_item_) throw std::out_of_range("replay: unknown item");
    return items_[static_cast<size_t>(key - first_item_)];
  }

  // replay.py:362-370
  void rotate(Chunk& chunk, Worker& worker) {
    const uint64_t old = chunk.uid;
    Chunk& succ = new_chunk(2, chunk.worker);
    Chunk& prev = chunks_.at(old);
    prev.refs -= 1;
    prev.succ = succ.uid;
    worker.cursor = Pos(succ.uid, 0);
    worker.open = &succ;
  }

  // replay.py:171-179
  // `worker` / `newest` (add() only): the worker stream this window continues and
  // the id of its last step, for selectors that take sliding windows one new
  // step at a time (Selector::insert_successor).
  void insert_item(const Pos& start, Worker* worker = nullptr, const StepId* newest = nullptr) {
    while (cfg_.capacity && size() >= cfg_.capacity) evict();
    const int64_t key = next_item_++;
    items_.push_back(start);
    const int64_t prev_key = worker ? worker->last_item : -1;
    if (worker) worker->last_item = key;
    if (selector_->needs_stepids()) {
      if (prev_key >= first_item_ && newest && selector_->insert_successor(key, prev_key, *newest)) return;
      if (!spans(start, cfg_.length, &scratch_))
        throw std::logic_error("replay: inserted window is incomplete");
      ids_.clear();
      for (const Span& s : scratch_)
        for (int64_t i = 0; i < s.count; ++i) ids_.push_back(make_stepid(s.uid, s.index + i));
      selector_->insert(key, ids_.data(), static_cast<int>(ids_.size()));
    } else {
      selector_->insert(key, nullptr, 0);
    }
  }

  // replay.py:181-191
  void evict() {
    const int64_t key = first_item_;
    selector_->remove(key);
    const uint64_t uid = items_.front().first;
    items_.pop_front();
    ++first_item_;
    // With N workers consecutive evictions cycle through N chunks: a small
    // direct-mapped table of nodes (stable addresses) instead of a hash lookup
    // per eviction.
    Chunk*& hint = evict_hint_[uid & (kHints - 1)];
    if (!hint || hint->uid != uid) hint = &chunks_.at(uid);
    Chunk& chunk = *hint;
    chunk.refs -= 1;
    if (chunk.refs < 1) {
      const uint64_t succ = chunk.succ;
      free_[chunk.slot / (cfg_.n_slots / cfg_.owners)].push_back(chunk.slot);
      hint = nullptr;
      chunks_.erase(uid);
      auto nx = chunks_.find(succ);
      if (nx != chunks_.end()) nx->second.refs -= 1;
    }
  }

  ReplayConfig cfg_;
  std::shared_ptr<Selector> selector_;
  std::unordered_map<uint64_t, Chunk> chunks_;
  std::vector<std::deque<int64_t>> free_;   // per owner, FIFO: a freed slot is recycled as late as possible
  std::vector<uint8_t> used_;               // per slot: some chunk has held it
  int64_t recycled_opens_ = 0;
  Ring<Pos> items_;
  int64_t first_item_ = 0;
  int64_t next_item_ = 0;
  uint64_t next_uid_ = 1;
  int64_t loaded_ = 0;
  std::deque<Worker> worker_pool_;                      // stable addresses, contiguous blocks
  std::unordered_map<int64_t, Worker*> workers_;
  std::vector<Worker*> dense_;
  Ring<Pos> fresh_;
  static constexpr uint64_t kHints = 256;
  Chunk* evict_hint_[kHints] = {};
  int64_t metrics_[3] = {0, 0, 0};
  mutable std::vector<Span> scratch_;
  std::vector<StepId> ids_;
};

}  // namespace emb
