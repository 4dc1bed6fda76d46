// Host index samplers, integer/float64 exact restatements of the reference's
// embodied/core/selectors.py (Fifo :7-26, Uniform :29-57, Prioritized :128-197,
// Mixture :200-228, SampleTree :231-354).  Pure host C++: this is bookkeeping,
// O(batch) per sample, and must reproduce the reference's numpy streams bit for
// bit; payload movement is what runs on the GPU.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <stdexcept>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "knobs.h"
#include "np_random.h"

namespace emb {

constexpr int kStepIdBytes = 20;

// A queue with indexing, for the hot paths of the index (items, the Uniform
// selector's key window, the online queue): a power-of-two ring over one array,
// so push_back / pop_front / operator[] are a mask and an add each
// (std::deque pays a division and two indirections per access, which showed up
// as a third of the per-step bookkeeping time).
template <typename T>
class Ring {
 public:
  bool empty() const { return count_ == 0; }
  size_t size() const { return count_; }
  T& operator[](size_t i) { return data_[(head_ + i) & mask_]; }
  const T& operator[](size_t i) const { return data_[(head_ + i) & mask_]; }
  T& front() { return data_[head_]; }
  const T& front() const { return data_[head_]; }
  void push_back(const T& v) {
    if (count_ == data_.size()) grow();
    data_[(head_ + count_) & mask_] = v;
    ++count_;
  }
  void pop_front() {
    head_ = (head_ + 1) & mask_;
    --count_;
  }
  void clear() { head_ = count_ = 0; }

 private:
  void grow() {
    const size_t cap = data_.empty() ? 64 : data_.size() * 2;
    std::vector<T> bigger(cap);
    for (size_t i = 0; i < count_; ++i) bigger[i] = data_[(head_ + i) & mask_];
    data_.swap(bigger);
    head_ = 0;
    mask_ = cap - 1;
  }
  std::vector<T> data_;
  size_t head_ = 0, count_ = 0, mask_ = 0;
};

struct StepId {
  uint8_t b[kStepIdBytes];
  bool operator==(const StepId& o) const { return !std::memcmp(b, o.b, kStepIdBytes); }
};

struct StepIdHash {
  size_t operator()(const StepId& s) const {
    uint64_t x, y;
    uint32_t z;
    std::memcpy(&x, s.b, 8);
    std::memcpy(&y, s.b + 8, 8);
    std::memcpy(&z, s.b + 16, 4);
    uint64_t h = x * 0x9E3779B97F4A7C15ull;
    h ^= (y + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
    h ^= (static_cast<uint64_t>(z) * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2));
    return static_cast<size_t>(h);
  }
};

// int64 key -> V for keys that arrive in increasing order and mostly leave
// oldest first (item ids of a Replay): a ring indexed by key - base serves them
// without hashing; any other pattern falls back to a hash map for good, with
// the same contents.  `V()` must mean "absent" (a null pointer / null first).
template <typename V>
class SlidingMap {
 public:
  static bool present(const V& v) { return !(v == V()); }
  size_t size() const { return live_; }
  bool empty() const { return live_ == 0; }
  V* find(int64_t key) {
    if (dense_) {
      const int64_t i = key - base_;
      if (i < 0 || i >= static_cast<int64_t>(window_.size()) || !present(window_[i])) return nullptr;
      return &window_[i];
    }
    auto it = sparse_.find(key);
    return it == sparse_.end() ? nullptr : &it->second;
  }
  bool count(int64_t key) { return find(key) != nullptr; }
  void put(int64_t key, const V& value) {
    if (dense_) {
      if (window_.empty()) base_ = key;
      const int64_t i = key - base_;
      if (i >= 0 && i < static_cast<int64_t>(window_.size())) {
        if (!present(window_[i])) ++live_;
        window_[i] = value;
        return;
      }
      if (i == static_cast<int64_t>(window_.size())) {
        window_.push_back(value);
        ++live_;
        return;
      }
      for (size_t j = 0; j < window_.size(); ++j)        // pattern broken: migrate
        if (present(window_[j])) sparse_[base_ + static_cast<int64_t>(j)] = window_[j];
      window_.clear();
      dense_ = false;
    }
    if (sparse_.emplace(key, value).second) ++live_;
    else sparse_[key] = value;
  }
  bool erase(int64_t key) {
    if (dense_) {
      V* slot = find(key);
      if (!slot) return false;
      *slot = V();
      --live_;
      while (!window_.empty() && !present(window_.front())) {
        window_.pop_front();
        ++base_;
      }
      return true;
    }
    if (!sparse_.erase(key)) return false;
    --live_;
    return true;
  }
  void clear() {
    window_.clear();
    sparse_.clear();
    live_ = 0;
    dense_ = true;
  }
  template <typename Fn>
  void for_each(Fn&& fn) const {
    if (dense_) {
      for (size_t j = 0; j < window_.size(); ++j)
        if (present(window_[j])) fn(base_ + static_cast<int64_t>(j), window_[j]);
    } else {
      for (const auto& kv : sparse_) fn(kv.first, kv.second);
    }
  }

 private:
  bool dense_ = true;
  int64_t base_ = 0;
  size_t live_ = 0;
  Ring<V> window_;
  std::unordered_map<int64_t, V> sparse_;
};

class Selector {
 public:
  virtual ~Selector() = default;
  virtual int64_t sample() = 0;
  virtual int64_t size() const = 0;
  virtual void insert(int64_t key, const StepId* steps, int n) = 0;
  virtual void remove(int64_t key) = 0;
  // Shortcut for sliding windows: item `key` is item `prev_key` moved on by one
  // step, its last step is `newest` (all its other steps are prev_key's steps
  // but the first).  True if the selector took it; false = not handled, nothing
  // changed, the caller inserts the full step list.
  virtual bool insert_successor(int64_t key, int64_t prev_key, const StepId& newest) { return false; }
  // False if insert() ignores the step ids (lets the replay skip building them).
  virtual bool needs_stepids() const { return true; }
  virtual bool can_prioritize() const { return false; }
  virtual void prioritize(const StepId*, const double*, int64_t) {}
};

class Fifo : public Selector {
 public:
  int64_t sample() override {
    if (q_.empty()) throw std::runtime_error("Fifo: empty");
    return q_.front();
  }
  int64_t size() const override { return static_cast<int64_t>(q_.size()); }
  bool needs_stepids() const override { return false; }
  void insert(int64_t key, const StepId*, int) override { q_.push_back(key); }
  void remove(int64_t key) override {
    if (!q_.empty() && q_.front() == key) {
      q_.pop_front();
      return;
    }
    auto it = std::find(q_.begin(), q_.end(), key);
    if (it == q_.end()) throw std::runtime_error("Fifo: unknown key");
    q_.erase(it);
  }

 private:
  std::deque<int64_t> q_;
};

class Uniform : public Selector {
 public:
  explicit Uniform(uint64_t seed) : rng_(seed) {}
  int64_t sample() override {
    if (keys_.empty()) throw std::runtime_error("Uniform: empty");
    return keys_[rng_.integers(static_cast<int64_t>(keys_.size()))];
  }
  int64_t size() const override { return static_cast<int64_t>(keys_.size()); }
  bool needs_stepids() const override { return false; }
  void insert(int64_t key, const StepId*, int) override {
    set_pos(key, static_cast<int64_t>(keys_.size()));
    keys_.push_back(key);
  }
  // Swap-with-last.  The reference asserts len >= 2 here (selectors.py:52),
  // which makes a capacity-1 replay raise on its second insert although its own
  // test expects it to work (tests/test_replay.py:49); we allow it.
  void remove(int64_t key) override {
    const int64_t at = take_pos(key);
    const int64_t tail = keys_.back();
    keys_.pop_back();
    if (at != static_cast<int64_t>(keys_.size())) {
      keys_[at] = tail;
      set_pos(tail, at);
    }
  }
  const std::vector<int64_t>& keys() const { return keys_; }

 private:
  // key -> position in keys_.  Replay hands out consecutive item ids and evicts
  // the oldest, so the live keys form a sliding window: a deque indexed by
  // key - base_ serves them without hashing; any other key pattern falls back
  // to the hash map for good.
  static constexpr int64_t kGone = -1;
  void set_pos(int64_t key, int64_t at) {
    if (dense_) {
      if (window_.empty()) base_ = key;
      const int64_t i = key - base_;
      if (i >= 0 && i < static_cast<int64_t>(window_.size())) { window_[i] = at; return; }
      if (i == static_cast<int64_t>(window_.size())) { window_.push_back(at); return; }
      for (size_t j = 0; j < window_.size(); ++j)      // pattern broken: migrate
        if (window_[j] != kGone) sparse_[base_ + static_cast<int64_t>(j)] = window_[j];
      window_.clear();
      dense_ = false;
    }
    sparse_[key] = at;
  }
  int64_t take_pos(int64_t key) {
    if (dense_) {
      const int64_t i = key - base_;
      if (i < 0 || i >= static_cast<int64_t>(window_.size()) || window_[i] == kGone)
        throw std::runtime_error("Uniform: unknown key");
      const int64_t at = window_[i];
      window_[i] = kGone;
      while (!window_.empty() && window_.front() == kGone) {
        window_.pop_front();
        ++base_;
      }
      return at;
    }
    auto it = sparse_.find(key);
    if (it == sparse_.end()) throw std::runtime_error("Uniform: unknown key");
    const int64_t at = it->second;
    sparse_.erase(it);
    return at;
  }

  NpRandom rng_;
  std::vector<int64_t> keys_;
  bool dense_ = true;
  int64_t base_ = 0;
  Ring<int64_t> window_;
  std::unordered_map<int64_t, int64_t> sparse_;
};

// b-ary sum tree (selectors.py:231-354).  Children keep list order because the
// draw at each level indexes them and float sums run left to right.
class SampleTree {
 public:
  // A node keeps its children's masses next to each other (`kid_mass[i]` ==
  // `kids[i]->mass`, always): re-summing a node and drawing among its children
  // read one short array instead of following sixteen pointers.
  struct Node {
    Node* up = nullptr;
    bool leaf = false;
    int32_t slot = 0;           // position among `up`'s children
    int64_t key = 0;
    double mass = 0.0;
    std::vector<Node*> kids;
    std::vector<double> kid_mass;
  };

  SampleTree(int branching, uint64_t seed) : branching_(branching), rng_(seed) {
    if (branching < 2) throw std::invalid_argument("SampleTree: branching < 2");
    root_ = make();
  }
  ~SampleTree() {
    destroy(root_);
    for (Node* node : spare_) delete node;
  }
  SampleTree(const SampleTree&) = delete;
  SampleTree& operator=(const SampleTree&) = delete;

  int64_t size() const { return static_cast<int64_t>(leaves_.size()); }
  double root_mass() const { return root_->mass; }
  const Node* root() const { return root_; }

  // Returns the leaf: it stays the same node until `remove(key)`.
  Node* insert(int64_t key, double mass) {
    if (leaves_.count(key)) throw std::runtime_error("SampleTree: duplicate key");
    Node* spot;
    if (!tail_) {
      spot = root_;
    } else {
      int climbed = 0;
      spot = tail_->up;
      while (spot && static_cast<int>(spot->kids.size()) >= branching_) {
        spot = spot->up;
        ++climbed;
      }
      if (!spot) {
        spot = make();
        attach(spot, root_);
        root_ = spot;
      }
      for (int i = 0; i < climbed; ++i) {
        Node* fresh = make();
        attach(spot, fresh);
        spot = fresh;
      }
    }
    Node* leaf = make();
    leaf->leaf = true;
    leaf->key = key;
    leaf->mass = mass;
    attach(spot, leaf);
    leaves_.put(key, leaf);
    tail_ = leaf;
    return leaf;
  }

  void remove(int64_t key) {
    Node** found = leaves_.find(key);
    if (!found) throw std::out_of_range("SampleTree: unknown key");
    Node* leaf = *found;
    leaves_.erase(key);
    Node* hole_parent = leaf->up;
    Node* tail_parent = tail_->up;
    detach(hole_parent, leaf);
    if (leaf != tail_) attach(hole_parent, tail_);
    Node* node = tail_parent;
    while (node->up && node->kids.empty()) {
      Node* above = node->up;
      detach(above, node);
      recycle(node);
      node = above;
    }
    recycle(leaf);
    if (node->kids.empty()) {
      tail_ = nullptr;
      return;
    }
    while (!node->leaf) node = node->kids.back();
    tail_ = node;
  }

  void update(int64_t key, double mass) {
    Node** found = leaves_.find(key);
    if (!found) throw std::out_of_range("SampleTree: unknown key");
    set_mass(*found, mass);
    resum((*found)->up);
  }

  // Several leaves at once: set every mass, then re-sum each ancestor (a node
  // is the fresh sum of its children, so the result is what one `update` per
  // leaf leaves behind; neighbouring leaves share their ancestors).
  void update_many(const int64_t* keys, const double* masses, int64_t n) {
    handles_.resize(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; ++i) {
      Node** found = leaves_.find(keys[i]);
      if (!found) throw std::out_of_range("SampleTree: unknown key");
      handles_[static_cast<size_t>(i)] = *found;
    }
    update_leaves(handles_.data(), masses, n);
  }
  void update_leaves(Node* const* leaves, const double* masses, int64_t n) {
    dirty_.clear();
    for (int64_t i = 0; i < n; ++i) {
      set_mass(leaves[i], masses[i]);
      if (leaves[i]->up) dirty_.push_back(leaves[i]->up);
    }
    while (!dirty_.empty()) {
      std::sort(dirty_.begin(), dirty_.end());
      dirty_.erase(std::unique(dirty_.begin(), dirty_.end()), dirty_.end());
      next_.clear();
      for (Node* node : dirty_) {
        double total = 0.0;
        for (double m : node->kid_mass) total += m;
        if (same_bits(total, node->mass)) continue;      // ancestors stay as they are
        set_mass(node, total);
        if (node->up) next_.push_back(node->up);
      }
      dirty_.swap(next_);
    }
  }

  int64_t sample() {
    if (leaves_.empty()) throw std::runtime_error("SampleTree: empty");
    Node* node = root_;
    while (!node->leaf) {
      const int k = static_cast<int>(node->kids.size());
      prob_.resize(k);
      cdf_.resize(k);
      const double* mass_ = node->kid_mass.data();
      const double total = np_pairwise_sum(mass_, k);
      if (!std::isfinite(total)) {
        int hot = 0;
        for (int i = 0; i < k; ++i) hot += std::isinf(mass_[i]) ? 1 : 0;
        for (int i = 0; i < k; ++i)
          prob_[i] = (std::isinf(mass_[i]) ? 1.0 : 0.0) / static_cast<double>(hot);
      } else if (total == 0) {
        for (int i = 0; i < k; ++i) prob_[i] = 1.0 / static_cast<double>(k);
      } else {
        for (int i = 0; i < k; ++i) prob_[i] = mass_[i] / total;
      }
      node = node->kids[rng_.choice(prob_.data(), k, cdf_.data())];
    }
    return node->key;
  }

 private:
  // A replay at capacity inserts and removes one leaf per step and worker: nodes
  // are taken from and handed back to a free list instead of the allocator.
  Node* make() {
    if (spare_.empty()) return new Node();
    Node* node = spare_.back();
    spare_.pop_back();
    return node;
  }
  void recycle(Node* node) {
    node->up = nullptr;
    node->leaf = false;
    node->slot = 0;
    node->key = 0;
    node->mass = 0.0;
    node->kids.clear();
    node->kid_mass.clear();
    if (spare_.size() < 4096) spare_.push_back(node);
    else delete node;
  }
  // Re-sum `node` and its ancestors.  A node whose fresh sum is bit for bit the
  // mass it holds leaves every ancestor as it is (each is the left-to-right sum
  // of its children's masses): the walk stops there.  With +inf masses around
  // -- `initial: inf` -- that is usually the first level.
  static bool same_bits(double a, double b) {
    uint64_t x, y;
    std::memcpy(&x, &a, 8);
    std::memcpy(&y, &b, 8);
    return x == y;
  }
  static void set_mass(Node* node, double mass) {
    node->mass = mass;
    if (node->up) node->up->kid_mass[static_cast<size_t>(node->slot)] = mass;
  }
  static void resum(Node* node) {
    while (node) {
      double total = 0.0;
      for (double m : node->kid_mass) total += m;
      if (same_bits(total, node->mass)) return;
      set_mass(node, total);
      node = node->up;
    }
  }
  static void detach(Node* parent, Node* child) {
    child->up = nullptr;
    auto& kids = parent->kids;
    const auto at = static_cast<std::ptrdiff_t>(child->slot);
    kids.erase(kids.begin() + at);
    parent->kid_mass.erase(parent->kid_mass.begin() + at);
    for (size_t i = static_cast<size_t>(at); i < kids.size(); ++i) kids[i]->slot = static_cast<int32_t>(i);
    resum(parent);
  }
  static void attach(Node* parent, Node* child) {
    if (child->up) detach(child->up, child);
    child->up = parent;
    child->slot = static_cast<int32_t>(parent->kids.size());
    parent->kids.push_back(child);
    parent->kid_mass.push_back(child->mass);
    resum(parent);
  }
  static void destroy(Node* node) {
    for (Node* kid : node->kids) destroy(kid);
    delete node;
  }

  int branching_;
  NpRandom rng_;
  Node* root_;
  Node* tail_ = nullptr;
  SlidingMap<Node*> leaves_;
  std::vector<double> prob_, cdf_;
  std::vector<Node*> dirty_, next_, handles_, spare_;
};

// Priority-proportional sampling over per-step priorities (selectors.py:128-197).
//
// The reference keeps, per item, the list of its step ids, per step id a
// priority and the list of items that contain it, and re-aggregates an item
// from its steps' priorities whenever one of them changes.  A Replay feeds this
// selector sliding windows: item k+1 of a worker stream is item k shifted by
// one step, and items leave oldest-first.  For that pattern ("stream mode") the
// steps of a stream live in ONE contiguous array, items are just their start
// position, "the items that contain step p" is the index range [p-n+1, p], and
// an insert touches one new step instead of n records.  Anything else -- items
// with arbitrary step lists, out-of-order removals -- switches the selector to
// the general representation for good (`migrate`), with identical results.
class Prioritized : public Selector {
 public:
  Prioritized(double exponent, double initial, bool zero_on_sample,
              double maxfrac, int branching, uint64_t seed)
      : exponent_(exponent), initial_(initial), zero_(zero_on_sample),
        maxfrac_(maxfrac), tree_(branching, seed) {
    if (maxfrac < 0 || maxfrac > 1) throw std::invalid_argument("maxfrac");
  }
  ~Prioritized() override {
    for (Stream* st : streams_) delete st;
  }

  int64_t sample() override {
    const int64_t key = tree_.sample();
    if (!zero_) return key;
    // selectors.py:163-168: the drawn item's steps go to priority 0.
    if (general_) {
      begin_touch();
      for (Step* step : items_.at(key).steps) {
        set_priority(*step, 0.0);
        touch_users(*step);
      }
      refresh_touched();
      return key;
    }
    const auto* own = owner_.find(key);
    if (!own) throw std::out_of_range("Prioritized: unknown key");
    Stream* st = own->first;
    const int64_t start = own->second;
    // The n steps of the drawn item are contiguous in the stream's arrays, and
    // the items that contain one of them are ONE range of start positions (the
    // union of touch_range over the n steps; the drawn item is in it).
    const double zero_powered = powered(0.0);
    double* prio = &st->prio[start - st->step0];
    double* pw = &st->powered[start - st->step0];
    for (int i = 0; i < st->n; ++i) prio[i] = 0.0, pw[i] = zero_powered;
    const int64_t last = st->item0 + static_cast<int64_t>(st->items.size()) - 1;
    refresh_drawn(st, std::max(st->item0, start - st->n + 1), std::min(last, start + st->n - 1), start);
    return key;
  }
  int64_t size() const override {
    return static_cast<int64_t>(general_ ? items_.size() : owner_.size());
  }

  void insert(int64_t key, const StepId* steps, int n) override {
    if (n <= 0) throw std::invalid_argument("Prioritized: item without steps");
    if (!general_ && !stream_insert(key, steps, n)) migrate();
    if (general_) general_insert(key, steps, n);
  }

  // The next window of a stream (what a Replay inserts step after step): one
  // new step record, no list of L step ids to build, hash and compare.
  bool insert_successor(int64_t key, int64_t prev_key, const StepId& newest) override {
    if (general_) return false;
    const auto* prev = owner_.find(prev_key);
    if (!prev || owner_.count(key)) return false;
    Stream* st = prev->first;
    const int64_t start = prev->second + 1;
    // prev_key must be the stream's newest item, ending at its newest step
    if (start != st->item0 + static_cast<int64_t>(st->items.size())) return false;
    const int64_t last = st->step0 + st->n_steps() - 1;
    // (`newest` is a new id: the caller's guarantee -- a Replay issues every step
    // id once.  The general insert() checks it.)
    if (start + st->n - 2 != last) return false;
    const double born = birth_priority(newest);
    const double pw = powered(born);
    st->push_step(newest, born, pw);
    where_log(newest, st, last + 1, true);
    st->items.push_back(key);
    owner_.put(key, std::make_pair(st, start));
    const double mass = newest_mass(*st, start, pw);
    st->leaves.push_back(tree_.insert(key, mass));
    st->mass.push_back(mass);
    return true;
  }

  void remove(int64_t key) override {
    if (!general_) {
      const auto* own = owner_.find(key);
      if (!own) throw std::out_of_range("Prioritized: unknown key");
      Stream* st = own->first;
      if (st->items[0] == key) {
        tree_.remove(key);
        owner_.erase(key);
        st->items.pop_front();
        st->leaves.pop_front();
        st->mass.pop_front();
        st->item0 += 1;
        // Steps in front of the oldest remaining item belong to no item any
        // more: their priority goes with them (selectors.py:180-185).
        const int64_t keep_from = st->items.size() ? st->item0 : st->step0 + st->n_steps();
        while (st->step0 < keep_from) {
          where_log(st->ids[0], nullptr, 0, false);
          st->pop_step();
          st->step0 += 1;
        }
        if (!st->items.size()) {
          streams_.erase(st);
          delete st;
        }
        return;
      }
      migrate();                      // not the oldest item of its stream
    }
    general_remove(key);
  }

  bool can_prioritize() const override { return true; }

  // selectors.py:143-158: set the priorities, then re-aggregate every item that
  // contains one of the steps -- once per item, whatever the order: a tree
  // node's mass is always the fresh sum of its children.
  void prioritize(const StepId* steps, const double* prios, int64_t n) override {
    if (general_) {
      begin_touch();
      for (int64_t i = 0; i < n; ++i) {
        auto it = steps_.find(steps[i]);
        if (it == steps_.end()) {          // in no item (yet, or any more): kept for an item that arrives later
          early_[steps[i]] = prios[i];
          continue;
        }
        set_priority(it->second, prios[i]);
        touch_users(it->second);
      }
      refresh_touched();
      return;
    }
    ranges_.clear();
    Stream* st = nullptr;
    int64_t pos = 0;
    for (int64_t i = 0; i < n; ++i) {
      // Rows of consecutive steps: try the slot after the previous one before
      // hashing the 20-byte id.
      if (st && pos + 1 < st->step0 + st->n_steps() && st->ids[pos + 1 - st->step0] == steps[i]) {
        pos += 1;
      } else {
        auto it = where().find(steps[i]);
        if (it == where_.end()) {
          st = nullptr;
          early_[steps[i]] = prios[i];       // in no item (yet, or any more): kept for an item that arrives later
          continue;
        }
        st = it->second.first;
        pos = it->second.second;
      }
      set_slot(*st, pos, prios[i]);
      touch_range(st, pos);
    }
    refresh_ranges();
  }

 private:
  // ---------------------------------------------------------- shared pieces --
  // prio ** exponent.  0, 1 and +inf are fixed points of x ** e for e > 0 (what
  // pow returns for them, exactly): the shipped PPO config (initial inf,
  // zero_on_sample) only ever holds those, and pow is ~40 ns a call.
  double powered(double prio) const {
    if (exponent_ == 1.0) return prio;
    if (exponent_ > 0 && (prio == 0.0 || prio == 1.0 || prio == INFINITY)) return prio;
    return std::pow(prio, exponent_);
  }
  // maxfrac * max + (1 - maxfrac) * mean over prio ** exponent, summed left to
  // right like the reference's sum() (selectors.py:187-197).
  double finish(double total, double top, int64_t count) const {
    const double mean = total / static_cast<double>(count);
    if (maxfrac_ != 0.0) return maxfrac_ * top + (1 - maxfrac_) * mean;
    return mean;
  }

  // A vector that is appended to at the back and consumed from the front.
  template <typename T>
  struct Sliding {
    std::vector<T> v;
    size_t off = 0;
    size_t size() const { return v.size() - off; }
    T& operator[](int64_t i) { return v[off + static_cast<size_t>(i)]; }
    const T& operator[](int64_t i) const { return v[off + static_cast<size_t>(i)]; }
    void push_back(const T& x) { v.push_back(x); }
    void pop_front() {
      ++off;
      if (off >= 256 && off * 2 >= v.size()) {
        v.erase(v.begin(), v.begin() + static_cast<std::ptrdiff_t>(off));
        off = 0;
      }
    }
  };

  // ------------------------------------------------------------ stream mode --
  struct Stream {
    int n = 0;              // steps per item
    int64_t step0 = 0;      // position of the first stored step
    int64_t item0 = 0;      // start position of items[0]; items[j] starts at item0 + j
    // Per step, one array each (the aggregation streams through `powered`).
    Sliding<double> prio, powered;
    Sliding<StepId> ids;
    // No stored step has ever held a NaN or negative powered priority: a window
    // with a +inf step then sums to +inf with maximum +inf whatever else it holds.
    bool plain = true;
    int64_t n_steps() const { return static_cast<int64_t>(ids.size()); }
    void push_step(const StepId& id, double p, double pw) {
      ids.push_back(id);
      prio.push_back(p);
      powered.push_back(pw);
      if (!(pw >= 0.0)) plain = false;
    }
    void pop_step() {
      ids.pop_front();
      prio.pop_front();
      powered.pop_front();
    }
    Sliding<int64_t> items; // keys
    Sliding<SampleTree::Node*> leaves;   // their tree leaves
    Sliding<double> mass;                // and the mass each leaf holds (a copy next to its
                                         // neighbours': comparing against it touches no tree node)
  };
  static constexpr int64_t kRegion = 1024;      // steps a refresh may scan for the +inf shortcut
  struct Range {
    Stream* st;
    int64_t lo, hi;         // item start positions, inclusive
  };

  void set_slot(Stream& st, int64_t pos, double prio) const {
    const double pw = powered(prio);
    st.prio[pos - st.step0] = prio;
    st.powered[pos - st.step0] = pw;
    if (!(pw >= 0.0)) st.plain = false;
  }
  // Mass of the window that ENDS with the step just pushed (`initial` powered =
  // pw): with `initial: inf` (ppo/configs.yaml:42) it is decided by that step --
  // what the left-to-right sum over the n steps gives, without walking them.
  double newest_mass(const Stream& st, int64_t start, double pw) const {
    if (pw == INFINITY && st.plain) return finish(INFINITY, INFINITY, st.n);
    return stream_mass(st, start);
  }
  double stream_mass(const Stream& st, int64_t start) const {
    double total = 0.0, top = -INFINITY;
    const double* p = &st.powered[start - st.step0];
    for (int i = 0; i < st.n; ++i) {
      const double v = p[i];
      total += v;
      top = (v > top) ? v : top;
    }
    return finish(total, top, st.n);
  }
  // Masses of the `count` items starting at start, start+1, ...  One pass over
  // the region's steps (running counts) sorts the windows: a window that holds a
  // +inf step and nothing NaN or negative sums to +inf with maximum +inf whatever
  // else it holds; a window of zeros only sums to +0 with maximum 0 (-0 steps
  // included: 0 + -0 = +0, and finish() adds a +0 mean to the maximum).  With
  // `initial: inf` and zero_on_sample that is nearly every window; only the
  // others are summed -- eight at a time, each lane its own left-to-right sum
  // (independent add chains instead of one 65-long dependent chain per item),
  // when no window of the range is decided by the counts.
  void stream_masses(const Stream& st, int64_t start, int64_t count, std::vector<double>* out) const {
    constexpr int kLanes = 8;
    const double* base = &st.powered[start - st.step0];
    out->resize(static_cast<size_t>(count));
    double* o = out->data();
    int64_t done = 0;
    const int64_t span = count + st.n - 1;
    if (span <= kRegion) {
      int32_t hot[kRegion + 1], odd[kRegion + 1], some[kRegion + 1];
      hot[0] = odd[0] = some[0] = 0;
      for (int64_t i = 0; i < span; ++i) {
        const double v = base[i];
        hot[i + 1] = hot[i] + (v == INFINITY ? 1 : 0);
        odd[i + 1] = odd[i] + ((v >= 0.0) ? 0 : 1);           // NaN or negative
        some[i + 1] = some[i] + ((v == 0.0) ? 0 : 1);
      }
      int64_t decided = 0;
      if (odd[span] == 0 && (hot[span] > 0 || some[span] < span))
        for (int64_t j = 0; j < count; ++j)
          decided += (hot[j + st.n] > hot[j] || some[j + st.n] == some[j]) ? 1 : 0;
      if (decided > 0) {
        const double all_inf = finish(INFINITY, INFINITY, st.n);
        const double all_zero = finish(0.0, 0.0, st.n);
        for (int64_t j = 0; j < count; ++j) {
          if (hot[j + st.n] > hot[j]) o[j] = all_inf;
          else if (some[j + st.n] == some[j]) o[j] = all_zero;
          else o[j] = stream_mass(st, start + j);
        }
        return;
      }
    }
    for (; done + kLanes <= count; done += kLanes) {
      double total[kLanes], top[kLanes];
      for (int j = 0; j < kLanes; ++j) total[j] = 0.0, top[j] = -INFINITY;
      const double* p = base + done;
      for (int i = 0; i < st.n; ++i) {
        for (int j = 0; j < kLanes; ++j) {
          const double v = p[i + j];
          total[j] += v;
          top[j] = (v > top[j]) ? v : top[j];
        }
      }
      for (int j = 0; j < kLanes; ++j) o[done + j] = finish(total[j], top[j], st.n);
    }
    for (; done < count; ++done) o[done] = stream_mass(st, start + done);
  }
  // Items of `st` that contain the step at `pos`.
  void touch_range(Stream* st, int64_t pos) {
    const int64_t last = st->item0 + static_cast<int64_t>(st->items.size()) - 1;
    const int64_t lo = std::max(st->item0, pos - st->n + 1), hi = std::min(last, pos);
    if (lo > hi) return;
    if (!ranges_.empty() && ranges_.back().st == st && lo <= ranges_.back().hi + 1 && hi >= ranges_.back().lo - 1) {
      ranges_.back().lo = std::min(ranges_.back().lo, lo);
      ranges_.back().hi = std::max(ranges_.back().hi, hi);
    } else {
      ranges_.push_back({st, lo, hi});
    }
  }
  // refresh_ranges for the ONE range a zero-on-sample draw touches, as a single
  // sliding pass: the counts of +inf and of non-zero steps of window j + 1 follow
  // from window j's by one step out and one step in.  In a stream that has never
  // held a NaN or negative powered priority (`plain`) a window with a +inf step
  // sums to +inf with maximum +inf and a window of zeros to +0 with maximum 0 --
  // the values stream_masses decides by its prefix counts; every other window is
  // summed left to right as there.  Same masses, same leaf updates.
  //
  // Which of the <= 2n - 1 windows can have changed at all (round 6): only the n
  // steps of the drawn item at `start` were written, all to zero.  A window that
  // holds a +inf step OUTSIDE them held it before and holds it now: its mass was
  // and is `all_inf`, bit for bit, and needs neither a look nor a leaf update.
  // With p_left the nearest +inf step left of the drawn item and p_right the
  // nearest one right of it, those are the windows that begin at or before p_left
  // or at or after p_right - n + 1; the pass runs over what lies between -- with
  // `initial: inf` (ppo/configs.yaml:42) and unsampled neighbours that is the
  // drawn window alone, found by reading two steps instead of sliding over 2n.
  void refresh_drawn(Stream* st, int64_t lo, int64_t hi, int64_t start) {
    if (!st->plain || lo > hi) {
      ranges_.clear();
      if (lo <= hi) ranges_.push_back({st, lo, hi});
      refresh_ranges();
      return;
    }
    const int n = st->n;
    {
      const double* pw = &st->powered[0];        // index: position - step0
      const int64_t s0 = st->step0;
      for (int64_t p = start - 1; p >= lo; --p)
        if (pw[p - s0] == INFINITY) {
          lo = p + 1;
          break;
        }
      for (int64_t p = start + n; p <= hi + n - 1; ++p)
        if (pw[p - s0] == INFINITY) {
          hi = p - n;
          break;
        }
      if (lo > hi) return;
    }
    const double* base = &st->powered[lo - st->step0];       // steps lo .. hi + n - 1
    double* held = &st->mass[lo - st->item0];
    SampleTree::Node* const* leaf = &st->leaves[lo - st->item0];
    const double all_inf = finish(INFINITY, INFINITY, n), all_zero = finish(0.0, 0.0, n);
    int hot = 0, some = 0;
    for (int i = 0; i < n; ++i) {
      hot += base[i] == INFINITY ? 1 : 0;
      some += base[i] != 0.0 ? 1 : 0;
    }
    leaves_.clear();
    masses_.clear();
    const int64_t count = hi - lo + 1;
    for (int64_t j = 0; j < count; ++j) {
      const double now = hot > 0 ? all_inf : some == 0 ? all_zero : stream_mass(*st, lo + j);
      uint64_t a, b;
      std::memcpy(&a, &now, 8);
      std::memcpy(&b, &held[j], 8);
      if (a != b) {
        held[j] = now;
        leaves_.push_back(leaf[j]);
        masses_.push_back(now);
      }
      if (j + 1 < count) {
        const double out = base[j], in = base[j + n];
        hot += (in == INFINITY ? 1 : 0) - (out == INFINITY ? 1 : 0);
        some += (in != 0.0 ? 1 : 0) - (out != 0.0 ? 1 : 0);
      }
    }
    if (!leaves_.empty()) tree_.update_leaves(leaves_.data(), masses_.data(), static_cast<int64_t>(leaves_.size()));
  }
  void refresh_ranges() {
    std::sort(ranges_.begin(), ranges_.end(), [](const Range& a, const Range& b) {
      return a.st != b.st ? a.st < b.st : a.lo < b.lo;
    });
    leaves_.clear();
    masses_.clear();
    Stream* st = nullptr;
    int64_t done = 0;        // next start position not yet emitted for `st`
    for (const Range& r : ranges_) {
      if (r.st != st) {
        st = r.st;
        done = r.lo;
      }
      const int64_t from = std::max(done, r.lo);
      if (from <= r.hi) {
        stream_masses(*st, from, r.hi - from + 1, &fresh_);
        // A leaf whose freshly aggregated mass is bit for bit what it holds
        // already needs no update: every ancestor is the left-to-right sum of its
        // children's current masses, so re-summing it would reproduce the value
        // it has (selectors.py:150-158 re-aggregates and updates every touched
        // item; the result is the same tree).  With priorities in {0, inf} --
        // zero_on_sample over `initial: inf` -- a drawn window changes its own
        // mass and almost never its neighbours': one leaf update per draw
        // instead of 2L - 1.
        for (int64_t start = from; start <= r.hi; ++start) {
          double& held = st->mass[start - st->item0];
          const double now = fresh_[static_cast<size_t>(start - from)];
          uint64_t a, b;
          std::memcpy(&a, &now, 8);
          std::memcpy(&b, &held, 8);
          if (a == b) continue;
          held = now;
          leaves_.push_back(st->leaves[start - st->item0]);
          masses_.push_back(now);
        }
      }
      done = std::max(done, r.hi + 1);
    }
    tree_.update_leaves(leaves_.data(), masses_.data(), static_cast<int64_t>(leaves_.size()));
  }

  // True if the item was taken in stream mode; false if it does not fit (the
  // caller migrates).  Nothing is modified when it returns false.
  bool stream_insert(int64_t key, const StepId* ids, int n) {
    if (n < 2 || owner_.count(key)) return false;
    auto newest = where().find(ids[n - 1]);
    if (newest != where_.end()) return false;           // the newest step must be new
    auto prev = where_.find(ids[n - 2]);
    Stream* st = nullptr;
    if (prev != where_.end()) {
      // Next window of an existing stream: ids[n-2] is that stream's newest
      // step, and ids[0 .. n-3] are the steps before it.
      st = prev->second.first;
      const int64_t pos = prev->second.second;
      const int64_t last = st->step0 + st->n_steps() - 1;
      if (st->n != n || pos != last) return false;
      const int64_t start = pos - (n - 2);
      if (start != st->item0 + static_cast<int64_t>(st->items.size())) return false;
      for (int i = 0; i < n - 2; ++i)
        if (!(st->ids[start + i - st->step0] == ids[i])) return false;
      const double born = birth_priority(ids[n - 1]);
      st->push_step(ids[n - 1], born, powered(born));
      where_.emplace(ids[n - 1], std::make_pair(st, pos + 1));
      st->items.push_back(key);
      owner_.put(key, std::make_pair(st, start));
      const double mass = stream_mass(*st, start);
      st->leaves.push_back(tree_.insert(key, mass));
      st->mass.push_back(mass);
      return true;
    }
    // First window of a new stream: none of its steps may be known, and they
    // must be distinct.
    for (int i = 0; i < n - 1; ++i) {
      if (where_.count(ids[i])) return false;
      for (int j = i + 1; j < n; ++j)
        if (ids[i] == ids[j]) return false;
    }
    st = new Stream();
    streams_.insert(st);
    st->n = n;
    for (int i = 0; i < n; ++i) {
      const double born = birth_priority(ids[i]);
      st->push_step(ids[i], born, powered(born));
      where_.emplace(ids[i], std::make_pair(st, static_cast<int64_t>(i)));
    }
    st->items.push_back(key);
    owner_.put(key, std::make_pair(st, int64_t{0}));
    const double mass = stream_mass(*st, 0);
    st->leaves.push_back(tree_.insert(key, mass));
    st->mass.push_back(mass);
    return true;
  }

  // Rebuild the general representation from the streams (the tree is shared and
  // stays as it is).  Items are re-created in key order = insertion order, so
  // every step's user list has the order the reference's would have.
  void migrate() {
    std::vector<int64_t> keys;
    keys.reserve(owner_.size());
    owner_.for_each([&](int64_t key, const std::pair<Stream*, int64_t>&) { keys.push_back(key); });
    std::sort(keys.begin(), keys.end());
    std::vector<StepId> ids;
    for (int64_t key : keys) {
      const auto [st, start] = *owner_.find(key);
      ids.clear();
      for (int64_t pos = start; pos < start + st->n; ++pos) ids.push_back(st->ids[pos - st->step0]);
      general_link(key, ids.data(), st->n);
      for (int64_t pos = start; pos < start + st->n; ++pos) {
        Step& step = steps_.at(st->ids[pos - st->step0]);
        step.prio = st->prio[pos - st->step0];
        step.powered = st->powered[pos - st->step0];
      }
    }
    for (Stream* st : streams_) delete st;
    streams_.clear();
    where_.clear();
    where_pending_.clear();
    where_stale_ = false;
    owner_.clear();
    general_ = true;
  }

  // ----------------------------------------------------------- general mode --
  struct Item;
  // One time step: its priority, priority ** exponent (computed when the
  // priority is set, not every time an item is aggregated) and the items that
  // contain it (live ones are users[head:]).
  struct Step {
    double prio = 0.0, powered = 0.0;
    std::vector<Item*> users;
    size_t head = 0;
    const StepId* id = nullptr;
  };
  struct Item {
    int64_t key = 0;
    uint64_t stamp = 0;
    std::vector<Step*> steps;
  };

  void set_priority(Step& step, double prio) const {
    step.prio = prio;
    step.powered = powered(prio);
  }
  double mass_of(const Item& item) const {
    double total = 0.0, top = -INFINITY;
    for (const Step* step : item.steps) {
      const double v = step->powered;
      total += v;
      top = (v > top) ? v : top;
    }
    return finish(total, top, static_cast<int64_t>(item.steps.size()));
  }
  // Item <-> step links; new steps start at the initial priority.
  Item& general_link(int64_t key, const StepId* steps, int n) {
    Item& mine = items_[key];
    mine.key = key;
    mine.steps.clear();
    mine.steps.reserve(n);
    for (int i = 0; i < n; ++i) {
      auto found = steps_.find(steps[i]);
      if (found == steps_.end()) {
        found = steps_.emplace(steps[i], Step()).first;
        found->second.id = &found->first;
        set_priority(found->second, birth_priority(steps[i]));
      }
      mine.steps.push_back(&found->second);       // node addresses are stable
    }
    for (Step* step : mine.steps) step->users.push_back(&mine);
    return mine;
  }
  void general_insert(int64_t key, const StepId* steps, int n) {
    tree_.insert(key, mass_of(general_link(key, steps, n)));
  }
  void general_remove(int64_t key) {
    tree_.remove(key);
    auto it = items_.find(key);
    Item* gone = &it->second;
    for (Step* step : gone->steps) {
      auto& users = step->users;
      // Oldest item first is the usual eviction order: it sits at `head`.
      // (An item may list the same step twice: then the step may be gone already.)
      if (step->head < users.size() && users[step->head] == gone) {
        ++step->head;
      } else {
        users.erase(std::find(users.begin() + static_cast<std::ptrdiff_t>(step->head), users.end(), gone));
      }
      if (step->head == users.size()) steps_.erase(*step->id);   // its priority goes with it (selectors.py:180-185)
    }
    items_.erase(it);
  }
  void begin_touch() {
    touched_.clear();
    ++epoch_;
  }
  void touch_users(Step& step) {
    for (size_t i = step.head; i < step.users.size(); ++i) {
      Item* item = step.users[i];
      if (item->stamp == epoch_) continue;
      item->stamp = epoch_;
      touched_.push_back(item);
    }
  }
  void refresh_touched() {
    keys_.resize(touched_.size());
    masses_.resize(touched_.size());
    for (size_t i = 0; i < touched_.size(); ++i) {
      keys_[i] = touched_[i]->key;
      masses_[i] = mass_of(*touched_[i]);
    }
    tree_.update_many(keys_.data(), masses_.data(), static_cast<int64_t>(touched_.size()));
  }

  double exponent_, initial_;
  // Priorities given to step ids that belong to NO item at that moment.  The
  // reference's table takes a priority for any step id (selectors.py:143-150,
  // `prios` is a defaultdict) and an item that arrives later aggregates it
  // (:187-197) -- unreachable through Replay.update (sampled steps are in
  // items), reachable through the selector protocol.  A new step record takes
  // its entry from here; like the reference's, entries of ids that never arrive
  // stay.
  std::unordered_map<StepId, double, StepIdHash> early_;
  double birth_priority(const StepId& id) {
    if (early_.empty()) return initial_;
    auto it = early_.find(id);
    if (it == early_.end()) return initial_;
    const double prio = it->second;
    early_.erase(it);
    return prio;
  }
  bool zero_;
  double maxfrac_;
  SampleTree tree_;
  bool general_ = false;
  // stream mode
  std::unordered_set<Stream*> streams_;
  // id -> stream, position.  Only `prioritize` by step id and the general insert
  // read it, so the per-step add / erase of the sliding-window fast path (two
  // random accesses into a table of 100 k ids: cache misses, a third of an
  // insert + evict) are logged and applied in order when somebody asks: the PPO
  // config (zero_on_sample addresses the drawn item's steps by position, the
  // agent sets no priorities) never does.
  struct WhereOp { StepId id; Stream* st; int64_t pos; bool add; };
  std::unordered_map<StepId, std::pair<Stream*, int64_t>, StepIdHash> where_;
  std::vector<WhereOp> where_pending_;
  bool where_stale_ = false;       // the backlog was dropped: rebuild from the streams when asked
  void where_log(const StepId& id, Stream* st, int64_t pos, bool add) {
    if (where_stale_) return;
    where_pending_.push_back({id, st, pos, add});
    static const size_t limit = [] {        // EMB_WHERE_BACKLOG: entries kept before giving up (tests)
      const char* e = emb::knob("EMB_WHERE_BACKLOG");
      const long v = e ? std::atol(e) : 0;
      return v > 0 ? static_cast<size_t>(v) : (size_t{1} << 16);
    }();
    if (where_pending_.size() >= limit) {
      // Nobody has asked for `limit` (65 536) steps: stop keeping a backlog at all.  The
      // table is rebuilt from the streams' own id arrays if it is ever needed.
      where_pending_.clear();
      where_pending_.shrink_to_fit();
      where_.clear();
      where_stale_ = true;
    }
  }
  std::unordered_map<StepId, std::pair<Stream*, int64_t>, StepIdHash>& where() {
    if (where_stale_) {
      where_.clear();
      for (Stream* st : streams_)
        for (int64_t i = 0; i < st->n_steps(); ++i)
          where_.emplace(st->ids[i], std::make_pair(st, st->step0 + i));
      where_stale_ = false;
    }
    for (const WhereOp& op : where_pending_) {
      if (op.add) where_.emplace(op.id, std::make_pair(op.st, op.pos));
      else where_.erase(op.id);
    }
    where_pending_.clear();
    return where_;
  }
  SlidingMap<std::pair<Stream*, int64_t>> owner_;                             // key -> stream, start
  std::vector<Range> ranges_;
  std::vector<SampleTree::Node*> leaves_;
  // general mode
  std::unordered_map<StepId, Step, StepIdHash> steps_;
  std::unordered_map<int64_t, Item> items_;
  std::vector<Item*> touched_;
  uint64_t epoch_ = 0;
  // both
  std::vector<int64_t> keys_;
  std::vector<double> masses_, fresh_;
};

class Mixture : public Selector {
 public:
  // Members in name-sorted order with non-zero fractions (caller's job, as in
  // selectors.py:205-211); fractions are float32 in the reference.
  Mixture(std::vector<std::shared_ptr<Selector>> members,
          const std::vector<float>& fractions, uint64_t seed)
      : members_(std::move(members)), rng_(seed) {
    if (members_.empty() || members_.size() != fractions.size())
      throw std::invalid_argument("Mixture: members/fractions mismatch");
    for (float f : fractions) frac_.push_back(static_cast<double>(f));
    cdf_.resize(frac_.size());
  }
  int64_t sample() override {
    const int pick = rng_.choice(frac_.data(), static_cast<int>(frac_.size()), cdf_.data());
    return members_[pick]->sample();
  }
  // The reference's Mixture has no __len__ (so Replay.sample raises with it,
  // replay.py:123); every member holds the same keys, report the first.
  int64_t size() const override { return members_[0]->size(); }
  void insert(int64_t key, const StepId* steps, int n) override {
    for (auto& m : members_) m->insert(key, steps, n);
  }
  // The next window of a stream: handed on when exactly one member wants step
  // ids (the usual uniform / priority / recency mix) -- that member is asked
  // first, and only if it takes the step do the others get their plain insert;
  // otherwise nothing has changed and the caller inserts the full step list.
  // (Members are independent: the order of their inserts changes no draw.)
  bool insert_successor(int64_t key, int64_t prev_key, const StepId& newest) override {
    Selector* picky = nullptr;
    int count = 0;
    for (auto& m : members_)
      if (m->needs_stepids()) {
        picky = m.get();
        ++count;
      }
    if (count != 1 || !picky->insert_successor(key, prev_key, newest)) return false;
    for (auto& m : members_)
      if (m.get() != picky) m->insert(key, nullptr, 0);
    return true;
  }
  void remove(int64_t key) override {
    for (auto& m : members_) m->remove(key);
  }
  bool can_prioritize() const override {
    for (auto& m : members_) if (m->can_prioritize()) return true;
    return false;
  }
  bool needs_stepids() const override {
    for (auto& m : members_) if (m->needs_stepids()) return true;
    return false;
  }
  void prioritize(const StepId* s, const double* p, int64_t n) override {
    for (auto& m : members_) if (m->can_prioritize()) m->prioritize(s, p, n);
  }

 private:
  std::vector<std::shared_ptr<Selector>> members_;
  std::vector<double> frac_, cdf_;
  NpRandom rng_;
};

// Age-biased sampling (selectors.py:60-125): the item inserted `age` inserts ago
// is drawn with probability proportional to uprobs[age].  The b-ary table of
// normalised block masses is built by the caller (numpy arithmetic, the
// reference's `_build`) and handed over level by level: level l has b^l rows of
// b probabilities.  A draw is one `choice` per level; while fewer items than
// table entries are held the age is scaled by items / entries; an age whose
// item is gone is drawn again (the reference sleeps 10 ms and retries).  The
// reference's `_sample` cannot run as written (an unbound `segment`); with that
// one token repaired its draws are these (golden sel_recency).
class Recency : public Selector {
 public:
  Recency(std::vector<double> table, int depth, int bfactor, int64_t entries, uint64_t seed)
      : table_(std::move(table)), depth_(depth), b_(bfactor), entries_(entries), rng_(seed) {
    if (depth_ < 1 || b_ < 2 || entries_ < 1) throw std::invalid_argument("Recency: bad table shape");
    int64_t rows = 1, need = 0;
    for (int l = 0; l < depth_; ++l) {
      offset_.push_back(need);
      need += rows * b_;
      rows *= b_;
    }
    if (static_cast<int64_t>(table_.size()) != need || entries_ > rows)
      throw std::invalid_argument("Recency: table size does not match depth / branching");
    cdf_.resize(b_);
  }
  int64_t sample() override {
    for (int attempt = 0; attempt < 1000; ++attempt) {
      int64_t age = 0;
      for (int l = 0; l < depth_; ++l)
        age = age * b_ + rng_.choice(table_.data() + offset_[l] + age * b_, b_, cdf_.data());
      const int64_t held = static_cast<int64_t>(key_of_.size());
      if (held < entries_)
        age = static_cast<int64_t>(static_cast<double>(age) / static_cast<double>(entries_) *
                                   static_cast<double>(held));
      const int64_t* key = key_of_.find(step_ - 1 - age);
      if (key) return *key - 1;
    }
    throw std::out_of_range("Recency: no live item found");
  }
  int64_t size() const override { return static_cast<int64_t>(key_of_.size()); }
  bool needs_stepids() const override { return false; }
  // (Both maps hold value + 1: SlidingMap reads a zero as "absent".  A Replay's
  // item ids and the insert counter both arrive in order and leave oldest first,
  // so neither map hashes.)
  void insert(int64_t key, const StepId*, int) override {
    step_of_.put(key, step_ + 1);
    key_of_.put(step_, key + 1);
    ++step_;
  }
  void remove(int64_t key) override {
    const int64_t* step = step_of_.find(key);
    if (!step) throw std::runtime_error("Recency: unknown key");
    key_of_.erase(*step - 1);
    step_of_.erase(key);
  }

 private:
  std::vector<double> table_;
  std::vector<int64_t> offset_;
  int depth_, b_;
  int64_t entries_;
  int64_t step_ = 0;
  SlidingMap<int64_t> step_of_, key_of_;
  std::vector<double> cdf_;
  NpRandom rng_;
};

// A selector implemented by the caller (the duck-typed Python protocol of
// selectors.py: __call__/__len__/__setitem__/__delitem__/prioritize) reached
// through C function pointers.
struct SelectorCallbacks {
  void* user;
  int64_t (*sample)(void* user);
  int64_t (*size)(void* user);
  void (*insert)(void* user, int64_t key, const uint8_t* stepids, int32_t n);
  void (*remove)(void* user, int64_t key);
  void (*prioritize)(void* user, const uint8_t* stepids, const double* prios, int64_t n);
};

class CallbackSelector : public Selector {
 public:
  explicit CallbackSelector(const SelectorCallbacks& cb) : cb_(cb) {}
  int64_t sample() override { return cb_.sample(cb_.user); }
  int64_t size() const override { return cb_.size(cb_.user); }
  void insert(int64_t key, const StepId* steps, int n) override {
    cb_.insert(cb_.user, key, reinterpret_cast<const uint8_t*>(steps), n);
  }
  void remove(int64_t key) override { cb_.remove(cb_.user, key); }
  bool can_prioritize() const override { return cb_.prioritize != nullptr; }
  void prioritize(const StepId* s, const double* p, int64_t n) override {
    if (cb_.prioritize) cb_.prioritize(cb_.user, reinterpret_cast<const uint8_t*>(s), p, n);
  }

 private:
  SelectorCallbacks cb_;
};

}  // namespace emb
