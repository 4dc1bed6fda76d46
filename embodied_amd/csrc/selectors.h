// Host index samplers, integer/float64 exact restatements of the reference's
// embodied/core/selectors.py (Fifo :7-26, Uniform :29-57, Prioritized :128-197,
// Mixture :200-228, SampleTree :231-354).  Pure host C++: this is bookkeeping,
// O(batch) per sample, and must reproduce the reference's numpy streams bit for
// bit; payload movement is what runs on the GPU.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "np_random.h"

namespace emb {

constexpr int kStepIdBytes = 20;

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

class Selector {
 public:
  virtual ~Selector() = default;
  virtual int64_t sample() = 0;
  virtual int64_t size() const = 0;
  virtual void insert(int64_t key, const StepId* steps, int n) = 0;
  virtual void remove(int64_t key) = 0;
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
  std::deque<int64_t> window_;
  std::unordered_map<int64_t, int64_t> sparse_;
};

// b-ary sum tree (selectors.py:231-354).  Children keep list order because the
// draw at each level indexes them and float sums run left to right.
class SampleTree {
 public:
  struct Node {
    Node* up = nullptr;
    bool leaf = false;
    int64_t key = 0;
    double mass = 0.0;
    std::vector<Node*> kids;
  };

  SampleTree(int branching, uint64_t seed) : branching_(branching), rng_(seed) {
    if (branching < 2) throw std::invalid_argument("SampleTree: branching < 2");
    root_ = new Node();
  }
  ~SampleTree() { destroy(root_); }
  SampleTree(const SampleTree&) = delete;
  SampleTree& operator=(const SampleTree&) = delete;

  int64_t size() const { return static_cast<int64_t>(leaves_.size()); }
  double root_mass() const { return root_->mass; }
  const Node* root() const { return root_; }

  void insert(int64_t key, double mass) {
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
        spot = new Node();
        attach(spot, root_);
        root_ = spot;
      }
      for (int i = 0; i < climbed; ++i) {
        Node* fresh = new Node();
        attach(spot, fresh);
        spot = fresh;
      }
    }
    Node* leaf = new Node();
    leaf->leaf = true;
    leaf->key = key;
    leaf->mass = mass;
    attach(spot, leaf);
    leaves_[key] = leaf;
    tail_ = leaf;
  }

  void remove(int64_t key) {
    auto it = leaves_.find(key);
    if (it == leaves_.end()) throw std::out_of_range("SampleTree: unknown key");
    Node* leaf = it->second;
    leaves_.erase(it);
    Node* hole_parent = leaf->up;
    Node* tail_parent = tail_->up;
    detach(hole_parent, leaf);
    if (leaf != tail_) attach(hole_parent, tail_);
    Node* node = tail_parent;
    while (node->up && node->kids.empty()) {
      Node* above = node->up;
      detach(above, node);
      delete node;
      node = above;
    }
    delete leaf;
    if (node->kids.empty()) {
      tail_ = nullptr;
      return;
    }
    while (!node->leaf) node = node->kids.back();
    tail_ = node;
  }

  void update(int64_t key, double mass) {
    auto it = leaves_.find(key);
    if (it == leaves_.end()) throw std::out_of_range("SampleTree: unknown key");
    it->second->mass = mass;
    resum(it->second->up);
  }

  int64_t sample() {
    if (leaves_.empty()) throw std::runtime_error("SampleTree: empty");
    Node* node = root_;
    while (!node->leaf) {
      const int k = static_cast<int>(node->kids.size());
      mass_.resize(k);
      prob_.resize(k);
      cdf_.resize(k);
      for (int i = 0; i < k; ++i) mass_[i] = node->kids[i]->mass;
      const double total = np_pairwise_sum(mass_.data(), k);
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
  static void resum(Node* node) {
    while (node) {
      double total = 0.0;
      for (Node* kid : node->kids) total += kid->mass;
      node->mass = total;
      node = node->up;
    }
  }
  static void detach(Node* parent, Node* child) {
    child->up = nullptr;
    auto& kids = parent->kids;
    kids.erase(std::find(kids.begin(), kids.end(), child));
    resum(parent);
  }
  static void attach(Node* parent, Node* child) {
    if (child->up) detach(child->up, child);
    child->up = parent;
    parent->kids.push_back(child);
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
  std::unordered_map<int64_t, Node*> leaves_;
  std::vector<double> mass_, prob_, cdf_;
};

class Prioritized : public Selector {
 public:
  Prioritized(double exponent, double initial, bool zero_on_sample,
              double maxfrac, int branching, uint64_t seed)
      : exponent_(exponent), initial_(initial), zero_(zero_on_sample),
        maxfrac_(maxfrac), tree_(branching, seed) {
    if (maxfrac < 0 || maxfrac > 1) throw std::invalid_argument("maxfrac");
  }

  int64_t sample() override {
    const int64_t key = tree_.sample();
    if (zero_) {
      const auto steps = items_.at(key);  // copy: prioritize() reads items_
      std::vector<double> zeros(steps.size(), 0.0);
      prioritize(steps.data(), zeros.data(), static_cast<int64_t>(steps.size()));
    }
    return key;
  }
  int64_t size() const override { return static_cast<int64_t>(items_.size()); }

  void insert(int64_t key, const StepId* steps, int n) override {
    if (n <= 0) throw std::invalid_argument("Prioritized: item without steps");
    auto& mine = items_[key];
    mine.assign(steps, steps + n);
    for (int i = 0; i < n; ++i) {
      users_[steps[i]].push_back(key);
      prio_.emplace(steps[i], initial_);
    }
    tree_.insert(key, mass_of(key));
  }

  void remove(int64_t key) override {
    tree_.remove(key);
    auto it = items_.find(key);
    for (const StepId& sid : it->second) {
      auto& group = users_[sid];
      group.erase(std::find(group.begin(), group.end(), key));
      if (group.empty()) {
        users_.erase(sid);
        prio_.erase(sid);
      }
    }
    items_.erase(it);
  }

  bool can_prioritize() const override { return true; }

  void prioritize(const StepId* steps, const double* prios, int64_t n) override {
    touched_.clear();
    for (int64_t i = 0; i < n; ++i) {
      auto it = users_.find(steps[i]);
      if (it == users_.end()) continue;  // step no longer in any item
      prio_[steps[i]] = prios[i];
      touched_.insert(touched_.end(), it->second.begin(), it->second.end());
    }
    std::sort(touched_.begin(), touched_.end());
    touched_.erase(std::unique(touched_.begin(), touched_.end()), touched_.end());
    for (int64_t key : touched_) tree_.update(key, mass_of(key));
  }

 private:
  // maxfrac * max + (1 - maxfrac) * mean over prio ** exponent (py:187-197).
  double mass_of(int64_t key) {
    const auto& steps = items_.at(key);
    double total = 0.0, top = -INFINITY;
    for (const StepId& sid : steps) {
      double v = prio_.at(sid);
      if (exponent_ != 1.0) v = std::pow(v, exponent_);
      total += v;
      top = (v > top) ? v : top;
    }
    const double mean = total / static_cast<double>(steps.size());
    if (maxfrac_ != 0.0) return maxfrac_ * top + (1 - maxfrac_) * mean;
    return mean;
  }

  double exponent_, initial_;
  bool zero_;
  double maxfrac_;
  SampleTree tree_;
  std::unordered_map<StepId, double, StepIdHash> prio_;
  std::unordered_map<StepId, std::vector<int64_t>, StepIdHash> users_;
  std::unordered_map<int64_t, std::vector<StepId>> items_;
  std::vector<int64_t> touched_;
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
