// Host-only entry points of include/embodied_hip.h: errors, knobs, the numpy-exact
// PRNG, the sample tree and the selectors.  No HIP call in this file.
#include "handles.h"

extern "C" {

const char* emb_last_error(void) { return g_error.c_str(); }
int32_t emb_abi_version(void) { return EMB_ABI_VERSION; }

int32_t emb_configure(const char* name, const char* value) {
  return guarded([&] {
    need(name && std::strncmp(name, "EMB_", 4) == 0, "configure: knob names start with EMB_");
    if (!emb::knob_known(name)) {
      std::string known;
      for (const char* k : emb::kKnobNames) known += std::string(known.empty() ? "" : ", ") + k;
      throw std::invalid_argument(std::string("configure: ") + name + " is not a knob of this library (" + known + ")");
    }
    if (emb::knob_set(name, value) != 0)
      throw std::invalid_argument(std::string("configure: ") + name +
                                  " is already in effect (knobs are read once: set them before the "
                                  "first call that uses them)");
  });
}

// ---------------------------------------------------------------------- rng --

int32_t emb_rng_create(const uint32_t* words, int32_t n_words, emb_rng_t** out) {
  return guarded([&] {
    need(words && n_words > 0 && out, "rng: bad arguments");
    *out = new emb_rng(std::vector<uint32_t>(words, words + n_words));
  });
}

int32_t emb_rng_integers(emb_rng_t* rng, int64_t high, int64_t count, int64_t* out) {
  return guarded([&] {
    need(rng && out && high >= 1 && count >= 0, "rng_integers: bad arguments");
    std::lock_guard<std::mutex> lock(rng->mu);
    for (int64_t i = 0; i < count; ++i) out[i] = rng->impl.integers(high);
  });
}

int32_t emb_rng_random(emb_rng_t* rng, int64_t count, double* out) {
  return guarded([&] {
    need(rng && out && count >= 0, "rng_random: bad arguments");
    std::lock_guard<std::mutex> lock(rng->mu);
    for (int64_t i = 0; i < count; ++i) out[i] = rng->impl.random();
  });
}

int32_t emb_rng_choice(emb_rng_t* rng, const double* p, int32_t k, int64_t count, int64_t* out) {
  return guarded([&] {
    need(rng && p && out && k >= 1 && count >= 0, "rng_choice: bad arguments");
    std::lock_guard<std::mutex> lock(rng->mu);
    std::vector<double> cdf(k);
    for (int64_t i = 0; i < count; ++i) out[i] = rng->impl.choice(p, k, cdf.data());
  });
}

int32_t emb_rng_destroy(emb_rng_t* rng) {
  delete rng;
  return EMB_OK;
}

int32_t emb_np_sum(const double* values, int64_t n, double* out) {
  return guarded([&] {
    need(values && out && n >= 0, "np_sum: bad arguments");
    *out = emb::np_pairwise_sum(values, n);
  });
}

// --------------------------------------------------------------------- tree --

int32_t emb_tree_create(int32_t branching, uint64_t seed, emb_tree_t** out) {
  return guarded([&] {
    need(out, "tree: out is null");
    *out = new emb_tree(branching, seed);
  });
}

#define TREE_OP(...)                                  \
  return guarded([&] {                                \
    need(tree, "tree handle is null");                \
    std::lock_guard<std::mutex> lock(tree->mu);       \
    __VA_ARGS__;                                      \
  })

int32_t emb_tree_insert(emb_tree_t* tree, int64_t key, double uprob) { TREE_OP(tree->impl.insert(key, uprob)); }
int32_t emb_tree_remove(emb_tree_t* tree, int64_t key) { TREE_OP(tree->impl.remove(key)); }
int32_t emb_tree_update(emb_tree_t* tree, int64_t key, double uprob) { TREE_OP(tree->impl.update(key, uprob)); }
int32_t emb_tree_sample(emb_tree_t* tree, int64_t* key) { TREE_OP(need(key, "tree_sample: null output"); *key = tree->impl.sample()); }
int32_t emb_tree_len(emb_tree_t* tree, int64_t* n) { TREE_OP(need(n, "tree_len: null output"); *n = tree->impl.size()); }
int32_t emb_tree_root_sum(emb_tree_t* tree, double* total) { TREE_OP(need(total, "tree_root_sum: null output"); *total = tree->impl.root_mass()); }

int32_t emb_tree_shape(emb_tree_t* tree, int64_t cap, int64_t* depths, int64_t* n_leaves,
                       int64_t* n_nodes) {
  TREE_OP({
    int64_t leaves = 0, nodes = 0;
    std::vector<std::pair<const emb::SampleTree::Node*, int64_t>> stack;
    stack.emplace_back(tree->impl.root(), 0);
    while (!stack.empty()) {
      auto [node, depth] = stack.back();
      stack.pop_back();
      ++nodes;
      if (node->leaf) {
        if (depths && leaves < cap) depths[leaves] = depth;
        ++leaves;
      }
      for (auto* kid : node->kids) stack.emplace_back(kid, depth + 1);
    }
    if (n_leaves) *n_leaves = leaves;
    if (n_nodes) *n_nodes = nodes;
  });
}

int32_t emb_tree_destroy(emb_tree_t* tree) {
  delete tree;
  return EMB_OK;
}

// ---------------------------------------------------------------- selectors --

static int32_t make_selector(emb_selector_t** out, std::shared_ptr<emb::Selector> impl) {
  auto* h = new emb_selector();
  h->impl = std::move(impl);
  *out = h;
  return EMB_OK;
}

int32_t emb_selector_create_fifo(emb_selector_t** out) {
  return guarded([&] { need(out, "out is null"); make_selector(out, std::make_shared<emb::Fifo>()); });
}

int32_t emb_selector_create_uniform(uint64_t seed, emb_selector_t** out) {
  return guarded([&] { need(out, "out is null"); make_selector(out, std::make_shared<emb::Uniform>(seed)); });
}

int32_t emb_selector_create_prioritized(double exponent, double initial, int32_t zero_on_sample,
                                        double maxfrac, int32_t branching, uint64_t seed,
                                        emb_selector_t** out) {
  return guarded([&] {
    need(out, "out is null");
    make_selector(out, std::make_shared<emb::Prioritized>(exponent, initial, zero_on_sample != 0,
                                                          maxfrac, branching, seed));
  });
}

int32_t emb_selector_create_mixture(emb_selector_t* const* members, const float* fractions,
                                    int32_t n, uint64_t seed, emb_selector_t** out) {
  return guarded([&] {
    need(members && fractions && n >= 1 && out, "mixture: bad arguments");
    std::vector<std::shared_ptr<emb::Selector>> impls;
    for (int i = 0; i < n; ++i) {
      need(members[i], "mixture: null member");
      impls.push_back(members[i]->impl);
    }
    make_selector(out, std::make_shared<emb::Mixture>(
                           std::move(impls), std::vector<float>(fractions, fractions + n), seed));
  });
}

int32_t emb_selector_create_recency(const double* table, int64_t table_len, int32_t depth,
                                    int32_t bfactor, int64_t entries, uint64_t seed,
                                    emb_selector_t** out) {
  return guarded([&] {
    need(table && table_len > 0 && out, "recency: bad arguments");
    make_selector(out, std::make_shared<emb::Recency>(
                           std::vector<double>(table, table + table_len), depth, bfactor, entries, seed));
  });
}

int32_t emb_selector_create_callback(const emb_selector_callbacks_t* cb, emb_selector_t** out) {
  return guarded([&] {
    need(cb && cb->sample && cb->size && cb->insert && cb->remove && out, "callback selector: bad arguments");
    emb::SelectorCallbacks c{cb->user, cb->sample, cb->size, cb->insert, cb->remove, cb->prioritize};
    make_selector(out, std::make_shared<emb::CallbackSelector>(c));
  });
}

#define SEL_OP(...)                                   \
  return guarded([&] {                                \
    need(sel, "selector handle is null");             \
    std::lock_guard<std::mutex> lock(*sel->mu);       \
    sel->gate->drain();                               \
    __VA_ARGS__;                                      \
  })

int32_t emb_selector_insert(emb_selector_t* sel, int64_t key, const uint8_t* stepids, int32_t n_steps) {
  SEL_OP(sel->impl->insert(key, reinterpret_cast<const emb::StepId*>(stepids), stepids ? n_steps : 0));
}
int32_t emb_selector_remove(emb_selector_t* sel, int64_t key) { SEL_OP(sel->impl->remove(key)); }
int32_t emb_selector_sample(emb_selector_t* sel, int64_t* key) { SEL_OP(need(key, "selector_sample: null output"); *key = sel->impl->sample()); }
int32_t emb_selector_len(emb_selector_t* sel, int64_t* n) { SEL_OP(need(n, "selector_len: null output"); *n = sel->impl->size()); }
int32_t emb_selector_prioritize(emb_selector_t* sel, const uint8_t* stepids, const double* prios, int64_t n) {
  SEL_OP(need(n >= 0 && (n == 0 || (stepids && prios)), "selector_prioritize: bad arguments");
         sel->impl->prioritize(reinterpret_cast<const emb::StepId*>(stepids), prios, n));
}
int32_t emb_selector_destroy(emb_selector_t* sel) {
  delete sel;
  return EMB_OK;
}

}  // extern "C"
