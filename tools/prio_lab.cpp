// Prioritized selector + ReplayIndex at the PPO replay shape (ppo/configs.yaml:42:
// exponent .8, maxfrac .5, initial inf, zero_on_sample), timed inside C++.
//   g++ -O3 -std=c++17 -I embodied_amd/csrc tools/prio_lab.cpp -o tools/build/prio_lab
#include <chrono>
#include <cstdio>
#include <limits>
#include <memory>
#include <vector>

#include "replay_index.h"

int main() {
  const int64_t n = 64, L = 65, cap = 100000;
  emb::ReplayConfig cfg;
  cfg.length = L;
  cfg.capacity = cap;
  cfg.chunksize = 1024;
  cfg.n_slots = (cap + L) / 1024 + 3 * n + 10;
  cfg.online = false;
  auto sel = std::make_shared<emb::Prioritized>(0.8, std::numeric_limits<double>::infinity(), true, 0.5, 16, 0);
  emb::ReplayIndex index(cfg, sel);
  emb::StepId sid;
  auto add_all = [&] { for (int64_t i = 0; i < n; ++i) index.add(i, &sid); };
  for (int64_t t = 0; t < cap / n + 3 * L; ++t) add_all();
  using clock = std::chrono::steady_clock;
  const int iters = 300;
  double adds = 0, draws = 0;
  bool online;
  for (int i = 0; i < iters; ++i) {
    auto a = clock::now();
    add_all();
    auto b = clock::now();
    for (int d = 0; d < 3; ++d) index.draw(true, &online);     // train_ratio 3/1024 * 64 = 0.19 batches of 16
    auto c = clock::now();
    adds += std::chrono::duration<double, std::micro>(b - a).count();
    draws += std::chrono::duration<double, std::micro>(c - b).count();
  }
  std::printf("64 inserts %.2f us; one zero-on-sample draw %.2f us\n", adds / iters, draws / iters / 3);
  return 0;
}
