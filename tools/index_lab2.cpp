// Component timings of the host index with the prioritized selector at the PPO replay shape
// (ppo/configs.yaml:42: exponent .8, maxfrac .5, initial inf, zero_on_sample): the sample tree alone,
// the index with Prioritized, the index with Uniform -- medians over 2000 iterations, inside C++.
//   g++ -O3 -std=c++17 -I embodied_amd/csrc tools/index_lab2.cpp -o /tmp/index_lab2 && taskset -c 8 /tmp/index_lab2
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <limits>
#include <memory>
#include <vector>
#include "replay_index.h"
using clk = std::chrono::steady_clock;
static double med(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main() {
  const int64_t n = 64, L = 65, cap = 100000;
  const double inf = std::numeric_limits<double>::infinity();
  // 1. the tree alone: 100k leaves, per iteration 64 x (insert newest, remove oldest)
  {
    emb::SampleTree tree(16, 0);
    int64_t next = 0, oldest = 0;
    for (; next < cap; ++next) tree.insert(next, inf);
    std::vector<double> t;
    for (int it = 0; it < 2000; ++it) {
      auto a = clk::now();
      for (int i = 0; i < n; ++i) { tree.insert(next++, inf); tree.remove(oldest++); }
      t.push_back(std::chrono::duration<double, std::micro>(clk::now() - a).count());
    }
    std::printf("tree alone: 64 x (insert + remove) median %.2f us\n", med(t));
    std::vector<double> d;
    for (int it = 0; it < 2000; ++it) {
      auto a = clk::now();
      for (int i = 0; i < 16; ++i) tree.sample();
      d.push_back(std::chrono::duration<double, std::micro>(clk::now() - a).count());
    }
    std::printf("tree alone: 16 x sample (all inf) median %.2f us\n", med(d));
  }
  // 2. the whole index with the prioritized selector
  emb::ReplayConfig cfg;
  cfg.length = L; cfg.capacity = cap; cfg.chunksize = 1024; cfg.n_slots = (cap + L) / 1024 + 3 * n + 10; cfg.online = false;
  auto sel = std::make_shared<emb::Prioritized>(0.8, inf, true, 0.5, 16, 0);
  emb::ReplayIndex index(cfg, sel);
  emb::StepId sid;
  auto add_all = [&] { for (int64_t i = 0; i < n; ++i) index.add(i, &sid); };
  for (int64_t t = 0; t < cap / n + 3 * L; ++t) add_all();
  std::vector<double> adds, draws;
  bool online;
  for (int i = 0; i < 2000; ++i) {
    auto a = clk::now();
    add_all();
    auto b = clk::now();
    if (i % 5 == 0) {
      for (int d = 0; d < 16; ++d) index.draw(true, &online);
      draws.push_back(std::chrono::duration<double, std::micro>(clk::now() - b).count());
    }
    adds.push_back(std::chrono::duration<double, std::micro>(b - a).count());
  }
  std::printf("index + prioritized: 64 adds median %.2f us; 16 zero-on-sample draws median %.2f us\n", med(adds), med(draws));
  // 3. the index with the uniform selector (what the bookkeeping itself costs)
  auto uni = std::make_shared<emb::Uniform>(0);
  emb::ReplayIndex index2(cfg, uni);
  auto add2 = [&] { for (int64_t i = 0; i < n; ++i) index2.add(i, &sid); };
  for (int64_t t = 0; t < cap / n + 3 * L; ++t) add2();
  std::vector<double> u;
  for (int i = 0; i < 2000; ++i) { auto a = clk::now(); add2(); u.push_back(std::chrono::duration<double, std::micro>(clk::now() - a).count()); }
  std::printf("index + uniform: 64 adds median %.2f us\n", med(u));
  return 0;
}
