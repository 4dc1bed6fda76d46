// Host index core alone, timed inside C++ (no interpreter, no ctypes): 64
// inserts per call at the BASELINE replay shape, warm and with the caches
// flushed between calls (what the real loop's interpreter + launches do).
//   g++ -O3 -std=c++17 -I embodied_amd/csrc tools/index_lab.cpp -o tools/build/index_lab
#include <chrono>
#include <cstdio>
#include <cstdlib>
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
  cfg.online = true;
  auto sel = std::make_shared<emb::Uniform>(0);
  emb::ReplayIndex index(cfg, sel);
  std::vector<int64_t> workers(n);
  for (int64_t i = 0; i < n; ++i) workers[i] = i;
  emb::StepId sid;
  auto add_all = [&] {
    for (int64_t i = 0; i < n; ++i) index.add(workers[i], &sid);
  };
  for (int64_t t = 0; t < cap / n + 3 * L; ++t) add_all();
  using clock = std::chrono::steady_clock;
  auto t0 = clock::now();
  const int iters = 3000;
  for (int i = 0; i < iters; ++i) add_all();
  double warm = std::chrono::duration<double, std::micro>(clock::now() - t0).count() / iters;
  std::vector<char> junk(64 << 20);
  double cold = 0;
  volatile char sink = 0;
  for (int i = 0; i < 200; ++i) {
    for (size_t j = 0; j < junk.size(); j += 64) junk[j] += 1;
    sink += junk[i];
    auto a = clock::now();
    add_all();
    cold += std::chrono::duration<double, std::micro>(clock::now() - a).count();
    bool online;
    for (int b = 0; b < 3; ++b) index.draw(true, &online);   // a train step drains the online queue
  }
  std::printf("warm %.2f us per 64 inserts, cold %.2f us\n", warm, cold / 200);
  return 0;
}
