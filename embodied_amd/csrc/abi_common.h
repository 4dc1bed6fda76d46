// What every translation unit behind include/embodied_hip.h shares: the error
// channel of the C ABI (a thread-local message + a status code, no exception
// crosses the boundary), the host-section profiler and the cycle clock.
#pragma once

#include "../../include/embodied_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "knobs.h"
#include "replay_index.h"     // emb::PoolFull

namespace emb_abi {

// EMB_HOST_PROFILE=1: cycle counts of the host sections of the hot entry points,
// printed to stderr when the process ends (tools/profile_step.py's native view).
// Off: one predictable branch per lap.
struct HostProfile {
  enum { kSlots = 32 };
  bool on = false;
  uint64_t cycles[kSlots] = {}, laps[kSlots] = {};
  const char* label[kSlots] = {};
  uint64_t tsc0 = 0;
  timespec wall0{};
  HostProfile() {
    const char* e = emb::knob("EMB_HOST_PROFILE");
    on = e && e[0] == '1';
    if (on) {
      clock_gettime(CLOCK_MONOTONIC, &wall0);
      tsc0 = __builtin_ia32_rdtsc();
    }
  }
  ~HostProfile() {
    if (!on) return;
    timespec now{};
    clock_gettime(CLOCK_MONOTONIC, &now);
    const double ns = (now.tv_sec - wall0.tv_sec) * 1e9 + (now.tv_nsec - wall0.tv_nsec);
    const double per_cycle = ns / static_cast<double>(__builtin_ia32_rdtsc() - tsc0);
    std::fprintf(stderr, "[emb host profile]  section                          laps      ns/lap\n");
    for (int i = 0; i < kSlots; ++i)
      if (laps[i])
        std::fprintf(stderr, "[emb host profile]  %-30s %8llu %10.0f\n", label[i],
                     static_cast<unsigned long long>(laps[i]), cycles[i] * per_cycle / laps[i]);
  }
};
inline HostProfile g_host_profile;
struct HostLap {
  uint64_t t;
  HostLap() : t(g_host_profile.on ? __builtin_ia32_rdtsc() : 0) {}
  void lap(int slot, const char* name) {
    if (!g_host_profile.on) return;
    const uint64_t now = __builtin_ia32_rdtsc();
    g_host_profile.cycles[slot] += now - t;
    g_host_profile.laps[slot] += 1;
    g_host_profile.label[slot] = name;
    t = __builtin_ia32_rdtsc();
  }
};

// Streams made by emb_stream_create_on_cus: how many compute units each may use.
// The persistent movers size their grids by it (a grid for 256 CUs queued onto
// 64 would run in rounds).  A handful of entries, read on every mover launch only
// while there are any.
struct CuStreams {
  std::atomic<int> count{0};
  std::mutex mu;
  std::vector<std::pair<hipStream_t, int>> streams;
  int cus_of(hipStream_t s) {
    if (count.load(std::memory_order_acquire) == 0) return 0;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto& entry : streams)
      if (entry.first == s) return entry.second;
    return 0;
  }
};
inline CuStreams g_cu_streams;

inline thread_local std::string g_error;

inline int32_t fail(int32_t code, const std::string& msg) {
  g_error = msg;
  return code;
}

struct HipFailure : std::runtime_error {
  explicit HipFailure(hipError_t e, const char* what)
      : std::runtime_error(std::string(what) + ": " + hipGetErrorString(e)) {}
};

#define HIP_OK(expr)                                   \
  do {                                                 \
    hipError_t e_ = (expr);                            \
    if (e_ != hipSuccess) throw HipFailure(e_, #expr); \
  } while (0)

template <typename Fn>
int32_t guarded(Fn&& fn) {
  try {
    fn();
    return EMB_OK;
  } catch (const emb::PoolFull& e) {
    return fail(EMB_ERR_POOL_FULL, e.what());
  } catch (const HipFailure& e) {
    return fail(EMB_ERR_HIP, e.what());
  } catch (const std::out_of_range& e) {
    return fail(EMB_ERR_NOT_FOUND, e.what());
  } catch (const std::invalid_argument& e) {
    return fail(EMB_ERR_INVALID, e.what());
  } catch (const std::runtime_error& e) {
    const std::string msg = e.what();
    return fail(msg.find("empty") != std::string::npos ? EMB_ERR_EMPTY : EMB_ERR_INVALID, msg);
  } catch (const std::exception& e) {
    return fail(EMB_ERR_INTERNAL, e.what());
  } catch (...) {
    return fail(EMB_ERR_INTERNAL, "unknown C++ exception");
  }
}

inline void need(bool ok, const char* msg) {
  if (!ok) throw std::invalid_argument(msg);
}

// Ticks of the time-stamp counter per microsecond, measured (not assumed): one
// (tsc, CLOCK_MONOTONIC) pair when the library is loaded, a second one at the
// first question -- at least 200 us later, waited for if need be.
struct TscClock {
  uint64_t tsc0;
  timespec wall0;
  TscClock() {
    clock_gettime(CLOCK_MONOTONIC, &wall0);
    tsc0 = __builtin_ia32_rdtsc();
  }
  double per_us() const {
    static const double value = [this] {
      for (;;) {
        timespec now;
        clock_gettime(CLOCK_MONOTONIC, &now);
        const uint64_t tsc = __builtin_ia32_rdtsc();
        const double us = (now.tv_sec - wall0.tv_sec) * 1e6 + (now.tv_nsec - wall0.tv_nsec) * 1e-3;
        if (us >= 200.0) {
          const double rate = static_cast<double>(tsc - tsc0) / us;
          return rate > 100.0 && rate < 20000.0 ? rate : 3000.0;     // 0.1 .. 20 GHz, else a guess
        }
      }
    }();
    return value;
  }
};
inline const TscClock g_tsc;
inline uint64_t tsc_ticks(double us) { return static_cast<uint64_t>(us * g_tsc.per_us()); }

}  // namespace emb_abi
