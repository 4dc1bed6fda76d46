// extern "C" surface of libembodied_hip.so — see include/embodied_hip.h.
#include "../../include/embodied_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: the symbols are taken with dlsym

#include <dlfcn.h>

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"
#include "knobs.h"
#include "np_random.h"
#include "replay_index.h"
#include "selectors.h"

namespace {

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
HostProfile g_host_profile;
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

thread_local std::string g_error;

int32_t fail(int32_t code, const std::string& msg) {
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

void need(bool ok, const char* msg) {
  if (!ok) throw std::invalid_argument(msg);
}

// Row tables travel host -> device through a small ring of pinned slots.  The
// kernel reads the device copy; a slot is reused only after the launch that
// read it has finished (event per slot).
class TableRing {
 public:
  struct Lease {
    int slot;
    uint8_t* host;
    uint8_t* device;
  };

  ~TableRing() { release_all(); }

  Lease acquire(size_t bytes, hipStream_t stream) {
    if (bytes > cap_) regrow(bytes, stream);
    if (events_.empty()) regrow(cap_ ? cap_ : 4096, stream);
    const int slot = next_;
    next_ = (next_ + 1) % kSlots;
    if (busy_[slot]) {
      HIP_OK(hipEventSynchronize(events_[slot]));
      busy_[slot] = false;
    }
    return {slot, host_ + slot * cap_, dev_ + slot * cap_};
  }

  void upload(const Lease& l, size_t bytes, hipStream_t stream) {
    HIP_OK(hipMemcpyAsync(l.device, l.host, bytes, hipMemcpyHostToDevice, stream));
  }

  void retire(const Lease& l, hipStream_t stream) {
    HIP_OK(hipEventRecord(events_[l.slot], stream));
    busy_[l.slot] = true;
  }

 private:
  static constexpr int kSlots = 32;

  void regrow(size_t bytes, hipStream_t stream) {
    size_t cap = 4096;
    while (cap < bytes) cap *= 2;
    for (int s = 0; s < static_cast<int>(events_.size()); ++s)
      if (busy_[s]) HIP_OK(hipEventSynchronize(events_[s]));
    (void)stream;
    if (host_) HIP_OK(hipHostFree(host_));
    if (dev_) HIP_OK(hipFree(dev_));
    host_ = dev_ = nullptr;
    HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&host_), cap * kSlots, hipHostMallocDefault));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&dev_), cap * kSlots));
    if (events_.empty()) {
      events_.resize(kSlots);
      for (auto& e : events_) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    busy_.assign(kSlots, false);
    cap_ = cap;
  }

  void release_all() {
    for (auto& e : events_) (void)hipEventDestroy(e);
    if (host_) (void)hipHostFree(host_);
    if (dev_) (void)hipFree(dev_);
  }

  size_t cap_ = 0;
  uint8_t* host_ = nullptr;
  uint8_t* dev_ = nullptr;
  std::vector<hipEvent_t> events_;
  std::vector<bool> busy_;
  int next_ = 0;
};

// Argument blocks the CPU writes STRAIGHT into device memory (large-BAR
// systems: all of VRAM is mapped into the host's address space; this is how the
// HIP runtime itself places kernel arguments in device memory).  Fine-grained
// memory, so the GPU reads what the host wrote without an L2 copy in between;
// write-combined stores + a store fence, ordered before the doorbell write of
// the launch that follows.  3.7 KB take ~0.8 us — no upload, no writer kernel.
// A slot is reused only after the launch that read it has finished.
class ArgRing {
 public:
  ~ArgRing() {
    for (auto& e : events_) (void)hipEventDestroy(e);
    if (dev_) (void)hipFree(dev_);
  }
  // False when this device cannot do it (no large BAR, allocation refused, or
  // EMB_ARGS_BAR=0): the caller falls back to the writer kernel.
  bool usable() {
    if (state_ == 0) {
      state_ = -1;
      const char* knob = emb::knob("EMB_ARGS_BAR");
      int dev = 0, large = 0;
      if (!(knob && knob[0] == '0') && hipGetDevice(&dev) == hipSuccess &&
          hipDeviceGetAttribute(&large, hipDeviceAttributeIsLargeBar, dev) == hipSuccess && large &&
          hipExtMallocWithFlags(reinterpret_cast<void**>(&dev_), kSlots * kSlotBytes,
                                hipDeviceMallocFinegrained) == hipSuccess) {
        events_.resize(kSlots / kGroup);
        bool ok = true;
        for (auto& e : events_) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        busy_.assign(kSlots / kGroup, false);
        if (ok) state_ = 1;
      }
      (void)hipGetLastError();
    }
    return state_ == 1;
  }
  // Copies `bytes` (<= 4 KiB) of arguments into the next slot; returns its
  // device address.  Call retire() after the launch that reads it.  Slots are
  // guarded in groups of kGroup with one event per group (an event record costs
  // the host ~3.7 us: once per group, not once per launch): a group is entered
  // again only after the event recorded behind its last launch is done.
  void* put(const void* args, size_t bytes, hipStream_t stream) {
    uint8_t* dst = take(stream);
    std::memcpy(dst, args, bytes);
    __builtin_ia32_sfence();
    return dst;
  }
  // The next slot itself (write-combined device memory: fill it front to back,
  // never read it), for callers that build their block in place; follow with
  // publish() before the launch and retire() after it.
  uint8_t* take(hipStream_t stream) {
    if (next_ % kGroup != 0 && stream != group_stream_) {
      // Another stream takes over in the middle of a group (actor / learner):
      // close the group on the stream that filled it so far, start a new one.
      HIP_OK(hipEventRecord(events_[next_ / kGroup], group_stream_));
      busy_[next_ / kGroup] = true;
      next_ = (next_ / kGroup + 1) * kGroup % kSlots;
    }
    slot_ = next_;
    next_ = (next_ + 1) % kSlots;
    const int group = slot_ / kGroup;
    if (slot_ % kGroup == 0) {
      if (busy_[group]) {
        HIP_OK(hipEventSynchronize(events_[group]));
        busy_[group] = false;
      }
      group_stream_ = stream;
    }
    return dev_ + static_cast<size_t>(slot_) * kSlotBytes;
  }
  static void publish() { __builtin_ia32_sfence(); }
  static constexpr size_t kSlotBytes = 4096;
  // After the launch that reads the slot `put` returned (same stream).
  void retire(hipStream_t stream) {
    if (slot_ % kGroup == kGroup - 1) {
      HIP_OK(hipEventRecord(events_[slot_ / kGroup], stream));
      busy_[slot_ / kGroup] = true;
    }
  }

 private:
  // 512 slots in groups of 64: one event record (~3.7 us of host time) per 64
  // launches; a group is entered again 448 launches after it was closed.
  static constexpr int kSlots = 512, kGroup = 64;
  hipStream_t group_stream_ = nullptr;
  int state_ = 0;          // 0 unknown, 1 usable, -1 not
  uint8_t* dev_ = nullptr;
  std::vector<hipEvent_t> events_;
  std::vector<bool> busy_;
  int next_ = 0, slot_ = 0;
};

struct KeyInfo {
  std::string name;
  int64_t rowbytes;
  uint8_t* pool;
};

// HIP-event pairs around gather launches (bench.py's roofline leg).
class LaunchTimer {
 public:
  ~LaunchTimer() {
    for (auto& p : pairs_) {
      (void)hipEventDestroy(p.first);
      (void)hipEventDestroy(p.second);
    }
  }
  bool enabled = false;
  bool discard = false;   // stamps only, never read: a small ring of pairs reused in turn
  // Stamp one launch in `every` (a stamped launch costs the host a few
  // microseconds more than a plain one: sampling keeps the timed loop close to
  // the un-instrumented one).
  int every = 1;
  uint64_t tick = 0;
  bool due() { return enabled && (tick++ % static_cast<uint64_t>(every) == 0); }
  // Next (start, stop) pair for hipExtLaunchKernelGGL, or nulls when disabled.
  // Pairs come from a fixed pool created when stamping is switched on (event
  // creation is far too slow to happen inside a timed region); when the pool is
  // used up the stamps are read — all but the newest few belong to launches
  // that finished long ago — and the pool starts over.
  void next(hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (!enabled) return;
    if (pairs_.empty()) reserve(discard ? 256 : kPool);
    if (used_ == pairs_.size()) {
      if (discard) used_ = 0;
      else collect();
    }
    *start = pairs_[used_].first;
    *stop = pairs_[used_].second;
    ++used_;
  }
  void reserve(size_t n) {
    while (pairs_.size() < n) {
      hipEvent_t a, b;
      HIP_OK(hipEventCreate(&a));
      HIP_OK(hipEventCreate(&b));
      pairs_.emplace_back(a, b);
    }
  }
  static constexpr size_t kPool = 2048;
  void collect() {
    for (size_t i = 0; i < used_; ++i) {
      HIP_OK(hipEventSynchronize(pairs_[i].second));
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, pairs_[i].first, pairs_[i].second));
      total_ms_ += ms;
      ++launches_;
    }
    used_ = 0;
  }
  void read(int64_t* launches, double* ms, bool reset) {
    collect();
    *launches = launches_;
    *ms = total_ms_;
    if (reset) {
      launches_ = 0;
      total_ms_ = 0;
    }
  }

 private:
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs_;
  size_t used_ = 0;
  int64_t launches_ = 0;
  double total_ms_ = 0;
};

// HIP_FORCE_DEV_KERNARG=0: the runtime leaves kernel arguments in host memory.
bool host_kernargs() {
  static const bool value = [] {
    if (const char* o = emb::knob("EMB_ARGS_INDIRECT")) return o[0] == '1';   // A/B override
    const char* e = std::getenv("HIP_FORCE_DEV_KERNARG");
    return e && e[0] == '0';
  }();
  return value;
}

// While a replay's gather launches are being timed (bench.py's roofline leg),
// its scatter launches carry a completion stamp too (stop event only, from a
// small ring, never read).  Why: a dispatch WITHOUT a completion signal leaves
// its end-of-kernel cache release to the window of the dispatch that follows,
// so a stamped gather behind an unstamped insert reads ~2 us long
// (tools/gather_lab.hip "pipeline study": plain copy 10.2 us alone, 12.2 us
// behind an unstamped tiny kernel, 10.3 us behind one with a stop stamp; wall
// time per pair is the same).  rocprofv3 gives every dispatch a signal, so this
// is also what makes the in-process number agree with the profiler's.
// EMB_STAMP_PRED=0 turns it off.
bool stamp_predecessors() {
  static const bool value = [] {
    const char* e = emb::knob("EMB_STAMP_PRED");
    return !(e && e[0] == '0');
  }();
  return value;
}

std::mutex g_ring_mu;
TableRing& global_ring() {
  static TableRing* ring = new TableRing();  // leaked on purpose: HIP may be gone at exit
  return *ring;
}

}  // namespace

static size_t emb_timer_pool() { return 2048; }

struct emb_rng {
  std::mutex mu;
  emb::NpRandom impl;
  explicit emb_rng(const std::vector<uint32_t>& w) : impl(w) {}
};

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
static const TscClock g_tsc;
static uint64_t tsc_ticks(double us) { return static_cast<uint64_t>(us * g_tsc.per_us()); }

// Deferred index work (emb_replay_publish): ONE job at a time, run by a helper
// thread while the caller goes on (its launch, then the interpreter's work up to
// the next library call).  Rules that make it race-free without the helper
// taking a lock: a job is posted only by a thread that holds the replay's and the
// selector handle's mutex; every operation that holds either of them drains the
// gate before it touches the index or the selector; the job touches nothing else.
// The helper spins for a while after a job (the next one is ~15 us away in a
// stepping loop), then sleeps; it is not joined (it keeps the gate alive itself)
// and a forked child starts its own.
std::atomic<uint64_t> g_fork_epoch{0};
struct DeferGate : std::enable_shared_from_this<DeferGate> {
  std::atomic<int> state{0};             // 0 idle, 1 posted or running
  void (*fn)(void*) = nullptr;
  void* ctx = nullptr;
  std::exception_ptr error;              // written by the helper before state -> 0
  std::mutex m;
  std::condition_variable cv;
  std::atomic<bool> sleeping{false};
  uint64_t started_epoch = ~uint64_t{0};
  pthread_t helper{};
  cpu_set_t helper_cpus;                 // where the helper may run: the poster's L3 group
  bool placed = false;
  // Self-check: cycles the draining threads spent waiting for jobs.  A helper
  // that cannot keep up (no CPU near the poster, an oversubscribed host) costs
  // more than it saves: deferral then pauses for a while.
  uint64_t wait_cycles = 0, jobs = 0, skip = 0;
  uint64_t job_cycles = 0;               // written by the helper while a job runs, read after a drain

  // "0-7,128-135" -> set
  static bool read_cpu_list(const char* path, cpu_set_t* out) {
    CPU_ZERO(out);
    FILE* f = std::fopen(path, "r");
    if (!f) return false;
    char text[512] = {};
    const bool got = std::fgets(text, sizeof(text), f) != nullptr;
    std::fclose(f);
    if (!got) return false;
    for (char* p = text; *p;) {
      char* end = nullptr;
      const long lo = std::strtol(p, &end, 10);
      if (end == p) break;
      long hi = lo;
      p = end;
      if (*p == '-') {
        hi = std::strtol(p + 1, &end, 10);
        p = end;
      }
      for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) CPU_SET(static_cast<int>(c), out);
      while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    }
    return CPU_COUNT(out) > 0;
  }

  // Keep the helper on CPUs that share the posting thread's L3: the two threads
  // hand the workers' records back and forth every step, and across CCXs (or
  // sockets) those cache-line transfers cost more than the job (measured: 4.0 M
  // env steps/s on the calling thread, 1.8-2.3 M with a helper the scheduler
  // had put elsewhere, 4.2-4.4 M with it next door).
  void place_helper() {
    const int cpu = sched_getcpu();
    if (cpu < 0 || (placed && CPU_ISSET(cpu, &helper_cpus))) return;
    char path[128];
    cpu_set_t l3, siblings, allowed, want;
    std::snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    CPU_ZERO(&allowed);
    if (!read_cpu_list(path, &l3) || sched_getaffinity(0, sizeof(allowed), &allowed) != 0) {
      // no cache topology to go by: leave the helper to the scheduler, and do not ask again
      for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &helper_cpus);
      placed = true;
      return;
    }
    CPU_AND(&want, &l3, &allowed);
    helper_cpus = want;                  // (membership test above: includes the poster's own CPU)
    placed = true;
    CPU_CLR(cpu, &want);
    std::snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
    if (read_cpu_list(path, &siblings)) {
      cpu_set_t without;
      CPU_XOR(&without, &want, &siblings);
      CPU_AND(&without, &without, &want);          // want minus the poster's SMT siblings
      if (CPU_COUNT(&without) > 0) want = without;
    }
    if (CPU_COUNT(&want) > 0) (void)pthread_setaffinity_np(helper, sizeof(want), &want);
  }

  // May this publish be deferred?  (caller holds the mutexes)  Only in a loop
  // that publishes every few tens of microseconds (< ~80 us): the helper polls between
  // jobs, and a poll that lasts a 100 us step of host simulators takes a CPU
  // from them for a 3 us job (measured with 64 env processes: 490 -> 270-370 k).
  // EMB_DEFER_MAX_GAP_US replaces the 80 us (the test suite steps slowly and
  // wants the deferred path all the same).
  uint64_t last_publish = 0;
  bool allowed() {
    static const uint64_t max_gap = [] {
      const char* e = emb::knob("EMB_DEFER_MAX_GAP_US");
      const double us = e ? std::atof(e) : 80.0;
      return us >= 1e9 ? ~uint64_t{0} : tsc_ticks(us);
    }();
    const uint64_t now = __builtin_ia32_rdtsc();
    const bool quick = now - last_publish < max_gap;
    last_publish = now;
    if (skip > 0) {
      --skip;
      return false;
    }
    return quick;
  }

  static void* run(void* self_owned) {
    std::shared_ptr<DeferGate> self(*static_cast<std::shared_ptr<DeferGate>*>(self_owned));
    delete static_cast<std::shared_ptr<DeferGate>*>(self_owned);
    DeferGate& g = *self;
    for (;;) {
      int spins = 0;
      while (g.state.load(std::memory_order_acquire) != 1 || g.fn == nullptr) {
        if (self.use_count() == 1) return nullptr;      // every replay / selector handle is gone
        if (++spins < 40000) {           // 0.1 - 1 ms of polling (a `pause` is 10 - 65 cycles by core), then sleep
          __builtin_ia32_pause();
          continue;
        }
        std::unique_lock<std::mutex> lock(g.m);
        g.sleeping.store(true);
        g.cv.wait_for(lock, std::chrono::milliseconds(200), [&] {
          return g.state.load() == 1;
        });
        g.sleeping.store(false);
        spins = 0;
      }
      void (*fn)(void*) = g.fn;
      g.fn = nullptr;
      const uint64_t began = __builtin_ia32_rdtsc();
      try {
        fn(g.ctx);
      } catch (...) {
        g.error = std::current_exception();
      }
      g.job_cycles += __builtin_ia32_rdtsc() - began;
      g.state.store(0, std::memory_order_release);
    }
  }

  // Caller holds the mutexes named above and has drained.
  void post(void (*f)(void*), void* c) {
    const uint64_t epoch = g_fork_epoch.load();
    if (started_epoch != epoch) {        // first job, or first job in a forked child
      auto* owned = new std::shared_ptr<DeferGate>(shared_from_this());
      pthread_t th;
      pthread_attr_t attr;
      pthread_attr_init(&attr);
      pthread_attr_setdetachstate(&attr, PTHREAD_CREATE_DETACHED);
      if (pthread_create(&th, &attr, &DeferGate::run, owned) != 0) {
        pthread_attr_destroy(&attr);
        delete owned;
        f(c);                            // no helper: do it here
        return;
      }
      pthread_attr_destroy(&attr);
      started_epoch = epoch;
      helper = th;
      placed = false;
    }
    place_helper();
    if (++jobs >= 4096) {
      // Deferral pays while the stepping thread waits for less than the jobs
      // take (it would have spent that time doing them): pause -- the next 2^15
      // publishes do their bookkeeping themselves, then try again -- only when it
      // waited more than ~2 us per job AND more than three quarters of the jobs'
      // own time (a uniform selector's 3 us job waited for in full is a loss: the
      // index then lives in two cores' caches; a prioritized selector's 9 us job
      // with a 5 us wait still saves 4).  (This thread has drained: the helper is
      // idle and job_cycles is complete.)
      if (wait_cycles / jobs > tsc_ticks(2.0) && wait_cycles > job_cycles / 4 * 3) skip = uint64_t{1} << 15;
      wait_cycles = jobs = job_cycles = 0;
    }
    ctx = c;
    fn = f;
    state.store(1, std::memory_order_seq_cst);
    if (sleeping.load(std::memory_order_seq_cst)) {
      std::lock_guard<std::mutex> lock(m);
      cv.notify_one();
    }
  }

  void drain() {
    if (state.load(std::memory_order_acquire) == 0 && !error) return;
    const uint64_t began = __builtin_ia32_rdtsc();
    for (int spins = 0; state.load(std::memory_order_acquire) != 0; ++spins) {
      if (spins < 4000) __builtin_ia32_pause();
      else sched_yield();                // the helper may be waiting for this very CPU
    }
    // (one long wait -- the helper lost its CPU for a time slice -- counts like a
    // slow job, not like a thousand of them)
    wait_cycles += std::min<uint64_t>(__builtin_ia32_rdtsc() - began, tsc_ticks(8.0));
    if (error) {
      std::exception_ptr e = error;
      error = nullptr;
      std::rethrow_exception(e);
    }
  }
};

// fork(): the parent finishes the job in flight first (the child would wait for a
// helper it does not have); the child's gates start helpers of their own.
std::mutex g_gates_mu;
std::vector<std::weak_ptr<DeferGate>> g_gates;
void gates_before_fork() {
  std::lock_guard<std::mutex> lock(g_gates_mu);
  for (auto& weak : g_gates)
    if (auto gate = weak.lock())
      while (gate->state.load(std::memory_order_acquire) != 0) sched_yield();
}
void gates_in_child() { g_fork_epoch.fetch_add(1); }
std::shared_ptr<DeferGate> make_gate() {
  static const bool hooked = [] {
    pthread_atfork(&gates_before_fork, nullptr, &gates_in_child);
    return true;
  }();
  (void)hooked;
  auto gate = std::make_shared<DeferGate>();
  std::lock_guard<std::mutex> lock(g_gates_mu);
  g_gates.erase(std::remove_if(g_gates.begin(), g_gates.end(),
                               [](const std::weak_ptr<DeferGate>& w) { return w.expired(); }),
                g_gates.end());
  g_gates.push_back(gate);
  return gate;
}

// EMB_PREDICT_ROWS=0: every early insert waits for the helper thread and reads
// the cursors (the A/B of the predicted rows, emb_replay::Predicted).
bool predict_rows() {
  static const bool value = [] {
    const char* e = emb::knob("EMB_PREDICT_ROWS");
    return !(e && e[0] == '0');
  }();
  return value;
}

// EMB_DEFER_INDEX=0: emb_replay_publish does its index bookkeeping itself.
// A process confined to one CPU keeps it too: the helper would only take turns
// with the thread that waits for it.
bool defer_index() {
  static const bool value = [] {
    const char* e = emb::knob("EMB_DEFER_INDEX");
    if (e && e[0] == '0') return false;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) < 2) return false;
    return true;
  }();
  return value;
}

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

// Pool accesses from several HIP streams (actor / learner split: inserts on one
// stream, sample + write-back on another) ordered with as few events as the
// hazards need -- an event record costs the host ~4 us, a stepping loop inserts
// every ~12 us:
//   read  (sample gather)            after every earlier WRITE on another stream;
//   write to live rows (update,      after every earlier write AND read on another
//     scatter_rows)                    stream;
//   write to fresh rows (add: rows   after the other streams' work only when a chunk
//     of the workers' open chunks)     slot has been opened since THIS stream's last look --
//                                      rows of an open chunk belong to no item, so no
//                                      gather reads them and no write-back targets
//                                      them, unless the slot was recycled.
// Nothing is recorded when work is issued: counters only.  The stream that has to
// wait records an event on the OTHER stream at that moment (it covers everything
// queued there so far) and waits for it.  The caller holds the replay's mutex.
struct StreamOrder {
  enum { kRead = 0, kWriteLive = 1, kWriteFresh = 2 };
  static constexpr int kMax = 6;
  struct Entry {
    hipStream_t stream = nullptr;
    uint64_t writes = 0, reads = 0;          // issued so far
    hipEvent_t event = nullptr;
    uint64_t cover_w = 0, cover_r = 0;       // what the event's last record covers
  };
  Entry e[kMax];
  uint64_t seen_w[kMax][kMax] = {}, seen_r[kMax][kMax] = {};    // [waiter][other]
  int n = 0;
  int64_t opens_seen[kMax] = {-1, -1, -1, -1, -1, -1};           // per inserting stream

  ~StreamOrder() {
    for (int i = 0; i < n; ++i)
      if (e[i].event) (void)hipEventDestroy(e[i].event);
  }
  int entry(hipStream_t stream) {
    for (int i = 0; i < n; ++i)
      if (e[i].stream == stream) return i;
    if (n == kMax) {
      // More streams than the table holds (not a stepping loop any more): drain
      // the device and start over.
      HIP_OK(hipDeviceSynchronize());
      for (int i = 0; i < n; ++i) {
        e[i].stream = nullptr;
        e[i].writes = e[i].reads = e[i].cover_w = e[i].cover_r = 0;
      }
      std::memset(seen_w, 0, sizeof(seen_w));
      std::memset(seen_r, 0, sizeof(seen_r));
      std::fill(opens_seen, opens_seen + kMax, int64_t{-1});
      n = 0;
    }
    e[n].stream = stream;
    if (!e[n].event) HIP_OK(hipEventCreateWithFlags(&e[n].event, hipEventDisableTiming));
    return n++;
  }
  void wait_for(int i, int j) {
    Entry& other = e[j];
    if (other.cover_w < other.writes || other.cover_r < other.reads) {
      HIP_OK(hipEventRecord(other.event, other.stream));
      other.cover_w = other.writes;
      other.cover_r = other.reads;
    }
    HIP_OK(hipStreamWaitEvent(e[i].stream, other.event, 0));
    seen_w[i][j] = other.cover_w;
    seen_r[i][j] = other.cover_r;
  }
  void before(int kind, hipStream_t stream, int64_t chunks_opened) {
    const int i = entry(stream);
    bool reads_too = kind == kWriteLive;
    if (kind == kWriteFresh) {
      if (chunks_opened == opens_seen[i]) return;
      opens_seen[i] = chunks_opened;
      reads_too = true;
    }
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      if (e[j].writes > seen_w[i][j] || (reads_too && e[j].reads > seen_r[i][j])) wait_for(i, j);
    }
  }
  void after(int kind, hipStream_t stream) {
    Entry& mine = e[entry(stream)];
    if (kind == kRead) ++mine.reads;
    else ++mine.writes;
  }
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

extern "C" {

const char* emb_last_error(void) { return g_error.c_str(); }
int32_t emb_abi_version(void) { return EMB_ABI_VERSION; }

int32_t emb_configure(const char* name, const char* value) {
  return guarded([&] {
    need(name && std::strncmp(name, "EMB_", 4) == 0, "configure: knob names start with EMB_");
    if (emb::knob_set(name, value) != 0)
      throw std::invalid_argument(std::string("configure: ") + name +
                                  " is already in effect (knobs are read once: set them before the "
                                  "first call that uses them)");
  });
}

int32_t emb_device_count(int32_t* count) {
  return guarded([&] {
    need(count, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
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
    // EMB_ARGS_DEVICE_MIN: smallest move (bytes) that gets a device copy of its arguments.
    // Default 4 MB for writes (a plain insert of 64 Atari steps, 1.8 MB, is cheaper
    // on the host without the copy) and 1 MB for gathers: a B = 1 or 2 sample of
    // 65 x 28 KB steps (1.8 / 3.7 MB) takes 8.8 us with its 4 KB of arguments in
    // host memory and 4.9 us with the device copy, for 0.4 us of host time.
    static const bool device_min_given = emb::knob("EMB_ARGS_DEVICE_MIN") != nullptr;
    static const int64_t device_min = [] {
      const char* e = emb::knob("EMB_ARGS_DEVICE_MIN");
      return e ? std::atoll(e) : int64_t{4} << 20;
    }();
    const int64_t least = (gather && !device_min_given) ? (int64_t{1} << 20) : device_min;
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

// ------------------------------------------------------------------ kernels --

int32_t emb_obs_stack(const void* src, const int32_t* env_ids, int64_t n, int64_t pixels,
                      int64_t channels, int32_t layout, int32_t out_dtype, float scale,
                      float offset, void* dst, void* stream) {
  return guarded([&] {
    need(src && dst && n >= 0 && pixels > 0 && channels > 0, "obs_stack: bad arguments");
    need(layout == EMB_LAYOUT_SAME || layout == EMB_LAYOUT_CHANNELS_FIRST, "obs_stack: bad layout");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!env_ids) {
      HIP_OK(emb::launch_obs_stack(static_cast<const uint8_t*>(src), nullptr, dst, n, pixels,
                                   channels, layout, out_dtype, scale, offset, s));
      return;
    }
    std::lock_guard<std::mutex> lock(g_ring_mu);
    auto lease = global_ring().acquire(n * sizeof(int32_t), s);
    std::memcpy(lease.host, env_ids, n * sizeof(int32_t));
    global_ring().upload(lease, n * sizeof(int32_t), s);
    HIP_OK(emb::launch_obs_stack(static_cast<const uint8_t*>(src),
                                 reinterpret_cast<const int32_t*>(lease.device), dst, n, pixels,
                                 channels, layout, out_dtype, scale, offset, s));
    global_ring().retire(lease, s);
  });
}

int32_t emb_mask_actions(const void* act, void* out, int64_t n, int64_t row_elems, int32_t dtype,
                         const void* is_last, void* stream) {
  return guarded([&] {
    need(act && out && is_last && n >= 0 && row_elems >= 0, "mask_actions: bad arguments");
    HIP_OK(emb::launch_mask_rows(act, out, n, row_elems, dtype, static_cast<const uint8_t*>(is_last),
                                 static_cast<hipStream_t>(stream)));
  });
}

static void rows_move(void* table, int64_t rowbytes, const int32_t* ids, int64_t n, void* batch,
                      bool gather, hipStream_t s) {
  need(table && batch && ids && rowbytes > 0 && n >= 0, "rows_gather/scatter: bad arguments");
  if (n == 0) return;
  std::lock_guard<std::mutex> lock(g_ring_mu);
  auto lease = global_ring().acquire(n * sizeof(int32_t), s);
  std::memcpy(lease.host, ids, n * sizeof(int32_t));
  global_ring().upload(lease, n * sizeof(int32_t), s);
  emb::MovePlan plan;
  plan.n_keys = 1;
  plan.key[0] = {static_cast<uint8_t*>(table), static_cast<uint8_t*>(batch), rowbytes};
  plan.n_rows = static_cast<int32_t>(n);
  plan.rows = reinterpret_cast<const int32_t*>(lease.device);
  HIP_OK(gather ? emb::launch_gather(plan, s) : emb::launch_scatter(plan, s));
  global_ring().retire(lease, s);
}

int32_t emb_rows_gather(const void* table, int64_t rowbytes, const int32_t* ids, int64_t n, void* dst,
                        void* stream) {
  return guarded([&] { rows_move(const_cast<void*>(table), rowbytes, ids, n, dst, true, static_cast<hipStream_t>(stream)); });
}

int32_t emb_rows_scatter(void* table, int64_t rowbytes, const int32_t* ids, int64_t n, const void* src,
                         void* stream) {
  return guarded([&] { rows_move(table, rowbytes, ids, n, const_cast<void*>(src), false, static_cast<hipStream_t>(stream)); });
}

int32_t emb_window(const void* src, void* dst, int64_t batch, int64_t total, int64_t start,
                   int64_t count, int64_t rowbytes, void* stream) {
  return guarded([&] {
    need(src && dst && batch >= 0 && start >= 0 && count >= 0 && start + count <= total && rowbytes > 0,
         "window: bad arguments");
    HIP_OK(emb::launch_window(static_cast<const uint8_t*>(src), static_cast<uint8_t*>(dst), batch,
                              total, start, count, rowbytes, static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_window_keys(int32_t n_keys, const void* const* src, void* const* dst,
                        const int64_t* rowbytes, int64_t batch, int64_t total, int64_t start,
                        int64_t count, void* stream) {
  return guarded([&] {
    need(n_keys >= 1 && src && dst && rowbytes && batch >= 0 && start >= 0 && count >= 0 &&
         start + count <= total, "window_keys: bad arguments");
    if (batch == 0 || count == 0) return;
    need(batch * total <= INT32_MAX, "window_keys: batch too large");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // A window is a gather whose "pool" is the source batch: sequence b is the
    // single span {b * total + start, count}.  All keys move in one launch.
    std::vector<int32_t> spans(3 * batch), rows;
    for (int64_t b = 0; b < batch; ++b) {
      spans[3 * b] = static_cast<int32_t>(b * total + start);
      spans[3 * b + 1] = static_cast<int32_t>(count);
      spans[3 * b + 2] = 0;
    }
    for (int lo = 0; lo < n_keys; lo += emb::kMaxKeys) {
      emb::MovePlan plan;
      for (int k = lo; k < std::min(n_keys, lo + emb::kMaxKeys); ++k) {
        need(src[k] && dst[k] && rowbytes[k] > 0, "window_keys: bad key");
        plan.key[plan.n_keys++] = {const_cast<uint8_t*>(static_cast<const uint8_t*>(src[k])),
                                   static_cast<uint8_t*>(dst[k]), rowbytes[k]};
      }
      plan.seq_len = static_cast<int32_t>(count);
      plan.n_rows = static_cast<int32_t>(batch * count);
      plan.spans_host = spans.data();
      plan.n_seq = static_cast<int32_t>(batch);
      if (emb::plan_fits_inline(plan)) {
        HIP_OK(emb::launch_gather(plan, s));
        continue;
      }
      if (rows.empty()) {
        rows.resize(batch * count);
        for (int64_t b = 0; b < batch; ++b)
          for (int64_t j = 0; j < count; ++j)
            rows[b * count + j] = static_cast<int32_t>(b * total + start + j);
      }
      plan.spans_host = nullptr;
      plan.n_seq = 0;
      std::lock_guard<std::mutex> lock(g_ring_mu);
      auto lease = global_ring().acquire(rows.size() * sizeof(int32_t), s);
      std::memcpy(lease.host, rows.data(), rows.size() * sizeof(int32_t));
      global_ring().upload(lease, rows.size() * sizeof(int32_t), s);
      plan.rows = reinterpret_cast<const int32_t*>(lease.device);
      HIP_OK(emb::launch_gather(plan, s));
      global_ring().retire(lease, s);
    }
  });
}

int32_t emb_scan_gae(const void* rew, const void* val, const void* last, const void* term, int64_t B,
                     int64_t T, float live_scale, float lam, void* adv, void* tar, void* stream) {
  return guarded([&] {
    need(rew && val && last && term && adv && tar && B >= 0 && T >= 1, "scan_gae: bad arguments");
    HostLap hp;
    HIP_OK(emb::launch_gae(static_cast<const float*>(rew), static_cast<const float*>(val),
                           static_cast<const uint8_t*>(last), static_cast<const uint8_t*>(term), B, T,
                           live_scale, lam, static_cast<float*>(adv), static_cast<float*>(tar),
                           static_cast<hipStream_t>(stream)));
    hp.lap(20, "gae: launch");
  });
}

int32_t emb_scan_gae_grouped(const void* rew, const void* val, const void* last, const void* term,
                             int64_t B, int64_t T, float live_scale, float lam, void* adv,
                             void* tar, int64_t group, int64_t group_stride, void* stream) {
  return guarded([&] {
    need(rew && val && last && term && adv && tar && B >= 0 && T >= 1 && group >= 0 &&
             group_stride >= 0 && group_stride % 4 == 0, "scan_gae_grouped: bad arguments");
    HIP_OK(emb::launch_gae(static_cast<const float*>(rew), static_cast<const float*>(val),
                           static_cast<const uint8_t*>(last), static_cast<const uint8_t*>(term), B, T,
                           live_scale, lam, static_cast<float*>(adv), static_cast<float*>(tar),
                           static_cast<hipStream_t>(stream), group, group_stride));
  });
}

int32_t emb_scan_lambda(const void* last, const void* term, const void* rew, const void* boot,
                        int64_t B, int64_t T, float disc, float lam, void* ret, void* stream) {
  return guarded([&] {
    need(last && term && rew && boot && ret && B >= 0 && T >= 1, "scan_lambda: bad arguments");
    HIP_OK(emb::launch_lambda_return(static_cast<const uint8_t*>(last), static_cast<const uint8_t*>(term),
                                     static_cast<const float*>(rew), static_cast<const float*>(boot), B, T,
                                     disc, lam, static_cast<float*>(ret), static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_scan_lambda_multi(int32_t n_problems, const emb_lambda_problem_t* problems, void* stream) {
  return guarded([&] {
    need(n_problems >= 0 && (problems || n_problems == 0), "scan_lambda_multi: bad arguments");
    std::vector<emb::LambdaProblem> list;
    list.reserve(n_problems);
    for (int i = 0; i < n_problems; ++i) {
      const emb_lambda_problem_t& q = problems[i];
      need(q.B >= 0 && q.T >= 1, "scan_lambda_multi: bad shape");
      if (q.B == 0 || q.T < 2) continue;
      need(q.last && q.term && q.rew && q.boot && q.ret, "scan_lambda_multi: null buffer");
      list.push_back({static_cast<const uint8_t*>(q.last), static_cast<const uint8_t*>(q.term),
                      static_cast<const float*>(q.rew), static_cast<const float*>(q.boot),
                      static_cast<float*>(q.ret), q.B, q.T, q.disc, q.lam});
    }
    HostLap hp;
    HIP_OK(emb::launch_lambda_return_multi(static_cast<int>(list.size()), list.data(),
                                           static_cast<hipStream_t>(stream)));
    hp.lap(23, "lambda-return (multi): launch");
  });
}

int32_t emb_scan_director(const void* rew, const void* cont, const void* value, int64_t T, int64_t B,
                          float discount, float lam, void* ret, void* stream) {
  return guarded([&] {
    need(rew && cont && value && ret && B >= 0 && T >= 1, "scan_director: bad arguments");
    HIP_OK(emb::launch_director_score(static_cast<const float*>(rew), static_cast<const float*>(cont),
                                      static_cast<const float*>(value), T, B, discount, lam,
                                      static_cast<float*>(ret), static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_abstract_traj(const void* reward, const void* cont, int64_t T, int64_t B, int32_t k,
                          void* reward_out, void* cont_out, void* stream) {
  return guarded([&] {
    need(cont && (reward || !reward_out) && T >= 1 && B >= 0 && k >= 1, "abstract_traj: bad arguments");
    HIP_OK(emb::launch_abstract_traj(static_cast<const float*>(reward), static_cast<const float*>(cont),
                                     T, B, k, static_cast<float*>(reward_out),
                                     static_cast<float*>(cont_out), static_cast<hipStream_t>(stream)));
  });
}

int32_t emb_synth_env_step(void* image, void* reward, void* is_first, void* is_last, void* is_terminal,
                           int64_t n, int64_t frame_bytes, int64_t env0, int64_t episode_len,
                           const void* reset, void* counters, int32_t turn, void* stream) {
  return guarded([&] {
    need(image && reward && is_first && is_last && is_terminal && counters && n >= 0 && episode_len >= 1,
         "synth_env_step: bad arguments");
    HostLap hp;
    HIP_OK(emb::launch_synth_env(static_cast<uint8_t*>(image), static_cast<float*>(reward),
                                 static_cast<uint8_t*>(is_first), static_cast<uint8_t*>(is_last),
                                 static_cast<uint8_t*>(is_terminal), n, frame_bytes, env0, episode_len,
                                 static_cast<const uint8_t*>(reset), static_cast<int32_t*>(counters),
                                 turn, static_cast<hipStream_t>(stream)));
    hp.lap(21, "synthetic env: launch");
  });
}

}  // extern "C"

// -------------------------------------------------------------- collectives --

namespace {

struct Rccl {
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

// One RCCL per process: the copy that is already loaded (torch bundles one with
// the same SONAME) if there is one, else the system's.
const Rccl& rccl() {
  static const Rccl table = [] {
    void* lib = nullptr;
    // EMB_RCCL_LIB=<path>: bind the ten symbols below from that library and no
    // other (a site's own RCCL build; the suite's loopback transport between
    // processes that share one GPU, tests/fake_rccl/).  No fallback: a path
    // that does not load is an error.
    if (const char* chosen = emb::knob("EMB_RCCL_LIB"); chosen && *chosen) {
      lib = dlopen(chosen, RTLD_NOW | RTLD_LOCAL);
      if (!lib) throw std::runtime_error(std::string("EMB_RCCL_LIB: cannot load ") + chosen + ": " + dlerror());
    }
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!lib) lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) throw std::runtime_error(std::string("cannot load RCCL: ") + dlerror());
    Rccl t;
    auto sym = [&](const char* name) {
      void* p = dlsym(lib, name);
      if (!p) throw std::runtime_error(std::string("RCCL lacks ") + name);
      return p;
    };
    t.get_unique_id = reinterpret_cast<decltype(t.get_unique_id)>(sym("ncclGetUniqueId"));
    t.comm_init_rank = reinterpret_cast<decltype(t.comm_init_rank)>(sym("ncclCommInitRank"));
    t.comm_destroy = reinterpret_cast<decltype(t.comm_destroy)>(sym("ncclCommDestroy"));
    t.all_gather = reinterpret_cast<decltype(t.all_gather)>(sym("ncclAllGather"));
    t.all_reduce = reinterpret_cast<decltype(t.all_reduce)>(sym("ncclAllReduce"));
    t.send = reinterpret_cast<decltype(t.send)>(sym("ncclSend"));
    t.recv = reinterpret_cast<decltype(t.recv)>(sym("ncclRecv"));
    t.group_start = reinterpret_cast<decltype(t.group_start)>(sym("ncclGroupStart"));
    t.group_end = reinterpret_cast<decltype(t.group_end)>(sym("ncclGroupEnd"));
    t.error_string = reinterpret_cast<decltype(t.error_string)>(sym("ncclGetErrorString"));
    return t;
  }();
  return table;
}

void rccl_ok(ncclResult_t r, const char* what) {
  if (r != ncclSuccess)
    throw std::runtime_error(std::string(what) + ": " + rccl().error_string(r));
}

}  // namespace

struct emb_comm {
  ncclComm_t comm = nullptr;
  int32_t rank = 0, world = 1;
  // emb_comm_exchange: the communicator's own stream, so that a train step's
  // collectives overlap whatever the caller's stream does next.
  hipStream_t side = nullptr;
  hipEvent_t forked = nullptr, done = nullptr;
  bool in_flight = false;
};

static void alltoall_on(emb_comm* comm, const void* send, void* recv, int64_t bytes_per_rank,
                        hipStream_t s) {
  const auto* from = static_cast<const uint8_t*>(send);
  auto* to = static_cast<uint8_t*>(recv);
  const size_t n = static_cast<size_t>(bytes_per_rank);
  // One fused group of point-to-point transfers: on xGMI every pair of GPUs
  // has its own link, so the n-1 blocks leave on n-1 links at once.
  rccl_ok(rccl().group_start(), "ncclGroupStart");
  ncclResult_t first = ncclSuccess;
  for (int32_t peer = 0; peer < comm->world && first == ncclSuccess; ++peer) {
    first = rccl().send(from + peer * n, n, ncclUint8, peer, comm->comm, s);
    if (first == ncclSuccess) first = rccl().recv(to + peer * n, n, ncclUint8, peer, comm->comm, s);
  }
  const ncclResult_t closed = rccl().group_end();
  rccl_ok(first, "ncclSend/ncclRecv");
  rccl_ok(closed, "ncclGroupEnd");
}

static ncclDataType_t grad_type(int32_t dtype) {
  switch (dtype) {
    case EMB_F32: return ncclFloat32;
    case EMB_BF16: return ncclBfloat16;
    case EMB_F16: return ncclFloat16;
    case EMB_F64: return ncclFloat64;
    default: need(false, "comm_allreduce_grads: dtype must be f16, bf16, f32 or f64");
  }
  return ncclFloat32;
}

extern "C" {

int32_t emb_comm_unique_id(uint8_t* id_out) {
  return guarded([&] {
    need(id_out, "comm_unique_id: null output");
    static_assert(sizeof(ncclUniqueId) == EMB_COMM_ID_BYTES, "RCCL id size");
    ncclUniqueId id;
    rccl_ok(rccl().get_unique_id(&id), "ncclGetUniqueId");
    std::memcpy(id_out, &id, sizeof(id));
  });
}

int32_t emb_comm_init(const uint8_t* id, int32_t rank, int32_t world, emb_comm_t** out) {
  return guarded([&] {
    need(id && out && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    auto comm = std::make_unique<emb_comm>();
    comm->rank = rank;
    comm->world = world;
    rccl_ok(rccl().comm_init_rank(&comm->comm, world, uid, rank), "ncclCommInitRank");
    HIP_OK(hipStreamCreateWithFlags(&comm->side, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&comm->forked, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&comm->done, hipEventDisableTiming));
    *out = comm.release();
  });
}

int32_t emb_comm_allgather_traj(emb_comm_t* comm, const void* send, void* recv,
                                int64_t bytes_per_rank, void* stream) {
  return guarded([&] {
    need(comm && send && recv && bytes_per_rank >= 0, "comm_allgather_traj: bad arguments");
    if (bytes_per_rank == 0) return;
    rccl_ok(rccl().all_gather(send, recv, static_cast<size_t>(bytes_per_rank), ncclUint8, comm->comm,
                              static_cast<hipStream_t>(stream)),
            "ncclAllGather");
  });
}

int32_t emb_comm_allgather_returns(emb_comm_t* comm, const void* send, void* recv, int64_t count,
                                   void* stream) {
  return guarded([&] {
    need(comm && send && recv && count >= 0, "comm_allgather_returns: bad arguments");
    if (count == 0) return;
    rccl_ok(rccl().all_gather(send, recv, static_cast<size_t>(count), ncclFloat32, comm->comm,
                              static_cast<hipStream_t>(stream)),
            "ncclAllGather");
  });
}

int32_t emb_comm_pmean_scalars(emb_comm_t* comm, void* values, int64_t count, void* stream) {
  return guarded([&] {
    need(comm && values && count >= 0, "comm_pmean_scalars: bad arguments");
    if (count == 0) return;
    rccl_ok(rccl().all_reduce(values, values, static_cast<size_t>(count), ncclFloat32, ncclAvg,
                              comm->comm, static_cast<hipStream_t>(stream)),
            "ncclAllReduce");
  });
}

int32_t emb_comm_alltoall_slices(emb_comm_t* comm, const void* send, void* recv,
                                 int64_t bytes_per_rank, void* stream) {
  return guarded([&] {
    need(comm && send && recv && bytes_per_rank >= 0, "comm_alltoall_slices: bad arguments");
    if (bytes_per_rank == 0) return;
    alltoall_on(comm, send, recv, bytes_per_rank, static_cast<hipStream_t>(stream));
  });
}

static int32_t allreduce_typed(emb_comm_t* comm, void* buf, int64_t count, int32_t dtype,
                               int32_t mean, void* stream) {
  return guarded([&] {
    need(comm && buf && count >= 0, "comm_allreduce_grads: bad arguments");
    const ncclDataType_t type = grad_type(dtype);
    if (count == 0) return;
    rccl_ok(rccl().all_reduce(buf, buf, static_cast<size_t>(count), type,
                              mean ? ncclAvg : ncclSum, comm->comm, static_cast<hipStream_t>(stream)),
            "ncclAllReduce");
  });
}

int32_t emb_comm_allreduce_grads(emb_comm_t* comm, void* buf, int64_t count, int32_t mean,
                                 void* stream) {
  return allreduce_typed(comm, buf, count, EMB_F32, mean, stream);
}

int32_t emb_comm_allreduce_grads_as(emb_comm_t* comm, void* buf, int64_t count, int32_t dtype,
                                    int32_t mean, void* stream) {
  return allreduce_typed(comm, buf, count, dtype, mean, stream);
}

int32_t emb_comm_exchange(emb_comm_t* comm, void* after_stream, const void* slices_send,
                          void* slices_recv, int64_t bytes_per_rank, void* grads, int64_t count,
                          int32_t dtype, int32_t mean) {
  return guarded([&] {
    need(comm && bytes_per_rank >= 0 && count >= 0, "comm_exchange: bad arguments");
    need(bytes_per_rank == 0 || (slices_send && slices_recv), "comm_exchange: null slice buffers");
    need(count == 0 || grads, "comm_exchange: null gradient buffer");
    const ncclDataType_t type = grad_type(count ? dtype : EMB_F32);
    if (bytes_per_rank == 0 && count == 0) return;
    HIP_OK(hipEventRecord(comm->forked, static_cast<hipStream_t>(after_stream)));
    HIP_OK(hipStreamWaitEvent(comm->side, comm->forked, 0));
    if (bytes_per_rank) alltoall_on(comm, slices_send, slices_recv, bytes_per_rank, comm->side);
    if (count)
      rccl_ok(rccl().all_reduce(grads, grads, static_cast<size_t>(count), type,
                                mean ? ncclAvg : ncclSum, comm->comm, comm->side),
              "ncclAllReduce");
    HIP_OK(hipEventRecord(comm->done, comm->side));
    comm->in_flight = true;
  });
}

int32_t emb_comm_wait(emb_comm_t* comm, void* stream) {
  return guarded([&] {
    need(comm, "comm_wait: null communicator");
    if (!comm->in_flight) return;
    HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), comm->done, 0));
    comm->in_flight = false;
  });
}

int32_t emb_comm_destroy(emb_comm_t* comm) {
  return guarded([&] {
    if (!comm) return;
    if (comm->side) {
      (void)hipStreamSynchronize(comm->side);
      (void)hipEventDestroy(comm->forked);
      (void)hipEventDestroy(comm->done);
      (void)hipStreamDestroy(comm->side);
    }
    if (comm->comm) rccl_ok(rccl().comm_destroy(comm->comm), "ncclCommDestroy");
    delete comm;
  });
}

}  // extern "C"
