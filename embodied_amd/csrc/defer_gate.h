// The helper thread that takes a publish's index bookkeeping off the stepping
// thread (emb_replay_publish).  INVARIANTS, stated once:
//
//  1. One job at a time per gate.  `state` is 0 (idle) or 1 (posted or running);
//     only a poster moves it 0 -> 1, only the helper 1 -> 0.
//  2. A job is posted only by a thread that holds the replay's mutex AND the
//     selector handle's mutex, and only after it has drained the gate.
//  3. Every operation that touches the index or the selector holds one of those
//     mutexes and drains the gate first (REP_OP / SEL_OP) -- with one exception:
//     an early insert on PREDICTED rows (emb_replay::Predicted) touches neither
//     and therefore does not drain.
//  4. The job touches the index, the selector and its own inputs (job_workers,
//     job_rows, defer_rows, defer_ids) -- nothing else; the posting thread does
//     not touch those between post() and the next drain().
//  5. Publication: the poster's writes (fn, ctx, the job's inputs) happen-before
//     the helper's reads through the release store / acquire load of `state`;
//     the helper's writes (index, error, job_cycles) happen-before the drainer's
//     reads the same way.  tests/sanitize runs this under ThreadSanitizer.
//  6. The helper is detached and keeps the gate alive (shared_ptr); it leaves
//     when it holds the last reference.  fork(): the parent finishes the job in
//     flight first; the child's gates start helpers of their own (fork epoch).
#pragma once

#include <pthread.h>
#include <sched.h>

#include "abi_common.h"

namespace emb_abi {

// Deferred index work (emb_replay_publish): ONE job at a time, run by a helper
// thread while the caller goes on (its launch, then the interpreter's work up to
// the next library call).  Rules that make it race-free without the helper
// taking a lock: a job is posted only by a thread that holds the replay's and the
// selector handle's mutex; every operation that holds either of them drains the
// gate before it touches the index or the selector; the job touches nothing else.
// The helper spins for a while after a job (the next one is ~15 us away in a
// stepping loop), then sleeps; it is not joined (it keeps the gate alive itself)
// and a forked child starts its own.
inline std::atomic<uint64_t> g_fork_epoch{0};
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
        // (wait_until on the system clock = pthread_cond_timedwait, which every
        // ThreadSanitizer intercepts; wait_for goes through pthread_cond_clockwait,
        // which gcc 11's does not -- it then misses the unlock inside the wait)
        g.cv.wait_until(lock, std::chrono::system_clock::now() + std::chrono::milliseconds(200), [&] {
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
inline std::mutex g_gates_mu;
inline std::vector<std::weak_ptr<DeferGate>> g_gates;
inline void gates_before_fork() {
  std::lock_guard<std::mutex> lock(g_gates_mu);
  for (auto& weak : g_gates)
    if (auto gate = weak.lock())
      while (gate->state.load(std::memory_order_acquire) != 0) sched_yield();
}
inline void gates_in_child() { g_fork_epoch.fetch_add(1); }
inline std::shared_ptr<DeferGate> make_gate() {
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
inline bool predict_rows() {
  static const bool value = [] {
    const char* e = emb::knob("EMB_PREDICT_ROWS");
    return !(e && e[0] == '0');
  }();
  return value;
}

// EMB_DEFER_INDEX=0: emb_replay_publish does its index bookkeeping itself.
// A process confined to one CPU keeps it too: the helper would only take turns
// with the thread that waits for it.
inline bool defer_index() {
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

}  // namespace emb_abi
