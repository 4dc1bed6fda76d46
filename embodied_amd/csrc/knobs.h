// Process-wide tuning knobs of libembodied_hip.so.
//
// Every knob has one name (EMB_...), is read ONCE, the first time the code that
// uses it runs, and can be set in two ways:
//   * emb_configure("EMB_...", "value")  (include/embodied_hip.h) -- from the
//     host program, before the knob's first use; refused afterwards, so a
//     setting can never be half in effect;
//   * the environment variable of the same name (what the shell A/Bs use).
// emb_configure wins over the environment.  INTEGRATION.md lists the knobs.
#pragma once

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace emb {

// Every knob the library reads (tests/test_abi_symbols.py checks this list
// against the knob("EMB_...") calls in csrc/): emb_configure refuses any other
// name, so a misspelt or retired knob is an error instead of a silent no-op.
constexpr const char* kKnobNames[] = {
    "EMB_ARGS_BAR",     "EMB_DEFER_INDEX", "EMB_DEFER_MAX_GAP_US", "EMB_GATHER_STORES", "EMB_HOST_PROFILE",
    "EMB_PREDICT_ROWS", "EMB_RCCL_LIB",    "EMB_SPAN_MOVER",       "EMB_WHERE_BACKLOG",
};
inline bool knob_known(const char* name) {
  for (const char* known : kKnobNames)
    if (std::string(known) == name) return true;
  return false;
}

struct KnobTable {
  std::mutex mu;
  std::map<std::string, std::string> given;      // emb_configure values
  std::map<std::string, bool> read;              // knobs already in effect
};
inline KnobTable& knob_table() {
  static KnobTable* table = new KnobTable();     // never destroyed: read from static initialisers
  return *table;
}

// The value of a knob (nullptr = not set) -- marks it as in effect.
inline const char* knob(const char* name) {
  KnobTable& t = knob_table();
  std::lock_guard<std::mutex> lock(t.mu);
  t.read[name] = true;
  auto it = t.given.find(name);
  if (it != t.given.end()) return it->second.c_str();      // (map nodes never move)
  return std::getenv(name);
}

// 0 = set; 1 = the knob has been read already (too late); value nullptr clears.
inline int knob_set(const char* name, const char* value) {
  KnobTable& t = knob_table();
  std::lock_guard<std::mutex> lock(t.mu);
  if (t.read.count(name)) return 1;
  if (value) t.given[name] = value;
  else t.given.erase(name);
  return 0;
}

}  // namespace emb
