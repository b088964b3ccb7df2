#include "tune.h"

#include <stdlib.h>

#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>

#include "../../include/flair_hip.h"

namespace flair {
namespace {
std::map<std::string, int>& table() {
  static std::map<std::string, int> t;
  return t;
}

// ctypes releases the GIL: two replicas driven from two Python threads, or a flair_tune_set beside a running step, reach
// this table concurrently.  Readers share the lock; a first use / an override takes it exclusively.
std::shared_mutex& table_lock() {
  static std::shared_mutex m;
  return m;
}
}  // namespace

int tune(const char* key, int dflt) {
  auto& t = table();
  {
    std::shared_lock<std::shared_mutex> rd(table_lock());
    auto it = t.find(key);
    if (it != t.end()) return it->second;
  }
  const char* e = getenv(key);
  const int v = e ? atoi(e) : dflt;
  std::unique_lock<std::shared_mutex> wr(table_lock());
  return t.emplace(key, v).first->second;   // (a racing first use computed the same value)
}

void tune_set(const char* key, int value) {
  std::unique_lock<std::shared_mutex> wr(table_lock());
  table()[key] = value;
}
}  // namespace flair

namespace flair { int set_debug_buffer(void* p); }

extern "C" int flair_debug_buffer(void* p) { return flair::set_debug_buffer(p); }

extern "C" int flair_tune_set(const char* key, int value) {
  if (!key) return -1;
  flair::tune_set(key, value);
  return 0;
}
