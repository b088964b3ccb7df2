#include "tune.h"

#include <stdlib.h>

#include <map>
#include <string>

#include "../../include/flair_hip.h"

namespace flair {
namespace {
std::map<std::string, int>& table() {
  static std::map<std::string, int> t;
  return t;
}
}  // namespace

int tune(const char* key, int dflt) {
  auto& t = table();
  auto it = t.find(key);
  if (it != t.end()) return it->second;
  const char* e = getenv(key);
  const int v = e ? atoi(e) : dflt;
  t[key] = v;
  return v;
}

void tune_set(const char* key, int value) { table()[key] = value; }
}  // namespace flair

namespace flair { int set_debug_buffer(void* p); }

extern "C" int flair_debug_buffer(void* p) { return flair::set_debug_buffer(p); }

extern "C" int flair_tune_set(const char* key, int value) {
  if (!key) return -1;
  flair::tune_set(key, value);
  return 0;
}
