#include "prof.h"

#include <map>
#include <string>
#include <string.h>
#include <vector>

#include "../../include/flair_hip.h"

namespace flair {
namespace {
struct Rec { const char* name; double flops, bytes; };
bool g_on = false;
std::vector<hipEvent_t> g_ev;   // 2 per record
std::vector<Rec> g_rec;
size_t g_cap = 0;
struct Agg { double ms = 0, flops = 0, bytes = 0; long n = 0; };
std::vector<std::pair<std::string, Agg>> g_agg;
}  // namespace

ProfScope::ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : slot(-1), s(stream) {
  if (!g_on || g_rec.size() >= g_cap) return;
  slot = (int)g_rec.size();
  g_rec.push_back({name, flops, bytes});
  (void)hipEventRecord(g_ev[2 * slot], s);
}
ProfScope::~ProfScope() {
  if (slot >= 0) (void)hipEventRecord(g_ev[2 * slot + 1], s);
}
}  // namespace flair

using namespace flair;

extern "C" {
int flair_profile_start(int max_records) {
  if (max_records < 1) return -1;
  while (g_ev.size() < (size_t)2 * max_records) {
    hipEvent_t e;
    // timing events without the system-scope fence: the default writes back and invalidates the caches at every record, which
    // slows the very kernels being timed (hip_runtime_api.h, hipEventDisableSystemFence)
    hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    if (rc != hipSuccess) return (int)rc;
    g_ev.push_back(e);
  }
  g_cap = max_records;
  g_rec.clear();
  g_agg.clear();
  g_on = true;
  return 0;
}

int flair_profile_stop(void) {
  g_on = false;
  hipError_t rc = hipDeviceSynchronize();
  if (rc != hipSuccess) return (int)rc;
  std::map<std::string, size_t> idx;
  for (size_t i = 0; i < g_rec.size(); ++i) {
    float ms = 0.f;
    rc = hipEventElapsedTime(&ms, g_ev[2 * i], g_ev[2 * i + 1]);
    if (rc != hipSuccess) return (int)rc;
    auto it = idx.find(g_rec[i].name);
    if (it == idx.end()) { idx[g_rec[i].name] = g_agg.size(); g_agg.push_back({g_rec[i].name, Agg()}); it = idx.find(g_rec[i].name); }
    Agg& a = g_agg[it->second].second;
    a.ms += ms; a.flops += g_rec[i].flops; a.bytes += g_rec[i].bytes; a.n += 1;
  }
  return (int)g_agg.size();
}

int flair_profile_kernel(int i, char* name, int name_cap, double* total_ms, int64_t* launches, double* flops, double* bytes) {
  if (i < 0 || i >= (int)g_agg.size()) return -1;
  strncpy(name, g_agg[i].first.c_str(), name_cap - 1);
  name[name_cap - 1] = 0;
  *total_ms = g_agg[i].second.ms; *launches = g_agg[i].second.n; *flops = g_agg[i].second.flops; *bytes = g_agg[i].second.bytes;
  return 0;
}
}
