// Optional in-library kernel timing: HIP events recorded on the launch stream around each kernel launch,
// aggregated per kernel name with the ALGORITHMIC flops / bytes of the launch (bench.py "roofline").
// Off by default; when off a scope costs one predictable branch.
#pragma once
#include <hip/hip_runtime.h>

namespace flair {
struct ProfScope {
  int slot;
  hipStream_t s;
  ProfScope(const char* name, double flops, double bytes, hipStream_t stream);
  ~ProfScope();
};
}  // namespace flair
