// Optional live timing of kernel launches with HIP events recorded on the launch
// stream (bench.py's roofline leg).  Off by default; not thread-safe; must be off
// while a stream is being captured into a hipGraph.
#pragma once
#include <hip/hip_runtime.h>

namespace selfc {

enum ProfClass { PROF_CONV3X3 = 0, PROF_CONV5_F = 1, PROF_CONV5_GH = 2, PROF_TRANSFORM = 3, PROF_CONV5_PLAIN = 4, PROF_STP = 5, PROF_FUSED_GH = 6, PROF_BWD = 7, PROF_NCLASS = 8 };

bool prof_enabled();
// returns an event already recorded on `s` (start marker) or nullptr when profiling is off
hipEvent_t prof_begin(hipStream_t s);
void prof_end(int cls, hipEvent_t start, hipStream_t s);

struct ProfScope {
  int cls; hipStream_t s; hipEvent_t e0;
  ProfScope(int c, hipStream_t st) : cls(c), s(st), e0(c >= 0 ? prof_begin(st) : nullptr) {}   // c < 0: timed by an enclosing scope
  ~ProfScope() { if (e0) prof_end(cls, e0, s); }
};

}  // namespace selfc
