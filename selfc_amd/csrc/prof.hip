#include "prof.hpp"
#include <vector>
#include "../../include/selfc_hip.h"

namespace selfc {
namespace {
struct Rec { int cls; hipEvent_t e0, e1; };
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t get_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

bool prof_enabled() { return g_on; }

hipEvent_t prof_begin(hipStream_t s) {
  if (!g_on) return nullptr;
  hipEvent_t e = get_event();
  if (e) (void)hipEventRecord(e, s);
  return e;
}

void prof_end(int cls, hipEvent_t start, hipStream_t s) {
  hipEvent_t e1 = get_event();
  if (!e1) { g_pool.push_back(start); return; }
  (void)hipEventRecord(e1, s);
  g_recs.push_back(Rec{cls, start, e1});
}
}  // namespace selfc

using namespace selfc;

extern "C" {

int selfc_profile_enable(int on) {
  g_on = on != 0;
  return SELFC_OK;
}

int selfc_profile_read(int cls, double* total_ms, long long* launches) {
  if (cls < 0 || cls >= PROF_NCLASS || !total_ms || !launches) return SELFC_EINVAL;
  double ms = 0.0;
  long long n = 0;
  for (const Rec& r : g_recs) {
    if (r.cls != cls) continue;
    if (hipEventSynchronize(r.e1) != hipSuccess) return SELFC_EINVAL;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) return SELFC_EINVAL;
    ms += t;
    ++n;
  }
  *total_ms = ms;
  *launches = n;
  return SELFC_OK;
}

int selfc_profile_reset(void) {
  for (const Rec& r : g_recs) { g_pool.push_back(r.e0); g_pool.push_back(r.e1); }
  g_recs.clear();
  return SELFC_OK;
}

}  // extern "C"

#ifdef SELFC_CLOCKS
#include "common.hpp"
namespace selfc {
static unsigned long long* g_clock_buf = nullptr;
unsigned long long* clock_probe_slot(int which) {
  if (!g_clock_buf) {
    if (hipMalloc(&g_clock_buf, 8 * 3 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    (void)hipMemset(g_clock_buf, 0, 8 * 3 * sizeof(unsigned long long));
  }
  return g_clock_buf + 3 * which;
}
}  // namespace selfc
extern "C" int selfc_debug_clocks(unsigned long long* out24, int reset) {      // diag library only: [slot][cycles, 100 MHz ticks, launches]
  if (!selfc::g_clock_buf) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  if (hipMemcpy(out24, selfc::g_clock_buf, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -3;
  if (reset) (void)hipMemset(selfc::g_clock_buf, 0, 24 * sizeof(unsigned long long));
  return 0;
}
#endif
