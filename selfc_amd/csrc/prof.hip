#include "prof.hpp"
#include <algorithm>
#include <vector>
#include "../../include/selfc_hip.h"
#include "common.hpp"

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

// ---- box calibration --------------------------------------------------------------------------------------------------
// MI355X boxes of one pool differ by up to 10 % on the same binary (sustained clock under load).  bench.py prints these two
// figures next to its value so that a reader can tell a slow box from a slow build: a register-only MFMA loop (no memory at
// all: the matrix pipes at whatever clock the chip holds under full MFMA load) and a large device-to-device copy.
namespace {
__global__ __launch_bounds__(256) void calib_mfma_kernel(float* __restrict__ out, const int iters) {
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (f16)(float)((threadIdx.x + i) & 3);
    b[i] = (f16)(float)((threadIdx.x * 3 + i) & 1);
  }
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = mfma_32x32x16(a, b, acc[j]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) sum += acc[j][e];
  if (sum == -1.f) out[0] = sum;     // never true (all products are >= 0): keeps the loop alive
}
}  // namespace

namespace {
// one wave that watches the shader-clock counter against the constant 100 MHz counter for `ticks` of the latter, next to
// whatever else the chip is running (MI355X_MICROARCH.md "DVFS give-back"); exits as soon as the time is up
__global__ __launch_bounds__(64) void clock_sample_kernel(unsigned long long* __restrict__ out, const unsigned long long ticks) {
  unsigned long long t0, r0, t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  do {
    __builtin_amdgcn_s_sleep(64);
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  } while (r1 - r0 < ticks);
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
}
}  // namespace

int selfc_profile_clock_sample(unsigned long long* out2, int micros, void* stream) {
  if (!out2 || micros <= 0 || micros > 500000) return SELFC_EINVAL;
  clock_sample_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out2, (unsigned long long)micros * 100ull);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? SELFC_OK : -(int)e - 1000;
}

// A HIP stream of the caller's own (non-blocking), outside torch's 32-entry pool: runtime.own_stream wraps it as a
// torch.cuda.ExternalStream.  Graph captures of this package fork onto / capture on such streams only - see runtime.own_stream.
int selfc_stream_create(void** out) {
  if (!out) return SELFC_EINVAL;
  hipStream_t s = nullptr;
  const hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  *out = (void*)s;
  return e == hipSuccess ? SELFC_OK : -(int)e - 1000;
}
int selfc_stream_destroy(void* stream) {
  const hipError_t e = hipStreamDestroy((hipStream_t)stream);
  return e == hipSuccess ? SELFC_OK : -(int)e - 1000;
}

// Node census of a captured hipGraph (abi 13): counts[0] = all nodes, [1] = kernel, [2] = memset, [3] = memcpy (any kind), [4] = every other
// type.  `graph` is a hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()).  Host-only; the tests hold "no memset node in
// any graph of the package" with it (a memset node is not ordered with its neighbours in a one-stream capture on this runtime),
// bench.py reports the captured training step's node count.
int selfc_graph_stats(void* graph, long long* counts) {
  if (!graph || !counts) return SELFC_EINVAL;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes((hipGraph_t)graph, nullptr, &n);
  if (e != hipSuccess) return -(int)e - 1000;
  for (int i = 0; i < 5; ++i) counts[i] = 0;
  counts[0] = (long long)n;
  if (!n) return SELFC_OK;
  hipGraphNode_t* nodes = new hipGraphNode_t[n];
  e = hipGraphGetNodes((hipGraph_t)graph, nodes, &n);
  for (size_t i = 0; e == hipSuccess && i < n; ++i) {
    hipGraphNodeType ty;
    e = hipGraphNodeGetType(nodes[i], &ty);
    if (e != hipSuccess) break;
    if (ty == hipGraphNodeTypeKernel) ++counts[1];
    else if (ty == hipGraphNodeTypeMemset) ++counts[2];
    else if (ty == hipGraphNodeTypeMemcpy || ty == hipGraphNodeTypeMemcpyFromSymbol || ty == hipGraphNodeTypeMemcpyToSymbol) ++counts[3];
    else ++counts[4];
  }
  delete[] nodes;
  return e == hipSuccess ? SELFC_OK : -(int)e - 1000;
}

int selfc_profile_calibrate(double* mfma_tflops, double* copy_GBps, void* stream) {
  if (!mfma_tflops || !copy_GBps) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -(int)hipErrorLaunchFailure - 1000;
  constexpr size_t BYTES = (size_t)512 << 20;
  char* buf = nullptr;
  if (hipMalloc(&buf, 2 * BYTES + 64) != hipSuccess) return -(int)hipErrorLaunchFailure - 1000;
  (void)hipMemsetAsync(buf, 0, 2 * BYTES + 64, s);
  constexpr int GRID = 2048, ITERS = 2048;
  const double flop = (double)GRID * 4 * ITERS * 4 * 32768.0;
  double best_m = 0.0, best_c = 0.0;
  int rc = SELFC_OK;
  for (int rep = 0; rep < 4 && rc == SELFC_OK; ++rep) {      // rep 0 warms the clocks up and is not counted
    float ms = 0.f;
    (void)hipEventRecord(e0, s);
    calib_mfma_kernel<<<GRID, 256, 0, s>>>(reinterpret_cast<float*>(buf + 2 * BYTES), ITERS);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = -(int)hipErrorLaunchFailure - 1000; break; }
    if (rep && ms > 0.f) best_m = std::max(best_m, flop / (ms * 1e-3) * 1e-12);
    (void)hipEventRecord(e0, s);
    (void)hipMemcpyAsync(buf + BYTES, buf, BYTES, hipMemcpyDeviceToDevice, s);
    (void)hipEventRecord(e1, s);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { rc = -(int)hipErrorLaunchFailure - 1000; break; }
    if (rep && ms > 0.f) best_c = std::max(best_c, 2.0 * BYTES / (ms * 1e-3) * 1e-9);
  }
  (void)hipFree(buf);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *mfma_tflops = best_m;
  *copy_GBps = best_c;
  return rc;
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
