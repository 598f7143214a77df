// Does a COLD instruction cache set the per-node floor of a graph of small kernels?  16 distinct kernels (same body, different
// template tag: 16 x N x 8 bytes of code) launched round-robin in one hipGraph, against the same number of nodes of ONE kernel.
//   hipcc -O3 --offload-arch=gfx950 launch_floor_ubench2.hip -o launch_floor_ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
template <int TAG, int N>
__global__ __launch_bounds__(192) void k(float* out, float a, float b) {
  float x = threadIdx.x + TAG;
#pragma unroll
  for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, a, b + (float)(i * 16 + TAG));      // distinct literal constants: no code sharing
  if (x == 123.456f) out[0] = x;
}
template <int N, int T>
static void launch(int i, int distinct, float* d, hipStream_t s) {
  if constexpr (T < 16) {
    if ((distinct ? i % 16 : 0) == T) { hipLaunchKernelGGL((k<T, N>), dim3(504), dim3(192), 0, s, d, 1.0001f, 0.5f); return; }
    launch<N, T + 1>(i, distinct, d, s);
  }
}
template <int N>
static void run_graph(float* d, hipStream_t s, int n, int distinct) {
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) launch<N, 0>(i, distinct, d, s);
  (void)hipStreamEndCapture(s, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, s);
  (void)hipStreamSynchronize(s);
  const int R = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) (void)hipGraphLaunch(ge, s);
  (void)hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("  N=%5d (%6d B of code per kernel), %2d distinct kernels: %.2f us per node\n", N, N * 8, distinct ? 16 : 1,
         std::chrono::duration<double, std::micro>(t2 - t0).count() / (R * n));
  (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
}
int main() {
  float* d; (void)hipMalloc(&d, 4096);
  hipStream_t s; (void)hipStreamCreate(&s);
  run_graph<64>(d, s, 400, 0);   run_graph<64>(d, s, 400, 1);
  run_graph<512>(d, s, 400, 0);  run_graph<512>(d, s, 400, 1);
  run_graph<1024>(d, s, 400, 0); run_graph<1024>(d, s, 400, 1);
  run_graph<2048>(d, s, 400, 0); run_graph<2048>(d, s, 400, 1);
  return 0;
}
