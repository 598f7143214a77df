// Per-launch floor of a dependent chain of tiny kernels: stream launches (C++ host loop) against ONE hipGraph of the same launches.
// Kernel = 504 x 192 threads, 64 dependent FMAs, optional 40 KB of dynamic LDS (what a data-gradient conv workgroup allocates).
//   hipcc -O3 --offload-arch=gfx950 launch_floor_ubench.hip -o launch_floor_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ __launch_bounds__(192) void k(float* out, float a, float b) {
  extern __shared__ float sm[];
  float x = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 64; ++i) x = __builtin_fmaf(x, a, b);
  if (x == 123.456f) { sm[threadIdx.x] = x; out[0] = sm[0]; }
}
static double run_stream(float* d, hipStream_t s, int n, size_t lds) {
  for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k, dim3(504), dim3(192), lds, s, d, 1.0001f, 0.5f);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(504), dim3(192), lds, s, d, 1.0001f, 0.5f);
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("  stream, %5zu B LDS: %.2f us per launch (host issue alone %.2f us)\n", lds, std::chrono::duration<double, std::micro>(t2 - t0).count() / n,
         std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
  return 0;
}
static void run_graph(float* d, hipStream_t s, int n, size_t lds) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(504), dim3(192), lds, s, d, 1.0001f, 0.5f);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  const int R = 20;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < R; ++i) hipGraphLaunch(ge, s);
  auto t1 = std::chrono::steady_clock::now();
  hipStreamSynchronize(s);
  auto t2 = std::chrono::steady_clock::now();
  printf("  graph of %d nodes, %5zu B LDS: %.2f us per node (host issue alone %.2f us per node)\n", n, lds,
         std::chrono::duration<double, std::micro>(t2 - t0).count() / (R * n), std::chrono::duration<double, std::micro>(t1 - t0).count() / (R * n));
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  hipStream_t s; hipStreamCreate(&s);
  for (size_t lds : {size_t(0), size_t(40064)}) {
    run_stream(d, s, 2000, lds);
    run_graph(d, s, 400, lds);
  }
  return 0;
}
