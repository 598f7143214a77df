// What does straight-line code cost a short kernel?  Each workgroup of a data-gradient conv launch runs through ~17 KB of code once;
// the instruction cache is cold at kernel start.  This probe runs N dependent FMAs per lane either unrolled (N x 8 bytes of code,
// every line fetched once) or as a rolled loop (one cache line), on the grid of such a launch (504 x 192), back to back on one
// stream, and reports the average launch-to-launch time.   hipcc -O3 --offload-arch=gfx950 ifetch_ubench.hip -o ifetch_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N, bool UNROLL>
__global__ __launch_bounds__(192) void k(float* out, float a, float b) {
  float x = threadIdx.x;
  if (UNROLL) {
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, a, b);
  } else {
#pragma unroll 1
    for (int i = 0; i < N; ++i) { x = __builtin_fmaf(x, a, b); asm volatile("" : "+v"(x)); }
  }
  if (x == 123.456f) out[0] = x;
}
template <int N, bool U>
void run(float* d, const char* name) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<N, U>), dim3(504), dim3(192), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e0, 0);
  const int R = 200;
  for (int i = 0; i < R; ++i) hipLaunchKernelGGL((k<N, U>), dim3(504), dim3(192), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s N=%5d  code ~%6d B  %.2f us per launch\n", name, N, U ? N * 8 : 64, ms * 1e3 / R);
}
int main() {
  float* d; hipMalloc(&d, 4096);
  run<64, true>(d, "unrolled"); run<64, false>(d, "rolled");
  run<256, true>(d, "unrolled"); run<256, false>(d, "rolled");
  run<1024, true>(d, "unrolled"); run<1024, false>(d, "rolled");
  run<2048, true>(d, "unrolled"); run<2048, false>(d, "rolled");
  run<4096, true>(d, "unrolled"); run<4096, false>(d, "rolled");
  return 0;
}
