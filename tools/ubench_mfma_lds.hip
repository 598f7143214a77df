// Micro-benchmark: what one CU sustains on the fused kernels' inner-loop SHAPE - MFMA 32x32x16 f16 fed from LDS through a
// 3-deep register ring, 4 or 8 waves per workgroup (1 or 2 per SIMD), with / without the per-chunk workgroup barrier and the
// one-LDS-store-per-step hook.  Answers, on the device and on random operands (MI355X_MICROARCH.md "DVFS give-back"), what a
// tile design can reach before it is written: cycles per MFMA per SIMD and the clock the chip holds.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_lds.hip -o tools/ubench_mfma_lds && tools/ubench_mfma_lds
//
// Per step a wave reads NA weight fragments (shared by all waves, like the A operands of the convs) and NB activation
// fragments (its own pixels) and issues NA x NB MFMAs.  Product shapes: fused F non-ring wave = <8 waves, NA 2, NB 1>
// (3 reads / 2 MFMAs), ring wave ~ <NA 2, NB 2> minus one; the "two M-tiles per wave on 4 waves" design = <4, 2, 2>.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int LDS_BYTES = 128 * 1024;
constexpr int NFRAG_A = 48;                 // weight fragments cycled through (48 KiB)

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

// MODE bits: 1 = LDS reads in the loop, 2 = barrier every 6 steps, 4 = one ds_write_b128 per step (the hook)
template <int WAVES, int NA, int NB, int MODE>
__global__ __launch_bounds__(WAVES * 64) void loop_kernel(const u32x4* __restrict__ src, float* __restrict__ out, int outer,
                                                          unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < LDS_BYTES / 16; i += WAVES * 64) *reinterpret_cast<u32x4*>(smem + i * 16) = src[i];
  __syncthreads();
  const unsigned char* wa = smem + lane * 16;                                     // weight fragments: 1 KiB each, lane-linear
  const unsigned char* wb = smem + NFRAG_A * 1024 + wave * (8 * 1024) + lane * 16;   // this wave's activation fragments (8 KiB window)
  unsigned char* wdst = smem + NFRAG_A * 1024 + 8 * 8 * 1024 + tid * 16;         // hook store target (never read)
  f32x16 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f16x8 rA[3][NA], rB[3][NB];
  constexpr int STEPS = 48;                                 // unrolled steps per outer iteration (multiple of 6 and 3)
  auto load_step = [&](const int st) __attribute__((always_inline)) {
    const int s = st % 3;
#pragma unroll
    for (int i = 0; i < NA; ++i) rA[s][i] = *reinterpret_cast<const f16x8*>(wa + ((st * NA + i) % NFRAG_A) * 1024);
#pragma unroll
    for (int j = 0; j < NB; ++j) rB[s][j] = *reinterpret_cast<const f16x8*>(wb + ((st * NB + j) % 8) * 1024);
  };
  load_step(0);
  load_step(1);
  load_step(2);
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  const u32x4 hookv = {1u, 2u, 3u, 4u};
  for (int it = 0; it < outer; ++it) {
    static_for<0, STEPS>([&](auto si) __attribute__((always_inline)) {
      constexpr int st = decltype(si)::value;
      if constexpr ((MODE & 1) != 0) load_step(st + 2);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int s = (MODE & 1) ? st % 3 : 0;
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rA[s][i], rB[s][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((MODE & 4) != 0) *reinterpret_cast<u32x4*>(wdst + (st % 2) * 8192) = hookv;
      if constexpr ((MODE & 2) != 0 && st % 6 == 2) __syncthreads();
    });
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
  out[(size_t)blockIdx.x * WAVES * 64 + tid] = sum;
  if (lane == 0) {
    stamps[((size_t)blockIdx.x * WAVES + wave) * 2 + 0] = t1 - t0;
    stamps[((size_t)blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0;
  }
}

// The same loop on v_mfma_f32_16x16x32_f16 (MI355X_MICROARCH.md "DVFS give-back" item 7: the clock the chip holds can depend on the
// MFMA shape).  One step = NA weight fragments (16 rows x 32 k) x NB activation fragments (32 k x 16 pixels) = NA x NB MFMAs of
// 16,384 FLOP; a 32-channel x 32-pixel wave tile of TWO convs is <NA 4, NB 2> (6 reads / 8 MFMAs = the bytes per FLOP of the
// 32x32x16 loop <NA 2, NB 1>), of one conv <NA 2, NB 2>.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int WAVES, int NA, int NB, int MODE>
__global__ __launch_bounds__(WAVES * 64) void loop16_kernel(const u32x4* __restrict__ src, float* __restrict__ out, int outer,
                                                            unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < LDS_BYTES / 16; i += WAVES * 64) *reinterpret_cast<u32x4*>(smem + i * 16) = src[i];
  __syncthreads();
  const unsigned char* wa = smem + lane * 16;
  const unsigned char* wb = smem + NFRAG_A * 1024 + wave * (8 * 1024) + lane * 16;
  unsigned char* wdst = smem + NFRAG_A * 1024 + 8 * 8 * 1024 + tid * 16;
  f32x4v acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  f16x8 rA[3][NA], rB[3][NB];
  constexpr int STEPS = 48;
  auto load_step = [&](const int st) __attribute__((always_inline)) {
    const int s = st % 3;
#pragma unroll
    for (int i = 0; i < NA; ++i) rA[s][i] = *reinterpret_cast<const f16x8*>(wa + ((st * NA + i) % NFRAG_A) * 1024);
#pragma unroll
    for (int j = 0; j < NB; ++j) rB[s][j] = *reinterpret_cast<const f16x8*>(wb + ((st * NB + j) % 8) * 1024);
  };
  load_step(0);
  load_step(1);
  load_step(2);
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  const u32x4 hookv = {1u, 2u, 3u, 4u};
  for (int it = 0; it < outer; ++it) {
    static_for<0, STEPS>([&](auto si) __attribute__((always_inline)) {
      constexpr int st = decltype(si)::value;
      if constexpr ((MODE & 1) != 0) load_step(st + 2);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int s = (MODE & 1) ? st % 3 : 0;
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(rA[s][i], rB[s][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr ((MODE & 4) != 0) *reinterpret_cast<u32x4*>(wdst + (st % 2) * 8192) = hookv;
      if constexpr ((MODE & 2) != 0 && st % 6 == 2) __syncthreads();
    });
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += acc[i][j][e];
  out[(size_t)blockIdx.x * WAVES * 64 + tid] = sum;
  if (lane == 0) {
    stamps[((size_t)blockIdx.x * WAVES + wave) * 2 + 0] = t1 - t0;
    stamps[((size_t)blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int WAVES, int NA, int NB, int MODE, int SHAPE = 0>
void run(const u32x4* src, float* out, unsigned long long* stamps, const char* label) {
  constexpr int STEPS = 48;
  const int grid = 256, outer = 400;
  auto kern = SHAPE ? loop16_kernel<WAVES, NA, NB, MODE> : loop_kernel<WAVES, NA, NB, MODE>;
  constexpr double FLOP = SHAPE ? 16384.0 : 32768.0, CYC = SHAPE ? 0.5 : 1.0;      // a 16x16x32 MFMA is half a 32x32x16
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), LDS_BYTES, 0, src, out, outer, stamps);
  CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), LDS_BYTES, 0, src, out, outer, stamps);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)grid * WAVES * 2);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (size_t i = 0; i < h.size(); i += 2) { cyc.push_back((double)h[i]); clk.push_back((double)h[i] / (double)h[i + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double med_cyc = cyc[cyc.size() / 2], med_clk = clk[clk.size() / 2];
  const double mfma_per_simd = (double)outer * STEPS * NA * NB * (WAVES / 4) * CYC;     // in 32x32x16 equivalents
  const double tf = (double)grid * WAVES * outer * STEPS * NA * NB * FLOP * reps / (ms * 1e-3) / 1e12;
  printf("%-58s waves %d NA %d NB %d reads/MFMA %.2f | %6.1f cyc/MFMA/SIMD  clock %.2f GHz  %7.1f TFLOP/s (chip, 256 WGs)  %.3f ms/launch\n",
         label, WAVES, NA, NB, (MODE & 1) ? (double)(NA + NB) / (NA * NB) : 0.0, med_cyc / mfma_per_simd, med_clk, tf, ms / reps);
  fflush(stdout);
}


// ---- a whole TILE of a single-conv, resident-weight, frame-walking kernel (the "persistent layer-wise" design): S k-steps of
// NB MFMAs per wave (one weight fragment, NB activation fragments), no barrier inside the K loop; behind step i < NST one
// halo piece of the next tile (global load into registers early, ds_write_b128 later); epilogue = LeakyReLU + pack + two
// 16-byte global stores per lane and M-tile; two workgroup barriers per tile (image complete / image free).
template <int WAVES, int NB, int S, int NST>
__global__ __launch_bounds__(WAVES * 64) void tile_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ out, int tiles,
                                                          unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < LDS_BYTES / 16; i += WAVES * 64) *reinterpret_cast<u32x4*>(smem + i * 16) = src[i];
  __syncthreads();
  const unsigned char* wa = smem + lane * 16;
  const unsigned char* wb = smem + NFRAG_A * 1024 + wave * (8 * 1024) + lane * 16;
  unsigned char* wdst = smem + NFRAG_A * 1024 + 8 * 8 * 1024 + tid * 16;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
  u32x4 hv[NST];
  for (int tl = 0; tl < tiles; ++tl) {
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][e] = 0.01f * e;
    f16x8 rA[3], rB[3][NB];
    auto load_step = [&](const int st) __attribute__((always_inline)) {
      const int s_ = st % 3;
      rA[s_] = *reinterpret_cast<const f16x8*>(wa + (st % NFRAG_A) * 1024);
#pragma unroll
      for (int j = 0; j < NB; ++j) rB[s_][j] = *reinterpret_cast<const f16x8*>(wb + ((st * NB + j) % 8) * 1024);
    };
    load_step(0);
    load_step(1);
    const u32x4* hsrc = src + ((size_t)(blockIdx.x * 131 + tl * 17) % 64) * 512 + tid;
    static_for<0, S>([&](auto si) __attribute__((always_inline)) {
      constexpr int st = decltype(si)::value;
      if constexpr (st + 2 < S) load_step(st + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rA[st % 3], rB[st % 3][j], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (st < NST) hv[st] = hsrc[st * 512];                       // next tile's halo: one load per step
      if constexpr (st >= S - NST) *reinterpret_cast<u32x4*>(wdst + ((st - (S - NST)) % 2) * 8192) = hv[st - (S - NST)];   // ... stored late
    });
    // epilogue: LeakyReLU, f16 pack, two 16-byte stores per lane and M-tile
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      unsigned r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a0 = acc[j][2 * e], a1 = acc[j][2 * e + 1];
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 hh = {(_Float16)fmaxf(a0, 0.2f * a0), (_Float16)fmaxf(a1, 0.2f * a1)};
        r[e] = __builtin_bit_cast(unsigned, hh);
      }
      u32x4* o = out + ((size_t)blockIdx.x * WAVES * 64 + tid) * 2 * NB + 2 * j;
      o[0] = u32x4{r[0], r[1], r[2], r[3]};
      o[1] = u32x4{r[4], r[5], r[6], r[7]};
    }
    __syncthreads();
    __syncthreads();
  }
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  if (lane == 0) {
    stamps[((size_t)blockIdx.x * WAVES + wave) * 2 + 0] = t1 - t0;
    stamps[((size_t)blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0;
  }
}

template <int WAVES, int NB, int S, int NST>
void run_tile(const u32x4* src, u32x4* out, unsigned long long* stamps, const char* label) {
  const int grid = 256, tiles = 200;
  auto kern = tile_kernel<WAVES, NB, S, NST>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), LDS_BYTES, 0, src, out, tiles, stamps);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), LDS_BYTES, 0, src, out, tiles, stamps);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)grid * WAVES * 2);
  CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> cyc, clk;
  for (size_t i = 0; i < h.size(); i += 2) { cyc.push_back((double)h[i]); clk.push_back((double)h[i] / (double)h[i + 1] * 0.1); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double per_tile = cyc[cyc.size() / 2] / tiles, floor_ = (double)S * NB * (WAVES / 4) * 32.0;
  const double tf = (double)grid * WAVES * tiles * S * NB * 32768.0 * reps / (ms * 1e-3) / 1e12;
  printf("%-44s waves %d tiles/wave %d steps %3d halo pieces %2d | %7.0f cyc/tile  MFMA floor %6.0f (%.0f %%)  clock %.2f GHz  %7.1f TFLOP/s\n",
         label, WAVES, NB, S, NST, per_tile, floor_, 100.0 * floor_ / per_tile, clk[clk.size() / 2], tf);
  fflush(stdout);
}

int main() {
  u32x4* src; float* out; unsigned long long* stamps;
  CK(hipMalloc(&src, 4 << 20)); CK(hipMemset(src, 0x3c, 4 << 20)); CK(hipMalloc(&out, 32 << 20)); CK(hipMalloc(&stamps, 256 * 8 * 2 * 8));
  std::vector<f16> h(LDS_BYTES / 2);
  srand(1);
  for (auto& v : h) v = (f16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  CK(hipMemcpy(src, h.data(), LDS_BYTES, hipMemcpyHostToDevice));
  // MODE: 1 reads, 2 barrier / 6 steps, 4 hook store
  run<8, 2, 1, 0>(src, out, stamps, "8 waves, bare MFMA (no LDS)");
  run<8, 2, 1, 1>(src, out, stamps, "8 waves, F non-ring shape, reads");
  run<8, 2, 1, 3>(src, out, stamps, "8 waves, F non-ring shape, reads + barrier");
  run<8, 2, 1, 7>(src, out, stamps, "8 waves, F non-ring shape, reads + barrier + store");
  run<8, 2, 2, 1>(src, out, stamps, "8 waves, 2 M-tiles, reads");
  run<8, 2, 2, 7>(src, out, stamps, "8 waves, 2 M-tiles, reads + barrier + store");
  run<8, 1, 1, 1>(src, out, stamps, "8 waves, 1 conv 1 tile (FM steps / G-H conv4), reads");
  run<8, 1, 2, 1>(src, out, stamps, "8 waves, 1 conv 2 tiles (G-H), reads");
  // 16x16x32 twins (cycles per 32x32x16 EQUIVALENT, i.e. per 32,768 FLOP)
  run<8, 4, 2, 0, 1>(src, out, stamps, "16x16x32: 8 waves, bare MFMA (no LDS)");
  run<8, 4, 2, 1, 1>(src, out, stamps, "16x16x32: 8 waves, F non-ring shape (2 convs), reads");
  run<8, 4, 2, 3, 1>(src, out, stamps, "16x16x32: 8 waves, F non-ring shape, reads + barrier");
  run<8, 4, 2, 7, 1>(src, out, stamps, "16x16x32: 8 waves, F non-ring shape, reads + barrier + store");
  run<8, 2, 2, 1, 1>(src, out, stamps, "16x16x32: 8 waves, 1 conv 1 tile (G-H conv4), reads");
  run<8, 2, 4, 1, 1>(src, out, stamps, "16x16x32: 8 waves, 1 conv 2 tiles (G-H), reads");
  run<8, 2, 1, 1>(src, out, stamps, "32x32x16 again: 8 waves, F non-ring shape, reads");
  run<8, 1, 1, 1>(src, out, stamps, "32x32x16 again: 8 waves, 1 conv 1 tile, reads");
  run<4, 2, 1, 0>(src, out, stamps, "4 waves, bare MFMA (no LDS)");
  run<4, 2, 2, 1>(src, out, stamps, "4 waves, 2 M-tiles, reads");
  run<4, 2, 2, 7>(src, out, stamps, "4 waves, 2 M-tiles, reads + barrier + store");
  run<4, 2, 3, 1>(src, out, stamps, "4 waves, 3 M-tiles, reads");
  run<4, 2, 3, 7>(src, out, stamps, "4 waves, 3 M-tiles, reads + barrier + store");
  run<4, 2, 4, 1>(src, out, stamps, "4 waves, 4 M-tiles, reads");
  run<4, 1, 2, 1>(src, out, stamps, "4 waves, 1 conv 2 tiles, reads");
  run<4, 1, 4, 1>(src, out, stamps, "4 waves, 1 conv 4 tiles, reads");
  // whole tiles of the single-conv design: S = 9 taps x (Cin / 16) k-steps; halo pieces = 16-byte pieces of an 18x18 image per thread
  u32x4* out4 = (u32x4*)out;
  run_tile<8, 1, 27, 4>(src, out4, stamps, "conv1 of F (48 ch), 8 waves x 1 tile");
  run_tile<8, 1, 45, 6>(src, out4, stamps, "conv2 of F (80 ch), 8 waves x 1 tile");
  run_tile<8, 1, 63, 9>(src, out4, stamps, "conv3 of F (112 ch), 8 waves x 1 tile");
  run_tile<8, 1, 81, 12>(src, out4, stamps, "conv4 of F (144 ch), 8 waves x 1 tile");
  run_tile<4, 2, 63, 18>(src, out4, stamps, "conv3 of F (112 ch), 4 waves x 2 tiles");
  run_tile<4, 2, 81, 24>(src, out4, stamps, "conv4 of F (144 ch), 4 waves x 2 tiles");
  return 0;
}
