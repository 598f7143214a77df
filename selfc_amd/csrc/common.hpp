// Shared device helpers for the gfx950 kernels of libselfc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace selfc {

// MFMA operand type.  `f16` is the name used throughout for "the 16-bit operand type": IEEE half by default
// (the parity-grade build); -DSELFC_OPERAND_BF16 builds the same kernels with bfloat16
// operands (libselfc_hip_bf16.so) - same MFMA rate, ~7x the rounding error.
#ifdef SELFC_OPERAND_BF16
typedef __bf16 f16;
#define SELFC_OPERAND_NAME "bf16"
#else
typedef _Float16 f16;
#define SELFC_OPERAND_NAME "f16"
#endif
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// native 16-byte vector: unlike HIP's uint4 struct, a load of it is a first-class value (no
// aggregate memcpy into a private array, which hipcc fails to promote out of scratch)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// LDS pixel stride of a staged activation tile: 32 f16 channels (64 B) + 16 B
// pad = 5 sixteen-byte slots; 5 is coprime with the 16 slots of a 256-B bank
// row, so 16 consecutive pixels read by one ds_read_b128 lane group hit 16
// different slots.
constexpr int PS = 80;

// LeakyReLU(0.2) = max(v, 0.2 v): two VALU ops (0.2f*v rounds exactly as in v >= 0 ? v : 0.2f*v).  The max is emitted
// directly: fmaxf() first canonicalises its operand (a second v_max_f32 v, v, v per element - a third of the epilogue's
// arithmetic); accumulators of finite f16 products are never signalling NaNs.
__device__ __forceinline__ float lrelu02(float v) {
  const float m = 0.2f * v;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
  return r;
}

// MI355X deals consecutive workgroup ids round-robin over its 8 XCDs (private
// L2 each).  Remap so that each XCD receives a CONTIGUOUS range of logical tile
// ids: neighbouring tiles (which share halo pixels and the same weights) then
// hit the same L2.  Bijective for any grid size.  Speed only, never correctness.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ f32x16 mfma_32x32x16(const f16x8 a, const f16x8 b, const f32x16 c) {
#ifdef SELFC_PROXY16
  // TIMING-ONLY diagnostic build (results are wrong): the same operand registers and FLOP count issued as two 16x16x32 MFMAs -
  // what the loops would cost (and what clock the chip would hold) on that instruction shape, before any kernel is rewritten
  f32x4 lo = {c[0], c[1], c[2], c[3]}, hi = {c[4], c[5], c[6], c[7]};
  lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, lo, 0, 0, 0);
  hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, hi, 0, 0, 0);
  f32x16 r = c;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
#endif
#ifdef SELFC_OPERAND_BF16
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 mfma_16x16x32(const f16x8 a, const f16x8 b, const f32x4 c) {
#ifdef SELFC_OPERAND_BF16
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE attribute: opt a kernel in to > 64 KiB of LDS once per
// (call site, device).  `done` is the call site's static bit mask (bit d = device d); atomic, so host threads driving
// different GPUs race benignly (the attribute call is idempotent).
inline hipError_t lds_optin(const void* fn, int bytes, std::atomic<unsigned long long>& done) {
  int d = 0;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (d & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// Raw BUFFER loads (SGPR descriptor + one 32-bit per-lane byte offset + an SGPR byte offset): the weight-fragment streams are read
// at `uniform base + chunk offset + 16 * tid`.  As flat / global loads hipcc materialises one 64-bit per-lane address PER CHUNK and
// keeps them all live across the tile loop (20-48 VGPRs in the 256-register fused kernels: fused_f_kernel<1> spilled); as buffer
// loads the chunk offset is an SGPR add and the lane offset is ONE VGPR.  Out-of-range lanes read zeros (num_records), which also
// replaces the min() clamps on the last partial piece.
typedef __amdgpu_buffer_rsrc_t buf_rsrc;
__device__ __forceinline__ buf_rsrc make_rsrc(const void* base, const unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);     // gfx9 word 3: 32-bit data format, no swizzle
}
__device__ __forceinline__ u32x4 buffer_load_b128(const buf_rsrc r, const unsigned lane_off, const unsigned uniform_off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, uniform_off, 0);
}

// Store with the same addressing; a lane whose offset is >= num_records (BUF_OOB) stores nothing - masked lanes need no branch.
constexpr unsigned BUF_OOB = 0x80000000u;       // resources built by make_rsrc are kept below 2 GiB where this is used
__device__ __forceinline__ void buffer_store_b128(const u32x4 v, const buf_rsrc r, const unsigned lane_off, const unsigned uniform_off) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, lane_off, uniform_off, 0);
}

__device__ __forceinline__ uint32_t pack2(float a, float b) {
  f16x2 h = {(f16)a, (f16)b};
  return __builtin_bit_cast(uint32_t, h);
}

// LeakyReLU(0.2) of two fp32 accumulators, then ONE rounding to the f16 pair.  A packed variant (round first, then v_pk_mul_f16 by
// 0.2 and v_pk_max_f16: 3 instead of 5 VALU instructions per pair) was measured in round 2: fused G/H -1.4 %, but the slope becomes
// fl16(0.2) = 0.19995 and negatives are rounded twice, which moved the chained-gradient check (tests/test_gpu_backward.py
// test_stack_backward_chain) from 4.4 % to 5.1 % against its 5 % bar - dropped.
__device__ __forceinline__ uint32_t lrelu_pack2(float a, float b) { return pack2(lrelu02(a), lrelu02(b)); }


// ---- diagnostic build only (make diag DIAG=-DSELFC_CLOCKS): the clock a kernel really runs at inside the benchmark's
// hipGraph.  Thread 0 of workgroup 0 reads the shader-clock counter and the constant 100 MHz counter at kernel entry and exit
// (MI355X_MICROARCH.md "DVFS give-back" item 6) and adds the differences to its slot; tools/clock_probe.py reads the sums.
// Nothing of this exists in the shipped library.
#ifdef SELFC_CLOCKS
struct ClockProbe { unsigned long long t0, r0; };
__device__ __forceinline__ void clock_probe_begin(ClockProbe& p) {
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(p.t0), "=s"(p.r0) :: "memory");
}
__device__ __forceinline__ void clock_probe_end(const ClockProbe& p, unsigned long long* slot, const bool leader) {
  unsigned long long t1, r1;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
  if (leader && slot) {
    atomicAdd(slot + 0, t1 - p.t0);
    atomicAdd(slot + 1, r1 - p.r0);
    atomicAdd(slot + 2, 1ull);
  }
}
unsigned long long* clock_probe_slot(int which);      // prof.hip: device buffer of 3 counters per slot (0 F pair 0, 1 F pair 1, 2 G/H)
#endif

}  // namespace selfc
