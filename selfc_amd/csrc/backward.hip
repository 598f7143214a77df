// Backward (training) kernels of the dense-block subnets and the affine coupling for gfx950 (MI355X).
//
// The reference trains through stock autograd (SURVEY 8b "Autograd"); this file supplies the gradients of
//   DenseBlock / D2DTInput   (Subnet_constructor.py:8-34, 98-133)   -> selfc_subnet_bwd
//   InvBlockExp coupling     (Inv_arch.py:21-33)                    -> selfc_coupling_bwd
//   FrequencyAnalyzer        (SelfC_GMM_arch_inv.py:62-82)          -> selfc_freq_fwd_bwd / selfc_freq_inv_bwd
//
// Dense block backward.  With f_k = lrelu(conv_k[x, f_1..f_{k-1}]) and out = conv5[x, f_1..f_4], the gradient
// of the pre-activations obeys a dense recursion in the opposite order:
//   dpre_j = lrelu'(f_j) * ( conv5^T(dOut)[f_j] + sum_{k>j} conv_k^T(dpre_k)[f_j] )
//   dx     =               conv5^T(dOut)[x]   + sum_k     conv_k^T(dpre_k)[x]
// so it runs on the same generic plane-list MFMA conv as the forward (csrc/dense_conv.hip, EPI_BWD) with
// transposed + tap-flipped weights (packing.py:pack_subnet_bwd) on a plane-blocked f16 gradient buffer
// [dpre4 dpre3 dpre2 dpre1 | dOut | conv5^T(dOut) x-part, f1, f2, f3].  Gradients pass through the MFMA as f16
// scaled by a power of two taken from max|dOut| (bwd_internal.hpp: grad_scale); every fp32 result is unscaled.
//
// Weight gradients: dW_k[o][c][tap] = sum_px dpre_k[px][o] * in[px + tap][c] is a GEMM whose reduction index is
// the pixel, while both operands are stored channel-minor.  wgrad_kernel stages a 16x16-pixel tile of each in
// LDS as [pixel][32 ch] rows and reads both MFMA operands with ds_read_b64_tr_b16 (the gfx950 transposing LDS
// read): a k-step is a 4x4 pixel patch, each 16-lane group fetches 4 consecutive pixels x 16 channels = 256
// contiguous bytes per 32-lane half, i.e. conflict-free.  Each wave keeps one 32x32 accumulator per tap and
// writes a partial; wgrad_finish_kernel sums the partials into the PyTorch weight layout.
#include <type_traits>
#include "common.hpp"
#include "prof.hpp"
#include "bwd_internal.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace {

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }
inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

// ---------------------------------------------------------------------------------------------------------
// scale selection and fp32 NHWC -> scaled f16 planes
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ g, size_t n, unsigned* __restrict__ amax_bits) {
  float m = 0.f;
  bool nan = false;                          // fmaxf drops NaNs: track them separately so that a diverged step stays visible
  const size_t n4 = n >> 2;
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(g);
  // four independent 16-byte loads per thread and trip (one at a time, a 14 MB gradient took 13 dependent memory round trips
  // per thread: 15 us for what is 3 us of HBM time - 61 such launches sit on the training step's critical path)
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = i + u * stride < n4 ? g4[i + u * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
      nan |= (v[u].x != v[u].x) | (v[u].y != v[u].y) | (v[u].z != v[u].z) | (v[u].w != v[u].w);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = g[(n4 << 2) + threadIdx.x];
    m = fmaxf(m, fabsf(v));
    nan |= v != v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  // non-negative floats order as uints, and a quiet NaN's bits (0x7fc00000) order above every finite value and infinity:
  // a NaN anywhere makes *amax NaN, grad_scale(NaN) is NaN and every gradient of the call comes out NaN, as the
  // reference's autograd would deliver it
  if (__any(nan)) { if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, 0x7fc00000u); }
  else if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits, __float_as_uint(m));
}

// item = (pixel, 8-channel chunk of the padded planes): 16-byte stores
// amax_copy (optional): the scale's source is handed on to the phase's own slot by this kernel (one lane) - the kernels that follow
// read the slot, this one reads the source, so no separate 4-byte copy node sits in front of it
__global__ __launch_bounds__(256) void grad_to_planes_kernel(const float* __restrict__ g, f16* __restrict__ planes, size_t npix,
                                                             int c, int cs, int nplanes, int lrelu, float sign, const float* __restrict__ amax,
                                                             float* __restrict__ amax_copy) {
  const float sc = amax ? sign * grad_scale(*amax) : sign;
  if (amax_copy && blockIdx.x == 0 && threadIdx.x == 0) *amax_copy = *amax;
  const size_t total = npix * (size_t)nplanes * 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int chunk = (int)(i % ((size_t)nplanes * 4));
  const size_t pix = i / ((size_t)nplanes * 4);
  const int ch0 = chunk * 8;
  f16x8 o;
  float in[8];
  if (ch0 + 8 <= c && (cs & 3) == 0) {           // whole chunk inside the row, rows 16-byte aligned: two 16-byte loads instead of eight scalar ones
    const float4 a = *reinterpret_cast<const float4*>(g + pix * cs + ch0), b = *reinterpret_cast<const float4*>(g + pix * cs + ch0 + 4);
    in[0] = a.x; in[1] = a.y; in[2] = a.z; in[3] = a.w; in[4] = b.x; in[5] = b.y; in[6] = b.z; in[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) in[e] = (ch0 + e < c) ? g[pix * cs + ch0 + e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = in[e];
    if (lrelu) v = lrelu02(v);
    o[e] = (f16)(v * sc);
  }
  *reinterpret_cast<f16x8*>(planes + (size_t)(ch0 >> 5) * npix * 32 + pix * 32 + (ch0 & 31)) = o;
}

// The same for the two dOut tensors of a G/H pair (blockIdx.y) under ONE scale: S from max(max|dOut_G|, max|dOut_H|), written to
// *amax_common for every kernel that follows.  A power-of-two scale is exact, and with the common maximum at (128, 256] the smaller
// tensor keeps f16's full mantissa while its own maximum is within 2^-22 of the larger one's - so the pair's input gradient can be ONE
// accumulation over both nets' planes (and one launch).
__global__ __launch_bounds__(256) void grad_to_planes2_kernel(const float* __restrict__ g0, const float* __restrict__ g1, f16* __restrict__ planes0,
                                                              f16* __restrict__ planes1, size_t npix, int c, int cs, int nplanes, float sign0, float sign1,
                                                              const float* __restrict__ amax0, const float* __restrict__ amax1, float* __restrict__ amax_common) {
  const float m0 = *amax0, m1 = *amax1;
  const float am = (m0 != m0) ? m0 : (m1 != m1) ? m1 : fmaxf(m0, m1);       // a NaN in either poisons the call (absmax_kernel's convention)
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *amax_common = am;
  const bool second = blockIdx.y != 0;
  const float* __restrict__ g = second ? g1 : g0;
  f16* __restrict__ planes = second ? planes1 : planes0;
  const float sc = (second ? sign1 : sign0) * grad_scale(am);
  const size_t total = npix * (size_t)nplanes * 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int chunk = (int)(i % ((size_t)nplanes * 4));
  const size_t pix = i / ((size_t)nplanes * 4);
  const int ch0 = chunk * 8;
  f16x8 o;
  float in[8];
  if (ch0 + 8 <= c && (cs & 3) == 0) {
    const float4 a = *reinterpret_cast<const float4*>(g + pix * cs + ch0), b = *reinterpret_cast<const float4*>(g + pix * cs + ch0 + 4);
    in[0] = a.x; in[1] = a.y; in[2] = a.z; in[3] = a.w; in[4] = b.x; in[5] = b.y; in[6] = b.z; in[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) in[e] = (ch0 + e < c) ? g[pix * cs + ch0 + e] : 0.f;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (f16)(in[e] * sc);
  *reinterpret_cast<f16x8*>(planes + (size_t)(ch0 >> 5) * npix * 32 + pix * 32 + (ch0 & 31)) = o;
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------------------------------------
struct WgArgs {
  const f16* P;          // single: gradient planes (blockIdx.z selects one); multi: the dpre planes (conv k reads plane 4-k)
  const f16* Q0;         // activation planes: a run of nq0 planes, then the run Q1
  const f16* Q1;
  float* part;           // [split][pair][taps][32 o][32 c]; pair = z*qtot + q (single) | blockIdx.y (multi)
  float* bpart;          // optional [split][gridDim.z | 4][32]: column sums of P (bias gradient), written by the q == 0 workgroups
  size_t plane;          // halfs per plane
  int N, T, H, W, tiles_x, tiles_y, ntiles;
  int nq0;
  int multi;             // conv1..4 of one dense block in a single launch: blockIdx.y enumerates (conv k, input plane q)
  int nqc1;              // multi: input planes of conv1 (conv k has nqc1 + k - 1)
  int ts;                // tile side in pixels: 16, or 12 where that wastes less of the frame (wg_tile_side; 9-tap and temporal kernels)
};

// (conv k, plane q) of pair index y in multi mode
__device__ __forceinline__ void wg_pair(int y, const int nqc1, int& k, int& q) {
  k = 1;
  while (y >= nqc1 + k - 1) { y -= nqc1 + k - 1; ++k; }
  q = y;
}

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

__device__ __forceinline__ f16x8 tr_frag(const unsigned char* lds, const int off0, const int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(f16x8, v);
}

// TAPS = 9: 3x3 spatial taps, 3 waves x 3 taps each.  TAPS = 3: temporal taps (frames n-1, n, n+1 of the clip, zero
// outside), 3 waves x 1 tap.  TAPS = 1: pointwise, 4 waves x 4 of the tile's 16 patches, reduced through LDS.
// Every workgroup writes ONE partial per tap, so the finish pass reads nsplit blocks.  The next tile's operands are
// fetched into registers while the current one is multiplied (the launches are short: exposed load latency per tile
// was most of their time).
// gx / gy: the launch's extent in x (pixel splits) and y (pairs) for THIS job (gridDim.x / gridDim.y of a one-job launch)
// TS: tile side (16, or 12: a 36x36 training latent is 3 x 3 tiles of 12 with NO empty 4x4 patches - 9 patches per tile instead of
// 16 of which 7 are skipped or half empty on average; the kernel is instruction-bound, so fewer patches per tile is the saving);
// lp / lq: the workgroup's LDS tiles (declared by the kernel: two instantiations in one kernel share them)
template <int TAPS, int TS = 16>
__device__ __forceinline__ void wgrad_body(const WgArgs& a, const int gx, const int gy, const int bx, const int by,
                                           unsigned char* const lp, unsigned char* const lq) {
  static_assert(TS == 16 || (TS == 12 && TAPS == 9), "tile side");
  constexpr int NW = TAPS == 1 ? 4 : 3, NT = NW * 64;
  constexpr int TPW = TAPS == 9 ? 3 : 1;                    // taps per wave
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int QW = TS + 2 * HALO, QPIX = QW * QW, QF = TAPS == 3 ? 3 : 1;
  constexpr int PPIX4 = TS * TS * 4, NP = TS / 4;
  constexpr int PI = (PPIX4 + NT - 1) / NT, QI = (QF * QPIX * 4 + NT - 1) / NT;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int kconv = 0, qi = by;
  if (a.multi) wg_pair(by, a.nqc1, kconv, qi);
  const f16* __restrict__ P = a.P + (size_t)(a.multi ? 4 - kconv : (int)blockIdx.z) * a.plane;
  const f16* __restrict__ Q = qi < a.nq0 ? a.Q0 + (size_t)qi * a.plane : a.Q1 + (size_t)(qi - a.nq0) * a.plane;
  const int H = a.H, W = a.W;
  const bool want_bias = a.bpart != nullptr && qi == 0;     // workgroup-uniform

  f32x16 acc[TPW], accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    accb[r] = 0.f;
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t][r] = 0.f;
  }
  f16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (f16)1.f;

  // ds_read_b64_tr_b16 addressing: lane 16g + 4q + p supplies block row q (a pixel), columns 4p..4p+3 of the
  // group's 16 channels, and receives channel (lane & 15) of the 4 pixels.  Groups g = 0,1 are the two channel
  // halves of the same pixels, g >> 1 is the MFMA k-half: pixels of patch rows 2h (first read) and 2h+1 (second).
  const int g = lane >> 4, h = g >> 1, q = (lane >> 2) & 3, p = lane & 3;
  const int choff = (16 * (g & 1) + 4 * p) * 2;

  u32x4 preg[PI], qreg[QI];
  unsigned okp = 0, okq = 0;
  auto fetch = [&](const int tile) __attribute__((always_inline)) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, n = tile / (a.tiles_x * a.tiles_y);
    const int tx0 = tx * TS, ty0 = ty * TS;
    const int tclip = n % a.T;
    okp = 0; okq = 0;
#pragma unroll
    for (int it = 0; it < PI; ++it) {
      const int i = min(tid + it * NT, PPIX4 - 1);
      const int px = i >> 2, ch = i & 3;
      const int y = ty0 + px / TS, x = tx0 + px % TS;
      okp |= (((y < H) & (x < W)) ? 1u : 0u) << it;
      const int yc = min(y, H - 1), xc = min(x, W - 1);
      preg[it] = *reinterpret_cast<const u32x4*>(P + ((size_t)(n * H + yc) * W + xc) * 32 + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < QI; ++it) {
      const int i = min(tid + it * NT, QF * QPIX * 4 - 1);
      const int f = i / (QPIX * 4), j = i - f * (QPIX * 4);
      const int px = j >> 2, ch = j & 3;
      const int hy = px / QW, hx = px - hy * QW;
      const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
      const int dt = TAPS == 3 ? f - 1 : 0;
      const bool tv = (tclip + dt >= 0) & (tclip + dt < a.T);
      okq |= ((tv & (y >= 0) & (y < H) & (x >= 0) & (x < W)) ? 1u : 0u) << it;
      const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
      const int nq = tv ? n + dt : n;
      qreg[it] = *reinterpret_cast<const u32x4*>(Q + ((size_t)(nq * H + yc) * W + xc) * 32 + ch * 8);
    }
  };

  // The tile loop in four straight-line forms (EX: the frame is whole tiles, no patch can lie outside - no skip test; BI: this workgroup
  // also sums the bias gradient): with the tests inside the unrolled patch loop every patch was its own basic block, and the compiler
  // moved the accumulators between VGPRs and AGPRs at every join (352 v_accvgpr_* for 100 MFMAs) in a kernel that is instruction-bound
  auto run = [&](auto ex_c, auto bi_c) __attribute__((always_inline)) {
  constexpr bool EX = decltype(ex_c)::value, BI = decltype(bi_c)::value;
  int tile = bx;
  if (tile < a.ntiles) fetch(tile);
  for (; tile < a.ntiles; tile += gx) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y;
    const int tx0 = tx * TS, ty0 = ty * TS;
    __syncthreads();                       // the previous tile's fragments have been read
#pragma unroll
    for (int it = 0; it < PI; ++it) {
      const int i = tid + it * NT;
      if (i < PPIX4) *reinterpret_cast<u32x4*>(lp + (i >> 2) * 64 + (i & 3) * 16) = ((okp >> it) & 1u) ? preg[it] : u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int it = 0; it < QI; ++it) {
      const int i = tid + it * NT;
      if (i < QF * QPIX * 4) *reinterpret_cast<u32x4*>(lq + (size_t)(i >> 2) * 64 + (i & 3) * 16) = ((okq >> it) & 1u) ? qreg[it] : u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    if (tile + gx < a.ntiles) fetch(tile + gx);
    // 16 patches of 4x4 pixels; patches wholly outside the image are skipped (wave-uniform: the transposing
    // read needs EXEC all ones).  TAPS == 1: each wave takes 4 of them; otherwise each wave takes all 16 for its taps.
    if constexpr (EX) {
      // whole tiles: the patch loop is straight-line, and the fragments of patch k + 1 are requested before the MFMAs of patch k
      // (as hipcc scheduled the plain loop every MFMA sat behind an s_waitcnt lgkmcnt(0) for the four transposing reads just in
      // front of it: profiles/r6/ab_experiments.txt r6z)
      constexpr int NIT = TAPS == 1 ? NP * NP / 4 : NP * NP;
      f16x8 afr[2], bfr[2][TPW];
      auto read_patch = [&](const int k, f16x8& af, f16x8 (&bf)[TPW]) __attribute__((always_inline)) {
        const int pi = TAPS == 1 ? wave + 4 * k : k;
        const int pr = pi / NP, pc = pi % NP;
        const int arow = ((4 * pr + 2 * h) * TS + 4 * pc + q) * 64 + choff;
        af = tr_frag(lp, arow, arow + TS * 64);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
          const int tap = TAPS == 9 ? wave * 3 + t : 0;
          const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap - 3 * (tap / 3) : 0;
          const int fr = TAPS == 3 ? wave : 0;
          const int brow = (fr * QPIX + (4 * pr + 2 * h + dy) * QW + 4 * pc + q + dx) * 64 + choff;
          bf[t] = tr_frag(lq, brow, brow + QW * 64);
        }
      };
      read_patch(0, afr[0], bfr[0]);
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        if (k + 1 < NIT) read_patch(k + 1, afr[(k + 1) & 1], bfr[(k + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = mfma_32x32x16(afr[k & 1], bfr[k & 1][t], acc[t]);
        if (BI) accb = mfma_32x32x16(afr[k & 1], ones, accb);       // (every wave: a wave test here is a branch per patch; wave 0's copy is stored)
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
    for (int pi = (TAPS == 1 ? wave : 0); pi < NP * NP; pi += (TAPS == 1 ? 4 : 1)) {
      const int pr = pi / NP, pc = pi % NP;
      if (ty0 + 4 * pr >= H || tx0 + 4 * pc >= W) continue;
      const int arow = ((4 * pr + 2 * h) * TS + 4 * pc + q) * 64 + choff;
      const f16x8 af = tr_frag(lp, arow, arow + TS * 64);
#pragma unroll
      for (int t = 0; t < TPW; ++t) {
        const int tap = TAPS == 9 ? wave * 3 + t : 0;
        const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap - 3 * (tap / 3) : 0;
        const int fr = TAPS == 3 ? wave : 0;
        const int brow = (fr * QPIX + (4 * pr + 2 * h + dy) * QW + 4 * pc + q + dx) * 64 + choff;
        const f16x8 bf = tr_frag(lq, brow, brow + QW * 64);
        acc[t] = mfma_32x32x16(af, bf, acc[t]);
      }
      if (BI) accb = mfma_32x32x16(af, ones, accb);       // (every wave: a wave test here is a branch per patch; wave 0's copy is stored)
    }
    }
  }
  };
  {
    const bool exact = (H % TS == 0) && (W % TS == 0);      // workgroup-uniform
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (exact) { if (want_bias) run(T_{}, T_{}); else run(T_{}, F_{}); }
    else { if (want_bias) run(F_{}, T_{}); else run(F_{}, F_{}); }
  }
  // D[o][c]: lane owns column c = lane & 31, rows o = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int npairs = a.multi ? gy : (int)(gy * gridDim.z);
  const int pair = a.multi ? by : (int)(blockIdx.z * gy + by);
  const size_t blk = (size_t)bx * npairs + pair;
  if (TAPS == 1) {
    // reduce the 4 waves' accumulators (and bias sums) through LDS, wave 0 writes
    __syncthreads();
    float* red = reinterpret_cast<float*>(lq);            // 3 x 4 KiB
    float* redb = reinterpret_cast<float*>(lp);           // 3 x 4 KiB
    if (wave > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        red[(wave - 1) * 1024 + r * 64 + lane] = acc[0][r];
        if (want_bias) redb[(wave - 1) * 1024 + r * 64 + lane] = accb[r];
      }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          acc[0][r] += red[w * 1024 + r * 64 + lane];
          if (want_bias) accb[r] += redb[w * 1024 + r * 64 + lane];
        }
    }
  }
  if (TAPS != 1 || wave == 0) {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int tap = TAPS == 9 ? wave * 3 + t : (TAPS == 3 ? wave : 0);
      float* __restrict__ base = a.part + ((blk * TAPS + tap) << 10);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int o = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        base[o * 32 + (lane & 31)] = acc[t][r];
      }
    }
    if (want_bias && wave == 0 && (lane & 31) == 0) {
      const int nb = a.multi ? 4 : (int)gridDim.z, pb = a.multi ? kconv - 1 : (int)blockIdx.z;
      float* __restrict__ bb = a.bpart + ((size_t)bx * nb + pb) * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) bb[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = accb[r];
    }
  }
}

// LDS of a weight-gradient workgroup: the gradient tile [16 x 16 px][32 ch] and the activation tile(s) (18 x 18 with the 3x3 taps' halo)
#define SELFC_WG_LDS(TAPS_) \
  __shared__ __attribute__((aligned(16))) unsigned char lp[256 * 64]; \
  __shared__ __attribute__((aligned(16))) unsigned char lq[((TAPS_) == 9 ? 18 * 18 : 256) * 64]
// 9-tap body on the job's tile side
__device__ __forceinline__ void wgrad9(const WgArgs& a, const int gx, const int gy, const int bx, const int by, unsigned char* lp, unsigned char* lq) {
  if (a.ts == 12) wgrad_body<9, 12>(a, gx, gy, bx, by, lp, lq);
  else wgrad_body<9, 16>(a, gx, gy, bx, by, lp, lq);
}

template <int TAPS>
__global__ __launch_bounds__((TAPS == 1 ? 4 : 3) * 64) void wgrad_kernel(const WgArgs a) {
  SELFC_WG_LDS(TAPS);
  if constexpr (TAPS == 9) wgrad9(a, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y, lp, lq);
  else wgrad_body<TAPS>(a, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y, lp, lq);
}

// multi mode only (conv1..4 of a dense block, grid.z unused by the body): the jobs of two nets of one geometry in one launch
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_pair_kernel(const WgArgs a, const WgArgs b) {
  SELFC_WG_LDS(9);
  if (blockIdx.z) wgrad9(b, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y, lp, lq);
  else wgrad9(a, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y, lp, lq);
}

// Up to WG_TABLE conv1..4 jobs (multi mode) in ONE launch: blockIdx.z = job, whose own extent is gx[z] x gy[z] (the rest of the grid
// returns at once).  A training step is a serial chain of launches that each under-fill the chip (one stream and three streams take
// the same time inside a replayed graph on this runtime): the weight gradients of a whole block stack - independent of everything
// behind them - are therefore ONE fat launch at the end of the stack's data-gradient chain instead of 24 thin ones beside it.
constexpr int WG_TABLE = 32;
struct WgTable {
  int n, xcd;
  int gx[WG_TABLE], gy[WG_TABLE];
  WgArgs job[WG_TABLE];
};
static_assert(sizeof(WgTable) <= 4096, "kernel argument limit");
// Workgroup -> (pixel split, pair): the grid is 1-D per job (x; a multiple of 8 workgroups, blockIdx.z = job) and workgroups reach the
// eight XCDs round robin, so L & 7 is the XCD.  All pairs of ONE split - which read the same tiles of the same planes: a dpre plane is an
// operand of 2..5 pairs, an activation plane of 1..4 - sit next to each other on ONE XCD and walk their tiles in step: the re-reads hit
// that XCD's L2 instead of going out to the Infinity Cache / HBM once per pair (the launch moved 5.0 GB per stack at 5.5 TB/s: it was
// bound by exactly that).  t.xcd == 0: plain x = split fastest (A/B).
__device__ __forceinline__ bool wg_map(const int L, const int gx, const int gy, const int xcd_mode, int& bx, int& by) {
  if (xcd_mode && gx >= 8) {               // (fewer than 8 splits: the XCD form would leave XCDs without work)
    const int j = L >> 3;
    by = j % gy;
    bx = (j / gy) * 8 + (L & 7);
  } else {
    bx = L % gx;
    by = L / gx;
  }
  return bx < gx && by < gy;
}
__global__ __launch_bounds__(192) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_table_kernel(const WgTable t) {
  const int z = blockIdx.z;
  int bx, by;
  if (!wg_map((int)blockIdx.x, t.gx[z], t.gy[z], t.xcd, bx, by)) return;          // workgroup-uniform
  SELFC_WG_LDS(9);
  wgrad9(t.job[z], t.gx[z], t.gy[z], bx, by, lp, lq);
}

// Temporal weight gradient (conv5 of D2DTInput: dW[o][c][tap] = sum_px g[n][px][o] * in[n + tap - 1][px][c] inside each clip).
// A workgroup owns (clip, 16x16 spatial tile) units and walks the clip's frames in order with a three-slot ring of
// activation tiles in LDS, so every frame's tile is loaded once (the generic kernel loaded three per frame) while the next
// frame's two tiles are prefetched into registers; wave w multiplies with the ring slot of frame t + w - 1.
// bz / nz: this workgroup's gradient plane and the number of gradient planes of its net (blockIdx.z / gridDim.z of a one-net launch)
template <int TS>
__device__ __forceinline__ void wgrad_temporal_body_ts(const WgArgs& a, const int bz, const int nz, const int gx, const int gy, const int bx, const int by,
                                                       unsigned char* const lp, unsigned char* const lq) {
  constexpr int NT = 192, PPIX4 = TS * TS * 4, PI = (PPIX4 + NT - 1) / NT, NP = TS / 4, SLOT = TS * TS * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qi = by;
  const f16* __restrict__ P = a.P + (size_t)bz * a.plane;
  const f16* __restrict__ Q = qi < a.nq0 ? a.Q0 + (size_t)qi * a.plane : a.Q1 + (size_t)(qi - a.nq0) * a.plane;
  const int H = a.H, W = a.W, T = a.T;
  const bool want_bias = a.bpart != nullptr && qi == 0;
  f32x16 acc, accb;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accb[r] = 0.f; }
  f16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (f16)1.f;
  const int g = lane >> 4, h = g >> 1, q = (lane >> 2) & 3, p = lane & 3;
  const int choff = (16 * (g & 1) + 4 * p) * 2;

  u32x4 preg[PI], qreg[PI];
  unsigned okm = 0;
  // fetch the P tile of frame t and the Q tile of frame t + 1 of the current unit (either may be outside the clip)
  auto fetch = [&](const int nbase, const int t, const int tx0, const int ty0) __attribute__((always_inline)) {
    okm = 0;
    const bool pv = (t >= 0) & (t < T), qv = t + 1 < T;
    const int np = nbase + max(0, min(t, T - 1)), nq = nbase + min(t + 1, T - 1);
#pragma unroll
    for (int it = 0; it < PI; ++it) {
      const int i = min(tid + it * NT, PPIX4 - 1);
      const int px = i >> 2, ch = i & 3;
      const int y = ty0 + px / TS, x = tx0 + px % TS;
      const bool in = (y < H) & (x < W);
      okm |= ((in & pv) ? 1u : 0u) << it;
      okm |= ((in & qv) ? 1u : 0u) << (16 + it);
      const int yc = min(y, H - 1), xc = min(x, W - 1);
      preg[it] = *reinterpret_cast<const u32x4*>(P + ((size_t)(np * H + yc) * W + xc) * 32 + ch * 8);
      qreg[it] = *reinterpret_cast<const u32x4*>(Q + ((size_t)(nq * H + yc) * W + xc) * 32 + ch * 8);
    }
  };
  auto store_q = [&](const int slot, const unsigned mask) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < PI; ++it) {
      const int i = tid + it * NT;
      if (i < PPIX4) *reinterpret_cast<u32x4*>(lq + (size_t)slot * SLOT + (i >> 2) * 64 + (i & 3) * 16) = ((mask >> it) & 1u) ? qreg[it] : u32x4{0u, 0u, 0u, 0u};
    }
  };

  const int nunits = (a.N / T) * a.tiles_x * a.tiles_y;
  // four straight-line forms of the unit loop (EX: whole tiles only, no patch test; BI: bias sums), as in wgrad_body
  auto run = [&](auto ex_c, auto bi_c) __attribute__((always_inline)) {
  constexpr bool EX = decltype(ex_c)::value, BI = decltype(bi_c)::value;
  for (int unit = bx; unit < nunits; unit += gx) {
    const int tx = unit % a.tiles_x, ty = (unit / a.tiles_x) % a.tiles_y, clip = unit / (a.tiles_x * a.tiles_y);
    const int tx0 = tx * TS, ty0 = ty * TS, nbase = clip * T;
    // ring slots: frame f lives in slot (f + 1) % 3; slot of frame -1 is zero, frame 0 is loaded up front
    __syncthreads();
    fetch(nbase, -1 + 0, tx0, ty0);                    // t = -1: P invalid (unused), Q = frame 0
    // (fetch's P part for t = -1 reads frame 0's P rows harmlessly; only the Q half is stored)
    store_q(1, okm >> 16);
    for (int i = tid; i < PPIX4; i += NT) *reinterpret_cast<u32x4*>(lq + (i >> 2) * 64 + (i & 3) * 16) = u32x4{0u, 0u, 0u, 0u};   // slot 0 = frame -1
    fetch(nbase, 0, tx0, ty0);                          // P[0], Q[1]
    for (int t = 0; t < T; ++t) {
      __syncthreads();                                  // frame t-1's fragments have been read
#pragma unroll
      for (int it = 0; it < PI; ++it) {
        const int i = tid + it * NT;
        if (i < PPIX4) *reinterpret_cast<u32x4*>(lp + (i >> 2) * 64 + (i & 3) * 16) = ((okm >> it) & 1u) ? preg[it] : u32x4{0u, 0u, 0u, 0u};
      }
      store_q((t + 2) % 3, okm >> 16);                  // frame t+1 (zeros past the clip end)
      __syncthreads();
      if (t + 1 < T) fetch(nbase, t + 1, tx0, ty0);     // P[t+1], Q[t+2] while frame t is multiplied
      const int slot = (t + wave) % 3;                  // frame t + wave - 1
#pragma unroll
      for (int pi = 0; pi < NP * NP; ++pi) {
        const int pr = pi / NP, pc = pi % NP;
        if (!EX && (ty0 + 4 * pr >= H || tx0 + 4 * pc >= W)) continue;
        const int row = ((4 * pr + 2 * h) * TS + 4 * pc + q) * 64 + choff;
        const f16x8 af = tr_frag(lp, row, row + TS * 64);
        const f16x8 bf = tr_frag(lq + (size_t)slot * SLOT, row, row + TS * 64);
        acc = mfma_32x32x16(af, bf, acc);
        if (BI) accb = mfma_32x32x16(af, ones, accb);      // (every wave; wave 0's copy is stored)
      }
    }
  }
  };
  {
    const bool exact = (H % TS == 0) && (W % TS == 0);      // workgroup-uniform
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (exact) { if (want_bias) run(T_{}, T_{}); else run(T_{}, F_{}); }
    else { if (want_bias) run(F_{}, T_{}); else run(F_{}, F_{}); }
  }
  const int npairs = gy * nz;
  const int pair = bz * gy + by;
  float* __restrict__ base = a.part + ((((size_t)bx * npairs + pair) * 3 + wave) << 10);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int o = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    base[o * 32 + (lane & 31)] = acc[r];
  }
  if (want_bias && wave == 0 && (lane & 31) == 0) {
    float* __restrict__ bb = a.bpart + ((size_t)bx * nz + bz) * 32;
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = accb[r];
  }
}

__device__ __forceinline__ void wgrad_temporal_body(const WgArgs& a, const int bz, const int nz, const int gx, const int gy, const int bx, const int by) {
  __shared__ __attribute__((aligned(16))) unsigned char lp[256 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char lq[3 * 256 * 64];
  if (a.ts == 12) wgrad_temporal_body_ts<12>(a, bz, nz, gx, gy, bx, by, lp, lq);
  else wgrad_temporal_body_ts<16>(a, bz, nz, gx, gy, bx, by, lp, lq);
}

__global__ __launch_bounds__(192) void wgrad_temporal_kernel(const WgArgs a) {
  wgrad_temporal_body(a, (int)blockIdx.z, (int)gridDim.z, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y);
}

// two nets of one geometry: blockIdx.z in [0, 2 Pn)
__global__ __launch_bounds__(192) void wgrad_temporal_pair_kernel(const WgArgs a, const WgArgs b) {
  const int nz = (int)gridDim.z >> 1;
  if ((int)blockIdx.z >= nz) wgrad_temporal_body(b, (int)blockIdx.z - nz, nz, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y);
  else wgrad_temporal_body(a, (int)blockIdx.z, nz, (int)gridDim.x, (int)gridDim.y, (int)blockIdx.x, (int)blockIdx.y);
}

// Up to WG_TABLE temporal (conv5) jobs in one launch (blockIdx.z = job), each with its own extent: gx unit splits x (nz gradient
// planes x gy input planes) pairs
struct WgTTable {
  int n, xcd;
  int gx[WG_TABLE], gy[WG_TABLE], nz[WG_TABLE];
  WgArgs job[WG_TABLE];
};
static_assert(sizeof(WgTTable) <= 4096, "kernel argument limit");
// blockIdx.z = job; x enumerates (unit split, (gradient plane, input plane) pair) through wg_map: the pairs of one split - every
// gradient plane against every input plane of the same tiles - next to each other on one XCD
__global__ __launch_bounds__(192) void wgrad_temporal_table_kernel(const WgTTable t) {
  const int j = blockIdx.z;
  int bx, pr;
  if (!wg_map((int)blockIdx.x, t.gx[j], t.gy[j] * t.nz[j], t.xcd, bx, pr)) return;   // workgroup-uniform
  wgrad_temporal_body(t.job[j], pr / t.gy[j], t.nz[j], t.gx[j], t.gy[j], bx, pr % t.gy[j]);
}

struct FinArgs {
  const float* part;
  const float* bpart;
  float* out[4];         // single: out[0] = (O, Ctot, ttot) fp32, PyTorch layout of the conv weight; multi: conv1..4
  float* bout[4];        // (O) or null
  int nW, Pn, qtot, ttot, O, Ctot, cin, nx;
  int multi, nqc1, npairs;
  const float* amax;
  float beta;
};

// thread = one element of the partial block layout (coalesced reads over the nW partials), scattered write;
// the threads past the weight elements reduce the bias partials
__device__ __forceinline__ void wgrad_finish_body(const FinArgs& a, const size_t e) {
  const size_t per = (size_t)a.npairs * a.ttot * 1024;
  const float inv = 1.f / grad_scale(*a.amax);
  if (e >= per) {
    const size_t o = e - per;
    const int nb = a.multi ? 4 : a.Pn;
    if (o >= (size_t)(a.multi ? 128 : a.O)) return;
    const int kb = (int)(o >> 5);
    float* bo = a.multi ? (kb == 0 ? a.bout[0] : kb == 1 ? a.bout[1] : kb == 2 ? a.bout[2] : a.bout[3]) : a.bout[0];
    if (!bo) return;
    float t = 0.f;
    for (int w = 0; w < a.nW; ++w) t += a.bpart[((size_t)w * nb + kb) * 32 + (o & 31)];
    float* dst = bo + (a.multi ? (o & 31) : o);
    *dst = (a.beta != 0.f) ? a.beta * *dst + t * inv : t * inv;
    return;
  }
  const int c = (int)(e & 31), oo = (int)((e >> 5) & 31);
  const size_t blk = e >> 10;
  const int tap = (int)(blk % a.ttot);
  const int pair = (int)(blk / a.ttot);
  int qq, o, Ctot;
  float* outp;
  if (a.multi) {
    int k = 1, y = pair;
    while (y >= a.nqc1 + k - 1) { y -= a.nqc1 + k - 1; ++k; }
    qq = y; o = oo; Ctot = a.cin + 32 * (k - 1);
    outp = k == 1 ? a.out[0] : k == 2 ? a.out[1] : k == 3 ? a.out[2] : a.out[3];
  } else {
    qq = pair % a.qtot;
    o = 32 * (pair / a.qtot) + oo; Ctot = a.Ctot;
    outp = a.out[0];
  }
  if (!outp) return;
  int ci;
  if (qq < a.nx) {
    ci = 32 * qq + c;
    if (ci >= a.cin) return;
  } else {
    ci = a.cin + 32 * (qq - a.nx) + c;
  }
  if (o >= a.O || ci >= Ctot) return;
  // the partials are nW strided reads per thread: sixteen in flight (the loop is latency-bound, not bandwidth-bound)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int w = 0;
  for (; w + 16 <= a.nW; w += 16) {
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = a.part[(size_t)(w + j) * per + e];
#pragma unroll
    for (int j = 0; j < 16; j += 4) { s0 += v[j]; s1 += v[j + 1]; s2 += v[j + 2]; s3 += v[j + 3]; }
  }
  for (; w + 4 <= a.nW; w += 4) {
    s0 += a.part[(size_t)w * per + e];
    s1 += a.part[(size_t)(w + 1) * per + e];
    s2 += a.part[(size_t)(w + 2) * per + e];
    s3 += a.part[(size_t)(w + 3) * per + e];
  }
  for (; w < a.nW; ++w) s0 += a.part[(size_t)w * per + e];
  float* dst = outp + ((size_t)o * Ctot + ci) * a.ttot + tap;
  const float v = ((s0 + s1) + (s2 + s3)) * inv;
  *dst = (a.beta != 0.f) ? a.beta * *dst + v : v;
}

__global__ __launch_bounds__(256) void wgrad_finish_kernel(const FinArgs a) {
  wgrad_finish_body(a, (size_t)blockIdx.x * 256 + threadIdx.x);
}

// two finish jobs in one launch (conv1..4 and conv5 of a subnet): blocks [0, nblk_a) do job a, the rest job b
__global__ __launch_bounds__(256) void wgrad_finish2_kernel(const FinArgs a, const FinArgs b, const unsigned nblk_a) {
  if (blockIdx.x < nblk_a) wgrad_finish_body(a, (size_t)blockIdx.x * 256 + threadIdx.x);
  else wgrad_finish_body(b, (size_t)(blockIdx.x - nblk_a) * 256 + threadIdx.x);
}

// Up to FIN_TABLE finish jobs in one launch (the deferred finishes of a whole block stack's weight gradients: one launch per
// 24 jobs instead of one per subnet - 54 launches of 11..26 us per training step, 0.74 TB/s).  Blocks [start[j], start[j+1]) do job j.
constexpr int FIN_TABLE = 24;
struct FinTable {
  int n;
  unsigned start[FIN_TABLE + 1];
  FinArgs job[FIN_TABLE];
};
static_assert(sizeof(FinTable) <= 4096, "kernel argument limit");

__global__ __launch_bounds__(256) void wgrad_finish_table_kernel(const FinTable t) {
  int j = 0;
  while (j + 1 < t.n && blockIdx.x >= t.start[j + 1]) ++j;               // workgroup-uniform
  wgrad_finish_body(t.job[j], (size_t)(blockIdx.x - t.start[j]) * 256 + threadIdx.x);
}

inline unsigned fin_blocks(const FinArgs& f) {
  const size_t per = (size_t)f.npairs * f.ttot * 1024;
  const bool bias = f.multi ? (f.bout[0] || f.bout[1] || f.bout[2] || f.bout[3]) : f.bout[0] != nullptr;
  return (unsigned)((per + (bias ? (f.multi ? 128 : f.O) : 0) + 255) / 256);
}

// ---------------------------------------------------------------------------------------------------------
// coupling and FrequencyAnalyzer gradients (elementwise / index shuffles, fp32)
// ---------------------------------------------------------------------------------------------------------
// s = clamp*(2 sigmoid(h) - 1)  =>  ds/dh = clamp*(1 - (s/clamp)^2)/2.
// The affine coupling itself as a stand-alone elementwise pass (Inv_arch.py:26-27,29-30): s = clamp (2 sigmoid(h) - 1),
// rev == 0: y2 = x2 e^s + g;  rev != 0: y2 = (x2 - g) / e^s.  Used where the fused conv5 epilogues do not apply: an InvBlockExp
// with channel_split_num > 3, composed from stand-alone subnets.  Any layout (all operands share it).
__global__ __launch_bounds__(256) void coupling_fwd_kernel(int rev, const float4* __restrict__ x2, const float4* __restrict__ g,
                                                           const float4* __restrict__ h, float4* __restrict__ y2, float4* __restrict__ s,
                                                           float clamp, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const float4 xv = x2[i], gv = g[i], hv = h[i];
  const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, ha[4] = {hv.x, hv.y, hv.z, hv.w};
  float oy[4], os[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    os[j] = clamp * (2.f / (1.f + expf(-ha[j])) - 1.f);
    oy[j] = rev ? (xa[j] - ga[j]) / expf(os[j]) : xa[j] * expf(os[j]) + ga[j];
  }
  y2[i] = make_float4(oy[0], oy[1], oy[2], oy[3]);
  s[i] = make_float4(os[0], os[1], os[2], os[3]);
}

// rev == 0 (y2 = x2*e^s + g, v = x2):   dx2 = dy2*e^s,   dh = dy2*x2*e^s * ds/dh,   dg = dy2
// rev != 0 (y2 = (x2-g)*e^-s, v = y2):  dx2 = dy2*e^-s,  dh = -dy2*y2 * ds/dh,      dg = -dx2
// Block-wide max |value| -> ONE atomic per block on the float bits (NaN -> 0x7fc00000: absmax_kernel's convention).  Atomics on
// one address serialise at the memory side (~90 per microsecond): the kernels below are grid-stride over <= 256 blocks, so a
// maximum costs <= 256 of them, not one per wave of the whole tensor (13,600 for a 14 MB gradient: +3 ms per training step,
// measured).  m: this thread's running max, nan: it saw a NaN; every thread of the block must call.
__device__ __forceinline__ void block_absmax(float m, bool nan, unsigned* bits) {
  __shared__ float red[8];
  __shared__ int rednan;
  if (threadIdx.x == 0) rednan = 0;
#pragma unroll
  for (int k = 32; k > 0; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  if (__any(nan) && (threadIdx.x & 63) == 0) atomicOr(&rednan, 1);
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = red[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) r = fmaxf(r, red[k]);
    if (rednan) atomicMax(bits, 0x7fc00000u);
    else if (r > 0.f) atomicMax(bits, __float_as_uint(r));
  }
}
__device__ __forceinline__ void track4(const float (&o)[4], float& m, bool& nan) {
  m = fmaxf(fmaxf(m, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
  nan |= (o[0] != o[0]) | (o[1] != o[1]) | (o[2] != o[2]) | (o[3] != o[3]);
}

__global__ __launch_bounds__(256) void coupling_bwd_kernel(int rev, const float4* __restrict__ v, const float4* __restrict__ s,
                                                           const float4* __restrict__ dy2, float4* __restrict__ dx2,
                                                           float4* __restrict__ dh, float clamp, size_t n4,
                                                           unsigned* __restrict__ dx2_amax, unsigned* __restrict__ dh_amax) {
  float mx = 0.f, mh = 0.f;
  bool nx = false, nh = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 vv = v[i], ss = s[i], dd = dy2[i];
    const float va[4] = {vv.x, vv.y, vv.z, vv.w}, sa[4] = {ss.x, ss.y, ss.z, ss.w}, da[4] = {dd.x, dd.y, dd.z, dd.w};
    float ox[4], oh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = sa[j] / clamp;
      const float dsdh = clamp * (1.f - u * u) * 0.5f;
      const float e = expf(rev ? -sa[j] : sa[j]);
      ox[j] = da[j] * e;
      oh[j] = (rev ? -da[j] * va[j] : ox[j] * va[j]) * dsdh;
    }
    dx2[i] = make_float4(ox[0], ox[1], ox[2], ox[3]);
    dh[i] = make_float4(oh[0], oh[1], oh[2], oh[3]);
    track4(ox, mx, nx);
    track4(oh, mh, nh);
  }
  // max |dx2| / max |dh| where they are produced: the max|dOut| of G's (reverse) / H's backward without a separate pass
  if (dx2_amax) block_absmax(mx, nx, dx2_amax);
  if (dh_amax) block_absmax(mh, nh, dh_amax);
}

// a += b with max |a| of the result (the two halves of y1's gradient meet here; the sum is F's dOut)
__global__ __launch_bounds__(256) void add_absmax_kernel(float4* __restrict__ a, const float4* __restrict__ b, size_t n4, unsigned* __restrict__ amax) {
  float m = 0.f;
  bool nan = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = a[i], y = b[i];
    const float o[4] = {x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w};
    a[i] = make_float4(o[0], o[1], o[2], o[3]);
    track4(o, m, nan);
  }
  if (amax) block_absmax(m, nan, amax);
}

// FrequencyAnalyzer forward (lo = 4x4 mean, hi[(sy*4+sx)*3+c] = x - lo): adjoint on the latent layout
//   dx[c][4Y+sy][4X+sx] = dhi[(sy*4+sx)*3+c] + (dlo[c] - sum_{sy',sx'} dhi[(sy'*4+sx')*3+c]) / 16
__global__ __launch_bounds__(256) void freq_fwd_bwd_kernel(const float* __restrict__ d1, const float* __restrict__ d2, float* __restrict__ dx,
                                                           int N, int H, int W, int c2p) {
  const int h = H / 4, w = W / 4;
  const size_t total = (size_t)N * h * w * 3;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % 3);
  const size_t pix = i / 3;
  const int X = (int)(pix % w), Y = (int)((pix / w) % h), n = (int)(pix / ((size_t)w * h));
  const float* hi = d2 + pix * c2p;
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) sum += hi[k * 3 + c];
  const float base = (d1[pix * 4 + c] - sum) * (1.f / 16.f);
#pragma unroll
  for (int k = 0; k < 16; ++k)
    dx[(((size_t)n * 3 + c) * H + 4 * Y + (k >> 2)) * W + 4 * X + (k & 3)] = hi[k * 3 + c] + base;
}

// FrequencyAnalyzer reverse (out = nearest_up4(x[:, :3]) + PixelShuffle4(x[:, 3:]), channel c*16+sy*4+sx): adjoint
__global__ __launch_bounds__(256) void freq_inv_bwd_kernel(const float* __restrict__ dout, float* __restrict__ d1, float* __restrict__ d2,
                                                           int N, int H, int W, int c2p) {
  const int h = H / 4, w = W / 4;
  const size_t total = (size_t)N * h * w * 3;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % 3);
  const size_t pix = i / 3;
  const int X = (int)(pix % w), Y = (int)((pix / w) % h), n = (int)(pix / ((size_t)w * h));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float v = dout[(((size_t)n * 3 + c) * H + 4 * Y + (k >> 2)) * W + 4 * X + (k & 3)];
    d2[pix * c2p + c * 16 + k] = v;
    sum += v;
  }
  d1[pix * 4 + c] = sum;
  if (c == 0) d1[pix * 4 + 3] = 0.f;
}

// scratch layout of one selfc_subnet_bwd call
struct BwdLayout {
  int nx, ng, hasx;
  size_t plane_b;       // bytes of one f16 plane
  size_t off_g, off_t5, off_xplane, off_amax, off_wg, off_wg5, total;
};

BwdLayout bwd_layout(int N, int H, int W, int cin, int cout) {
  BwdLayout L{};
  L.nx = (cin + 31) / 32;
  L.ng = (cout + 31) / 32;
  L.hasx = cin <= 3;
  const size_t npix = (size_t)N * H * W;
  L.plane_b = npix * 64;
  L.off_g = 4 * L.plane_b;
  L.off_t5 = L.off_g + (size_t)L.ng * L.plane_b;
  L.off_xplane = L.off_t5 + (size_t)(L.nx + 3) * L.plane_b;
  L.off_amax = L.off_xplane + (L.hasx ? L.plane_b : 0);
  L.off_wg = up256(L.off_amax + 256);
  // largest weight-gradient job: conv4 (1 P plane, nx+3 Q planes, 9 taps) or conv5 (ng P planes, nx+4 Q planes, 9 | 3 taps)
  const size_t a4 = bwd_wgrad14_scratch_bytes(N, H, W, L.nx);
  const size_t a5 = bwd_wgrad_scratch_bytes(N, H, W, L.ng, L.nx + 4, 9);
  L.off_wg5 = up256(L.off_wg + a4);                   // own region each: both compute kernels run before ONE finish launch
  L.total = up256(L.off_wg5 + a5);
  return L;
}

// pixel splits of a weight-gradient job: enough workgroups to fill the chip (~2 per CU), bounded by the tile count and
// by 12 MiB of partials (written once, read once by the finish pass)
// SELFC_WG_TILES: when the launches of many subnets are deferred into ONE (wgrad_table_kernel) there is no chip to fill per job:
// every workgroup then takes at least this many tiles (fewer partials to write and to reduce)
// Default: 16 while the job has at most 256 tiles (training crops at 1..4 septuplets per rank: more, shorter workgroups), 32 beyond
// (8 per rank: fewer partials) - same-box sweep on the final kernels, profiles/r6/ab_experiments.txt r6aa.
static long wg_min_tiles(long units) {
  static const long v = getenv("SELFC_WG_TILES") ? atol(getenv("SELFC_WG_TILES")) : 0;
  return v >= 1 ? v : (units <= 256 ? 16 : 32);
}
// SELFC_BWD_WG_THIN (phase flag of the calling entry point): ONE workgroup per pair walks every tile - a launch of a few hundred
// long-lived workgroups (a stack: ~290) that leaves most of every CU to the kernels of another stream (the next stack's data chain)
static thread_local bool g_wg_thin = false;
int wgrad_nsplit_table(int nsplit, long units) {
  if (g_wg_thin) return 1;
  const long cap = (units + wg_min_tiles(units) - 1) / wg_min_tiles(units);
  return (int)(nsplit > cap ? (cap < 1 ? 1 : cap) : nsplit);
}
struct ThinScope {
  bool old;
  explicit ThinScope(bool on) : old(g_wg_thin) { g_wg_thin = on; }
  ~ThinScope() { g_wg_thin = old; }
};

// tile side of the 9-tap and temporal weight-gradient kernels: 12 where 12x12 tiles cover the frame with less waste than 16x16 ones
// (36x36 training latents: 3 x 3 tiles of 12, no empty patch; 16: 3 x 3 tiles 56 % full).  SELFC_WG_TS=16: always 16 (A/B)
int wg_tile_side(int H, int W) {
  static const int force = getenv("SELFC_WG_TS") ? atoi(getenv("SELFC_WG_TS")) : 0;
  if (force == 16 || force == 12) return force;
  const long a16 = (long)((H + 15) / 16) * ((W + 15) / 16) * 256, a12 = (long)((H + 11) / 12) * ((W + 11) / 12) * 144;
  return a12 < a16 ? 12 : 16;
}
inline int wg_tiles(int v, int ts) { return (v + ts - 1) / ts; }

int wgrad_nsplit(int N, int H, int W, int npairs, int ttot) {
  const int ts9 = ttot == 1 ? 16 : wg_tile_side(H, W);
  const long ntiles = (long)N * wg_tiles(H, ts9) * wg_tiles(W, ts9);
  long ns = (768 + (long)npairs - 1) / (long)npairs;           // 512 / 1024 / 1536 / 2048 measured: all slower (profiles/r4/ab_experiments.txt)
  const long cap = 3072 / ((long)npairs * ttot);
  if (ns > cap) ns = cap;
  if (ns > ntiles) ns = ntiles;
  return (int)(ns < 1 ? 1 : ns);
}

}  // namespace

namespace selfc {

// *amax = 0 as a KERNEL, not hipMemsetAsync (round 5).  Inside a captured step a memset NODE in front of the atomic-max kernel was not
// ordered with the kernels around it when the capture ran on ONE stream: the maximum kept its previous value (or was zeroed late), the
// power-of-two gradient scale came out wrong and the replayed step's gradient norm differed from the eager step's by 1.5 % at 8 septuplets
// (tools/experiments/graph_vs_eager_bits.py: with this kernel the replayed and the eager steps are bit-identical on one stream and on
// three; the three-stream capture happened to be right with the memset too).  The same mechanism left torch's multi-block reductions
// (a semaphore zeroed by a memset) without output.  No memset node is left in any graph of the package.
__global__ void zero1_kernel(float* p) { *p = 0.f; }

int bwd_absmax(const float* g, size_t n, float* amax, hipStream_t s) {
  hipLaunchKernelGGL(zero1_kernel, dim3(1), dim3(1), 0, s, amax);
  int rc = hip_rc(hipGetLastError());
  if (rc) return rc;
  const size_t nb = (n + 256 * 32 - 1) / (256 * 32);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(nb < 1 ? 1 : (nb > 256 ? 256 : nb))), dim3(256), 0, s, g, n, (unsigned*)amax);
  return hip_rc(hipGetLastError());
}

int bwd_to_planes(const float* x, void* planes, size_t npix, int c, int cs, int nplanes, int lrelu, float sign,
                  const float* amax, hipStream_t s, float* amax_copy) {
  const size_t items = npix * (size_t)nplanes * 4;
  hipLaunchKernelGGL(grad_to_planes_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s,
                     x, (f16*)planes, npix, c, cs, nplanes, lrelu, sign, amax, amax_copy);
  return hip_rc(hipGetLastError());
}

size_t bwd_wgrad_scratch_bytes(int N, int H, int W, int Pn, int qtot, int ttot) {
  const int ns = wgrad_nsplit(N, H, W, Pn * qtot, ttot);
  return up256((size_t)ns * (Pn < 4 ? 4 : Pn) * 32 * sizeof(float)) + (size_t)ns * Pn * qtot * ttot * 4096;
}

size_t bwd_wgrad14_scratch_bytes(int N, int H, int W, int nqc1) {
  const int npairs = 4 * nqc1 + 6;
  const int ns = wgrad_nsplit(N, H, W, npairs, 9);
  return up256((size_t)ns * 4 * 32 * sizeof(float)) + (size_t)ns * npairs * 9 * 4096;
}

namespace {
template <int TAPS>
int launch_wgrad_any(const WgArgs& a, int nsplit, int gy, int gz, hipStream_t s) {
  hipLaunchKernelGGL(wgrad_kernel<TAPS>, dim3((unsigned)nsplit, (unsigned)gy, (unsigned)gz), dim3((TAPS == 1 ? 4 : 3) * 64), 0, s, a);
  return hip_rc(hipGetLastError());
}
}  // namespace

// a deferred weight-gradient launch (host side): the kernel arguments + its own grid extent
struct WgJob {
  WgArgs a;
  int kind;              // 0: conv1..4 (wgrad_body<9>, multi), 1: temporal conv5
  int gx, gy, nz;        // pixel splits, pairs (kind 0) / input planes (kind 1), gradient planes (kind 1)
};

static void wg_job_set(WgJob* dst, const WgArgs& a, int kind, int gx, int gy, int nz);
static int bwd_wgrad_impl(const WgradJob& j, const float* amax, void* scratch, int N, int T, int H, int W, hipStream_t s, FinArgs* defer,
                          WgJob* defer_wg = nullptr) {
  const size_t npix = (size_t)N * H * W, plane = npix * 32;
  const int qtot = j.Qn[0] + j.Qn[1];
  const int ttot = j.taps == 9 ? 9 : (j.temporal ? 3 : 1);
  const int ts = ttot == 1 ? 16 : wg_tile_side(H, W);
  const int tiles_x = wg_tiles(W, ts), tiles_y = wg_tiles(H, ts);
  int nsplit = wgrad_nsplit(N, H, W, j.Pn * qtot, ttot);
  if (ttot == 3) {                                   // the temporal kernel's units are (clip, tile): never more splits than units
    const int nunits = (N / T) * tiles_x * tiles_y;
    if (nsplit > nunits) nsplit = nunits;
    if (defer_wg && defer) nsplit = wgrad_nsplit_table(nsplit, (long)nunits * T);
  }
  float* bpart = (float*)scratch;
  float* part = (float*)((unsigned char*)scratch + up256((size_t)nsplit * (j.Pn < 4 ? 4 : j.Pn) * 32 * sizeof(float)));
  if (!j.wout && !j.bout) return SELFC_OK;
  WgArgs a{};
  a.P = (const f16*)j.P; a.Q0 = (const f16*)j.Q[0]; a.nq0 = j.Qn[0]; a.Q1 = (const f16*)j.Q[1];
  a.part = part; a.bpart = j.bout ? bpart : nullptr; a.plane = plane;
  a.N = N; a.T = T; a.H = H; a.W = W; a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.ntiles = N * tiles_x * tiles_y; a.ts = ts;
  int rc = SELFC_OK;
  const bool dwg = defer_wg && defer && ttot == 3;           // only the temporal job can be deferred (the table kernels cover multi + temporal)
  if (dwg) wg_job_set(defer_wg, a, 1, nsplit, qtot, j.Pn);
  else rc = ttot == 9 ? launch_wgrad_any<9>(a, nsplit, qtot, j.Pn, s)
          : ttot == 1 ? launch_wgrad_any<1>(a, nsplit, qtot, j.Pn, s) : SELFC_OK;
  if (ttot == 3 && !dwg) {
    hipLaunchKernelGGL(wgrad_temporal_kernel, dim3((unsigned)nsplit, (unsigned)qtot, (unsigned)j.Pn), dim3(192), 0, s, a);
    rc = hip_rc(hipGetLastError());
  }
  if (rc) return rc;
  FinArgs f{};
  f.part = part; f.bpart = bpart; f.out[0] = j.wout; f.bout[0] = j.bout; f.nW = nsplit; f.Pn = j.Pn; f.qtot = qtot; f.ttot = ttot;
  f.O = j.O; f.Ctot = j.Ctot; f.cin = j.cin; f.nx = j.nx; f.npairs = j.Pn * qtot; f.amax = amax; f.beta = j.beta;
  if (defer) { *defer = f; return SELFC_OK; }          // the caller finishes this job together with another one
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3(fin_blocks(f)), dim3(256), 0, s, f);
  return hip_rc(hipGetLastError());
}

int bwd_wgrad(const WgradJob& j, const float* amax, void* scratch, int N, int T, int H, int W, hipStream_t s) {
  return bwd_wgrad_impl(j, amax, scratch, N, T, H, W, s, nullptr);
}

// conv1..conv4 of one dense block in one launch + one finish.  dpre: the four gradient planes [dpre4 dpre3 dpre2 dpre1];
// inputs of conv k = the first (nqc1 + k - 1) planes of the run Q0 (nq0 planes) followed by Q1.
// arguments of the conv1..4 job (one launch) and of its finish
static void wgrad14_args(const void* dpre, const void* Q0, int nq0, const void* Q1, int nqc1, int cin, int nx,
                         float* const* wout, float* const* bout, float beta, const float* amax, void* scratch,
                         int N, int T, int H, int W, WgArgs& a, FinArgs& f, int& nsplit, int& npairs, bool table = false) {
  const size_t npix = (size_t)N * H * W, plane = npix * 32;
  npairs = 4 * nqc1 + 6;
  nsplit = wgrad_nsplit(N, H, W, npairs, 9);
  const int ts = wg_tile_side(H, W);
  if (table) nsplit = wgrad_nsplit_table(nsplit, (long)N * wg_tiles(H, ts) * wg_tiles(W, ts));
  float* bpart = (float*)scratch;
  float* part = (float*)((unsigned char*)scratch + up256((size_t)nsplit * 4 * 32 * sizeof(float)));
  a = WgArgs{};
  a.P = (const f16*)dpre; a.Q0 = (const f16*)Q0; a.nq0 = nq0; a.Q1 = (const f16*)Q1;
  a.part = part; a.bpart = bout ? bpart : nullptr; a.plane = plane;
  a.N = N; a.T = T; a.H = H; a.W = W; a.tiles_x = wg_tiles(W, ts); a.tiles_y = wg_tiles(H, ts); a.ntiles = N * a.tiles_x * a.tiles_y; a.ts = ts;
  a.multi = 1; a.nqc1 = nqc1;
  f = FinArgs{};
  f.part = part; f.bpart = bpart; f.nW = nsplit; f.Pn = 4; f.ttot = 9; f.O = 32; f.cin = cin; f.nx = nx;
  f.multi = 1; f.nqc1 = nqc1; f.npairs = npairs; f.amax = amax; f.beta = beta;
  for (int k = 0; k < 4; ++k) { f.out[k] = wout ? wout[k] : nullptr; f.bout[k] = bout ? bout[k] : nullptr; }
}

int bwd_wgrad14(const void* dpre, const void* Q0, int nq0, const void* Q1, int nqc1, int cin, int nx,
                float* const* wout, float* const* bout, float beta, const float* amax, void* scratch,
                int N, int T, int H, int W, hipStream_t s, FinArgs* defer, WgJob* defer_wg = nullptr) {
  WgArgs a; FinArgs f; int nsplit, npairs;
  wgrad14_args(dpre, Q0, nq0, Q1, nqc1, cin, nx, wout, bout, beta, amax, scratch, N, T, H, W, a, f, nsplit, npairs, defer_wg && defer);
  if (defer_wg && defer) { *defer_wg = WgJob{a, 0, nsplit, npairs, 1}; *defer = f; return SELFC_OK; }
  int rc = launch_wgrad_any<9>(a, nsplit, npairs, 1, s);
  if (rc) return rc;
  if (defer) { *defer = f; return SELFC_OK; }
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3(fin_blocks(f)), dim3(256), 0, s, f);
  return hip_rc(hipGetLastError());
}

// run deferred weight-gradient jobs: all conv1..4 jobs as one launch per WG_TABLE, all temporal jobs as another
static int launch_wg_jobs(const WgJob* jobs, int njobs, hipStream_t s) {
  for (int kind = 0; kind < 2; ++kind) {
    int j = 0;
    while (j < njobs) {
      static const int xcd_mode = getenv("SELFC_WG_XCD") ? atoi(getenv("SELFC_WG_XCD")) : 1;
      WgTable t{};
      WgTTable tt{};
      int mx = 0, n = 0;                       // mx: workgroups per job slice (x), a multiple of 8 in XCD mode
      for (; j < njobs && n < WG_TABLE; ++j) {
        if (jobs[j].kind != kind) continue;
        const int pairs = kind == 0 ? jobs[j].gy : jobs[j].gy * jobs[j].nz;
        if (kind == 0) { t.job[n] = jobs[j].a; t.gx[n] = jobs[j].gx; t.gy[n] = jobs[j].gy; }
        else { tt.job[n] = jobs[j].a; tt.gx[n] = jobs[j].gx; tt.gy[n] = jobs[j].gy; tt.nz[n] = jobs[j].nz; }
        const int need = (xcd_mode && jobs[j].gx >= 8) ? ((jobs[j].gx + 7) / 8) * 8 * pairs : jobs[j].gx * pairs;
        mx = need > mx ? need : mx;
        ++n;
      }
      if (!n) break;
      mx = (mx + 7) & ~7;                      // every job's slice starts on XCD 0
      if (kind == 0) {
        t.n = n; t.xcd = xcd_mode;
        hipLaunchKernelGGL(wgrad_table_kernel, dim3((unsigned)mx, 1, (unsigned)n), dim3(192), 0, s, t);
      } else {
        tt.n = n; tt.xcd = xcd_mode;
        hipLaunchKernelGGL(wgrad_temporal_table_kernel, dim3((unsigned)mx, 1, (unsigned)n), dim3(192), 0, s, tt);
      }
      const int rc = hip_rc(hipGetLastError());
      if (rc) return rc;
    }
  }
  return SELFC_OK;
}

static void wg_job_set(WgJob* dst, const WgArgs& a, int kind, int gx, int gy, int nz) { *dst = WgJob{a, kind, gx, gy, nz}; }

// launch up to FIN_TABLE deferred finish jobs per kernel
static int launch_fin_jobs(const FinArgs* jobs, int njobs, hipStream_t s) {
  for (int j0 = 0; j0 < njobs; j0 += FIN_TABLE) {
    FinTable t{};
    t.n = njobs - j0 < FIN_TABLE ? njobs - j0 : FIN_TABLE;
    unsigned nb = 0;
    for (int j = 0; j < t.n; ++j) { t.job[j] = jobs[j0 + j]; t.start[j] = nb; nb += fin_blocks(jobs[j0 + j]); }
    t.start[t.n] = nb;
    if (!nb) continue;
    hipLaunchKernelGGL(wgrad_finish_table_kernel, dim3(nb), dim3(256), 0, s, t);
    const int rc = hip_rc(hipGetLastError());
    if (rc) return rc;
  }
  return SELFC_OK;
}

}  // namespace selfc

extern "C" {

size_t selfc_subnet_bwd_scratch_bytes(int N, int H, int W, int cin, int cout) {
  if (N <= 0 || H <= 0 || W <= 0 || cin < 1 || cout < 1) return 0;
  return bwd_layout(N, H, W, cin, cout).total;
}

int selfc_subnet_bwd(const selfc_subnet_bw* bw, int kind, const void* dense, const float* xin, const float* dout, float sign,
                     float* dx, int accumulate_dx, float* const* wgrad, float* const* bgrad, float beta,
                     void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout, void* stream) {
  return selfc_subnet_bwd_phase(SELFC_BWD_DATA | SELFC_BWD_WEIGHTS, bw, kind, dense, xin, dout, sign, dx, accumulate_dx, wgrad, bgrad,
                                beta, scratch, scratch_bytes, N, T, H, W, cin, cout, stream);
}

int selfc_subnet_bwd_phase(int phases, const selfc_subnet_bw* bw, int kind, const void* dense, const float* xin, const float* dout,
                           float sign, float* dx, int accumulate_dx, float* const* wgrad, float* const* bgrad, float beta,
                           void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout, void* stream) {
  return selfc_subnet_bwd_phase_x(phases, bw, kind, dense, xin, dout, sign, dx, accumulate_dx, wgrad, bgrad, beta, scratch, scratch_bytes,
                                  N, T, H, W, cin, cout, nullptr, nullptr, stream);
}

int selfc_subnet_bwd_phase_x(int phases, const selfc_subnet_bw* bw, int kind, const void* dense, const float* xin, const float* dout,
                             float sign, float* dx, int accumulate_dx, float* const* wgrad, float* const* bgrad, float beta,
                             void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout,
                             const float* dout_amax, float* dx_amax_out, void* stream) {
  return selfc_subnet_bwd_phase_d(phases, bw, kind, dense, xin, dout, sign, dx, accumulate_dx, wgrad, bgrad, beta, scratch, scratch_bytes,
                                  N, T, H, W, cin, cout, dout_amax, dx_amax_out, nullptr, nullptr, stream);
}

size_t selfc_fin_job_bytes(void) { return sizeof(FinArgs); }
size_t selfc_wg_job_bytes(void) { return sizeof(WgJob); }

int selfc_wgrad_run_jobs(const void* jobs, int njobs, void* stream) {
  if (!jobs || njobs < 0) return SELFC_EINVAL;
  return launch_wg_jobs((const WgJob*)jobs, njobs, (hipStream_t)stream);
}

int selfc_wgrad_finish_jobs(const void* jobs, int njobs, void* stream) {
  if (!jobs || njobs < 0) return SELFC_EINVAL;
  return launch_fin_jobs((const FinArgs*)jobs, njobs, (hipStream_t)stream);
}

int selfc_subnet_bwd_phase_d(int phases, const selfc_subnet_bw* bw, int kind, const void* dense, const float* xin, const float* dout,
                             float sign, float* dx, int accumulate_dx, float* const* wgrad, float* const* bgrad, float beta,
                             void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout,
                             const float* dout_amax, float* dx_amax_out, void* fin_jobs, void* wg_jobs, void* stream) {
  if (!(phases & (SELFC_BWD_DATA | SELFC_BWD_WEIGHTS))) return SELFC_EINVAL;
  if (wg_jobs && (!fin_jobs || kind != SELFC_SUBNET_D2DT)) return SELFC_EINVAL;      // deferred launches need deferred finishes; temporal conv5 only
  ThinScope thin((phases & SELFC_BWD_WG_THIN) != 0 && wg_jobs != nullptr);
  if (!bw || !dense || !dout || !scratch || !bw->wt5 || !bw->wtx || !bw->wtd[0] || !bw->wtd[1] || !bw->wtd[2]) return SELFC_EINVAL;
  if (N <= 0 || T <= 0 || N % T || H <= 0 || W <= 0 || cin < 1 || cin > 96 || cout < 1 || cout > 96) return SELFC_EINVAL;
  if (kind != SELFC_SUBNET_D2DT && kind != SELFC_SUBNET_DB2D) return SELFC_EINVAL;
  const BwdLayout L = bwd_layout(N, H, W, cin, cout);
  if (L.hasx && !xin) return SELFC_EINVAL;
  if (scratch_bytes < L.total) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_BWD, s);
  unsigned char* sb = (unsigned char*)scratch;
  const size_t npix = (size_t)N * H * W, plane = npix * 32;
  f16* gb = (f16*)sb;                                  // planes 0..3: dpre4, dpre3, dpre2, dpre1
  f16* gpl = (f16*)(sb + L.off_g);                     // dOut planes
  f16* t5 = (f16*)(sb + L.off_t5);                     // conv5^T(dOut): nx x-groups, f1, f2, f3
  f16* xpl = (f16*)(sb + L.off_xplane);                // f16 copy of the input when it is not in `dense`
  float* amax = (float*)(sb + L.off_amax);
  void* wgs = sb + L.off_wg;
  const f16* dn = (const f16*)dense;
  const f16* feat = dn + (size_t)(L.hasx ? 0 : L.nx) * plane;      // f1..f4
  const int coutp = (cout + 3) & ~3, cinp = (cin + 3) & ~3;
  const bool d2dt = kind == SELFC_SUBNET_D2DT;
  int rc;
  if (!(phases & SELFC_BWD_DATA)) goto weights;

  // 1. scale + scaled f16 planes of dOut.  dout_amax: the producer of dOut already took max|dOut| (selfc_coupling_bwd_x,
  //    selfc_add_absmax, dx_amax_out of the call that wrote it): no pass over dOut, and the to-planes kernel hands the value on
  //    to this phase's slot (a 4-byte copy node in front of every subnet cost 0.06 - 0.17 ms of the captured step)
  if (!dout_amax && (rc = bwd_absmax(dout, npix * coutp, amax, s))) return rc;
  if ((rc = bwd_to_planes(dout, gpl, npix, cout, coutp, L.ng, 0, sign, dout_amax ? dout_amax : amax, s, dout_amax ? amax : nullptr))) return rc;
  if (L.hasx && (rc = selfc_nhwc_to_planes(xin, xpl, npix, cin, stream))) return rc;      // (stays in the data phase: xin may be rewritten before the weight phase runs on its side stream)

  // 2. conv5^T(dOut): x-groups and f1..f3 as addend planes, f4 masked straight into dpre4.  Temporal conv5 (D2DTInput):
  //    the frame-walking temporal-conv kernel, one 32-channel output plane per blockIdx.y; 3x3 conv5 (DenseBlock): the
  //    generic plane conv.
  if (d2dt) {
    if ((rc = bwd_tconv5T(gpl, L.ng, bw->wt5, L.nx + 4, t5, feat + 3 * plane, L.nx + 3, gb, N, T, H, W, s))) return rc;
  } else {
    BwdConv c{};
    c.in = gpl; c.nplanes_in = L.ng; c.kt = 1; c.sp1 = 0; c.w = bw->wt5;
    c.ngroups = L.nx + 4; c.out_planes = t5;
    c.mask = feat + 3 * plane; c.mask_z = L.nx + 3; c.alt = gb;
    c.amax = amax;
    if ((rc = bwd_conv_planes(c, N, T, H, W, s))) return rc;
  }
  // 3 + 4 as ONE launch (csrc/dgrad_chain.hip: the chain in LDS, bit-identical values) while its workgroups - one per 12x16
  // tile and frame, one per CU - need at most two rounds (F: 512 workgroups; a G/H pair: four rounds, 1,024): since the chain's
  // MFMAs are fed one tap ahead (r6y) it beats the layer-wise launches up to there (same-box sweep r6aa: 7.98 -> 7.75 ms at four
  // septuplets per rank, 11.98 -> 11.88 at eight; in round 5, with hipcc's read - wait - multiply loop, one round was the limit).
  // SELFC_BWD_CHAIN=0 / 1: never / always (tests).
  {
    static const int chain_env = getenv("SELFC_BWD_CHAIN") ? atoi(getenv("SELFC_BWD_CHAIN")) : -1;
    static const long chain_max = getenv("SELFC_BWD_CHAIN_MAX1") ? atol(getenv("SELFC_BWD_CHAIN_MAX1")) : 512;
    const long chain_wgs = (long)N * ((H + 11) / 12) * ((W + 15) / 16);
    if (chain_env == 1 || (chain_env < 0 && chain_wgs <= chain_max)) {
      const void* wtd[3] = {bw->wtd[0], bw->wtd[1], bw->wtd[2]};
      if ((rc = bwd_dgrad_chain(gb, t5, feat, wtd, bw->wtx, dx, L.nx, cinp, accumulate_dx, amax, dx_amax_out, N, H, W, s))) return rc;
      goto weights;
    }
  }
  // 3. dpre3, dpre2, dpre1
  for (int j = 3; j >= 1; --j) {
    BwdConv c{};
    c.in = gb; c.nplanes_in = 4 - j; c.kt = 1; c.sp1 = 0; c.w = bw->wtd[3 - j];
    c.ngroups = 1; c.out_planes = gb + (size_t)(4 - j) * plane;
    c.add = t5 + (size_t)(L.nx + j - 1) * plane;
    c.mask = feat + (size_t)(j - 1) * plane; c.mask_z = 0;
    c.amax = amax;
    if ((rc = bwd_conv_planes(c, N, T, H, W, s))) return rc;
  }
  // 4. dx
  if (dx) {
    BwdConv c{};
    c.in = gb; c.nplanes_in = 4; c.kt = 1; c.sp1 = 0; c.w = bw->wtx;
    c.ngroups = L.nx; c.add = t5; c.mask_z = -1;
    c.plain = dx; c.coutp = cinp; c.accumulate = accumulate_dx; c.amax = amax; c.amax_out = dx_amax_out;
    if ((rc = bwd_conv_planes(c, N, T, H, W, s))) return rc;
  }
weights:
  if (!(phases & SELFC_BWD_WEIGHTS) || (!wgrad && !bgrad)) return SELFC_OK;

  // 5. weight / bias gradients: conv1..4 in one launch, conv5 in another (reads only what the data phase left in scratch)
  {
    // input planes: [x planes][f1..]; with cin <= 3 the x plane is the scratch copy and the features start `dense`
    const void* q0 = L.hasx ? (const void*)xpl : (const void*)dn;
    const int nq0 = L.hasx ? 1 : L.nx + 4;
    FinArgs fa{}, fb{};
    WgJob* wj = (WgJob*)wg_jobs;
    if ((rc = bwd_wgrad14(gb, q0, nq0, feat, L.nx, cin, L.nx, wgrad, bgrad, beta, amax, wgs, N, T, H, W, s, &fa, wj))) return rc;
    WgradJob j{};
    j.P = gpl; j.Pn = L.ng;
    j.Q[0] = q0; j.Qn[0] = nq0;
    if (L.hasx) { j.Q[1] = feat; j.Qn[1] = 4; }
    j.temporal = d2dt ? 1 : 0;
    j.taps = d2dt ? 1 : 9;
    j.wout = wgrad ? wgrad[4] : nullptr;
    j.bout = bgrad ? bgrad[4] : nullptr;
    j.O = cout; j.Ctot = cin + 128; j.cin = cin; j.nx = L.nx; j.beta = beta;
    const bool has5 = j.wout || j.bout;
    if (wj) wj[1] = WgJob{WgArgs{}, 1, 0, 0, 0};           // (stays empty when conv5 wants no gradient: extent 0)
    if ((rc = bwd_wgrad_impl(j, amax, sb + L.off_wg5, N, T, H, W, s, &fb, wj ? wj + 1 : nullptr))) return rc;
    if (fin_jobs) {          // deferred: the caller finishes many subnets' jobs in one launch (selfc_wgrad_finish_jobs); a job without
      FinArgs* out = (FinArgs*)fin_jobs;       // outputs has zero blocks there
      out[0] = fa;
      out[1] = has5 ? fb : FinArgs{};
      return SELFC_OK;
    }
    if (has5) {
      const unsigned na = fin_blocks(fa);
      hipLaunchKernelGGL(wgrad_finish2_kernel, dim3(na + fin_blocks(fb)), dim3(256), 0, s, fa, fb, na);
    } else {
      hipLaunchKernelGGL(wgrad_finish_kernel, dim3(fin_blocks(fa)), dim3(256), 0, s, fa);
    }
    if ((rc = hip_rc(hipGetLastError()))) return rc;
  }
  return SELFC_OK;
}

// ---------------------------------------------------------------------------------------------------------
// G and H of one InvBlockExp as ONE backward call (abi 13): the two subnets read the same input (y1 / x1), have the same
// shapes, and their input gradients add up - every step of selfc_subnet_bwd_phase_x runs once for both (grid dimension = net),
// under one gradient scale (grad_to_planes2_kernel), and the input gradient is a single conv over the eight dpre planes.
// Against two calls on two streams: half the launches, no fork / join (a cross-queue dependency costs ~10 us inside a replayed
// graph on this runtime), no d1 += d1h pass, and launches that fill the chip on a training crop.  D2DTInput subnets only.
// ---------------------------------------------------------------------------------------------------------
namespace {
struct PairLayout {
  int nx, ng;
  size_t plane_b;
  size_t off_gb, off_g, off_t5, off_xplane, off_dxh, off_amax, off_wg[2], off_wg5[2], total;
};
PairLayout pair_layout(int N, int H, int W, int cin, int cout) {
  PairLayout L{};
  L.nx = (cin + 31) / 32;
  L.ng = (cout + 31) / 32;
  L.plane_b = (size_t)N * H * W * 64;
  L.off_gb = 0;                                             // 8 planes: G's dpre4..1, H's dpre4..1
  L.off_g = 8 * L.plane_b;                                  // dOut planes: G's ng, H's ng
  L.off_t5 = L.off_g + 2 * (size_t)L.ng * L.plane_b;        // conv5^T(dOut): (nx + 3) planes per net
  L.off_xplane = L.off_t5 + 2 * (size_t)(L.nx + 3) * L.plane_b;
  L.off_dxh = L.off_xplane + L.plane_b;                       // H's input gradient on the chain path (fp32, <= 4 channels per pixel)
  L.off_amax = L.off_dxh + up256((size_t)N * H * W * 4 * sizeof(float));
  size_t o = up256(L.off_amax + 256);
  const size_t a4 = bwd_wgrad14_scratch_bytes(N, H, W, L.nx);
  const size_t a5 = bwd_wgrad_scratch_bytes(N, H, W, L.ng, L.nx + 4, 3);
  for (int q = 0; q < 2; ++q) {
    L.off_wg[q] = o; o = up256(o + a4);
    L.off_wg5[q] = o; o = up256(o + a5);
  }
  L.total = o;
  return L;
}
}  // namespace

size_t selfc_gh_bwd_pair_scratch_bytes(int N, int H, int W, int cin, int cout) {
  if (N <= 0 || H <= 0 || W <= 0 || cin < 1 || cin > 3 || cout < 1 || cout > 96) return 0;
  return pair_layout(N, H, W, cin, cout).total;
}

int selfc_gh_bwd_pair(int phases, const selfc_subnet_bw* bw_g, const selfc_subnet_bw* bw_h, const void* dense_g, const void* dense_h,
                      const float* xin, const float* dout_g, const float* dout_h, float sign_g, float sign_h,
                      float* dx, int accumulate_dx, float* const* wgrad_g, float* const* bgrad_g, float* const* wgrad_h, float* const* bgrad_h,
                      float beta, void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout,
                      const float* amax_g, const float* amax_h, float* dx_amax_out, void* fin_jobs, void* wg_jobs, void* stream) {
  if (!(phases & (SELFC_BWD_DATA | SELFC_BWD_WEIGHTS))) return SELFC_EINVAL;
  if (wg_jobs && !fin_jobs) return SELFC_EINVAL;
  ThinScope thin((phases & SELFC_BWD_WG_THIN) != 0 && wg_jobs != nullptr);
  if (!bw_g || !bw_h || !dense_g || !dense_h || !xin || !dout_g || !dout_h || !scratch) return SELFC_EINVAL;
  for (const selfc_subnet_bw* bw : {bw_g, bw_h})
    if (!bw->wt5 || !bw->wtx || !bw->wtd[0] || !bw->wtd[1] || !bw->wtd[2]) return SELFC_EINVAL;
  if (N <= 0 || T <= 0 || N % T || H <= 0 || W <= 0 || cin < 1 || cin > 3 || cout < 1 || cout > 96) return SELFC_EINVAL;
  const PairLayout L = pair_layout(N, H, W, cin, cout);
  if (scratch_bytes < L.total) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_BWD, s);
  unsigned char* sb = (unsigned char*)scratch;
  const size_t npix = (size_t)N * H * W, plane = npix * 32;
  f16* gb[2] = {(f16*)(sb + L.off_gb), (f16*)(sb + L.off_gb) + 4 * plane};
  f16* gpl[2] = {(f16*)(sb + L.off_g), (f16*)(sb + L.off_g) + (size_t)L.ng * plane};
  f16* t5[2] = {(f16*)(sb + L.off_t5), (f16*)(sb + L.off_t5) + (size_t)(L.nx + 3) * plane};
  f16* xpl = (f16*)(sb + L.off_xplane);
  float* amax = (float*)(sb + L.off_amax);                 // [0] the pair's common maximum, [1], [2] the nets' own when the caller has none
  const f16* feat[2] = {(const f16*)dense_g, (const f16*)dense_h};       // cin <= 3: the dense buffers start at f1
  const selfc_subnet_bw* bw[2] = {bw_g, bw_h};
  const int coutp = (cout + 3) & ~3, cinp = (cin + 3) & ~3;
  int rc;
  if (phases & SELFC_BWD_DATA) {
    // 1. both dOut tensors as scaled f16 planes under one scale
    if (!amax_g) { if ((rc = bwd_absmax(dout_g, npix * coutp, amax + 1, s))) return rc; amax_g = amax + 1; }
    if (!amax_h) { if ((rc = bwd_absmax(dout_h, npix * coutp, amax + 2, s))) return rc; amax_h = amax + 2; }
    {
      const size_t items = npix * (size_t)L.ng * 4;
      hipLaunchKernelGGL(grad_to_planes2_kernel, dim3((unsigned)((items + 255) / 256), 2), dim3(256), 0, s, dout_g, dout_h, gpl[0], gpl[1], npix,
                         cout, coutp, L.ng, sign_g, sign_h, amax_g, amax_h, amax);
      if ((rc = hip_rc(hipGetLastError()))) return rc;
    }
    if ((rc = selfc_nhwc_to_planes(xin, xpl, npix, cin, stream))) return rc;
    // 2. conv5^T(dOut) of both nets: x-groups and f1..f3 as addend planes, f4 masked straight into dpre4
    if ((rc = bwd_tconv5T_pair(gpl[0], gpl[1], L.ng, bw_g->wt5, bw_h->wt5, L.nx + 4, t5[0], t5[1], feat[0] + 3 * plane, feat[1] + 3 * plane,
                               L.nx + 3, gb[0], gb[1], N, T, H, W, s))) return rc;
    // 3. dpre3, dpre2, dpre1 of both nets: the chain kernel while its workgroups fit the chip in one round, else layer by layer
    static const int chain_env = getenv("SELFC_BWD_CHAIN") ? atoi(getenv("SELFC_BWD_CHAIN")) : -1;
    static const long chain_max = getenv("SELFC_BWD_CHAIN_MAX2") ? atol(getenv("SELFC_BWD_CHAIN_MAX2")) : 1024;
    const long chain_wgs = 2L * N * ((H + 11) / 12) * ((W + 15) / 16);
    if (chain_env == 1 || (chain_env < 0 && chain_wgs <= chain_max)) {
      const void* wtd0[3] = {bw_g->wtd[0], bw_g->wtd[1], bw_g->wtd[2]};
      const void* wtd1[3] = {bw_h->wtd[0], bw_h->wtd[1], bw_h->wtd[2]};
      static const int chain_dx = getenv("SELFC_BWD_CHAIN_DX") ? atoi(getenv("SELFC_BWD_CHAIN_DX")) : 1;
      if (dx && chain_dx && cinp <= 4) {
        // the chains' own dx layers (G into dx, H into a scratch buffer) + one add with the maximum: 26 + 5 us instead of 19 + 18 us for
        // the eight-stage conv below on one 36x36 septuplet
        float* dxh = (float*)(sb + L.off_dxh);
        if ((rc = bwd_dgrad_chain_pair(gb[0], gb[1], t5[0], t5[1], feat[0], feat[1], wtd0, wtd1, L.nx, amax, N, H, W, s,
                                       bw_g->wtx, bw_h->wtx, dx, dxh, cinp, accumulate_dx))) return rc;
        if ((rc = selfc_add_absmax(dx, dxh, npix * cinp, dx_amax_out, stream))) return rc;
        dx = nullptr;                           // done
      } else if ((rc = bwd_dgrad_chain_pair(gb[0], gb[1], t5[0], t5[1], feat[0], feat[1], wtd0, wtd1, L.nx, amax, N, H, W, s))) return rc;
    } else {
      for (int j = 3; j >= 1; --j) {
        BwdConv c[2];
        for (int q = 0; q < 2; ++q) {
          c[q] = BwdConv{};
          c[q].in = gb[q]; c[q].nplanes_in = 4 - j; c[q].kt = 1; c[q].sp1 = 0; c[q].w = bw[q]->wtd[3 - j];
          c[q].ngroups = 1; c[q].out_planes = gb[q] + (size_t)(4 - j) * plane;
          c[q].add = t5[q] + (size_t)(L.nx + j - 1) * plane;
          c[q].mask = feat[q] + (size_t)(j - 1) * plane; c[q].mask_z = 0;
          c[q].amax = amax;
        }
        if ((rc = bwd_conv_planes_pair(c[0], c[1], N, T, H, W, s))) return rc;
      }
    }
    // 4. dx: one conv over the eight dpre planes of both nets (+ both conv5^T x-parts)
    if (dx) {
      BwdConv c{};
      c.in = gb[0]; c.nplanes_in = 4; c.kt = 1; c.sp1 = 0; c.w = bw_g->wtx;
      c.in2 = gb[1]; c.nplanes_in2 = 4; c.w2 = bw_h->wtx; c.add2 = t5[1];
      c.ngroups = L.nx; c.add = t5[0]; c.mask_z = -1;
      c.plain = dx; c.coutp = cinp; c.accumulate = accumulate_dx; c.amax = amax; c.amax_out = dx_amax_out;
      if ((rc = bwd_conv_planes(c, N, T, H, W, s))) return rc;
    }
  }
  if (!(phases & SELFC_BWD_WEIGHTS) || (!wgrad_g && !bgrad_g && !wgrad_h && !bgrad_h)) return SELFC_OK;
  // 5. weight / bias gradients: conv1..4 of both nets in one launch, the temporal conv5 of both in another
  {
    float* const* wg[2] = {wgrad_g, wgrad_h};
    float* const* bg[2] = {bgrad_g, bgrad_h};
    WgArgs a14[2], a5[2];
    FinArgs f14[2], f5[2];
    int nsplit14 = 0, npairs14 = 0, nsplit5 = 0;
    const int qtot = 1 + 4;                                  // x plane (the scratch copy) + f1..f4
    for (int q = 0; q < 2; ++q) {
      wgrad14_args(gb[q], xpl, 1, feat[q], L.nx, cin, L.nx, wg[q], bg[q], beta, amax, sb + L.off_wg[q], N, T, H, W, a14[q], f14[q], nsplit14, npairs14,
                   wg_jobs != nullptr);
      // conv5 (temporal taps): P = the dOut planes, Q = [x | f1..f4]
      nsplit5 = wgrad_nsplit(N, H, W, L.ng * qtot, 3);
      const int ts5 = wg_tile_side(H, W);
      const int nunits = (N / T) * wg_tiles(W, ts5) * wg_tiles(H, ts5);
      if (nsplit5 > nunits) nsplit5 = nunits;
      if (wg_jobs) nsplit5 = wgrad_nsplit_table(nsplit5, (long)nunits * T);
      float* bpart = (float*)(sb + L.off_wg5[q]);
      float* part = (float*)(sb + L.off_wg5[q] + up256((size_t)nsplit5 * (L.ng < 4 ? 4 : L.ng) * 32 * sizeof(float)));
      float* wout = wg[q] ? wg[q][4] : nullptr;
      float* bout = bg[q] ? bg[q][4] : nullptr;
      WgArgs& a = a5[q];
      a = WgArgs{};
      a.P = gpl[q]; a.Q0 = xpl; a.nq0 = 1; a.Q1 = feat[q];
      a.part = part; a.bpart = bout ? bpart : nullptr; a.plane = plane;
      a.N = N; a.T = T; a.H = H; a.W = W; a.tiles_x = wg_tiles(W, ts5); a.tiles_y = wg_tiles(H, ts5); a.ntiles = N * a.tiles_x * a.tiles_y; a.ts = ts5;
      FinArgs& f = f5[q];
      f = FinArgs{};
      if (wout || bout) {
        f.part = part; f.bpart = bpart; f.out[0] = wout; f.bout[0] = bout; f.nW = nsplit5; f.Pn = L.ng; f.qtot = qtot; f.ttot = 3;
        f.O = cout; f.Ctot = cin + 128; f.cin = cin; f.nx = L.nx; f.npairs = L.ng * qtot; f.amax = amax; f.beta = beta;
      }
    }
    const FinArgs jobs[4] = {f14[0], f5[0], f14[1], f5[1]};
    if (wg_jobs) {           // the launches themselves are deferred too: the caller runs a whole stack's jobs as two launches
      WgJob* wj = (WgJob*)wg_jobs;
      for (int q = 0; q < 2; ++q) {
        wj[2 * q] = WgJob{a14[q], 0, nsplit14, npairs14, 1};
        wj[2 * q + 1] = WgJob{a5[q], 1, nsplit5, qtot, L.ng};
      }
      for (int j = 0; j < 4; ++j) ((FinArgs*)fin_jobs)[j] = jobs[j];
      return SELFC_OK;
    }
    hipLaunchKernelGGL(wgrad_pair_kernel, dim3((unsigned)nsplit14, (unsigned)npairs14, 2), dim3(192), 0, s, a14[0], a14[1]);
    if ((rc = hip_rc(hipGetLastError()))) return rc;
    hipLaunchKernelGGL(wgrad_temporal_pair_kernel, dim3((unsigned)nsplit5, (unsigned)qtot, (unsigned)(2 * L.ng)), dim3(192), 0, s, a5[0], a5[1]);
    if ((rc = hip_rc(hipGetLastError()))) return rc;
    if (fin_jobs) {
      for (int j = 0; j < 4; ++j) ((FinArgs*)fin_jobs)[j] = jobs[j];
      return SELFC_OK;
    }
    return launch_fin_jobs(jobs, 4, s);
  }
}

// ---------------------------------------------------------------------------------------------------------
// ReconstructionLoss (models/modules/loss.py:5-21) as two launches: v = (x - t)^2 ('l2') or sqrt((x - t)^2 + eps) ('l1'), the
// reference's mean over the four axes (equal group sizes: sum / count), AND dv/dx in the same pass (the gradient of the loss w.r.t.
// x does not depend on the loss value) - the torch expression is ~10 element-wise / reduction launches forward and ~12 backward, each
// a 4.5-us node of a replayed training step.  Deterministic: a fixed block -> chunk map, a tree per block, partials summed in order
// by one block (no atomics, no memset: capturable).  x / t: n_outer rows of `inner` contiguous floats at row strides sx / st (a
// channel slice of an NCHW tensor is such a view).
// ---------------------------------------------------------------------------------------------------------
namespace {
// sum of n doubles by ONE wave, the same order in every caller: lane l adds elements l, l + 64, ... in sequence, then a fixed
// xor tree (deterministic; a single thread walking 512 partials in global memory took 23 us)
__device__ __forceinline__ double wave_sum(const double* __restrict__ p, const int n, const int lane) {
  double t = 0.0;
  for (int i = lane; i < n; i += 64) t += p[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
  return t;
}
constexpr int LOSS_BLOCKS = 512;
__global__ __launch_bounds__(256) void recon_loss_kernel(const float* __restrict__ x, size_t sx, const float* __restrict__ t, size_t st, size_t n_outer,
                                                         size_t inner, int l1, float eps, float gscale, float* __restrict__ grad, double* __restrict__ partial) {
  const size_t total = n_outer * inner;
  const size_t chunk = (total + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < total ? lo + chunk : total;
  double acc = 0.0;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    const size_t r = i / inner, c = i - r * inner;
    const float d = x[r * sx + c] - t[r * st + c];
    float v, g;
    if (l1) { v = sqrtf(d * d + eps); g = d / v; }
    else { v = d * d; g = 2.f * d; }
    acc += (double)v;
    if (grad) grad[i] = g * gscale;
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(64) void recon_loss_final_kernel(const double* __restrict__ partial, int n, double scale, float* __restrict__ out) {
  const double s_ = wave_sum(partial, n, threadIdx.x);
  if (threadIdx.x == 0) *out = (float)(s_ * scale);
}
}  // namespace

int selfc_recon_loss_blocks(void) { return LOSS_BLOCKS; }

int selfc_recon_loss(const float* x, size_t stride_x, const float* t, size_t stride_t, size_t n_outer, size_t inner, int l1, float eps,
                     float weight, float* grad, double* partial, float* out, void* stream) {
  if (!x || !t || !partial || !out || n_outer == 0 || inner == 0 || stride_x < inner || stride_t < inner) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const double inv = (double)weight / ((double)n_outer * (double)inner);
  hipLaunchKernelGGL(recon_loss_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, s, x, stride_x, t, stride_t, n_outer, inner, l1, eps, (float)inv, grad, partial);
  int rc = hip_rc(hipGetLastError());
  if (rc) return rc;
  hipLaunchKernelGGL(recon_loss_final_kernel, dim3(1), dim3(64), 0, s, partial, LOSS_BLOCKS, inv, out);
  return hip_rc(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------
// clip_grad_norm_ + Adam on ONE flat buffer as two launches (models/SelfC_model.py:172-176: nn.utils.clip_grad_norm_ then
// optimizer.step(); torch.optim.Adam's capturable multi-tensor step is 16 launches + ~10 for the staged norm and the clip - 0.23 ms of
// 4.5-us nodes at the end of every replayed step).  Launch 1: per-block sums of squares of the gradient (fixed block -> chunk map,
// double, no atomics) and step += 1.  Launch 2: every block sums the partials in the same order (the norm, bit-identical in every
// block), scales its gradient slice by min(1, max_norm / (norm + 1e-6)) IN PLACE (the caller's .grad views hold the clipped
// gradient, as after clip_grad_norm_) and applies torch's Adam update in torch's operation order:
//   g += wd * p;  m = lerp(m, g, 1 - b1);  v = b2 * v + (1 - b2) g^2;
//   p += m / ((sqrt(v) / sqrt(1 - b2^t) + eps) / (lr / (b1^t - 1)))
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int ADAM_BLOCKS = 512;
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial, float* __restrict__ step) {
  const size_t chunk = ((n + gridDim.x - 1) / gridDim.x + 3) & ~(size_t)3;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  double acc = 0.0;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) { const float v = g[i]; acc += (double)v * (double)v; }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0];
    if (blockIdx.x == 0 && step) *step += 1.f;
  }
}
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                                                        const double* __restrict__ partial, int nb, float max_norm, const float* __restrict__ lr_dev, float lr_host,
                                                        float beta1, float beta2, float omb1, float omb2, float eps, float wd, const float* __restrict__ step_dev,
                                                        float step_host, float* __restrict__ norm_out) {
  __shared__ float s_coef;
  if (threadIdx.x < 64) {
    const double t = wave_sum(partial, nb, threadIdx.x);
    if (threadIdx.x == 0) {
      const float norm = (float)sqrt(t);
      if (blockIdx.x == 0 && norm_out) *norm_out = norm;
      s_coef = max_norm > 0.f ? fminf(max_norm / (norm + 1e-6f), 1.f) : 1.f;
    }
  }
  __syncthreads();
  const float coef = s_coef;
  const float lr = lr_dev ? *lr_dev : lr_host;
  const float t = step_dev ? *step_dev : step_host;              // already incremented
  const float ss = 1.f / ((powf(beta1, t) - 1.f) / lr);           // torch: bias_correction1 = beta1^t - 1; /= lr; reciprocal  (negative step size)
  const float bc2s = sqrtf(-(powf(beta2, t) - 1.f));              // sqrt(1 - beta2^t)
  const size_t chunk = ((n + gridDim.x - 1) / gridDim.x + 3) & ~(size_t)3;
  const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  for (size_t i = lo + threadIdx.x; i < hi; i += 256) {
    float gi = g[i] * coef;
    g[i] = gi;
    const float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    float mi = m[i], vi = v[i];
    mi = mi + omb1 * (gi - mi);                                   // lerp (weight 1 - beta1 rounded from double, as torch passes it)
    vi = vi * beta2 + omb2 * gi * gi;                             // mul_, addcmul_ (value 1 - beta2 likewise)
    m[i] = mi; v[i] = vi;
    const float den = (sqrtf(vi) / bc2s + eps) / ss;
    p[i] = pi + mi / den;
  }
}
}  // namespace

int selfc_clip_adam_blocks(void) { return ADAM_BLOCKS; }

int selfc_clip_adam(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double* partial, float max_norm,
                    const float* lr_dev, float lr_host, double beta1_d, double beta2_d, float eps, float weight_decay,
                    float* step_dev, float step_host, float* norm_out, void* stream) {
  const float beta1 = (float)beta1_d, beta2 = (float)beta2_d;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !partial || n == 0) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(ADAM_BLOCKS), dim3(256), 0, s, grad, n, partial, step_dev);
  int rc = hip_rc(hipGetLastError());
  if (rc) return rc;
  hipLaunchKernelGGL(clip_adam_kernel, dim3(ADAM_BLOCKS), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, n, partial, ADAM_BLOCKS, max_norm,
                     lr_dev, lr_host, beta1, beta2, (float)(1.0 - (double)beta1_d), (float)(1.0 - (double)beta2_d), eps, weight_decay, step_dev, step_host, norm_out);
  return hip_rc(hipGetLastError());
}

int selfc_coupling_fwd(int rev, const float* x2, const float* g, const float* h, float* y2, float* s, float clamp, size_t n, void* stream) {
  if (!x2 || !g || !h || !y2 || !s || n == 0 || (n & 3)) return SELFC_EINVAL;
  ProfScope prof(PROF_CONV5_GH, (hipStream_t)stream);
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(coupling_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rev,
                     (const float4*)x2, (const float4*)g, (const float4*)h, (float4*)y2, (float4*)s, clamp, n4);
  return hip_rc(hipGetLastError());
}

int selfc_coupling_bwd(int rev, const float* v, const float* s, const float* dy2, float* dx2, float* dh, float clamp,
                       size_t n, void* stream) {
  return selfc_coupling_bwd_x(rev, v, s, dy2, dx2, dh, clamp, n, nullptr, nullptr, stream);
}

int selfc_coupling_bwd_x(int rev, const float* v, const float* s, const float* dy2, float* dx2, float* dh, float clamp,
                         size_t n, float* dx2_amax, float* dh_amax, void* stream) {
  if (!v || !s || !dy2 || !dx2 || !dh || n == 0 || (n & 3) || clamp == 0.f) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  const size_t n4 = n / 4;
  size_t nb = (n4 + 255) / 256;
  if ((dx2_amax || dh_amax) && nb > 256) nb = 256;        // a maximum is one atomic per block (block_absmax)
  hipLaunchKernelGGL(coupling_bwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, rev,
                     (const float4*)v, (const float4*)s, (const float4*)dy2, (float4*)dx2, (float4*)dh, clamp, n4,
                     (unsigned*)dx2_amax, (unsigned*)dh_amax);
  return hip_rc(hipGetLastError());
}

int selfc_add_absmax(float* a, const float* b, size_t n, float* amax, void* stream) {
  if (!a || !b || n == 0 || (n & 3)) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  const size_t n4 = n / 4;
  size_t nb = (n4 + 255) / 256;
  if (amax && nb > 256) nb = 256;
  hipLaunchKernelGGL(add_absmax_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (float4*)a, (const float4*)b, n4, (unsigned*)amax);
  return hip_rc(hipGetLastError());
}

int selfc_freq_fwd_bwd(const float* d1, const float* d2, float* dx, int N, int H, int W, void* stream) {
  if (!d1 || !d2 || !dx || N <= 0 || H <= 0 || W <= 0 || (H & 3) || (W & 3)) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  const size_t total = (size_t)N * (H / 4) * (W / 4) * 3;
  hipLaunchKernelGGL(freq_fwd_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d1, d2, dx, N, H, W, 48);
  return hip_rc(hipGetLastError());
}

int selfc_freq_inv_bwd(const float* dout, float* d1, float* d2, int N, int H, int W, void* stream) {
  if (!dout || !d1 || !d2 || N <= 0 || H <= 0 || W <= 0 || (H & 3) || (W & 3)) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  const size_t total = (size_t)N * (H / 4) * (W / 4) * 3;
  hipLaunchKernelGGL(freq_inv_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, d1, d2, N, H, W, 48);
  return hip_rc(hipGetLastError());
}

// ---- building blocks for gradients that are orchestrated from the host side (STP head, selfc_amd/autograd.py) ----
int selfc_bwd_scale(const float* g, size_t n, float* amax, void* stream) {
  if (!g || !amax || n == 0) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  return bwd_absmax(g, n, amax, (hipStream_t)stream);
}

int selfc_bwd_to_planes(const float* x, void* planes, size_t npix, int c, int cs, int lrelu, float sign, const float* amax, void* stream) {
  if (!x || !planes || npix == 0 || c < 1 || cs < c) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  return bwd_to_planes(x, planes, npix, c, cs, (c + 31) / 32, lrelu, sign, amax, (hipStream_t)stream);
}

int selfc_bwd_conv_planes(const void* in, int nplanes_in, int kt, int sp1, const void* w, int ngroups, void* out_planes,
                          const void* add, const void* mask, int mask_z, float* plain, int coutp, int accumulate,
                          const float* amax, int N, int T, int H, int W, void* stream) {
  if (!in || !w || nplanes_in < 1 || (kt != 1 && kt != 3) || ngroups < 1 || N <= 0 || T <= 0 || N % T || H <= 0 || W <= 0) return SELFC_EINVAL;
  if ((out_planes == nullptr) == (plain == nullptr)) return SELFC_EINVAL;
  if (plain && (!amax || coutp < 4 || (coutp & 3))) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  BwdConv c{};
  c.in = in; c.nplanes_in = nplanes_in; c.kt = kt; c.sp1 = sp1; c.w = w; c.ngroups = ngroups; c.out_planes = out_planes;
  c.add = add; c.mask = mask; c.mask_z = mask_z; c.plain = plain; c.coutp = coutp; c.accumulate = accumulate; c.amax = amax;
  return bwd_conv_planes(c, N, T, H, W, (hipStream_t)stream);
}

size_t selfc_bwd_wgrad_scratch_bytes(int N, int H, int W, int Pn, int Qn, int taps) {
  if (N <= 0 || H <= 0 || W <= 0 || Pn < 1 || Qn < 1 || (taps != 1 && taps != 9)) return 0;
  return bwd_wgrad_scratch_bytes(N, H, W, Pn, Qn, taps);
}

int selfc_bwd_wgrad(const void* P, int Pn, const void* Q, int Qn, int taps, float* wout, int O, int Ctot, float* bout, float beta,
                    const float* amax, void* scratch, size_t scratch_bytes, int N, int T, int H, int W, void* stream) {
  if (!P || !Q || !amax || !scratch || Pn < 1 || Qn < 1 || (taps != 1 && taps != 9)) return SELFC_EINVAL;
  if (O < 1 || O > 32 * Pn || Ctot < 1 || Ctot > 32 * Qn || N <= 0 || T <= 0 || N % T || H <= 0 || W <= 0) return SELFC_EINVAL;
  if (scratch_bytes < selfc_bwd_wgrad_scratch_bytes(N, H, W, Pn, Qn, taps)) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  WgradJob j{};
  j.P = P; j.Pn = Pn; j.Q[0] = Q; j.Qn[0] = Qn; j.taps = taps; j.temporal = 0;
  j.wout = wout; j.O = O; j.Ctot = Ctot; j.cin = Ctot; j.nx = Qn; j.bout = bout; j.beta = beta;
  return bwd_wgrad(j, amax, scratch, N, T, H, W, (hipStream_t)stream);
}

}  // extern "C"
