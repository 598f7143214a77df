// The data-gradient chain of one dense block as ONE launch (backward of Subnet_constructor.py:27-30 / :126-129, the four 3x3
// convs with their concats and LeakyReLUs), for gfx950.
//
// csrc/backward.hip runs the chain layer by layer on the generic plane conv (csrc/dense_conv.hip, EPI_BWD):
//     dpre3 = m3 (t3 + W4^T[f3] dpre4)
//     dpre2 = m2 (t2 + W4^T[f2] dpre4 + W3^T[f2] dpre3)
//     dpre1 = m1 (t1 + W4^T[f1] dpre4 + W3^T[f1] dpre3 + W2^T[f1] dpre2)
//     dx    =     tx + W4^T[x ] dpre4 + W3^T[x ] dpre3 + W2^T[x ] dpre2 + W1^T[x] dpre1
// (t = conv5^T(dOut), m = LeakyReLU'(saved feature)) - four or five launches of ~4 us of MFMA work each on the 36x36 latents of a
// training crop, in series, 32 subnets in series per step.  Here one workgroup owns a 12x16 output tile of one frame and keeps
// the chain in LDS (halo recompute: dpre_j on the tile grown by j pixels; dpre4 comes in with a 4-pixel halo), the same
// decomposition as csrc/fused_gh.hip forward.  Same fragments (packing.pack_subnet_bwd), same stage / tap / k order and the same
// MFMA as the layer-wise path, so every value it stores is BIT-IDENTICAL to that path's (tests/test_gpu_train.py).
//
// LDS: images G4 (20x24), G3 (18x22), G2 (16x20), G1 (14x18) of 32 f16 channels at the conv kernels' 80-byte pixel pitch and
// 256-byte-multiple row pitches (40 + 31.5 + 28 + 21 KiB) + one 18-KiB weight stage = 138.5 KiB: one workgroup of 8 waves per CU.
// M-tiles are 2 rows x 16 columns (dense_conv.hip's): a region 22 / 20 / 18 columns wide takes two column blocks, the lanes of
// the second block beyond the region multiply whatever the image row holds there (their results are not stored; an MFMA's
// output column depends on its own input column only).
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include "common.hpp"
#include "prof.hpp"
#include "bwd_internal.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace selfc {

struct DgArgs {
  const f16* g4;          // dpre4: plane 0 of the chain buffer [N][H][W][32]
  f16* gb;                // the chain buffer: dpre_j is written to plane 4 - j
  const f16* add;         // conv5^T(dOut): planes [nx x-groups][f1][f2][f3]
  const f16* feat;        // saved features f1, f2, f3 (the masks)
  const f16* wtd[3];      // fragments of dpre3, dpre2, dpre1 (stages: dpre4, dpre3, ..; 18 fragments each)
  const f16* wtx;         // fragments of dx: nx groups x 4 stages x 18
  float* dx;              // fp32 NHWC rows of stride cinp, or null (no input gradient wanted)
  const float* amax;      // max|dOut| of the call: dx is stored as value / grad_scale(*amax)
  float* amax_out;        // optional: atomic max |stored dx|
  size_t plane;           // halfs per plane
  int N, H, W, tiles_x, tiles_y;
  int nx, cinp, acc_dx;
#ifdef SELFC_DEV
  int ablate;                   // dev build, timing only: 1024 = no epilogue operand loads
  unsigned long long* stamps;   // dev build: 8 x u64 per workgroup (100-MHz clock): entry, dpre4 staged, after dpre3 / dpre2 / dpre1, end, stage count
#endif
};

namespace {

#ifdef SELFC_DEV
#define DGSTAMP(i) do { if (a.stamps && threadIdx.x == 0 && blockIdx.x < 512) a.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DGSTAMP(i) do { } while (0)
#endif

constexpr int TW = 16;
constexpr int NWAVE = 8, NT = NWAVE * 64;
// TH = tile rows: 12 (three tiles down a 36-row training latent), or 6 where that still fits one round - twice the workgroups,
// each with less serial work (launches of 63 / 126 tile-frames at 12 rows are all latency)
template <int TH, int K>
struct Img {
  static constexpr int rows = TH + 2 * K, cols = TW + 2 * K;
  static constexpr int rowb = (cols * PS + 255) / 256 * 256;
  static constexpr int bytes = rows * rowb;
};
template <int TH, int K> constexpr int img_off() {
  return K == 4 ? 0 : K == 3 ? Img<TH, 4>::bytes : K == 2 ? Img<TH, 4>::bytes + Img<TH, 3>::bytes : Img<TH, 4>::bytes + Img<TH, 3>::bytes + Img<TH, 2>::bytes;
}
template <int TH> constexpr int off_w() { return img_off<TH, 1>() + Img<TH, 1>::bytes; }
template <int TH> constexpr int dg_lds() { return off_w<TH>() + 2 * 18 * 1024; }      // TWO fragment buffers: stage g lives in buffer g & 1
static_assert(dg_lds<12>() <= 160 * 1024, "LDS budget");
constexpr int WITER = (18 * 64 + NT - 1) / NT;      // 3

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int TH>
__device__ __forceinline__ void dgrad_chain_body(const DgArgs& a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const lw0 = smem + off_w<TH>();
  DGSTAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int wg = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);
  const int tx = wg % a.tiles_x, ty = (wg / a.tiles_x) % a.tiles_y, n = wg / (a.tiles_x * a.tiles_y);
  const int tx0 = tx * TW, ty0 = ty * TH;
  const int H = a.H, W = a.W;

  // flat list of the weight stages: dpre3 (1), dpre2 (2), dpre1 (3), dx (4 per x-group)
  const int nstage_all = 6 + (a.dx ? 4 * a.nx : 0);
  auto wsrc_of = [&](const int g) __attribute__((always_inline)) -> const u32x4* {
    const f16* p;
    if (g < 1) p = a.wtd[0];
    else if (g < 3) p = a.wtd[1] + (size_t)(g - 1) * 18 * 512;
    else if (g < 6) p = a.wtd[2] + (size_t)(g - 3) * 18 * 512;
    else p = a.wtx + (size_t)(g - 6) * 18 * 512;          // group z = (g - 6) / 4 follows group z - 1 (stride 4 x 18 fragments)
    return reinterpret_cast<const u32x4*>(p);
  };
  u32x4 wreg[WITER];
  auto load_w = [&](const int g) __attribute__((always_inline)) {
    const u32x4* __restrict__ src = wsrc_of(g);
#pragma unroll
    for (int it = 0; it < WITER; ++it) wreg[it] = src[min(tid + it * NT, 18 * 64 - 1)];
  };
  // Two LDS fragment buffers: the fragments of stage g + 1 (requested when stage g - 1 started) go to LDS when stage g STARTS, into
  // the buffer stage g - 1 read - behind that stage's closing barrier - so a stage is [fragments g + 1 to LDS, request g + 2, MFMAs
  // of g, ONE barrier] where it used to be [request g + 1, MFMAs, barrier, fragments to LDS, barrier] (r6ab).
  auto store_w = [&](const int g) __attribute__((always_inline)) {
    unsigned char* const lw = lw0 + (g & 1) * (18 * 1024);
#pragma unroll
    for (int it = 0; it < WITER; ++it) {
      const int i = tid + it * NT;
      if (i < 18 * 64) *reinterpret_cast<u32x4*>(lw + i * 16) = wreg[it];
    }
  };

  load_w(0);
  // ---- dpre4 with its 4-pixel halo -> G4 (zero outside the frame: the convs' zero padding)
  {
    constexpr int NITEM = Img<TH, 4>::rows * Img<TH, 4>::cols * 4;
    constexpr int AITER = (NITEM + NT - 1) / NT;
    u32x4 v[AITER];
    bool ok[AITER];
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
      const int i = min(tid + it * NT, NITEM - 1);
      const int p = i >> 2, q = i & 3;
      const int hy = p / Img<TH, 4>::cols, hx = p - hy * Img<TH, 4>::cols;
      const int y = ty0 + hy - 4, x = tx0 + hx - 4;
      ok[it] = (y >= 0) & (y < H) & (x >= 0) & (x < W);
      const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
      v[it] = *reinterpret_cast<const u32x4*>(a.g4 + ((size_t)(n * H + yc) * W + xc) * 32 + q * 8);
    }
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
      const int i = tid + it * NT;
      if (i < NITEM) {
        const int p = i >> 2, q = i & 3;
        const int hy = p / Img<TH, 4>::cols, hx = p - hy * Img<TH, 4>::cols;
        *reinterpret_cast<u32x4*>(smem + hy * Img<TH, 4>::rowb + hx * PS + q * 16) = ok[it] ? v[it] : u32x4{0u, 0u, 0u, 0u};
      }
    }
  }
  store_w(0);
  load_w(1);
  __syncthreads();
  DGSTAMP(1);

  int gstage = 0;        // the stage whose fragments are in LDS
  float bwmax = 0.f;
  bool bwnan = false;

  // one layer: J = 3, 2, 1 -> dpre_J on the tile grown by J; J = 0 -> one 32-channel group of dx on the tile
  auto layer = [&](auto jtag, const int zg) __attribute__((always_inline)) {
    constexpr int J = decltype(jtag)::value;
    constexpr int RH = TH + 2 * J, RW = TW + 2 * J, NBX = (RW + 15) / 16, NMT = (RH / 2) * NBX, MT = (NMT + NWAVE - 1) / NWAVE;
    constexpr int NST = 4 - J;
    f32x16 acc[MT];
    int rr[MT], cc[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = min(wave + NWAVE * mi, NMT - 1);
      const int by = m / NBX, bx = m - by * NBX;
      rr[mi] = 2 * by + ((lane & 31) >> 4);
      cc[mi] = 16 * bx + (lane & 15);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
    }
    // the epilogue's operands (conv5^T addend, saved feature = the mask; for dx the addend and, when accumulating, the old value)
    // are fetched BEFORE the MFMA stages: out-of-frame and out-of-region lanes read pixel 0 and ignore it
    bool inimg[MT];
    size_t pixv[MT];
    f16x4 tadd[MT][4], tmask[J > 0 ? MT : 1][4];
    float4 told[J == 0 ? MT : 1][4];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int y = ty0 + rr[mi] - J, x = tx0 + cc[mi] - J;
      inimg[mi] = (cc[mi] < RW) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
      pixv[mi] = inimg[mi] ? (size_t)(n * H + y) * W + x : 0;
#ifdef SELFC_DEV
      if (a.ablate & 1024) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { tadd[mi][g] = f16x4{}; if constexpr (J > 0) tmask[mi][g] = f16x4{}; else told[mi][g] = make_float4(0.f, 0.f, 0.f, 0.f); }
        continue;
      }
#endif
      const f16* __restrict__ ad = a.add + (size_t)(J > 0 ? a.nx + J - 1 : zg) * a.plane + pixv[mi] * 32 + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) tadd[mi][g] = *reinterpret_cast<const f16x4*>(ad + 8 * g);
      if constexpr (J > 0) {
        const f16* __restrict__ mk = a.feat + (size_t)(J - 1) * a.plane + pixv[mi] * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) tmask[mi][g] = *reinterpret_cast<const f16x4*>(mk + 8 * g);
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int oc = min(32 * zg + 8 * g + 4 * half, a.cinp - 4);
          told[mi][g] = a.acc_dx ? *reinterpret_cast<const float4*>(a.dx + pixv[mi] * a.cinp + oc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    static_for<0, NST>([&](auto stag) __attribute__((always_inline)) {
      constexpr int S = decltype(stag)::value, K = 4 - S, SH = K - J - 1;
      constexpr int ROWB = Img<TH, K>::rowb;
      if (gstage + 1 < nstage_all) store_w(gstage + 1);
      if (gstage + 2 < nstage_all) load_w(gstage + 2);
      const unsigned char* const lw = lw0 + (gstage & 1) * (18 * 1024);
      const unsigned char* const img = smem + img_off<TH, K>();
      // The operands of tap t + 1 are requested BEFORE the MFMAs of tap t, into a second register set (written as it falls out of the
      // loop nest - read, wait, MFMA, with the M-tile test around each pair - hipcc put an s_waitcnt lgkmcnt(0) in front of EVERY
      // MFMA: ~3 us per stage for 0.7 us of matrix time, r6v).  The reads of a wave's unused M-tile slot go to a valid tile and
      // are dropped.
      f16x8 afr[2][2], bfr[2][2][MT];
      auto read_tap = [&](const int tap, f16x8 (&af)[2], f16x8 (&bf)[2][MT]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          af[ks] = *reinterpret_cast<const f16x8*>(lw + (tap * 2 + ks) * 1024 + lane * 16);
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
            bf[ks][mi] = *reinterpret_cast<const f16x8*>(img + (rr[mi] + SH + tap / 3) * ROWB + (cc[mi] + SH + tap % 3) * PS + half * 16 + ks * 32);
        }
      };
      read_tap(0, afr[0], bfr[0]);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) read_tap(tap + 1, afr[(tap + 1) & 1], bfr[(tap + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mi = 0; mi < MT; ++mi)
            if (wave + NWAVE * mi < NMT) acc[mi] = mfma_32x32x16(afr[tap & 1][ks], bfr[tap & 1][ks][mi], acc[mi]);       // wave-uniform
        __builtin_amdgcn_sched_barrier(0);
      }
      ++gstage;
      if constexpr (S + 1 < NST) __syncthreads();      // every wave is done with this stage's fragment buffer, and the next stage's is written
    });
    // ---- epilogue (no barrier in front of it: it reads accumulators and writes G_J, which nobody reads before the barrier behind it -
    // the same barrier closes the layer's last stage)
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      if (wave + NWAVE * mi >= NMT) continue;
      const int r = rr[mi], c = cc[mi];
      if (c >= RW) continue;
      const bool in_img = inimg[mi];
      const size_t pix = pixv[mi];
      float v[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[g][j] = acc[mi][4 * g + j] + (float)tadd[mi][g][j];
      if constexpr (J > 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[g][j] *= ((float)tmask[mi][g][j] > 0.f) ? 1.f : 0.2f;
        const bool centre = in_img & (r >= J) & (r < J + TH) & (c >= J) & (c < J + TW);
        unsigned char* const lp = smem + img_off<TH, (J > 0 ? J : 1)>() + r * Img<TH, (J > 0 ? J : 1)>::rowb + c * PS + 8 * half;
        f16* const gp = a.gb + (size_t)(4 - J) * a.plane + pix * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 u;
          u.x = in_img ? pack2(v[g][0], v[g][1]) : 0u;
          u.y = in_img ? pack2(v[g][2], v[g][3]) : 0u;
          *reinterpret_cast<uint2*>(lp + 16 * g) = u;
          if (centre) *reinterpret_cast<uint2*>(gp + 8 * g) = u;
        }
      } else {
        if (!in_img) continue;
        const float inv = 1.f / grad_scale(*a.amax);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int oc = 32 * zg + 8 * g + 4 * half;
          if (oc < a.cinp) {
            float4* dst = reinterpret_cast<float4*>(a.dx + pix * a.cinp + oc);
            float4 o = make_float4(v[g][0] * inv, v[g][1] * inv, v[g][2] * inv, v[g][3] * inv);
            const float4 old = told[mi][g];
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            *dst = o;
            bwmax = fmaxf(fmaxf(bwmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            bwnan |= (o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (o.w != o.w);
          }
        }
      }
    }
    __syncthreads();
  };

  layer(std::integral_constant<int, 3>{}, 0);
  DGSTAMP(2);
  layer(std::integral_constant<int, 2>{}, 0);
  DGSTAMP(3);
  layer(std::integral_constant<int, 1>{}, 0);
  DGSTAMP(4);
  if (a.dx) {
    for (int z = 0; z < a.nx; ++z) layer(std::integral_constant<int, 0>{}, z);
    if (a.amax_out) {        // one atomic per wave, the convention of dense_conv.hip's EPI_BWD (NaN -> 0x7fc00000)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bwmax = fmaxf(bwmax, __shfl_xor(bwmax, o));
      unsigned* const bits = reinterpret_cast<unsigned*>(a.amax_out);
      if (__any(bwnan)) { if (lane == 0) atomicMax(bits, 0x7fc00000u); }
      else if (lane == 0 && bwmax > 0.f) atomicMax(bits, __float_as_uint(bwmax));
    }
  }
#ifdef SELFC_DEV
  if (a.stamps) { __builtin_amdgcn_s_waitcnt(0); DGSTAMP(5); if (threadIdx.x == 0 && blockIdx.x < 512) { a.stamps[blockIdx.x * 8 + 6] = 100 + nstage_all; } }
#endif
}

template <int TH>
__global__ __launch_bounds__(NT) void dgrad_chain_kernel(const DgArgs a) {
  dgrad_chain_body<TH>(a);
}

// the chains of two nets of one geometry (a G/H pair) in one launch: blockIdx.y picks the argument set
template <int TH>
__global__ __launch_bounds__(NT) void dgrad_chain_pair_kernel(const DgArgs a, const DgArgs b) {
  if (blockIdx.y) dgrad_chain_body<TH>(b);
  else dgrad_chain_body<TH>(a);
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

}  // namespace

// dpre3, dpre2, dpre1 of TWO nets of one geometry in one launch (no input gradient: the pair's dx is one conv over both nets' planes,
// bwd_conv_planes with a second source)
int bwd_dgrad_chain_pair(void* gb0, void* gb1, const void* add0, const void* add1, const void* feat0, const void* feat1,
                         const void* const* wtd0, const void* const* wtd1, int nx, const float* amax, int N, int H, int W, hipStream_t s,
                         const void* wtx0, const void* wtx1, float* dx0, float* dx1, int cinp, int acc0) {
  DgArgs a{}, b{};
  // optional input gradients: net 0 into dx0 (accumulated when acc0), net 1 into dx1 (the caller adds the two: each net's dx layer
  // runs on its own LDS-resident chain - on a small problem that is cheaper than a separate eight-stage conv over both nets' planes)
  a.wtx = (const f16*)wtx0; b.wtx = (const f16*)wtx1; a.dx = dx0; b.dx = dx1; a.cinp = b.cinp = cinp; a.acc_dx = acc0; b.acc_dx = 0;
  a.g4 = (const f16*)gb0; a.gb = (f16*)gb0; a.add = (const f16*)add0; a.feat = (const f16*)feat0;
  b.g4 = (const f16*)gb1; b.gb = (f16*)gb1; b.add = (const f16*)add1; b.feat = (const f16*)feat1;
  for (int i = 0; i < 3; ++i) { a.wtd[i] = (const f16*)wtd0[i]; b.wtd[i] = (const f16*)wtd1[i]; }
  a.amax = b.amax = amax;
  a.plane = b.plane = (size_t)N * H * W * 32;
  a.N = b.N = N; a.H = b.H = H; a.W = b.W = W;
  a.tiles_x = b.tiles_x = (W + TW - 1) / TW;
  a.nx = b.nx = nx;
  static const int th_env = getenv("SELFC_BWD_CHAIN_TH") ? atoi(getenv("SELFC_BWD_CHAIN_TH")) : 0;
  const bool th6 = th_env ? th_env == 6 : 2L * N * a.tiles_x * ((H + 5) / 6) <= 256;
#ifdef SELFC_DEV
  static const int dev_abl = getenv("SELFC_ABLATE") ? atoi(getenv("SELFC_ABLATE")) : 0;
  if (dev_abl & 512) { a.stamps = dev_stamp_slot(); b.stamps = dev_stamp_slot(); }
  a.ablate = b.ablate = dev_abl;
#endif
  static std::atomic<unsigned long long> optin12{0}, optin6{0};
  ProfScope prof(-1, s);
  if (th6) {
    if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&dgrad_chain_pair_kernel<6>), dg_lds<6>(), optin6); e != hipSuccess) return hip_rc(e);
    a.tiles_y = b.tiles_y = (H + 5) / 6;
    hipLaunchKernelGGL(dgrad_chain_pair_kernel<6>, dim3((unsigned)(a.tiles_x * a.tiles_y * N), 2), dim3(NT), dg_lds<6>(), s, a, b);
    return hip_rc(hipGetLastError());
  }
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&dgrad_chain_pair_kernel<12>), dg_lds<12>(), optin12); e != hipSuccess) return hip_rc(e);
  a.tiles_y = b.tiles_y = (H + 11) / 12;
  hipLaunchKernelGGL(dgrad_chain_pair_kernel<12>, dim3((unsigned)(a.tiles_x * a.tiles_y * N), 2), dim3(NT), dg_lds<12>(), s, a, b);
  return hip_rc(hipGetLastError());
}

// dpre3, dpre2, dpre1 (planes 1..3 of gb) and, with dx, the input gradient - steps 3 and 4 of selfc_subnet_bwd_phase_x as one launch
int bwd_dgrad_chain(void* gb, const void* add, const void* feat, const void* const* wtd, const void* wtx, float* dx, int nx, int cinp,
                    int accumulate_dx, const float* amax, float* amax_out, int N, int H, int W, hipStream_t s) {
  DgArgs a{};
  a.g4 = (const f16*)gb; a.gb = (f16*)gb; a.add = (const f16*)add; a.feat = (const f16*)feat;
  for (int i = 0; i < 3; ++i) a.wtd[i] = (const f16*)wtd[i];
  a.wtx = (const f16*)wtx; a.dx = dx; a.amax = amax; a.amax_out = amax_out;
  a.plane = (size_t)N * H * W * 32;
  a.N = N; a.H = H; a.W = W;
  a.tiles_x = (W + TW - 1) / TW;
  a.nx = nx; a.cinp = cinp; a.acc_dx = accumulate_dx;
  static const int th_env = getenv("SELFC_BWD_CHAIN_TH") ? atoi(getenv("SELFC_BWD_CHAIN_TH")) : 0;
  const bool th6 = th_env ? th_env == 6 : (long)N * a.tiles_x * ((H + 5) / 6) <= 256;
#ifdef SELFC_DEV
  static const int dev_abl = getenv("SELFC_ABLATE") ? atoi(getenv("SELFC_ABLATE")) : 0;
  if (dev_abl & 512) a.stamps = dev_stamp_slot();
  a.ablate = dev_abl;
#endif
  static std::atomic<unsigned long long> optin12{0}, optin6{0};
  if (th6) {
    if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&dgrad_chain_kernel<6>), dg_lds<6>(), optin6); e != hipSuccess) return hip_rc(e);
    a.tiles_y = (H + 5) / 6;
    ProfScope prof(-1, s);
    hipLaunchKernelGGL(dgrad_chain_kernel<6>, dim3((unsigned)(a.tiles_x * a.tiles_y * N)), dim3(NT), dg_lds<6>(), s, a);
    return hip_rc(hipGetLastError());
  }
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&dgrad_chain_kernel<12>), dg_lds<12>(), optin12); e != hipSuccess) return hip_rc(e);
  a.tiles_y = (H + 11) / 12;
  ProfScope prof(-1, s);
  hipLaunchKernelGGL(dgrad_chain_kernel<12>, dim3((unsigned)(a.tiles_x * a.tiles_y * N)), dim3(NT), dg_lds<12>(), s, a);
  return hip_rc(hipGetLastError());
}

}  // namespace selfc
