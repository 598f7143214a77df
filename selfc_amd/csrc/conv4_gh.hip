// conv4 of the G / H dense blocks (Subnet_constructor.py:129, cin = 3) as its own persistent launch, for gfx950.
//
// csrc/fused_gh.hip fused to depth 3 (conv1..3, launch_fused_gh3) leaves conv4 = 47 % of the block's MACs.  Run as a
// fourth fused conv it needs the 24x24 input halo and three grown feature regions; run here it reads the 18x18 halos of
// f1..f3 back from L2 (62 KB per tile and net) but computes no halo twice, keeps ALL of its 57 weight fragments resident
// in LDS (no streaming, no chunk barriers) and has two barriers per tile.
//
// One 512-thread workgroup owns ONE spatial 16x16 tile and walks frames (per-workgroup constant halo geometry, as in
// csrc/fused_f.hip).  Wave w = output rows 2w, 2w+1 (one 32-pixel M-tile); K = [im2col48 of y1: 3 steps][f1, f2, f3:
// 18 (tap, k-step) steps each]; operands through a 3-deep register ring.  The next frame's halo is prefetched into
// registers under the MFMA steps and written to the (single) LDS image between two barriers.
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"
#include "prof.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace selfc {

struct G4Args {
  const float* x1;          // [N][H][W][4] fp32 (y1)
  const f16* w[2];          // conv4 fragments of the fused stream (packing.py: pack_fused_gh, fragments 63..119)
  const float* bias[2];     // 32 floats
  f16* dense[2];            // plane-blocked [4][N][H][W][32]: planes 0..2 = f1..f3 (read), plane 3 = f4 (written)
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
  size_t plane;
};

namespace {

constexpr int TS = 16, HS = 18;                   // tile side, halo side
constexpr int NWAVE = 8, NTHR = NWAVE * 64;
constexpr int PF = 1536;                          // feature image row pitch: >= HS * PS, a multiple of 256 (2-row M-tiles)
constexpr int F_BYTES = HS * PF;                  // 27,648
constexpr int XPITCH = HS * 8;
constexpr int X_BYTES = HS * XPITCH;              // 2,592
constexpr int NFRAG = 57;
constexpr int OFF_F = 0, OFF_X = 3 * F_BYTES, OFF_W = OFF_X + ((X_BYTES + 255) & ~255), OFF_B = OFF_W + NFRAG * 1024;
constexpr int G4_LDS = OFF_B + 128;
static_assert(G4_LDS <= 160 * 1024, "LDS budget");
constexpr int NPIECE = HS * HS * 4;               // 16-byte pieces of one feature halo
constexpr int FITER = (NPIECE + NTHR - 1) / NTHR; // 3

__global__ __launch_bounds__(NTHR) void conv4_gh_kernel(const G4Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  const int net = blockIdx.y;
  const int stile = blockIdx.x % a.ntiles, f0 = blockIdx.x / a.ntiles, gf = gridDim.x / a.ntiles;
  if (f0 >= a.N) return;
  const int ty0 = (stile / a.tiles_x) * TS, tx0 = (stile % a.tiles_x) * TS;
  const int py = 2 * wave + ((lane >> 4) & 1), px = lane & 15;
  const size_t fpix = (size_t)a.H * a.W;
  f16* const dense = net ? a.dense[1] : a.dense[0];

  // ---- per-workgroup constant halo geometry: feature pieces (map: piece i & 3 of halo pixel i >> 2) and the y1 pixel
  unsigned goff[FITER], loff[FITER], fok = 0;
#pragma unroll
  for (int it = 0; it < FITER; ++it) {
    const int i = tid + it * NTHR;
    const int p = min(i >> 2, HS * HS - 1), hy = p / HS, hx = p - hy * HS;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool ok = (i < NPIECE) & (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
    const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
    goff[it] = (unsigned)(yc * a.W + xc) * 64u + (unsigned)(i & 3) * 16u;
    loff[it] = (unsigned)(hy * PF + hx * PS + (i & 3) * 16);
    fok |= (ok ? 1u : 0u) << it;
  }
  unsigned xgo = 0;
  bool xok = false;
  {
    const int p = min(tid, HS * HS - 1), hy = p / HS, hx = p - hy * HS;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    xok = (tid < HS * HS) & (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
    xgo = (unsigned)(min(max(y, 0), a.H - 1) * a.W + min(max(x, 0), a.W - 1)) * 16u;
  }
  u32x4 fv[3][FITER];
  float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
  auto h_load = [&](const int n) __attribute__((always_inline)) {
    const char* fr = reinterpret_cast<const char*>(dense) + (size_t)n * fpix * 64;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int it = 0; it < FITER; ++it) fv[j][it] = *reinterpret_cast<const u32x4*>(fr + (size_t)j * a.plane * 2 + goff[it]);
    xv = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.x1) + (size_t)n * fpix * 16 + xgo);
  };
  auto h_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int it = 0; it < FITER; ++it)
        if ((fok >> it) & 1u) *reinterpret_cast<u32x4*>(smem + OFF_F + j * F_BYTES + loff[it]) = fv[j][it];
    if (xok) {
      uint2 u;
      u.x = pack2(xv.x, xv.y);
      u.y = pack2(xv.z, 0.f);
      *reinterpret_cast<uint2*>(smem + OFF_X + tid * 8) = u;
    }
  };

  // ---- prologue: zero the images once (pieces outside the frame are never written), weights, bias, first halo
  for (int i = tid; i < OFF_W / 16; i += NTHR) *reinterpret_cast<u32x4*>(smem + i * 16) = u32x4{0u, 0u, 0u, 0u};
  h_load(f0);
  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(net ? a.w[1] : a.w[0]);
    for (int i = tid; i < NFRAG * 64; i += NTHR) *reinterpret_cast<u32x4*>(smem + OFF_W + i * 16) = wsrc[i];
    if (tid < 32) reinterpret_cast<float*>(smem + OFF_B)[tid] = (net ? a.bias[1] : a.bias[0])[tid];
  }
  __syncthreads();
  h_store();
  __syncthreads();

  // per-lane operand bases: tap (dy, dx) of output pixel (py, px) reads halo pixel (py + dy, px + dx)
  const unsigned char* const wl = smem + OFF_W + lane * 16;
  const unsigned char* const fb = smem + OFF_F + py * PF + px * PS + half * 16;
  const unsigned char* const xb = smem + OFF_X + (py * HS + px) * 8;
  // im2col48 k-step ks: K = 12 taps x (c0 c1 c2 0); this lane's 8 K-entries are the two taps 4 ks + 2 half + {0, 1}
  int xo[3][2];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int t0 = 4 * ks + e, t1 = 4 * ks + 2 + e;
      const int o0 = t0 < 9 ? ((t0 / 3) * HS + t0 % 3) * 8 : -1;
      const int o1 = t1 < 9 ? ((t1 / 3) * HS + t1 % 3) * 8 : -1;
      xo[ks][e] = half ? o1 : o0;
    }
  const int y = ty0 + py, x = tx0 + px;
  const bool inimg = (y < a.H) & (x < a.W);
  const unsigned ooff = (unsigned)((y * a.W + x) * 32 + 8 * half);

  for (int n = f0; n < a.N; n += gf) {
    const bool more = n + gf < a.N;
    if (more) h_load(n + gf);                     // lands while this tile computes
    f32x16 acc;
    {
      const float* bl = reinterpret_cast<const float*>(smem + OFF_B) + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bl + 8 * g);
        acc[4 * g + 0] = b.x; acc[4 * g + 1] = b.y; acc[4 * g + 2] = b.z; acc[4 * g + 3] = b.w;
      }
    }
    f16x8 rA[3], rB[3];
    auto load_step = [&](const int st) __attribute__((always_inline)) {
      rA[st % 3] = *reinterpret_cast<const f16x8*>(wl + st * 1024);
      if (st < 3) {
        uint2 p0 = make_uint2(0u, 0u), p1 = make_uint2(0u, 0u);
        if (xo[st][0] >= 0) p0 = *reinterpret_cast<const uint2*>(xb + xo[st][0]);
        if (xo[st][1] >= 0) p1 = *reinterpret_cast<const uint2*>(xb + xo[st][1]);
        const u32x4 u = {p0.x, p0.y, p1.x, p1.y};
        rB[st % 3] = __builtin_bit_cast(f16x8, u);
      } else {
        const int fs = st - 3, j = fs / 18, f = fs - 18 * j, tap = f >> 1, ks = f & 1;
        rB[st % 3] = *reinterpret_cast<const f16x8*>(fb + j * F_BYTES + (tap / 3) * PF + (tap % 3) * PS + ks * 32);
      }
    };
    load_step(0);
    load_step(1);
#pragma unroll
    for (int st = 0; st < NFRAG; ++st) {
      if (st + 2 < NFRAG) load_step(st + 2);
      __builtin_amdgcn_sched_barrier(0);
      acc = mfma_32x32x16(rA[st % 3], rB[st % 3], acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                              // every wave is done with the images
    if (more) h_store();
    // epilogue: LeakyReLU, f16, half swap, two 16-byte stores per lane into plane 3
    {
      uint32_t r[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        r[g][0] = pack2(lrelu02(acc[4 * g + 0]), lrelu02(acc[4 * g + 1]));
        r[g][1] = pack2(lrelu02(acc[4 * g + 2]), lrelu02(acc[4 * g + 3]));
      }
      f16* d = dense + (size_t)3 * a.plane + (size_t)n * fpix * 32 + ooff;
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        u32x4 v;
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
          const auto sw = __builtin_amdgcn_permlane32_swap(r[2 * gp][dd], r[2 * gp + 1][dd], false, false);
          v[dd] = sw[0];
          v[2 + dd] = sw[1];
        }
        if (inimg) *reinterpret_cast<u32x4*>(d + 16 * gp) = v;
      }
    }
    __syncthreads();                              // the next tile's images are complete
  }
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

}  // namespace

int launch_conv4_gh(G4Args& a, hipStream_t s) {
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv4_gh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G4_LDS);
    if (e != hipSuccess) return hip_rc(e);
    attr_done = true;
  }
  a.tiles_x = (a.W + TS - 1) / TS;
  a.tiles_y = (a.H + TS - 1) / TS;
  a.ntiles = a.tiles_x * a.tiles_y;
  a.plane = (size_t)a.N * a.H * a.W * 32;
  static const int maxwg = getenv("SELFC_CONV4GH_MAXWG") ? atoi(getenv("SELFC_CONV4GH_MAXWG")) : 128;
  static const int minrounds = getenv("SELFC_CONV4GH_MINROUNDS") ? atoi(getenv("SELFC_CONV4GH_MINROUNDS")) : 2;
  const int gmax = a.ntiles >= maxwg ? 1 : (maxwg / a.ntiles < a.N ? maxwg / a.ntiles : a.N);
  int rounds = (a.N + gmax - 1) / gmax;
  if (rounds < minrounds) rounds = a.N < minrounds ? a.N : minrounds;
  const int gfr = (a.N + rounds - 1) / rounds;
  ProfScope prof(PROF_FUSED_GH, s);
  hipLaunchKernelGGL(conv4_gh_kernel, dim3((unsigned)(gfr * a.ntiles), 2), dim3(NTHR), G4_LDS, s, a);
  return hip_rc(hipGetLastError());
}

}  // namespace selfc
