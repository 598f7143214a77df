// Dense-block convolutions + fused affine coupling for gfx950 (MI355X).
//
//   conv3x3_kernel   conv1..4 of DenseBlock / D2DTInput (Subnet_constructor.py:27-30,
//                    126-129) as a direct (im2col-free) convolution on a concat-free
//                    NHWC f16 dense buffer: reads channel ranges ("stages") of the
//                    buffer, appends 32 channels.  With EPI != LRELU it is the 3x3
//                    conv5 of the 2-D DenseBlock (:31) with the coupling of
//                    Inv_arch.py:25-31 fused into the epilogue.
//   tconv5_kernel    conv5 of D2DTInput (:106,130): 3-tap temporal conv, frames of a
//                    clip walked sequentially so every dense feature is read once;
//                    coupling fused into the epilogue.
//
// MFMA orientation: D[outch][pixel] = sum_k W[outch][k] * act[k][pixel], i.e. the
// packed weights are the A operand and the activations the B operand.  A lane
// then owns one pixel and 4-channel groups of it, so epilogue loads/stores are
// 8/16-byte vectors on NHWC rows.  Operands are f16 (weights and activations
// rounded once), accumulation and all coupling arithmetic are fp32.
#include <stdlib.h>
#include "common.hpp"
#include "prof.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace selfc {
// csrc/fused_gh.hip
struct FGArgs {
  const float* x1;
  const f16* w[2];
  const float* bias[2][4];
  f16* dense[2];
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
  size_t plane;
  int ablate;
  unsigned long long* stamps;
};
int launch_fused_gh(FGArgs& a, hipStream_t s, int nets);
// csrc/fused_f.hip
int launch_fused_f(void* dense, const void* w, const float* const* bias, int N, int H, int W, hipStream_t s,
                   const void* w5p, float* pf, const float* b5, const float* x1, float* x1out, int T, int rev, int keep_features);

}  // namespace selfc
#include "bwd_internal.hpp"

namespace {

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

// Developer timing aid: with -DSELFC_DEV the env var SELFC_ABLATE switches parts of conv3x3_kernel off
// (1 MFMA loop, 2 global loads, 4 LDS staging, 8 epilogue stores, 16 im2col fill, 32 weight loads, 64 activation
// loads) - results are then wrong, only the timing is meaningful.  Compiled out of the production library.
#ifdef SELFC_DEV
#define ABL(a, bit) ((a).ablate & (bit))
#else
#define ABL(a, bit) false
#endif

enum { EPI_LRELU = 0, EPI_PLAIN = 1, EPI_F = 2, EPI_GH = 3, EPI_BWD = 4, EPI_T5B = 5 };

struct C3Stage {
  int coff;   // first channel of the stage in the dense buffer
  int width;  // 16 or 32 channels
  int kind;   // 0: 9-tap stage of the dense buffer; 1: im2col stage built from x1 (K = 9*c1 <= 32); 2: generic mode, a plane of the
              // SECOND source (C3Args::gen_split)
  int dt;     // generic mode: frame offset of a temporal tap (-1, 0, +1), 0 otherwise
};

// The stage list of a conv is regular, so it is described by four integers and
// decoded with scalar arithmetic (a by-value array indexed by the stage loop
// would be spilled to scratch):
//   has_im2col : stage 0 is the im2col stage, stages s >= 1 are features s-1
//   otherwise  : stages 0..n_in-1 cover the cin16 input channels in 32-wide
//                pieces (last one 16 wide when cin16 % 32 == 16), followed by
//                features at channel fbase + 32*i.
struct C3Args {
  const f16* dense[2];   // per net (blockIdx.z, or the G/H loop of EPI_GH)
  const f16* w[2];       // packed A fragments
  const float* bias[2];  // 32 floats
  const float* x1;       // NHWC4 fp32 source of the im2col stage
  f16* out[2];           // EPI_LRELU: dense buffer to append to
  int N, H, W;           // frames, latent size
  size_t plane;          // halfs per 32-channel plane of a dense buffer = N*H*W*32
  int c1;                // channels of the im2col source
  int nstages;
  int out_coff;
  int tiles_x, tiles_y;
  int has_im2col, cin16, n_in, fbase;
  // generic mode (gen_planes > 0): the conv reads planes [0, gen_planes) of dense[0], optionally at
  // three temporal taps (gen_tt == 3: frames n-1, n, n+1 inside a clip of T frames, zero outside), and
  // blockIdx.z selects a 32-channel output group (weights w[0] + z*wz_stride, bias[0] + 32 z).
  int gen_planes, gen_tt, T;
  size_t wz_stride;
  int gen_sp1;           // generic mode: centre spatial tap only ((kt,1,1) kernels: 2 fragments per stage)
  // generic mode, two sources (the input gradient of a G/H PAIR in one launch, csrc/backward.hip): stages [0, gen_split) read planes
  // 0.. of dense[0] with the fragments of w[0], stages [gen_split, nstages) planes 0.. of dense[1] with w[1] (+ z*wz_stride2);
  // gen_split == 0: one source.  No temporal taps in this mode.
  int gen_split;
  size_t wz_stride2;
  // EPI_BWD (generic mode, csrc/backward.hip): v = acc + add[z] (f16 plane, optional); group z == bw_mask_z is
  // multiplied by LeakyReLU'(bw_mask) (1 where the saved feature is > 0, else 0.2; bw_mask_z -2: every group against its own
  // mask plane, -3: the same with ReLU' = 0 where the feature is 0).  Output: f16 plane
  // out[0] + (out_coff/32 + z) planes (group bw_mask_z goes to bw_alt when that is set), or, with `plain`,
  // fp32 NHWC rows of stride coutp scaled by 1/grad_scale(*bw_amax), added to the old value when bw_acc.
  const f16* bw_add;
  const f16* bw_add2;      // second addend planes (the pair's other net), or null
  const f16* bw_mask;
  f16* bw_alt;
  const float* bw_amax;
  float* bw_amax_out;      // EPI_BWD plain output: atomic max |stored value| (float bits; NaN -> 0x7fc00000, see backward.hip absmax_kernel)
  int bw_mask_z, bw_acc;
  // coupling / plain epilogue (EPI != LRELU)
  float* x1io;           // EPI_F: y1 = x1 +- F: read here ...     [N][H][W][4]
  float* x2io;           // EPI_GH: x2 of the coupling: read here  [N][H][W][c2p]
  float* x1out, *x2out;  // ... and written here (the same buffers unless selfc_latent.x1_out / x2_out are set)
  f16* fd;               // EPI_GH: f16 copy of y2 into planes 0.. of the F dense buffer, or null
  float* s_out;          // EPI_GH: optional s                      [N][H][W][c2p]
  float* plain;          // EPI_PLAIN: fp32 NHWC output, stride coutp
  int c2p, coutp, rev;
  float clamp;
  int ablate;            // -DSELFC_DEV only (see ABL above)
#ifdef SELFC_DEV
  unsigned long long* stamps;   // 8 x u64 per workgroup of this launch (100-MHz clock): entry, stage 0 staged, stages done, rows ready, stores issued, stores done
#endif
};

// ---------------------------------------------------------------------------------
// conv3x3: workgroup = TH x TW output pixels of one frame, NW waves, MT 32-pixel
// M-tiles (2 rows x 16 cols) per wave.  Per stage: the (TH+2)x(TW+2) halo tile of
// <=32 channels and the stage's weight fragments are staged in LDS (register
// prefetch of stage s+1 is issued before the MFMAs of stage s).
// ---------------------------------------------------------------------------------
template <bool GEN>
__device__ __forceinline__ C3Stage stage_of(const C3Args& a, const int s) {
  if (GEN) {
    if (a.gen_split && s >= a.gen_split) return C3Stage{32 * (s - a.gen_split), 32, 2, 0};
    const int ti = s / a.gen_planes;
    return C3Stage{32 * (s - ti * a.gen_planes), 32, 0, a.gen_tt == 3 ? ti - 1 : 0};
  }
  if (a.has_im2col) return s == 0 ? C3Stage{0, 32, 1, 0} : C3Stage{a.fbase + 32 * (s - 1), 32, 0, 0};
  if (s < a.n_in) return C3Stage{32 * s, (a.cin16 - 32 * s >= 32) ? 32 : 16, 0, 0};
  return C3Stage{a.fbase + 32 * (s - a.n_in), 32, 0, 0};
}

#ifdef SELFC_DEV
#define C3STAMP(i) do { if (a.stamps && threadIdx.x == 0 && blockIdx.x < 512 && blockIdx.z == 0) a.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define C3STAMP(i) do { } while (0)
#endif

// GEN: generic plane-list mode (temporal taps, output groups) as a template parameter so that the hot
// non-generic instantiations keep their register budget (as a runtime flag it cost 21 VGPRs = one wave/SIMD).
// ROWEPI (EPI_BWD only): the row epilogue of conv3x3_body.hpp - 165 + 62 registers = 2 waves per SIMD, chosen by the launcher while
// the launch's workgroups fit the chip two per CU (the training crops); larger problems keep the 132-register form (3 per SIMD).
template <int TH, int TW, int NW, int MT, int EPI, bool GEN, bool ROWEPI = false>
__global__ __launch_bounds__(NW * 64) void conv3x3_kernel(const C3Args a) {
#include "conv3x3_body.hpp"
}

// The same body as a device function, for the pair kernel only (136 VGPRs = 4 workgroups per CU this way; included twice behind a
// reference alias it took 137).  ROWEPI off: the row epilogue's registers would take the pair from 4 to 2 workgroups per CU (r6r).
template <int TH, int TW, int NW, int MT, int EPI, bool GEN>
__device__ __forceinline__ void conv3x3_pair_body(const C3Args& a) {
  constexpr bool ROWEPI = false;
#include "conv3x3_body.hpp"
}

// Two independent convs of the same geometry in ONE launch (blockIdx.y picks the argument set): the data-gradient layers of a G/H
// pair (csrc/backward.hip).  On a training crop one net's launch is half a round of workgroups, and the two nets' launches used to
// run side by side on two streams - with a cross-queue dependency (~10 us each on this runtime) at the fork and at the join.
template <int TH, int TW, int NW, int MT, int EPI, bool GEN>
__global__ __launch_bounds__(NW * 64) void conv3x3_pair_kernel(const C3Args a0, const C3Args a1) {
  if (blockIdx.y) conv3x3_pair_body<TH, TW, NW, MT, EPI, GEN>(a1);
  else conv3x3_pair_body<TH, TW, NW, MT, EPI, GEN>(a0);
}

// ---------------------------------------------------------------------------------
// tconv5: out[t] = W0 d[t-1] + W1 d[t] + W2 d[t+1] (+bias), zero outside the clip.
// Workgroup = 8 waves x 16 pixels of one clip; all weight fragments live in LDS for
// the lifetime of the workgroup; each wave walks t = 0..T-1, loads d[t] once as MFMA
// B fragments straight from global memory (no cross-wave reuse -> no LDS), and adds
// its three contributions to rotating accumulators for out[t+1], out[t], out[t-1].
// MFMA 16x16x32: A = W[16 outch][32 k] (LDS), B = d[32 k][16 px].
// ---------------------------------------------------------------------------------
struct T5Args {
  const f16* dense[2];
  const f16* w;           // fragments [3 taps][NETS][KS][OT][64 lanes][8]
  const float* bias[2];   // OT*16 floats (zero padded)
  const float* x1;        // HASX: NHWC4 fp32, channels 0..2 are the first three K entries
  int B, T, HW;           // clips, frames per clip, pixels per frame
  size_t plane;           // halfs per 32-channel plane = B*T*HW*32
  float* x1io;
  float* x2io;
  float* x1out, *x2out;
  f16* fd;
  float* s_out;
  float* plain;
  int c2p, coutp, rev;
  float clamp;
  // EPI_T5B (csrc/backward.hip: conv5^T of a temporal dense block): blockIdx.y = 32-channel output plane, weights
  // w + y*wz_stride; f16 planes out (no bias); plane mask_z is multiplied by LeakyReLU'(maskp) and stored to altp
  f16* outp;
  const f16* maskp;
  f16* altp;
  int mask_z;
  size_t wz_stride;
  // EPI_T5B, two nets in one launch (the G/H pair of csrc/backward.hip): blockIdx.y in [pair_planes, 2 pair_planes) is output plane
  // y - pair_planes of the SECOND net: input dense[1], weights w2, outputs outp2 / altp2, mask maskp2.  0: one net.
  int pair_planes;
  const f16* w2;
  f16* outp2;
  const f16* maskp2;
  f16* altp2;
};

// Waves per workgroup.  Measured and dropped: 16 waves for the HBM-bound F conv5 (same bytes in flight on half the CUs,
// leaving the rest to the MFMA-bound kernels of other streams): 0.29 -> 0.33 ms per step, headline -0.5 %.
template <int EPI> constexpr int t5_waves() { return 8; }

// FP ("frame-parallel", small problems only): a workgroup owns 16 pixels of one clip and wave w computes output frame t = w from the
// three input frames t-1, t, t+1 (loaded by that wave) instead of eight pixel groups each walking the clip's frames in sequence - 3x the
// loads, no serial walk: on one 36x36 training septuplet the walk is 11 workgroups x 7 dependent steps (27 us), this is 81 x 1.  Every
// output accumulates its taps in the walk's order (tap 0 on frame t-1, tap 1 on t, tap 2 on t+1; k-steps ascending): bit-identical.
template <int NETS, int OT, int KD, int HASX, int EPI, bool FP = false>
__global__ __launch_bounds__(t5_waves<EPI>() * 64) void tconv5_kernel(const T5Args a) {
  constexpr int KS = KD + HASX;
  constexpr int NTH = t5_waves<EPI>() * 64, PXWG = FP ? 16 : t5_waves<EPI>() * 16;
  constexpr int NFRAG = 3 * NETS * KS * OT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {
    // every load of a thread's share is issued before the first LDS store: as a rolled loop (load, wait, store, branch) the
    // up-to-twelve 16-byte pieces per thread were twelve serial L2 round trips at the head of EVERY workgroup of every launch
    const bool net2w = EPI == EPI_T5B && a.pair_planes && (int)blockIdx.y >= a.pair_planes;
    const uint4* __restrict__ wsrc = reinterpret_cast<const uint4*>((net2w ? a.w2 : a.w) +
                                                                     (EPI == EPI_T5B ? (size_t)(blockIdx.y - (net2w ? a.pair_planes : 0)) * a.wz_stride : 0));
    constexpr int WIT = (NFRAG * 64 + NTH - 1) / NTH;
    uint4 wv[WIT];
#pragma unroll
    for (int it = 0; it < WIT; ++it) wv[it] = wsrc[min(tid + it * NTH, NFRAG * 64 - 1)];
#pragma unroll
    for (int it = 0; it < WIT; ++it) {
      const int i = tid + it * NTH;
      if (i < NFRAG * 64) *reinterpret_cast<uint4*>(smem + (size_t)i * 16) = wv[it];
    }
  }
  __syncthreads();

  const int tiles = (a.HW + PXWG - 1) / PXWG;
  const int wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int b = wg / tiles;
  const int p0 = (wg % tiles) * PXWG + (FP ? 0 : wave * 16);
  const int pl = p0 + (lane & 15);
  const bool pvalid = pl < a.HW;
  const int pc = pvalid ? pl : a.HW - 1;   // clamp loads of masked lanes to a valid pixel
  const int kq = lane >> 4;                // this lane's 8-channel group of a 32-wide k-step
  if (p0 >= a.HW) return;                  // whole wave outside (no barriers below)
  const bool net2 = EPI == EPI_T5B && a.pair_planes && (int)blockIdx.y >= a.pair_planes;     // workgroup-uniform
  if (FP && wave >= a.T) return;           // frame-parallel: one wave per frame of the clip (T <= 8; no barriers below)

  f32x4 accp[NETS][OT], accc[NETS][OT], accn[NETS][OT];
#pragma unroll
  for (int q = 0; q < NETS; ++q)
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      accp[q][o] = f32x4{0.f, 0.f, 0.f, 0.f};
      accc[q][o] = f32x4{0.f, 0.f, 0.f, 0.f};
      accn[q][o] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

  u32x4 bcur[NETS][KD], bnxt[NETS][KD];
  float4 xcur = make_float4(0.f, 0.f, 0.f, 0.f), xnxt = xcur;

  auto load_frame = [&](u32x4 (&dst)[NETS][KD], float4& xd, const int t) __attribute__((always_inline)) {
    const size_t pix = (size_t)(b * a.T + t) * a.HW + pc;
#pragma unroll
    for (int q = 0; q < NETS; ++q)
#pragma unroll
      for (int ks = 0; ks < KD; ++ks)
        dst[q][ks] = *reinterpret_cast<const u32x4*>(((q || net2) ? a.dense[1] : a.dense[0]) + (size_t)ks * a.plane + pix * 32 + kq * 8);
    if (HASX) xd = *reinterpret_cast<const float4*>(a.x1 + pix * 4);
  };

  // biases are fetched once, and the x2 rows an epilogue needs are prefetched before the MFMA phase of
  // the frame in which it runs: no global load sits in front of its consumer inside the epilogue
  float4 ebias[NETS][OT];
#pragma unroll
  for (int q = 0; q < NETS; ++q)
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const int oc = o * 16 + kq * 4;
      ebias[q][o] = EPI == EPI_T5B ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>((q ? a.bias[1] : a.bias[0]) + oc);
    }
  float4 xrow[OT];
  auto prefetch_x2 = [&](const int t) __attribute__((always_inline)) {
    if (EPI != EPI_GH) return;
    const size_t pix = (size_t)(b * a.T + t) * a.HW + pc;
#pragma unroll
    for (int o = 0; o < OT; ++o) {
      const int oc = min(o * 16 + kq * 4, a.c2p - 4);
      xrow[o] = *reinterpret_cast<const float4*>(a.x2io + pix * a.c2p + oc);
    }
  };

  auto epilogue = [&](const int t, f32x4 (&acc)[NETS][OT]) __attribute__((always_inline)) {
    if (!pvalid) return;
    const size_t pix = (size_t)(b * a.T + t) * a.HW + pl;
    if (EPI == EPI_T5B) {
      const int zg = (int)blockIdx.y - (net2 ? a.pair_planes : 0);
      const f16* __restrict__ mkp = net2 ? a.maskp2 : a.maskp;
      f16* const altp = net2 ? a.altp2 : a.altp;
      const bool masked = mkp != nullptr && zg == a.mask_z;
      f16* __restrict__ dst = (masked && altp ? altp : (net2 ? a.outp2 : a.outp) + (size_t)zg * a.plane) + pix * 32 + kq * 4;
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        float v[4] = {acc[0][o][0], acc[0][o][1], acc[0][o][2], acc[0][o][3]};
        if (masked) {
          const f16x4 m = *reinterpret_cast<const f16x4*>(mkp + pix * 32 + o * 16 + kq * 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] *= ((float)m[j] > 0.f) ? 1.f : 0.2f;
        }
        uint2 u;
        u.x = pack2(v[0], v[1]);
        u.y = pack2(v[2], v[3]);
        *reinterpret_cast<uint2*>(dst + o * 16) = u;
      }
    } else if (EPI == EPI_PLAIN) {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        const int oc = o * 16 + kq * 4;
        if (oc < a.coutp) {
          const float4 bb = ebias[0][o];
          *reinterpret_cast<float4*>(a.plain + pix * a.coutp + oc) =
              make_float4(acc[0][o][0] + bb.x, acc[0][o][1] + bb.y, acc[0][o][2] + bb.z, acc[0][o][3] + bb.w);
        }
      }
    } else if (EPI == EPI_F) {
      if (kq == 0) {
        const float4 bb = ebias[0][0];
        float4 v = *reinterpret_cast<float4*>(a.x1io + pix * 4);
        const float sgn = a.rev ? -1.f : 1.f;
        v.x += sgn * (acc[0][0][0] + bb.x);
        v.y += sgn * (acc[0][0][1] + bb.y);
        v.z += sgn * (acc[0][0][2] + bb.z);
        v.w += sgn * (acc[0][0][3] + bb.w);
        *reinterpret_cast<float4*>(a.x1out + pix * 4) = v;
      }
    } else {
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        const int oc = o * 16 + kq * 4;
        if (oc < a.c2p) {
          const float4 bg = ebias[0][o];
          const float4 bh = ebias[NETS - 1][o];
          const float4 xv = xrow[o];
          const float xin[4] = {xv.x, xv.y, xv.z, xv.w};
          const float gb[4] = {bg.x, bg.y, bg.z, bg.w}, hb[4] = {bh.x, bh.y, bh.z, bh.w};
          // s = clamp (2 sigmoid(h) - 1); y2 = x2 e^s + g  /  (x2 - g) e^-s  (Inv_arch.py:26-27,29-30) on the hardware
          // exp2 / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp; the libm expf + IEEE divisions were ~45 VALU instructions
          // per element and a third of this kernel's time); rev is folded into the exponent's sign, no division.
          float yo[4], so[4];
          const float esgn = a.rev ? -1.44269504f : 1.44269504f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float gv = acc[0][o][j] + gb[j];
            const float hv = acc[NETS - 1][o][j] + hb[j];
            const float s = a.clamp * (2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.44269504f * hv)) - 1.f);
            so[j] = s;
            const float es = __builtin_amdgcn_exp2f(esgn * s);
            yo[j] = a.rev ? (xin[j] - gv) * es : xin[j] * es + gv;
          }
          *reinterpret_cast<float4*>(a.x2out + pix * a.c2p + oc) = make_float4(yo[0], yo[1], yo[2], yo[3]);
          if (a.s_out) *reinterpret_cast<float4*>(a.s_out + pix * a.c2p + oc) = make_float4(so[0], so[1], so[2], so[3]);
          if (a.fd) {
            uint2 u;
            u.x = pack2(yo[0], yo[1]);
            u.y = pack2(yo[2], yo[3]);
            *reinterpret_cast<uint2*>(a.fd + (size_t)(oc >> 5) * a.plane + pix * 32 + (oc & 31)) = u;
          }
        }
      }
    }
  };

  if constexpr (FP) {
    const int t = wave;
    u32x4 bprv[NETS][KD];
    float4 xprv = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool vp = t >= 1, vn = t + 1 < a.T;                       // wave-uniform: frames outside the clip contribute nothing
    prefetch_x2(t);
    load_frame(bprv, xprv, vp ? t - 1 : t);
    load_frame(bcur, xcur, t);
    load_frame(bnxt, xnxt, vn ? t + 1 : t);
    auto tap_step = [&](const int tap, u32x4 (&bb)[NETS][KD], const float4& xx) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < NETS; ++q) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          f16x8 bf;
          if (HASX && ks == 0) {
            u32x4 u = {0u, 0u, 0u, 0u};
            if (kq == 0) {
              u.x = pack2(xx.x, xx.y);
              u.y = pack2(xx.z, 0.f);
            }
            bf = __builtin_bit_cast(f16x8, u);
          } else {
            bf = __builtin_bit_cast(f16x8, bb[q][ks - HASX]);
          }
#pragma unroll
          for (int o = 0; o < OT; ++o) {
            constexpr size_t TAPSTRIDE = (size_t)NETS * KS * OT * 1024;
            const unsigned char* wb = smem + ((size_t)((q * KS + ks) * OT + o) * 64 + lane) * 16 + (size_t)tap * TAPSTRIDE;
            accc[q][o] = mfma_16x16x32(*reinterpret_cast<const f16x8*>(wb), bf, accc[q][o]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    };
    if (vp) tap_step(0, bprv, xprv);
    tap_step(1, bcur, xcur);
    if (vn) tap_step(2, bnxt, xnxt);
    epilogue(t, accc);
    return;
  }
  load_frame(bcur, xcur, 0);
  for (int t = 0; t < a.T; ++t) {
    // (Measured and dropped: a THREE-frame register ring - biases in LDS, A fragments three at a time, SGPR-base
    // addressing to make room - i.e. two frames of loads in flight per wave: 0.65 -> 0.72 ms per step.  At ~5 TB/s the
    // kernel is no longer short of bytes in flight.)
    // x2 rows of the epilogue first: the VM counter retires loads in issue order, so waiting for these (at the
    // epilogue) must not mean waiting for the younger loads of frame t+1 as well - those stay in flight through
    // the epilogue and into the next iteration
    if (t >= 1) prefetch_x2(t - 1);
    __builtin_amdgcn_sched_barrier(0);
    if (t + 1 < a.T) load_frame(bnxt, xnxt, t + 1);
#pragma unroll
    for (int q = 0; q < NETS; ++q) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f16x8 bf;
        if (HASX && ks == 0) {
          // K entries 0..2 = the three input channels of this frame; lanes of k-groups 1..3 carry zeros
          u32x4 u = {0u, 0u, 0u, 0u};
          if (kq == 0) {
            u.x = pack2(xcur.x, xcur.y);
            u.y = pack2(xcur.z, 0.f);
          }
          bf = __builtin_bit_cast(f16x8, u);
        } else {
          bf = __builtin_bit_cast(f16x8, bcur[q][ks - HASX]);
        }
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          const unsigned char* wb = smem + ((size_t)((q * KS + ks) * OT + o) * 64 + lane) * 16;
          constexpr size_t TAPSTRIDE = (size_t)NETS * KS * OT * 1024;
          const f16x8 a0 = *reinterpret_cast<const f16x8*>(wb);
          const f16x8 a1 = *reinterpret_cast<const f16x8*>(wb + TAPSTRIDE);
          const f16x8 a2 = *reinterpret_cast<const f16x8*>(wb + 2 * TAPSTRIDE);
          accn[q][o] = mfma_16x16x32(a0, bf, accn[q][o]);  // -> out[t+1]
          accc[q][o] = mfma_16x16x32(a1, bf, accc[q][o]);  // -> out[t]
          accp[q][o] = mfma_16x16x32(a2, bf, accp[q][o]);  // -> out[t-1]
        }
        // keep the scheduler from hoisting every k-step's weight reads (it spills at 90 fragments)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (t >= 1) epilogue(t - 1, accp);
#pragma unroll
    for (int q = 0; q < NETS; ++q)
#pragma unroll
      for (int o = 0; o < OT; ++o) {
        accp[q][o] = accc[q][o];
        accc[q][o] = accn[q][o];
        accn[q][o] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
    for (int q = 0; q < NETS; ++q)
#pragma unroll
      for (int ks = 0; ks < KD; ++ks) bcur[q][ks] = bnxt[q][ks];
    xcur = xnxt;
  }
  prefetch_x2(a.T - 1);
  epilogue(a.T - 1, accp);
}

// fp32 NHWC (stride cinp) -> f16 channels [0, cin32) of the plane-blocked dense buffer (zero padded).
// One thread per (pixel, 8-channel group): 32 contiguous bytes in, 16 contiguous bytes out, neighbouring threads on
// neighbouring bytes (a thread per pixel walking its channels ran at 0.9 TB/s).
__global__ void nhwc_to_dense_kernel(const float* __restrict__ x, f16* __restrict__ d, size_t npix, int cin, int cinp, int cin32) {
  const int ngroups = cin32 >> 3;
  const size_t gi = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t pix = gi / ngroups;
  if (pix >= npix) return;
  const int c0 = (int)(gi - pix * ngroups) * 8;
  float v[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = c0 + 4 * h;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < cinp) r = *reinterpret_cast<const float4*>(x + pix * cinp + c);
    v[4 * h + 0] = c + 0 < cin ? r.x : 0.f;
    v[4 * h + 1] = c + 1 < cin ? r.y : 0.f;
    v[4 * h + 2] = c + 2 < cin ? r.z : 0.f;
    v[4 * h + 3] = c + 3 < cin ? r.w : 0.f;
  }
  const u32x4 u = {pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
  *reinterpret_cast<u32x4*>(d + (size_t)(c0 >> 5) * npix * 32 + pix * 32 + (c0 & 31)) = u;
}

// ---------------------------------------------------------------------------------
// host-side launch helpers
// ---------------------------------------------------------------------------------
// Two tile shapes: 16x16 pixels on 4 waves, and 12x16 on 3 waves for frame heights that 12 divides better (training
// crops: a 36-row LR frame is 3 x 12 exactly but 3 x 16 = 48 rows of tiles, i.e. 25 % fewer MFMAs and workgroups).
template <int TH, int TW>
constexpr int c3_lds() { return (TH + 2) * ((((TW + 2) * PS + 255) / 256) * 256) + 18 * 1024 + 128; }   // halo image + weight stage + 32 biases

template <int TH, int TW, int NW, int MT, int EPI, bool GEN>
int launch_conv3x3_cfg(C3Args& a, int nets_z, hipStream_t s) {
  a.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = (a.H + TH - 1) / TH;
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N), 1, (unsigned)nets_z);
  ProfScope prof(EPI == EPI_LRELU ? PROF_CONV3X3 : EPI == EPI_F ? PROF_CONV5_F : EPI == EPI_GH ? PROF_CONV5_GH : EPI == EPI_BWD ? -1 : PROF_CONV5_PLAIN, s);
  if (EPI == EPI_BWD && (size_t)grid.x * grid.z <= 2 * 256)
    hipLaunchKernelGGL((conv3x3_kernel<TH, TW, NW, MT, EPI, GEN, EPI == EPI_BWD>), grid, dim3(NW * 64), (c3_lds<TH, TW>()), s, a);
  else
    hipLaunchKernelGGL((conv3x3_kernel<TH, TW, NW, MT, EPI, GEN>), grid, dim3(NW * 64), (c3_lds<TH, TW>()), s, a);
  return hip_rc(hipGetLastError());
}

template <int TH, int TW, int NW, int MT, int EPI, bool GEN>
int launch_conv3x3_pair_cfg(C3Args& a, C3Args& b, int nets_z, hipStream_t s) {
  a.tiles_x = b.tiles_x = (a.W + TW - 1) / TW;
  a.tiles_y = b.tiles_y = (a.H + TH - 1) / TH;
  const dim3 grid((unsigned)(a.tiles_x * a.tiles_y * a.N), 2, (unsigned)nets_z);
  ProfScope prof(-1, s);
  hipLaunchKernelGGL((conv3x3_pair_kernel<TH, TW, NW, MT, EPI, GEN>), grid, dim3(NW * 64), (c3_lds<TH, TW>()), s, a, b);
  return hip_rc(hipGetLastError());
}

#ifdef SELFC_DEV
// timing aid (SELFC_ABLATE & 512): per-workgroup phase stamps of the last 256 EPI_BWD launches; tools/experiments/c3_stamps.py
unsigned long long* g_c3_stamps = nullptr;
int g_c3_launch = 0;
unsigned long long* c3_stamp_slot() {
  if (!g_c3_stamps) {
    if (hipMalloc(&g_c3_stamps, 256 * 512 * 8 * 8) != hipSuccess) return nullptr;
    (void)hipMemset(g_c3_stamps, 0, 256 * 512 * 8 * 8);
  }
  return g_c3_stamps + (size_t)(g_c3_launch++ % 256) * 512 * 8;
}
#endif

// two convs of one geometry (same N, H, W, stage count) in one launch
template <int EPI, bool GEN>
int launch_conv3x3_pair(C3Args& a, C3Args& b, int nets_z, hipStream_t s) {
#ifdef SELFC_DEV
  static const int ablate = getenv("SELFC_ABLATE") ? atoi(getenv("SELFC_ABLATE")) : 0;
  a.ablate = b.ablate = ablate;
  a.stamps = b.stamps = nullptr;
  if (EPI == EPI_BWD && (ablate & 512)) { a.stamps = c3_stamp_slot(); b.stamps = c3_stamp_slot(); }
#endif
  const int rows16 = (a.H + 15) / 16 * 16, rows12 = (a.H + 11) / 12 * 12;
  if (rows12 < rows16) return launch_conv3x3_pair_cfg<12, 16, 3, 2, EPI, GEN>(a, b, nets_z, s);
  return launch_conv3x3_pair_cfg<16, 16, 4, 2, EPI, GEN>(a, b, nets_z, s);
}

template <int EPI, bool GEN = false>
int launch_conv3x3(C3Args& a, int nets_z, hipStream_t s) {
#ifdef SELFC_DEV
  static const int ablate = getenv("SELFC_ABLATE") ? atoi(getenv("SELFC_ABLATE")) : 0;
  a.ablate = ablate;
  a.stamps = (EPI == EPI_BWD && (ablate & 512)) ? c3_stamp_slot() : nullptr;
#endif
  const int rows16 = (a.H + 15) / 16 * 16, rows12 = (a.H + 11) / 12 * 12;
  if (rows12 < rows16) return launch_conv3x3_cfg<12, 16, 3, 2, EPI, GEN>(a, nets_z, s);
  return launch_conv3x3_cfg<16, 16, 4, 2, EPI, GEN>(a, nets_z, s);
}

// stage description of conv `layer` (1..5) of a dense block with `cin` inputs.
// Dense layout: cin <= 3 : [f1 f2 f3 f4] (input comes from x1 through the im2col stage)
//               cin  > 3 : [x (cin, zero padded to cin32) | f1 f2 f3 f4]
void build_stages(C3Args& a, int cin, int layer) {
  if (cin <= 3) {
    a.has_im2col = 1; a.cin16 = 0; a.n_in = 0; a.fbase = 0;
    a.nstages = 1 + (layer - 1);
  } else {
    a.has_im2col = 0;
    a.cin16 = (cin + 15) & ~15;
    a.n_in = (a.cin16 + 31) / 32;
    a.fbase = (cin + 31) & ~31;
    a.nstages = a.n_in + (layer - 1);
  }
}

inline int dense_channels(int cin) { return cin <= 3 ? 128 : ((cin + 31) & ~31) + 128; }

template <int NETS, int OT, int KD, int HASX, int EPI>
int launch_t5(const T5Args& a, hipStream_t s) {
  constexpr int lds = 3 * NETS * (KD + HASX) * OT * 1024;
  if constexpr (EPI == EPI_GH || EPI == EPI_T5B) {
    // frame-parallel instance for small problems (SELFC_T5_FP_MAX pixel-frames, default two 36x36 training septuplets; 0: never)
    static const long fp_max = getenv("SELFC_T5_FP_MAX") ? atol(getenv("SELFC_T5_FP_MAX")) : 20000;
    if (a.T <= t5_waves<EPI>() && (long)a.B * a.T * a.HW <= fp_max) {
      static std::atomic<unsigned long long> optin_fp{0};
      if (lds > 64 * 1024)
        if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&tconv5_kernel<NETS, OT, KD, HASX, EPI, true>), lds, optin_fp); e != hipSuccess) return hip_rc(e);
      const int tiles_fp = (a.HW + 15) / 16;
      ProfScope prof(EPI == EPI_GH ? PROF_CONV5_GH : -1, s);
      hipLaunchKernelGGL((tconv5_kernel<NETS, OT, KD, HASX, EPI, true>), dim3((unsigned)(tiles_fp * a.B), EPI == EPI_T5B ? (unsigned)(a.pair_planes ? 2 * a.pair_planes : a.coutp) : 1u),
                         dim3(t5_waves<EPI>() * 64), lds, s, a);
      return hip_rc(hipGetLastError());
    }
  }
  static std::atomic<unsigned long long> optin{0};
  if (lds > 64 * 1024)
    if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&tconv5_kernel<NETS, OT, KD, HASX, EPI>), lds, optin); e != hipSuccess) return hip_rc(e);
  constexpr int pxwg = t5_waves<EPI>() * 16;
  const int tiles = (a.HW + pxwg - 1) / pxwg;
  ProfScope prof(EPI == EPI_F ? PROF_CONV5_F : EPI == EPI_GH ? PROF_CONV5_GH : EPI == EPI_T5B ? -1 : PROF_CONV5_PLAIN, s);
  hipLaunchKernelGGL((tconv5_kernel<NETS, OT, KD, HASX, EPI>), dim3((unsigned)(tiles * a.B), EPI == EPI_T5B ? (unsigned)(a.pair_planes ? 2 * a.pair_planes : a.coutp) : 1u), dim3(t5_waves<EPI>() * 64), lds, s, a);
  return hip_rc(hipGetLastError());
}

// dispatch over the compiled (outch tiles, dense k-steps) combinations.  Only what a caller can reach is instantiated: F's
// conv5 (one net, no fp32 x input) has one outch tile and 5..7 dense k-steps (c2 <= 32 / 64 / 96); the G/H pair always has the
// four feature planes and 1..3 outch tiles (c2 <= 48: every shipped net; the four-tile instance spilled 18 VGPRs and is not built -
// a block with a wider x2 runs composed from stand-alone subnets, modules/Inv_arch.py); stand-alone subnets (EPI_PLAIN) keep the
// full table.
template <int NETS, int HASX, int EPI>
int dispatch_t5(const T5Args& a, int ot, int kd, hipStream_t s) {
#define SELFC_T5_CASE(OT_, KD_) \
  if (ot == OT_ && kd == KD_) return launch_t5<NETS, OT_, KD_, HASX, EPI>(a, s);
  if constexpr (NETS == 1) {
    SELFC_T5_CASE(1, 4) SELFC_T5_CASE(1, 5) SELFC_T5_CASE(1, 6) SELFC_T5_CASE(1, 7)
    if constexpr (EPI != EPI_F) {          // stand-alone subnets (selfc_subnet_run): up to 128 output channels
      SELFC_T5_CASE(2, 4) SELFC_T5_CASE(2, 5) SELFC_T5_CASE(2, 6)
      SELFC_T5_CASE(3, 4) SELFC_T5_CASE(3, 5) SELFC_T5_CASE(3, 6)
      SELFC_T5_CASE(4, 4) SELFC_T5_CASE(4, 5) SELFC_T5_CASE(4, 6)
    }
  } else {
    SELFC_T5_CASE(1, 4) SELFC_T5_CASE(2, 4) SELFC_T5_CASE(3, 4)
  }
#undef SELFC_T5_CASE
  return SELFC_EINVAL;
}

bool latent_ok(const selfc_latent* l) {
  if (!l || !l->x1 || !l->x2 || !l->fd || !l->gd || !l->hd) return false;
  if (l->N <= 0 || l->T <= 0 || l->N % l->T || l->H <= 0 || l->W <= 0) return false;
  if (l->c1 < 1 || l->c1 > 3 || l->c2 < 4) return false;
  if (l->kind == SELFC_SUBNET_D2DT) return l->c2 <= 48;
  if (l->kind == SELFC_SUBNET_DB2D) return l->c2 <= 32;
  return false;
}

// conv1..4 of one subnet (or of the G/H pair when wb != null) on its dense buffer(s)
int run_conv1to4(const selfc_subnet_w* wa, const selfc_subnet_w* wb, void* da, void* db, const float* x1,
                 int cin, int N, int H, int W, hipStream_t s) {
  for (int layer = 1; layer <= 4; ++layer) {
    C3Args a{};
    a.dense[0] = (const f16*)da; a.out[0] = (f16*)da;
    a.w[0] = (const f16*)wa->w3[layer - 1]; a.bias[0] = wa->b3[layer - 1];
    if (wb) {
      a.dense[1] = (const f16*)db; a.out[1] = (f16*)db;
      a.w[1] = (const f16*)wb->w3[layer - 1]; a.bias[1] = wb->b3[layer - 1];
    }
    a.x1 = x1;
    a.N = N; a.H = H; a.W = W; a.plane = (size_t)N * H * W * 32; a.c1 = cin <= 3 ? cin : 0;
    build_stages(a, cin, layer);
    a.out_coff = (cin <= 3 ? 0 : ((cin + 31) & ~31)) + 32 * (layer - 1);
    const int rc = launch_conv3x3<EPI_LRELU>(a, wb ? 2 : 1, s);
    if (rc) return rc;
  }
  return SELFC_OK;
}

int check_subnet(const selfc_subnet_w* w, bool need5) {
  for (int i = 0; i < 4; ++i)
    if (!w->w3[i] || !w->b3[i]) return SELFC_EINVAL;
  if (need5 && (!w->w5 || !w->b5)) return SELFC_EINVAL;
  return SELFC_OK;
}

int run_F(const selfc_invblock_w* blk, const selfc_latent* l, int rev, float* x1in, float* x1out, hipStream_t s) {
  static const bool no_fuse_f = getenv("SELFC_NO_FUSE_F") != nullptr;     // developer A/B switch
  int rc;
  if (blk->F.wfused && l->c2 == 48 && !no_fuse_f) {
    // two pairwise-fused launches; with the partial-product fragments they also cover the temporal conv5 (rc 1)
    const bool t5 = l->kind == SELFC_SUBNET_D2DT && l->c1 <= 3;
    rc = launch_fused_f(l->fd, blk->F.wfused, blk->F.b3, l->N, l->H, l->W, s, t5 ? blk->F.w5p : nullptr, t5 ? l->pf : nullptr,
                        blk->F.b5, x1in, x1out, l->T, rev, (l->flags & SELFC_LAT_KEEP_FEATURES) != 0);
    if (rc == 1) return SELFC_OK;
  } else
    rc = run_conv1to4(&blk->F, nullptr, l->fd, nullptr, nullptr, l->c2, l->N, l->H, l->W, s);
  if (rc) return rc;
  const int FC = dense_channels(l->c2);
  if (l->kind == SELFC_SUBNET_D2DT) {
    T5Args a{};
    a.dense[0] = (const f16*)l->fd; a.w = (const f16*)blk->F.w5; a.bias[0] = blk->F.b5;
    a.B = l->N / l->T; a.T = l->T; a.HW = l->H * l->W; a.plane = (size_t)l->N * l->H * l->W * 32;
    a.x1io = x1in; a.x1out = x1out; a.rev = rev;
    return dispatch_t5<1, 0, EPI_F>(a, 1, FC / 32, s);
  }
  C3Args a{};
  a.dense[0] = (const f16*)l->fd; a.w[0] = (const f16*)blk->F.w5; a.bias[0] = blk->F.b5;
  a.N = l->N; a.H = l->H; a.W = l->W; a.plane = (size_t)l->N * l->H * l->W * 32;
  build_stages(a, l->c2, 5);
  a.x1io = x1in; a.x1out = x1out; a.rev = rev;
  return launch_conv3x3<EPI_F>(a, 1, s);
}

// x1: the side G / H read (forward: the y1 run_F just wrote); x2in -> x2out: the coupling
int run_GH(const selfc_invblock_w* blk, const selfc_latent* l, int rev, const float* x1, float* x2in, float* x2out, hipStream_t s) {
  int rc;
  static const bool no_fuse = getenv("SELFC_NO_FUSE") != nullptr;     // developer A/B switch
  if (blk->G.wfused && blk->H.wfused && l->c1 == 3 && !no_fuse) {
    FGArgs fa{};
    fa.x1 = x1;
    fa.w[0] = (const f16*)blk->G.wfused; fa.w[1] = (const f16*)blk->H.wfused;
    for (int i = 0; i < 4; ++i) { fa.bias[0][i] = blk->G.b3[i]; fa.bias[1][i] = blk->H.b3[i]; }
    fa.dense[0] = (f16*)l->gd; fa.dense[1] = (f16*)l->hd;
    fa.N = l->N; fa.H = l->H; fa.W = l->W;
    rc = launch_fused_gh(fa, s, 2);
  } else {
    rc = run_conv1to4(&blk->G, &blk->H, l->gd, l->hd, x1, l->c1, l->N, l->H, l->W, s);
  }
  if (rc) return rc;
  const int c2p = (l->c2 + 3) & ~3;
  if (l->kind == SELFC_SUBNET_D2DT) {
    T5Args a{};
    a.dense[0] = (const f16*)l->gd; a.dense[1] = (const f16*)l->hd;
    a.w = (const f16*)blk->G.w5; a.bias[0] = blk->G.b5; a.bias[1] = blk->H.b5;
    a.x1 = x1;
    a.B = l->N / l->T; a.T = l->T; a.HW = l->H * l->W; a.plane = (size_t)l->N * l->H * l->W * 32;
    a.x2io = x2in; a.x2out = x2out; a.fd = (f16*)((!rev && l->fd_next) ? l->fd_next : l->fd); a.s_out = l->s_out; a.c2p = c2p;
    a.rev = rev; a.clamp = blk->clamp;
    return dispatch_t5<2, 1, EPI_GH>(a, (l->c2 + 15) / 16, 4, s);
  }
  C3Args a{};
  a.dense[0] = (const f16*)l->gd; a.dense[1] = (const f16*)l->hd;
  a.w[0] = (const f16*)blk->G.w5; a.w[1] = (const f16*)blk->H.w5;
  a.bias[0] = blk->G.b5; a.bias[1] = blk->H.b5;
  a.x1 = x1; a.c1 = l->c1;
  a.N = l->N; a.H = l->H; a.W = l->W; a.plane = (size_t)l->N * l->H * l->W * 32;
  build_stages(a, l->c1, 5);
  a.x2io = x2in; a.x2out = x2out; a.fd = (f16*)((!rev && l->fd_next) ? l->fd_next : l->fd); a.s_out = l->s_out; a.c2p = c2p;
  a.rev = rev; a.clamp = blk->clamp;
  return launch_conv3x3<EPI_GH>(a, 1, s);
}

}  // namespace

namespace selfc {
#ifdef SELFC_DEV
unsigned long long* dev_stamp_slot() { return c3_stamp_slot(); }     // csrc/dgrad_chain.hip stamps its launches into the same ring
#endif

// conv5^T of a temporal dense block: out plane z (z < nplanes_out) = sum over the three temporal taps of W_z[tap] g[t + tap - 1];
// g = ng scaled f16 gradient planes; plane mask_z is multiplied by LeakyReLU'(mask) and stored to `alt`.
int bwd_tconv5T(const void* g, int ng, const void* w, int nplanes_out, void* out_planes, const void* mask, int mask_z, void* alt,
                int N, int T, int H, int W, hipStream_t s) {
  T5Args a{};
  a.dense[0] = (const f16*)g; a.w = (const f16*)w;
  a.B = N / T; a.T = T; a.HW = H * W; a.plane = (size_t)N * H * W * 32;
  a.outp = (f16*)out_planes; a.maskp = (const f16*)mask; a.altp = (f16*)alt; a.mask_z = mask_z;
  a.coutp = nplanes_out;                                  // grid.y
  a.wz_stride = (size_t)3 * ng * 2 * 512;                 // halfs per output plane: 3 taps x ng k-steps x 2 out tiles
  if (ng == 1) return launch_t5<1, 2, 1, 0, EPI_T5B>(a, s);
  if (ng == 2) return launch_t5<1, 2, 2, 0, EPI_T5B>(a, s);
  if (ng == 3) return launch_t5<1, 2, 3, 0, EPI_T5B>(a, s);
  return SELFC_EINVAL;
}

// the same for two nets of one geometry in ONE launch (grid.y = 2 x nplanes_out)
int bwd_tconv5T_pair(const void* g0, const void* g1, int ng, const void* w0, const void* w1, int nplanes_out, void* out0, void* out1,
                     const void* mask0, const void* mask1, int mask_z, void* alt0, void* alt1, int N, int T, int H, int W, hipStream_t s) {
  T5Args a{};
  a.dense[0] = (const f16*)g0; a.dense[1] = (const f16*)g1; a.w = (const f16*)w0; a.w2 = (const f16*)w1;
  a.B = N / T; a.T = T; a.HW = H * W; a.plane = (size_t)N * H * W * 32;
  a.outp = (f16*)out0; a.outp2 = (f16*)out1; a.maskp = (const f16*)mask0; a.maskp2 = (const f16*)mask1;
  a.altp = (f16*)alt0; a.altp2 = (f16*)alt1; a.mask_z = mask_z;
  a.coutp = nplanes_out; a.pair_planes = nplanes_out;
  a.wz_stride = (size_t)3 * ng * 2 * 512;
  if (ng == 1) return launch_t5<1, 2, 1, 0, EPI_T5B>(a, s);
  if (ng == 2) return launch_t5<1, 2, 2, 0, EPI_T5B>(a, s);
  if (ng == 3) return launch_t5<1, 2, 3, 0, EPI_T5B>(a, s);
  return SELFC_EINVAL;
}

// Generic plane-list conv with the EPI_BWD epilogue (see C3Args); used by csrc/backward.hip for the data
// gradients of a dense block.  `in` = first of nplanes_in contiguous f16 planes; kt temporal taps (1 | 3);
// sp1: (kt,1,1) kernel instead of (kt,3,3); ngroups 32-channel output groups.
static C3Args bwd_conv_args(const BwdConv& c, int N, int T, int H, int W) {
  C3Args a{};
  a.dense[0] = (const f16*)c.in; a.out[0] = (f16*)c.out_planes;
  a.w[0] = (const f16*)c.w;
  a.N = N; a.H = H; a.W = W; a.plane = (size_t)N * H * W * 32;
  a.gen_planes = c.nplanes_in; a.gen_tt = c.kt; a.T = T; a.gen_sp1 = c.sp1;
  a.nstages = c.nplanes_in * c.kt;
  a.wz_stride = (size_t)a.nstages * (c.sp1 ? 2 : 18) * 512;
  if (c.in2) {             // second source: nplanes_in2 more stages (BwdConv::in2)
    a.dense[1] = (const f16*)c.in2; a.w[1] = (const f16*)c.w2;
    a.gen_split = a.nstages; a.nstages += c.nplanes_in2;
    a.wz_stride2 = (size_t)c.nplanes_in2 * 18 * 512;
    a.bw_add2 = (const f16*)c.add2;
  }
  a.out_coff = 0;
  a.bw_add = (const f16*)c.add; a.bw_mask = (const f16*)c.mask; a.bw_mask_z = c.mask_z; a.bw_alt = (f16*)c.alt;
  a.bw_amax = c.amax; a.bw_acc = c.accumulate; a.bw_amax_out = c.amax_out;
  a.plain = c.plain; a.coutp = c.coutp;
  return a;
}

int bwd_conv_planes(const BwdConv& c, int N, int T, int H, int W, hipStream_t s) {
  if (c.in2 && (c.kt != 1 || c.sp1)) return SELFC_EINVAL;
  C3Args a = bwd_conv_args(c, N, T, H, W);
  return launch_conv3x3<EPI_BWD, true>(a, c.ngroups, s);
}

// two data-gradient convs of one geometry (same plane counts, taps, groups) in one launch
int bwd_conv_planes_pair(const BwdConv& c0, const BwdConv& c1, int N, int T, int H, int W, hipStream_t s) {
  if (c0.nplanes_in != c1.nplanes_in || c0.kt != c1.kt || c0.sp1 != c1.sp1 || c0.ngroups != c1.ngroups || c0.in2 || c1.in2) return SELFC_EINVAL;
  C3Args a = bwd_conv_args(c0, N, T, H, W), b = bwd_conv_args(c1, N, T, H, W);
  return launch_conv3x3_pair<EPI_BWD, true>(a, b, c0.ngroups, s);
}

}  // namespace selfc

extern "C" {

const char* selfc_version(void) { return "selfc_hip gfx950 abi12 operands=" SELFC_OPERAND_NAME; }
int selfc_abi_version(void) { return 13; }

int selfc_invblock_run(const selfc_invblock_w* blk, const selfc_latent* lat, int rev, void* stream) {
  if (!blk || !latent_ok(lat)) return SELFC_EINVAL;
  const bool db2d = lat->kind == SELFC_SUBNET_DB2D;
  if (check_subnet(&blk->F, true) || check_subnet(&blk->G, true) || check_subnet(&blk->H, db2d)) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  // in place unless the caller names separate output buffers (autograd keeps the inputs for the backward pass instead of cloning them)
  float* const x1in = (float*)lat->x1, *const x2in = (float*)lat->x2;
  float* const x1o = lat->x1_out ? (float*)lat->x1_out : x1in, *const x2o = lat->x2_out ? (float*)lat->x2_out : x2in;
  if (!rev) {
    if ((rc = run_F(blk, lat, 0, x1in, x1o, s))) return rc;        // y1 = x1 + F(x2)
    return run_GH(blk, lat, 0, x1o, x2in, x2o, s);                 // y2 = x2*exp(s(y1)) + G(y1)
  }
  if ((rc = run_GH(blk, lat, 1, x1in, x2in, x2o, s))) return rc;   // y2 = (x2 - G(x1)) / exp(s(x1))
  return run_F(blk, lat, 1, x1in, x1o, s);                         // y1 = x1 - F(y2)
}

int selfc_invstack_run(const selfc_invblock_w* blks, int nblk, const selfc_latent* lat, int rev, void* stream) {
  if (!blks || nblk < 0) return SELFC_EINVAL;
  for (int i = 0; i < nblk; ++i) {
    const int rc = selfc_invblock_run(&blks[rev ? nblk - 1 - i : i], lat, rev, stream);
    if (rc) return rc;
  }
  return SELFC_OK;
}

int selfc_subnet_run(const selfc_subnet_w* w, int kind, const float* xin, float* yout, void* dense,
                     int N, int T, int H, int W, int cin, int cout, void* stream) {
  if (!w || !yout || !dense || N <= 0 || T <= 0 || N % T || H <= 0 || W <= 0 || cin < 1 || cout < 1) return SELFC_EINVAL;
  if (!xin && cin <= 3) return SELFC_EINVAL;          // a 3-channel input is read from its fp32 rows by conv1 itself
  if (check_subnet(w, true)) return SELFC_EINVAL;
  if (cin > 96) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int DC = dense_channels(cin);
  const int coutp = (cout + 3) & ~3;
  if (cin > 3 && xin) {        // xin == NULL: the producer already wrote the input planes (selfc_globalagg_run_d)
    const size_t npix = (size_t)N * H * W;
    hipLaunchKernelGGL(nhwc_to_dense_kernel, dim3((unsigned)((npix * (((cin + 31) & ~31) >> 3) + 255) / 256)), dim3(256), 0, s,
                       xin, (f16*)dense, npix, cin, (cin + 3) & ~3, (cin + 31) & ~31);
    int rc = hip_rc(hipGetLastError());
    if (rc) return rc;
  }
  int rc;
  static const bool no_fuse = getenv("SELFC_NO_FUSE") != nullptr;     // developer A/B switch
  if (cin == 3 && w->wfused && !no_fuse) {        // conv1..4 as ONE persistent launch (the G / H kernel on a single net)
    FGArgs fa{};
    fa.x1 = xin;
    fa.w[0] = fa.w[1] = (const f16*)w->wfused;
    for (int i = 0; i < 4; ++i) fa.bias[0][i] = fa.bias[1][i] = w->b3[i];
    fa.dense[0] = fa.dense[1] = (f16*)dense;
    fa.N = N; fa.H = H; fa.W = W;
    rc = launch_fused_gh(fa, s, 1);
  } else
    rc = run_conv1to4(w, nullptr, dense, nullptr, cin <= 3 ? xin : nullptr, cin, N, H, W, s);
  if (rc) return rc;
  if (kind == SELFC_SUBNET_D2DT) {
    if (cout > 64) return SELFC_EINVAL;
    T5Args a{};
    a.dense[0] = (const f16*)dense; a.w = (const f16*)w->w5; a.bias[0] = w->b5;
    a.x1 = cin <= 3 ? xin : nullptr;
    a.B = N / T; a.T = T; a.HW = H * W; a.plane = (size_t)N * H * W * 32;
    a.plain = yout; a.coutp = coutp;
    const int ot = (cout + 15) / 16;
    if (cin <= 3) return dispatch_t5<1, 1, EPI_PLAIN>(a, ot, 4, s);
    return dispatch_t5<1, 0, EPI_PLAIN>(a, ot, DC / 32, s);
  }
  if (kind != SELFC_SUBNET_DB2D || cout > 32) return SELFC_EINVAL;
  C3Args a{};
  a.dense[0] = (const f16*)dense; a.w[0] = (const f16*)w->w5; a.bias[0] = w->b5;
  a.x1 = cin <= 3 ? xin : nullptr; a.c1 = cin <= 3 ? cin : 0;
  a.N = N; a.H = H; a.W = W; a.plane = (size_t)N * H * W * 32;
  build_stages(a, cin, 5);
  a.plain = yout; a.coutp = coutp;
  return launch_conv3x3<EPI_PLAIN>(a, 1, s);
}


int selfc_nhwc_to_planes(const float* x, void* dense, size_t npix, int cin, void* stream) {
  if (!x || !dense || npix == 0 || cin < 1) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_TRANSFORM, s);
  hipLaunchKernelGGL(nhwc_to_dense_kernel, dim3((unsigned)((npix * (((cin + 31) & ~31) >> 3) + 255) / 256)), dim3(256), 0, s,
                     x, (f16*)dense, npix, cin, (cin + 3) & ~3, (cin + 31) & ~31);
  return hip_rc(hipGetLastError());
}

int selfc_conv_planes_run(void* dense, int nplanes_in, int kt, const void* w, const float* bias, int cout,
                          int out_plane, float* plain, int N, int T, int H, int W, void* stream) {
  if (!dense || !w || !bias || nplanes_in < 1 || (kt != 1 && kt != 3) || cout < 32 || cout % 32) return SELFC_EINVAL;
  if (N <= 0 || T <= 0 || N % T || H <= 0 || W <= 0) return SELFC_EINVAL;
  if ((out_plane < 0) == (plain == nullptr)) return SELFC_EINVAL;      // exactly one kind of output
  if (out_plane >= 0 && out_plane < nplanes_in) return SELFC_EINVAL;    // appended planes must not alias the inputs
  hipStream_t s = (hipStream_t)stream;
  C3Args a{};
  a.dense[0] = (const f16*)dense; a.out[0] = (f16*)dense;
  a.w[0] = (const f16*)w; a.bias[0] = bias;
  a.N = N; a.H = H; a.W = W; a.plane = (size_t)N * H * W * 32;
  a.gen_planes = nplanes_in; a.gen_tt = kt; a.T = T;
  a.nstages = nplanes_in * kt;
  a.wz_stride = (size_t)a.nstages * 18 * 512;          // halfs of one 32-channel output group's fragments
  const int zg = cout / 32;
  if (out_plane >= 0) {
    a.out_coff = out_plane * 32;
    return launch_conv3x3<EPI_LRELU, true>(a, zg, s);
  }
  a.plain = plain; a.coutp = cout;
  return launch_conv3x3<EPI_PLAIN, true>(a, zg, s);
}

}  // extern "C"

#ifdef SELFC_DEV
extern "C" int selfc_dev_c3_stamps(const char* path) {
  if (!g_c3_stamps) return -1;
  const size_t n = 256 * 512 * 8;
  unsigned long long* h = (unsigned long long*)malloc(n * 8);
  if (hipMemcpy(h, g_c3_stamps, n * 8, hipMemcpyDeviceToHost) != hipSuccess) { free(h); return -2; }
  FILE* f = fopen(path, "wb");
  if (!f) { free(h); return -3; }
  fwrite(h, 8, n, f);
  fclose(f);
  free(h);
  return g_c3_launch;
}
#endif
