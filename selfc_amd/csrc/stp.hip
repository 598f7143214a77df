// STP (self-conditioned latent predictor) kernels for gfx950: GlobalAgg, the 1x1x1 MLP head and
// the GMM sampler of SelfC_GMM_arch_inv.py:257-285,327-344,382-394.  Activations between STP
// modules are fp32 NHWC [N][H*W][64].
//
//   GlobalAgg(x) = x + (proj1(x) viewed (b, C*h*w, T)) @ A,   A = softmax(q k^T / C) over the
//   last axis, q/k = proj2/proj3(fc(adaptive_avg_pool2d(x, 32x32))).
// Everything is linear in x except A, so it is computed as
//   (1) gagg_pool : g[n][c] = sum_px x[n][px][c] * wmap[px]      (fc o adaptive pooling folded into
//                   one HxW weight map on the host; one read of x, deterministic 2-stage sum)
//   (2) gagg_attn : per clip, q/k projections and the TxT softmax (tiny)
//   (3) gagg_mix  : y[t2] = x[t2] + W1 (sum_t1 A[t1][t2] x[t1]) + b1 sum_t1 A[t1][t2]
//                   temporal mix in fp32 registers, then ONE 64x64 MFMA contraction per frame.
#include <type_traits>
#include "common.hpp"
#include "prof.hpp"
#include "bwd_internal.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace {

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

constexpr int POOL_CHUNK = 512;  // pixels per partial sum
constexpr int BWD_CHUNK = 96;    // pixels per partial of the GlobalAgg backward reduction: a training frame (36x36 latent) is 14 chunks,
                                 // 112 workgroups per 8 clips instead of 24 (the kernel was pure latency: 126 us)
constexpr int TMAX = 8;

// ---- (1) weighted global pooling: partial[n][chunk][64]
__global__ __launch_bounds__(256) void gagg_pool_kernel(const float* __restrict__ x, const float* __restrict__ wmap,
                                                        float* __restrict__ partial, int HW, int nchunk) {
  __shared__ float4 red[16][16];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int cq = threadIdx.x & 15, pr = threadIdx.x >> 4;
  const int p0 = chunk * POOL_CHUNK, p1 = min(p0 + POOL_CHUNK, HW);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* base = x + (size_t)n * HW * 64 + cq * 4;
  for (int p = p0 + pr; p < p1; p += 16) {
    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)p * 64);
    const float w = wmap[p];
    acc.x += v.x * w; acc.y += v.y * w; acc.z += v.z * w; acc.w += v.w * w;
  }
  red[pr][cq] = acc;
  __syncthreads();
  if (pr == 0) {
    float4 s = red[0][cq];
    for (int r = 1; r < 16; ++r) {
      const float4 v = red[r][cq];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(partial + ((size_t)n * nchunk + chunk) * 64 + cq * 4) = s;
  }
}

// ---- (2) per clip: g = sum(partials) + fc.bias; q = proj2(g), k = proj3(g); A = softmax(q k^T / C, dim=-1), left in LDS.
// Runs as the PROLOGUE of every mix workgroup of the clip (a few kFLOP, redundantly): as a launch of its own it was one
// workgroup per clip and 16 us of pure latency in front of the mix, six times per STP pass.  Built for latency: thread
// (matrix, output channel, half of the 64 inputs) holds its half row of W2 / W3 in registers (8 float4 straight from L2, no
// LDS staging), reads g as float4 LDS broadcasts and meets its partner lane through one shuffle - the first version (one
// thread per output walking 64 scalar LDS reads, three barriers of 64-step dependent loops) cost 13 us per workgroup.
struct AttnLds {
  float g[TMAX][64], q[TMAX][64], k[TMAX][64], m[TMAX][TMAX];
};

template <int TT>
__device__ __forceinline__ void clip_attention(AttnLds& L, int b, int tid, const float* __restrict__ partial, int nchunk,
                                               const float* __restrict__ fcbp, const float* __restrict__ w2, const float* __restrict__ b2,
                                               const float* __restrict__ w3, const float* __restrict__ b3, int Trt, float creal) {
  const int T = TT ? TT : Trt;          // clip length as a compile-time constant where the launcher knows it (see gagg_attn_bwd_kernel)
  const int mat = tid >> 7, c = (tid >> 1) & 63, jh = tid & 1;
  float4 wr[8];
  {
    const float4* wrow = reinterpret_cast<const float4*>((mat ? w3 : w2) + c * 64 + jh * 32);
#pragma unroll
    for (int i = 0; i < 8; ++i) wr[i] = wrow[i];
  }
  const float bias = (mat ? b3 : b2)[c];
  const float fcb = *fcbp;
  if (tid < T * 16) {       // g: one thread per (frame, 4 channels); the chunk partials are fetched 16 at a time (independent
    const int t = tid >> 4, c4 = tid & 15;          // loads in flight - one by one they were 14 L2 round trips in a row)
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* p = reinterpret_cast<const float4*>(partial + (size_t)(b * T + t) * nchunk * 64) + c4;
    for (int j0 = 0; j0 < nchunk; j0 += 16) {
      float4 v[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) v[jj] = j0 + jj < nchunk ? p[(size_t)(j0 + jj) * 16] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) { s.x += v[jj].x; s.y += v[jj].y; s.z += v[jj].z; s.w += v[jj].w; }
    }
    *reinterpret_cast<float4*>(&L.g[t][c4 * 4]) = make_float4(s.x + fcb, s.y + fcb, s.z + fcb, s.w + fcb);
  }
  __syncthreads();
  float acc[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    acc[t] = 0.f;
    if (t < T) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 gv = *reinterpret_cast<const float4*>(&L.g[t][jh * 32 + 4 * i]);
        acc[t] += gv.x * wr[i].x + gv.y * wr[i].y + gv.z * wr[i].z + gv.w * wr[i].w;
      }
    }
    acc[t] += __shfl_xor(acc[t], 1);
  }
  if (jh == 0) {
    float (*dst)[64] = mat ? L.k : L.q;
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < T) dst[t][c] = acc[t] + bias;
  }
  __syncthreads();
  if (tid < T * T) {
    const int t1 = tid / T, t2 = tid % T;
    float s = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const float4 qv = *reinterpret_cast<const float4*>(&L.q[t1][4 * jj]);
      const float4 kv = *reinterpret_cast<const float4*>(&L.k[t2][4 * jj]);
      s += qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w;
    }
    L.m[t1][t2] = s / creal;        // the reference divides by the module's REAL channel count C (64; 24 in the codec variant)
  }
  __syncthreads();
  if (tid < T) {            // softmax over the last axis of row tid
    float mx = L.m[tid][0];
    for (int jj = 1; jj < T; ++jj) mx = fmaxf(mx, L.m[tid][jj]);
    float e[TMAX], s = 0.f;
    for (int jj = 0; jj < T; ++jj) { e[jj] = expf(L.m[tid][jj] - mx); s += e[jj]; }
    for (int jj = 0; jj < T; ++jj) L.m[tid][jj] = e[jj] / s;
  }
  __syncthreads();
}

// ---- (3) temporal mix + 64x64 projection + residual.  Wave = 16 pixels of one clip.
// DENSE: the consumer is a D2DTInput - the result goes straight into planes 0..1 of ITS plane-blocked f16 operand buffer
// ([plane][N][HW][32], the rounding its nhwc_to_dense pass would apply) instead of an fp32 row the next launch converts.
template <bool DENSE, int TT>
__global__ __launch_bounds__(256) void gagg_mix_kernel(const float* __restrict__ x, float* __restrict__ y, f16* __restrict__ dense,
                                                       size_t plane, const float* __restrict__ partial, int nchunk,
                                                       const float* __restrict__ fcbp, const float* __restrict__ w2,
                                                       const float* __restrict__ b2, const float* __restrict__ w3,
                                                       const float* __restrict__ b3, float* __restrict__ attn_out,
                                                       const f16* __restrict__ w1, const float* __restrict__ b1, int Trt, int HW,
                                                       float creal) {
  const int T = TT ? TT : Trt;
  __shared__ AttnLds L;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles = (HW + 63) / 64;
  const int b = blockIdx.x / tiles;
  clip_attention<TT>(L, b, threadIdx.x, partial, nchunk, fcbp, w2, b2, w3, b3, T, creal);
  if (attn_out && blockIdx.x % tiles == 0 && threadIdx.x < T * T)
    attn_out[(size_t)b * T * T + threadIdx.x] = L.m[threadIdx.x / T][threadIdx.x % T];
  const int p0 = (blockIdx.x % tiles) * 64 + wave * 16;
  if (p0 >= HW) return;                    // no barriers below
  const int pl = p0 + (lane & 15);
  const bool pvalid = pl < HW;
  const int pc = pvalid ? pl : HW - 1;
  const int kq = lane >> 4;

  // this lane's 16 channels (k = 32 ks + 8 kq + j) of every frame of the clip.  (Requesting them BEFORE the prologue so that
  // both share one memory round trip was measured: 160 more live registers across it, 34 -> 45 us at four clips.)
  float xs[TMAX][16];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (t < T) {
      const float* p = x + ((size_t)(b * T + t) * HW + pc) * 64 + kq * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float4 v0 = *reinterpret_cast<const float4*>(p + ks * 32);
        const float4 v1 = *reinterpret_cast<const float4*>(p + ks * 32 + 4);
        xs[t][8 * ks + 0] = v0.x; xs[t][8 * ks + 1] = v0.y; xs[t][8 * ks + 2] = v0.z; xs[t][8 * ks + 3] = v0.w;
        xs[t][8 * ks + 4] = v1.x; xs[t][8 * ks + 5] = v1.y; xs[t][8 * ks + 6] = v1.z; xs[t][8 * ks + 7] = v1.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) xs[t][j] = 0.f;
    }
  }
  // proj1 weights as A fragments [otile 4][kstep 2]: W1[16 o + (lane&15)][32 ks + 8 kq + j]
  f16x8 wf[4][2];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      wf[o][ks] = *reinterpret_cast<const f16x8*>(w1 + ((size_t)(o * 2 + ks) * 64 + lane) * 8);

#pragma unroll
  for (int t2 = 0; t2 < TMAX; ++t2) {
    if (t2 >= T) break;
    float xm[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xm[j] = 0.f;
    float colsum = 0.f;
#pragma unroll
    for (int t1 = 0; t1 < TMAX; ++t1) {
      if (t1 < T) {
        const float a = L.m[t1][t2];
        colsum += a;
#pragma unroll
        for (int j = 0; j < 16; ++j) xm[j] += a * xs[t1][j];
      }
    }
    f16x8 bf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) bf[ks][j] = (f16)xm[8 * ks + j];
    const size_t pix = (size_t)(b * T + t2) * HW + pc;   // masked lanes: clamped address, store suppressed
    // proj1's output rows are PERMUTED at packing time (packing.gagg_row_perm): row 4 kq + i of output tile o is channel
    // oc + i with oc = 32 (o >> 1) + 8 kq + 4 (o & 1) - one of the 16 channels this lane already holds in xs (the residual
    // comes from registers; it used to be four dependent L2 reads per frame, the kernel's longest wait)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma_16x16x32(wf[o][0], bf[0], acc);
      acc = mfma_16x16x32(wf[o][1], bf[1], acc);
      const int oc = 32 * (o >> 1) + 8 * kq + 4 * (o & 1);
      const int xi = 8 * (o >> 1) + 4 * (o & 1);          // index of channel oc in xs[.][16]
      const float xr[4] = {xs[t2][xi], xs[t2][xi + 1], xs[t2][xi + 2], xs[t2][xi + 3]};      // t2 is unrolled: plain registers
      const float4 bb = *reinterpret_cast<const float4*>(b1 + oc);
      const float4 r = make_float4(xr[0] + acc[0] + bb.x * colsum, xr[1] + acc[1] + bb.y * colsum,
                                   xr[2] + acc[2] + bb.z * colsum, xr[3] + acc[3] + bb.w * colsum);
      if (pvalid) {
        if constexpr (DENSE) {
          uint2 h;
          h.x = pack2(r.x, r.y);
          h.y = pack2(r.z, r.w);
          *reinterpret_cast<uint2*>(dense + (size_t)(oc >> 5) * plane + pix * 32 + (oc & 31)) = h;
        } else {
          *reinterpret_cast<float4*>(y + pix * 64 + oc) = r;
        }
      }
    }
  }
}

// ---- pointwise (1x1x1) conv: out[px][o] = act( sum_k W[o][k] * in_act(in[px][k]) + b[o] )
// Workgroup = 4 waves x 2 pixel tiles of 16; the input fragments stay in registers, the weights
// stream through LDS in blocks of <= 64 KiB of fragments.
template <int KS, bool IN_F32, bool OUT_F32>
__global__ __launch_bounds__(256) void pwconv_kernel(const void* __restrict__ in, void* __restrict__ out,
                                                     const f16* __restrict__ w, const float* __restrict__ bias,
                                                     size_t npix, int cin, int ot_total, int cout_stride,
                                                     int lrelu_in, int lrelu_out) {
  constexpr int MT = 2;
  constexpr int OTB = 64 / KS;                       // output tiles per LDS block
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4;
  const size_t p0 = ((size_t)blockIdx.x * 4 + wave) * (16 * MT);
  f16x8 bf[MT][KS];
  size_t pl[MT];
  bool pv[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    pl[m] = p0 + m * 16 + (lane & 15);
    pv[m] = pl[m] < npix;
    const size_t pc = pv[m] ? pl[m] : npix - 1;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (IN_F32) {
        const float* p = reinterpret_cast<const float*>(in) + pc * cin + ks * 32 + kq * 8;
        const float4 v0 = *reinterpret_cast<const float4*>(p);
        const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) bf[m][ks][j] = (f16)(lrelu_in == 1 ? lrelu02(v[j]) : lrelu_in == 2 ? fmaxf(v[j], 0.f) : v[j]);
      } else {
        bf[m][ks] = *reinterpret_cast<const f16x8*>(reinterpret_cast<const f16*>(in) + pc * cin + ks * 32 + kq * 8);
      }
    }
  }
  for (int ob = 0; ob < ot_total; ob += OTB) {
    const int nt = min(OTB, ot_total - ob);
    __syncthreads();
    {
      const u32x4* src = reinterpret_cast<const u32x4*>(w) + (size_t)ob * KS * 64;
      for (int i = tid; i < nt * KS * 64; i += 256) *reinterpret_cast<u32x4*>(smem + (size_t)i * 16) = src[i];
    }
    __syncthreads();
    for (int o = 0; o < nt; ++o) {
      f32x4 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f16x8 af = *reinterpret_cast<const f16x8*>(smem + ((size_t)(o * KS + ks) * 64 + lane) * 16);
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32(af, bf[m][ks], acc[m]);
      }
      const int oc = (ob + o) * 16 + kq * 4;
      const float4 bb = *reinterpret_cast<const float4*>(bias + oc);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (!pv[m]) continue;
        float v[4] = {acc[m][0] + bb.x, acc[m][1] + bb.y, acc[m][2] + bb.z, acc[m][3] + bb.w};
        if (lrelu_out == 1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = lrelu02(v[j]);
        } else if (lrelu_out == 2) {          // ReLU (the gmm_thin head, SelfC_GMM_arch_inv.py:345-354)
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (OUT_F32) {
          *reinterpret_cast<float4*>(reinterpret_cast<float*>(out) + pl[m] * cout_stride + oc) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 u;
          u.x = pack2(v[0], v[1]);
          u.y = pack2(v[2], v[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<f16*>(out) + pl[m] * cout_stride + oc) = u;
        }
      }
    }
  }
}

// ---- the whole GMM head + the GMM sample in one kernel (SelfC_GMM_arch_inv.py:327-344,371-394, eval / sampling path) -----
//   feat (fp32, 64) -lrelu-> conv 64->128 -act-> conv 128->256 -act-> conv 256->720 -> sample v (48)
// Layer by layer the head wrote and re-read its 128- and 256-channel activations (pixel-major f16 rows) and, before the
// sampler was fused, the 720-channel output (578 MB fp32 at 4 x 7 x 64 x 112 pixels).  Here a wave keeps its 32 pixels in
// registers from the input row to the sample:
//  * layers 0 and 1: the OUTPUT ROWS of W0 / W1 are permuted at packing time (packing.py: head_row_perm) so that the MFMA
//    result of tile pair (2 s, 2 s + 1) - lane (pixel, kq) holds rows 4 kq + e of each - IS the next layer's B fragment of
//    k-step s (channels 32 s + 8 kq + 0..7): bias, activation, one pack, no cross-lane movement, no LDS round trip;
//  * layer 2: output channels permuted to [k][pi | log-sigma | mu][c] (group g = 3 k + j of 48 channels = 3 MFMA tiles), so
//    that for each mixture component k the softmax over the hf axis c, the clamp / exp of log-sigma and the weighted sum all
//    happen on the accumulators: lane (pixel = lane & 15, kq = lane >> 4) owns c = 16 i + 4 kq + e of tile i; the softmax's
//    max / sum cross the four kq lane groups by two xor shuffles.
// eps rows are [px][k * 48 + c] (k-major, unlike selfc_gmm_sample's c-major rows); v rows have stride vstride.
// Weights: ONE fragment stream [W0: 8 tiles x 2 | W1: 16 tiles x 4 | W2: 45 tiles x 8] = 440 KiB through an 80-KiB LDS buffer
// (W0 and W1 together, then W2 ten tiles at a time).  NW waves x 2 pixel tiles of 16 per workgroup: every workgroup streams
// the whole set, so the L2 -> LDS weight traffic is inversely proportional to the pixels per workgroup (4 waves: 421 us for
// the last layer alone, 8 waves: 207 us).
template <int K, int NW, int MT>
__global__ __launch_bounds__(NW * 64) void stp_head_gmm_kernel(const float* __restrict__ feat, const f16* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ eps, float* __restrict__ v, size_t npix, int vstride, int act) {
  constexpr int HF = 48, TI = HF / 16;
  constexpr int KS0 = 2, KS1 = 4, KS2 = 8;           // k-steps of 32 channels: 64 -> 128 -> 256 -> 720
  constexpr int OT0 = 8, OT1 = 16;
  constexpr int F01 = OT0 * KS0 + OT1 * KS1;        // 80 fragments: W0 | W1
  constexpr int NT = 3 * K * TI;                     // 45 output tiles of layer 2
  // The stream moves in CHUNKS of 40 fragments (40 KiB: 5 tiles of layer 2; W0 | W1 = chunks 0, 1) through a ring of three LDS
  // buffers, copied by global_load_lds (no VGPRs - the kernel sits at 246): chunk c+2 is requested when chunk c starts, a
  // counted vmcnt + one raw barrier per chunk (cdna_hip_programming.md, "Pipelining across barriers").  With one 80-KiB buffer
  // filled through registers between two __syncthreads the only workgroup of the CU stood still for every refill.
  constexpr int CH = 40, OTB = CH / KS2, NCH = (F01 + NT * KS2) / CH, GL = CH / NW;
  static_assert(F01 == 2 * CH && (NT * KS2) % CH == 0 && CH % NW == 0, "chunking");
  constexpr int NBIAS = (OT0 + OT1 + NT) * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4;
  const size_t p0 = ((size_t)blockIdx.x * NW + wave) * (16 * MT);
  size_t pl[MT];
  bool pv[MT];
  f16x8 b0[MT][KS0];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    pl[m] = p0 + m * 16 + (lane & 15);
    pv[m] = pl[m] < npix;
    const size_t pc = pv[m] ? pl[m] : npix - 1;
    pl[m] = pc;
#pragma unroll
    for (int ks = 0; ks < KS0; ++ks) {
      const float* p = feat + pc * 64 + ks * 32 + kq * 8;
      const float4 v0 = *reinterpret_cast<const float4*>(p);
      const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
      const u32x4 u = {pack2(lrelu02(v0.x), lrelu02(v0.y)), pack2(lrelu02(v0.z), lrelu02(v0.w)),
                       pack2(lrelu02(v1.x), lrelu02(v1.y)), pack2(lrelu02(v1.z), lrelu02(v1.w))};
      b0[m][ks] = __builtin_bit_cast(f16x8, u);
    }
  }
  auto stage = [&](const int c) __attribute__((always_inline)) {        // chunk c -> buffer c % 3: GL requests per thread
    const f16* src = w + (size_t)c * CH * 512 + lane * 8;
    unsigned char* dst = smem + (c % 3) * (CH * 1024);
#pragma unroll
    for (int i = 0; i < GL; ++i) {
      const int f = i * NW + wave;                                         // one fragment (1 KiB) per wave and request
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)f * 512),
                                       (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
    }
  };
  // the 1,104 biases live in LDS behind the ring: an ordinary global load per tile would make hipcc drain the whole VM queue
  // (vmcnt(0) at its first use while LDS-DMA requests are in flight) and with it the prefetched chunks
  float* __restrict__ lbias = reinterpret_cast<float*>(smem + 3 * CH * 1024);
  for (int i = tid; i < NBIAS / 4; i += NW * 64) reinterpret_cast<float4*>(lbias)[i] = reinterpret_cast<const float4*>(bias)[i];
  stage(0);
  stage(1);
  stage(2);
  // chunks 0, 1 (W0 | W1) landed, chunk 2 in flight; the lbias ds_writes above retired (lgkmcnt) BEFORE the raw barrier - a
  // bare s_barrier waits for no counter, so without it a wave could pass with its bias rows still in flight
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(GL) : "memory");
  __builtin_amdgcn_s_barrier();

  const float slope = act == 2 ? 0.f : 0.2f;
  // one tile of layer L (KSL k-steps over the fragments at LDS offset fbase), bias, activation, packed to two dwords
  auto dense_tile = [&](auto ksl, const int fbase, const int t, const float* __restrict__ bl, const f16x8 (&bin)[MT][decltype(ksl)::value],
                        uint2 (&res)[MT]) __attribute__((always_inline)) {
    constexpr int KSL = decltype(ksl)::value;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KSL; ++ks) {
      const f16x8 af = *reinterpret_cast<const f16x8*>(smem + ((size_t)(fbase + t * KSL + ks) * 64 + lane) * 16);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32(af, bin[m][ks], acc[m]);
    }
    const float4 bb = *reinterpret_cast<const float4*>(bl + t * 16 + kq * 4);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float r[4] = {acc[m][0] + bb.x, acc[m][1] + bb.y, acc[m][2] + bb.z, acc[m][3] + bb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], r[e] * slope);        // LeakyReLU(0.2) | ReLU
      res[m].x = pack2(r[0], r[1]);
      res[m].y = pack2(r[2], r[3]);
    }
  };

  f16x8 b1[MT][KS1];
#pragma unroll
  for (int s = 0; s < KS1; ++s) {
    uint2 lo[MT], hi[MT];
    dense_tile(std::integral_constant<int, KS0>{}, 0, 2 * s, lbias, b0, lo);
    dense_tile(std::integral_constant<int, KS0>{}, 0, 2 * s + 1, lbias, b0, hi);
#pragma unroll
    for (int m = 0; m < MT; ++m) b1[m][s] = __builtin_bit_cast(f16x8, u32x4{lo[m].x, lo[m].y, hi[m].x, hi[m].y});
  }
  f16x8 bf[MT][KS2];
#pragma unroll
  for (int s = 0; s < KS2; ++s) {
    uint2 lo[MT], hi[MT];
    dense_tile(std::integral_constant<int, KS1>{}, OT0 * KS0, 2 * s, lbias + OT0 * 16, b1, lo);
    dense_tile(std::integral_constant<int, KS1>{}, OT0 * KS0, 2 * s + 1, lbias + OT0 * 16, b1, hi);
#pragma unroll
    for (int m = 0; m < MT; ++m) bf[m][s] = __builtin_bit_cast(f16x8, u32x4{lo[m].x, lo[m].y, hi[m].x, hi[m].y});
    __builtin_amdgcn_sched_barrier(0);
  }

  // LDS byte address of this lane's four biases of layer-2 tile 0
  const unsigned bias2_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + 3 * CH * 1024 + ((OT0 + OT1) * 16 + kq * 4) * 4;
  f32x4 out[MT][TI], pi[MT][TI], sg[MT][TI];
  float4 ep[MT][TI];
  float mx[MT], inv[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < TI; ++i) out[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto tile = [&](const int gt, f32x4 (&acc)[MT]) __attribute__((always_inline)) {
    const int c = 2 + gt / OTB;                      // chunk of this tile (workgroup-uniform)
    if (gt % OTB == 0) {
      // own requests of chunk c retired (those of chunk c+1 may stay in flight), then the barrier: everybody's part of
      // chunk c is in LDS and everybody is done reading chunk c-1, whose buffer chunk c+2 takes over
      if (c == 2 || c + 1 >= NCH) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(GL) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (c == 2) { stage(3); stage(4); }            // buffers 0, 1 held W0 | W1 until this barrier
      else if (c + 2 < NCH) stage(c + 2);
    }
    const unsigned char* buf = smem + (c % 3) * (CH * 1024);
    const int o = gt % OTB;
    __builtin_amdgcn_sched_barrier(0);
    // The tile's bias row, read by hand: an LDS load hipcc can see gets an s_waitcnt vmcnt(0) in front of it whenever LDS-DMA
    // requests are pending (it cannot tell the ring from the bias rows), which drained the chunk requested a few lines up
    // right at every chunk boundary (the weight fragments' own reads carry no memory operand and escape that rule).
    f32x4 bb;
    asm volatile("ds_read_b128 %0, %1" : "=v"(bb) : "v"(bias2_lds + (unsigned)(gt * 64)));
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      const f16x8 af = *reinterpret_cast<const f16x8*>(buf + ((size_t)(o * KS2 + ks) * 64 + lane) * 16);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m] = mfma_16x16x32(af, bf[m][ks], acc[m]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb));
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] += bb;
    __builtin_amdgcn_sched_barrier(0);               // one tile at a time: the unrolled 45 tiles' reads hoisted together spilled 366 registers
  };

  // a real loop over the mixture components (fully unrolled, the 45 tiles' bias / noise loads were hoisted to the top
  // and the input fragments lived in scratch)
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < TI; ++i) ep[m][i] = *reinterpret_cast<const float4*>(eps + pl[m] * (HF * K) + k * HF + i * 16 + kq * 4);
    // pi logits: softmax over the 48 values of c (per pixel)
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      f32x4 acc[MT];
      tile((3 * k + 0) * TI + i, acc);
#pragma unroll
      for (int m = 0; m < MT; ++m) pi[m][i] = acc[m];
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      float a = pi[m][0][0];
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) a = fmaxf(a, pi[m][i][e]);
      a = fmaxf(a, __shfl_xor(a, 16, 64));
      a = fmaxf(a, __shfl_xor(a, 32, 64));
      mx[m] = a;
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { pi[m][i][e] = __builtin_amdgcn_exp2f(1.44269504f * (pi[m][i][e] - a)); sum += pi[m][i][e]; }
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      inv[m] = __builtin_amdgcn_rcpf(sum);
    }
    // log-sigma: clamp(-7, 7), exp
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      f32x4 acc[MT];
      tile((3 * k + 1) * TI + i, acc);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int e = 0; e < 4; ++e) sg[m][i][e] = __builtin_amdgcn_exp2f(1.44269504f * fminf(fmaxf(acc[m][e], -7.f), 7.f));
    }
    // mu: v[c] += pi (eps sigma + mu)
#pragma unroll
    for (int i = 0; i < TI; ++i) {
      f32x4 acc[MT];
      tile((3 * k + 2) * TI + i, acc);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const float e4[4] = {ep[m][i].x, ep[m][i].y, ep[m][i].z, ep[m][i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) out[m][i][e] += (pi[m][i][e] * inv[m]) * (e4[e] * sg[m][i][e] + acc[m][e]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    if (!pv[m]) continue;
#pragma unroll
    for (int i = 0; i < TI; ++i)
      *reinterpret_cast<float4*>(v + pl[m] * vstride + i * 16 + kq * 4) = make_float4(out[m][i][0], out[m][i][1], out[m][i][2], out[m][i][3]);
  }
}

// ---- GMM sample (SelfC_GMM_arch_inv.py:382-394): raw[px][c*K*3 + k*3 + {pi-logit, log-sigma, mu}],
// pi = softmax over the hf_dim axis c (per k), v[c] = sum_k pi (eps * exp(clamp(ls,-7,7)) + mu).
// 16 lanes per pixel, lane j owns c in {j, j+16, j+32}.
template <int K>
__global__ __launch_bounds__(256) void gmm_sample_kernel(const float* __restrict__ raw, const float* __restrict__ eps,
                                                         float* __restrict__ v, size_t npix) {
  constexpr int HF = 48, CP = HF / 16;
  const size_t px = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int j = threadIdx.x & 15;
  const bool ok = px < npix;
  const size_t pc = ok ? px : npix - 1;
  float r[CP][K * 3], e[CP][K];
#pragma unroll
  for (int i = 0; i < CP; ++i) {
    const int c = j + 16 * i;
    const float* p = raw + pc * (HF * K * 3) + (size_t)c * K * 3;
#pragma unroll
    for (int q = 0; q < K * 3; ++q) r[i][q] = p[q];
    const float* pe = eps + pc * (HF * K) + (size_t)c * K;
#pragma unroll
    for (int q = 0; q < K; ++q) e[i][q] = pe[q];
  }
  float out[CP];
#pragma unroll
  for (int i = 0; i < CP; ++i) out[i] = 0.f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float mx = r[0][3 * k];
#pragma unroll
    for (int i = 1; i < CP; ++i) mx = fmaxf(mx, r[i][3 * k]);
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 16));
    float ex[CP], s = 0.f;
#pragma unroll
    for (int i = 0; i < CP; ++i) { ex[i] = expf(r[i][3 * k] - mx); s += ex[i]; }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) s += __shfl_xor(s, d, 16);
#pragma unroll
    for (int i = 0; i < CP; ++i) {
      const float ls = fminf(fmaxf(r[i][3 * k + 1], -7.f), 7.f);
      out[i] += (ex[i] / s) * (e[i][k] * expf(ls) + r[i][3 * k + 2]);
    }
  }
  if (ok) {
#pragma unroll
    for (int i = 0; i < CP; ++i) v[px * HF + j + 16 * i] = out[i];
  }
}

// ---- the same sampler for any (hf_dim, K) and either std convention: one thread per pixel (the Haar variant's head has
// hf_dim = 9: SelfC_arch_inv.py:151-162, std = exp(0.5 logvar) there, :179-186).  raw rows have stride raw_stride.
__global__ __launch_bounds__(256) void gmm_sample_generic_kernel(const float* __restrict__ raw, const float* __restrict__ eps,
                                                                 float* __restrict__ v, size_t npix, int hf, int K, int raw_stride,
                                                                 int v_stride, float ls_scale) {
  const size_t px = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (px >= npix) return;
  const float* r = raw + px * (size_t)raw_stride;
  const float* e = eps + px * (size_t)(hf * K);
  float* o = v + px * (size_t)v_stride;
  for (int c = 0; c < hf; ++c) o[c] = 0.f;
  for (int k = 0; k < K; ++k) {
    float mx = r[k * 3];
    for (int c = 1; c < hf; ++c) mx = fmaxf(mx, r[(c * K + k) * 3]);
    float s = 0.f;
    for (int c = 0; c < hf; ++c) s += expf(r[(c * K + k) * 3] - mx);
    for (int c = 0; c < hf; ++c) {
      const float* q = r + (c * K + k) * 3;
      const float ls = fminf(fmaxf(q[1], -7.f), 7.f);
      o[c] += (expf(q[0] - mx) / s) * (e[c * K + k] * expf(ls_scale * ls) + q[2]);
    }
  }
}

// gradient of gmm_sample_generic_kernel w.r.t. raw (any hf_dim, K, raw / dv row strides, log-sigma scale): one thread per
// pixel; pad columns of a raw row (beyond hf*K*3) receive 0
__global__ __launch_bounds__(256) void gmm_sample_generic_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ eps,
                                                                     const float* __restrict__ dv, float* __restrict__ draw, size_t npix,
                                                                     int hf, int K, int raw_stride, int v_stride, float ls_scale) {
  const size_t px = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (px >= npix) return;
  const float* r = raw + px * (size_t)raw_stride;
  const float* e = eps + px * (size_t)(hf * K);
  const float* g = dv + px * (size_t)v_stride;
  float* o = draw + px * (size_t)raw_stride;
  for (int i = hf * K * 3; i < raw_stride; ++i) o[i] = 0.f;
  for (int k = 0; k < K; ++k) {
    float mx = r[k * 3];
    for (int c = 1; c < hf; ++c) mx = fmaxf(mx, r[(c * K + k) * 3]);
    float s = 0.f;
    for (int c = 0; c < hf; ++c) s += expf(r[(c * K + k) * 3] - mx);
    float dot = 0.f;
    for (int c = 0; c < hf; ++c) {
      const float* q = r + (c * K + k) * 3;
      const float ls = fminf(fmaxf(q[1], -7.f), 7.f);
      dot += (expf(q[0] - mx) / s) * g[c] * (e[c * K + k] * expf(ls_scale * ls) + q[2]);
    }
    for (int c = 0; c < hf; ++c) {
      const float* q = r + (c * K + k) * 3;
      const float ls = fminf(fmaxf(q[1], -7.f), 7.f), sg = expf(ls_scale * ls), pi = expf(q[0] - mx) / s;
      float* d = o + (c * K + k) * 3;
      d[0] = pi * (g[c] * (e[c * K + k] * sg + q[2]) - dot);
      d[1] = (q[1] >= -7.f && q[1] <= 7.f) ? g[c] * pi * e[c * K + k] * sg * ls_scale : 0.f;
      d[2] = g[c] * pi;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// gradients (training): GMM sampler, GlobalAgg, small elementwise helpers
// ---------------------------------------------------------------------------------------------------------
// d raw of v[c] = sum_k pi[c][k] (eps exp(clamp(ls)) + mu), pi = softmax over c: same lane layout as the sampler
template <int K>
__global__ __launch_bounds__(256) void gmm_sample_bwd_kernel(const float* __restrict__ raw, const float* __restrict__ eps,
                                                             const float* __restrict__ dv, float* __restrict__ draw, size_t npix) {
  constexpr int HF = 48, CP = HF / 16;
  const size_t px = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int j = threadIdx.x & 15;
  const bool ok = px < npix;
  const size_t pc = ok ? px : npix - 1;
  float r[CP][K * 3], e[CP][K], g[CP];
#pragma unroll
  for (int i = 0; i < CP; ++i) {
    const int c = j + 16 * i;
    const float* p = raw + pc * (HF * K * 3) + (size_t)c * K * 3;
#pragma unroll
    for (int q = 0; q < K * 3; ++q) r[i][q] = p[q];
    const float* pe = eps + pc * (HF * K) + (size_t)c * K;
#pragma unroll
    for (int q = 0; q < K; ++q) e[i][q] = pe[q];
    g[i] = dv[pc * HF + c];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float mx = r[0][3 * k];
#pragma unroll
    for (int i = 1; i < CP; ++i) mx = fmaxf(mx, r[i][3 * k]);
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) mx = fmaxf(mx, __shfl_xor(mx, d, 16));
    float ex[CP], s = 0.f;
#pragma unroll
    for (int i = 0; i < CP; ++i) { ex[i] = expf(r[i][3 * k] - mx); s += ex[i]; }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) s += __shfl_xor(s, d, 16);
    float pi[CP], dpi[CP], sg[CP], dot = 0.f;
#pragma unroll
    for (int i = 0; i < CP; ++i) {
      pi[i] = ex[i] / s;
      sg[i] = expf(fminf(fmaxf(r[i][3 * k + 1], -7.f), 7.f));
      dpi[i] = g[i] * (e[i][k] * sg[i] + r[i][3 * k + 2]);
      dot += pi[i] * dpi[i];
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) dot += __shfl_xor(dot, d, 16);
    if (ok) {
#pragma unroll
      for (int i = 0; i < CP; ++i) {
        float* o = draw + px * (HF * K * 3) + (size_t)(j + 16 * i) * K * 3 + 3 * k;
        const float lsr = r[i][3 * k + 1];
        o[0] = pi[i] * (dpi[i] - dot);
        o[1] = (lsr >= -7.f && lsr <= 7.f) ? g[i] * pi[i] * e[i][k] * sg[i] : 0.f;
        o[2] = g[i] * pi[i];
      }
    }
  }
}

__global__ __launch_bounds__(256) void lrelu_bwd_kernel(float4* __restrict__ dx, const float4* __restrict__ x, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 d = dx[i];
  const float4 v = x[i];
  d.x *= v.x > 0.f ? 1.f : 0.2f; d.y *= v.y > 0.f ? 1.f : 0.2f; d.z *= v.z > 0.f ? 1.f : 0.2f; d.w *= v.w > 0.f ? 1.f : 0.2f;
  dx[i] = d;
}

// f16 rows [npix][C] -> f16 planes [C/32][npix][32] (C a multiple of 32): item = (pixel, 16-byte chunk)
__global__ __launch_bounds__(256) void rows_to_planes_kernel(const f16* __restrict__ rows, f16* __restrict__ planes, size_t npix, int C) {
  const int cpp = C / 8;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix * (size_t)cpp) return;
  const int chunk = (int)(i % cpp);
  const size_t pix = i / cpp;
  const int ch0 = chunk * 8;
  *reinterpret_cast<u32x4*>(planes + (size_t)(ch0 >> 5) * npix * 32 + pix * 32 + (ch0 & 31)) =
      *reinterpret_cast<const u32x4*>(rows + pix * C + ch0);
}

// GlobalAgg backward, step 1: per clip and 512-pixel chunk, dAx[t1][t2] = sum_{px,c} x[t1] dz[t2] and dyo[t][o] = sum_px dy[t][o]
template <int TT>
__global__ __launch_bounds__(256) void gagg_bwd_reduce_kernel(const float* __restrict__ x, const float* __restrict__ dz, const float* __restrict__ dy,
                                                              float* __restrict__ pdA, float* __restrict__ pdyo, int Trt, int HW, int nchunk) {
  const int T = TT ? TT : Trt;
  __shared__ float redA[16][TMAX * TMAX + 1];
  __shared__ float redY[16][TMAX][64];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int cq = threadIdx.x & 15, pr = threadIdx.x >> 4;
  const int p0 = chunk * BWD_CHUNK, p1 = min(p0 + BWD_CHUNK, HW);
  float accA[TMAX][TMAX];
  float4 accY[TMAX];
#pragma unroll
  for (int t1 = 0; t1 < TMAX; ++t1) {
    accY[t1] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t2 = 0; t2 < TMAX; ++t2) accA[t1][t2] = 0.f;
  }
  for (int p = p0 + pr; p < p1; p += 16) {
    float4 xv[TMAX], zv[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) {
      if (t < T) {
        const size_t o = ((size_t)(b * T + t) * HW + p) * 64 + cq * 4;
        xv[t] = *reinterpret_cast<const float4*>(x + o);
        zv[t] = *reinterpret_cast<const float4*>(dz + o);
        const float4 d = *reinterpret_cast<const float4*>(dy + o);
        accY[t].x += d.x; accY[t].y += d.y; accY[t].z += d.z; accY[t].w += d.w;
      } else {
        xv[t] = zv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int t1 = 0; t1 < TMAX; ++t1)
#pragma unroll
      for (int t2 = 0; t2 < TMAX; ++t2)
        accA[t1][t2] += xv[t1].x * zv[t2].x + xv[t1].y * zv[t2].y + xv[t1].z * zv[t2].z + xv[t1].w * zv[t2].w;
  }
  // sum over the 16 channel-quad lanes (consecutive lanes), then over the 16 pixel rows through LDS
#pragma unroll
  for (int t1 = 0; t1 < TMAX; ++t1)
#pragma unroll
    for (int t2 = 0; t2 < TMAX; ++t2) {
      float v = accA[t1][t2];
#pragma unroll
      for (int d = 1; d < 16; d <<= 1) v += __shfl_xor(v, d, 16);
      if (cq == 0) redA[pr][t1 * TMAX + t2] = v;
    }
#pragma unroll
  for (int t = 0; t < TMAX; ++t) *reinterpret_cast<float4*>(&redY[pr][t][cq * 4]) = accY[t];
  __syncthreads();
  const size_t ob = (size_t)b * nchunk + chunk;
  if (threadIdx.x < TMAX * TMAX) {
    float v = 0.f;
    for (int r = 0; r < 16; ++r) v += redA[r][threadIdx.x];
    pdA[ob * 64 + threadIdx.x] = v;
  }
  for (int i = threadIdx.x; i < TMAX * 64; i += 256) {
    const int t = i >> 6, c = i & 63;
    float v = 0.f;
    for (int r = 0; r < 16; ++r) v += redY[r][t][c];
    pdyo[(ob * TMAX + t) * 64 + c] = v;
  }
}

// step 2: per clip - recompute g, q, k, A; then the gradients of everything upstream of A (tiny)
// TT: the clip length as a compile-time constant (0: run time).  With a run-time T every per-frame statement of this kernel sat
// behind its own `t < T` branch - 367 branches, as many drained wait counters, no load issued ahead of its use: 80 us for one wave.
template <int TT>
__global__ __launch_bounds__(64) void gagg_attn_bwd_kernel(const float* __restrict__ partial, int nchunk, int nchunkb, const float* __restrict__ fcbp,
                                                           const float* __restrict__ w2, const float* __restrict__ b2,
                                                           const float* __restrict__ w3, const float* __restrict__ b3,
                                                           const float* __restrict__ b1, const float* __restrict__ pdA,
                                                           const float* __restrict__ pdyo, float* __restrict__ A, float* __restrict__ dg,
                                                           float* __restrict__ db1c, float* __restrict__ dw2c, float* __restrict__ db2c,
                                                           float* __restrict__ dw3c, float* __restrict__ db3c, float* __restrict__ dfcbc, int Trt) {
  const int T = TT ? TT : Trt;
  __shared__ float g[TMAX][64], q[TMAX][64], k[TMAX][64], m[TMAX][TMAX], a[TMAX][TMAX], dA[TMAX][TMAX], dm[TMAX][TMAX];
  __shared__ float dyo[TMAX][64], dq[TMAX][64], dk[TMAX][64], dyb[TMAX], red[64];
  // one wave per clip and nothing to hide latency behind: everything that comes from global memory is fetched up front with
  // independent wide loads and then lives in LDS / registers (a dependent global load per loop iteration - proj2 / proj3 columns,
  // b1, the chunk sums - made this launch 80 us on the training step's main queue)
  __shared__ __attribute__((aligned(16))) float w2s[64][64], w3s[64][64];
  __shared__ float b1s[64];
  const int b = blockIdx.x, c = threadIdx.x;
  const float fcb = *fcbp;
  float4 w2r[16], w3r[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    w2r[j] = *reinterpret_cast<const float4*>(w2 + c * 64 + 4 * j);
    w3r[j] = *reinterpret_cast<const float4*>(w3 + c * 64 + 4 * j);
  }
  b1s[c] = b1[c];
  // the chunk sums: every frame's loads are issued together (eight partial accumulators per frame; one load - wait - add at a time
  // over ~35 chunks x 7 frames was most of this launch's 80 us)
  {
    float sg[TMAX], sy[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) sg[t] = sy[t] = 0.f;
#pragma unroll 4
    for (int j = 0; j < nchunk; ++j) {
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < T) sg[t] += partial[((size_t)(b * T + t) * nchunk + j) * 64 + c];
    }
#pragma unroll 4
    for (int j = 0; j < nchunkb; ++j) {
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < T) sy[t] += pdyo[(((size_t)b * nchunkb + j) * TMAX + t) * 64 + c];
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < T) { g[t][c] = sg[t] + fcb; dyo[t][c] = sy[t]; }
  }
  __syncthreads();
  // this thread's rows of proj2 / proj3 are read ONCE, as 16-byte loads, and stay in registers (they used to be re-read
  // scalar by scalar for every frame: ~1,800 dependent loads per thread made this 8-wave launch 78 us of pure latency)
#pragma unroll
  for (int j = 0; j < 16; ++j) {                  // rows -> LDS: the columns (d g) are read from there
    *reinterpret_cast<float4*>(&w2s[c][4 * j]) = w2r[j];
    *reinterpret_cast<float4*>(&w3s[c][4 * j]) = w3r[j];
  }
  {
    const float bq = b2[c], bk = b3[c];
    for (int t = 0; t < T; ++t) {
      float sq = bq, sk = bk;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float4 gv = *reinterpret_cast<const float4*>(&g[t][4 * j]);
        sq += gv.x * w2r[j].x + gv.y * w2r[j].y + gv.z * w2r[j].z + gv.w * w2r[j].w;
        sk += gv.x * w3r[j].x + gv.y * w3r[j].y + gv.z * w3r[j].z + gv.w * w3r[j].w;
      }
      q[t][c] = sq;
      k[t][c] = sk;
    }
  }
  __syncthreads();
  if (c < T * T) {
    const int t1 = c / T, t2 = c % T;
    float s = 0.f;
    for (int j = 0; j < 64; ++j) s += q[t1][j] * k[t2][j];
    m[t1][t2] = s / 64.0f;
    float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
    int j = 0;
    for (; j + 3 < nchunkb; j += 4) {
      d0 += pdA[((size_t)b * nchunkb + j) * 64 + t1 * TMAX + t2];
      d1 += pdA[((size_t)b * nchunkb + j + 1) * 64 + t1 * TMAX + t2];
      d2 += pdA[((size_t)b * nchunkb + j + 2) * 64 + t1 * TMAX + t2];
      d3 += pdA[((size_t)b * nchunkb + j + 3) * 64 + t1 * TMAX + t2];
    }
    for (; j < nchunkb; ++j) d0 += pdA[((size_t)b * nchunkb + j) * 64 + t1 * TMAX + t2];
    dA[t1][t2] = (d0 + d1) + (d2 + d3);
  }
  if (c < T) {
    float s = 0.f;
    for (int o = 0; o < 64; ++o) s += dyo[c][o] * b1s[o];
    dyb[c] = s;
  }
  __syncthreads();
  if (c < T) {
    float mx = m[c][0];
    for (int j = 1; j < T; ++j) mx = fmaxf(mx, m[c][j]);
    float e[TMAX], s = 0.f;
    for (int j = 0; j < T; ++j) { e[j] = expf(m[c][j] - mx); s += e[j]; }
    float dot = 0.f;
    for (int j = 0; j < T; ++j) {
      a[c][j] = e[j] / s;
      A[((size_t)b * T + c) * T + j] = a[c][j];
      dot += a[c][j] * (dA[c][j] + dyb[j]);
    }
    for (int j = 0; j < T; ++j) dm[c][j] = a[c][j] * (dA[c][j] + dyb[j] - dot) / 64.0f;
  }
  __syncthreads();
  {
    float s = 0.f;                                  // db1[o] = sum_t2 dyo[t2][o] * colsum[t2]
    for (int t2 = 0; t2 < T; ++t2) {
      float cs = 0.f;
      for (int t1 = 0; t1 < T; ++t1) cs += a[t1][t2];
      s += dyo[t2][c] * cs;
    }
    db1c[(size_t)b * 64 + c] = s;
  }
  float sq2 = 0.f, sk2 = 0.f;
  for (int t = 0; t < T; ++t) {
    float vq = 0.f, vk = 0.f;
    for (int j = 0; j < T; ++j) {
      vq += dm[t][j] * k[j][c];
      vk += dm[j][t] * q[j][c];
    }
    dq[t][c] = vq;
    dk[t][c] = vk;
    sq2 += vq;
    sk2 += vk;
  }
  db2c[(size_t)b * 64 + c] = sq2;
  db3c[(size_t)b * 64 + c] = sk2;
  __syncthreads();
  {
    float dqc[TMAX], dkc[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) { dqc[t] = t < T ? dq[t][c] : 0.f; dkc[t] = t < T ? dk[t][c] : 0.f; }
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {                 // row c of the two outer products, 16 bytes at a time
      float4 v2 = make_float4(0.f, 0.f, 0.f, 0.f), v3 = v2;
#pragma unroll
      for (int t = 0; t < TMAX; ++t) {
        if (t < T) {
          const float4 gv = *reinterpret_cast<const float4*>(&g[t][4 * j]);
          v2.x += dqc[t] * gv.x; v2.y += dqc[t] * gv.y; v2.z += dqc[t] * gv.z; v2.w += dqc[t] * gv.w;
          v3.x += dkc[t] * gv.x; v3.y += dkc[t] * gv.y; v3.z += dkc[t] * gv.z; v3.w += dkc[t] * gv.w;
        }
      }
      *reinterpret_cast<float4*>(dw2c + ((size_t)b * 64 + c) * 64 + 4 * j) = v2;
      *reinterpret_cast<float4*>(dw3c + ((size_t)b * 64 + c) * 64 + 4 * j) = v3;
    }
  }
  float tot = 0.f;
  {
    float vt[TMAX];
#pragma unroll
    for (int t = 0; t < TMAX; ++t) vt[t] = 0.f;
#pragma unroll 8
    for (int o = 0; o < 64; ++o) {                 // column c of proj2 / proj3 from the LDS copy, all frames from one read
      const float a2 = w2s[o][c], a3 = w3s[o][c];
#pragma unroll
      for (int t = 0; t < TMAX; ++t)
        if (t < T) vt[t] += dq[t][o] * a2 + dk[t][o] * a3;
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
      if (t < T) {
        dg[((size_t)b * T + t) * 64 + c] = vt[t];
        tot += vt[t];
      }
  }
  red[c] = tot;
  __syncthreads();
  if (c == 0) {
    float v = 0.f;
    for (int j = 0; j < 64; ++j) v += red[j];
    dfcbc[b] = v;
  }
}

// step 3: dx[t1] = dy[t1] + sum_t2 A[t1][t2] dz[t2] + dg[t1] wmap[px];  dwmap[b][px] = sum_{t,c} dg[t][c] x[t][px][c];
// z[t2] = sum_t1 A[t1][t2] x[t1] as f16 planes (the activation operand of proj1's weight gradient)
template <int TT>
__global__ __launch_bounds__(256) void gagg_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ dz, const float* __restrict__ dy,
                                                          const float* __restrict__ A, const float* __restrict__ dg, const float* __restrict__ wmap,
                                                          float* __restrict__ dx, float* __restrict__ dwmapc, f16* __restrict__ zp, int Trt, int HW, size_t npix_all,
                                                          unsigned* __restrict__ amax_out) {
  const int T = TT ? TT : Trt;
  float bmax = 0.f;                      // max |dx| of this thread (amax_out: the scale of the subnet backward that consumes dx)
  bool bnan = false;
  const int b = blockIdx.y;
  const int cq = threadIdx.x & 15;
  const int p = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool ok = p < HW;
  const int pc = ok ? p : HW - 1;
  const float* Ab = A + (size_t)b * T * T;
  float4 xv[TMAX], zv[TMAX];
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    if (t < T) {
      const size_t o = ((size_t)(b * T + t) * HW + pc) * 64 + cq * 4;
      xv[t] = *reinterpret_cast<const float4*>(x + o);
      zv[t] = *reinterpret_cast<const float4*>(dz + o);
    } else {
      xv[t] = zv[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float wm = wmap[pc];
  float dw = 0.f;
#pragma unroll
  for (int t1 = 0; t1 < TMAX; ++t1) {
    if (t1 < T) {
      const size_t o = ((size_t)(b * T + t1) * HW + pc) * 64 + cq * 4;
      const float4 d = *reinterpret_cast<const float4*>(dy + o);
      const float4 gg = *reinterpret_cast<const float4*>(dg + ((size_t)b * T + t1) * 64 + cq * 4);
      float4 r = make_float4(d.x + gg.x * wm, d.y + gg.y * wm, d.z + gg.z * wm, d.w + gg.w * wm);
      float4 zz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t2 = 0; t2 < TMAX; ++t2) {
        if (t2 < T) {
          const float a12 = Ab[t1 * T + t2], a21 = Ab[t2 * T + t1];
          r.x += a12 * zv[t2].x; r.y += a12 * zv[t2].y; r.z += a12 * zv[t2].z; r.w += a12 * zv[t2].w;
          zz.x += a21 * xv[t2].x; zz.y += a21 * xv[t2].y; zz.z += a21 * xv[t2].z; zz.w += a21 * xv[t2].w;   // z[t1] = sum_t A[t][t1] x[t]
        }
      }
      dw += gg.x * xv[t1].x + gg.y * xv[t1].y + gg.z * xv[t1].z + gg.w * xv[t1].w;
      if (ok) {
        *reinterpret_cast<float4*>(dx + o) = r;
        bmax = fmaxf(fmaxf(bmax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
        bnan |= (r.x != r.x) | (r.y != r.y) | (r.z != r.z) | (r.w != r.w);
        uint2 u;
        u.x = pack2(zz.x, zz.y);
        u.y = pack2(zz.z, zz.w);
        const size_t pix = (size_t)(b * T + t1) * HW + pc;
        *reinterpret_cast<uint2*>(zp + (size_t)(cq >> 3) * npix_all * 32 + pix * 32 + (cq & 7) * 4) = u;
      }
    }
  }
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) dw += __shfl_xor(dw, d, 16);
  if (ok && cq == 0) dwmapc[(size_t)b * HW + p] = dw;
  if (amax_out) {                        // ONE atomic per workgroup (same-address atomics serialise at ~90 per microsecond: one per wave
#pragma unroll                           // of 648 workgroups cost this kernel 27 us), absmax_kernel's convention (NaN -> 0x7fc00000)
    for (int o = 32; o > 0; o >>= 1) bmax = fmaxf(bmax, __shfl_xor(bmax, o));
    __shared__ unsigned smax[4];
    const unsigned bits = __any(bnan) ? 0x7fc00000u : __float_as_uint(bmax);
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned m = max(max(smax[0], smax[1]), max(smax[2], smax[3]));      // non-negative floats and the NaN pattern order as unsigned
      if (m) atomicMax(amax_out, m);
    }
  }
}

// dst_i[j] = beta * dst_i[j] + sum_r src_i[r][j] for up to 8 small row-major matrices in ONE launch: the per-clip partial
// gradients of GlobalAgg summed over the clips straight into their destination (the trainer's flat gradient buffer)
struct RowSumArgs { const float* src[SELFC_ROWSUM_MAX]; float* dst[SELFC_ROWSUM_MAX]; int rows[SELFC_ROWSUM_MAX]; int end[SELFC_ROWSUM_MAX]; float beta[SELFC_ROWSUM_MAX]; int n; };
__global__ __launch_bounds__(256) void rowsum_accum_kernel(const RowSumArgs a) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= a.end[a.n - 1]) return;
  int seg = 0;
#pragma unroll
  for (int i = 0; i < SELFC_ROWSUM_MAX - 1; ++i) seg += (i < a.n - 1 && j >= a.end[i]) ? 1 : 0;
  const int lo = seg ? a.end[seg - 1] : 0, len = a.end[seg] - lo, col = j - lo;
  const float* src = a.src[seg] + col;
  float acc = 0.f;
  for (int r = 0; r < a.rows[seg]; ++r) acc += src[(size_t)r * len];
  float* d = a.dst[seg] + col;
  const float beta = a.beta[seg];
  *d = beta != 0.f ? beta * *d + acc : acc;
}

struct GaggBwdLayout {
  size_t plane_b, off_dyp, off_zp, off_dz, off_amax, off_pool, off_A, off_pdA, off_pdyo, off_dg, off_wg, total;
  int nchunk, nchunkb;
};

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

GaggBwdLayout gagg_bwd_layout(int N, int T, int H, int W) {
  GaggBwdLayout L{};
  const int HW = H * W, B = N / T;
  L.nchunk = (HW + POOL_CHUNK - 1) / POOL_CHUNK;
  L.nchunkb = (HW + BWD_CHUNK - 1) / BWD_CHUNK;
  L.plane_b = (size_t)N * HW * 64;
  size_t o = 0;
  L.off_dyp = o; o += 2 * L.plane_b;
  L.off_zp = o; o += 2 * L.plane_b;
  L.off_dz = o; o += 4 * L.plane_b;
  L.off_amax = o; o += 256;
  L.off_pool = o; o = up256(o + (size_t)N * L.nchunk * 64 * 4);
  L.off_A = o; o = up256(o + (size_t)B * T * T * 4);
  L.off_pdA = o; o = up256(o + (size_t)B * L.nchunkb * 64 * 4);
  L.off_pdyo = o; o = up256(o + (size_t)B * L.nchunkb * TMAX * 64 * 4);
  L.off_dg = o; o = up256(o + (size_t)N * 64 * 4);
  L.off_wg = o; o = up256(o + bwd_wgrad_scratch_bytes(N, H, W, 2, 2, 1));
  L.total = o;
  return L;
}

template <int KS, bool IN_F32, bool OUT_F32>
int launch_pw(const void* in, void* out, const void* w, const float* bias, size_t npix, int cin, int cout,
              int cout_stride, int lrelu_in, int lrelu_out, hipStream_t s) {
  constexpr int lds = 64 * 1024;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&pwconv_kernel<KS, IN_F32, OUT_F32>), lds, optin); e != hipSuccess) return hip_rc(e);
  ProfScope prof(PROF_STP, s);
  const unsigned grid = (unsigned)((npix + 127) / 128);
  hipLaunchKernelGGL((pwconv_kernel<KS, IN_F32, OUT_F32>), dim3(grid), dim3(256), lds, s, in, out, (const f16*)w, bias,
                     npix, cin, (cout + 15) / 16, cout_stride, lrelu_in, lrelu_out);
  return hip_rc(hipGetLastError());
}

}  // namespace

extern "C" {

int selfc_globalagg_run(const float* x, float* y, const float* wmap, const float* fc_bias, const void* w1, const float* b1,
                        const float* w2, const float* b2, const float* w3, const float* b3,
                        float* partial, float* attn, int N, int T, int HW, void* stream) {
  if (!y) return SELFC_EINVAL;
  return selfc_globalagg_run_d(x, y, nullptr, wmap, fc_bias, w1, b1, w2, b2, w3, b3, partial, attn, N, T, HW, 64, stream);
}

int selfc_globalagg_run_d(const float* x, float* y, void* dense_out, const float* wmap, const float* fc_bias, const void* w1,
                          const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                          float* partial, float* attn, int N, int T, int HW, int c_real, void* stream) {
  if (!x || !wmap || !fc_bias || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !partial) return SELFC_EINVAL;
  if ((y == nullptr) == (dense_out == nullptr)) return SELFC_EINVAL;        // exactly one kind of output
  if (N <= 0 || T <= 0 || T > TMAX || N % T || HW <= 0 || x == y || c_real < 1 || c_real > 64) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int nchunk = (HW + POOL_CHUNK - 1) / POOL_CHUNK;
  const int B = N / T;
  {
    ProfScope prof(PROF_STP, s);
    hipLaunchKernelGGL(gagg_pool_kernel, dim3(nchunk, N), dim3(256), 0, s, x, wmap, partial, HW, nchunk);
  }
  int rc = hip_rc(hipGetLastError());
  if (rc) return rc;
  {
    ProfScope prof(PROF_STP, s);
    const int tiles = (HW + 63) / 64;
    const size_t plane = (size_t)N * HW * 32;
#define SELFC_GMIX(D_, TT_, Y_, DN_) hipLaunchKernelGGL((gagg_mix_kernel<D_, TT_>), dim3((unsigned)(tiles * B)), dim3(256), 0, s, x, Y_, DN_, plane, partial, \
                                                     nchunk, fc_bias, w2, b2, w3, b3, attn, (const f16*)w1, b1, T, HW, (float)c_real)
    // clip length 7 (GlobalVar temporal length of the shipped configurations) and 3 (the codec variant's segments) are compiled in
    if (dense_out) {
      if (T == 7) SELFC_GMIX(true, 7, nullptr, (f16*)dense_out);
      else if (T == 3) SELFC_GMIX(true, 3, nullptr, (f16*)dense_out);
      else SELFC_GMIX(true, 0, nullptr, (f16*)dense_out);
    } else {
      if (T == 7) SELFC_GMIX(false, 7, y, nullptr);
      else if (T == 3) SELFC_GMIX(false, 3, y, nullptr);
      else SELFC_GMIX(false, 0, y, nullptr);
    }
#undef SELFC_GMIX
  }
  return hip_rc(hipGetLastError());
}

size_t selfc_globalagg_partial_floats(int N, int HW) {
  return (size_t)N * ((HW + POOL_CHUNK - 1) / POOL_CHUNK) * 64;
}

int selfc_pwconv_run(const void* in, int in_is_f32, void* out, int out_is_f32, const void* w, const float* bias,
                     size_t npix, int cin, int cout, int cout_stride, int lrelu_in, int lrelu_out, void* stream) {
  if (!in || !out || !w || !bias || npix == 0 || cin % 32 || cin < 32 || cin > 256) return SELFC_EINVAL;
  if (cout < 16 || cout % 16 || cout_stride < cout || (cout_stride & 3)) return SELFC_EINVAL;   // whole 16-channel tiles are stored
  hipStream_t s = (hipStream_t)stream;
  const int ks = cin / 32;
#define SELFC_PW(KS_) \
  if (ks == KS_) { \
    if (in_is_f32 && out_is_f32) return launch_pw<KS_, true, true>(in, out, w, bias, npix, cin, cout, cout_stride, lrelu_in, lrelu_out, s); \
    if (in_is_f32) return launch_pw<KS_, true, false>(in, out, w, bias, npix, cin, cout, cout_stride, lrelu_in, lrelu_out, s); \
    if (out_is_f32) return launch_pw<KS_, false, true>(in, out, w, bias, npix, cin, cout, cout_stride, lrelu_in, lrelu_out, s); \
    return launch_pw<KS_, false, false>(in, out, w, bias, npix, cin, cout, cout_stride, lrelu_in, lrelu_out, s); \
  }
  SELFC_PW(1) SELFC_PW(2) SELFC_PW(4) SELFC_PW(8)
#undef SELFC_PW
  return SELFC_EINVAL;
}

int selfc_stp_head_gmm(const float* feat, const void* w, const float* bias, const float* eps, float* v, size_t npix,
                       int hf_dim, int K, int v_stride, int act, void* stream) {
  if (!feat || !w || !bias || !eps || !v || npix == 0 || hf_dim != 48 || K != 5 || v_stride < hf_dim || (v_stride & 3)) return SELFC_EINVAL;
  if (act != 1 && act != 2) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  constexpr int lds = 120 * 1024 + (8 + 16 + 45) * 64, GMM_NW = 8, GMM_MT = 2;    // measured: 4 waves x 4 pixel tiles (512 registers, 48 spilled) 245 us against 162
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&stp_head_gmm_kernel<5, GMM_NW, GMM_MT>), lds, optin); e != hipSuccess) return hip_rc(e);
  ProfScope prof(PROF_STP, s);
  constexpr int pxwg = GMM_NW * 16 * GMM_MT;
  hipLaunchKernelGGL((stp_head_gmm_kernel<5, GMM_NW, GMM_MT>), dim3((unsigned)((npix + pxwg - 1) / pxwg)), dim3(GMM_NW * 64), lds, s,
                     feat, (const f16*)w, bias, eps, v, npix, v_stride, act);
  return hip_rc(hipGetLastError());
}

int selfc_gmm_sample(const float* raw, const float* eps, float* v, size_t npix, int hf_dim, int K, void* stream) {
  if (!raw || !eps || !v || npix == 0 || hf_dim != 48) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_STP, s);
  const unsigned grid = (unsigned)((npix + 15) / 16);
  if (K == 5) hipLaunchKernelGGL(gmm_sample_kernel<5>, dim3(grid), dim3(256), 0, s, raw, eps, v, npix);
  else if (K == 3) hipLaunchKernelGGL(gmm_sample_kernel<3>, dim3(grid), dim3(256), 0, s, raw, eps, v, npix);
  else if (K == 1) hipLaunchKernelGGL(gmm_sample_kernel<1>, dim3(grid), dim3(256), 0, s, raw, eps, v, npix);
  else return SELFC_EINVAL;
  return hip_rc(hipGetLastError());
}

int selfc_gmm_sample_generic(const float* raw, const float* eps, float* v, size_t npix, int hf_dim, int K, int raw_stride,
                             int v_stride, float logsigma_scale, void* stream) {
  if (!raw || !eps || !v || npix == 0 || hf_dim < 1 || K < 1 || raw_stride < hf_dim * K * 3 || v_stride < hf_dim) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_STP, s);
  hipLaunchKernelGGL(gmm_sample_generic_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, raw, eps, v, npix, hf_dim, K,
                     raw_stride, v_stride, logsigma_scale);
  return hip_rc(hipGetLastError());
}

int selfc_gmm_sample_generic_bwd(const float* raw, const float* eps, const float* dv, float* draw, size_t npix, int hf_dim, int K,
                                 int raw_stride, int v_stride, float logsigma_scale, void* stream) {
  if (!raw || !eps || !dv || !draw || npix == 0 || hf_dim < 1 || K < 1 || raw_stride < hf_dim * K * 3 || v_stride < hf_dim) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_BWD, s);
  hipLaunchKernelGGL(gmm_sample_generic_bwd_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, raw, eps, dv, draw, npix,
                     hf_dim, K, raw_stride, v_stride, logsigma_scale);
  return hip_rc(hipGetLastError());
}

int selfc_gmm_sample_bwd(const float* raw, const float* eps, const float* dv, float* draw, size_t npix, int hf_dim, int K, void* stream) {
  if (!raw || !eps || !dv || !draw || npix == 0 || hf_dim != 48) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_BWD, s);
  const unsigned grid = (unsigned)((npix + 15) / 16);
  if (K == 5) hipLaunchKernelGGL(gmm_sample_bwd_kernel<5>, dim3(grid), dim3(256), 0, s, raw, eps, dv, draw, npix);
  else if (K == 3) hipLaunchKernelGGL(gmm_sample_bwd_kernel<3>, dim3(grid), dim3(256), 0, s, raw, eps, dv, draw, npix);
  else if (K == 1) hipLaunchKernelGGL(gmm_sample_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, raw, eps, dv, draw, npix);
  else return SELFC_EINVAL;
  return hip_rc(hipGetLastError());
}

int selfc_lrelu_bwd(float* dx, const float* x, size_t n, void* stream) {
  if (!dx || !x || n == 0 || (n & 3)) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  hipLaunchKernelGGL(lrelu_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float4*)dx, (const float4*)x, n / 4);
  return hip_rc(hipGetLastError());
}

int selfc_f16_rows_to_planes(const void* rows, void* planes, size_t npix, int C, void* stream) {
  if (!rows || !planes || npix == 0 || C < 32 || C % 32) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  const size_t items = npix * (size_t)(C / 8);
  hipLaunchKernelGGL(rows_to_planes_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f16*)rows, (f16*)planes, npix, C);
  return hip_rc(hipGetLastError());
}

int selfc_rowsum_accum(const selfc_rowsum* job, void* stream) {
  if (!job || job->n <= 0 || job->n > SELFC_ROWSUM_MAX) return SELFC_EINVAL;
  RowSumArgs a{};
  int total = 0;
  for (int i = 0; i < job->n; ++i) {
    if (!job->src[i] || !job->dst[i] || job->len[i] <= 0 || job->rows[i] <= 0) return SELFC_EINVAL;
    a.src[i] = job->src[i]; a.dst[i] = job->dst[i]; a.rows[i] = job->rows[i]; a.beta[i] = job->beta[i];
    a.end[i] = (total += job->len[i]);
  }
  a.n = job->n;
  hipLaunchKernelGGL(rowsum_accum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return hip_rc(hipGetLastError());
}

size_t selfc_globalagg_bwd_scratch_bytes(int N, int T, int H, int W) {
  if (N <= 0 || T <= 0 || T > TMAX || N % T || H <= 0 || W <= 0) return 0;
  return gagg_bwd_layout(N, T, H, W).total;
}

int selfc_globalagg_bwd(const float* x, const float* dy, float* dx, const float* wmap, const float* fc_bias, const void* w1t,
                        const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                        float* dw1, float* db1_clip, float* dw2_clip, float* db2_clip, float* dw3_clip, float* db3_clip,
                        float* dfcb_clip, float* dwmap_clip, void* scratch, size_t scratch_bytes,
                        int N, int T, int H, int W, void* stream) {
  return selfc_globalagg_bwd_x(x, dy, dx, wmap, fc_bias, w1t, b1, w2, b2, w3, b3, dw1, db1_clip, dw2_clip, db2_clip, dw3_clip, db3_clip,
                               dfcb_clip, dwmap_clip, scratch, scratch_bytes, N, T, H, W, nullptr, nullptr, stream);
}

int selfc_globalagg_bwd_x(const float* x, const float* dy, float* dx, const float* wmap, const float* fc_bias, const void* w1t,
                          const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                          float* dw1, float* db1_clip, float* dw2_clip, float* db2_clip, float* dw3_clip, float* db3_clip,
                          float* dfcb_clip, float* dwmap_clip, void* scratch, size_t scratch_bytes,
                          int N, int T, int H, int W, const float* dy_amax, float* dx_amax_out, void* stream) {
  if (!x || !dy || !dx || !wmap || !fc_bias || !w1t || !b1 || !w2 || !b2 || !w3 || !b3 || !scratch) return SELFC_EINVAL;
  if (!dw1 || !db1_clip || !dw2_clip || !db2_clip || !dw3_clip || !db3_clip || !dfcb_clip || !dwmap_clip) return SELFC_EINVAL;
  if (N <= 0 || T <= 0 || T > TMAX || N % T || H <= 0 || W <= 0 || x == dx) return SELFC_EINVAL;
  const GaggBwdLayout L = gagg_bwd_layout(N, T, H, W);
  if (scratch_bytes < L.total) return SELFC_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_BWD, s);
  unsigned char* sb = (unsigned char*)scratch;
  const int HW = H * W, B = N / T;
  const size_t npix = (size_t)N * HW;
  f16* dyp = (f16*)(sb + L.off_dyp);
  f16* zp = (f16*)(sb + L.off_zp);
  float* dz = (float*)(sb + L.off_dz);
  float* amax = (float*)(sb + L.off_amax);
  float* pool = (float*)(sb + L.off_pool);
  float* A = (float*)(sb + L.off_A);
  float* pdA = (float*)(sb + L.off_pdA);
  float* pdyo = (float*)(sb + L.off_pdyo);
  float* dg = (float*)(sb + L.off_dg);
  int rc;
  // dz = W1^T dy on the MFMA (scaled f16 planes in, fp32 rows out)
  // dy_amax: the producer of dy already took max|dy| (a subnet backward's dx epilogue); the to-planes kernel hands it on to this call's slot
  if (!dy_amax && (rc = bwd_absmax(dy, npix * 64, amax, s))) return rc;
  if ((rc = bwd_to_planes(dy, dyp, npix, 64, 64, 2, 0, 1.f, dy_amax ? dy_amax : amax, s, dy_amax ? amax : nullptr))) return rc;
  {
    BwdConv c{};
    c.in = dyp; c.nplanes_in = 2; c.kt = 1; c.sp1 = 1; c.w = w1t; c.ngroups = 2; c.mask_z = -1;
    c.plain = dz; c.coutp = 64; c.accumulate = 0; c.amax = amax;
    if ((rc = bwd_conv_planes(c, N, T, H, W, s))) return rc;
  }
  hipLaunchKernelGGL(gagg_pool_kernel, dim3(L.nchunk, N), dim3(256), 0, s, x, wmap, pool, HW, L.nchunk);
  if (T == 7) hipLaunchKernelGGL(gagg_bwd_reduce_kernel<7>, dim3(L.nchunkb, B), dim3(256), 0, s, x, dz, dy, pdA, pdyo, T, HW, L.nchunkb);
  else if (T == 3) hipLaunchKernelGGL(gagg_bwd_reduce_kernel<3>, dim3(L.nchunkb, B), dim3(256), 0, s, x, dz, dy, pdA, pdyo, T, HW, L.nchunkb);
  else hipLaunchKernelGGL(gagg_bwd_reduce_kernel<0>, dim3(L.nchunkb, B), dim3(256), 0, s, x, dz, dy, pdA, pdyo, T, HW, L.nchunkb);
#define SELFC_GATTN(TT_) hipLaunchKernelGGL(gagg_attn_bwd_kernel<TT_>, dim3(B), dim3(64), 0, s, pool, L.nchunk, L.nchunkb, fc_bias, w2, b2, w3, b3, b1, pdA, \
                                             pdyo, A, dg, db1_clip, dw2_clip, db2_clip, dw3_clip, db3_clip, dfcb_clip, T)
  if (T == 7) SELFC_GATTN(7);                 // GlobalVar temporal length of the shipped configurations (3: the codec variant's segments)
  else if (T == 3) SELFC_GATTN(3);
  else SELFC_GATTN(0);
#undef SELFC_GATTN
  if (T == 7) {
    hipLaunchKernelGGL(gagg_bwd_dx_kernel<7>, dim3((unsigned)((HW + 15) / 16), (unsigned)B), dim3(256), 0, s, x, dz, dy, A, dg, wmap, dx,
                     dwmap_clip, zp, T, HW, npix, (unsigned*)dx_amax_out);
  } else if (T == 3) {
    hipLaunchKernelGGL(gagg_bwd_dx_kernel<3>, dim3((unsigned)((HW + 15) / 16), (unsigned)B), dim3(256), 0, s, x, dz, dy, A, dg, wmap, dx,
                     dwmap_clip, zp, T, HW, npix, (unsigned*)dx_amax_out);
  } else {
    hipLaunchKernelGGL(gagg_bwd_dx_kernel<0>, dim3((unsigned)((HW + 15) / 16), (unsigned)B), dim3(256), 0, s, x, dz, dy, A, dg, wmap, dx,
                     dwmap_clip, zp, T, HW, npix, (unsigned*)dx_amax_out);
  }
  if ((rc = hip_rc(hipGetLastError()))) return rc;
  WgradJob j{};
  j.P = dyp; j.Pn = 2; j.Q[0] = zp; j.Qn[0] = 2; j.taps = 1; j.temporal = 0;
  j.wout = dw1; j.O = 64; j.Ctot = 64; j.cin = 64; j.nx = 2; j.beta = 0.f;
  return bwd_wgrad(j, amax, sb + L.off_wg, N, T, H, W, s);
}

}  // extern "C"
