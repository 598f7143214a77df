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
#include "common.hpp"
#include "prof.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace {

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

constexpr int POOL_CHUNK = 512;  // pixels per partial sum
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

// ---- (2) per clip: g = sum(partials) + fc.bias; q = proj2(g), k = proj3(g); A = softmax(q k^T / 64, dim=-1)
__global__ __launch_bounds__(64) void gagg_attn_kernel(const float* __restrict__ partial, int nchunk, float fcb,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       const float* __restrict__ w3, const float* __restrict__ b3,
                                                       float* __restrict__ A, int T) {
  __shared__ float g[TMAX][64], q[TMAX][64], k[TMAX][64], m[TMAX][TMAX];
  const int b = blockIdx.x, c = threadIdx.x;
  for (int t = 0; t < T; ++t) {
    float s = 0.f;
    const float* p = partial + (size_t)(b * T + t) * nchunk * 64 + c;
    for (int j = 0; j < nchunk; ++j) s += p[(size_t)j * 64];
    g[t][c] = s + fcb;
  }
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    float sq = b2[c], sk = b3[c];
    for (int j = 0; j < 64; ++j) {
      sq += g[t][j] * w2[c * 64 + j];
      sk += g[t][j] * w3[c * 64 + j];
    }
    q[t][c] = sq;
    k[t][c] = sk;
  }
  __syncthreads();
  if (c < T * T) {
    const int t1 = c / T, t2 = c % T;
    float s = 0.f;
    for (int j = 0; j < 64; ++j) s += q[t1][j] * k[t2][j];
    m[t1][t2] = s / 64.0f;
  }
  __syncthreads();
  if (c < T) {            // softmax over the last axis of row c
    float mx = m[c][0];
    for (int j = 1; j < T; ++j) mx = fmaxf(mx, m[c][j]);
    float e[TMAX], s = 0.f;
    for (int j = 0; j < T; ++j) { e[j] = expf(m[c][j] - mx); s += e[j]; }
    for (int j = 0; j < T; ++j) A[((size_t)b * T + c) * T + j] = e[j] / s;
  }
}

// ---- (3) temporal mix + 64x64 projection + residual.  Wave = 16 pixels of one clip.
__global__ __launch_bounds__(256) void gagg_mix_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       const float* __restrict__ A, const f16* __restrict__ w1,
                                                       const float* __restrict__ b1, int T, int HW) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles = (HW + 63) / 64;
  const int b = blockIdx.x / tiles;
  const int p0 = (blockIdx.x % tiles) * 64 + wave * 16;
  if (p0 >= HW) return;
  const int pl = p0 + (lane & 15);
  const bool pvalid = pl < HW;
  const int pc = pvalid ? pl : HW - 1;
  const int kq = lane >> 4;
  const float* Ab = A + (size_t)b * T * T;

  // this lane's 16 channels (k = 32 ks + 8 kq + j) of every frame of the clip
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

  for (int t2 = 0; t2 < T; ++t2) {
    float xm[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) xm[j] = 0.f;
    float colsum = 0.f;
#pragma unroll
    for (int t1 = 0; t1 < TMAX; ++t1) {
      if (t1 < T) {
        const float a = Ab[t1 * T + t2];
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
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      acc = mfma_16x16x32(wf[o][0], bf[0], acc);
      acc = mfma_16x16x32(wf[o][1], bf[1], acc);
      const int oc = o * 16 + kq * 4;
      const float4 xr = *reinterpret_cast<const float4*>(x + pix * 64 + oc);
      const float4 bb = *reinterpret_cast<const float4*>(b1 + oc);
      if (pvalid)
        *reinterpret_cast<float4*>(y + pix * 64 + oc) =
            make_float4(xr.x + acc[0] + bb.x * colsum, xr.y + acc[1] + bb.y * colsum,
                        xr.z + acc[2] + bb.z * colsum, xr.w + acc[3] + bb.w * colsum);
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
        for (int j = 0; j < 8; ++j) bf[m][ks][j] = (f16)(lrelu_in ? lrelu02(v[j]) : v[j]);
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
        if (lrelu_out) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = lrelu02(v[j]);
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

template <int KS, bool IN_F32, bool OUT_F32>
int launch_pw(const void* in, void* out, const void* w, const float* bias, size_t npix, int cin, int cout,
              int cout_stride, int lrelu_in, int lrelu_out, hipStream_t s) {
  constexpr int lds = 64 * 1024;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_kernel<KS, IN_F32, OUT_F32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return hip_rc(e);
    attr_done = true;
  }
  ProfScope prof(PROF_STP, s);
  const unsigned grid = (unsigned)((npix + 127) / 128);
  hipLaunchKernelGGL((pwconv_kernel<KS, IN_F32, OUT_F32>), dim3(grid), dim3(256), lds, s, in, out, (const f16*)w, bias,
                     npix, cin, (cout + 15) / 16, cout_stride, lrelu_in, lrelu_out);
  return hip_rc(hipGetLastError());
}

}  // namespace

extern "C" {

int selfc_globalagg_run(const float* x, float* y, const float* wmap, float fc_bias, const void* w1, const float* b1,
                        const float* w2, const float* b2, const float* w3, const float* b3,
                        float* partial, float* attn, int N, int T, int HW, void* stream) {
  if (!x || !y || !wmap || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !partial || !attn) return SELFC_EINVAL;
  if (N <= 0 || T <= 0 || T > TMAX || N % T || HW <= 0 || x == y) return SELFC_EINVAL;
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
    hipLaunchKernelGGL(gagg_attn_kernel, dim3(B), dim3(64), 0, s, partial, nchunk, fc_bias, w2, b2, w3, b3, attn, T);
  }
  rc = hip_rc(hipGetLastError());
  if (rc) return rc;
  {
    ProfScope prof(PROF_STP, s);
    const int tiles = (HW + 63) / 64;
    hipLaunchKernelGGL(gagg_mix_kernel, dim3((unsigned)(tiles * B)), dim3(256), 0, s, x, y, attn, (const f16*)w1, b1, T, HW);
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

}  // extern "C"
