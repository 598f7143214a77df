// HBM-bound split/merge transforms and layout kernels (gfx950).
//   Haar butterfly + band shuffle      <- Inv_arch.py:64-81
//   FrequencyAnalyzer fwd / rev        <- SelfC_GMM_arch_inv.py:62-82
//   NCHW <-> latent (narrow / cat)     <- Inv_arch.py:22,33
//   Quantization                       <- Quantization.py:7-17
// One thread per low-resolution pixel; reads of NCHW planes and writes of NHWC
// rows are 16-B vectors, consecutive lanes touch consecutive addresses.
#include "common.hpp"
#include "prof.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace {

constexpr int TPB = 256;

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }
inline unsigned nblocks(size_t n) { return (unsigned)((n + TPB - 1) / TPB); }

// ---------------------------------------------------------------- Haar
// y[:, k*C + c] = band k of channel c; sums in the order ((a+b)+c)+d so the
// fp32 bits equal the reference's depthwise conv2d (pinned by tests/golden/g1).
__global__ void haar_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                int N, int C, int h, int w) {
  const size_t total = (size_t)N * C * h * w;
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int xx = (int)(i % w);
  const int yy = (int)((i / w) % h);
  const int c = (int)((i / ((size_t)w * h)) % C);
  const int n = (int)(i / ((size_t)w * h * C));
  const int W = 2 * w;
  const float* p = x + (((size_t)n * C + c) * (2 * h) + 2 * yy) * W + 2 * xx;
  const float2 r0 = *reinterpret_cast<const float2*>(p);
  const float2 r1 = *reinterpret_cast<const float2*>(p + W);
  const float a = r0.x, b = r0.y, cc = r1.x, d = r1.y;
  const size_t plane = (size_t)h * w;
  float* o = y + ((size_t)n * 4 * C + c) * plane + (size_t)yy * w + xx;
  o[0] = (((a + b) + cc) + d) / 4.0f;
  o[(size_t)C * plane] = (((a - b) + cc) - d) / 4.0f;
  o[(size_t)2 * C * plane] = (((a + b) - cc) - d) / 4.0f;
  o[(size_t)3 * C * plane] = (((a - b) - cc) + d) / 4.0f;
}

__global__ void haar_inv_kernel(const float* __restrict__ y, float* __restrict__ x,
                                int N, int C, int h, int w) {
  const size_t total = (size_t)N * C * h * w;
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int xx = (int)(i % w);
  const int yy = (int)((i / w) % h);
  const int c = (int)((i / ((size_t)w * h)) % C);
  const int n = (int)(i / ((size_t)w * h * C));
  const size_t plane = (size_t)h * w;
  const float* p = y + ((size_t)n * 4 * C + c) * plane + (size_t)yy * w + xx;
  const float ll = p[0], hl = p[(size_t)C * plane], lh = p[(size_t)2 * C * plane], hh = p[(size_t)3 * C * plane];
  const int W = 2 * w;
  float* o = x + (((size_t)n * C + c) * (2 * h) + 2 * yy) * W + 2 * xx;
  float2 r0, r1;
  r0.x = ((ll + hl) + lh) + hh;
  r0.y = ((ll - hl) + lh) - hh;
  r1.x = ((ll + hl) - lh) - hh;
  r1.y = ((ll - hl) - lh) + hh;
  *reinterpret_cast<float2*>(o) = r0;
  *reinterpret_cast<float2*>(o + W) = r1;
}

// Indirect tensor addresses (abi 10).  A hipGraph bakes kernel arguments in; the module API must read the caller's tensor and
// write a FRESH output tensor on every call.  With IND the pointer argument is the device address of a SLOT that holds the
// real base address (written by selfc_set_pointers on the same stream right before the replay), plus an element offset
// (the part of the batch this launch owns): one scalar load per wave, no copy of the tensor, the graph stays as captured.
template <bool IND, class T>
__device__ __forceinline__ T* resolve(T* p, const size_t off) {
  if (!IND) return p;
  return *reinterpret_cast<T* const*>(p) + off;
}

// ---------------------------------------------------------------- FrequencyAnalyzer
// forward: lo = KxK block mean (row-major sequential sum / K^2, the order of
// torch's CPU avg_pool2d), hi[(sy*K+sx)*3+c] = x - lo.
template <int K, bool IND = false>
__global__ void freq_fwd_kernel(const float* __restrict__ xarg, float* __restrict__ x1, float* __restrict__ x2,
                                f16* __restrict__ fd, int FC, int N, int h, int w, size_t ioff = 0) {
  const float* __restrict__ x = resolve<IND>(xarg, ioff);
  constexpr int C2 = 3 * K * K;
  const size_t total = (size_t)N * h * w;
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int xx = (int)(i % w);
  const int yy = (int)((i / w) % h);
  const int n = (int)(i / ((size_t)w * h));
  const int H = K * h, W = K * w;
  float lo[4] = {0.f, 0.f, 0.f, 0.f};
  float hi[C2];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* p = x + (((size_t)n * 3 + c) * H + (size_t)K * yy) * W + (size_t)K * xx;
    float v[K][K];
#pragma unroll
    for (int sy = 0; sy < K; ++sy) {
      if (K == 4) {
        const float4 r = *reinterpret_cast<const float4*>(p + (size_t)sy * W);
        v[sy][0] = r.x; v[sy][1] = r.y; v[sy][2] = r.z; v[sy][3] = r.w;
      } else {
        const float2 r = *reinterpret_cast<const float2*>(p + (size_t)sy * W);
        v[sy][0] = r.x; v[sy][1] = r.y;
      }
    }
    float s = 0.f;
#pragma unroll
    for (int sy = 0; sy < K; ++sy)
#pragma unroll
      for (int sx = 0; sx < K; ++sx) s += v[sy][sx];
    const float m = s / (float)(K * K);
    lo[c] = m;
#pragma unroll
    for (int sy = 0; sy < K; ++sy)
#pragma unroll
      for (int sx = 0; sx < K; ++sx) hi[(sy * K + sx) * 3 + c] = v[sy][sx] - m;
  }
  *reinterpret_cast<float4*>(x1 + i * 4) = make_float4(lo[0], lo[1], lo[2], 0.f);
  float4* o2 = reinterpret_cast<float4*>(x2 + i * C2);
#pragma unroll
  for (int j = 0; j < C2 / 4; ++j) o2[j] = make_float4(hi[4 * j], hi[4 * j + 1], hi[4 * j + 2], hi[4 * j + 3]);
  if (fd != nullptr) {   // plane-blocked dense buffer: [C/32][N*h*w][32]
#pragma unroll
    for (int j = 0; j < C2 / 4; ++j) {
      uint2 u;
      u.x = pack2(hi[4 * j], hi[4 * j + 1]);
      u.y = pack2(hi[4 * j + 2], hi[4 * j + 3]);
      *reinterpret_cast<uint2*>(fd + (size_t)((4 * j) >> 5) * total * 32 + i * 32 + ((4 * j) & 31)) = u;
    }
  }
}

// reverse: out[c][K*y+sy][K*x+sx] = lo[c] + hf[c*K^2 + sy*K + sx]  (nn.PixelShuffle order)
template <int K, bool IND = false>
__global__ void freq_inv_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                float* __restrict__ xarg, int N, int h, int w, size_t ioff = 0) {
  float* __restrict__ x = resolve<IND>(xarg, ioff);
  constexpr int C2 = 3 * K * K;
  const size_t total = (size_t)N * h * w;
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const int xx = (int)(i % w);
  const int yy = (int)((i / w) % h);
  const int n = (int)(i / ((size_t)w * h));
  const int H = K * h, W = K * w;
  const float4 l4 = *reinterpret_cast<const float4*>(x1 + i * 4);
  const float lo[3] = {l4.x, l4.y, l4.z};
  float hf[C2];
  const float4* p2 = reinterpret_cast<const float4*>(x2 + i * C2);
#pragma unroll
  for (int j = 0; j < C2 / 4; ++j) {
    const float4 r = p2[j];
    hf[4 * j] = r.x; hf[4 * j + 1] = r.y; hf[4 * j + 2] = r.z; hf[4 * j + 3] = r.w;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float* o = x + (((size_t)n * 3 + c) * H + (size_t)K * yy) * W + (size_t)K * xx;
#pragma unroll
    for (int sy = 0; sy < K; ++sy) {
      if (K == 4) {
        float4 r;
        r.x = lo[c] + hf[c * 16 + sy * 4 + 0];
        r.y = lo[c] + hf[c * 16 + sy * 4 + 1];
        r.z = lo[c] + hf[c * 16 + sy * 4 + 2];
        r.w = lo[c] + hf[c * 16 + sy * 4 + 3];
        *reinterpret_cast<float4*>(o + (size_t)sy * W) = r;
      } else {
        float2 r;
        r.x = lo[c] + hf[c * 4 + sy * 2 + 0];
        r.y = lo[c] + hf[c * 4 + sy * 2 + 1];
        *reinterpret_cast<float2*>(o + (size_t)sy * W) = r;
      }
    }
  }
}

// ---------------------------------------------------------------- NCHW <-> latent
template <bool IND = false>
__global__ void nchw_to_latent_kernel(const float* __restrict__ xarg, float* __restrict__ x1, float* __restrict__ x2,
                                      f16* __restrict__ fd, int FC, int N, int c1, int c2, int c2p, size_t HW, size_t ioff = 0) {
  const float* __restrict__ x = resolve<IND>(xarg, ioff);
  const size_t total = (size_t)N * HW;
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const size_t n = i / HW, p = i - n * HW;
  const float* src = x + n * (size_t)(c1 + c2) * HW + p;
  float v1[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < c1; ++c) v1[c] = src[(size_t)c * HW];
  *reinterpret_cast<float4*>(x1 + i * 4) = make_float4(v1[0], v1[1], v1[2], v1[3]);
  for (int c0 = 0; c0 < c2p; c0 += 4) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (c0 + j < c2) ? src[(size_t)(c1 + c0 + j) * HW] : 0.f;
    *reinterpret_cast<float4*>(x2 + i * c2p + c0) = make_float4(v[0], v[1], v[2], v[3]);
    if (fd != nullptr) {
      uint2 u;
      u.x = pack2(v[0], v[1]);
      u.y = pack2(v[2], v[3]);
      *reinterpret_cast<uint2*>(fd + (size_t)(c0 >> 5) * total * 32 + i * 32 + (c0 & 31)) = u;
    }
  }
}

template <bool IND = false>
__global__ void latent_to_nchw_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                      float* __restrict__ yarg, int N, int c1, int c2, int c2p, size_t HW, size_t ioff = 0) {
  float* __restrict__ y = resolve<IND>(yarg, ioff);
  const size_t total = (size_t)N * HW;
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const size_t n = i / HW, p = i - n * HW;
  float* dst = y + n * (size_t)(c1 + c2) * HW + p;
  const float4 a = *reinterpret_cast<const float4*>(x1 + i * 4);
  const float v1[4] = {a.x, a.y, a.z, a.w};
  for (int c = 0; c < c1; ++c) dst[(size_t)c * HW] = v1[c];
  for (int c0 = 0; c0 < c2p; c0 += 4) {
    const float4 r = *reinterpret_cast<const float4*>(x2 + i * c2p + c0);
    const float v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < c2) dst[(size_t)(c1 + c0 + j) * HW] = v[j];
  }
}

template <bool IND = false>
__global__ void nchw_to_nhwc4_kernel(const float* __restrict__ xarg, float* __restrict__ y, int N, int C, int Cp, size_t HW, size_t ioff = 0) {
  const float* __restrict__ x = resolve<IND>(xarg, ioff);
  const size_t total = (size_t)N * HW;
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const size_t n = i / HW, p = i - n * HW;
  const float* src = x + n * (size_t)C * HW + p;
  for (int c0 = 0; c0 < Cp; c0 += 4) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (c0 + j < C) ? src[(size_t)(c0 + j) * HW] : 0.f;
    *reinterpret_cast<float4*>(y + i * Cp + c0) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

template <bool IND = false>
__global__ void nhwc4_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ yarg, int N, int C, int Cp, size_t HW, size_t ioff = 0) {
  float* __restrict__ y = resolve<IND>(yarg, ioff);
  const size_t total = (size_t)N * HW;
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= total) return;
  const size_t n = i / HW, p = i - n * HW;
  float* dst = y + n * (size_t)C * HW + p;
  for (int c0 = 0; c0 < Cp; c0 += 4) {
    const float4 r = *reinterpret_cast<const float4*>(x + i * Cp + c0);
    const float v[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < C) dst[(size_t)(c0 + j) * HW] = v[j];
  }
}

// Quantization.py:7-17: clamp(x,0,1); round(x*255)/255 with round-half-to-even (torch.round).
__global__ void quantize_kernel(float* __restrict__ x, size_t n4, const float qv, const int clip) {
  const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<float4*>(x)[i];
  float* f = reinterpret_cast<float*>(&v);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float t = (f[j] != f[j] || !clip) ? f[j] : fminf(fmaxf(f[j], 0.f), 1.f);   // fminf / fmaxf would turn a NaN into 0: torch.clamp keeps it
    f[j] = rintf(t * qv) / qv;
  }
  reinterpret_cast<float4*>(x)[i] = v;
}

// Every product and sum is rounded separately, as the reference's torch expression evaluates it.  hipcc's
// default -ffp-contract=fast may fuse EITHER product of a*b + c*d into the FMA, and it picked different ones
// for the two SLP-packed lanes (Y(a) != Y(a) by half an ulp); HIP's __fmul_rn / __fadd_rn are plain * and +
// and do not stop that, the pragma does.
__device__ __forceinline__ float y_of(float r, float g, float b) {
#pragma clang fp contract(off)
  float s = r * 65.481f;
  s = s + g * 128.553f;
  s = s + b * 24.966f;
  s = s + 16.0f;
  return s / 255.0f;
}

// Y-channel squared error of test_rescaling.py's metric (data/util.py:239-245, utils/util.py:198-221):
// Y = (65.481 R + 128.553 G + 24.966 B + 16) / 255; partial[n][blk] = sum over the block's pixels of (Ya - Yb)^2.
__global__ __launch_bounds__(256) void y_sse_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                    double* __restrict__ partial, int HW, int nblk) {
  __shared__ double red[256];
  const int n = blockIdx.y;
  const float* pa = a + (size_t)n * 3 * HW;
  const float* pb = b + (size_t)n * 3 * HW;
  double acc = 0.0;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += nblk * 256) {
    // every product and sum rounded separately, as the reference's torch expression does (no FMA contraction)
    const float ya = y_of(pa[p], pa[HW + p], pa[2 * HW + p]);
    const float yb = y_of(pb[p], pb[HW + p], pb[2 * HW + p]);
    const double d = (double)ya - (double)yb;
    acc += d * d;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[(size_t)n * nblk + blockIdx.x] = red[0];
}

// Y-channel SSIM of test_rescaling.py:110-122 (utils/util.py:396-441 with an 11-tap sigma-1.5 window, no padding,
// K1 = 0.01, K2 = 0.03, data_range 1): each 16x16 block of the (H-10)x(W-10) SSIM map stages the 26x26 Y patches of both
// images in LDS, filters the five moments horizontally then vertically (as the reference's two conv2d passes do) and
// adds its part of the map sum to partial[n][block] (fp64, deterministic two-stage sum).
__global__ __launch_bounds__(256) void y_ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ win,
                                                     double* __restrict__ partial, int H, int W, int nbx, int nby) {
  __shared__ float ya[26][27], yb[26][27];
  __shared__ float hz[5][26][17];             // horizontally filtered x, y, xx, yy, xy at 26 rows x 16 columns
  __shared__ float wk[11];
  __shared__ double red[256];
  const int n = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  const int tid = threadIdx.x;
  const size_t HW = (size_t)H * W;
  const float* pa = a + (size_t)n * 3 * HW;
  const float* pb = b + (size_t)n * 3 * HW;
  if (tid < 11) wk[tid] = win[tid];
  for (int i = tid; i < 26 * 26; i += 256) {
    const int r = i / 26, c = i - r * 26;
    const int y = min(by * 16 + r, H - 1), x = min(bx * 16 + c, W - 1);
    const size_t p = (size_t)y * W + x;
    ya[r][c] = y_of(pa[p], pa[HW + p], pa[2 * HW + p]);
    yb[r][c] = y_of(pb[p], pb[HW + p], pb[2 * HW + p]);
  }
  __syncthreads();
  for (int i = tid; i < 26 * 16; i += 256) {
    const int r = i >> 4, c = i & 15;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float u = ya[r][c + k], v = yb[r][c + k], w = wk[k];
      s0 += w * u; s1 += w * v; s2 += w * (u * u); s3 += w * (v * v); s4 += w * (u * v);
    }
    hz[0][r][c] = s0; hz[1][r][c] = s1; hz[2][r][c] = s2; hz[3][r][c] = s3; hz[4][r][c] = s4;
  }
  __syncthreads();
  const int r = tid >> 4, c = tid & 15;
  double val = 0.0;
  if (by * 16 + r < H - 10 && bx * 16 + c < W - 10) {
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 11; ++k)
#pragma unroll
      for (int q = 0; q < 5; ++q) m[q] += wk[k] * hz[q][r + k][c];
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    const float mu1 = m[0], mu2 = m[1];
    const float s1 = m[2] - mu1 * mu1, s2 = m[3] - mu2 * mu2, s12 = m[4] - mu1 * mu2;
    const float cs = (2.f * s12 + c2) / (s1 + s2 + c2);
    val = (double)(((2.f * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs);
  }
  red[tid] = val;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) partial[((size_t)n * nby + by) * nbx + bx] = red[0];
}

// Guassian_downsample (models/Guassian.py:34-51, scale 4): out[oy][ox] = sum_{i,j in [-6,6]} g[i][j] *
// x[reflect(4 oy + i)][reflect(4 ox + j)] per plane ('reflect' = mirror without repeating the edge sample).
__global__ __launch_bounds__(256) void gauss_down4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ g, int planes, int H, int W) {
  __shared__ float gk[169];
  if (threadIdx.x < 169) gk[threadIdx.x] = g[threadIdx.x];
  __syncthreads();
  const int h = H / 4, w = W / 4;
  const size_t total = (size_t)planes * h * w;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ox = (int)(i % w), oy = (int)((i / w) % h);
  const size_t pl = i / ((size_t)w * h);
  const float* p = x + pl * (size_t)H * W;
  float acc = 0.f;
  for (int a = -6; a <= 6; ++a) {
    int yy = 4 * oy + a;
    yy = yy < 0 ? -yy : (yy >= H ? 2 * H - 2 - yy : yy);
    for (int b = -6; b <= 6; ++b) {
      int xx = 4 * ox + b;
      xx = xx < 0 ? -xx : (xx >= W ? 2 * W - 2 - xx : xx);
      acc += gk[(a + 6) * 13 + (b + 6)] * p[(size_t)yy * W + xx];
    }
  }
  y[i] = acc;
}

}  // namespace

extern "C" {

int selfc_gauss_down4(const float* x, float* y, const float* g169, int planes, int H, int W, void* stream) {
  if (!x || !y || !g169 || planes <= 0 || H < 8 || W < 8 || (H & 3) || (W & 3)) return SELFC_EINVAL;   // reflect needs 6 < H, W
  const size_t total = (size_t)planes * (H / 4) * (W / 4);
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(gauss_down4_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, g169, planes, H, W);
  return hip_rc(hipGetLastError());
}

int selfc_y_ssim(const float* a, const float* b, const float* win11, double* partial, int N, int H, int W, void* stream) {
  if (!a || !b || !win11 || !partial || N <= 0 || H < 11 || W < 11) return SELFC_EINVAL;
  const int nbx = (W - 10 + 15) / 16, nby = (H - 10 + 15) / 16;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(y_ssim_kernel, dim3(nbx, nby, N), dim3(256), 0, (hipStream_t)stream, a, b, win11, partial, H, W, nbx, nby);
  return hip_rc(hipGetLastError());
}

int selfc_y_sse_blocks(int HW) { const int nb = (HW + 256 * 8 - 1) / (256 * 8); return nb < 1 ? 1 : (nb > 256 ? 256 : nb); }

int selfc_y_sse(const float* a, const float* b, double* partial, int N, int HW, void* stream) {
  if (!a || !b || !partial || N <= 0 || HW <= 0) return SELFC_EINVAL;
  const int nblk = selfc_y_sse_blocks(HW);
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(y_sse_kernel, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, a, b, partial, HW, nblk);
  return hip_rc(hipGetLastError());
}

int selfc_haar_fwd_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return SELFC_EINVAL;
  const size_t total = (size_t)N * C * (H / 2) * (W / 2);
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(haar_fwd_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, x, y, N, C, H / 2, W / 2);
  return hip_rc(hipGetLastError());
}

int selfc_haar_inv_nchw(const float* y, float* x, int N, int C, int h, int w, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || h <= 0 || w <= 0) return SELFC_EINVAL;
  const size_t total = (size_t)N * C * h * w;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(haar_inv_kernel, dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, y, x, N, C, h, w);
  return hip_rc(hipGetLastError());
}

int selfc_freq_fwd(const float* x, float* x1, float* x2, void* fd, int FC, int N, int H, int W, int k, void* stream) {
  if (!x || !x1 || !x2 || N <= 0 || H <= 0 || W <= 0 || (k != 4 && k != 2) || H % k || W % k) return SELFC_EINVAL;
  if (fd && (FC < 3 * k * k || (FC & 3))) return SELFC_EINVAL;
  const int h = H / k, w = W / k;
  const size_t total = (size_t)N * h * w;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  if (k == 4)
    hipLaunchKernelGGL((freq_fwd_kernel<4, false>), dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, x, x1, x2, (f16*)fd, FC, N, h, w, (size_t)0);
  else
    hipLaunchKernelGGL((freq_fwd_kernel<2, false>), dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, x, x1, x2, (f16*)fd, FC, N, h, w, (size_t)0);
  return hip_rc(hipGetLastError());
}

int selfc_freq_inv(const float* x1, const float* x2, float* x, int N, int h, int w, int k, void* stream) {
  if (!x || !x1 || !x2 || N <= 0 || h <= 0 || w <= 0 || (k != 4 && k != 2)) return SELFC_EINVAL;
  const size_t total = (size_t)N * h * w;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  if (k == 4)
    hipLaunchKernelGGL((freq_inv_kernel<4, false>), dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, x1, x2, x, N, h, w, (size_t)0);
  else
    hipLaunchKernelGGL((freq_inv_kernel<2, false>), dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, x1, x2, x, N, h, w, (size_t)0);
  return hip_rc(hipGetLastError());
}

int selfc_nchw_to_latent(const float* x, float* x1, float* x2, void* fd, int FC, int N, int c1, int c2, int H, int W, void* stream) {
  if (!x || !x1 || !x2 || N <= 0 || c1 < 1 || c1 > 4 || c2 < 1 || H <= 0 || W <= 0) return SELFC_EINVAL;
  const int c2p = (c2 + 3) & ~3;
  if (fd && (FC < c2p || (FC & 3))) return SELFC_EINVAL;
  const size_t HW = (size_t)H * W;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(nchw_to_latent_kernel<false>, dim3(nblocks(N * HW)), dim3(TPB), 0, (hipStream_t)stream, x, x1, x2, (f16*)fd, FC, N, c1, c2, c2p, HW, (size_t)0);
  return hip_rc(hipGetLastError());
}

int selfc_latent_to_nchw(const float* x1, const float* x2, float* y, int N, int c1, int c2, int H, int W, void* stream) {
  if (!y || !x1 || !x2 || N <= 0 || c1 < 1 || c1 > 4 || c2 < 1 || H <= 0 || W <= 0) return SELFC_EINVAL;
  const int c2p = (c2 + 3) & ~3;
  const size_t HW = (size_t)H * W;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(latent_to_nchw_kernel<false>, dim3(nblocks(N * HW)), dim3(TPB), 0, (hipStream_t)stream, x1, x2, y, N, c1, c2, c2p, HW, (size_t)0);
  return hip_rc(hipGetLastError());
}

int selfc_nchw_to_nhwc4(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SELFC_EINVAL;
  const size_t HW = (size_t)H * W;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel<false>, dim3(nblocks(N * HW)), dim3(TPB), 0, (hipStream_t)stream, x, y, N, C, (C + 3) & ~3, HW, (size_t)0);
  return hip_rc(hipGetLastError());
}

int selfc_nhwc4_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SELFC_EINVAL;
  const size_t HW = (size_t)H * W;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(nhwc4_to_nchw_kernel<false>, dim3(nblocks(N * HW)), dim3(TPB), 0, (hipStream_t)stream, x, y, N, C, (C + 3) & ~3, HW, (size_t)0);
  return hip_rc(hipGetLastError());
}

// ---- the same transforms with the tensor that belongs to the CALLER addressed through a pointer slot (abi 10) ----
namespace {
__global__ void set_pointers_kernel(void** table, const void* p0, const void* p1, const void* p2, const void* p3, const int n) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (n > 0) table[0] = const_cast<void*>(p0);
    if (n > 1) table[1] = const_cast<void*>(p1);
    if (n > 2) table[2] = const_cast<void*>(p2);
    if (n > 3) table[3] = const_cast<void*>(p3);
  }
}
}  // namespace

int selfc_set_pointers(void** table, int n, const void* p0, const void* p1, const void* p2, const void* p3, void* stream) {
  if (!table || n < 1 || n > 4) return SELFC_EINVAL;
  hipLaunchKernelGGL(set_pointers_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, table, p0, p1, p2, p3, n);
  return hip_rc(hipGetLastError());
}

int selfc_freq_fwd_ind(const float* const* xslot, size_t xoff, float* x1, float* x2, void* fd, int FC, int N, int H, int W, int k, void* stream) {
  if (!xslot || !x1 || !x2 || N <= 0 || H <= 0 || W <= 0 || (k != 4 && k != 2) || H % k || W % k) return SELFC_EINVAL;
  if (fd && (FC < 3 * k * k || (FC & 3))) return SELFC_EINVAL;
  const int h = H / k, w = W / k;
  const size_t total = (size_t)N * h * w;
  const float* xa = reinterpret_cast<const float*>(xslot);
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  if (k == 4)
    hipLaunchKernelGGL((freq_fwd_kernel<4, true>), dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, xa, x1, x2, (f16*)fd, FC, N, h, w, xoff);
  else
    hipLaunchKernelGGL((freq_fwd_kernel<2, true>), dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, xa, x1, x2, (f16*)fd, FC, N, h, w, xoff);
  return hip_rc(hipGetLastError());
}

int selfc_freq_inv_ind(const float* x1, const float* x2, float* const* xslot, size_t xoff, int N, int h, int w, int k, void* stream) {
  if (!xslot || !x1 || !x2 || N <= 0 || h <= 0 || w <= 0 || (k != 4 && k != 2)) return SELFC_EINVAL;
  const size_t total = (size_t)N * h * w;
  float* xa = reinterpret_cast<float*>(const_cast<float**>(xslot));
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  if (k == 4)
    hipLaunchKernelGGL((freq_inv_kernel<4, true>), dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, x1, x2, xa, N, h, w, xoff);
  else
    hipLaunchKernelGGL((freq_inv_kernel<2, true>), dim3(nblocks(total)), dim3(TPB), 0, (hipStream_t)stream, x1, x2, xa, N, h, w, xoff);
  return hip_rc(hipGetLastError());
}

int selfc_nchw_to_latent_ind(const float* const* xslot, size_t xoff, float* x1, float* x2, void* fd, int FC, int N, int c1, int c2, int H, int W, void* stream) {
  if (!xslot || !x1 || !x2 || N <= 0 || c1 < 1 || c1 > 4 || c2 < 1 || H <= 0 || W <= 0) return SELFC_EINVAL;
  const int c2p = (c2 + 3) & ~3;
  if (fd && (FC < c2p || (FC & 3))) return SELFC_EINVAL;
  const size_t HW = (size_t)H * W;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(nchw_to_latent_kernel<true>, dim3(nblocks(N * HW)), dim3(TPB), 0, (hipStream_t)stream, reinterpret_cast<const float*>(xslot), x1, x2,
                     (f16*)fd, FC, N, c1, c2, c2p, HW, xoff);
  return hip_rc(hipGetLastError());
}

int selfc_latent_to_nchw_ind(const float* x1, const float* x2, float* const* yslot, size_t yoff, int N, int c1, int c2, int H, int W, void* stream) {
  if (!yslot || !x1 || !x2 || N <= 0 || c1 < 1 || c1 > 4 || c2 < 1 || H <= 0 || W <= 0) return SELFC_EINVAL;
  const int c2p = (c2 + 3) & ~3;
  const size_t HW = (size_t)H * W;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(latent_to_nchw_kernel<true>, dim3(nblocks(N * HW)), dim3(TPB), 0, (hipStream_t)stream, x1, x2,
                     reinterpret_cast<float*>(const_cast<float**>(yslot)), N, c1, c2, c2p, HW, yoff);
  return hip_rc(hipGetLastError());
}

int selfc_nchw_to_nhwc4_ind(const float* const* xslot, size_t xoff, float* y, int N, int C, int H, int W, void* stream) {
  if (!xslot || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SELFC_EINVAL;
  const size_t HW = (size_t)H * W;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(nchw_to_nhwc4_kernel<true>, dim3(nblocks(N * HW)), dim3(TPB), 0, (hipStream_t)stream, reinterpret_cast<const float*>(xslot), y, N, C,
                     (C + 3) & ~3, HW, xoff);
  return hip_rc(hipGetLastError());
}

int selfc_nhwc4_to_nchw_ind(const float* x, float* const* yslot, size_t yoff, int N, int C, int H, int W, void* stream) {
  if (!x || !yslot || N <= 0 || C <= 0 || H <= 0 || W <= 0) return SELFC_EINVAL;
  const size_t HW = (size_t)H * W;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(nhwc4_to_nchw_kernel<true>, dim3(nblocks(N * HW)), dim3(TPB), 0, (hipStream_t)stream, x,
                     reinterpret_cast<float*>(const_cast<float**>(yslot)), N, C, (C + 3) & ~3, HW, yoff);
  return hip_rc(hipGetLastError());
}

int selfc_quantize_inplace_v(float* x, size_t n, float quant_v, int is_clip, void* stream) {
  if (!x || (n & 3) || !(quant_v > 0.f)) return SELFC_EINVAL;
  if (n == 0) return SELFC_OK;
  ProfScope prof(PROF_TRANSFORM, (hipStream_t)stream);
  hipLaunchKernelGGL(quantize_kernel, dim3(nblocks(n / 4)), dim3(TPB), 0, (hipStream_t)stream, x, n / 4, quant_v, is_clip ? 1 : 0);
  return hip_rc(hipGetLastError());
}

int selfc_quantize_inplace(float* x, size_t n, void* stream) { return selfc_quantize_inplace_v(x, n, 255.0f, 1, stream); }

}  // extern "C"
