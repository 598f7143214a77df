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
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = fabsf(g[i]);
    m = (v == v) ? fmaxf(m, v) : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax_bits, __float_as_uint(m));   // non-negative floats order as uints
}

// item = (pixel, 8-channel chunk of the padded planes): 16-byte stores
__global__ __launch_bounds__(256) void grad_to_planes_kernel(const float* __restrict__ g, f16* __restrict__ planes, size_t npix,
                                                             int c, int cs, int nplanes, int lrelu, float sign, const float* __restrict__ amax) {
  const float sc = amax ? sign * grad_scale(*amax) : sign;
  const size_t total = npix * (size_t)nplanes * 4;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int chunk = (int)(i % ((size_t)nplanes * 4));
  const size_t pix = i / ((size_t)nplanes * 4);
  const int ch0 = chunk * 8;
  f16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ch = ch0 + e;
    float v = (ch < c) ? g[pix * cs + ch] : 0.f;
    if (lrelu) v = lrelu02(v);
    o[e] = (f16)(v * sc);
  }
  *reinterpret_cast<f16x8*>(planes + (size_t)(ch0 >> 5) * npix * 32 + pix * 32 + (ch0 & 31)) = o;
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients
// ---------------------------------------------------------------------------------------------------------
struct WgArgs {
  const f16* P;     // gradient planes (blockIdx.z selects one)
  const f16* Q;     // activation planes (blockIdx.y selects one)
  float* part;      // [wave partial][Pn][qtot][ttot][32 o][32 c]
  size_t plane;     // halfs per plane
  int N, T, H, W, tiles_x, tiles_y, ntiles;
  int dt;           // TAPS == 1: Q is read at frame n + dt of the clip (zero outside)
  int q0, qtot, tap0, ttot;
};

typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));

__device__ __forceinline__ f16x8 tr_frag(const unsigned char* lds, const int off0, const int off1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(lds + off1));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(f16x8, v);
}

template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgArgs a) {
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int QW = 16 + 2 * HALO, QPIX = QW * QW;
  __shared__ __attribute__((aligned(16))) unsigned char lp[256 * 64];
  __shared__ __attribute__((aligned(16))) unsigned char lq[QPIX * 64];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f16* __restrict__ P = a.P + (size_t)blockIdx.z * a.plane;
  const f16* __restrict__ Q = a.Q + (size_t)blockIdx.y * a.plane;
  const int H = a.H, W = a.W;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ds_read_b64_tr_b16 addressing: lane 16g + 4q + p supplies block row q (a pixel), columns 4p..4p+3 of the
  // group's 16 channels, and receives channel (lane & 15) of the 4 pixels.  Groups g = 0,1 are the two channel
  // halves of the same pixels, g >> 1 is the MFMA k-half: pixels of patch rows 2h (first read) and 2h+1 (second).
  const int g = lane >> 4, h = g >> 1, q = (lane >> 2) & 3, p = lane & 3;
  const int choff = (16 * (g & 1) + 4 * p) * 2;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, n = tile / (a.tiles_x * a.tiles_y);
    const int tx0 = tx * 16, ty0 = ty * 16;
    const int tc = n % a.T + a.dt;
    const bool tv = (tc >= 0) & (tc < a.T);
    const int nq = tv ? n + a.dt : n;
    __syncthreads();                       // the previous tile's fragments have been read
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = tid + it * 256;
      const int px = i >> 2, ch = i & 3;
      const int y = ty0 + (px >> 4), x = tx0 + (px & 15);
      const bool ok = (y < H) & (x < W);
      const int yc = min(y, H - 1), xc = min(x, W - 1);
      const u32x4 v = *reinterpret_cast<const u32x4*>(P + ((size_t)(n * H + yc) * W + xc) * 32 + ch * 8);
      *reinterpret_cast<u32x4*>(lp + px * 64 + ch * 16) = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < QPIX * 4; i += 256) {
      const int px = i >> 2, ch = i & 3;
      const int hy = px / QW, hx = px - hy * QW;
      const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
      const bool ok = tv & (y >= 0) & (y < H) & (x >= 0) & (x < W);
      const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
      const u32x4 v = *reinterpret_cast<const u32x4*>(Q + ((size_t)(nq * H + yc) * W + xc) * 32 + ch * 8);
      *reinterpret_cast<u32x4*>(lq + px * 64 + ch * 16) = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    // 16 patches of 4x4 pixels, 4 per wave; patches wholly outside the image are skipped (wave-uniform test:
    // the transposing read needs EXEC all ones)
    for (int pi = wave; pi < 16; pi += 4) {
      const int pr = pi >> 2, pc = pi & 3;
      if (ty0 + 4 * pr >= H || tx0 + 4 * pc >= W) continue;
      const int arow = ((4 * pr + 2 * h) * 16 + 4 * pc + q) * 64 + choff;
      const f16x8 af = tr_frag(lp, arow, arow + 16 * 64);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int dy = TAPS == 9 ? t / 3 : 0, dx = TAPS == 9 ? t % 3 : 0;
        const int brow = ((4 * pr + 2 * h + dy) * QW + 4 * pc + q + dx) * 64 + choff;
        const f16x8 bf = tr_frag(lq, brow, brow + QW * 64);
        acc[t] = mfma_32x32x16(af, bf, acc[t]);
      }
    }
  }
  // D[o][c]: lane owns column c = lane & 31, rows o = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const size_t w = (size_t)blockIdx.x * 4 + wave;
  float* __restrict__ base = a.part + ((((w * gridDim.z + blockIdx.z) * a.qtot + a.q0 + blockIdx.y) * a.ttot + a.tap0) << 10);
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      base[(size_t)t * 1024 + o * 32 + (lane & 31)] = acc[t][r];
    }
}

struct FinArgs {
  const float* part;
  float* out;            // (O, Ctot, ttot) fp32, PyTorch layout of the conv weight
  int nW, Pn, qtot, ttot, O, Ctot, cin, nx;
  const float* amax;
  float beta;
};

// thread = one element of the partial block layout (coalesced reads over the nW partials), scattered write
__global__ __launch_bounds__(256) void wgrad_finish_kernel(const FinArgs a) {
  const size_t per = (size_t)a.Pn * a.qtot * a.ttot * 1024;
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= per) return;
  const int c = (int)(e & 31), oo = (int)((e >> 5) & 31);
  const size_t blk = e >> 10;
  const int tap = (int)(blk % a.ttot), qq = (int)((blk / a.ttot) % a.qtot), pz = (int)(blk / ((size_t)a.ttot * a.qtot));
  const int o = 32 * pz + oo;
  int ci;
  if (qq < a.nx) {
    ci = 32 * qq + c;
    if (ci >= a.cin) return;
  } else {
    ci = a.cin + 32 * (qq - a.nx) + c;
  }
  if (o >= a.O || ci >= a.Ctot) return;
  float sum = 0.f;
  for (int w = 0; w < a.nW; ++w) sum += a.part[(size_t)w * per + e];
  float* dst = a.out + ((size_t)o * a.Ctot + ci) * a.ttot + tap;
  const float v = sum / grad_scale(*a.amax);
  *dst = (a.beta != 0.f) ? a.beta * *dst + v : v;
}

// bias gradient: column sums of the gradient planes.  grid (nsplit, Pn); partb[(split*Pn + pz)*32 + ch]
__global__ __launch_bounds__(256) void bias_partial_kernel(const f16* __restrict__ P, size_t plane, size_t npix, float* __restrict__ partb) {
  __shared__ float red[64][33];
  const int tid = threadIdx.x, chunk = tid & 3, pl = tid >> 2;
  const f16* __restrict__ src = P + (size_t)blockIdx.y * plane;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (size_t px = (size_t)blockIdx.x * 64 + pl; px < npix; px += (size_t)gridDim.x * 64) {
    const f16x8 v = *reinterpret_cast<const f16x8*>(src + px * 32 + chunk * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[pl][chunk * 8 + e] = s[e];
  __syncthreads();
  if (tid < 32) {
    float t = 0.f;
    for (int i = 0; i < 64; ++i) t += red[i][tid];
    partb[((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 32 + tid] = t;
  }
}

__global__ void bias_finish_kernel(const float* __restrict__ partb, int nsplit, int Pn, int O, float* __restrict__ out,
                                   const float* __restrict__ amax, float beta) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= O) return;
  float t = 0.f;
  for (int s = 0; s < nsplit; ++s) t += partb[((size_t)s * Pn + (o >> 5)) * 32 + (o & 31)];
  const float v = t / grad_scale(*amax);
  out[o] = (beta != 0.f) ? beta * out[o] + v : v;
}

// ---------------------------------------------------------------------------------------------------------
// coupling and FrequencyAnalyzer gradients (elementwise / index shuffles, fp32)
// ---------------------------------------------------------------------------------------------------------
// s = clamp*(2 sigmoid(h) - 1)  =>  ds/dh = clamp*(1 - (s/clamp)^2)/2.
// rev == 0 (y2 = x2*e^s + g, v = x2):   dx2 = dy2*e^s,   dh = dy2*x2*e^s * ds/dh,   dg = dy2
// rev != 0 (y2 = (x2-g)*e^-s, v = y2):  dx2 = dy2*e^-s,  dh = -dy2*y2 * ds/dh,      dg = -dx2
__global__ __launch_bounds__(256) void coupling_bwd_kernel(int rev, const float4* __restrict__ v, const float4* __restrict__ s,
                                                           const float4* __restrict__ dy2, float4* __restrict__ dx2,
                                                           float4* __restrict__ dh, float clamp, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
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
  int nx, ng, hasx, nsplit, bsplit;
  size_t plane_b;       // bytes of one f16 plane
  size_t off_g, off_t5, off_xplane, off_amax, off_partb, off_part, total;
};

BwdLayout bwd_layout(int N, int H, int W, int cin, int cout) {
  BwdLayout L{};
  L.nx = (cin + 31) / 32;
  L.ng = (cout + 31) / 32;
  L.hasx = cin <= 3;
  const size_t npix = (size_t)N * H * W;
  L.plane_b = npix * 64;
  const int ntiles = N * ((H + 15) / 16) * ((W + 15) / 16);
  (void)ntiles;
  L.nsplit = bwd_wgrad_nsplit(N, H, W);
  L.bsplit = BWD_BSPLIT;
  L.off_g = 4 * L.plane_b;
  L.off_t5 = L.off_g + (size_t)L.ng * L.plane_b;
  L.off_xplane = L.off_t5 + (size_t)(L.nx + 3) * L.plane_b;
  L.off_amax = L.off_xplane + (L.hasx ? L.plane_b : 0);
  L.off_partb = up256(L.off_amax + 256);
  L.off_part = up256(L.off_partb + (size_t)L.bsplit * L.ng * 32 * sizeof(float));
  // largest partial: conv4 (1 P plane, nx+3 Q planes, 9 taps) or conv5 (ng P planes, nx+4 Q planes, 9 | 3 taps)
  const size_t a4 = (size_t)(L.nx + 3) * 9, a5 = (size_t)L.ng * (L.nx + 4) * 9;
  L.total = up256(L.off_part + (size_t)L.nsplit * 4 * (a4 > a5 ? a4 : a5) * 4096);
  return L;
}

template <int TAPS>
int launch_wgrad(const WgArgs& a, int nsplit, int Qn, int Pn, hipStream_t s) {
  hipLaunchKernelGGL(wgrad_kernel<TAPS>, dim3((unsigned)nsplit, (unsigned)Qn, (unsigned)Pn), dim3(256), 0, s, a);
  return hip_rc(hipGetLastError());
}

}  // namespace

namespace selfc {

int bwd_absmax(const float* g, size_t n, float* amax, hipStream_t s) {
  int rc = hip_rc(hipMemsetAsync(amax, 0, sizeof(float), s));
  if (rc) return rc;
  const size_t nb = (n + 256 * 16 - 1) / (256 * 16);
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)(nb < 1 ? 1 : (nb > 2048 ? 2048 : nb))), dim3(256), 0, s, g, n, (unsigned*)amax);
  return hip_rc(hipGetLastError());
}

int bwd_to_planes(const float* x, void* planes, size_t npix, int c, int cs, int nplanes, int lrelu, float sign,
                  const float* amax, hipStream_t s) {
  const size_t items = npix * (size_t)nplanes * 4;
  hipLaunchKernelGGL(grad_to_planes_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s,
                     x, (f16*)planes, npix, c, cs, nplanes, lrelu, sign, amax);
  return hip_rc(hipGetLastError());
}

int bwd_wgrad_nsplit(int N, int H, int W) {
  const long ntiles = (long)N * ((H + 15) / 16) * ((W + 15) / 16);
  return (int)(ntiles < 24 ? ntiles : 24);
}

size_t bwd_wgrad_part_bytes(int nsplit, int Pn, int qtot, int ttot) {
  return (size_t)nsplit * 4 * Pn * qtot * ttot * 4096;
}

int bwd_wgrad(const WgradJob& j, const float* amax, float* part, float* partb, int nsplit, int N, int T, int H, int W, hipStream_t s) {
  const size_t npix = (size_t)N * H * W, plane = npix * 32;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
  const int qtot = j.Qn[0] + j.Qn[1];
  const int ttot = j.taps == 9 ? 9 : (j.temporal ? 3 : 1);
  int rc;
  if (j.wout) {
    for (int dt = j.temporal ? -1 : 0; dt <= (j.temporal ? 1 : 0); ++dt) {
      WgArgs a{};
      a.P = (const f16*)j.P; a.part = part; a.plane = plane;
      a.N = N; a.T = T; a.H = H; a.W = W; a.tiles_x = tiles_x; a.tiles_y = tiles_y; a.ntiles = N * tiles_x * tiles_y;
      a.dt = dt; a.qtot = qtot; a.ttot = ttot; a.tap0 = j.temporal ? dt + 1 : 0;
      int q0 = 0;
      for (int r = 0; r < 2; ++r) {
        if (j.Qn[r] <= 0) continue;
        a.Q = (const f16*)j.Q[r]; a.q0 = q0;
        rc = j.taps == 9 ? launch_wgrad<9>(a, nsplit, j.Qn[r], j.Pn, s) : launch_wgrad<1>(a, nsplit, j.Qn[r], j.Pn, s);
        if (rc) return rc;
        q0 += j.Qn[r];
      }
    }
    FinArgs f{};
    f.part = part; f.out = j.wout; f.nW = nsplit * 4; f.Pn = j.Pn; f.qtot = qtot; f.ttot = ttot;
    f.O = j.O; f.Ctot = j.Ctot; f.cin = j.cin; f.nx = j.nx; f.amax = amax; f.beta = j.beta;
    const size_t per = (size_t)j.Pn * qtot * ttot * 1024;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, s, f);
    if ((rc = hip_rc(hipGetLastError()))) return rc;
  }
  if (j.bout) {
    hipLaunchKernelGGL(bias_partial_kernel, dim3((unsigned)BWD_BSPLIT, (unsigned)j.Pn), dim3(256), 0, s, (const f16*)j.P, plane, npix, partb);
    hipLaunchKernelGGL(bias_finish_kernel, dim3((unsigned)((j.O + 63) / 64)), dim3(64), 0, s, partb, BWD_BSPLIT, j.Pn, j.O, j.bout, amax, j.beta);
    if ((rc = hip_rc(hipGetLastError()))) return rc;
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
  float* partb = (float*)(sb + L.off_partb);
  float* part = (float*)(sb + L.off_part);
  const f16* dn = (const f16*)dense;
  const f16* feat = dn + (size_t)(L.hasx ? 0 : L.nx) * plane;      // f1..f4
  const int coutp = (cout + 3) & ~3, cinp = (cin + 3) & ~3;
  const bool d2dt = kind == SELFC_SUBNET_D2DT;
  int rc;

  // 1. scale + scaled f16 planes of dOut
  if ((rc = bwd_absmax(dout, npix * coutp, amax, s))) return rc;
  if ((rc = bwd_to_planes(dout, gpl, npix, cout, coutp, L.ng, 0, sign, amax, s))) return rc;
  if (L.hasx && (rc = selfc_nhwc_to_planes(xin, xpl, npix, cin, stream))) return rc;

  // 2. conv5^T(dOut): x-groups and f1..f3 as addend planes, f4 masked straight into dpre4
  {
    BwdConv c{};
    c.in = gpl; c.nplanes_in = L.ng; c.kt = d2dt ? 3 : 1; c.sp1 = d2dt ? 1 : 0; c.w = bw->wt5;
    c.ngroups = L.nx + 4; c.out_planes = t5;
    c.mask = feat + 3 * plane; c.mask_z = L.nx + 3; c.alt = gb;
    c.amax = amax;
    if ((rc = bwd_conv_planes(c, N, T, H, W, s))) return rc;
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
    c.plain = dx; c.coutp = cinp; c.accumulate = accumulate_dx; c.amax = amax;
    if ((rc = bwd_conv_planes(c, N, T, H, W, s))) return rc;
  }
  if (!wgrad && !bgrad) return SELFC_OK;

  // 5. weight / bias gradients
  for (int k = 1; k <= 5; ++k) {
    const int nfeat = k <= 4 ? k - 1 : 4;
    WgradJob j{};
    j.P = k <= 4 ? gb + (size_t)(4 - k) * plane : gpl;
    j.Pn = k <= 4 ? 1 : L.ng;
    // input planes: [x planes][f1..]; with cin <= 3 the x plane is the scratch copy and the features start `dense`
    if (L.hasx) { j.Q[0] = xpl; j.Qn[0] = 1; j.Q[1] = feat; j.Qn[1] = nfeat; }
    else { j.Q[0] = dn; j.Qn[0] = L.nx + nfeat; }
    j.temporal = (k == 5 && d2dt) ? 1 : 0;
    j.taps = j.temporal ? 1 : 9;
    j.wout = wgrad ? wgrad[k - 1] : nullptr;
    j.bout = bgrad ? bgrad[k - 1] : nullptr;
    j.O = k <= 4 ? 32 : cout; j.Ctot = cin + 32 * nfeat; j.cin = cin; j.nx = L.nx; j.beta = beta;
    if ((rc = bwd_wgrad(j, amax, part, partb, L.nsplit, N, T, H, W, s))) return rc;
  }
  return SELFC_OK;
}

int selfc_coupling_bwd(int rev, const float* v, const float* s, const float* dy2, float* dx2, float* dh, float clamp,
                       size_t n, void* stream) {
  if (!v || !s || !dy2 || !dx2 || !dh || n == 0 || (n & 3) || clamp == 0.f) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(coupling_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rev,
                     (const float4*)v, (const float4*)s, (const float4*)dy2, (float4*)dx2, (float4*)dh, clamp, n4);
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
  return up256((size_t)BWD_BSPLIT * Pn * 32 * sizeof(float)) + bwd_wgrad_part_bytes(bwd_wgrad_nsplit(N, H, W), Pn, Qn, taps);
}

int selfc_bwd_wgrad(const void* P, int Pn, const void* Q, int Qn, int taps, float* wout, int O, int Ctot, float* bout, float beta,
                    const float* amax, void* scratch, size_t scratch_bytes, int N, int T, int H, int W, void* stream) {
  if (!P || !Q || !amax || !scratch || Pn < 1 || Qn < 1 || (taps != 1 && taps != 9)) return SELFC_EINVAL;
  if (O < 1 || O > 32 * Pn || Ctot < 1 || Ctot > 32 * Qn || N <= 0 || T <= 0 || N % T || H <= 0 || W <= 0) return SELFC_EINVAL;
  if (scratch_bytes < selfc_bwd_wgrad_scratch_bytes(N, H, W, Pn, Qn, taps)) return SELFC_EINVAL;
  ProfScope prof(PROF_BWD, (hipStream_t)stream);
  unsigned char* sb = (unsigned char*)scratch;
  WgradJob j{};
  j.P = P; j.Pn = Pn; j.Q[0] = Q; j.Qn[0] = Qn; j.taps = taps; j.temporal = 0;
  j.wout = wout; j.O = O; j.Ctot = Ctot; j.cin = Ctot; j.nx = Qn; j.bout = bout; j.beta = beta;
  return bwd_wgrad(j, amax, (float*)(sb + up256((size_t)BWD_BSPLIT * Pn * 32 * sizeof(float))), (float*)sb,
                   bwd_wgrad_nsplit(N, H, W), N, T, H, W, (hipStream_t)stream);
}

}  // extern "C"
