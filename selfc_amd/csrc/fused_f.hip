// Pairwise-fused conv1..conv4 of the F dense block (D2DTInput / DenseBlock with cin = 48,
// Subnet_constructor.py:27-30,126-129) for gfx950.
//
// F's four 3x3 convs read the concat [x2 (48) | f1 | f2 | f3]; fused all four at once the 48-channel
// input halo does not fit the CU's LDS next to three feature images.  Two launches do:
//
//   pair 0 : conv1 on the 18x18 region around a 16x16 tile (halo recompute) -> FM image in LDS,
//            conv2 on the tile from [x2 | FM];           inputs x2 (48 ch) on 20x20,  writes f1, f2
//   pair 1 : conv3 on 18x18 -> FM, conv4 on the tile;    inputs [x2 | f1 | f2] (112 ch) on 20x20, writes f3, f4
//
// One persistent 512-thread workgroup per CU walks tiles.  Per tile the 20x20 input halo sits in LDS
// ("image", 16 k-channels = 32 B pieces, pixel pitch 112 / 240 B); the first conv of the pair and the
// input part of the second conv consume the SAME activation fragment with two weight fragments
// ("merged" steps: 1 B read feeds 2 MFMAs), the 68 ring pixels of the 18x18 region are three extra
// M-tiles on waves 0..2; then the second conv's last 18 steps read FM.  Pair 0 keeps its whole 72-KiB
// weight stream resident in LDS (no per-chunk barrier), pair 1 streams its 144 KiB through an LDS double
// buffer in 18-fragment chunks like csrc/fused_gh.hip.  The next tile's input halo is prefetched into
// registers while the current tile computes.
//
// LDS banking (MI355X_MICROARCH.md, LDS): ds_read_b128 is served in lane groups {0-3,12-15,20-27},... -
// an M-tile is 2 rows x 16 columns; with a row pitch of (16 q + 1) sixteen-byte slots the ring's COLUMN
// pixels (stride = one row) fall on 16 different slots, and rotating the second row's column order by
// kappa = (pitch/16)^-1 mod 16 keeps the two rows of a lane group on disjoint slots.
#include <stdio.h>
#include <stdlib.h>
#include "common.hpp"
#include "prof.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace selfc {

struct FFArgs {
  f16* dense;               // F dense buffer, plane-blocked [6][N][H][W][32]: x2 (2 planes), f1..f4
  const f16* w;             // fragment stream of this pair (packing.py: pack_fused_f)
  const float* bias[2];     // 32 floats: first / second conv of the pair
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
  size_t plane;
};

namespace {

constexpr int TS = 16, IS = 20, FS = 18;
constexpr int NWAVE = 8, NTHR = NWAVE * 64;
constexpr int NRING = FS * FS - TS * TS;          // 68 ring pixels
constexpr int CHF = 18;                           // fragments per streamed weight chunk

template <int PAIR>
struct Geo {
  static constexpr int NIN = PAIR ? 7 : 3;                 // 16-channel k-steps per input pixel
  static constexpr int NCHK = 2 * NIN;                     // 16-byte pieces per input pixel
  static constexpr int PITCH = NCHK * 16 + 16;             // 112 / 240 B: odd number of 16-B slots
  static constexpr int KAPPA = PAIR ? 15 : 7;              // (PITCH/16)^-1 mod 16  (x 1 slot of row skew)
  static constexpr int ROWP = PAIR ? 4880 : 2320;          // >= IS*PITCH, = 16 (16 q + 1)
  static constexpr int FROW = PAIR ? 1456 : 1584;          // FM row pitch: >= FS*PS, slots = 5*KAPPA mod 16
  static constexpr int IMG_BYTES = IS * ROWP;
  static constexpr int FM_BYTES = FS * FROW;
  static constexpr int S1 = 9 * NIN;                       // merged steps (tap-major, k-step minor)
  static constexpr int NFRAG = 2 * S1 + 18;                // 72 / 144
  static constexpr bool RES = PAIR == 0;                   // whole stream resident in LDS
  static constexpr int RD = 3;                             // operand ring depth (fragments fetched RD-1 steps ahead)
  static constexpr int W_BYTES = RES ? NFRAG * 1024 : 2 * CHF * 1024;
  static constexpr int OFF_IMG = 0, OFF_FM = IMG_BYTES, OFF_W = OFF_FM + FM_BYTES, OFF_B = OFF_W + W_BYTES;
  static constexpr int LDS = OFF_B + 256;
  static constexpr int NITEM = IS * IS * NCHK;             // 16-byte pieces of one input halo
  static constexpr int NPLANE_IN = PAIR ? 4 : 2;
  static constexpr int OUT1 = PAIR ? 4 : 2, OUT2 = OUT1 + 1;   // planes the two convs append
  static_assert(ROWP >= IS * PITCH && (ROWP / 16) % 16 == 1, "image row pitch");
  static_assert(FROW >= FS * PS && (FROW / 16) % 16 == (5 * KAPPA) % 16, "FM row pitch");
  static_assert(((PITCH / 16) * KAPPA) % 16 == 1, "kappa");
  static_assert(LDS <= 160 * 1024, "LDS budget");
  static_assert(NFRAG % CHF == 0 && (2 * S1) % CHF == 0, "chunking");
};

struct Lane {
  int tid, lane, wave, half;
  int py, px;            // this lane's pixel of the wave's centre M-tile (tile coordinates)
  int rr, rc;            // ring pixel (region coordinates), waves 0..2
  bool rvalid;
};

// ---- one run of merged steps [G0, G0+NS): acc1c += W1 B, acc2 += W2 B, (RING) acc1r += W1 Br ----------
// wl: LDS address of the fragment pair of step G0 (+ lane*16); fragments alternate [W1 step][W2 step].
template <int PAIR, int G0, int NS, bool RING>
__device__ __forceinline__ void merged_steps(const unsigned char* __restrict__ wl, const unsigned char* __restrict__ bc,
                                             const unsigned char* __restrict__ br, f32x16& acc1c, f32x16& acc1r, f32x16& acc2) {
  using G = Geo<PAIR>;
  constexpr int RD = G::RD;
  f16x8 rA1[RD], rA2[RD], rBc[RD], rBr[RD];
  auto load_step = [&](const int st) __attribute__((always_inline)) {
    // K order: source group (x2: 3 k-steps per tap; f1, f2: 2 each), tap-major inside a group
    const int g = G0 + st;
    const int tap = g < 27 ? g / 3 : g < 45 ? (g - 27) >> 1 : (g - 45) >> 1;
    const int ks = g < 27 ? g % 3 : g < 45 ? 3 + ((g - 27) & 1) : 5 + ((g - 45) & 1);
    const int off = (tap / 3) * G::ROWP + (tap % 3) * G::PITCH + ks * 32;
    const int s = st % RD;
    rA1[s] = *reinterpret_cast<const f16x8*>(wl + (2 * st) * 1024);
    rA2[s] = *reinterpret_cast<const f16x8*>(wl + (2 * st + 1) * 1024);
    rBc[s] = *reinterpret_cast<const f16x8*>(bc + off);
    if (RING) rBr[s] = *reinterpret_cast<const f16x8*>(br + off);
  };
  load_step(0);
  if (RD > 2 && NS > 1) load_step(1);
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    if (st + RD - 1 < NS) load_step(st + RD - 1);
    __builtin_amdgcn_sched_barrier(0);
    const int s = st % RD;
    acc1c = mfma_32x32x16(rA1[s], rBc[s], acc1c);
    acc2 = mfma_32x32x16(rA2[s], rBc[s], acc2);
    if (RING) acc1r = mfma_32x32x16(rA1[s], rBr[s], acc1r);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the 18 FM steps of the second conv --------------------------------------------------------------
template <int PAIR>
__device__ __forceinline__ void fm_steps(const unsigned char* __restrict__ wl, const unsigned char* __restrict__ fb, f32x16& acc2) {
  using G = Geo<PAIR>;
  f16x8 rA[3], rB[3];
  auto load_step = [&](const int st) __attribute__((always_inline)) {
    const int tap = st >> 1, ks = st & 1;
    rA[st % 3] = *reinterpret_cast<const f16x8*>(wl + st * 1024);
    rB[st % 3] = *reinterpret_cast<const f16x8*>(fb + (tap / 3) * G::FROW + (tap % 3) * PS + ks * 32);
  };
  load_step(0);
  load_step(1);
#pragma unroll
  for (int st = 0; st < 18; ++st) {
    if (st + 2 < 18) load_step(st + 2);
    __builtin_amdgcn_sched_barrier(0);
    acc2 = mfma_32x32x16(rA[st % 3], rB[st % 3], acc2);
    __builtin_amdgcn_sched_barrier(0);
  }
}

__device__ __forceinline__ f32x16 bias_init(const float* bl, const int half) {
  f32x16 b;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(bl + 8 * g + 4 * half);
    b[4 * g + 0] = v.x; b[4 * g + 1] = v.y; b[4 * g + 2] = v.z; b[4 * g + 3] = v.w;
  }
  return b;
}

// bias is already in the accumulator: LeakyReLU, f16, half-swap -> v[gp] = 8 contiguous channels 16 gp + 8 half
__device__ __forceinline__ void lrelu_pack(const f32x16& acc, const bool keep, u32x4 (&v)[2]) {
  uint32_t r[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    r[g][0] = pack2(lrelu02(acc[4 * g + 0]), lrelu02(acc[4 * g + 1]));
    r[g][1] = pack2(lrelu02(acc[4 * g + 2]), lrelu02(acc[4 * g + 3]));
  }
  const uint32_t m = keep ? 0xffffffffu : 0u;
#pragma unroll
  for (int gp = 0; gp < 2; ++gp)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const auto sw = __builtin_amdgcn_permlane32_swap(r[2 * gp][d], r[2 * gp + 1][d], false, false);
      v[gp][d] = sw[0] & m;
      v[gp][2 + d] = sw[1] & m;
    }
}

template <int PAIR>
__global__ __launch_bounds__(NTHR) void fused_f_kernel(const FFArgs a) {
  using G = Geo<PAIR>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Lane c;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave = c.tid >> 6;
  c.half = c.lane >> 5;
  {
    const int i = c.lane & 15, row2 = (c.lane >> 4) & 1;
    c.py = 2 * c.wave + row2;
    c.px = row2 ? ((i - G::KAPPA) & 15) : i;
    const int rho = 32 * c.wave + (c.lane & 31);
    c.rvalid = (c.wave < 3) & (rho < NRING);
    const int q = min(rho, NRING - 1);
    c.rr = q < 18 ? 0 : q < 36 ? 17 : q < 52 ? q - 35 : q - 51;
    c.rc = q < 18 ? q : q < 36 ? q - 18 : q < 52 ? 0 : 17;
  }
  const int total = a.ntiles * a.N;
  int t = blockIdx.x;
  if (t >= total) return;
  const bool ring = c.wave < 3;
  const u32x4* __restrict__ wsrc = reinterpret_cast<const u32x4*>(a.w);

  // ---- next-tile input halo: global -> registers (x_load), registers -> LDS image (x_store) ----------
  // Loaded in PARTS that follow the K order (part 0: x2 = planes 0, 1; part 1: f1; part 2: f2): a part of the
  // image is dead as soon as every wave has finished that source group's steps, so pair 1 refills it and starts
  // the next part's loads at that chunk barrier - 5 staging registers instead of 11.
  constexpr int XMAX = 5;
  u32x4 xv[XMAX];
  unsigned xok = 0;
  auto item = [&](const int part, const int it, int& seg, int& piece, int& hy, int& hx) __attribute__((always_inline)) {
    // pieces are enumerated plane by plane (coalesced 64-byte pixels): plane 0: 4, plane 1: 2, planes 2, 3: 4 per pixel
    int p;
    if (part == 0) {
      const int i = min(c.tid + it * NTHR, 2400 - 1);
      if (i < 1600) { seg = 0; p = i >> 2; piece = i & 3; }
      else { seg = 1; p = (i - 1600) >> 1; piece = i & 1; }
    } else {
      const int i = min(c.tid + it * NTHR, 1600 - 1);
      seg = part + 1; p = i >> 2; piece = i & 3;
    }
    hy = p / IS;
    hx = p - hy * IS;
  };
  auto x_load = [&](const int part, const int tile) __attribute__((always_inline)) {
    xok = 0;
    const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, n = tile / a.ntiles;
    const int niter = part == 0 ? 5 : 4;
#pragma unroll
    for (int it = 0; it < XMAX; ++it) {
      if (it < niter) {
        int seg, piece, hy, hx;
        item(part, it, seg, piece, hy, hx);
        const int y = ty * TS + hy - 2, x = tx * TS + hx - 2;
        const bool ok = (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
        const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
        xv[it] = *reinterpret_cast<const u32x4*>(a.dense + (size_t)seg * a.plane + ((size_t)(n * a.H + yc) * a.W + xc) * 32 + piece * 8);
        xok |= (ok ? 1u : 0u) << it;
      }
    }
  };
  auto x_store = [&](const int part) __attribute__((always_inline)) {
    const int niter = part == 0 ? 5 : 4, nitem = part == 0 ? 2400 : 1600;
#pragma unroll
    for (int it = 0; it < XMAX; ++it) {
      if (it < niter && c.tid + it * NTHR < nitem) {
        int seg, piece, hy, hx;
        item(part, it, seg, piece, hy, hx);
        const int slot = (seg == 0 ? 0 : seg == 1 ? 4 : seg == 2 ? 6 : 10) + piece;
        *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + hy * G::ROWP + hx * G::PITCH + slot * 16) =
            ((xok >> it) & 1u) ? xv[it] : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  constexpr int NPART = PAIR ? 3 : 1;

  // ---- weights ------------------------------------------------------------------------------------------
  constexpr int WITER = (CHF * 64 + NTHR - 1) / NTHR;   // 3 (streamed chunks)
  u32x4 wreg[WITER];
  auto w_prefetch = [&](const int chunk) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < WITER; ++it) {
      const int i = min(c.tid + it * NTHR, CHF * 64 - 1);
      wreg[it] = wsrc[chunk * CHF * 64 + i];
    }
  };
  auto w_commit = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < WITER; ++it) {
      const int i = c.tid + it * NTHR;
      if (i < CHF * 64) *reinterpret_cast<u32x4*>(smem + G::OFF_W + buf * CHF * 1024 + i * 16) = wreg[it];
    }
  };

  // ---- prologue ---------------------------------------------------------------------------------------
  if (c.tid < 64) reinterpret_cast<float*>(smem + G::OFF_B)[c.tid] = (c.tid < 32 ? a.bias[0] : a.bias[1])[c.tid & 31];
  x_load(0, t);
  if (G::RES) {
    for (int i = c.tid; i < G::NFRAG * 64; i += NTHR) *reinterpret_cast<u32x4*>(smem + G::OFF_W + i * 16) = wsrc[i];
  } else {
    w_prefetch(0);
    w_commit(0);
  }
  x_store(0);
#pragma unroll
  for (int part = 1; part < NPART; ++part) {
    x_load(part, t);
    x_store(part);
  }
  __syncthreads();

  const unsigned char* const img = smem + G::OFF_IMG;
  const unsigned char* const bc = img + (c.py + 1) * G::ROWP + (c.px + 1) * G::PITCH + c.half * 16;
  const unsigned char* const br = img + c.rr * G::ROWP + c.rc * G::PITCH + c.half * 16;
  const unsigned char* const fb = smem + G::OFF_FM + c.py * G::FROW + c.px * PS + c.half * 16;
  const float* const lb = reinterpret_cast<const float*>(smem + G::OFF_B);
  int par = 0;   // streamed mode: buffer holding the current chunk

  for (; t < total; t += gridDim.x) {
    const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, n = t / a.ntiles;
    const int ty0 = ty * TS, tx0 = tx * TS;
    const int tnext = t + gridDim.x;
    const bool more = tnext < total;
    if (more) x_load(0, tnext);                    // lands while this tile computes

    f32x16 acc1c = bias_init(lb, c.half), acc1r = acc1c, acc2 = bias_init(lb + 32, c.half);

    // ---- merged phase -------------------------------------------------------------------------------
    if (G::RES) {
      const unsigned char* wl = smem + G::OFF_W + c.lane * 16;
      if (ring) merged_steps<PAIR, 0, G::S1, true>(wl, bc, br, acc1c, acc1r, acc2);
      else merged_steps<PAIR, 0, G::S1, false>(wl, bc, br, acc1c, acc1r, acc2);
    } else {
      constexpr int NMC = 2 * G::S1 / CHF;         // 7 merged chunks of 9 steps
#pragma unroll
      for (int ch = 0; ch < NMC; ++ch) {
        w_prefetch(ch + 1);
        const unsigned char* wl = smem + G::OFF_W + par * CHF * 1024 + c.lane * 16;
        if (ring) {
          switch (ch) {   // compile-time step base per chunk
            case 0: merged_steps<PAIR, 0, 9, true>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 1: merged_steps<PAIR, 9, 9, true>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 2: merged_steps<PAIR, 18, 9, true>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 3: merged_steps<PAIR, 27, 9, true>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 4: merged_steps<PAIR, 36, 9, true>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 5: merged_steps<PAIR, 45, 9, true>(wl, bc, br, acc1c, acc1r, acc2); break;
            default: merged_steps<PAIR, 54, 9, true>(wl, bc, br, acc1c, acc1r, acc2); break;
          }
        } else {
          switch (ch) {
            case 0: merged_steps<PAIR, 0, 9, false>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 1: merged_steps<PAIR, 9, 9, false>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 2: merged_steps<PAIR, 18, 9, false>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 3: merged_steps<PAIR, 27, 9, false>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 4: merged_steps<PAIR, 36, 9, false>(wl, bc, br, acc1c, acc1r, acc2); break;
            case 5: merged_steps<PAIR, 45, 9, false>(wl, bc, br, acc1c, acc1r, acc2); break;
            default: merged_steps<PAIR, 54, 9, false>(wl, bc, br, acc1c, acc1r, acc2); break;
          }
        }
        if (ch + 1 < NMC) {
          w_commit(par ^ 1);
          __syncthreads();
          par ^= 1;
          // x2 steps end with chunk 2, f1 steps with chunk 4: refill that part of the image, start the next part's loads
          if (ch == 2 && more) { x_store(0); x_load(1, tnext); }
          if (ch == 4 && more) { x_store(1); x_load(2, tnext); }
        }
      }
    }

    // ---- epilogue of the first conv: FM image (zero outside the frame) + centre crop to HBM ----------
    {
      f16* __restrict__ dplane = a.dense + (size_t)G::OUT1 * a.plane;
      u32x4 v[2];
      const int y = ty0 + c.py, x = tx0 + c.px;
      const bool in = (y < a.H) & (x < a.W);
      lrelu_pack(acc1c, in, v);
      unsigned char* fdst = smem + G::OFF_FM + (c.py + 1) * G::FROW + (c.px + 1) * PS + 16 * c.half;
      *reinterpret_cast<u32x4*>(fdst) = v[0];
      *reinterpret_cast<u32x4*>(fdst + 32) = v[1];
      if (in) {
        f16* d = dplane + ((size_t)(n * a.H + y) * a.W + x) * 32 + 8 * c.half;
        *reinterpret_cast<u32x4*>(d) = v[0];
        *reinterpret_cast<u32x4*>(d + 16) = v[1];
      }
      if (ring) {
        const int ry = ty0 + c.rr - 1, rx = tx0 + c.rc - 1;
        const bool rin = (ry >= 0) & (ry < a.H) & (rx >= 0) & (rx < a.W);
        lrelu_pack(acc1r, rin, v);
        if (c.rvalid) {
          unsigned char* rdst = smem + G::OFF_FM + c.rr * G::FROW + c.rc * PS + 16 * c.half;
          *reinterpret_cast<u32x4*>(rdst) = v[0];
          *reinterpret_cast<u32x4*>(rdst + 32) = v[1];
        }
      }
    }
    if (!G::RES) w_commit(par ^ 1);                 // the FM chunk
    __syncthreads();                                // FM complete; every wave is done with the input image
    if (!G::RES) par ^= 1;

    // ---- FM phase + epilogue of the second conv -------------------------------------------------------
    if (!G::RES) w_prefetch(0);                     // wraps to the next tile's first chunk
    fm_steps<PAIR>(smem + G::OFF_W + (G::RES ? 2 * G::S1 * 1024 : par * CHF * 1024) + c.lane * 16, fb, acc2);
    {
      const int y = ty0 + c.py, x = tx0 + c.px;
      const bool in = (y < a.H) & (x < a.W);
      u32x4 v[2];
      lrelu_pack(acc2, true, v);
      if (in) {
        f16* d = a.dense + (size_t)G::OUT2 * a.plane + ((size_t)(n * a.H + y) * a.W + x) * 32 + 8 * c.half;
        *reinterpret_cast<u32x4*>(d) = v[0];
        *reinterpret_cast<u32x4*>(d + 16) = v[1];
      }
    }
    if (more) x_store(NPART - 1);
    if (!G::RES) w_commit(par ^ 1);
    __syncthreads();                                // image of the next tile visible; FM free again
    if (!G::RES) par ^= 1;
  }
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

template <int PAIR>
int launch_pair(FFArgs& a, int maxwg, hipStream_t s) {
  using G = Geo<PAIR>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_f_kernel<PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return hip_rc(e);
    attr_done = true;
  }
  const int total = a.ntiles * a.N;
  // persistent workgroups, every one walks the same number of tiles (no straggler round)
  const int rounds = (total + maxwg - 1) / maxwg;
  const int gx = (total + rounds - 1) / rounds;
  hipLaunchKernelGGL(fused_f_kernel<PAIR>, dim3((unsigned)gx), dim3(NTHR), G::LDS, s, a);
  return hip_rc(hipGetLastError());
}

}  // namespace

// conv1..conv4 of F (cin = 48) on its dense buffer: two launches.  w = [pair 0: 72 fragments][pair 1: 144 fragments].
int launch_fused_f(void* dense, const void* w, const float* const* bias, int N, int H, int W, hipStream_t s) {
  static const int maxwg = getenv("SELFC_FUSEDF_MAXWG") ? atoi(getenv("SELFC_FUSEDF_MAXWG")) : 256;
  FFArgs a{};
  a.dense = (f16*)dense;
  a.N = N; a.H = H; a.W = W;
  a.tiles_x = (W + TS - 1) / TS;
  a.tiles_y = (H + TS - 1) / TS;
  a.ntiles = a.tiles_x * a.tiles_y;
  a.plane = (size_t)N * H * W * 32;
  ProfScope prof(PROF_CONV3X3, s);
  a.w = (const f16*)w;
  a.bias[0] = bias[0]; a.bias[1] = bias[1];
  int rc = launch_pair<0>(a, maxwg > 0 ? maxwg : 256, s);
  if (rc) return rc;
  a.w = (const f16*)w + (size_t)Geo<0>::NFRAG * 512;
  a.bias[0] = bias[2]; a.bias[1] = bias[3];
  return launch_pair<1>(a, maxwg > 0 ? maxwg : 256, s);
}

}  // namespace selfc
