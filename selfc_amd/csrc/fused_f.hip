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
// weight stream resident in LDS (no per-chunk barrier), pair 1 streams its 144 KiB through three LDS
// buffers of 12 fragments with the chunk barrier in the MIDDLE of a chunk, so the operand ring never drains
// at a chunk boundary (see the kernel).  The next tile's input halo is prefetched into registers while the
// current tile computes.
//
// LDS banking (MI355X_MICROARCH.md, LDS): ds_read_b128 is served in lane groups {0-3,12-15,20-27},... -
// an M-tile is 2 rows x 16 columns; with a row pitch of (16 q + 1) sixteen-byte slots the ring's COLUMN
// pixels (stride = one row) fall on 16 different slots, and rotating the second row's column order by
// kappa = (pitch/16)^-1 mod 16 keeps the two rows of a lane group on disjoint slots.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
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
  const f16* w5p;           // optional: conv5 partial-product fragments of this pair (pair 0: x2, f1, f2 = 7; pair 1: f3, f4 = 4)
  float* pf;                // optional: partial products of this pair, fp32 [3 taps][N][H][W][4]
  int store_feat;           // 1: the pair's two feature planes go to HBM; 0 (pair 1 on the inference path with partial
                            //    products): nothing reads f3 / f4 afterwards - they stay in registers / the FM image
  unsigned long long* stamps;   // diagnostic build only (-DSELFC_STAMPS): per wave 7 phase sums + lifetime
};

namespace {

constexpr int TS = 16, IS = 20, FS = 18;
static_assert(IS == 20, "x_store_item_from divides by IS with a 16-bit reciprocal");
constexpr int NWAVE = 8, NTHR = NWAVE * 64;
constexpr int NRING = FS * FS - TS * TS;          // 68 ring pixels
constexpr int CHF = 18;                           // fragments per streamed weight chunk

#ifdef SELFC_STAMPS
#define STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define STAMP_ADD(k, a, b) phase[k] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(k, a, b)
#endif

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
  static constexpr int W_BYTES = RES ? NFRAG * 1024 : 3 * 12 * 1024;   // resident stream | three 12-fragment chunk buffers
  static constexpr int OFF_IMG = 0, OFF_FM = IMG_BYTES, OFF_W = OFF_FM + FM_BYTES, OFF_B = OFF_W + W_BYTES;
  // conv5 partial-product A fragments (rows 4 tap + oc, 11 used): pair 0 keeps its 7 whole, pair 1 (LDS-bound) its 4
  // compacted to 12 rows x 32 B (rows >= 11 of a fragment are zero: those lanes read the shared zero row)
  static constexpr int NP = PAIR ? 4 : 7;
  static constexpr int OFF_P = OFF_B + 256;
  static constexpr int P_BYTES = PAIR ? NP * 384 : NP * 1024;
  static constexpr int LDS = OFF_P + P_BYTES;
  static constexpr int NITEM = IS * IS * NCHK;             // 16-byte pieces of one input halo
  static constexpr int NPLANE_IN = PAIR ? 4 : 2;
  static constexpr int OUT1 = PAIR ? 4 : 2, OUT2 = OUT1 + 1;   // planes the two convs append
  static_assert(ROWP >= IS * PITCH && (ROWP / 16) % 16 == 1, "image row pitch");
  static_assert(FROW >= FS * PS && (FROW / 16) % 16 == (5 * KAPPA) % 16, "FM row pitch");
  static_assert(((PITCH / 16) * KAPPA) % 16 == 1, "kappa");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

// compile-time loop: f(integral_constant<int, I>) for I = 0..N-1.  (#pragma unroll gave up on the 63-step body with
// barriers inside and left the tap / k-step decode to run-time branches.)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

struct Lane {
  int tid, lane, wave, half;
  int py, px;            // this lane's pixel of the wave's centre M-tile (tile coordinates)
  int rr, rc;            // ring pixel (region coordinates), waves 0..2
  bool rvalid;
};

// ---- one run of merged steps [G0, G0+NS): acc1c += W1 B, acc2 += W2 B, (RING) acc1r += W1 Br ----------
// wl: LDS address of the fragment pair of step G0 (+ lane*16); fragments alternate [W1 step][W2 step].
// post(integral_constant<step>) runs behind the MFMAs of every step: the kernels hang their global loads and LDS
// stores there ONE PER STEP - issued as a burst behind a barrier by all eight waves at once they queue on the
// CU's address path while the matrix pipes run dry (measured: 23 % of pair 1's time).
template <int PAIR, int G0, int NS, bool RING, class Post>
__device__ __forceinline__ void merged_steps(const unsigned char* __restrict__ wl, const unsigned char* __restrict__ bc,
                                             const unsigned char* __restrict__ br, f32x16& acc1c, f32x16& acc1r, f32x16& acc2,
                                             Post&& post) {
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
  static_for<0, NS>([&](auto si) __attribute__((always_inline)) {
    constexpr int st = decltype(si)::value;
    if constexpr (st + RD - 1 < NS) load_step(st + RD - 1);
    __builtin_amdgcn_sched_barrier(0);
    constexpr int s = st % RD;
    acc1c = mfma_32x32x16(rA1[s], rBc[s], acc1c);
    acc2 = mfma_32x32x16(rA2[s], rBc[s], acc2);
    if (RING) acc1r = mfma_32x32x16(rA1[s], rBr[s], acc1r);
    __builtin_amdgcn_sched_barrier(0);
    post(std::integral_constant<int, G0 + st>{});
  });
}

// ---- the 18 FM steps of the second conv --------------------------------------------------------------
template <int PAIR, class Post>
__device__ __forceinline__ void fm_steps(const unsigned char* __restrict__ wl, const unsigned char* __restrict__ fb, f32x16& acc2, Post&& post) {
  using G = Geo<PAIR>;
  f16x8 rA[3], rB[3];
  auto load_step = [&](const int st) __attribute__((always_inline)) {
    const int tap = st >> 1, ks = st & 1;
    rA[st % 3] = *reinterpret_cast<const f16x8*>(wl + st * 1024);
    rB[st % 3] = *reinterpret_cast<const f16x8*>(fb + (tap / 3) * G::FROW + (tap % 3) * PS + ks * 32);
  };
  load_step(0);
  load_step(1);
  static_for<0, 18>([&](auto si) __attribute__((always_inline)) {
    constexpr int st = decltype(si)::value;
    if constexpr (st + 2 < 18) load_step(st + 2);
    __builtin_amdgcn_sched_barrier(0);
    acc2 = mfma_32x32x16(rA[st % 3], rB[st % 3], acc2);
    __builtin_amdgcn_sched_barrier(0);
    post(si);
  });
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
    r[g][0] = lrelu_pack2(acc[4 * g + 0], acc[4 * g + 1]);
    r[g][1] = lrelu_pack2(acc[4 * g + 2], acc[4 * g + 3]);
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
#ifdef SELFC_CLOCKS
  ClockProbe ckp;
  clock_probe_begin(ckp);
#endif
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
  // Frame walk: workgroup b owns ONE spatial tile (b % ntiles) and visits frames f0, f0 + gf, ... - the halo geometry
  // (global offsets inside a frame, LDS offsets, which pixels fall outside the frame) is then a per-workgroup constant,
  // computed once, and a tile's 14 halo loads cost one instruction each (the per-tile address arithmetic - ~55 VALU
  // instructions per 16-byte piece - used to outweigh the MFMAs of the steps it was hung behind).
  // (Measured and dropped: xcd_swizzle(blockIdx.x) here, so that neighbouring tiles' halos meet in one XCD's L2 - fused F
  // -0.5 %, headline -1.5 %.)
  const int stile = blockIdx.x % a.ntiles, f0 = blockIdx.x / a.ntiles, gf = gridDim.x / a.ntiles;
  if (f0 >= a.N) return;
  const int ty0 = (stile / a.tiles_x) * TS, tx0 = (stile % a.tiles_x) * TS;
#ifdef SELFC_STAMPS
  unsigned long long phase[7] = {0, 0, 0, 0, 0, 0, 0};   // 0 tile setup, 1 merged MFMA, 2 epilogue 1, 3 mid barrier, 4 FM MFMA, 5 epilogue 2 + image store, 6 end barrier
  STAMP(tk0);
#endif
  const bool ring = c.wave < 3;
  const u32x4* __restrict__ wsrc = reinterpret_cast<const u32x4*>(a.w);

  // ---- input halo (20x20 pixels) -------------------------------------------------------------------------
  // map A (64-byte planes: x2 channels 0..31, f1, f2): piece i = tid + 512 it (it < 4, i < 1600) = 16-byte piece i & 3 of
  // halo pixel i >> 2;  map B (x2 channels 32..47): i = tid + 512 it (it < 2, i < 800) = piece i & 1 of pixel i >> 1.
  // goff: byte offset inside one frame of a plane; ok: bit it set
  // when the piece exists and its pixel lies inside the frame; ex: bit it set when the piece exists.  Pieces outside the
  // frame are the convs' zero padding: the workgroup's FIRST fill stores zeros there, later fills leave them alone.
  unsigned goffA[4], goffB[2], okA = 0, okB = 0, exA = 0, exB = 0;
  {
    auto geom = [&](const int pix, const bool exists, unsigned& goff) __attribute__((always_inline)) {
      const int p = min(pix, IS * IS - 1);
      const int hy = p / IS, hx = p - hy * IS;
      const int y = ty0 + hy - 2, x = tx0 + hx - 2;
      const bool ok = exists & (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
      const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
      goff = (unsigned)(yc * a.W + xc) * 64u;
      return ok;
    };
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int i = c.tid + it * NTHR;
      const bool ok = geom(i >> 2, i < 1600, goffA[it]);
      goffA[it] += (i & 3) * 16;
      okA |= (ok ? 1u : 0u) << it;
      exA |= (i < 1600 ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = c.tid + it * NTHR;
      const bool ok = geom(i >> 1, i < 800, goffB[it]);
      goffB[it] += (i & 1) * 16;
      okB |= (ok ? 1u : 0u) << it;
      exB |= (i < 800 ? 1u : 0u) << it;
    }
  }
  const size_t frame_bytes = (size_t)a.H * a.W * 64;
  const char* const dbase = reinterpret_cast<const char*>(a.dense);
  // part 0 = x2: items 0..3 map A on plane 0, items 4, 5 map B on plane 1; part 1 = f1 (plane 2), part 2 = f2 (plane 3)
  constexpr int XMAX = 6;
  constexpr unsigned FILL_BITS = 0u;
  u32x4 xv[XMAX];
  const char* lframe = dbase;        // frame the next x_load_item reads (plane 0)
  auto x_target = [&](const int n) __attribute__((always_inline)) { lframe = dbase + (size_t)n * frame_bytes; };
  auto x_load_item_to = [&](const int part, const int it, u32x4 (&v)[XMAX]) __attribute__((always_inline)) {
    const size_t pb = (size_t)a.plane * 2;      // bytes per plane
    if (part == 0 && it >= 4) v[it] = *reinterpret_cast<const u32x4*>(lframe + pb + goffB[it - 4]);
    else v[it] = *reinterpret_cast<const u32x4*>(lframe + (part == 0 ? 0 : part + 1) * pb + goffA[it]);
  };
  auto x_store_item_from = [&](const int part, const int it, const u32x4 (&v)[XMAX], const bool fill = false) __attribute__((always_inline)) {
    // the LDS offset is recomputed here (a handful of VALU instructions) rather than pinned in registers all kernel long
    int tidl = c.tid;
    asm volatile("" : "+v"(tidl));
    const bool mb = part == 0 && it >= 4;
    // 24-bit multiplies only (full rate; v_mul_lo / v_mul_hi / 64-bit mads are quarter rate and made these steps
    // VALU-bound): p / 20 = (p * 3277) >> 16 for p < 16384
    const unsigned i = (unsigned)tidl + (unsigned)((mb ? it - 4 : it) * NTHR);
    const unsigned p = mb ? i >> 1 : i >> 2, piece = mb ? i & 1u : i & 3u;
    const unsigned hy = __umul24(p, 3277u) >> 16;
    unsigned hx;      // p - 20 hy as ONE full-rate v_mad_i32_i24 (hipcc turns every C spelling of it into a quarter-rate 64-bit mad)
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(hx) : "v"(hy), "s"(-IS), "v"(p));
    const unsigned loff = __umul24(hy, (unsigned)G::ROWP) + __umul24(hx, (unsigned)G::PITCH) + piece * 16u +
                          (mb ? 64u : part == 0 ? 0u : part == 1 ? 96u : 160u);
    const bool ok = ((mb ? okB >> (it - 4) : okA >> it) & 1u) != 0;
    if (fill) {
      if ((mb ? exB >> (it - 4) : exA >> it) & 1u) *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + loff) = ok ? v[it] : u32x4{FILL_BITS, FILL_BITS, FILL_BITS, FILL_BITS};
    } else if (ok) *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + loff) = v[it];
  };
  auto x_load_item = [&](const int part, const int it) __attribute__((always_inline)) { x_load_item_to(part, it, xv); };
  auto x_store_item = [&](const int part, const int it) __attribute__((always_inline)) { x_store_item_from(part, it, xv); };
  auto x_load_to = [&](const int part, u32x4 (&v)[XMAX]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < XMAX; ++it)
      if (it < (part == 0 ? 6 : 4)) x_load_item_to(part, it, v);
  };
  // the workgroup's first fill (prologue): every existing piece is written, zeros where the frame ends
  auto x_store_from = [&](const int part, const u32x4 (&v)[XMAX]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < XMAX; ++it)
      if (it < (part == 0 ? 6 : 4)) x_store_item_from(part, it, v, true);
  };
#ifdef SELFC_POISON_LDS
  // test build: the image starts as f16 NaNs, so a read of a byte no fill wrote shows up in the parity tests
  for (int i = c.tid; i < G::IMG_BYTES / 16; i += NTHR) *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + i * 16) = u32x4{0x7e007e00u, 0x7e007e00u, 0x7e007e00u, 0x7e007e00u};
  __syncthreads();
#endif

  const unsigned char* const img = smem + G::OFF_IMG;
  const unsigned char* const bc = img + (c.py + 1) * G::ROWP + (c.px + 1) * G::PITCH + c.half * 16;
  const unsigned char* const br = img + c.rr * G::ROWP + c.rc * G::PITCH + c.half * 16;
  const unsigned char* const fb = smem + G::OFF_FM + c.py * G::FROW + c.px * PS + c.half * 16;
  const float* const lb = reinterpret_cast<const float*>(smem + G::OFF_B);

  // epilogue of the pair's first conv: FM image (zero outside the frame) + centre crop to HBM; vc = the centre
  // pixels' 32 features as two MFMA B fragments (k-step gp = channels 16 gp + 8 half ..) for the conv5 partial products
  auto epilogue1 = [&](const f32x16& acc1c, const f32x16& acc1r, const int n, u32x4 (&vc)[2]) __attribute__((always_inline)) {
    f16* __restrict__ dplane = a.dense + (size_t)G::OUT1 * a.plane;
    const int y = ty0 + c.py, x = tx0 + c.px;
    const bool in = (y < a.H) & (x < a.W);
    lrelu_pack(acc1c, in, vc);
    unsigned char* fdst = smem + G::OFF_FM + (c.py + 1) * G::FROW + (c.px + 1) * PS + 16 * c.half;
    *reinterpret_cast<u32x4*>(fdst) = vc[0];
    *reinterpret_cast<u32x4*>(fdst + 32) = vc[1];
    if (in & (a.store_feat != 0)) {
      f16* d = dplane + ((size_t)(n * a.H + y) * a.W + x) * 32 + 8 * c.half;
      *reinterpret_cast<u32x4*>(d) = vc[0];
      *reinterpret_cast<u32x4*>(d + 16) = vc[1];
    }
    if (ring) {
      u32x4 v[2];
      const int ry = ty0 + c.rr - 1, rx = tx0 + c.rc - 1;
      const bool rin = (ry >= 0) & (ry < a.H) & (rx >= 0) & (rx < a.W);
      lrelu_pack(acc1r, rin, v);
      if (c.rvalid) {
        unsigned char* rdst = smem + G::OFF_FM + c.rr * G::FROW + c.rc * PS + 16 * c.half;
        *reinterpret_cast<u32x4*>(rdst) = v[0];
        *reinterpret_cast<u32x4*>(rdst + 32) = v[1];
      }
    }
  };
  auto epilogue2 = [&](const f32x16& acc2, const int n, u32x4 (&v)[2]) __attribute__((always_inline)) {
    const int y = ty0 + c.py, x = tx0 + c.px;
    lrelu_pack(acc2, true, v);
    if ((y < a.H) & (x < a.W) & (a.store_feat != 0)) {
      f16* d = a.dense + (size_t)G::OUT2 * a.plane + ((size_t)(n * a.H + y) * a.W + x) * 32 + 8 * c.half;
      *reinterpret_cast<u32x4*>(d) = v[0];
      *reinterpret_cast<u32x4*>(d + 16) = v[1];
    }
  };
  // ---- conv5 partial products (F's temporal conv5 has 3 outputs x 3 taps = 9 rows): P[row][px] += W5[row][16 ch] d[16 ch][px]
  // on the features while they are in registers / LDS, instead of a later pass over all 176 dense channels.
  const bool do_p = a.w5p != nullptr && a.pf != nullptr;
  const unsigned char* const pfr = smem + G::OFF_P + (PAIR ? min(c.lane & 31, 11) * 32 + c.half * 16 : c.lane * 16);
  auto p_frag = [&](const int j) __attribute__((always_inline)) {
    return *reinterpret_cast<const f16x8*>(pfr + j * (PAIR ? 384 : 1024));
  };
  auto p_feat = [&](f32x16& accp, const int j0, const u32x4 (&v)[2]) __attribute__((always_inline)) {
    accp = mfma_32x32x16(p_frag(j0), __builtin_bit_cast(f16x8, v[0]), accp);
    accp = mfma_32x32x16(p_frag(j0 + 1), __builtin_bit_cast(f16x8, v[1]), accp);
  };
  auto p_store = [&](const f32x16& accp, const int n) __attribute__((always_inline)) {
    const int y = ty0 + c.py, x = tx0 + c.px;
    if ((y < a.H) & (x < a.W)) {
      // accumulator rows (e & 3) + 8 (e >> 2) + 4 half, fragment rows 4 tap + oc: half 0 owns taps 0 (e 0-3) and 2
      // (e 4-7), half 1 tap 1 (e 0-3); one float4 (3 outputs + a zero) per pixel and tap: pf[tap][N][H][W][4]
      const size_t pix = (size_t)(n * a.H + y) * a.W + x, tapsz = (size_t)a.N * a.H * a.W * 4;
      *reinterpret_cast<float4*>(a.pf + (c.half ? tapsz : 0) + pix * 4) = make_float4(accp[0], accp[1], accp[2], accp[3]);
      if (c.half == 0) *reinterpret_cast<float4*>(a.pf + 2 * tapsz + pix * 4) = make_float4(accp[4], accp[5], accp[6], accp[7]);
    }
  };
  if (do_p) {
    const u32x4* __restrict__ psrc = reinterpret_cast<const u32x4*>(a.w5p);
    if (PAIR == 0) {
      if (c.tid < G::NP * 64) *reinterpret_cast<u32x4*>(smem + G::OFF_P + c.tid * 16) = psrc[c.tid];
    } else if (c.tid < G::NP * 24) {       // item = (fragment j, row r of 12, half h): 16 bytes
      const int j = c.tid / 24, r = (c.tid % 24) >> 1, h = c.tid & 1;
      const u32x4 val = r < 11 ? psrc[j * 64 + h * 32 + r] : u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(smem + G::OFF_P + j * 384 + r * 32 + h * 16) = val;
    }
  }

  if (c.tid < 64) reinterpret_cast<float*>(smem + G::OFF_B)[c.tid] = (c.tid < 32 ? a.bias[0] : a.bias[1])[c.tid & 31];

  if constexpr (G::RES) {
    // =====================================================================================================
    // pair 0: the whole 72-fragment stream is resident.  The first tile starts on the first 18 fragments (9
    // merged steps); the other 54 are already in flight and are stored behind those steps (a serial 72-KiB
    // fill was 22 % of the kernel's time).
    // =====================================================================================================
    constexpr int RITER = ((G::NFRAG - CHF) * 64 + NTHR - 1) / NTHR;   // 7
    constexpr int HITER = (CHF * 64 + NTHR - 1) / NTHR;                // 3
    u32x4 wrest[RITER];
    {
      u32x4 whead[HITER];
      x_target(f0);
      x_load_to(0, xv);
#pragma unroll
      for (int it = 0; it < HITER; ++it) whead[it] = wsrc[min(c.tid + it * NTHR, CHF * 64 - 1)];
#pragma unroll
      for (int it = 0; it < RITER; ++it) wrest[it] = wsrc[CHF * 64 + min(c.tid + it * NTHR, (G::NFRAG - CHF) * 64 - 1)];
      x_store_from(0, xv);
#pragma unroll
      for (int it = 0; it < HITER; ++it) {
        const int i = c.tid + it * NTHR;
        if (i < CHF * 64) *reinterpret_cast<u32x4*>(smem + G::OFF_W + i * 16) = whead[it];
      }
    }
    __syncthreads();
    bool first = true;
    const unsigned char* const wl = smem + G::OFF_W + c.lane * 16;

    for (int n = f0; n < a.N; n += gf) {
      const bool more = n + gf < a.N;
      STAMP(ts0);
      if (more) x_target(n + gf);
      f32x16 acc1c = bias_init(lb, c.half), acc1r = acc1c, acc2 = bias_init(lb + 32, c.half);
      STAMP(ts1);
      STAMP_ADD(0, ts0, ts1);
      // the next tile's halo: one load behind each of the first six merged steps, one store behind each of the first
      // six FM steps (the image is dead once every wave has passed the barrier in front of the FM phase)
      auto post_m = [&](auto gi) __attribute__((always_inline)) {
        constexpr int g = decltype(gi)::value;
        if constexpr (g < 6) { if (more) x_load_item(0, g); }
      };
      auto post_f = [&](auto si) __attribute__((always_inline)) {
        constexpr int st = decltype(si)::value;
        if constexpr (st < 6) { if (more) x_store_item(0, st); }
      };
      if (first) {
        if (ring) merged_steps<PAIR, 0, 9, true>(wl, bc, br, acc1c, acc1r, acc2, post_m);
        else merged_steps<PAIR, 0, 9, false>(wl, bc, br, acc1c, acc1r, acc2, post_m);
#pragma unroll
        for (int it = 0; it < RITER; ++it) {
          const int i = c.tid + it * NTHR;
          if (i < (G::NFRAG - CHF) * 64) *reinterpret_cast<u32x4*>(smem + G::OFF_W + CHF * 1024 + i * 16) = wrest[it];
        }
        __syncthreads();
        first = false;
        if (ring) merged_steps<PAIR, 9, G::S1 - 9, true>(wl + CHF * 1024, bc, br, acc1c, acc1r, acc2, post_m);
        else merged_steps<PAIR, 9, G::S1 - 9, false>(wl + CHF * 1024, bc, br, acc1c, acc1r, acc2, post_m);
      } else {
        if (ring) merged_steps<PAIR, 0, G::S1, true>(wl, bc, br, acc1c, acc1r, acc2, post_m);
        else merged_steps<PAIR, 0, G::S1, false>(wl, bc, br, acc1c, acc1r, acc2, post_m);
      }
      STAMP(ts2);
      STAMP_ADD(1, ts1, ts2);
      u32x4 vc[2];
      epilogue1(acc1c, acc1r, n, vc);
      f32x16& accp = acc1c;     // the first conv's accumulator is dead: its registers take the conv5 partial products
      if (do_p) {        // x2 (centre tap of the image, still intact before the barrier) and f1
#pragma unroll
        for (int e = 0; e < 16; ++e) accp[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
          accp = mfma_32x32x16(p_frag(ks), *reinterpret_cast<const f16x8*>(bc + G::ROWP + G::PITCH + ks * 32), accp);
        p_feat(accp, 3, vc);
      }
      STAMP(ts3);
      STAMP_ADD(2, ts2, ts3);
      __syncthreads();                                // FM complete; every wave is done with the input image
      STAMP(ts4);
      STAMP_ADD(3, ts3, ts4);
      fm_steps<PAIR>(wl + 2 * G::S1 * 1024, fb, acc2, post_f);
      STAMP(ts5);
      STAMP_ADD(4, ts4, ts5);
      epilogue2(acc2, n, vc);
      if (do_p) {
        p_feat(accp, 5, vc);                          // f2
        p_store(accp, n);
      }
      STAMP(ts6);
      STAMP_ADD(5, ts5, ts6);
      __syncthreads();                                // image of the next tile visible; FM free again
      STAMP(ts7);
      STAMP_ADD(6, ts6, ts7);
    }
  } else {
    // =====================================================================================================
    // pair 1: 144 fragments = 12 chunks of 12 through THREE LDS buffers.  The only workgroup barrier of a chunk
    // sits in its MIDDLE (after 3 of its 6 merged steps): behind barrier c every wave has left chunk c-1, so
    // chunk c+2 is committed into that buffer there, and it is visible to everybody by barrier c+1 - before any
    // wave starts chunk c+2.  The operand ring therefore runs across chunk boundaries and across the barrier
    // (its fragments are already in registers), instead of draining and refilling once per chunk.  Those same
    // barriers are where a source group's part of the image dies (x2 after step 26 = barrier 4, f1 after step 44
    // = barrier 7, f2 after step 62 = barrier 10), so the next tile's parts are stored there.
    // =====================================================================================================
    constexpr int CH3 = 12, CHI = CH3 * 64;          // fragments / 16-byte items per chunk
    static_assert(G::NFRAG % CH3 == 0 && (G::NFRAG / CH3) % 3 == 0 && 3 * CH3 * 1024 <= G::W_BYTES, "three-buffer chunking");
    u32x4 wset[2][2];
    auto w_prefetch_item = [&](const int chunk, const int set, const int it) __attribute__((always_inline)) {
      wset[set][it] = wsrc[chunk * CHI + min(c.tid + it * NTHR, CHI - 1)];
    };
    auto w_commit_item = [&](const int buf, const int set, const int it) __attribute__((always_inline)) {
      const int i = c.tid + it * NTHR;
      if (i < CHI) *reinterpret_cast<u32x4*>(smem + G::OFF_W + buf * CH3 * 1024 + i * 16) = wset[set][it];
    };
    {
      u32x4 xb[XMAX], xc[XMAX];
      x_target(f0);
      x_load_to(0, xv);
      w_prefetch_item(0, 0, 0); w_prefetch_item(0, 0, 1);
      w_prefetch_item(1, 1, 0); w_prefetch_item(1, 1, 1);
      x_load_to(1, xb);
      x_load_to(2, xc);
      x_store_from(0, xv);
      w_commit_item(0, 0, 0); w_commit_item(0, 0, 1);
      w_commit_item(1, 1, 0); w_commit_item(1, 1, 1);
      w_prefetch_item(2, 0, 0); w_prefetch_item(2, 0, 1);
      w_prefetch_item(3, 1, 0); w_prefetch_item(3, 1, 1);
      x_store_from(1, xb);
      x_store_from(2, xc);
    }
    __syncthreads();
    const unsigned char* const wb = smem + G::OFF_W + c.lane * 16;
    constexpr int NCHUNK = G::NFRAG / CH3;   // 12

    for (int n = f0; n < a.N; n += gf) {
      const bool more = n + gf < a.N;
      STAMP(ts0);
      if (more) x_target(n + gf);
      f32x16 acc1c = bias_init(lb, c.half), acc1r = acc1c, acc2 = bias_init(lb + 32, c.half);
      STAMP(ts1);
      STAMP_ADD(0, ts0, ts1);

      u32x4 vc[2];
      f32x16& accp = acc1c;     // dead after epilogue 1: its registers take the conv5 partial products
      auto wfrag = [&](const int f) __attribute__((always_inline)) { return wb + ((f / CH3) % 3) * CH3 * 1024 + (f % CH3) * 1024; };
      // Behind barrier cidx (merged step 6 cidx + 2): steps +1, +2 commit chunk cidx+2, steps +3, +4 prefetch chunk cidx+4.
      auto chunk_task = [&](const int cidx, const int j) __attribute__((always_inline)) {
        if (j == 1 || j == 2) w_commit_item((cidx + 2) % 3, cidx & 1, j - 1);
        if (j == 3 || j == 4) w_prefetch_item((cidx + 4) % NCHUNK, cidx & 1, j - 3);
      };
      auto post_m = [&](auto gi) __attribute__((always_inline)) {
        constexpr int g = decltype(gi)::value;
        if constexpr (g % 6 == 2) {
          if constexpr (g / 6 == 10) {
            epilogue1(acc1c, acc1r, n, vc);
            if (do_p) {                                          // f3 (acc1c is dead from here on)
#pragma unroll
              for (int e = 0; e < 16; ++e) accp[e] = 0.f;
              p_feat(accp, 0, vc);
            }
          }
          __syncthreads();
        } else if constexpr (g >= 3) {
          chunk_task((g - 3) / 6, g - (6 * ((g - 3) / 6) + 2));
        }
        if (more) {
          if constexpr (g < 6) x_load_item(0, g);
          if constexpr (g >= 27 && g < 33) {                  // x2 died at barrier 4 (step 26)
            x_store_item(0, g - 27);
            if constexpr (g - 27 < 4) x_load_item(1, g - 27);
          }
          if constexpr (g >= 45 && g < 49) {                  // f1 died at barrier 7 (step 44)
            x_store_item(1, g - 45);
            x_load_item(2, g - 45);
          }
        }
      };
      auto post_f = [&](auto si) __attribute__((always_inline)) {
        constexpr int st = decltype(si)::value;
        if constexpr (st < 4) chunk_task(10, st + 1);            // behind barrier 10 (merged step 62)
        if constexpr (st == 11) __syncthreads();                 // barrier 11
        if constexpr (st >= 12 && st < 16) chunk_task(11, st - 11);
        if constexpr (st < 4) { if (more) x_store_item(2, st); } // f2 died at barrier 10
      };

      auto body = [&](auto ring_tag) __attribute__((always_inline)) {
        constexpr bool RING = decltype(ring_tag)::value;
        f16x8 rA1[3], rA2[3], rBc[3], rBr[3];
        auto load_m = [&](const int g) __attribute__((always_inline)) {
          const int tap = g < 27 ? g / 3 : g < 45 ? (g - 27) >> 1 : (g - 45) >> 1;
          const int ks = g < 27 ? g % 3 : g < 45 ? 3 + ((g - 27) & 1) : 5 + ((g - 45) & 1);
          const int off = (tap / 3) * G::ROWP + (tap % 3) * G::PITCH + ks * 32;
          const int s = g % 3;
          rA1[s] = *reinterpret_cast<const f16x8*>(wfrag(2 * g));
          rA2[s] = *reinterpret_cast<const f16x8*>(wfrag(2 * g + 1));
          rBc[s] = *reinterpret_cast<const f16x8*>(bc + off);
          if (RING) rBr[s] = *reinterpret_cast<const f16x8*>(br + off);
        };
        load_m(0);
        load_m(1);
        static_for<0, G::S1>([&](auto gi) __attribute__((always_inline)) {
          constexpr int g = decltype(gi)::value;
          if constexpr (g + 2 < G::S1) load_m(g + 2);
          __builtin_amdgcn_sched_barrier(0);
          constexpr int s = g % 3;
#if defined(SELFC_STAMPS) && defined(SELFC_STEP_STAMPS)     // diagnostic: the step split into its MFMA group (bucket 2) and its hook (bucket 3)
          STAMP(tsa);
#endif
          acc1c = mfma_32x32x16(rA1[s], rBc[s], acc1c);
          acc2 = mfma_32x32x16(rA2[s], rBc[s], acc2);
          if (RING) acc1r = mfma_32x32x16(rA1[s], rBr[s], acc1r);
          __builtin_amdgcn_sched_barrier(0);
#if defined(SELFC_STAMPS) && defined(SELFC_STEP_STAMPS)
          STAMP(tsb);
          STAMP_ADD(2, tsa, tsb);
#endif
          post_m(gi);
#if defined(SELFC_STAMPS) && defined(SELFC_STEP_STAMPS)
          STAMP(tsc);
          STAMP_ADD(3, tsb, tsc);
#endif
        });
      };
      if (ring) body(std::true_type{});
      else body(std::false_type{});
      STAMP(ts2);
      STAMP_ADD(1, ts1, ts2);

      // FM phase: fragments 126..143 = second half of chunk 10 and chunk 11 (barrier 11 behind FM step 11)
      {
        f16x8 rA[3], rB[3];
        auto load_f = [&](const int st) __attribute__((always_inline)) {
          const int tap = st >> 1, ks = st & 1;
          rA[st % 3] = *reinterpret_cast<const f16x8*>(wfrag(2 * G::S1 + st));
          rB[st % 3] = *reinterpret_cast<const f16x8*>(fb + (tap / 3) * G::FROW + (tap % 3) * PS + ks * 32);
        };
        load_f(0);
        load_f(1);
        static_for<0, 18>([&](auto si) __attribute__((always_inline)) {
          constexpr int st = decltype(si)::value;
          if constexpr (st + 2 < 18) load_f(st + 2);
          __builtin_amdgcn_sched_barrier(0);
          acc2 = mfma_32x32x16(rA[st % 3], rB[st % 3], acc2);
          __builtin_amdgcn_sched_barrier(0);
          post_f(si);
        });
      }
      STAMP(ts5);
      STAMP_ADD(4, ts2, ts5);
      epilogue2(acc2, n, vc);
      if (do_p) {
        p_feat(accp, 2, vc);                          // f4
        p_store(accp, n);
      }
      STAMP(ts6);
      STAMP_ADD(5, ts5, ts6);
    }
  }
#ifdef SELFC_CLOCKS
  clock_probe_end(ckp, a.stamps, blockIdx.x == 0 && c.tid == 0);
#endif
#ifdef SELFC_STAMPS
  STAMP(tk1);
  if (a.stamps && c.lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * NWAVE + c.wave) * 8;
#pragma unroll
    for (int i = 0; i < 7; ++i) o[i] = phase[i];
    o[7] = tk1 - tk0;
  }
#endif
}

// y1 = x1 +- (b5 + conv5 of F) from the partial products of the two pairs: out[t] = sum over pairs of
// P[t-1][tap 0] + P[t][tap 1] + P[t+1][tap 2], zero outside the clip (Subnet_constructor.py:130, Inv_arch.py:25,31).
__global__ __launch_bounds__(256) void f_couple_kernel(const float* __restrict__ pf, const float* __restrict__ bias, const float* x1, float* x1out,
                                                       const int N, const int T, const int HW, const float sgn, const int nsets) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, npx = (size_t)N * HW;
  if (i >= npx) return;
  const int t = (int)(i / HW) % T;
  float4 o = make_float4(bias[0], bias[1], bias[2], 0.f);
  for (int pair = 0; pair < nsets; ++pair) {      // pf[set][tap][N][H][W][4]: two sets from the pair kernels, one from csrc/split_f.hip
    const float4* p = reinterpret_cast<const float4*>(pf) + (size_t)pair * 3 * npx + i;
    const float4 q1 = p[npx];
    o.x += q1.x; o.y += q1.y; o.z += q1.z;
    if (t > 0) {
      const float4 q = p[-(ptrdiff_t)HW];
      o.x += q.x; o.y += q.y; o.z += q.z;
    }
    if (t + 1 < T) {
      const float4 q = p[2 * npx + HW];
      o.x += q.x; o.y += q.y; o.z += q.z;
    }
  }
  float4 v = *reinterpret_cast<const float4*>(x1 + i * 4);
  v.x += sgn * o.x; v.y += sgn * o.y; v.z += sgn * o.z;
  *reinterpret_cast<float4*>(x1out + i * 4) = v;
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

template <int PAIR>
int launch_pair(FFArgs& a, int maxwg, hipStream_t s) {
  using G = Geo<PAIR>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&fused_f_kernel<PAIR>), G::LDS, optin); e != hipSuccess) return hip_rc(e);
  // Frame walk (see the kernel): gf workgroups per spatial tile, workgroup b visits frames b / ntiles + k gf.  gf is
  // sized so that about maxwg workgroups exist, every one of them walks (nearly) the same number of frames, and -
  // when there are frames enough - at least three, which amortises the prologue (weights, first halo).
  const int gmax = a.ntiles >= maxwg ? 1 : (maxwg / a.ntiles < a.N ? maxwg / a.ntiles : a.N);
  static const int minrounds = getenv("SELFC_FUSEDF_MINROUNDS") ? atoi(getenv("SELFC_FUSEDF_MINROUNDS")) : 3;
  int rounds = (a.N + gmax - 1) / gmax;
  if (rounds < minrounds) rounds = a.N < minrounds ? a.N : minrounds;
  const int gfr = (a.N + rounds - 1) / rounds;
  const int gx = gfr * a.ntiles;
#ifdef SELFC_STAMPS
  static unsigned long long* dbg = nullptr;
  constexpr int DBG_WG = 1024;
  if (!dbg) { (void)hipMalloc(&dbg, DBG_WG * NWAVE * 8 * sizeof(unsigned long long)); (void)hipMemset(dbg, 0, DBG_WG * NWAVE * 8 * sizeof(unsigned long long)); }
  a.stamps = gx <= DBG_WG ? dbg : nullptr;
  if (getenv("SELFC_STAMP_DUMP_F")) {      // diagnostic: dump the previous launch's sums (of this pair), then continue
    static unsigned long long host[DBG_WG * NWAVE * 8];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
    char fn[512];
    snprintf(fn, sizeof(fn), "%s.%d", getenv("SELFC_STAMP_DUMP_F"), PAIR);
    FILE* f = fopen(fn, "w");
    if (f) {
      for (int i = 0; i < DBG_WG * NWAVE; ++i) {
        for (int j = 0; j < 8; ++j) fprintf(f, "%llu ", host[8 * i + j]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
  }
#endif
#ifdef SELFC_CLOCKS
  a.stamps = clock_probe_slot(PAIR);
#endif
  hipLaunchKernelGGL(fused_f_kernel<PAIR>, dim3((unsigned)gx), dim3(NTHR), G::LDS, s, a);
  return hip_rc(hipGetLastError());
}

}  // namespace

// csrc/fused_f16.hip: the same two pair launches on the 16x16x32 kernels
int launch_fused_f16_pairs(void* dense, const void* w16, const float* const* bias, int N, int H, int W, hipStream_t s,
                           const void* w5p16, float* pf, int keep_features);

// conv1..conv4 of F (cin = 48) on its dense buffer: two launches.  w = [pair 0: 72 fragments][pair 1: 144 fragments]
// [the 16x16x32 stream: 74 + 146 fragments], w5p = [11 fragments][6 fragments of the 16x16x32 kernels] (packing.subnet_pack_entries).
// The 16x16x32 kernels (csrc/fused_f16.hip) are the default; SELFC_F_MFMA32=1 selects the 32x32x16 kernels of this file.
// With w5p (11 partial-product fragments), pf and x1 the temporal conv5 + coupling y1 = x1 +- F is done here as well:
// the two launches emit the conv5 partial products and f_couple_kernel sums them (returns 1: conv5 handled).
int launch_fused_f(void* dense, const void* w, const float* const* bias, int N, int H, int W, hipStream_t s,
                   const void* w5p, float* pf, const float* b5, const float* x1, float* x1out, int T, int rev, int keep_features) {
  static const int maxwg = getenv("SELFC_FUSEDF_MAXWG") ? atoi(getenv("SELFC_FUSEDF_MAXWG")) : 256;
  static const bool no_p = getenv("SELFC_NO_F5P") != nullptr;     // developer A/B switch
  const bool with_p = w5p && pf && b5 && x1 && T > 0 && !no_p;
  FFArgs a{};
  a.dense = (f16*)dense;
  a.N = N; a.H = H; a.W = W;
  a.tiles_x = (W + TS - 1) / TS;
  a.tiles_y = (H + TS - 1) / TS;
  a.ntiles = a.tiles_x * a.tiles_y;
  a.plane = (size_t)N * H * W * 32;
  static const bool mfma32 = getenv("SELFC_F_MFMA32") != nullptr;
  bool use32 = mfma32;
  if (!use32) {
    ProfScope prof(PROF_CONV3X3, s);
    const int rc = launch_fused_f16_pairs(dense, (const f16*)w + (size_t)(Geo<0>::NFRAG + Geo<1>::NFRAG) * 512, bias, N, H, W, s,
                                          with_p ? (const f16*)w5p + (size_t)(Geo<0>::NP + Geo<1>::NP) * 512 : nullptr, with_p ? pf : nullptr, keep_features);
    if (rc == 2) use32 = true;         // SELFC_F16_TOO_BIG: beyond the 16x16x32 kernels' 2-GiB buffer addressing, nothing launched
    else if (rc || !with_p) return rc;
  }
  if (use32) {
    ProfScope prof(PROF_CONV3X3, s);
    a.w = (const f16*)w;
    a.bias[0] = bias[0]; a.bias[1] = bias[1];
    a.w5p = with_p ? (const f16*)w5p : nullptr;
    a.pf = with_p ? pf : nullptr;
    a.store_feat = 1;                                  // f1, f2: pair 1 reads them
    int rc = launch_pair<0>(a, maxwg > 0 ? maxwg : 256, s);
    if (rc) return rc;
    a.w = (const f16*)w + (size_t)Geo<0>::NFRAG * 512;
    a.bias[0] = bias[2]; a.bias[1] = bias[3];
    a.w5p = with_p ? (const f16*)w5p + (size_t)Geo<0>::NP * 512 : nullptr;
    a.pf = with_p ? pf + (size_t)N * H * W * 12 : nullptr;
    // f3, f4 feed only conv5: with the partial products taken here nothing reads them again unless the caller keeps the
    // dense buffer for a backward pass (selfc_latent.flags & SELFC_LAT_KEEP_FEATURES) - 128 B per pixel-frame of dead stores
    static const bool force_keep = getenv("SELFC_F_KEEP_FEATURES") != nullptr;     // developer A/B switch
    a.store_feat = (keep_features || !with_p || force_keep) ? 1 : 0;
    rc = launch_pair<1>(a, maxwg > 0 ? maxwg : 256, s);
    if (rc || !with_p) return rc;
  }
  ProfScope prof(PROF_CONV5_F, s);
  const size_t npx = (size_t)N * H * W;
  hipLaunchKernelGGL(f_couple_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, s, pf, b5, x1, x1out, N, T, H * W, rev ? -1.f : 1.f, 2);
  const int rc = hip_rc(hipGetLastError());
  return rc ? rc : 1;
}

}  // namespace selfc
