// Pairwise-fused conv1..conv4 of the F dense block (D2DTInput / DenseBlock with cin = 48, Subnet_constructor.py:27-30,126-129)
// for gfx950, designed around v_mfma_f32_16x16x32_f16.
//
// Same decomposition as csrc/fused_f.hip (pair 0: conv1 on the 18x18 region around a 16x16 tile -> FM image in LDS, conv2 on
// the tile; pair 1: conv3 / conv4 with [x2 | f1 | f2] as input; one persistent 512-thread workgroup per CU owns ONE spatial
// tile and walks frames; the pair's first conv and the input part of its second conv consume the SAME activation fragment;
// F's temporal conv5 leaves as partial products), but on the 16-cycle MFMA shape.  Why: this chip is POWER-bound under the
// stack's workload - the shader clock it holds is what the instruction mix lets it hold, and
// 16x16x32 delivers the same MACs at ~12 % less energy than 32x32x16 (same operand bytes from LDS, a quarter of the
// accumulator registers per instruction): a timing-only proxy of the old kernels on this shape measured +3.5 % on the headline
// for F alone.  What the shape changes:
//
//   * a k-step is 32 deep = two 16-channel groups of the old K order; x2's 27 groups per pixel pair up as nine whole-tap
//     steps (channels 0..31) + five steps that pair the taps' last 16 channels (packing.f16_steps) - one group of padding;
//   * an M-tile is ONE row of 16 pixels; lane (q = lane >> 4, i = lane & 15) reads the 16 bytes of k-octet q of pixel pos(i).
//     ds_read_b128 is served in lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ..: eight lanes of octet q and eight of
//     octet q + 1.  With a pixel pitch of 6 or 14 sixteen-byte slots (NO padding: 96 / 224 B) the slot residues of 16
//     consecutive pixels are the eight even residues, each twice (pixels p and p + 8); pos() puts p < 8 on the lanes
//     {0-3,12-15} and p >= 8 on {4-11}, so a group touches the evens (octet q) and the odds (octet q + 1): conflict-free
//     without padding bytes.  Row pitches = 10 slots mod 16 give the ring's COLUMN blocks the same property;
//   * output rows are permuted at packing time (row 4 q + e of block rb = channel 8 q + 4 rb + e), so a lane's accumulators
//     are 8 consecutive channels of one pixel: bias, LeakyReLU, one pack, ONE 16-byte store - no permlane, and the packed
//     registers ARE the B fragment of the conv5 partial products (one 16-cycle MFMA per feature and pixel row).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "prof.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace selfc {

constexpr int SELFC_F16_TOO_BIG = 2;      // launch_fused_f16_pairs: not launched, the caller falls back (same value in csrc/fused_f.hip)

struct F16Args {
  f16* dense;               // F dense buffer, plane-blocked [6][N][H][W][32]: x2 (2 planes), f1..f4
  const f16* w;             // fragment stream of this pair (packing.pack_fused_f16)
  const float* bias[2];     // 32 floats: first / second conv of the pair
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
  size_t plane;
  const f16* w5p;           // optional: conv5 partial-product fragments of this pair (packing.pack_f5_partial16)
  float* pf;                // optional: partial products of this pair, fp32 [3 taps][N][H][W][4]
  int store_feat;           // 1: the pair's two feature planes go to HBM; 0: nothing reads them afterwards
#ifdef SELFC_STAMPS
  unsigned long long* stamps;   // diagnostic build: per wave 7 phase sums + lifetime
#endif
};

namespace {

constexpr int TS = 16, IS = 20, FS = 18;
constexpr int NWAVE = 8, NTHR = NWAVE * 64;
constexpr int FPIX = 96;                          // FM pixel pitch: 64 B of features + 32 B (6 slots: see the header)
constexpr int FROW = FS * FPIX;                   // 1728
constexpr int FM_BYTES = FS * FROW;               // 31104

#ifdef SELFC_STAMPS
#define STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define STAMP_ADD(k, a, b) phase[k] += (b) - (a)
#else
#define STAMP(var)
#define STAMP_ADD(k, a, b)
#endif

template <int PAIR>
struct Geo {
  static constexpr int NOCT = PAIR ? 14 : 6;               // 16-byte octets per input pixel
  static constexpr int PITCH = NOCT * 16;                  // 96 / 224 B, no padding
  static constexpr int ROWP = PAIR ? 4512 : 1952;          // >= IS * PITCH, /16 = 10 (mod 16)
  static constexpr int IMG_BYTES = IS * ROWP;
  static constexpr int NX = 14;                            // x2 steps (9 whole-tap + 5 paired)
  static constexpr int NM = NX + 18 * PAIR;                // merged steps: 14 / 32
  static constexpr int NFRAG = 4 * NM + 18;                // 74 / 146
  static constexpr bool RES = PAIR == 0;                   // whole stream resident in LDS
  static constexpr int CH = 12;                            // fragments per streamed chunk buffer
  static constexpr int W_BYTES = RES ? NFRAG * 1024 : 3 * CH * 1024;
  static constexpr int NP = PAIR ? 2 : 4;                  // conv5 partial-product fragments
  static constexpr int OFF_IMG = 0, OFF_FM = IMG_BYTES, OFF_W = OFF_FM + FM_BYTES, OFF_B = OFF_W + W_BYTES;
  static constexpr int OFF_P = OFF_B + 256;
  static constexpr int LDS = OFF_P + NP * 1024;
  static constexpr int NPLANE_IN = PAIR ? 4 : 2;
  static constexpr int OUT1 = PAIR ? 4 : 2, OUT2 = OUT1 + 1;   // planes the two convs append
  static_assert(ROWP >= IS * PITCH && (ROWP / 16) % 16 == 10, "image row pitch");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// byte offset of merged step s inside the image, relative to the step's lane pointer (see step_kind)
template <int PAIR>
__host__ __device__ constexpr int step_off(const int s) {
  using G = Geo<PAIR>;
  if (s < 9) return (s / 3) * G::ROWP + (s % 3) * G::PITCH;                  // A: tap s, channels 0..31
  if (s < 12) return (s - 9) * G::ROWP;                                      // P: taps 3r, 3r+1 (r = s - 9), channels 32..47
  if (s == 12) return 2 * G::PITCH;                                          // R: taps 2, 5
  if (s == 13) return 2 * G::ROWP + 2 * G::PITCH;                            // S: tap 8
  const int t = (s - 14) % 9, f = (s - 14) / 9;                              // f1 / f2: tap t, 32 channels
  return (t / 3) * G::ROWP + (t % 3) * G::PITCH + 96 + 64 * f;
}
// which lane pointer a merged step uses: 0 = A (base + 16 q), 1 = P (+ (q>>1) PITCH + 64 + 16 (q&1)), 2 = R (+ (q>>1) ROWP + ..), 3 = S
__host__ __device__ constexpr int step_kind(const int s) { return s < 9 ? 0 : s < 12 ? 1 : s == 12 ? 2 : s == 13 ? 3 : 0; }

struct BPtr {                  // the four per-lane base pointers of one 16-pixel block
  const unsigned char* k[4];
};
template <int PAIR>
__device__ __forceinline__ BPtr make_bptr(const unsigned char* base, const int q) {
  using G = Geo<PAIR>;
  BPtr p;
  p.k[0] = base + 16 * q;
  p.k[1] = base + (q >> 1) * G::PITCH + 64 + 16 * (q & 1);
  p.k[2] = base + (q >> 1) * G::ROWP + 64 + 16 * (q & 1);
  p.k[3] = base + 64 + 16 * (q & 1);
  return p;
}

// LeakyReLU + f16 pack of a lane's 8 consecutive channels (blocks rb = 0, 1 of one pixel); masked pixels give zeros
__device__ __forceinline__ u32x4 lrelu_pack8(const f32x4& a0, const f32x4& a1, const bool keep) {
  u32x4 v;
  v[0] = lrelu_pack2(a0[0], a0[1]);
  v[1] = lrelu_pack2(a0[2], a0[3]);
  v[2] = lrelu_pack2(a1[0], a1[1]);
  v[3] = lrelu_pack2(a1[2], a1[3]);
  const uint32_t m = keep ? 0xffffffffu : 0u;
  v[0] &= m; v[1] &= m; v[2] &= m; v[3] &= m;
  return v;
}

template <int PAIR>
__global__ __launch_bounds__(NTHR) void fused_f16_kernel(const F16Args a) {
  using G = Geo<PAIR>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, i = lane & 15;
  // position of this lane's pixel inside a 16-pixel block (header: lanes {0-3,12-15} take positions 0..7, {4-11} take 8..15)
  const int pos = i < 4 ? i : (i < 12 ? i + 4 : i - 8);
  const bool ring = wave < 5;                    // waves 0..4 own one of the five ring blocks (68 ring pixels)

  // workgroups go to the XCDs round robin: the swizzle gives each XCD a contiguous run of (frame group, tile) pairs, so the tiles
  // that share halo pixels of one frame read them through the same L2
  const int lbid = xcd_swizzle((int)blockIdx.x, (int)gridDim.x);          // (-23 % FETCH_SIZE on the headline's launches)
  const int stile = lbid % a.ntiles, f0 = lbid / a.ntiles, gf = gridDim.x / a.ntiles;
  if (f0 >= a.N) return;
  const int ty0 = (stile / a.tiles_x) * TS, tx0 = (stile % a.tiles_x) * TS;
#ifdef SELFC_STAMPS
  unsigned long long phase[7] = {0, 0, 0, 0, 0, 0, 0};   // 0 tile setup, 1 merged steps, 2 epilogue 1 (+ P), 3 mid barrier, 4 FM steps, 5 epilogue 2 (+ P store), 6 end barrier
  STAMP(tk0);
#endif

  // ---- ring pixel of this lane (region coordinates 0..17): block 0 top row, 1 bottom row, 2 left column, 3 right column,
  // 4 the four corner-side leftovers (0,16) (0,17) (17,16) (17,17) - its other lanes re-read pixel (0,16) and store nothing
  int rr, rc;
  bool rvalid = true;
  {
    const int b = wave;
    rr = b == 0 ? 0 : b == 1 ? 17 : (b == 4 ? (pos & 2 ? 17 : 0) : pos + 1);
    rc = b == 2 ? 0 : b == 3 ? 17 : (b == 4 ? 16 + (pos & 1) : pos);
    if (b == 4 && pos >= 4) { rr = 0; rc = 16; rvalid = false; }
    if (b > 4) { rr = 0; rc = 0; rvalid = false; }
  }

  // ---- input halo (20x20 pixels): the same piece maps as csrc/fused_f.hip
  // map A (64-byte planes: x2 channels 0..31, f1, f2): piece j = tid + 512 it (it < 4, j < 1600) = 16-byte piece j & 3 of halo
  // pixel j >> 2;  map B (x2 channels 32..47): j = tid + 512 it (it < 2, j < 800) = piece j & 1 of pixel j >> 1.
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
      const int j = tid + it * NTHR;
      const bool ok = geom(j >> 2, j < 1600, goffA[it]);
      goffA[it] += (j & 3) * 16;
      okA |= (ok ? 1u : 0u) << it;
      exA |= (j < 1600 ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int j = tid + it * NTHR;
      const bool ok = geom(j >> 1, j < 800, goffB[it]);
      goffB[it] += (j & 1) * 16;
      okB |= (ok ? 1u : 0u) << it;
      exB |= (j < 800 ? 1u : 0u) << it;
    }
  }
  const size_t frame_bytes = (size_t)a.H * a.W * 64, plane_bytes = (size_t)a.plane * 2;
  const buf_rsrc drs = make_rsrc(a.dense, (unsigned)(plane_bytes * (G::NPLANE_IN + 2)));   // < 2 GiB: launch_fused_f16_pairs checks
  constexpr int XMAX = 6;
  u32x4 xv[XMAX];
  size_t lframe = 0;                 // byte offset of the frame the next x_load_item reads (plane 0)
  // part 0 = x2: items 0..3 map A on plane 0, items 4, 5 map B on plane 1; part 1 = f1 (plane 2), part 2 = f2 (plane 3)
  auto x_load_item_to = [&](const int part, const int it, u32x4 (&v)[XMAX]) __attribute__((always_inline)) {
    const bool mb = part == 0 && it >= 4;
    const int pl = mb ? 1 : part == 0 ? 0 : part + 1;
    const unsigned go = mb ? goffB[it - 4] : goffA[it];
    v[it] = buffer_load_b128(drs, go, (unsigned)(lframe + pl * plane_bytes));
  };
  // fill = the workgroup's first fill: every existing piece is written, zeros where the frame ends (the convs' zero padding);
  // later fills leave the out-of-frame pieces alone
  auto x_store_item_from = [&](const int part, const int it, const u32x4 (&v)[XMAX], const bool fill) __attribute__((always_inline)) {
    const bool mb = part == 0 && it >= 4;
    // the LDS offset is recomputed (a handful of full-rate VALU instructions) rather than pinned in registers all kernel long:
    // p / 20 = (p * 3277) >> 16 for p < 16384, 24-bit multiplies only
    int tidl = tid;
    asm volatile("" : "+v"(tidl));
    const unsigned j = (unsigned)tidl + (unsigned)((mb ? it - 4 : it) * NTHR);
    const unsigned p = mb ? j >> 1 : j >> 2, piece = mb ? j & 1u : j & 3u;
    const unsigned hy = __umul24(p, 3277u) >> 16;
    unsigned hx;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(hx) : "v"(hy), "s"(-IS), "v"(p));
    const unsigned lo = __umul24(hy, (unsigned)G::ROWP) + __umul24(hx, (unsigned)G::PITCH) + piece * 16u +
                        (mb ? 64u : part == 0 ? 0u : part == 1 ? 96u : 160u);
    const bool ok = ((mb ? okB >> (it - 4) : okA >> it) & 1u) != 0;
    const bool ex = ((mb ? exB >> (it - 4) : exA >> it) & 1u) != 0;
    if (fill) {
      if (ex) *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + lo) = ok ? v[it] : u32x4{0u, 0u, 0u, 0u};
    } else if (ok) *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + lo) = v[it];
  };
  auto x_load_item = [&](const int part, const int it) __attribute__((always_inline)) { x_load_item_to(part, it, xv); };
  auto x_store_item = [&](const int part, const int it) __attribute__((always_inline)) { x_store_item_from(part, it, xv, false); };
  auto x_load_to = [&](const int part, u32x4 (&v)[XMAX]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < XMAX; ++it)
      if (it < (part == 0 ? 6 : 4)) x_load_item_to(part, it, v);
  };
  auto x_store_from = [&](const int part, const u32x4 (&v)[XMAX]) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < XMAX; ++it)
      if (it < (part == 0 ? 6 : 4)) x_store_item_from(part, it, v, true);
  };

  // ---- operand pointers: centre blocks cb = 0, 1 = tile rows 2 wave + cb (the second row is the immediate offset ROWP / FROW)
  const unsigned char* const img = smem + G::OFF_IMG;
  const int py = 2 * wave;
  const BPtr bc = make_bptr<PAIR>(img + (py + 1) * G::ROWP + (pos + 1) * G::PITCH, q);
  const BPtr br = make_bptr<PAIR>(img + rr * G::ROWP + rc * G::PITCH, q);
  const unsigned char* const fb = smem + G::OFF_FM + py * FROW + pos * FPIX + 16 * q;
  const float* const lb = reinterpret_cast<const float*>(smem + G::OFF_B);

  // ---- output geometry (per-workgroup constants)
  bool cin_frame[2];
  unsigned coff[2], poff[2];
  const size_t tap_bytes = (size_t)a.N * a.H * a.W * 16;
  const buf_rsrc prs = make_rsrc(a.pf, (unsigned)(3 * tap_bytes));
#pragma unroll
  for (int cb = 0; cb < 2; ++cb) {
    const int y = ty0 + py + cb, x = tx0 + pos;
    cin_frame[cb] = (y < a.H) & (x < a.W);
    coff[cb] = (cin_frame[cb] & (a.store_feat != 0)) ? (unsigned)((y * a.W + x) * 64 + 16 * q) : BUF_OOB;
    poff[cb] = (cin_frame[cb] & (q < 3)) ? (unsigned)((y * a.W + x) * 16) + (unsigned)q * (unsigned)tap_bytes : BUF_OOB;   // lane group q = tap q
  }
  const bool rin = rvalid & (ty0 + rr - 1 >= 0) & (ty0 + rr - 1 < a.H) & (tx0 + rc - 1 >= 0) & (tx0 + rc - 1 < a.W);
  auto crop_store = [&](const int plane_idx, const int n, const int cb, const u32x4 v) __attribute__((always_inline)) {
    const size_t so = (size_t)plane_idx * plane_bytes + (size_t)n * frame_bytes;
    buffer_store_b128(v, drs, coff[cb], (unsigned)so);
  };
  // epilogue of the pair's first conv: FM image (zero outside the frame) + centre crop to HBM; vc[cb] = the pixel's 8 channels
  // of octet q = the B fragment of the conv5 partial products
  auto epilogue1 = [&](const f32x4 (&acc1c)[2][2], const f32x4 (&acc1r)[2], const int n, u32x4 (&vc)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      vc[cb] = lrelu_pack8(acc1c[0][cb], acc1c[1][cb], cin_frame[cb]);
      *reinterpret_cast<u32x4*>(smem + G::OFF_FM + (py + cb + 1) * FROW + (pos + 1) * FPIX + 16 * q) = vc[cb];
      crop_store(G::OUT1, n, cb, vc[cb]);
    }
    if (ring) {
      const u32x4 v = lrelu_pack8(acc1r[0], acc1r[1], rin);
      if (rvalid) *reinterpret_cast<u32x4*>(smem + G::OFF_FM + rr * FROW + rc * FPIX + 16 * q) = v;
    }
  };
  auto epilogue2 = [&](const f32x4 (&acc2)[2][2], const int n, u32x4 (&v)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      v[cb] = lrelu_pack8(acc2[0][cb], acc2[1][cb], true);
      crop_store(G::OUT2, n, cb, v[cb]);
    }
  };
  // ---- conv5 partial products: P[4 tap + oc][px] += W5[4 tap + oc][32 ch] d[32 ch][px], one 16x16x32 MFMA per feature and block
  const bool do_p = a.w5p != nullptr && a.pf != nullptr;
  auto p_frag = [&](const int j) __attribute__((always_inline)) {
    return *reinterpret_cast<const f16x8*>(smem + G::OFF_P + j * 1024 + lane * 16);
  };
  auto p_feat = [&](f32x4 (&accp)[2], const int j, const u32x4 (&v)[2]) __attribute__((always_inline)) {
    const f16x8 af = p_frag(j);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) accp[cb] = mfma_16x16x32(af, __builtin_bit_cast(f16x8, v[cb]), accp[cb]);
  };
  auto p_store = [&](const f32x4 (&accp)[2], const int n) __attribute__((always_inline)) {
    // lane group q holds rows 4 q + e = tap q, outputs e (row 4 q + 3 has zero weights): one float4 per pixel and tap
    const size_t so = (size_t)n * a.H * a.W * 16;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      buffer_store_b128(__builtin_bit_cast(u32x4, accp[cb]), prs, poff[cb], (unsigned)so);
    }
  };
  if (do_p && tid < G::NP * 64) *reinterpret_cast<u32x4*>(smem + G::OFF_P + tid * 16) = reinterpret_cast<const u32x4*>(a.w5p)[tid];
  if (tid < 64) reinterpret_cast<float*>(smem + G::OFF_B)[tid] = (tid < 32 ? a.bias[0] : a.bias[1])[tid & 31];

  // accumulators start at the bias: block rb, value e = channel 8 q + 4 rb + e
  auto bias4 = [&](const int conv, const int rb) __attribute__((always_inline)) {
    const float4 v = *reinterpret_cast<const float4*>(lb + 32 * conv + 8 * q + 4 * rb);
    return f32x4{v.x, v.y, v.z, v.w};
  };

  // ---- one merged step s: operands [A1 rb0, A1 rb1, A2 rb0, A2 rb1] at wl (+ lane * 16), Bc blocks 0 / 1, Br
  struct Ops { f16x8 a1[2], a2[2], bcv[2], brv; };
  auto load_m = [&](const int s, const unsigned char* wl, Ops& o, const bool RING) __attribute__((always_inline)) {
    const int kind = step_kind(s), off = step_off<PAIR>(s);
    o.a1[0] = *reinterpret_cast<const f16x8*>(wl);
    o.a1[1] = *reinterpret_cast<const f16x8*>(wl + 1024);
    o.a2[0] = *reinterpret_cast<const f16x8*>(wl + 2048);
    o.a2[1] = *reinterpret_cast<const f16x8*>(wl + 3072);
    o.bcv[0] = *reinterpret_cast<const f16x8*>(bc.k[kind] + off);
    o.bcv[1] = *reinterpret_cast<const f16x8*>(bc.k[kind] + off + G::ROWP);
    if (RING) o.brv = *reinterpret_cast<const f16x8*>(br.k[kind] + off);
  };
  auto mfma_m = [&](const Ops& o, f32x4 (&acc1c)[2][2], f32x4 (&acc1r)[2], f32x4 (&acc2)[2][2], const bool RING) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        acc1c[rb][cb] = mfma_16x16x32(o.a1[rb], o.bcv[cb], acc1c[rb][cb]);
        acc2[rb][cb] = mfma_16x16x32(o.a2[rb], o.bcv[cb], acc2[rb][cb]);
      }
    if (RING) {
      acc1r[0] = mfma_16x16x32(o.a1[0], o.brv, acc1r[0]);
      acc1r[1] = mfma_16x16x32(o.a1[1], o.brv, acc1r[1]);
    }
  };
  struct FOps { f16x8 a2[2], b[2]; };
  auto load_f = [&](const int t, const unsigned char* wl, FOps& o) __attribute__((always_inline)) {
    const int off = (t / 3) * FROW + (t % 3) * FPIX;
    o.a2[0] = *reinterpret_cast<const f16x8*>(wl);
    o.a2[1] = *reinterpret_cast<const f16x8*>(wl + 1024);
    o.b[0] = *reinterpret_cast<const f16x8*>(fb + off);
    o.b[1] = *reinterpret_cast<const f16x8*>(fb + off + FROW);
  };
  auto mfma_f = [&](const FOps& o, f32x4 (&acc2)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc2[rb][cb] = mfma_16x16x32(o.a2[rb], o.b[cb], acc2[rb][cb]);
  };

  const u32x4* __restrict__ wsrc = reinterpret_cast<const u32x4*>(a.w);

  if constexpr (G::RES) {
    // =====================================================================================================
    // pair 0: the whole 74-fragment stream is resident.  The first tile-frame starts on the first five steps' 20 fragments;
    // the other 54 are already in flight and are stored behind those steps.
    // =====================================================================================================
    constexpr int HEAD = 20, HITER = (HEAD * 64 + NTHR - 1) / NTHR, RITER = ((G::NFRAG - HEAD) * 64 + NTHR - 1) / NTHR;   // 3, 7
    u32x4 wrest[RITER];
    {
      u32x4 whead[HITER];
      lframe = (size_t)f0 * frame_bytes;
      x_load_to(0, xv);
#pragma unroll
      for (int it = 0; it < HITER; ++it) whead[it] = wsrc[min(tid + it * NTHR, HEAD * 64 - 1)];
#pragma unroll
      for (int it = 0; it < RITER; ++it) wrest[it] = wsrc[HEAD * 64 + min(tid + it * NTHR, (G::NFRAG - HEAD) * 64 - 1)];
      x_store_from(0, xv);
#pragma unroll
      for (int it = 0; it < HITER; ++it) {
        const int j = tid + it * NTHR;
        if (j < HEAD * 64) *reinterpret_cast<u32x4*>(smem + G::OFF_W + j * 16) = whead[it];
      }
    }
    __syncthreads();
    bool first = true;
    const unsigned char* const wl = smem + G::OFF_W + lane * 16;

    for (int n = f0; n < a.N; n += gf) {
      const bool more = n + gf < a.N;
      STAMP(ts0);
      if (more) lframe = (size_t)(n + gf) * frame_bytes;
      f32x4 acc1c[2][2], acc1r[2], acc2[2][2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        acc1r[rb] = bias4(0, rb);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) { acc1c[rb][cb] = bias4(0, rb); acc2[rb][cb] = bias4(1, rb); }
      }
      STAMP(ts1);
      STAMP_ADD(0, ts0, ts1);
      auto merged = [&](auto ring_tag) __attribute__((always_inline)) {
        constexpr bool RING = decltype(ring_tag)::value;
        Ops ops[2];
        load_m(0, wl, ops[0], RING);
        static_for<0, G::NM>([&](auto si) __attribute__((always_inline)) {
          constexpr int s = decltype(si)::value;
          if constexpr (s == 5) {
            if (first) {                 // the rest of the stream lands behind the first five steps (first tile-frame only)
#pragma unroll
              for (int it = 0; it < RITER; ++it) {
                const int j = tid + it * NTHR;
                if (j < (G::NFRAG - HEAD) * 64) *reinterpret_cast<u32x4*>(smem + G::OFF_W + HEAD * 1024 + j * 16) = wrest[it];
              }
              __syncthreads();
              first = false;
              load_m(5, wl + 5 * 4096, ops[1], RING);
            }
          }
          if constexpr (s + 1 < G::NM) {
            if constexpr (s + 1 == 5) { if (!first) load_m(s + 1, wl + (s + 1) * 4096, ops[(s + 1) & 1], RING); }
            else load_m(s + 1, wl + (s + 1) * 4096, ops[(s + 1) & 1], RING);
          }
          __builtin_amdgcn_sched_barrier(0);
          mfma_m(ops[s & 1], acc1c, acc1r, acc2, RING);
          __builtin_amdgcn_sched_barrier(0);
          // the next tile-frame's halo: one load behind each of the first six steps
          if constexpr (s < 6) { if (more) x_load_item(0, s); }
        });
      };
      if (ring) merged(std::true_type{});
      else merged(std::false_type{});
      STAMP(ts2);
      STAMP_ADD(1, ts1, ts2);

      u32x4 vc[2];
      epilogue1(acc1c, acc1r, n, vc);
      f32x4 accp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      if (do_p) {        // x2 (centre tap of the image, still intact before the barrier) and f1
        const f16x8 a0 = p_frag(0), a1 = p_frag(1);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const f16x8 b0 = *reinterpret_cast<const f16x8*>(bc.k[0] + G::ROWP + G::PITCH + cb * G::ROWP);
          const f16x8 b1 = *reinterpret_cast<const f16x8*>(bc.k[3] + G::ROWP + G::PITCH + cb * G::ROWP);
          accp[cb] = mfma_16x16x32(a0, b0, accp[cb]);
          accp[cb] = mfma_16x16x32(a1, b1, accp[cb]);
        }
        p_feat(accp, 2, vc);
      }
      STAMP(ts3);
      STAMP_ADD(2, ts2, ts3);
      __syncthreads();                                // FM complete; every wave is done with the input image
      STAMP(ts4);
      STAMP_ADD(3, ts3, ts4);
      {
        const unsigned char* const wf = wl + G::NM * 4096;
        FOps fo[2];
        load_f(0, wf, fo[0]);
        static_for<0, 9>([&](auto ti) __attribute__((always_inline)) {
          constexpr int t = decltype(ti)::value;
          if constexpr (t + 1 < 9) load_f(t + 1, wf + (t + 1) * 2048, fo[(t + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          mfma_f(fo[t & 1], acc2);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (t < 6) { if (more) x_store_item(0, t); }      // the image is dead: the next tile-frame's halo goes in
        });
      }
      STAMP(ts5);
      STAMP_ADD(4, ts4, ts5);
      epilogue2(acc2, n, vc);
      if (do_p) {
        p_feat(accp, 3, vc);                          // f2
        p_store(accp, n);
      }
      STAMP(ts6);
      STAMP_ADD(5, ts5, ts6);
      __syncthreads();                                // image of the next tile-frame visible; FM free again
      STAMP(ts7);
      STAMP_ADD(6, ts6, ts7);
    }
  } else {
    // =====================================================================================================
    // pair 1: 146 fragments through THREE 12-fragment LDS buffers in 14 chunks per tile-frame: chunks 0..9 = three merged
    // steps each, 10 = steps 30, 31, 11 / 12 / 13 = FM steps 0..2 / 3..5 / 6..8.  Barrier c follows the first step of chunk c
    // (c = 11: it IS the barrier in front of the FM phase): behind it every wave has left chunk c - 1, so chunk c + 2 is
    // committed into that buffer there and is visible by barrier c + 1, before any wave starts chunk c + 2; the registers it
    // frees take chunk c + 4 (two register sets, by chunk parity: 14 is even, so parities carry over to the next tile-frame).
    // 14 chunks on three buffers: chunk c of tile-frame k uses buffer (c + 2 k) mod 3 - a triple of offsets that rotates by two
    // per tile-frame.  Image parts die at barriers too (x2 after step 13 -> barrier 5, f1 after step 22 -> barrier 8, f2 ->
    // barrier 11): the next tile-frame's parts are stored behind them.
    // =====================================================================================================
    constexpr int CH = G::CH, NCHUNK = 14;
    auto cfirst = [](const int c) constexpr { return c <= 10 ? 12 * c : 128 + 6 * (c - 11); };      // first fragment of chunk c
    auto cfrags = [](const int c) constexpr { return c < 10 ? 12 : c == 10 ? 8 : 6; };               // its fragments
    const buf_rsrc wrs = make_rsrc(a.w, (unsigned)G::NFRAG * 1024u);
    u32x4 wset[2][2];
    auto w_prefetch = [&](const int c, const int it) __attribute__((always_inline)) {       // chunk c (0..13) into set c & 1
      const int items = cfrags(c) * 64;
      if (it * NTHR >= items) return;
      const unsigned j = (unsigned)(tid + it * NTHR);
      wset[c & 1][it] = buffer_load_b128(wrs, (j < (unsigned)items ? j : (unsigned)tid % (unsigned)items) * 16u, (unsigned)(cfirst(c) * 1024));
    };
    unsigned wb3[3] = {0u, (unsigned)(CH * 1024), (unsigned)(2 * CH * 1024)};       // buffer offsets of chunks = 0, 1, 2 (mod 3) of THIS tile-frame
    auto w_commit = [&](const int c, const unsigned bufoff, const int it) __attribute__((always_inline)) {
      const int items = cfrags(c) * 64;
      if (it * NTHR >= items) return;
      if (tid + it * NTHR < items) *reinterpret_cast<u32x4*>(smem + G::OFF_W + bufoff + (tid + it * NTHR) * 16) = wset[c & 1][it];
    };
    {
      u32x4 xb[XMAX], xc[XMAX];
      lframe = (size_t)f0 * frame_bytes;
      x_load_to(0, xv);
      w_prefetch(0, 0); w_prefetch(0, 1);
      w_prefetch(1, 0); w_prefetch(1, 1);
      x_load_to(1, xb);
      x_load_to(2, xc);
      x_store_from(0, xv);
      w_commit(0, wb3[0], 0); w_commit(0, wb3[0], 1);
      w_commit(1, wb3[1], 0); w_commit(1, wb3[1], 1);
      w_prefetch(2, 0); w_prefetch(2, 1);
      w_prefetch(3, 0); w_prefetch(3, 1);
      x_store_from(1, xb);
      x_store_from(2, xc);
    }
    __syncthreads();
    const unsigned char* const wl0 = smem + G::OFF_W + lane * 16;

    for (int n = f0; n < a.N; n += gf) {
      const bool more = n + gf < a.N;
      STAMP(ts0);
      if (more) lframe = (size_t)(n + gf) * frame_bytes;
      f32x4 acc1c[2][2], acc1r[2], acc2[2][2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        acc1r[rb] = bias4(0, rb);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) { acc1c[rb][cb] = bias4(0, rb); acc2[rb][cb] = bias4(1, rb); }
      }
      STAMP(ts1);
      STAMP_ADD(0, ts0, ts1);
      u32x4 vc[2];
      f32x4 accp[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      // LDS address of merged step s / FM step t: chunk, position inside the chunk
      auto wl_m = [&](const int s) __attribute__((always_inline)) {
        const int c = s < 30 ? s / 3 : 10, p = s < 30 ? s % 3 : s - 30;
        return wl0 + wb3[c % 3] + p * 4096;
      };
      auto wl_f = [&](const int t) __attribute__((always_inline)) {
        const int c = 11 + t / 3, p = t % 3;
        return wl0 + wb3[c % 3] + p * 2048;
      };
      // hooks behind barrier c: commit chunk c + 2, then prefetch chunk c + 4; chunks past 13 are the next tile-frame's 0.. -
      // the same stream, but ITS buffer triple (rotated by two)
      auto commit2 = [&](const int c, const int it) __attribute__((always_inline)) {
        const int cc = c + 2;
        if (cc < NCHUNK) w_commit(cc, wb3[cc % 3], it);
        else w_commit(cc - NCHUNK, wb3[(cc - NCHUNK + 2) % 3], it);
      };
      auto prefetch4 = [&](const int c, const int it) __attribute__((always_inline)) { w_prefetch((c + 4) % NCHUNK, it); };

      auto merged = [&](auto ring_tag) __attribute__((always_inline)) {
        constexpr bool RING = decltype(ring_tag)::value;
        Ops ops[2];
        load_m(0, wl_m(0), ops[0], RING);
        static_for<0, G::NM>([&](auto si) __attribute__((always_inline)) {
          constexpr int s = decltype(si)::value;
          constexpr int c = s < 30 ? s / 3 : 10, p = s < 30 ? s % 3 : s - 30;
          if constexpr (s + 1 < G::NM) load_m(s + 1, wl_m(s + 1), ops[(s + 1) & 1], RING);
          __builtin_amdgcn_sched_barrier(0);
          mfma_m(ops[s & 1], acc1c, acc1r, acc2, RING);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (p == 0) __syncthreads();            // barrier c
          // weight hand-over: chunks of three steps commit behind step 1 and prefetch behind step 2; chunk 10 (two steps)
          // does both behind its second step
          if constexpr (c < 10) {
            if constexpr (p == 1) { commit2(c, 0); commit2(c, 1); }
            if constexpr (p == 2) { prefetch4(c, 0); prefetch4(c, 1); }
          } else if constexpr (p == 1) {
            commit2(c, 0); commit2(c, 1);
            prefetch4(c, 0); prefetch4(c, 1);
          }
          // the next tile-frame's image
          if (more) {
            if constexpr (s < 6) x_load_item(0, s);
            if constexpr (s >= 16 && s < 22) {                // x2 died at barrier 5 (behind step 15)
              x_store_item(0, s - 16);
              if constexpr (s - 16 < 4) x_load_item(1, s - 16);
            }
            if constexpr (s >= 25 && s < 29) {                // f1 died at barrier 8 (behind step 24)
              x_store_item(1, s - 25);
              x_load_item(2, s - 25);
            }
          }
        });
      };
      if (ring) merged(std::true_type{});
      else merged(std::false_type{});
      STAMP(ts2);
      STAMP_ADD(1, ts1, ts2);

      epilogue1(acc1c, acc1r, n, vc);
      if (do_p) p_feat(accp, 0, vc);                          // f3
      STAMP(ts3);
      STAMP_ADD(2, ts2, ts3);
      __syncthreads();                                        // barrier 11: FM complete, f2 dead, chunk 10 left
      STAMP(ts4);
      STAMP_ADD(3, ts3, ts4);
      {
        FOps fo[2];
        load_f(0, wl_f(0), fo[0]);
        static_for<0, 9>([&](auto ti) __attribute__((always_inline)) {
          constexpr int t = decltype(ti)::value;
          constexpr int c = 11 + t / 3, p = t % 3;
          if constexpr (t + 1 < 9) load_f(t + 1, wl_f(t + 1), fo[(t + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
          mfma_f(fo[t & 1], acc2);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (c > 11 && p == 0) __syncthreads();                         // barriers 12, 13 (11 sits in front of the phase)
          if constexpr (c == 11) {                                                 // behind barrier 11: steps 0, 1
            if constexpr (p == 0) { commit2(c, 0); commit2(c, 1); }
            if constexpr (p == 1) { prefetch4(c, 0); prefetch4(c, 1); }
          } else {                                                                 // behind barriers 12, 13: steps 1, 2 of the chunk
            if constexpr (p == 1) { commit2(c, 0); commit2(c, 1); }
            if constexpr (p == 2) { prefetch4(c, 0); prefetch4(c, 1); }
          }
          if constexpr (t < 4) { if (more) x_store_item(2, t); }                   // f2 died at barrier 11
        });
      }
      STAMP(ts5);
      STAMP_ADD(4, ts4, ts5);
      epilogue2(acc2, n, vc);
      if (do_p) {
        p_feat(accp, 1, vc);                          // f4
        p_store(accp, n);
      }
      STAMP(ts6);
      STAMP_ADD(5, ts5, ts6);
      // the next tile-frame's chunk c uses the buffer this one's chunk c + 2 used (14 = 2 mod 3)
      const unsigned t0 = wb3[0], t1 = wb3[1];
      wb3[0] = wb3[2]; wb3[1] = t0; wb3[2] = t1;
    }
  }
#ifdef SELFC_STAMPS
  STAMP(tk1);
  if (a.stamps && lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)blockIdx.x * NWAVE + wave) * 8;
#pragma unroll
    for (int k = 0; k < 7; ++k) o[k] = phase[k];
    o[7] = tk1 - tk0;
  }
#endif
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

template <int PAIR>
int launch_pair16(F16Args& a, int maxwg, hipStream_t s) {
  using G = Geo<PAIR>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&fused_f16_kernel<PAIR>), G::LDS, optin); e != hipSuccess) return hip_rc(e);
  // frame walk: the launch geometry of csrc/fused_f.hip (gf workgroups per spatial tile), with at least TWO frames per workgroup
  // when there are (the 32x32x16 kernels wanted three; measured on the headline: 2 -> +1 %, 4 -> -1 % against 3)
  const int gmax = a.ntiles >= maxwg ? 1 : (maxwg / a.ntiles < a.N ? maxwg / a.ntiles : a.N);
  static const int minrounds = getenv("SELFC_FUSEDF_MINROUNDS") ? atoi(getenv("SELFC_FUSEDF_MINROUNDS")) : 2;
  int rounds = (a.N + gmax - 1) / gmax;
  // (not when every tile-frame gets a workgroup of its own in ONE round - training crops at 1..4 septuplets per rank: there the
  // launch is latency, and one tile-frame per workgroup halves it: captured training step 7.75 -> 7.43 ms at one septuplet)
  if (rounds < minrounds && (long)a.N * a.ntiles > maxwg) rounds = a.N < minrounds ? a.N : minrounds;
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
  hipLaunchKernelGGL(fused_f16_kernel<PAIR>, dim3((unsigned)gx), dim3(NTHR), G::LDS, s, a);
  return hip_rc(hipGetLastError());
}

}  // namespace

// The two pair launches of launch_fused_f (csrc/fused_f.hip) on the 16x16x32 kernels.  w16 = [pair 0: 74 fragments][pair 1: 146],
// w5p16 = [pair 0: 4][pair 1: 2] (or null), pf = [2 pairs][3 taps][N][H][W][4] (or null).  Returns 0, an error code, or
// SELFC_F16_TOO_BIG (nothing launched): the kernels address the dense buffer and pf through 32-bit buffer-resource offsets, so
// six planes (resp. the three taps of pf) must stay below 2 GiB - about 5.5 M pixel-frames, 43 1080p frames in one call; beyond
// that launch_fused_f takes the flat-addressed 32x32x16 kernels of csrc/fused_f.hip.
int launch_fused_f16_pairs(void* dense, const void* w16, const float* const* bias, int N, int H, int W, hipStream_t s,
                           const void* w5p16, float* pf, int keep_features) {
  if ((size_t)N * H * W * 64 * (Geo<1>::NPLANE_IN + 2) >= 0x7fff0000ull || (size_t)N * H * W * 48 >= 0x7fff0000ull) return SELFC_F16_TOO_BIG;
  static const int maxwg = getenv("SELFC_FUSEDF_MAXWG") ? atoi(getenv("SELFC_FUSEDF_MAXWG")) : 256;
  F16Args a{};
  a.dense = (f16*)dense;
  a.N = N; a.H = H; a.W = W;
  a.tiles_x = (W + TS - 1) / TS;
  a.tiles_y = (H + TS - 1) / TS;
  a.ntiles = a.tiles_x * a.tiles_y;
  a.plane = (size_t)N * H * W * 32;
  const bool with_p = w5p16 && pf;
  a.w = (const f16*)w16;
  a.bias[0] = bias[0]; a.bias[1] = bias[1];
  a.w5p = with_p ? (const f16*)w5p16 : nullptr;
  a.pf = with_p ? pf : nullptr;
  a.store_feat = 1;                                  // f1, f2: pair 1 reads them
  int rc = launch_pair16<0>(a, maxwg > 0 ? maxwg : 256, s);
  if (rc) return rc;
  a.w = (const f16*)w16 + (size_t)Geo<0>::NFRAG * 512;
  a.bias[0] = bias[2]; a.bias[1] = bias[3];
  a.w5p = with_p ? (const f16*)w5p16 + (size_t)Geo<0>::NP * 512 : nullptr;
  a.pf = with_p ? pf + (size_t)N * H * W * 12 : nullptr;
  a.store_feat = (keep_features || !with_p) ? 1 : 0;
  return launch_pair16<1>(a, maxwg > 0 ? maxwg : 256, s);
}

}  // namespace selfc
