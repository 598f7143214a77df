// Fused conv1..conv4 of the G / H dense blocks (Subnet_constructor.py:126-129, cin = 3) for gfx950.
//
// One persistent workgroup owns a 16x16 output tile at a time and computes all four 3x3 convs of
// one net with every intermediate feature kept in LDS (halo recompute: conv k is evaluated on the
// tile grown by 4-k pixels), instead of four launches that each round-trip the features through
// HBM with 1-3 waves of short-lived workgroups.  Per tile and net:
//
//   X  : y1 halo (24x24 px, 4 x f16 per pixel: c0 c1 c2 0)          <- x1 buffer (fp32 NHWC4)
//   F1 : lrelu(conv1) on 22x22      F2 : conv2 on 20x20      F3 : conv3 on 18x18     (f16, LDS)
//   conv4 on 16x16 -> HBM only.  The centre 16x16 crops of F1..F3 also go to HBM (planes 0..2 of
//   the net's dense buffer) because the temporal conv5 reads all four features.
//   Features outside the image are stored as ZERO (they are the next conv's zero padding).
//
// MFMA 32x32x16 f16, D[outch][pixel]; an M-tile is 32 pixels of the conv's region laid out so that every
// ds_read_b128 lane group hits 16 distinct 16-byte slots in EVERY feature image it reads (mtile_geom:
// two-row checkerboard tiles + column strips for region widths 20 / 18 / 16; conv1: 32 consecutive pixels).
// The 3-channel input enters as an "im2col48" stage: K = 12 taps x 4 (c0 c1 c2 0), so a lane's 8
// K-entries are two whole pixels of X = two ds_read_b64, no repacking (3 MFMAs instead of 2).
// Weights: the packed fragment stream of the four convs (120 KiB per net) is read in chunks of
// <= 21 fragments through an LDS double buffer; the stream wraps around from conv4 to the next
// tile's conv1, so the prefetch never stalls at a tile boundary.  One barrier per chunk.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "prof.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace selfc {

struct FGArgs {
  const float* x1;          // [N][H][W][4] fp32 (y1)
  const f16* w[2];          // fused fragment stream per net (packing.py: pack_fused_gh), 120 fragments
  const float* bias[2][4];  // 32 floats per conv
  f16* dense[2];            // plane-blocked [4][N][H][W][32]
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
  size_t plane;
  int ablate;               // unused (kept for ABI stability of the internal struct)
  unsigned long long* stamps;   // diagnostic build only (-DSELFC_STAMPS): per wave 4 phase-cycle sums
};

namespace {

// DEPTH convs are fused.  The geometry below is written for DEPTH in {3, 4}; the product is depth 4 (conv1..4).  A depth-3
// build + conv4 as its own persistent launch was measured in round 1 (-3 % ... +3 %) and removed.
constexpr int DEPTH = 4;
constexpr int TS = 16;
constexpr int XS = TS + 2 * DEPTH;         // X halo side (24 / 22)
constexpr int XPITCH = XS * 8;             // bytes
constexpr int X_BYTES = XS * XPITCH;       // 4608
// LDS row pitches of F1..F3 (bytes).  A ds_read_b128 is served in lane groups G0 = {0-3, 12-15, 20-27}, G1 = {4-11, 16-19, 28-31}
// (and the same + 32): 16 lanes that must hit 16 different 16-byte slots (mod 16).  With the 80-byte pixel a pixel (r, c) of an
// image of pitch p slots sits on slot p r + 5 c.  Round 5: M-tiles are no longer 32 consecutive pixels of the region (which is
// conflict-free for ONE reader per image only: F1's conv3 / conv4 reads and F2's conv4 reads took 8 LDS cycles instead of 4 -
// 44 % more B-read cycles per tile, SQ_LDS_BANK_CONFLICT 19 % of the LDS cycles) but (mtile_geom):
//   * main tiles = two image rows x 16 columns, each lane group a CHECKERBOARD of them (G0: odd columns of the first row + even
//     columns of the second; G1 the opposite): 5 c takes every residue once over 16 columns, so the second row's eight slots
//     avoid the first row's eight for EVERY even p;
//   * the region's columns 16.. (20x20: 4 columns, 18x18: 2) = strip tiles of 4 columns x 8 rows (a lane group = the even or the
//     odd rows of it: slots 2 p k + {0, 5, 10, 15}) or 2 columns x 16 rows (a group = 8 consecutive rows: p k + {0, 5}).
// p1 = 2 (mod 16) serves conv2's 4-column strips and conv3's 2-column strips, p2 = 2 (mod 4) conv3's strips, p3 is even:
// tools/lds_conflicts_fused_gh.py replays every B read of a tile - 4.0 LDS cycles per read everywhere but conv2's last strip
// tile (clamped rows: 4.15), 6,120 -> 4,284 cycles per tile and net.
constexpr int P1 = 1824, P2 = 1632, P3 = DEPTH == 4 ? 1440 : 16;
static_assert(DEPTH != 4 || (P1 >= (TS + 6) * 80 && (P1 / 16) % 16 == 2 && P2 >= (TS + 4) * 80 && (P2 / 16) % 4 == 2 && P3 >= (TS + 2) * 80 && (P3 / 16) % 2 == 0),
              "pitches: see above");
constexpr int F1_BYTES = (TS + 2 * (DEPTH - 1)) * P1, F2_BYTES = (TS + 2 * (DEPTH - 2)) * P2, F3_BYTES = DEPTH == 4 ? (TS + 2) * P3 : 0;
constexpr int WCH = 21;                    // fragments per weight chunk buffer
constexpr int W_BYTES = WCH * 1024;
// depth 3: the whole 63-fragment stream of conv1..3 stays RESIDENT in LDS (no streaming, no hand-over waits)
constexpr bool WRES = DEPTH == 3;
constexpr int NFRAG_RES = 63;
constexpr int OFF_F1 = 0, OFF_F2 = OFF_F1 + F1_BYTES, OFF_F3 = OFF_F2 + F2_BYTES;
constexpr int OFF_X = OFF_F3 + F3_BYTES;   // 2 buffers
constexpr int OFF_W = OFF_X + 2 * X_BYTES; // 2 buffers
constexpr int OFF_B = OFF_W + (WRES ? NFRAG_RES * 1024 : 2 * W_BYTES); // biases of the four convs: 4 x 32 floats
// conv1's three fragments stay RESIDENT: streamed like the rest, conv1's short phase (3 MFMA steps) had to wait for conv2's
// first chunk right behind the previous tile's conv4 feature stores - loads and stores retire in order on one VM counter,
// so that wait was the stores' full write latency, once per tile.  Now conv4's last chunk streams conv2's first chunk in
// (committed BEFORE conv4's stores) and conv1 touches no global memory before its own epilogue.
constexpr int OFF_W1 = OFF_B + 512;
constexpr int FG_LDS = OFF_W1 + 3 * 1024;
static_assert(FG_LDS <= 160 * 1024, "LDS budget");
#ifndef SELFC_RD1
#define SELFC_RD1 3
#endif
#ifndef SELFC_RD2
#define SELFC_RD2 3
#endif
constexpr int RD1 = SELFC_RD1, RD2 = SELFC_RD2;   // operand ring depths (one / two M-tiles per wave)
constexpr int NWAVE = 8, NTHR = NWAVE * 64;   // 2 waves per SIMD: one wave's epilogue / LDS latency hides under the other's MFMAs
constexpr int WITER = (WCH * 64 + NTHR - 1) / NTHR;   // 3

template <int J> struct FeatGeom;
template <> struct FeatGeom<1> { static constexpr int off = OFF_F1, pitch = P1; };
template <> struct FeatGeom<2> { static constexpr int off = OFF_F2, pitch = P2; };
template <> struct FeatGeom<3> { [[maybe_unused]] static constexpr int off = OFF_F3, pitch = P3; };

// fragment offsets of the fused stream: per conv [im2col48: 3][feature j: 18 each]
constexpr int LAYER_OFF[5] = {0, 0, 3, 24, 63};

#ifdef SELFC_STAMPS
#define STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#ifdef SELFC_STAMP_BYCONV      // the four slots = MFMA-loop cycles of conv1..conv4 instead of the four phases
#define STAMP_ADD(k, a, b)
#define STAMP_LOOP(K, a, b) c.phase[(K) - 1] += (b) - (a)
#else
#define STAMP_ADD(k, a, b) c.phase[k] += (b) - (a)
#define STAMP_LOOP(K, a, b) c.phase[1] += (b) - (a)
#endif
#else
#define STAMP(var)
#define STAMP_ADD(k, a, b)
#define STAMP_LOOP(K, a, b)
#endif

typedef __attribute__((address_space(1))) f16 gf16;                 // explicitly global: a laundered pointer would otherwise become flat
typedef __attribute__((address_space(1))) u32x4 gu32x4;

struct Ctx {
#ifdef SELFC_STAMPS
  unsigned long long phase[4];   // 0 setup/prefetch, 1 MFMA loop, 2 epilogue, 3 commit + barrier
#endif
  unsigned char* smem;
  const u32x4* wsrc;
  gf16* dense;        // this net's dense buffer (kernel argument, kept opaque: see the kernel)
  int tid, lane, wave, half;
  int par;            // weight buffer that holds the CURRENT chunk
  bool first;         // resident mode: the workgroup's first tile streams the fragments INTO their resident places
  u32x4 wreg[WITER];
};

template <int OFF, int NFR>
__device__ __forceinline__ void w_prefetch(Ctx& c) {
  if (WRES && !c.first) return;
#pragma unroll
  for (int it = 0; it < WITER; ++it) {
    const int i = min(c.tid + it * NTHR, NFR * 64 - 1);
    c.wreg[it] = c.wsrc[OFF * 64 + i];
  }
}
template <int OFF, int NFR>
__device__ __forceinline__ void w_commit(Ctx& c) {
  if (WRES && !c.first) return;
  // streamed: the other half of the double buffer; resident (depth 3, first tile only): the chunk's own place
  unsigned char* dst = c.smem + OFF_W + (WRES ? OFF * 1024 : (c.par ^ 1) * W_BYTES);
#pragma unroll
  for (int it = 0; it < WITER; ++it) {
    const int i = c.tid + it * NTHR;
    // Wait for the prefetch load on EVERY path, not only inside the branch: a wave whose lanes all skip the store never
    // executed the branch's s_waitcnt, so hipcc kept the load "pending" and, when it later reused the register in the
    // middle of an epilogue, drained the whole VM queue there (vmcnt(0)) - i.e. waited for the feature stores just issued.
    // (Removes those drains from the ISA; measured time unchanged)
    u32x4 v = c.wreg[it];
    asm volatile("" : "+v"(v));
    if (i < NFR * 64) *reinterpret_cast<u32x4*>(dst + i * 16) = v;
  }
}

// ---- epilogue of conv K: bias is already in the accumulator; LeakyReLU, zero outside the image, f16 -> LDS feature image
// (+ centre crop to HBM).  Split in PIECES so that the epilogues of conv2 / conv3 can be hung, a few instructions at a
// time, behind the MFMA steps of the NEXT conv's first chunk (which reads only older features): piece p < 4 MT packs
// 4 channels of M-tile p / 4, piece 4 MT + m swaps / masks / stores M-tile m.
template <int K> struct ConvGeom {
  static constexpr int R = TS + 2 * (DEPTH - K), NPX = R * R, NTL = (NPX + 31) / 32, MT = (NTL + NWAVE - 1) / NWAVE;
};
// Which waves own a SECOND M-tile: conv3's three (tiles 8..10) go to waves 0..2, conv2's five (tiles 8..12) to waves 3..7 -
// so that no SIMD (waves w and w + 4) carries more than one of each, and a wave is in one of only TWO classes:
// X (waves 0..2: one tile in conv2, two in conv3) and Y (waves 3..7: two in conv2, one in conv3).
template <int K> constexpr int second_tile_first_wave() { return (DEPTH == 4 && K == 2) ? 3 : 0; }
template <int K>
__device__ __forceinline__ void mtile_geom(const Ctx& c, const int m, int& r, int& cc, bool& valid) {
  using G = ConvGeom<K>;
  constexpr int W0 = second_tile_first_wave<K>();
  static_assert(G::NTL - NWAVE <= NWAVE - W0, "second tiles fit the waves from W0 on");
  const int mt = m == 0 ? c.wave : NWAVE + c.wave - W0;
  const bool own = (m == 0 || c.wave >= W0) & (mt < G::NTL);
  if constexpr (K == 1 || DEPTH != 4) {
    // conv1 reads only the X halo (ds_read_b64): 32 consecutive pixels of the 22x22 region per M-tile
    const int q = mt * 32 + (c.lane & 31);
    valid = own & (q < G::NPX);
    const int qc = min(q, G::NPX - 1);
    r = qc / G::R;
    cc = qc - r * G::R;
  } else {
  // conv2..4 (regions 20 / 18 / 16 wide): checkerboard main tiles + column-strip tiles, see the pitch comment above.
  // lane -> (group, index in the group's hardware order): lanes {0-3, 12-15, 20-27} are group 0
  constexpr int NMAIN = G::R / 2, LW = G::R - 16;
  static_assert(NMAIN + (LW * G::R + 31) / 32 == G::NTL && (LW == 0 || LW == 2 || LW == 4), "tile count of the region");
  const int l = c.lane & 31;
  const int g = (0x96 >> (l >> 2)) & 1, idx = ((l >> 3) << 2) | (l & 3);
  int rr;
  if (mt < NMAIN) {
    const int hi = idx >> 3;                    // first / second row of the pair
    rr = 2 * mt + hi;
    cc = 2 * (idx & 7) + (g ^ hi ^ 1);
  } else if (LW == 4) {
    rr = 8 * (mt - NMAIN) + 2 * (idx >> 2) + g;
    cc = 16 + (idx & 3);
  } else {
    rr = 16 * (mt - NMAIN) + 8 * g + (idx >> 1);
    cc = 16 + (idx & 1);
  }
  valid = own & (rr < G::R);
  r = min(rr, G::R - 1);
  }
}
template <int K, int MT>
__device__ __forceinline__ void epi_piece(const Ctx& c, const FGArgs& a, const int net, const size_t fofs, const int ty0, const int tx0,
                                          const f32x16 (&acc)[MT], uint32_t (&rr)[MT][4][2], const int p) {
  if (p < 4 * MT) {
    const int m = p >> 2, g = p & 3;
    rr[m][g][0] = lrelu_pack2(acc[m][4 * g + 0], acc[m][4 * g + 1]);
    rr[m][g][1] = lrelu_pack2(acc[m][4 * g + 2], acc[m][4 * g + 3]);
    return;
  }
  const int m = p - 4 * MT;
  int r, cc;
  bool valid;
  mtile_geom<K>(c, m, r, cc, valid);
  gf16* __restrict__ dplane = c.dense + (size_t)(K - 1) * a.plane;
  // the region leaves the image only for tiles on the frame border (wave-uniform test)
  const bool border = (ty0 - (DEPTH - K) < 0) | (tx0 - (DEPTH - K) < 0) | (ty0 + TS + (DEPTH - K) > a.H) | (tx0 + TS + (DEPTH - K) > a.W);
  const int ar = r - (DEPTH - K), ac = cc - (DEPTH - K);
  const int y = ty0 + ar, x = tx0 + ac;
  const bool inimg = (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
  // features outside the image are the next conv's zero padding: one AND per packed dword, applied after the
  // half swap (lanes l and l + 32 hold the same pixel); laundered so that it stays an AND on the 8 packed
  // dwords instead of 16 selects on the floats in front of the conversion
  uint32_t keep = (border && !inimg) ? 0u : 0xffffffffu;
  asm volatile("" : "+v"(keep));
#pragma unroll
  for (int gp = 0; gp < 2; ++gp) {
    u32x4 v;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const auto sw = __builtin_amdgcn_permlane32_swap(rr[m][2 * gp][d], rr[m][2 * gp + 1][d], false, false);
      v[d] = sw[0] & keep;
      v[2 + d] = sw[1] & keep;
    }
    if (valid) {
      if (K < DEPTH) {
        constexpr int pitch = FeatGeom<(K < DEPTH ? K : 1)>::pitch;
        *reinterpret_cast<u32x4*>(c.smem + FeatGeom<(K < DEPTH ? K : 1)>::off + r * pitch + cc * PS + (16 * gp + 8 * c.half) * 2) = v;
      }
      const bool centre = (ar >= 0) & (ar < TS) & (ac >= 0) & (ac < TS);
      if (centre && inimg) {
        gf16* const dst = dplane + fofs + (unsigned)((y * a.W + x) * 32 + 16 * gp + 8 * c.half);
        *(gu32x4*)dst = v;
      }
    }
  }
}

struct AccPair { f32x16 a[2]; };      // accumulators of conv2 / conv3 (2 M-tiles per wave) handed to the next conv

// One 3x3 conv K (1..4) of the current tile.  NEXT_OFF/NEXT_N: the chunk that follows this conv's
// last chunk in the stream (prefetched during that chunk).
// prev: accumulators of conv K-1 whose epilogue is still pending (K = 3, 4): it runs, piece by piece, behind the MFMA
// steps of this conv's first chunk (im2col + F1: neither reads what that epilogue writes).  out: K = 2, 3 leave their
// accumulators there instead of running their epilogue.
// HAS2 / PREV2: does this wave own a SECOND M-tile of conv K / of conv K-1?  conv2's 20x20 and conv3's 18x18 regions are 13 and
// 11 tiles of 32 pixels for 8 waves: three (conv2) / five (conv3) waves have no second tile.  They used to run its MFMAs, operand
// reads and epilogue pieces on clamped addresses with the stores masked - a quarter of conv3's and a fifth of conv2's issue
// slots; the kernel's time is the sum of what the two waves of a SIMD issue.  Two wave classes (see
// mtile_geom) run their own straight-line instance of the whole tile loop (dispatch once, in the kernel; identical barriers).
template <int K, bool HAS2 = true, bool PREV2 = true>
__device__ __forceinline__ void conv_fused(Ctx& c, const FGArgs& a, const int net, const int xbuf,
                                           const size_t fofs, const int ty0, const int tx0, AccPair* prev, AccPair* out) {
  constexpr int MT = ConvGeom<K>::MT;
  constexpr int MTE = (MT == 2 && !HAS2) ? 1 : MT;        // M-tiles this wave really computes
  constexpr int NCH = K == 1 ? 1 : K - 1;          // weight chunks of this conv: [im2col(+f1)], [f2], [f3]
  constexpr bool DEFER = K >= 2 && K < DEPTH, PENDING = K >= 3;
  static_assert(!DEFER || MT == 2, "AccPair");
  unsigned char* const smem = c.smem;

  int r[MT], cc[MT];
  bool valid[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) mtile_geom<K>(c, m, r[m], cc[m], valid[m]);
  // accumulators start at the bias (row = outch (e&3) + 8*(e>>2) + 4*half), read from the LDS copy
  f32x16 acc_local[DEFER ? 1 : MT];
  f32x16 (&acc)[MT] = *reinterpret_cast<f32x16 (*)[MT]>(DEFER ? &out->a[0] : &acc_local[0]);
  uint32_t prr[2][4][2];                            // packed halves of the pending epilogue (PENDING)
  {
    const float* bl = reinterpret_cast<const float*>(smem + OFF_B) + 32 * (K - 1) + 4 * c.half;
    f32x16 binit;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *reinterpret_cast<const float4*>(bl + 8 * g);
      binit[4 * g + 0] = b.x; binit[4 * g + 1] = b.y; binit[4 * g + 2] = b.z; binit[4 * g + 3] = b.w;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[m] = binit;
  }

  // per-lane X offsets of the two pixels (taps) each im2col48 k-step needs: tap = 2*(2ks+half)+{0,1}
  int xo[3][2];
#pragma unroll
  for (int ks = 0; ks < 3; ++ks)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int t0 = 4 * ks + e, t1 = 4 * ks + 2 + e;          // half 0 / half 1
      const int o0 = t0 < 9 ? ((t0 / 3) * XS + t0 % 3) * 8 : 0;
      const int o1 = t1 < 9 ? ((t1 / 3) * XS + t1 % 3) * 8 : 0;
      xo[ks][e] = c.half ? o1 : o0;
    }

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    STAMP(ts0);
    // ---- prefetch the chunk that follows (compile-time position in the stream)
    if (ch + 1 < NCH) {
      if (ch == 0) w_prefetch<LAYER_OFF[K] + 21, 18>(c);
      else w_prefetch<LAYER_OFF[K] + 39, 18>(c);
    } else {
      if (K == DEPTH) w_prefetch<LAYER_OFF[2], 21>(c);           // wraps to the next tile's conv2 (conv1's fragments are resident)
      else if (K == 1) { /* conv2's first chunk is already in place */ }
      else if (K == 2) w_prefetch<LAYER_OFF[3], 21>(c);
      else w_prefetch<LAYER_OFF[4], 21>(c);
    }
    const unsigned char* wb = (K == 1 ? smem + OFF_W1
                                      : smem + OFF_W + (WRES ? (LAYER_OFF[K] + (ch == 0 ? 0 : ch == 1 ? 21 : 39)) * 1024 : c.par * W_BYTES)) + c.lane * 16;
    const unsigned char* xb = smem + OFF_X + xbuf * X_BYTES;
    // The chunk is a flat list of MFMA steps: [3 im2col48 k-steps (chunk 0 only)] + [18 (tap, k-step)
    // steps of feature J = ch + 1 (K >= 2)].  One wave per SIMD has no other wave to hide LDS latency
    // behind, so operand fragments are fetched two steps ahead into a 3-deep register ring; the
    // sched_barriers pin "reads of step s+2, then MFMAs of step s".
    const int NIM = ch == 0 ? 3 : 0;
    const int NS = NIM + (K >= 2 ? 18 : 0);
    const int J = ch + 1;
    const int pitch = J == 1 ? P1 : J == 2 ? P2 : P3;
    const unsigned char* fb = smem + (J == 1 ? OFF_F1 : J == 2 ? OFF_F2 : OFF_F3);
    int pb[MT], xbase[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      pb[m] = (r[m] + K - J - 1) * pitch + (cc[m] + K - J - 1) * PS + c.half * 16;
      xbase[m] = ((r[m] + K - 1) * XS + (cc[m] + K - 1)) * 8;
    }
    // ring depth: fragments are fetched RD-1 steps ahead.  A step is MT MFMAs = MT x 32 cycles of matrix pipe per wave, so
    // conv4 (one M-tile per wave) needs a deeper ring than the others to cover the same LDS latency.
    constexpr int RD = MT == 1 ? RD1 : RD2;
    f16x8 ringA[RD];
    f16x8 ringB[RD][MT];
    auto load_step = [&](const int st, f16x8& A, f16x8 (&B)[MT]) __attribute__((always_inline)) {
      A = *reinterpret_cast<const f16x8*>(wb + st * 1024);
      if (st < NIM) {
#pragma unroll
        for (int m = 0; m < MTE; ++m) {
          // tap slots 9..11 carry ZERO weights (packing.pack_fused_gh): their lanes read tap 0 instead of branching around the
          // load - any finite value will do, and a branch per load split the step into basic blocks (no read / MFMA interleave)
          const uint2 p0 = *reinterpret_cast<const uint2*>(xb + xbase[m] + xo[st][0]);
          const uint2 p1 = *reinterpret_cast<const uint2*>(xb + xbase[m] + xo[st][1]);
          const u32x4 u = {p0.x, p0.y, p1.x, p1.y};
          B[m] = __builtin_bit_cast(f16x8, u);
        }
      } else {
        const int fs = st - NIM, tap = fs >> 1, ks = fs & 1;
#pragma unroll
        for (int m = 0; m < MTE; ++m)
          B[m] = *reinterpret_cast<const f16x8*>(fb + pb[m] + (tap / 3) * pitch + (tap % 3) * PS + ks * 32);
      }
    };
    STAMP(ts1);
    STAMP_ADD(0, ts0, ts1);
#pragma unroll
    for (int i = 0; i < RD - 1; ++i)
      if (i < NS) load_step(i, ringA[i], ringB[i]);
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      if (st + RD - 1 < NS) load_step(st + RD - 1, ringA[(st + RD - 1) % RD], ringB[(st + RD - 1) % RD]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < MTE; ++m) {
        acc[m] = mfma_32x32x16(ringA[st % RD], ringB[st % RD][m], acc[m]);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the pending epilogue of conv K-1: one piece (ten-odd VALU instructions, or one M-tile's stores) per step
      // (packs behind steps 1..8; then this chunk's weight hand-over, so that no wait on the VM counter follows the
      // feature stores; then the two M-tiles' stores behind steps 9 and 10)
      // (a wave without a second tile in conv K-1 skips that tile's four pack pieces and its store piece)
      if (PENDING && ch == 0 && st >= 1 && st <= 10 && (PREV2 || st <= 4 || st == 9))
        epi_piece<(PENDING ? K - 1 : 1), 2>(c, a, net, fofs, ty0, tx0, prev->a, prr, st - 1);
      if (PENDING && ch == 0 && st == 8) w_commit<LAYER_OFF[K] + 21, 18>(c);
    }

    STAMP(ts2);
    STAMP_LOOP(K, ts1, ts2);
    // ---- hand the weight buffers over FIRST (next chunk -> the other buffer): global stores share the VM counter with
    // loads on gfx9, so waiting for the prefetched fragments behind the epilogue's feature stores meant waiting for
    // those stores to be acknowledged (~1-2 us at every conv boundary)
    if (PENDING && ch == 0) { /* handed over behind step 8 */ }
    else if (ch + 1 < NCH) { if (ch == 0) w_commit<LAYER_OFF[K] + 21, 18>(c); else w_commit<LAYER_OFF[K] + 39, 18>(c); }
    else if (K == DEPTH) w_commit<LAYER_OFF[2], 21>(c);
    else if (K == 1) { /* nothing streamed during conv1 */ }
    else w_commit<LAYER_OFF[K < 4 ? K + 1 : 2], 21>(c);
    STAMP(ts3);
    STAMP_ADD(3, ts2, ts3);
    if (ch == NCH - 1 && !DEFER) {
      // ---- epilogue now (conv1: its output is read by conv2's very first feature step; conv4: end of the tile)
      uint32_t rr[MT][4][2];
#pragma unroll
      for (int p = 0; p < 5 * MT; ++p) epi_piece<K, MT>(c, a, net, fofs, ty0, tx0, acc, rr, p);
    }
    if (K == DEPTH && ch == NCH - 1) return;   // the tile loop has the tile's last barrier
    __syncthreads();
    if (K != 1) c.par ^= 1;                    // conv1 consumed no streamed buffer
    STAMP(ts4);
    STAMP_ADD(2, ts3, ts4);
  }
}

__global__ __launch_bounds__(NTHR) void fused_gh_kernel(const FGArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Ctx c;
  c.smem = smem;
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave = c.tid >> 6;
  c.half = c.lane >> 5;
#ifdef SELFC_CLOCKS
  ClockProbe ckp;
  clock_probe_begin(ckp);
#endif
  c.par = 0;
  c.first = true;
#ifdef SELFC_STAMPS
  c.phase[0] = c.phase[1] = c.phase[2] = c.phase[3] = 0;
  STAMP(tk0);
#endif
  const int net = blockIdx.y;
  c.wsrc = reinterpret_cast<const u32x4*>(net ? a.w[1] : a.w[0]);
  {
    // Opaque to hipcc: under SGPR pressure it re-loaded a.dense[net] from the kernel-argument segment (s_load_dwordx2) in
    // front of EVERY feature store, and a scalar load can only be awaited with lgkmcnt(0) - which also drains the wave's
    // LDS operand ring (seven drains per tile).  Laundered, the pointer lives in SGPRs for the whole kernel.  (ISA clean-up;
    // measured time unchanged)
    unsigned long long d = reinterpret_cast<unsigned long long>(net ? a.dense[1] : a.dense[0]);
    asm volatile("" : "+s"(d));
    c.dense = (gf16*)d;
  }
  // Frame walk: a workgroup owns ONE spatial tile (blockIdx.x % ntiles) and visits frames f0, f0 + gf, ...: everything
  // that depends on the tile position (halo offsets, border masks, output offsets) is a per-workgroup constant that the
  // compiler keeps in registers instead of re-deriving ~400 VALU instructions' worth of it per tile.
  const int stile = blockIdx.x % a.ntiles, f0 = blockIdx.x / a.ntiles, gf = gridDim.x / a.ntiles;
  if (f0 >= a.N) return;
  const int ty0 = (stile / a.tiles_x) * TS, tx0 = (stile % a.tiles_x) * TS;
  const size_t fpix = (size_t)a.H * a.W;

  constexpr int XITER = (XS * XS + NTHR - 1) / NTHR;   // 2
  float4 xv[XITER];
  unsigned xgo[XITER];
  unsigned xok = 0;   // bit it: halo pixel it is inside the image (mask applied at store time, not after the load)
#pragma unroll
  for (int it = 0; it < XITER; ++it) {
    const int p = min(c.tid + it * NTHR, XS * XS - 1);
    const int hy = p / XS, hx = p - hy * XS;
    const int y = ty0 + hy - DEPTH, x = tx0 + hx - DEPTH;
    const bool ok = (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
    const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
    xgo[it] = (unsigned)(yc * a.W + xc) * 4u;
    xok |= (ok ? 1u : 0u) << it;
  }
  auto x_load = [&](const int n) __attribute__((always_inline)) {
    const float* fr = a.x1 + (size_t)n * fpix * 4;
#pragma unroll
    for (int it = 0; it < XITER; ++it) xv[it] = *reinterpret_cast<const float4*>(fr + xgo[it]);
  };
  auto x_store = [&](const int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < XITER; ++it) {
      const int p = c.tid + it * NTHR;
      if (p < XS * XS) {
        uint2 u;
        u.x = pack2(xv[it].x, xv[it].y);
        u.y = pack2(xv[it].z, 0.f);
        if (!((xok >> it) & 1u)) u = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(smem + OFF_X + buf * X_BYTES + p * 8) = u;
      }
    }
  };

  // prologue: biases -> LDS, first tile's X halo and the first weight chunk
  if (c.tid < 128) {
    const float* bsrc = net ? a.bias[1][c.tid >> 5] : a.bias[0][c.tid >> 5];
    reinterpret_cast<float*>(smem + OFF_B)[c.tid] = bsrc[c.tid & 31];
  }
  x_load(f0);
  w_prefetch<LAYER_OFF[2], 21>(c);
  if (c.tid < 3 * 64) *reinterpret_cast<u32x4*>(smem + OFF_W1 + c.tid * 16) = c.wsrc[LAYER_OFF[1] * 64 + c.tid];   // conv1: resident
  x_store(0);
  c.par = 1;            // w_commit writes buffer par^1 = 0
  c.first = true;
  w_commit<LAYER_OFF[2], 21>(c);
  __syncthreads();
  c.par = 0;

  auto tile_loop = [&](auto two2c, auto two3c) __attribute__((always_inline)) {
  constexpr bool TWO2 = decltype(two2c)::value, TWO3 = decltype(two3c)::value;
  int xbuf = 0;
  for (int n = f0; n < a.N; n += gf) {
    const bool more = n + gf < a.N;
    if (more) x_load(n + gf);                     // lands while this tile computes
    const size_t fofs = (size_t)n * fpix * 32;
    AccPair acc2, acc3;
    conv_fused<1>(c, a, net, xbuf, fofs, ty0, tx0, nullptr, nullptr);
    conv_fused<2, TWO2>(c, a, net, xbuf, fofs, ty0, tx0, nullptr, &acc2);
    if (DEPTH == 3) {
      if (more) x_store(xbuf ^ 1);                // before the last conv's feature stores (see conv_fused: one VM counter)
      conv_fused<3>(c, a, net, xbuf, fofs, ty0, tx0, &acc2, nullptr);
    } else {
      conv_fused<3, TWO3, TWO2>(c, a, net, xbuf, fofs, ty0, tx0, &acc2, &acc3);
      if (more) x_store(xbuf ^ 1);
      conv_fused<(DEPTH == 4 ? 4 : 3), true, TWO3>(c, a, net, xbuf, fofs, ty0, tx0, &acc3, nullptr);
    }
    STAMP(tt0);
    __syncthreads();            // (round 5: a build without this barrier - no hazard needs it, conv1's own barrier follows - measured 0: ab_experiments.txt)
    c.par ^= 1;
    c.first = false;
    xbuf ^= 1;
    STAMP(tt1);
    STAMP_ADD(3, tt0, tt1);
  }
  };
  // second M-tile owners (mtile_geom): conv3 waves 0..2, conv2 waves 3..7
  static_assert(DEPTH != 4 || (ConvGeom<3>::NTL - NWAVE == 3 && ConvGeom<2>::NTL - NWAVE == 5 && second_tile_first_wave<2>() == 3),
                "the two wave classes below");
  if (c.wave < 3) tile_loop(std::false_type{}, std::true_type{});
  else tile_loop(std::true_type{}, std::false_type{});
#ifdef SELFC_CLOCKS
  clock_probe_end(ckp, a.stamps, blockIdx.x == 0 && blockIdx.y == 0 && c.tid == 0);
#endif
#ifdef SELFC_STAMPS
  STAMP(tk1);
  if (a.stamps && c.lane == 0) {
    unsigned long long* o = a.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * NWAVE + c.wave) * 5;
    o[0] = c.phase[0]; o[1] = c.phase[1]; o[2] = c.phase[2]; o[3] = c.phase[3]; o[4] = tk1 - tk0;
  }
#endif
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

}  // namespace

// Called from dense_conv.hip when the block carries fused fragment streams: run_GH with nets = 2 (G and H of a coupling block),
// selfc_subnet_run with nets = 1 (a stand-alone 3-channel subnet: the first subnet of the STP chain, SelfC_GMM_arch_inv.py:305).
int launch_fused_gh(FGArgs& a, hipStream_t s, int nets) {
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&fused_gh_kernel), FG_LDS, optin); e != hipSuccess) return hip_rc(e);
  a.tiles_x = (a.W + TS - 1) / TS;
  a.tiles_y = (a.H + TS - 1) / TS;
  a.ntiles = a.tiles_x * a.tiles_y;
  a.plane = (size_t)a.N * a.H * a.W * 32;
  // persistent workgroups, about 128 per net (one per CU overall): gfr workgroups per spatial tile, each walking
  // frames f0, f0 + gfr, ... - sized so that every workgroup walks (nearly) the same number of frames (no straggler
  // round) and unused CUs stay free for kernels of other streams
  static const int maxwg2 = getenv("SELFC_FUSEDGH_MAXWG") ? atoi(getenv("SELFC_FUSEDGH_MAXWG")) : 128;
  const int maxwg = nets == 2 ? maxwg2 : 2 * maxwg2;
  const int gmax = a.ntiles >= maxwg ? 1 : (maxwg / a.ntiles < a.N ? maxwg / a.ntiles : a.N);
  static const int minrounds = getenv("SELFC_FUSEDGH_MINROUNDS") ? atoi(getenv("SELFC_FUSEDGH_MINROUNDS")) : 2;
  int rounds = (a.N + gmax - 1) / gmax;
  if (rounds < minrounds && (long)a.N * a.ntiles > maxwg) rounds = a.N < minrounds ? a.N : minrounds;   // (one round of single tile-frames stays: see fused_f16.hip)
  const int gfr = (a.N + rounds - 1) / rounds;
  const int gx = gfr * a.ntiles;
#ifdef SELFC_STAMPS
  static unsigned long long* dbg = nullptr;
  if (!dbg) (void)hipMalloc(&dbg, 256 * NWAVE * 5 * sizeof(unsigned long long));
  a.stamps = nets * gx <= 256 ? dbg : nullptr;
  if (getenv("SELFC_STAMP_DUMP")) {      // diagnostic: dump the previous launch's sums, then continue
    static unsigned long long host[256 * NWAVE * 5];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
    FILE* f = fopen(getenv("SELFC_STAMP_DUMP"), "w");
    if (f) {
      for (int i = 0; i < 256 * NWAVE; ++i)
        fprintf(f, "%llu %llu %llu %llu %llu\n", host[5 * i], host[5 * i + 1], host[5 * i + 2], host[5 * i + 3], host[5 * i + 4]);
      fclose(f);
    }
  }
#endif
#ifdef SELFC_CLOCKS
  a.stamps = clock_probe_slot(2);
#endif
  ProfScope prof(PROF_FUSED_GH, s);
  hipLaunchKernelGGL(fused_gh_kernel, dim3((unsigned)gx, (unsigned)nets), dim3(NTHR), FG_LDS, s, a);
  return hip_rc(hipGetLastError());
}

}  // namespace selfc
