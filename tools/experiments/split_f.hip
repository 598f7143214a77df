// conv1..conv4 of the F dense block (D2DTInput / DenseBlock with cin = 48, Subnet_constructor.py:27-30,126-129) as FOUR
// persistent single-conv launches for gfx950 - the un-fused alternative to csrc/fused_f.hip (selected per call by
// launch_split_f).
//
// One 512-thread workgroup per CU owns ONE 16x16 spatial tile and walks frames.  Per launch ONE conv: its whole fragment
// stream (27 / 45 / 63 / 81 KiB) is RESIDENT in LDS, the 18x18 input halo of all its input channels sits in an LDS image
// (16 k-channels = 32 B pieces, pixel pitch an odd number of 16-byte slots), a wave owns one M-tile of 2 rows x 16 columns
// and runs one MFMA per k-step: no ring tiles, no weight streaming, no barrier except where a source group's part of the image
// dies and the next frame's part is stored over it.  conv4's 144 input channels do not fit next to its 81 KiB of weights, so it
// runs as two phases over the same image region: [x2 | f1], then [f2 | f3].  conv4 also takes F's temporal conv5 as partial
// products (all 11 fragments: x2, f1..f3 from the images' centre taps, f4 from its own registers) - f_couple_kernel sums one
// set instead of two.
//
// Against the pair kernels this trades HBM bytes for cycles: every conv re-reads its inputs (1,215 B per pixel-frame instead of
// 777), but nothing is computed twice (no 18x18 halo recompute), nothing is re-streamed per tile, and the K loop is one MFMA
// and two fragment reads per wave and step.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include "common.hpp"
#include "prof.hpp"
#include "../../include/selfc_hip.h"

using namespace selfc;

namespace selfc {

struct SFArgs {
  f16* dense;               // F dense buffer, plane-blocked [6][N][H][W][32]: x2 (2 planes), f1..f4
  const f16* w;             // this conv's fragment stream (packing.py: pack_split_f)
  const float* bias;        // 32 floats
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
  size_t plane;
  const f16* w5p;           // conv4 only, optional: the 11 conv5 partial-product fragments (packing.py: pack_f5_partial)
  float* pf;                // conv4 only, optional: partial products, fp32 [3 taps][N][H][W][4]
  int store_feat;           // 0: conv4 on the inference path - nothing reads f4 afterwards
};

namespace {

constexpr int TS = 16, IS = 18;
constexpr int NWAVE = 8, NTHR = NWAVE * 64;

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// NKA / NKB: 16-channel k-steps per pixel of the phase A / phase B image (NKB = 0: one phase).  conv1 <3,0>, conv2 <5,0>,
// conv3 <7,0>, conv4 <5,4>.  Both phases share the pixel pitch of phase A (one lane <-> pixel map for the accumulator).
template <int NKA, int NKB>
struct SG {
  static constexpr int NK = NKA;
  static constexpr int PITCH = NKA * 32 + 16;                                    // odd number of 16-byte slots
  static constexpr int KAPPA = NKA == 3 ? 7 : NKA == 5 ? 3 : NKA == 7 ? 15 : 11;   // (PITCH/16)^-1 mod 16
  static constexpr int ROWP = NKA == 3 ? 2064 : NKA == 5 ? 3344 : NKA == 7 ? 4368 : 5648;   // >= IS*PITCH, slots = 1 mod 16
  static constexpr int IMG_BYTES = IS * ROWP;
  static constexpr int SA = 9 * NKA, SB = 9 * NKB, S = SA + SB;                  // k-steps
  static constexpr int W_BYTES = S * 1024;
  static constexpr int NGA = 1 + (NKA - 3) / 2;                                  // source groups of phase A: x2, then features
  static constexpr int NGB = NKB / 2;
  static constexpr bool LAST = NKB > 0;                                          // conv4: conv5 partial products
  static constexpr int OUT = NKB > 0 ? 5 : 1 + NGA;                              // plane this conv appends (f_k = plane 1 + k)
  static constexpr int OFF_IMG = 0, OFF_W = IMG_BYTES, OFF_B = OFF_W + W_BYTES, OFF_P = OFF_B + 128;
  static constexpr int LDS = OFF_P + (LAST ? 11 * 1024 : 0);
  static_assert(ROWP >= IS * PITCH && (ROWP / 16) % 16 == 1, "image row pitch");
  static_assert(((PITCH / 16) * KAPPA) % 16 == 1, "kappa");
  static_assert(LDS <= 160 * 1024, "LDS budget");
  // first k-step and length (in k-steps) of source group g of a phase: x2 has 3 k-steps per tap, a feature 2
  static constexpr int gstart(bool phaseB, int g) { return phaseB ? 18 * g : (g == 0 ? 0 : 27 + 18 * (g - 1)); }
  static constexpr int glen(bool phaseB, int g) { return (!phaseB && g == 0) ? 27 : 18; }
  static constexpr int gitems(bool phaseB, int g) { return (!phaseB && g == 0) ? 5 : 3; }    // 16-byte pieces per thread
};

__device__ __forceinline__ f32x16 bias_init(const float* bl, const int half) {
  f32x16 b;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(bl + 8 * g + 4 * half);
    b[4 * g + 0] = v.x; b[4 * g + 1] = v.y; b[4 * g + 2] = v.z; b[4 * g + 3] = v.w;
  }
  return b;
}

// bias is already in the accumulator: LeakyReLU, f16, half-swap -> v[gp] = 8 contiguous channels 16 gp + 8 half ..
__device__ __forceinline__ void lrelu_pack(const f32x16& acc, u32x4 (&v)[2]) {
  uint32_t r[4][2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    r[g][0] = lrelu_pack2(acc[4 * g + 0], acc[4 * g + 1]);
    r[g][1] = lrelu_pack2(acc[4 * g + 2], acc[4 * g + 3]);
  }
#pragma unroll
  for (int gp = 0; gp < 2; ++gp)
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const auto sw = __builtin_amdgcn_permlane32_swap(r[2 * gp][d], r[2 * gp + 1][d], false, false);
      v[gp][d] = sw[0];
      v[gp][2 + d] = sw[1];
    }
}

template <int NKA, int NKB>
__global__ __launch_bounds__(NTHR) void split_f_kernel(const SFArgs a) {
  using G = SG<NKA, NKB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
  int py, px;
  {
    const int i = lane & 15, row2 = (lane >> 4) & 1;
    py = 2 * wave + row2;
    px = row2 ? ((i - G::KAPPA) & 15) : i;
  }
  const int stile = blockIdx.x % a.ntiles, f0 = blockIdx.x / a.ntiles, gf = gridDim.x / a.ntiles;
  if (f0 >= a.N) return;
  const int ty0 = (stile / a.tiles_x) * TS, tx0 = (stile % a.tiles_x) * TS;

  // ---- halo geometry (18x18 pixels): per-workgroup constants.  map A (whole 64-byte planes): piece i = tid + 512 it (it < 3,
  // i < 1296) = 16-byte piece i & 3 of halo pixel i >> 2; map B (x2 channels 32..47, 32 bytes of plane 1): i = tid + 512 it
  // (it < 2, i < 648) = piece i & 1 of pixel i >> 1.  goff: byte offset inside one frame of a plane; loff: offset of the pixel
  // in the LDS image; ok bit: the piece exists and its pixel lies inside the frame (pieces outside are never stored: the
  // image is zeroed once and those bytes stay the convs' zero padding).
  unsigned goffA[3], loffA[3], goffB[2], loffB[2], okA = 0, okB = 0;
  {
    auto geom = [&](const int pix, const bool exists, unsigned& goff, unsigned& loff) __attribute__((always_inline)) {
      const int p = min(pix, IS * IS - 1);
      const int hy = p / IS, hx = p - hy * IS;
      const int y = ty0 + hy - 1, x = tx0 + hx - 1;
      const bool ok = exists & (y >= 0) & (y < a.H) & (x >= 0) & (x < a.W);
      const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
      goff = (unsigned)(yc * a.W + xc) * 64u;
      loff = (unsigned)(hy * G::ROWP + hx * G::PITCH);
      return ok;
    };
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int i = tid + it * NTHR;
      const bool ok = geom(i >> 2, i < IS * IS * 4, goffA[it], loffA[it]);
      goffA[it] += (i & 3) * 16;
      loffA[it] += (i & 3) * 16;
      okA |= (ok ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i = tid + it * NTHR;
      const bool ok = geom(i >> 1, i < IS * IS * 2, goffB[it], loffB[it]);
      goffB[it] += (i & 1) * 16;
      loffB[it] += (i & 1) * 16 + 64;
      okB |= (ok ? 1u : 0u) << it;
    }
  }
  const size_t frame_bytes = (size_t)a.H * a.W * 64, plane_bytes = a.plane * 2;
  const char* const dbase = reinterpret_cast<const char*>(a.dense);
  // source group g of a phase: phase A g = 0 is x2 (planes 0, 1), g >= 1 feature g (plane 1 + g); phase B g = 0, 1 are f2, f3
  // (planes 3, 4).  item it of a group: x2 items 0..2 map A on plane 0, items 3, 4 map B on plane 1; a feature: items 0..2 map A.
  // items in flight: ALL pieces of the data that replaces an image are requested in the first steps of a tile (one per step) and
  // stored group by group as the groups die - a frame's pieces need 1-3 us to arrive when every workgroup asks at once, more than
  // one source group's 18 k-steps last.  One-phase conv k: 5 + 3 (k - 1) pieces; conv4: 6 (its phase-B image) + 8 (the next
  // frame's phase-A image).
  constexpr int XMAX = NKB > 0 ? 14 : 5 + 3 * (G::NGA - 1);
  u32x4 xv[XMAX];
  auto item_load = [&](const char* frame, const bool phaseB, const int g, const int it, const int slot) __attribute__((always_inline)) {
    if (!phaseB && g == 0) {
      xv[slot] = it < 3 ? *reinterpret_cast<const u32x4*>(frame + goffA[it])
                        : *reinterpret_cast<const u32x4*>(frame + plane_bytes + goffB[it - 3]);
    } else {
      const int pl = phaseB ? 3 + g : 1 + g;
      xv[slot] = *reinterpret_cast<const u32x4*>(frame + (size_t)pl * plane_bytes + goffA[it]);
    }
  };
  // first: the very first image of the workgroup - pieces outside the frame are stored once, as zeros (the convs' padding), and
  // never again: the set is a per-workgroup constant, so no zero fill of the image is needed
  auto item_store = [&](const bool phaseB, const int g, const int it, const int slot, const bool first = false) __attribute__((always_inline)) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    if (!phaseB && g == 0) {
      if (it < 3) {
        const bool ok = (okA >> it) & 1u;
        if (ok | (first & ((tid + it * NTHR) < IS * IS * 4))) *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + loffA[it]) = ok ? xv[slot] : z;
      } else {
        const bool ok = (okB >> (it - 3)) & 1u;
        if (ok | (first & ((tid + (it - 3) * NTHR) < IS * IS * 2))) *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + loffB[it - 3]) = ok ? xv[slot] : z;
      }
    } else {
      const int goff_px = phaseB ? 64 * g : 96 + 64 * (g - 1);
      const bool ok = (okA >> it) & 1u;
      if (ok | (first & ((tid + it * NTHR) < IS * IS * 4))) *reinterpret_cast<u32x4*>(smem + G::OFF_IMG + loffA[it] + goff_px) = ok ? xv[slot] : z;
    }
  };
  // register slot of item `it` of group g of a phase-A image / of conv4's phase-B image
  auto slot_a = [](const int g, const int it) { return (g == 0 ? 0 : 5 + 3 * (g - 1)) + it; };

  // ---- prologue: the first image's pieces are requested first; under that round trip the weights become resident, the bias and
  // (conv4) the conv5 partial-product fragments land; then the image is stored (zeros where it lies outside the frame) ----
  {
    const char* fr = dbase + (size_t)f0 * frame_bytes;
    static_for<0, G::NGA>([&](auto gi) __attribute__((always_inline)) {
      constexpr int g = decltype(gi)::value;
      static_for<0, G::gitems(false, g)>([&](auto ii) __attribute__((always_inline)) { item_load(fr, false, g, decltype(ii)::value, slot_a(g, decltype(ii)::value)); });
    });
  }
  {
    const u32x4* __restrict__ wsrc = reinterpret_cast<const u32x4*>(a.w);
    for (int i = tid; i < G::S * 64; i += NTHR) *reinterpret_cast<u32x4*>(smem + G::OFF_W + i * 16) = wsrc[i];
  }
  if (tid < 32) reinterpret_cast<float*>(smem + G::OFF_B)[tid] = a.bias[tid];
  const bool do_p = G::LAST && a.w5p != nullptr && a.pf != nullptr;
  if (G::LAST && do_p) {
    const u32x4* __restrict__ psrc = reinterpret_cast<const u32x4*>(a.w5p);
    for (int i = tid; i < 11 * 64; i += NTHR) *reinterpret_cast<u32x4*>(smem + G::OFF_P + i * 16) = psrc[i];
  }
  static_for<0, G::NGA>([&](auto gi) __attribute__((always_inline)) {
    constexpr int g = decltype(gi)::value;
    static_for<0, G::gitems(false, g)>([&](auto ii) __attribute__((always_inline)) { item_store(false, g, decltype(ii)::value, slot_a(g, decltype(ii)::value), true); });
  });
  __syncthreads();

  const unsigned char* const img = smem + G::OFF_IMG;
  const unsigned char* const bc = img + py * G::ROWP + px * G::PITCH + half * 16;      // image pixel (0,0) = tile pixel (-1,-1)
  const unsigned char* const wl = smem + G::OFF_W + lane * 16;
  const float* const lb = reinterpret_cast<const float*>(smem + G::OFF_B);
  const unsigned char* const pfr = smem + G::OFF_P + lane * 16;

  for (int n = f0; n < a.N; n += gf) {
    const bool more = n + gf < a.N;
    const char* const nfr = dbase + (size_t)(n + gf) * frame_bytes;      // only dereferenced when `more`
    f32x16 acc = bias_init(lb, half);
    f32x16 accp;
    if (G::LAST) {
#pragma unroll
      for (int e = 0; e < 16; ++e) accp[e] = 0.f;
    }

    // one phase: k-steps [S0, S0 + NS) over the source groups of that phase's image, one MFMA per step.  Hooks hang ONE memory
    // operation behind a step's MFMA: the pieces that will replace an image are all requested behind the first steps of the tile;
    // in a one-phase conv the barrier sits behind the LAST step of a group (its part of the image is dead for every wave) and the
    // group's new pieces are stored behind the first steps of the next group; the last group's pieces (and conv4's whole images) are
    // stored by the caller between two barriers.
    auto phase = [&](auto phase_b_tag, const char* ldframe, const char* ldframe2, const bool ld_on) __attribute__((always_inline)) {
      constexpr bool PB = decltype(phase_b_tag)::value;          // which image this phase reads
      constexpr int NS = PB ? G::SB : G::SA, S0 = PB ? G::SA : 0, NG = PB ? G::NGB : G::NGA;
      f16x8 rA[3], rB[3];
      auto load_step = [&](const int st) __attribute__((always_inline)) {
        // group-major, tap-major inside a group, k-step minor
        int g = 0, s = st;
        if (!PB) { if (st >= 27) { g = 1 + (st - 27) / 18; s = (st - 27) % 18; } }
        else { g = st / 18; s = st % 18; }
        const int kpt = (!PB && g == 0) ? 3 : 2;
        const int tap = s / kpt, ks = s % kpt + (PB ? 2 * g : (g == 0 ? 0 : 3 + 2 * (g - 1)));
        rA[st % 3] = *reinterpret_cast<const f16x8*>(wl + (S0 + st) * 1024);
        rB[st % 3] = *reinterpret_cast<const f16x8*>(bc + (tap / 3) * G::ROWP + (tap % 3) * G::PITCH + ks * 32);
      };
      load_step(0);
      load_step(1);
      static_for<0, NS>([&](auto si) __attribute__((always_inline)) {
        constexpr int st = decltype(si)::value;
        if constexpr (st + 2 < NS) load_step(st + 2);
        __builtin_amdgcn_sched_barrier(0);
        acc = mfma_32x32x16(rA[st % 3], rB[st % 3], acc);
        __builtin_amdgcn_sched_barrier(0);
        // ---- hooks ----
        constexpr int g = PB ? st / 18 : (st < 27 ? 0 : 1 + (st - 27) / 18);      // group of THIS phase step st belongs to
        constexpr int s = st - G::gstart(PB, g);
        if constexpr (NKB == 0) {
          // loads: piece j of the next frame's image behind step j (all groups, first steps of the tile)
          if constexpr (st < XMAX) {
            constexpr int lg = st < 5 ? 0 : 1 + (st - 5) / 3, li = st < 5 ? st : (st - 5) % 3;
            if (ld_on) item_load(ldframe, false, lg, li, st);
          }
          // group g's part of the image dies behind its last step (barrier); its new pieces go in behind the next group's first steps
          if constexpr (s == G::glen(PB, g) - 1 && g < NG - 1) __syncthreads();
          if constexpr (g >= 1 && s < G::gitems(false, g - 1)) {
            if (ld_on) item_store(false, g - 1, s, slot_a(g - 1, s));
          }
        } else if constexpr (!PB) {
          // conv4, phase A: this frame's phase-B image (slots 0..5), then the next frame's phase-A image (slots 6..13)
          if constexpr (st < 6) item_load(ldframe, true, st / 3, st % 3, st);
          else if constexpr (st < 14) {
            constexpr int j = st - 6, lg = j < 5 ? 0 : 1, li = j < 5 ? j : j - 5;
            if (ld_on) item_load(ldframe2, false, lg, li, st);
          }
        }
      });
    };

    if constexpr (NKB == 0) {
      phase(std::false_type{}, nfr, nfr, more);
      u32x4 v[2];
      lrelu_pack(acc, v);
      {
        const int y = ty0 + py, x = tx0 + px;
        if ((y < a.H) & (x < a.W) & (a.store_feat != 0)) {
          f16* d = a.dense + (size_t)G::OUT * a.plane + ((size_t)(n * a.H + y) * a.W + x) * 32 + 8 * half;
          *reinterpret_cast<u32x4*>(d) = v[0];
          *reinterpret_cast<u32x4*>(d + 16) = v[1];
        }
      }
      __syncthreads();                                   // the last group's part of the image is dead
      if (more) {
        constexpr int g = G::NGA - 1;
        static_for<0, G::gitems(false, g)>([&](auto ii) __attribute__((always_inline)) {
          item_store(false, g, decltype(ii)::value, slot_a(g, decltype(ii)::value));
        });
      }
      __syncthreads();                                   // the next frame's image is complete
    } else {
      // conv4.  phase A reads [x2 | f1] while this frame's [f2 | f3] arrive in registers
      phase(std::false_type{}, dbase + (size_t)n * frame_bytes, nfr, more);
      if (do_p) {                                        // conv5 partial products of x2, f1: the image's centre tap
#pragma unroll
        for (int ks = 0; ks < 5; ++ks)
          accp = mfma_32x32x16(*reinterpret_cast<const f16x8*>(pfr + ks * 1024),
                               *reinterpret_cast<const f16x8*>(bc + G::ROWP + G::PITCH + ks * 32), accp);
      }
      __syncthreads();                                   // image A is dead
      static_for<0, 2>([&](auto gi) __attribute__((always_inline)) {
        static_for<0, 3>([&](auto ii) __attribute__((always_inline)) {
          item_store(true, decltype(gi)::value, decltype(ii)::value, 3 * decltype(gi)::value + decltype(ii)::value);
        });
      });
      __syncthreads();                                   // image B is complete
      // phase B reads [f2 | f3] while the next frame's [x2 | f1] arrive
      phase(std::true_type{}, nfr, nfr, more);
      u32x4 v[2];
      lrelu_pack(acc, v);
      if (do_p) {                                        // f2, f3 from image B's centre tap, f4 from the packed registers
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          accp = mfma_32x32x16(*reinterpret_cast<const f16x8*>(pfr + (5 + ks) * 1024),
                               *reinterpret_cast<const f16x8*>(bc + G::ROWP + G::PITCH + ks * 32), accp);
        accp = mfma_32x32x16(*reinterpret_cast<const f16x8*>(pfr + 9 * 1024), __builtin_bit_cast(f16x8, v[0]), accp);
        accp = mfma_32x32x16(*reinterpret_cast<const f16x8*>(pfr + 10 * 1024), __builtin_bit_cast(f16x8, v[1]), accp);
      }
      {
        const int y = ty0 + py, x = tx0 + px;
        if ((y < a.H) & (x < a.W)) {
          if (a.store_feat != 0) {
            f16* d = a.dense + (size_t)G::OUT * a.plane + ((size_t)(n * a.H + y) * a.W + x) * 32 + 8 * half;
            *reinterpret_cast<u32x4*>(d) = v[0];
            *reinterpret_cast<u32x4*>(d + 16) = v[1];
          }
          if (do_p) {
            // accumulator rows (e & 3) + 8 (e >> 2) + 4 half, fragment rows 4 tap + oc: half 0 owns taps 0 (e 0-3) and 2 (e 4-7),
            // half 1 tap 1 (e 0-3); one float4 (3 outputs + a zero) per pixel and tap: pf[tap][N][H][W][4]
            const size_t pix = (size_t)(n * a.H + y) * a.W + x, tapsz = (size_t)a.N * a.H * a.W * 4;
            *reinterpret_cast<float4*>(a.pf + (half ? tapsz : 0) + pix * 4) = make_float4(accp[0], accp[1], accp[2], accp[3]);
            if (half == 0) *reinterpret_cast<float4*>(a.pf + 2 * tapsz + pix * 4) = make_float4(accp[4], accp[5], accp[6], accp[7]);
          }
        }
      }
      __syncthreads();                                   // image B is dead
      if (more) {
        static_for<0, 5>([&](auto ii) __attribute__((always_inline)) { item_store(false, 0, decltype(ii)::value, 6 + decltype(ii)::value); });
        static_for<0, 3>([&](auto ii) __attribute__((always_inline)) { item_store(false, 1, decltype(ii)::value, 11 + decltype(ii)::value); });
      }
      __syncthreads();                                   // the next frame's image A is complete
    }
  }
}

inline int hip_rc(hipError_t e) { return e == hipSuccess ? SELFC_OK : -(int)e - 1000; }

template <int NKA, int NKB>
int launch_one(SFArgs& a, int maxwg, hipStream_t s) {
  using G = SG<NKA, NKB>;
  static std::atomic<unsigned long long> optin{0};
  if (hipError_t e = lds_optin(reinterpret_cast<const void*>(&split_f_kernel<NKA, NKB>), G::LDS, optin); e != hipSuccess) return hip_rc(e);
  // frame walk as in fused_f.hip: gfr workgroups per spatial tile, every one walking (nearly) the same number of frames
  const int gmax = a.ntiles >= maxwg ? 1 : (maxwg / a.ntiles < a.N ? maxwg / a.ntiles : a.N);
  static const int minrounds = getenv("SELFC_SPLITF_MINROUNDS") ? atoi(getenv("SELFC_SPLITF_MINROUNDS")) : 3;
  int rounds = (a.N + gmax - 1) / gmax;
  if (rounds < minrounds) rounds = a.N < minrounds ? a.N : minrounds;
  const int gfr = (a.N + rounds - 1) / rounds;
  hipLaunchKernelGGL((split_f_kernel<NKA, NKB>), dim3((unsigned)(gfr * a.ntiles)), dim3(NTHR), G::LDS, s, a);
  return hip_rc(hipGetLastError());
}

}  // namespace

void launch_f_couple(const float* pf, const float* b5, float* x1, int N, int T, int HW, float sgn, int nsets, hipStream_t s);   // fused_f.hip

// conv1..conv4 of F (cin = 48) on its dense buffer as four single-conv launches.  w = [conv1: 27][conv2: 45][conv3: 63]
// [conv4: 81] fragments (packing.py: pack_split_f).  With w5p (11 partial-product fragments), pf and x1 the temporal conv5 +
// coupling y1 = x1 +- F is done here as well (returns 1: conv5 handled).
int launch_split_f(void* dense, const void* w, const float* const* bias, int N, int H, int W, hipStream_t s,
                   const void* w5p, float* pf, const float* b5, float* x1, int T, int rev, int keep_features) {
  static const int maxwg = getenv("SELFC_SPLITF_MAXWG") ? atoi(getenv("SELFC_SPLITF_MAXWG")) : 256;
  const bool with_p = w5p && pf && b5 && x1 && T > 0;
  SFArgs a{};
  a.dense = (f16*)dense;
  a.N = N; a.H = H; a.W = W;
  a.tiles_x = (W + TS - 1) / TS;
  a.tiles_y = (H + TS - 1) / TS;
  a.ntiles = a.tiles_x * a.tiles_y;
  a.plane = (size_t)N * H * W * 32;
  const int mw = maxwg > 0 ? maxwg : 256;
  {
    ProfScope prof(PROF_CONV3X3, s);
    int rc;
    a.store_feat = 1;
    a.w = (const f16*)w; a.bias = bias[0];
    if ((rc = launch_one<3, 0>(a, mw, s))) return rc;
    a.w += (size_t)27 * 512; a.bias = bias[1];
    if ((rc = launch_one<5, 0>(a, mw, s))) return rc;
    a.w += (size_t)45 * 512; a.bias = bias[2];
    if ((rc = launch_one<7, 0>(a, mw, s))) return rc;
    a.w += (size_t)63 * 512; a.bias = bias[3];
    a.w5p = with_p ? (const f16*)w5p : nullptr;
    a.pf = with_p ? pf : nullptr;
    a.store_feat = (keep_features || !with_p) ? 1 : 0;
    if ((rc = launch_one<5, 4>(a, mw, s))) return rc;
    if (!with_p) return SELFC_OK;
  }
  ProfScope prof(PROF_CONV5_F, s);
  launch_f_couple(pf, b5, x1, N, T, H * W, rev ? -1.f : 1.f, 1, s);
  const int rc = hip_rc(hipGetLastError());
  return rc ? rc : 1;
}

}  // namespace selfc
