// The body of conv3x3_kernel / conv3x3_pair_kernel (csrc/dense_conv.hip), included INSIDE those __global__ functions with
//   template parameters TH, TW, NW, MT, EPI, GEN in scope, a constexpr bool ROWEPI, and the kernel argument struct named `a`.
// Not a function: as a __device__ function taking the arguments by reference (round 6, first version) the non-generic instantiations
// lost their register budget - conv3x3_kernel<16,16,4,2,EPI_LRELU,false> went from 134 to 178 VGPRs, i.e. from 3 to 2 waves per SIMD,
// and the STP's layer-wise convs from 77 to 90 us at 1080p (profiles/r6/ab_experiments.txt r6t).
  static_assert(TH * TW == NW * MT * 32, "tile must be covered by the waves' M-tiles");
  static_assert(TW % 16 == 0 && TH % 2 == 0, "M-tiles are 2 rows x 16 cols");
  constexpr int HWD = TW + 2, NPIX = (TH + 2) * HWD;
  constexpr int NT = NW * 64;
  // Row pitch of the LDS halo image, rounded to a whole 256-B bank row: a ds_read_b128 lane group
  // mixes pixels {0-3,12-15} of one tile row with {4-11} of the next; with the pitch a multiple of
  // 16 slots both rows see the same pixel->slot map (5*px mod 16) and the two sets are disjoint.
  constexpr int ROWB = ((HWD * PS + 255) / 256) * 256;
  constexpr int ACT_BYTES = (TH + 2) * ROWB;
  constexpr int AITER = (NPIX * 4 + NT - 1) / NT;
  constexpr int WITER = (18 * 64 + NT - 1) / NT;
  constexpr int NNETS = (EPI == EPI_GH) ? 2 : 1;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const lact = smem;
  unsigned char* const lw = smem + ACT_BYTES;
  C3STAMP(0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tx = wg % a.tiles_x;
  const int ty = (wg / a.tiles_x) % a.tiles_y;
  const int n = wg / (a.tiles_x * a.tiles_y);
  const int tx0 = tx * TW, ty0 = ty * TH;
  const int H = a.H, W = a.W;

  // this lane's pixel in each of the wave's M-tiles
  int pbase[MT];      // byte offset of the pixel's tap (0,0) in the LDS halo tile (+ k-half)
  int py[MT], px[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int mt = wave * MT + m;
    const int by = mt / (TW / 16), bx = mt % (TW / 16);
    py[m] = 2 * by + ((lane & 31) >> 4);
    px[m] = 16 * bx + (lane & 15);
    pbase[m] = py[m] * ROWB + px[m] * PS + (lane >> 5) * 16;
  }

  f32x16 acc[NNETS][MT];
#pragma unroll
  for (int q = 0; q < NNETS; ++q)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[q][m][r] = 0.f;

  // conv1-4 epilogue biases: copied to LDS now (no global latency in the epilogue, no long-lived registers -
  // holding them in VGPRs cost the kernel its third wave per SIMD)
  const int zg = GEN ? (int)blockIdx.z : 0;     // generic mode: 32-channel output group
  float* const lbias = reinterpret_cast<float*>(smem + ACT_BYTES + 18 * 1024);
  if (EPI == EPI_LRELU && tid < 32) {
    const float* __restrict__ bias = GEN ? a.bias[0] + 32 * zg : (blockIdx.z ? a.bias[1] : a.bias[0]);
    lbias[tid] = bias[tid];
  }
  unsigned gofs[AITER];   // halfs, inside one plane
  int lofs[AITER];        // bytes, inside the LDS halo image
  unsigned okmask = 0;    // bit it: item it is an in-image pixel
#pragma unroll
  for (int it = 0; it < AITER; ++it) {
    const int i = tid + it * NT;
    const int p = min(i >> 2, NPIX - 1), q = i & 3;
    const int hy = p / HWD, hx = p - hy * HWD;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    const bool ok = (y >= 0) & (y < H) & (x >= 0) & (x < W);
    const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
    gofs[it] = (unsigned)(((n * H + yc) * W + xc) * 32 + q * 8);
    lofs[it] = hy * ROWB + hx * PS + q * 16;
    okmask |= (ok ? 1u : 0u) << it;
  }

  // EPI_BWD with f16 plane output: the epilogue works on ROWS of an M-tile - lane l owns channels 8*(l&3).. of pixel l>>2 of a
  // 16-pixel row, so every global access of a wave is one contiguous KiB (the accumulator layout gives 8-byte pieces of 32
  // different pixels per instruction: the epilogue was 1.6 of the 3.0 ms the data-gradient convs cost a step at 8 per rank,
  // profiles/r6/ab_experiments.txt r6r).  The addends and the mask do not depend on the conv: they are fetched under the last
  // stage's MFMAs.  The accumulators change lanes through LDS (wave-private, after the last stage's barrier).
  // They land in the stage-prefetch registers, which the last stage no longer needs (registers of their own took the G/H pair launch
  // from 4 to 2 workgroups per CU, i.e. to two rounds).
  constexpr int EROWS = (EPI == EPI_BWD) ? 2 * MT : 1;
  constexpr int NEX = 3 * EROWS > AITER + WITER ? 3 * EROWS - AITER - WITER : 1;
  u32x4 areg[AITER];
  u32x4 wreg[WITER];
  u32x4 epx[NEX];
  auto ep_slot = [&](const int i) __attribute__((always_inline)) -> u32x4& {
    return i < AITER ? areg[i] : (i < AITER + WITER ? wreg[i - AITER] : epx[i - AITER - WITER]);
  };
  auto epi_row = [&](const int e, bool& ok) __attribute__((always_inline)) -> size_t {
    const int mt = wave * MT + (e >> 1);
    const int y = ty0 + 2 * (mt / (TW / 16)) + (e & 1), x = tx0 + 16 * (mt % (TW / 16)) + (lane >> 2);
    ok = (y < H) & (x < W);
    return ((size_t)(n * H + min(y, H - 1)) * W + min(x, W - 1)) * 32 + 8 * (lane & 3);
  };
  auto epi_bwd_prefetch = [&]() __attribute__((always_inline)) {
    const int z = zg;
    const bool masked = a.bw_mask && (a.bw_mask_z <= -2 || z == a.bw_mask_z);
#pragma unroll
    for (int e = 0; e < EROWS; ++e) {
      bool ok;
      const size_t o = epi_row(e, ok);
      if (ABL(a, 128)) continue;
      if (a.bw_add) ep_slot(e) = *reinterpret_cast<const u32x4*>(a.bw_add + (size_t)z * a.plane + o);
      if (a.bw_add2) ep_slot(EROWS + e) = *reinterpret_cast<const u32x4*>(a.bw_add2 + (size_t)z * a.plane + o);
      if (masked) ep_slot(2 * EROWS + e) = *reinterpret_cast<const u32x4*>(a.bw_mask + (a.bw_mask_z <= -2 ? (size_t)z * a.plane : 0) + o);
    }
  };

#pragma unroll
  for (int net_i = 0; net_i < NNETS; ++net_i) {
    constexpr bool gen = GEN;
    const int net = (EPI == EPI_GH) ? net_i : (gen ? 0 : (int)blockIdx.z);
    // ternaries, not a.dense[net]: a dynamically indexed by-value array goes to scratch
    const f16* __restrict__ dense = net ? a.dense[1] : a.dense[0];
    const u32x4* __restrict__ wsrc = reinterpret_cast<const u32x4*>((net ? a.w[1] : a.w[0]) + (gen ? (size_t)blockIdx.z * a.wz_stride : 0));
    const int tclip = gen ? n % a.T : 0;                 // frame index inside its clip (temporal taps)
    const long frame_stride = (long)H * W * 32;           // halfs per frame inside a plane

    int fragbase = 0;      // fragments consumed by earlier stages

    // -- staging helpers -------------------------------------------------------
    // Item i = tid + it*NT covers 16-byte chunk q = i&3 of halo pixel p = i>>2.  Its global offset
    // inside a 32-channel plane, its LDS offset and its validity do not depend on the stage, so they
    // are computed once (the per-stage address math was as expensive to issue as the stage's MFMAs).
    // A 16-wide stage simply stages the (zero) pad half of its plane as well.
    auto load_stage = [&](const C3Stage st, const int fb) __attribute__((always_inline)) {
      if (ABL(a, 2)) return;
      const bool tv = (tclip + st.dt >= 0) & (tclip + st.dt < (gen ? a.T : 1 << 30));
      const bool second = GEN && st.kind == 2;             // workgroup-uniform: the pair's other net (gen_split)
      const f16* __restrict__ src = (second ? a.dense[1] : dense) + (size_t)(st.coff >> 5) * a.plane + (tv ? st.dt * frame_stride : 0);
      if (!ABL(a, 64))
#pragma unroll
      for (int it = 0; it < AITER; ++it) areg[it] = *reinterpret_cast<const u32x4*>(src + gofs[it]);
      const int nfr = (GEN && a.gen_sp1) ? 2 : 9 * (st.width >> 4);
      const u32x4* __restrict__ ws = second ? reinterpret_cast<const u32x4*>(a.w[1] + (size_t)blockIdx.z * a.wz_stride2) : wsrc;
      const int fbb = second ? fb - 18 * a.gen_split : fb;
      if (!ABL(a, 32))
#pragma unroll
      for (int it = 0; it < WITER; ++it) {
        const int i = min(tid + it * NT, nfr * 64 - 1);  // unconditional (clamped): keeps wreg in registers
        wreg[it] = ws[(size_t)fbb * 64 + i];
      }
    };
    auto store_stage = [&](const C3Stage st) __attribute__((always_inline)) {
      if (ABL(a, 4)) return;
      const bool tv = (tclip + st.dt >= 0) & (tclip + st.dt < (gen ? a.T : 1 << 30));
      const unsigned okm = tv ? okmask : 0u;               // a temporal tap outside the clip is zero padding too
#pragma unroll
      for (int it = 0; it < AITER; ++it) {
        // out-of-image pixels become the conv's zero padding here (a select right after the load
        // would make the compiler drain the prefetch before the MFMA phase)
        if (tid + it * NT < NPIX * 4)
          *reinterpret_cast<u32x4*>(lact + lofs[it]) = ((okm >> it) & 1u) ? areg[it] : u32x4{0u, 0u, 0u, 0u};
      }
      const int nfr = (GEN && a.gen_sp1) ? 2 : 9 * (st.width >> 4);
#pragma unroll
      for (int it = 0; it < WITER; ++it) {
        const int i = tid + it * NT;
        if (i < nfr * 64) *reinterpret_cast<u32x4*>(lw + i * 16) = wreg[it];
      }
    };
    // im2col stage: row of pixel p holds x1[p + tap][c] at k = tap*c1 + c, zero above 9*c1.
    // Two phases so that no global-load latency is serialised: (1) the (TH+2)x(TW+2) halo of
    // x1 goes to LDS as 4 x f16 per pixel (parked in the weight area, which is filled last),
    // (2) rows are assembled LDS -> LDS.
    auto fill_im2col = [&]() __attribute__((always_inline)) {
      if (ABL(a, 16)) return;
      const int c1 = a.c1;
      constexpr int XITER = (NPIX + NT - 1) / NT;
      unsigned char* const lx = lw + 4096;          // 8 B per halo pixel, after the 2 im2col weight fragments
      float4 xv[XITER];
#pragma unroll
      for (int it = 0; it < XITER; ++it) {
        const int p = tid + it * NT;
        const int hy = p / HWD, hx = p - hy * HWD;
        const int y = ty0 + hy - 1, x = tx0 + hx - 1;
        const bool ok = (p < NPIX) & (y >= 0) & (y < H) & (x >= 0) & (x < W);
        const int yc = min(max(y, 0), H - 1), xc = min(max(x, 0), W - 1);
        const float4 v = *reinterpret_cast<const float4*>(a.x1 + ((size_t)(n * H + yc) * W + xc) * 4);
        xv[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      for (int i = tid; i < 2 * 64; i += NT) *reinterpret_cast<u32x4*>(lw + i * 16) = wsrc[i];
#pragma unroll
      for (int it = 0; it < XITER; ++it) {
        const int p = tid + it * NT;
        if (p < NPIX) {
          uint2 u;
          u.x = pack2(xv[it].x, xv[it].y);
          u.y = pack2(xv[it].z, xv[it].w);
          *reinterpret_cast<uint2*>(lx + p * 8) = u;
        }
      }
      __syncthreads();
      if (c1 == 3) {
        // one output pixel per item: 9 x 8-byte reads of the halo, one 64-byte row (k = tap*3 + c) out
        for (int p = tid; p < TH * TW; p += NT) {
          const int ly = p / TW, lxx = p - ly * TW;
          f16 rowv[32];
#pragma unroll
          for (int k = 27; k < 32; ++k) rowv[k] = (f16)0.f;
#pragma unroll
          for (int tap = 0; tap < 9; ++tap) {
            const f16x4 sv = *reinterpret_cast<const f16x4*>(lx + ((ly + tap / 3) * HWD + (lxx + tap % 3)) * 8);
            rowv[tap * 3 + 0] = sv[0];
            rowv[tap * 3 + 1] = sv[1];
            rowv[tap * 3 + 2] = sv[2];
          }
          unsigned char* row = lact + (ly + 1) * ROWB + (lxx + 1) * PS;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rowv[8 * j + e];
            *reinterpret_cast<f16x8*>(row + 16 * j) = o;
          }
        }
      } else {
        for (int i = tid; i < TH * TW * 10; i += NT) {
          const int p = i / 10, tap = i - p * 10;
          const int ly = p / TW, lxx = p - ly * TW;
          f16* row = reinterpret_cast<f16*>(lact + (ly + 1) * ROWB + (lxx + 1) * PS);
          if (tap == 9) {
            for (int k = 9 * c1; k < 32; ++k) row[k] = (f16)0.f;
          } else {
            const f16* src = reinterpret_cast<const f16*>(lx + ((ly + tap / 3) * HWD + (lxx + tap % 3)) * 8);
            for (int c = 0; c < c1; ++c) row[tap * c1 + c] = src[c];
          }
        }
      }
    };

    // -- prologue: stage 0 --------------------------------------------------------
    if (net_i > 0) __syncthreads();  // previous net's last MFMAs are done with LDS
    if (a.has_im2col) {
      fill_im2col();
    } else {
      const C3Stage st0 = stage_of<GEN>(a, 0);
      load_stage(st0, 0);
      store_stage(st0);
    }
    __syncthreads();

    C3STAMP(1);
    for (int s = 0; s < a.nstages; ++s) {
      const C3Stage st = stage_of<GEN>(a, s);
      const C3Stage stn = stage_of<GEN>(a, s + 1);
      const int nfr = (st.kind == 1 || (GEN && a.gen_sp1)) ? 2 : 9 * (st.width >> 4);
      const bool more = s + 1 < a.nstages;
      if (more) load_stage(stn, fragbase + nfr);
      else if (ROWEPI && EPI == EPI_BWD && !a.plain) epi_bwd_prefetch();

      if (ABL(a, 1)) {
      } else if (st.kind == 1 || (GEN && a.gen_sp1)) {
        constexpr int CTR = ROWB + PS;  // centre tap
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const f16x8 af = *reinterpret_cast<const f16x8*>(lw + ks * 1024 + lane * 16);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const f16x8 bf = *reinterpret_cast<const f16x8*>(lact + pbase[m] + CTR + ks * 32);
            acc[net_i][m] = mfma_32x32x16(af, bf, acc[net_i][m]);
          }
        }
      } else if (st.width == 32) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const f16x8 af = *reinterpret_cast<const f16x8*>(lw + (tap * 2 + ks) * 1024 + lane * 16);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              const f16x8 bf = *reinterpret_cast<const f16x8*>(lact + pbase[m] + (tap / 3) * ROWB + (tap % 3) * PS + ks * 32);
              acc[net_i][m] = mfma_32x32x16(af, bf, acc[net_i][m]);
            }
          }
        }
      } else {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const f16x8 af = *reinterpret_cast<const f16x8*>(lw + tap * 1024 + lane * 16);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const f16x8 bf = *reinterpret_cast<const f16x8*>(lact + pbase[m] + (tap / 3) * ROWB + (tap % 3) * PS);
            acc[net_i][m] = mfma_32x32x16(af, bf, acc[net_i][m]);
          }
        }
      }
      fragbase += nfr;
      if (more) {
        __syncthreads();
        store_stage(stn);
        __syncthreads();
      }
    }
  }

  C3STAMP(2);
  // -- epilogue -------------------------------------------------------------------
  // acc[..][m][r]: pixel = lane&31 of M-tile m, outch = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int half = lane >> 5;
  if (ABL(a, 8)) {   // keep the accumulators alive without the stores
    float keep = 0.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) keep += acc[0][m][0] + acc[NNETS - 1][m][5];
    if (keep == 123.456f) a.out[0][0] = (f16)keep;
    return;
  }
  if (ROWEPI && EPI == EPI_BWD && !a.plain) {
    constexpr int EP = 144;      // bytes per pixel of the fp32 exchange image (32 channels + 16: conflict-free 16-byte writes)
    __syncthreads();             // every wave is done with the halo image and the weights
    unsigned char* const ex = smem + wave * (MT * 32 * EP);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(ex + (m * 32 + (lane & 31)) * EP + (8 * g + 4 * half) * 4) =
            make_float4(acc[0][m][4 * g], acc[0][m][4 * g + 1], acc[0][m][4 * g + 2], acc[0][m][4 * g + 3]);
    const bool masked = a.bw_mask && (a.bw_mask_z <= -2 || zg == a.bw_mask_z);
    const float mslope = a.bw_mask_z == -3 ? 0.f : 0.2f;
    f16* const obase = (a.bw_alt && zg == a.bw_mask_z) ? a.bw_alt : a.out[0] + (size_t)((a.out_coff >> 5) + zg) * a.plane;
    C3STAMP(3);
#pragma unroll
    for (int e = 0; e < EROWS; ++e) {
      bool ok;
      const size_t o = epi_row(e, ok);
      const unsigned char* src = ex + ((e >> 1) * 32 + (e & 1) * 16 + (lane >> 2)) * EP + (lane & 3) * 32;
      const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 16);
      float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      if (a.bw_add) {
        const f16x8 t = __builtin_bit_cast(f16x8, ep_slot(e));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)t[j];
      }
      if (a.bw_add2) {
        const f16x8 t = __builtin_bit_cast(f16x8, ep_slot(EROWS + e));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += (float)t[j];
      }
      if (masked) {
        const f16x8 t = __builtin_bit_cast(f16x8, ep_slot(2 * EROWS + e));
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= ((float)t[j] > 0.f) ? 1.f : mslope;
      }
      if (ABL(a, 256) && v[0] + v[5] != 123.456f) continue;
      if (ok) *reinterpret_cast<u32x4*>(obase + o) = u32x4{pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
    }
#ifdef SELFC_DEV
    C3STAMP(4);
    if (a.stamps) { __builtin_amdgcn_s_waitcnt(0); C3STAMP(5); if (threadIdx.x == 0 && blockIdx.x < 512 && blockIdx.z == 0) a.stamps[blockIdx.x * 8 + 6] = a.nstages; }
#endif
    return;
  }
  float bwmax = 0.f;           // EPI_BWD plain output with bw_amax_out: max |stored value| of this lane
  bool bwnan = false;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int y = ty0 + py[m], x = tx0 + px[m];
    if (y >= H || x >= W) continue;
    const size_t pix = (size_t)(n * H + y) * W + x;
    if (EPI == EPI_LRELU) {
      // lanes l and l+32 own the same pixel and interleaved 4-channel groups; one half-swap per
      // dword hands each lane 8 contiguous channels -> two 16-byte stores per M-tile
      f16* dst = ((blockIdx.z && !zg) ? a.out[1] : a.out[0]) + (size_t)((a.out_coff >> 5) + zg) * a.plane + pix * 32 + 8 * half;
      uint32_t r[4][2];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(lbias + 8 * g + 4 * half);
        r[g][0] = pack2(lrelu02(acc[0][m][4 * g + 0] + b.x), lrelu02(acc[0][m][4 * g + 1] + b.y));
        r[g][1] = pack2(lrelu02(acc[0][m][4 * g + 2] + b.z), lrelu02(acc[0][m][4 * g + 3] + b.w));
      }
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        // before: r[2gp] = {lower: ch 16gp+0..3, upper: 16gp+4..7}, r[2gp+1] = {lower: 16gp+8..11, upper: 16gp+12..15}
        // swap(r[2gp].upper <-> r[2gp+1].lower): lower lane holds ch 16gp+0..7, upper lane ch 16gp+8..15
        u32x4 v;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const auto sw = __builtin_amdgcn_permlane32_swap(r[2 * gp][d], r[2 * gp + 1][d], false, false);
          v[d] = sw[0];
          v[2 + d] = sw[1];
        }
        *reinterpret_cast<u32x4*>(dst + 16 * gp) = v;
      }
    } else if (EPI == EPI_BWD) {
      const bool masked = a.bw_mask && (a.bw_mask_z <= -2 || zg == a.bw_mask_z);     // -2 / -3: every group, mask planes z
      const float mslope = a.bw_mask_z == -3 ? 0.f : 0.2f;                           // -3: ReLU' instead of LeakyReLU'
      float v[4][4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[g][j] = acc[0][m][4 * g + j];
      if (a.bw_add) {
        const f16* __restrict__ ad = a.bw_add + (size_t)zg * a.plane + pix * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f16x4 t = *reinterpret_cast<const f16x4*>(ad + 8 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[g][j] += (float)t[j];
        }
      }
      if (a.bw_add2) {
        const f16* __restrict__ ad = a.bw_add2 + (size_t)zg * a.plane + pix * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f16x4 t = *reinterpret_cast<const f16x4*>(ad + 8 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[g][j] += (float)t[j];
        }
      }
      if (masked) {
        const f16* __restrict__ mk = a.bw_mask + (a.bw_mask_z <= -2 ? (size_t)zg * a.plane : 0) + pix * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f16x4 t = *reinterpret_cast<const f16x4*>(mk + 8 * g);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[g][j] *= ((float)t[j] > 0.f) ? 1.f : mslope;
        }
      }
      if (a.plain) {
        const float am = *a.bw_amax;
        const float inv = 1.f / grad_scale(am);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int oc = 32 * zg + 8 * g + 4 * half;
          if (oc < a.coutp) {
            float4* dst = reinterpret_cast<float4*>(a.plain + pix * a.coutp + oc);
            float4 o = make_float4(v[g][0] * inv, v[g][1] * inv, v[g][2] * inv, v[g][3] * inv);
            if (a.bw_acc) {
              const float4 old = *dst;
              o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            *dst = o;
            bwmax = fmaxf(fmaxf(bwmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            bwnan |= (o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (o.w != o.w);
          }
        }
      } else if (!ROWEPI) {     // f16 plane output in the accumulator layout (ROWEPI: the row epilogue above)
        f16* dst = ((a.bw_alt && zg == a.bw_mask_z) ? a.bw_alt : a.out[0] + (size_t)((a.out_coff >> 5) + zg) * a.plane) + pix * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2 u;
          u.x = pack2(v[g][0], v[g][1]);
          u.y = pack2(v[g][2], v[g][3]);
          *reinterpret_cast<uint2*>(dst + 8 * g) = u;
        }
      }
    } else if (EPI == EPI_PLAIN) {
      const float* __restrict__ bias = a.bias[0];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oc = 32 * zg + 8 * g + 4 * half;
        if (oc < a.coutp) {
          const float4 b = *reinterpret_cast<const float4*>(bias + oc);
          *reinterpret_cast<float4*>(a.plain + pix * a.coutp + oc) =
              make_float4(acc[0][m][4 * g] + b.x, acc[0][m][4 * g + 1] + b.y, acc[0][m][4 * g + 2] + b.z, acc[0][m][4 * g + 3] + b.w);
        }
      }
    } else if (EPI == EPI_F) {
      // y1 = x1 + F(x2)  (Inv_arch.py:25)  /  y1 = x1 - F(y2)  (:31); outch 0..3 live in g == 0, half == 0
      if (half == 0) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias[0]);
        float4 v = *reinterpret_cast<float4*>(a.x1io + pix * 4);
        const float sgn = a.rev ? -1.f : 1.f;
        v.x += sgn * (acc[0][m][0] + b.x);
        v.y += sgn * (acc[0][m][1] + b.y);
        v.z += sgn * (acc[0][m][2] + b.z);
        v.w += sgn * (acc[0][m][3] + b.w);
        *reinterpret_cast<float4*>(a.x1out + pix * 4) = v;
      }
    } else {  // EPI_GH: s = clamp*(2*sigmoid(H)-1); y2 = x2*exp(s)+G  /  (x2-G)/exp(s)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int oc = 8 * g + 4 * half;
        if (oc < a.c2p) {
          const float4 bg = *reinterpret_cast<const float4*>(a.bias[0] + oc);
          const float4 bh = *reinterpret_cast<const float4*>(a.bias[1] + oc);
          const float4 xv = *reinterpret_cast<const float4*>(a.x2io + pix * a.c2p + oc);
          const float xin[4] = {xv.x, xv.y, xv.z, xv.w};
          const float gb[4] = {bg.x, bg.y, bg.z, bg.w}, hb[4] = {bh.x, bh.y, bh.z, bh.w};
          float yo[4], so[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float gv = acc[0][m][4 * g + j] + gb[j];
            const float hv = acc[1][m][4 * g + j] + hb[j];
            const float s = a.clamp * (2.f / (1.f + expf(-hv)) - 1.f);
            so[j] = s;
            yo[j] = a.rev ? (xin[j] - gv) / expf(s) : xin[j] * expf(s) + gv;
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
  }
  if (EPI == EPI_BWD) {
    if (a.plain && a.bw_amax_out) {        // one atomic per wave: the same max (and NaN convention) absmax_kernel would find in `plain`
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bwmax = fmaxf(bwmax, __shfl_xor(bwmax, o));
      unsigned* const bits = reinterpret_cast<unsigned*>(a.bw_amax_out);
      if (__any(bwnan)) { if (lane == 0) atomicMax(bits, 0x7fc00000u); }
      else if (lane == 0 && bwmax > 0.f) atomicMax(bits, __float_as_uint(bwmax));
    }
  }
