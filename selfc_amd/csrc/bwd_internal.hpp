// Internal interface between csrc/dense_conv.hip (the generic plane-list conv) and csrc/backward.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace selfc {

// Gradients travel through the MFMA as f16 scaled by a power of two S chosen from max|dOut| of the subnet call,
// so that the largest element lands in (128, 256]: far from both the f16 overflow limit and its subnormals.
__host__ __device__ inline float grad_scale(float amax) {
  if (amax != amax) return amax;        // NaN in dOut: poison the whole call (see absmax_kernel)
  return (amax > 0.f && amax < 3.0e38f) ? exp2f(8.f - ceilf(log2f(amax))) : 1.f;
}

struct BwdConv {
  const void* in;         // first input plane (f16 [N][H][W][32] each), nplanes_in contiguous planes
  int nplanes_in, kt, sp1;
  const void* w;          // packing.pack_planes_generic fragments
  int ngroups;            // 32-channel output groups (blockIdx.z)
  void* out_planes;       // f16 output: group z -> plane z of this buffer (unless plain)
  const void* add;        // optional f16 addend planes (group z -> plane z)
  const void* mask;       // optional saved feature plane: group mask_z is multiplied by LeakyReLU'
  int mask_z;
  void* alt;              // optional: group mask_z is stored here instead of out_planes + mask_z planes
  float* plain;           // fp32 NHWC output (stride coutp) instead of planes: (acc + add) / S (+ old)
  int coutp, accumulate;
  const float* amax;      // device: max|dOut| of this subnet call (defines S)
  float* amax_out;        // optional (plain output): atomic max |value| over everything this call stores - the max|dOut| of the
                          // subnet call that consumes `plain` next, taken where the values are produced (no separate pass)
  // optional second source (kt == 1, 3x3 taps): nplanes_in2 more input planes `in2` with the fragments `w2` (one output group's
  // worth per group, as `w`) and a second addend `add2` - the input gradient of a G/H pair as ONE conv over both nets' planes
  const void* in2;
  int nplanes_in2;
  const void* w2;
  const void* add2;
};
int bwd_conv_planes(const BwdConv& c, int N, int T, int H, int W, hipStream_t s);
int bwd_conv_planes_pair(const BwdConv& c0, const BwdConv& c1, int N, int T, int H, int W, hipStream_t s);
int bwd_tconv5T_pair(const void* g0, const void* g1, int ng, const void* w0, const void* w1, int nplanes_out, void* out0, void* out1,
                     const void* mask0, const void* mask1, int mask_z, void* alt0, void* alt1, int N, int T, int H, int W, hipStream_t s);
// conv5^T of a temporal dense block on the temporal-conv kernel (weights: packing.pack_t5_bwd)
int bwd_tconv5T(const void* g, int ng, const void* w, int nplanes_out, void* out_planes, const void* mask, int mask_z, void* alt,
                int N, int T, int H, int W, hipStream_t s);

// csrc/dgrad_chain.hip: dpre3, dpre2, dpre1 (planes 1..3 of gb, plane 0 = dpre4) and, with dx, the input gradient as ONE launch
int bwd_dgrad_chain(void* gb, const void* add, const void* feat, const void* const* wtd, const void* wtx, float* dx, int nx, int cinp,
                    int accumulate_dx, const float* amax, float* amax_out, int N, int H, int W, hipStream_t s);

int bwd_dgrad_chain_pair(void* gb0, void* gb1, const void* add0, const void* add1, const void* feat0, const void* feat1,
                         const void* const* wtd0, const void* const* wtd1, int nx, const float* amax, int N, int H, int W, hipStream_t s,
                         const void* wtx0 = nullptr, const void* wtx1 = nullptr, float* dx0 = nullptr, float* dx1 = nullptr, int cinp = 0, int acc0 = 0);

// csrc/backward.hip building blocks (also used by the STP gradients in csrc/stp.hip)
int bwd_absmax(const float* g, size_t n, float* amax, hipStream_t s);    // *amax = max|g| (zeroed first)
// fp32 rows (stride cs, c valid channels) -> f16 planes [nplanes][npix][32]: sign * S(*amax) * (lrelu ? LeakyReLU(x) : x);
// amax == nullptr: no scaling (activations)
int bwd_to_planes(const float* x, void* planes, size_t npix, int c, int cs, int nplanes, int lrelu, float sign,
                  const float* amax, hipStream_t s, float* amax_copy = nullptr);

struct WgradJob {
  const void* P; int Pn;            // gradient planes (scaled f16)
  const void* Q[2]; int Qn[2];      // activation planes: two contiguous runs (the second may be empty)
  int taps;                         // 9: 3x3 spatial taps; 1: pointwise
  int temporal;                     // taps == 1 only: frames n-1, n, n+1 of the clip -> weight (O, Ctot, 3)
  float* wout; int O, Ctot, cin, nx; // output layout (O, Ctot, ttot); channel map: first nx planes = cin inputs, rest features
  float* bout;                      // optional bias gradient (O)
  float beta;
};
// scratch of one job: per-split partial blocks + bias partials (ttot = 9, 3 (temporal) or 1)
size_t bwd_wgrad_scratch_bytes(int N, int H, int W, int Pn, int qtot, int ttot);
int bwd_wgrad(const WgradJob& j, const float* amax, void* scratch, int N, int T, int H, int W, hipStream_t s);
size_t bwd_wgrad14_scratch_bytes(int N, int H, int W, int nqc1);

#ifdef SELFC_DEV
unsigned long long* dev_stamp_slot();    // timing aid (SELFC_ABLATE & 512): 512 x 8 u64 of the next launch, tools/experiments/c3_stamps.py
#endif

}  // namespace selfc
