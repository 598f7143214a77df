// Internal interface between csrc/dense_conv.hip (the generic plane-list conv) and csrc/backward.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace selfc {

// Gradients travel through the MFMA as f16 scaled by a power of two S chosen from max|dOut| of the subnet call,
// so that the largest element lands in (128, 256]: far from both the f16 overflow limit and its subnormals.
__host__ __device__ inline float grad_scale(float amax) {
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
};
int bwd_conv_planes(const BwdConv& c, int N, int T, int H, int W, hipStream_t s);

}  // namespace selfc
