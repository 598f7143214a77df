/* selfc_hip.h - C ABI of libselfc_hip.so, the MI355X (gfx950) implementation of
 * SelfC's invertible-rescaling hot path.
 *
 * The reference (tianyuan168326/SelfC) has NO native code and no FFI: every op
 * below replaces a composition of stock torch ops inside a Python nn.Module.
 * Each entry point cites the reference expression it replaces (paths relative
 * to codes/).  The binding a reference maintainer would add is a ctypes stub;
 * it is shown in INTEGRATION.md and implemented in selfc_amd/_lib.py.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by
 *     the caller; no allocation, no host synchronisation inside any call;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all
 *     work is enqueued on it, calls are re-entrant and hipGraph-capturable;
 *   - return value: 0 on success, SELFC_EINVAL (-1) for a shape/argument the
 *     kernels do not cover (nothing is launched), or -(hipError_t) - 1000 when
 *     the HIP runtime refused a launch;
 *   - "NCHW" tensors are the reference's own layout (fp32, contiguous).
 *
 * Internal ("latent") layout used between kernels, for an InvBlockExp with
 * channel split (c1, c2), c1 <= 3:
 *   x1    fp32 [N][H][W][4]        first c1 channels real
 *   x2    fp32 [N][H][W][c2p]      c2p = roundup(c2, 4)
 *   fd    f16  [FC/32][N][H][W][32]  F subnet dense buffer, FC = roundup(c2,32)+128 channels
 *                                  [x2 as f16 | zero pad | f1 f2 f3 f4], stored as 32-channel
 *                                  planes so that each group of a pixel is one contiguous 64 B
 *   gd,hd f16  [4][N][H][W][32]    G / H subnet dense feature buffers (f1..f4)
 * All workspaces must be zero-initialised once by the caller (pad channels are never written).
 * N = B*T frames, clip-major (frame n = b*T + t), as in Subnet_constructor.py:119-124.
 */
#ifndef SELFC_HIP_H
#define SELFC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SELFC_OK 0
#define SELFC_EINVAL (-1)

#define SELFC_SUBNET_D2DT 0   /* Subnet_constructor.py:98-133  (temporal conv5) */
#define SELFC_SUBNET_DB2D 1   /* Subnet_constructor.py:8-34    (3x3 conv5)      */

/* library / build identification: returns a static string "selfc_hip gfx950 <abi>" */
const char* selfc_version(void);
/* number of bytes the packed weights of one conv occupy; mirrors selfc_amd/packing.py */
int selfc_abi_version(void);

/* ---- split / merge transforms (HBM-bound) --------------------------------- */

/* HaarDownsampling.forward(x, rev=False): Inv_arch.py:64-73.
 * x NCHW (N,C,H,W) -> y NCHW (N,4C,H/2,W/2), y[:,k*C+c] = band k. H, W even. */
int selfc_haar_fwd_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream);
/* HaarDownsampling.forward(x, rev=True): Inv_arch.py:74-81. y (N,4C,h,w) -> x (N,C,2h,2w). */
int selfc_haar_inv_nchw(const float* y, float* x, int N, int C, int h, int w, void* stream);

/* FrequencyAnalyzer.forward(x, rev=False): SelfC_GMM_arch_inv.py:73-78 (k = 4 or 2).
 * x NCHW (N,3,H,W) -> latent x1 (lo, 3 ch), x2 (3k^2 ch, channel (sy*k+sx)*3+c),
 * and, when fd != NULL, the f16 copy of x2 in channels [0,3k^2) of the F dense
 * buffer (FC = its channel count, used for validation only). */
int selfc_freq_fwd(const float* x, float* x1, float* x2, void* fd, int FC,
                   int N, int H, int W, int k, void* stream);
/* FrequencyAnalyzer.forward(x, rev=True): SelfC_GMM_arch_inv.py:79-82 (nn.PixelShuffle
 * channel order c*k^2+sy*k+sx - deliberately not the inverse of the forward). */
int selfc_freq_inv(const float* x1, const float* x2, float* x, int N, int h, int w, int k, void* stream);

/* NCHW (N,c1+c2,H,W) <-> latent (x1,x2[,fd]); the narrow/cat of Inv_arch.py:22,33. */
int selfc_nchw_to_latent(const float* x, float* x1, float* x2, void* fd, int FC,
                         int N, int c1, int c2, int H, int W, void* stream);
int selfc_latent_to_nchw(const float* x1, const float* x2, float* y,
                         int N, int c1, int c2, int H, int W, void* stream);

/* Quantization.forward on the LR channels, in place on x1: Quantization.py:7-17
 * (clamp to [0,1], round-half-even(x*255)/255). n = number of floats. */
int selfc_quantize_inplace(float* x, size_t n, void* stream);
/* The same for any class-level setting of the reference (Quantization.quant_v, Quantization.is_clip: Quantization.py:9-13,20-22):
 * optional clamp to [0,1], then round-half-even(x*quant_v)/quant_v. */
int selfc_quantize_inplace_v(float* x, size_t n, float quant_v, int is_clip, void* stream);

/* Y-channel squared error of test_rescaling.py's PSNR (rgb_to_ycbcr data/util.py:239-245, calculate_psnr
 * utils/util.py:198-221): a, b NCHW (N,3,H,W) RGB in [0,1]; partial[n][blk] (double, selfc_y_sse_blocks(HW)
 * per frame) = partial sums of (Ya - Yb)^2; PSNR_n = 10 log10(HW / sum_blk partial[n][blk]). Deterministic. */
int selfc_y_sse_blocks(int HW);
int selfc_y_sse(const float* a, const float* b, double* partial, int N, int HW, void* stream);

/* Y-channel SSIM sums of calculate_ssim(rgb_to_ycbcr(a), rgb_to_ycbcr(b)) (utils/util.py:396-441,597-605; 11-tap window
 * win11, no padding, data_range 1): a, b (N,3,H,W) RGB in [0,1]; partial[N][ceil((H-10)/16)][ceil((W-10)/16)] doubles, the
 * sum over a frame's entries divided by (H-10)*(W-10) is its SSIM. */
int selfc_y_ssim(const float* a, const float* b, const float* win11, double* partial, int N, int H, int W, void* stream);

/* Guassian_downsample(x, scale=4) of feed_data's "sr_bd" LR target (models/Guassian.py:7-52, SelfC_model.py:128):
 * per plane, 13x13 Gaussian g169 (sigma 1.6, row-major) at stride 4 with reflect padding; x (planes,H,W) ->
 * y (planes,H/4,W/4); H, W multiples of 4 and >= 8. */
int selfc_gauss_down4(const float* x, float* y, const float* g169, int planes, int H, int W, void* stream);

/* ---- dense-block subnets --------------------------------------------------- */

typedef struct {
  const void* w3[4];     /* packed f16 MFMA A-fragments of conv1..conv4 (packing.py) */
  const float* b3[4];    /* 32 fp32 biases each                                       */
  const void* w5;        /* packed conv5 fragments                                    */
  const float* b5;       /* conv5 bias, zero-padded to a multiple of 32 floats        */
  const void* wfused;    /* optional: fused conv1..4 fragment stream.  cin == 3 subnets (G, H):
                            packing.py:pack_fused_gh - when G and H both carry one, their conv1..4 run as
                            ONE persistent fused launch (csrc/fused_gh.hip).  cin == 48 (F of SelfC-large):
                            packing.py:pack_fused_f - conv1+conv2 and conv3+conv4 run as two pairwise-fused
                            launches (csrc/fused_f.hip) */
  const void* w5p;       /* optional (F with wfused, temporal conv5): packing.py:pack_f5_partial - the fused F
                            launches then also emit the conv5 partial products and a small kernel replaces the
                            conv5 pass over the 176 dense channels (needs selfc_latent.pf) */
} selfc_subnet_w;

typedef struct {
  selfc_subnet_w F;      /* F: c2 -> c1 */
  selfc_subnet_w G;      /* G: c1 -> c2 ; G.w5 holds the G+H conv5 fragments interleaved */
  selfc_subnet_w H;      /* H: c1 -> c2 ; H.w5 unused (NULL)                             */
  float clamp;           /* InvBlockExp.clamp, Inv_arch.py:15 */
} selfc_invblock_w;

typedef struct {
  int kind;              /* SELFC_SUBNET_D2DT / SELFC_SUBNET_DB2D */
  int N, T, H, W;        /* frames (B*T), temporal length, latent height/width */
  int c1, c2;            /* channel split (c1 <= 3, c2 <= 48 for D2DT, c2 <= 32 for DB2D; wider: composed from selfc_subnet_run) */
  float* x1;             /* latent state, updated in place */
  float* x2;
  void* fd;              /* workspaces (see layout above) */
  void* gd;
  void* hd;
  float* s_out;          /* optional: InvBlockExp.s as fp32 [N][H][W][c2p], or NULL */
  float* pf;             /* optional workspace: F conv5 partial products, fp32 [2 pairs][3 taps][N][H][W][4] (see w5p), or NULL */
  int flags;             /* SELFC_LAT_* (abi 7) */
  void* fd_next;         /* optional (abi 8), rev == 0 only: where the G/H epilogue puts the f16 copy of the updated x2 (the NEXT
                            block's F input) instead of this block's own `fd` - a training forward that keeps one private `fd` per
                            block (its backward then finds F's input planes untouched).  NULL: `fd` (blocks chained in place). */
  float* x1_out;         /* optional (abi 9): where the updated x1 / x2 go instead of over their inputs, which then stay intact - a */
  float* x2_out;         /* training forward keeps them for the backward pass without cloning.  NULL: in place. */
} selfc_latent;

/* The caller reads the dense feature buffers (fd / gd / hd planes f1..f4) after the call - the training forward, whose
 * backward consumes them (selfc_subnet_bwd).  Without it (inference) a kernel may keep features that nothing else reads
 * on chip: the pairwise-fused F launches then do not store f3 / f4 (128 B per pixel-frame per block and direction). */
#define SELFC_LAT_KEEP_FEATURES 1

/* InvBlockExp.forward(x, rev): Inv_arch.py:21-33 on the latent layout.
 * Precondition: channels [0,c2) of `fd` hold x2 as f16 when rev == 0 (every
 * producer in this library maintains that).  Postcondition: the same holds for
 * the updated x2, so blocks chain without touching NCHW. */
int selfc_invblock_run(const selfc_invblock_w* blk, const selfc_latent* lat, int rev, void* stream);
/* nblk consecutive InvBlockExp (the op loops of SelfC_GMM_arch_inv.py:455-456 and
 * :486-487): blocks 0..nblk-1 when rev == 0, nblk-1..0 when rev != 0. */
int selfc_invstack_run(const selfc_invblock_w* blks, int nblk, const selfc_latent* lat, int rev, void* stream);

/* One stand-alone subnet (DenseBlock.forward / D2DTInput.forward,
 * Subnet_constructor.py:26-34,115-133): input NHWC fp32 `xin` with channel
 * stride cinp = roundup(cin,4), output NHWC fp32 `yout` with stride
 * coutp = roundup(cout,4); `dense` is a zero-initialised [DC/32][N][H][W][32] f16
 * workspace with DC = (cin <= 3 ? 128 : roundup(cin,32)+128).  cin > 3 and xin == NULL:
 * the input planes of `dense` are already filled (by selfc_globalagg_run_d) and are used as they are. */
int selfc_subnet_run(const selfc_subnet_w* w, int kind, const float* xin, float* yout, void* dense,
                     int N, int T, int H, int W, int cin, int cout, void* stream);

/* Generic dense-block conv on a plane-blocked f16 buffer [P][N][H][W][32] (used for FeatureCalapseBlock,
 * Subnet_constructor.py:280-324: gc = 128, (3,3,3) conv1 / conv5): reads planes [0, nplanes_in), kt = 1
 * ((1,3,3) Conv3d) or 3 ((3,3,3): frames n-1, n, n+1 of a T-frame clip, zero outside), cout a multiple of
 * 32.  out_plane >= 0: LeakyReLU(0.2) output appended as f16 planes out_plane .. out_plane + cout/32 - 1;
 * out_plane < 0: no activation, fp32 NHWC rows of stride cout into `plain`.  `w`: pack_conv_planes. */
int selfc_conv_planes_run(void* dense, int nplanes_in, int kt, const void* w, const float* bias, int cout,
                          int out_plane, float* plain, int N, int T, int H, int W, void* stream);
/* fp32 NHWC rows (stride roundup(cin,4)) -> f16 planes [0, roundup(cin,32)/32) of a plane-blocked buffer. */
int selfc_nhwc_to_planes(const float* x, void* dense, size_t npix, int cin, void* stream);

/* NHWC(4-padded) fp32 <-> NCHW fp32 helpers for the stand-alone subnet entry. */
int selfc_nchw_to_nhwc4(const float* x, float* y, int N, int C, int H, int W, void* stream);
int selfc_nhwc4_to_nchw(const float* x, float* y, int N, int C, int H, int W, void* stream);

/* ---- training: gradients of the dense-block subnets and the coupling (csrc/backward.hip) --------------
 *
 * The reference trains through stock autograd (SelfC_model.py:153-176 optimize_parameters); these entry points are
 * what the autograd.Functions of the boundary modules call.  Gradients pass through the MFMA as f16 scaled by a
 * power of two taken from max|dout| of the call; all results are fp32 and unscaled. */
typedef struct {
  const void* wt5;      /* conv5^T  : dOut planes -> [x groups | f1 f2 f3 f4]   (packing.py:pack_subnet_bwd) */
  const void* wtd[3];   /* conv_k^T : [dpre4 ..] -> f3, f2, f1 */
  const void* wtx;      /* conv_k^T : [dpre4 dpre3 dpre2 dpre1] -> x */
} selfc_subnet_bw;

size_t selfc_subnet_bwd_scratch_bytes(int N, int H, int W, int cin, int cout);
/* Backward of DenseBlock.forward / D2DTInput.forward (Subnet_constructor.py:26-34,119-133) given the dense buffer
 * the forward left behind (`dense`: [x planes when cin > 3 | f1..f4], see selfc_subnet_run) and, when cin <= 3,
 * the fp32 NHWC(4) input `xin`.  `dout`: fp32 NHWC rows of stride roundup(cout,4), used as sign*dout.
 * dx (optional): fp32 NHWC rows of stride roundup(cin,4), overwritten or (accumulate_dx) added to.
 * wgrad[k] / bgrad[k] (k = 0..4, each optional): gradients of conv{k+1}.weight / .bias in the reference's own
 * layout ((32|cout, cin+32k, [1|3,] 3|1, 3|1) contiguous fp32), written as beta*old + new. */
int selfc_subnet_bwd(const selfc_subnet_bw* bw, int kind, const void* dense, const float* xin, const float* dout, float sign,
                     float* dx, int accumulate_dx, float* const* wgrad, float* const* bgrad, float beta,
                     void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout, void* stream);
/* The same in two phases, so that a caller can put the weight gradients on a second stream: SELFC_BWD_DATA computes dx and
 * leaves the scaled gradient planes in `scratch`; SELFC_BWD_WEIGHTS (same arguments, same untouched scratch, ordered after
 * the data phase) produces wgrad / bgrad. */
#define SELFC_BWD_DATA 1
#define SELFC_BWD_WEIGHTS 2
/* (abi 13, with wg_jobs of selfc_subnet_bwd_phase_d / selfc_gh_bwd_pair) the deferred weight-gradient jobs are built THIN: one workgroup
 * per (conv, input plane) pair walking every tile - a launch meant to run in the background of another stream's kernels */
#define SELFC_BWD_WG_THIN 8
int selfc_subnet_bwd_phase(int phases, const selfc_subnet_bw* bw, int kind, const void* dense, const float* xin, const float* dout,
                           float sign, float* dx, int accumulate_dx, float* const* wgrad, float* const* bgrad, float beta,
                           void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout, void* stream);
/* The affine coupling of InvBlockExp (Inv_arch.py:26-27 / 29-30) as a stand-alone elementwise pass over n fp32 elements
 * (n % 4 == 0; any layout, shared by all operands): s = clamp*(2*sigmoid(h)-1); rev == 0: y2 = x2*e^s + g; rev != 0:
 * y2 = (x2-g)/e^s.  The fused conv5 epilogues do this for channel_split_num <= 3; a block with a wider split is composed from
 * stand-alone subnets (selfc_subnet_run) and this pass. */
int selfc_coupling_fwd(int rev, const float* x2, const float* g, const float* h, float* y2, float* s, float clamp, size_t n, void* stream);
/* Gradient of the affine coupling of InvBlockExp (Inv_arch.py:24-32) w.r.t. its x2 path and H's output, n = npix*c2p
 * fp32 elements: rev == 0: v = x2 (input), dx2 = dy2*e^s, dh = dx2*x2*ds/dh (dG = dy2);
 * rev != 0: v = y2 (output), dx2 = dy2*e^-s, dh = -dy2*y2*ds/dh (dG = -dx2); ds/dh = clamp*(1-(s/clamp)^2)/2. */
int selfc_coupling_bwd(int rev, const float* v, const float* s, const float* dy2, float* dx2, float* dh, float clamp,
                       size_t n, void* stream);
/* (abi 12) max|dOut| where dOut is produced, instead of a pass over it per subnet call: a training step issues 61 such passes on
 * its critical path.  `selfc_coupling_bwd_x` also takes max|dx2| / max|dh| (either pointer may be NULL), `selfc_add_absmax` is
 * a += b with max|a|, and `selfc_subnet_bwd_phase_x` takes the finished max of its dOut (`dout_amax`, NULL: it takes the max itself)
 * and leaves max|dx| of what it stored in `dx_amax_out` (NULL: not wanted).  Every max slot is one float the CALLER zeroes first; all
 * use atomic max on the float bits with the NaN convention of the internal pass, so gradients are bit-identical either way. */
int selfc_coupling_bwd_x(int rev, const float* v, const float* s, const float* dy2, float* dx2, float* dh, float clamp,
                         size_t n, float* dx2_amax, float* dh_amax, void* stream);
int selfc_add_absmax(float* a, const float* b, size_t n, float* amax, void* stream);
int selfc_subnet_bwd_phase_x(int phases, const selfc_subnet_bw* bw, int kind, const void* dense, const float* xin, const float* dout,
                             float sign, float* dx, int accumulate_dx, float* const* wgrad, float* const* bgrad, float beta,
                             void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout,
                             const float* dout_amax, float* dx_amax_out, void* stream);
/* (abi 13) Fewer, fatter launches for the training backward (the reference's autograd launches one ATen op per layer and tensor,
 * models/SelfC_model.py:148-183; here a replayed step is made of graph nodes, and on a 36x36 training latent every node is latency).
 * - `selfc_subnet_bwd_phase_d` = `selfc_subnet_bwd_phase_x` with deferred weight-gradient finishes: with `fin_jobs` (room for 2 jobs of
 *   `selfc_fin_job_bytes()` each, HOST memory) the weights phase leaves its two partial-sum reductions as job descriptors instead
 *   of launching them; `selfc_wgrad_finish_jobs` runs any number of collected jobs (24 per launch), in order.  The scratch buffers of
 *   the deferred calls must stay untouched until that launch has run.  fin_jobs NULL: as before.
 * - `selfc_gh_bwd_pair`: G and H of ONE InvBlockExp (Inv_arch.py:18-20,26-30: same input, same shapes, input gradients add up) in one
 *   call: every step of the subnet backward runs once for both nets (a grid dimension), under one power-of-two gradient scale taken
 *   from max(max|dOut_G|, max|dOut_H|) (`amax_g` / `amax_h`: the finished maxima, NULL = taken here), and dx (+= when accumulate_dx)
 *   is a single conv over both nets' gradient planes.  cin <= 3, D2DTInput subnets.  `fin_jobs`: room for 4 jobs, or NULL.
 *   Scratch: `selfc_gh_bwd_pair_scratch_bytes`. */
size_t selfc_fin_job_bytes(void);
int selfc_wgrad_finish_jobs(const void* jobs, int njobs, void* stream);
/* - `wg_jobs` (with fin_jobs; room for 2 - pair: 4 - jobs of `selfc_wg_job_bytes()` each, HOST memory): the weights phase launches
 *   NOTHING and leaves its weight-gradient launches as job descriptors too; `selfc_wgrad_run_jobs` runs any number of them as ONE launch
 *   per kind (conv1..4 / temporal conv5) and 32 jobs - the weight gradients of a whole block stack behind its data-gradient chain: on a
 *   training crop each such launch under-fills the chip, and inside a replayed graph side streams buy no overlap on this runtime.
 *   Call order: selfc_wgrad_run_jobs, then selfc_wgrad_finish_jobs; every scratch / saved-feature buffer of the deferred calls stays
 *   untouched until then. */
size_t selfc_wg_job_bytes(void);
int selfc_wgrad_run_jobs(const void* jobs, int njobs, void* stream);
int selfc_subnet_bwd_phase_d(int phases, const selfc_subnet_bw* bw, int kind, const void* dense, const float* xin, const float* dout,
                             float sign, float* dx, int accumulate_dx, float* const* wgrad, float* const* bgrad, float beta,
                             void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout,
                             const float* dout_amax, float* dx_amax_out, void* fin_jobs, void* wg_jobs, void* stream);
size_t selfc_gh_bwd_pair_scratch_bytes(int N, int H, int W, int cin, int cout);
int selfc_gh_bwd_pair(int phases, const selfc_subnet_bw* bw_g, const selfc_subnet_bw* bw_h, const void* dense_g, const void* dense_h,
                      const float* xin, const float* dout_g, const float* dout_h, float sign_g, float sign_h,
                      float* dx, int accumulate_dx, float* const* wgrad_g, float* const* bgrad_g, float* const* wgrad_h, float* const* bgrad_h,
                      float beta, void* scratch, size_t scratch_bytes, int N, int T, int H, int W, int cin, int cout,
                      const float* amax_g, const float* amax_h, float* dx_amax_out, void* fin_jobs, void* wg_jobs, void* stream);
/* (abi 13) ReconstructionLoss (models/modules/loss.py:5-21) and its gradient in two launches: *out = weight * mean(v), v = (x-t)^2
 * (l1 == 0) or sqrt((x-t)^2 + eps) (l1 != 0); grad (dense [n_outer][inner], or NULL) = weight / count * dv/dx (the gradient w.r.t. t is
 * its negative).  x / t: n_outer rows of `inner` contiguous floats at row strides stride_x / stride_t (elements).  partial: scratch of
 * selfc_recon_loss_blocks() doubles.  Deterministic, no atomics, capturable. */
int selfc_recon_loss_blocks(void);
int selfc_recon_loss(const float* x, size_t stride_x, const float* t, size_t stride_t, size_t n_outer, size_t inner, int l1, float eps,
                     float weight, float* grad, double* partial, float* out, void* stream);
/* (abi 13) nn.utils.clip_grad_norm_(max_norm) + torch.optim.Adam.step() (models/SelfC_model.py:172-176) on ONE flat fp32 buffer of n
 * elements as two launches: the gradient's L2 norm (deterministic two-stage sum, *norm_out, the value clip_grad_norm_ returns), the
 * gradient scaled IN PLACE by min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: no clipping) and Adam's update of param / exp_avg /
 * exp_avg_sq in torch's operation order (amsgrad and maximize off).  step: `step_dev` (device float, incremented here - a capturable
 * optimizer's state) or, when NULL, `step_host` = the already incremented count; lr likewise (`lr_dev` or `lr_host`).  partial:
 * scratch of selfc_clip_adam_blocks() doubles.  No atomics, no memset: capturable. */
int selfc_clip_adam_blocks(void);
int selfc_clip_adam(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, double* partial, float max_norm,
                    const float* lr_dev, float lr_host, double beta1, double beta2, float eps, float weight_decay,
                    float* step_dev, float step_host, float* norm_out, void* stream);
/* Adjoint of selfc_freq_fwd (latent grads d1 [N][h][w][4], d2 [N][h][w][48] -> dx NCHW (N,3,H,W)) and of selfc_freq_inv
 * (dout NCHW -> d1, d2). */
int selfc_freq_fwd_bwd(const float* d1, const float* d2, float* dx, int N, int H, int W, void* stream);
int selfc_freq_inv_bwd(const float* dout, float* d1, float* d2, int N, int H, int W, void* stream);

/* Building blocks of the host-orchestrated gradients (STP head: selfc_amd/autograd.py).  A "plane" is a 32-channel
 * f16 slab [N*H*W][32]; gradient planes are scaled by S = grad scale of *amax (selfc_bwd_scale), fp32 results are not. */
int selfc_bwd_scale(const float* g, size_t n, float* amax, void* stream);                 /* *amax = max|g| (device) */
/* fp32 rows (stride cs, c channels) -> roundup(c,32)/32 planes: sign * S * (lrelu ? LeakyReLU(x) : x); amax NULL: S = 1 */
int selfc_bwd_to_planes(const float* x, void* planes, size_t npix, int c, int cs, int lrelu, float sign, const float* amax, void* stream);
int selfc_f16_rows_to_planes(const void* rows, void* planes, size_t npix, int C, void* stream);   /* f16 [npix][C] -> planes */
/* Generic plane-list conv with the gradient epilogue: in = nplanes_in contiguous planes, (kt, sp1 ? 1x1 : 3x3) kernel
 * packed by packing.py:pack_planes_generic, ngroups 32-channel output groups.  v = acc + add[z] (optional planes);
 * v *= LeakyReLU'(mask) for group mask_z (mask = one plane) or for every group (mask_z == -2, mask = ngroups planes;
 * mask_z == -3: the same with ReLU' - 0 instead of 0.2 where the saved activation is not positive: the 'gmm_thin' head).
 * Output: f16 planes out_planes[z], or (plain) fp32 rows of stride coutp holding v / S (+ old when accumulate). */
int selfc_bwd_conv_planes(const void* in, int nplanes_in, int kt, int sp1, const void* w, int ngroups, void* out_planes,
                          const void* add, const void* mask, int mask_z, float* plain, int coutp, int accumulate,
                          const float* amax, int N, int T, int H, int W, void* stream);
size_t selfc_bwd_wgrad_scratch_bytes(int N, int H, int W, int Pn, int Qn, int taps);
/* wout (O, Ctot, taps) = beta*wout + sum_px P[px][o] * Q[px + tap][c] / S;  bout (O) likewise from column sums of P.
 * P: Pn gradient planes, Q: Qn activation planes, taps 9 (3x3) or 1. */
int selfc_bwd_wgrad(const void* P, int Pn, const void* Q, int Qn, int taps, float* wout, int O, int Ctot, float* bout, float beta,
                    const float* amax, void* scratch, size_t scratch_bytes, int N, int T, int H, int W, void* stream);

/* ---- STP (self-conditioned latent predictor), activations fp32 NHWC [N][H*W][64] -------------
 *
 * GlobalAgg.forward: SelfC_GMM_arch_inv.py:265-285.  y = x + (proj1(x) viewed (b,C*h*w,T)) @ A,
 * A = softmax(proj2(g) proj3(g)^T / C, dim=-1), g = fc(adaptive_avg_pool2d(x, 32x32)).
 * `wmap` [H*W] is fc folded through the adaptive pooling (host, selfc_amd/packing.py:pool_weight_map),
 * `w1` proj1 as pointwise fragments (pack_pointwise), w2/b2/w3/b3 the fp32 Linear(64,64) params,
 * `partial` selfc_globalagg_partial_floats(N,HW) floats, `attn` (N/T)*T*T floats. C = 64, T <= 8, x != y. */
int selfc_globalagg_run(const float* x, float* y, const float* wmap, const float* fc_bias /* device, 1 float */, const void* w1, const float* b1,
                        const float* w2, const float* b2, const float* w3, const float* b3,
                        float* partial, float* attn, int N, int T, int HW, void* stream);
/* The general entry.  c_real: a GlobalAgg(c) with c < 64 (codec variant: c = stp_hidden_c = 24, SelfC_Codec_arch_inv.py:103-131)
 * runs with activations and parameters zero-padded to 64 channels by the caller (exact: padded channels stay 0) and passes the
 * module's real channel count, which is the softmax temperature of `/C` (:120).  Exactly one of `y` / `dense_out` is given:
 * `dense_out` = the plane-blocked f16 operand buffer ([plane][N][HW][32]) of the D2DTInput that consumes the result - its planes
 * 0..1 are written directly (the rounding selfc_subnet_run's own input conversion applies; call it with xin = NULL afterwards).
 * The clip attention (tiny) is the prologue of every mix workgroup; `attn` may be NULL (else it receives A, (N/T)*T*T floats). */
int selfc_globalagg_run_d(const float* x, float* y, void* dense_out, const float* wmap, const float* fc_bias, const void* w1,
                          const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                          float* partial, float* attn, int N, int T, int HW, int c_real, void* stream);
size_t selfc_globalagg_partial_floats(int N, int HW);

/* Pointwise conv = the Conv3d(.,.,1) layers of STPNet.tail_gmm (SelfC_GMM_arch_inv.py:327-354):
 * out[px][o] = act_out(sum_k W[o][k] act_in(in[px][k]) + b[o]); act flag 0 = none, 1 = LeakyReLU(0.2), 2 = ReLU (gmm_thin head).
 * in: fp32 or f16 rows of cin (32 | cin <= 256) channels; out: fp32 or f16 rows of stride cout_stride;
 * cout a multiple of 16; `w` from pack_pointwise, bias zero-padded to cout. */
int selfc_pwconv_run(const void* in, int in_is_f32, void* out, int out_is_f32, const void* w, const float* bias,
                     size_t npix, int cin, int cout, int cout_stride, int lrelu_in, int lrelu_out, void* stream);

/* GMM sample of STPNet.forward (SelfC_GMM_arch_inv.py:382-394): raw fp32 [npix][hf_dim*K*3] in the
 * reference's (hf_dim, K, 3) order, eps fp32 [npix][hf_dim*K]; v fp32 [npix][hf_dim] (= the x2 latent
 * layout).  pi = softmax over the hf_dim axis, log-sigma clamped to [-7,7].  hf_dim = 48, K in {1,3,5}. */
int selfc_gmm_sample(const float* raw, const float* eps, float* v, size_t npix, int hf_dim, int K, void* stream);
/* Any (hf_dim, K): raw rows of stride raw_stride >= hf_dim*K*3, v rows of stride v_stride.  logsigma_scale = 1 gives the
 * sampler above; 0.5 is STP v1's `std = exp(0.5 logvar)` (SelfC_arch_inv.py:151-162,179-186; hf_dim = 9 there). */
int selfc_gmm_sample_generic(const float* raw, const float* eps, float* v, size_t npix, int hf_dim, int K, int raw_stride,
                             int v_stride, float logsigma_scale, void* stream);
/* The whole GMM head and the GMM sample in one kernel (sampling path of SelfC_GMM_arch_inv.py:327-344,371-394):
 * feat (fp32 rows [npix][64]) -LeakyReLU-> Conv3d 1x1x1 64->128 -act-> 128->256 -act-> 256->hf_dim*K*3 -> v; no activation
 * of the head is written to memory.  act: 1 = LeakyReLU(0.2) ('gmm').  w: ONE fragment stream [W0 | W1 | W2], each layer
 * packed by packing.py:pack_pointwise after a permutation of its OUTPUT channels - W0 / W1 by packing.py:head_row_perm (the
 * MFMA result layout becomes the next layer's operand layout), W2 by packing.py:gmm_head_perm (new channel
 * (3 k + j) * hf_dim + c = reference channel (c*K + k)*3 + j); bias: the three permuted biases back to back (128 + 256 + 720);
 * eps: fp32 rows [npix][k * hf_dim + c]; v: fp32 rows of stride v_stride.  hf_dim = 48, K = 5. */
int selfc_stp_head_gmm(const float* feat, const void* w, const float* bias, const float* eps, float* v, size_t npix,
                       int hf_dim, int K, int v_stride, int act, void* stream);

/* ---- live kernel timing (bench.py roofline leg) ------------------------------
 * HIP events are recorded on the launch stream around every kernel launch while
 * enabled.  Classes: 0 dense 3x3 conv (conv1..4), 1 conv5+coupling of F,
 * 2 conv5+coupling of G/H, 3 split/merge/layout transforms, 4 stand-alone conv5, 5 STP kernels,
 * 6 fused G/H conv1..4.
 * Not thread-safe; keep disabled while capturing a hipGraph.  The reference has
 * no counterpart (its only timing is commented-out time.time(), SelfC_model.py:194). */
int selfc_profile_enable(int on);
int selfc_profile_read(int cls, double* total_ms, long long* launches);   /* waits for the recorded events */
int selfc_profile_reset(void);
/* Box calibration (bench.py prints it beside its value; boxes of one pool differ by up to 10 % on one binary): the f16
 * MFMA rate of a register-only 32x32x16 loop (TFLOP/s at the clock the chip holds under full matrix load) and the rate
 * of a 512-MiB device-to-device copy (read + write bytes, GB/s).  The ONE entry point that allocates (1 GiB, freed before
 * it returns) and waits on the host; not capturable.  No reference counterpart. */
int selfc_profile_calibrate(double* mfma_tflops, double* copy_GBps, void* stream);
/* Shader clock under load: enqueues ONE wave on `stream` that watches the shader-clock counter against the constant
 * 100 MHz counter for `micros` microseconds (<= 500,000) and then writes out2[0] = shader cycles, out2[1] = 100 MHz ticks
 * (device memory, 2 x u64).  Launch it on a side stream next to the workload; GHz = 0.1 * out2[0] / out2[1]. */
int selfc_profile_clock_sample(unsigned long long* out2, int micros, void* stream);
/* (abi 11) A non-blocking HIP stream owned by the caller (hipStreamCreateWithFlags / hipStreamDestroy).  The host side captures its
 * hipGraphs on, and forks onto, streams of its own rather than streams of torch's shared 32-entry pool (runtime.own_stream). */
int selfc_stream_create(void** out);
int selfc_stream_destroy(void* stream);
/* (abi 13) Node census of a captured hipGraph_t: counts[5] = {all, kernel, memset, memcpy, other}.  Host-only.  No reference counterpart
 * (the reference launches eagerly); used by the tests ("no memset node in any graph of the package") and by bench.py (nodes per step). */
int selfc_graph_stats(void* graph, long long* counts);

/* ---- indirect tensor addresses (abi 10): the module API as ONE replayed hipGraph ----
 * The reference's callers pass a new input tensor to every netG(x=..., rev=...) call and own the tensors it returns
 * (SelfC_model.py:213-230; torch.cat / convs always return fresh storage, SURVEY 8b "ownership").  A hipGraph bakes kernel
 * arguments in, so the graph-side transforms that touch a CALLER's tensor take the device address of a pointer SLOT
 * instead of the tensor: selfc_set_pointers (one tiny launch on the same stream, right before the replay) stores the
 * call's base addresses into slots 0..n-1 of `table`, and each *_ind transform reads `*slot + off` elements (off = the
 * part of the batch this launch owns).  Same arithmetic, same layouts, same error codes as the direct entry points. */
int selfc_set_pointers(void** table, int n, const void* p0, const void* p1, const void* p2, const void* p3, void* stream);
int selfc_freq_fwd_ind(const float* const* xslot, size_t xoff, float* x1, float* x2, void* fd, int FC, int N, int H, int W, int k, void* stream);
int selfc_freq_inv_ind(const float* x1, const float* x2, float* const* xslot, size_t xoff, int N, int h, int w, int k, void* stream);
int selfc_nchw_to_latent_ind(const float* const* xslot, size_t xoff, float* x1, float* x2, void* fd, int FC, int N, int c1, int c2, int H, int W, void* stream);
int selfc_latent_to_nchw_ind(const float* x1, const float* x2, float* const* yslot, size_t yoff, int N, int c1, int c2, int H, int W, void* stream);
int selfc_nchw_to_nhwc4_ind(const float* const* xslot, size_t xoff, float* y, int N, int C, int H, int W, void* stream);
int selfc_nhwc4_to_nchw_ind(const float* x, float* const* yslot, size_t yoff, int N, int C, int H, int W, void* stream);

/* ---- STP gradients (csrc/stp.hip) ---- */
/* d raw of selfc_gmm_sample given dv: raw/draw [npix][hf_dim*K*3], eps [npix][hf_dim*K], dv [npix][hf_dim]. */
int selfc_gmm_sample_bwd(const float* raw, const float* eps, const float* dv, float* draw, size_t npix, int hf_dim, int K, void* stream);
/* The same for selfc_gmm_sample_generic (any hf_dim / K / row strides / log-sigma scale: STP v1's head, SelfC_arch_inv.py:151-186);
 * draw rows have raw's stride, columns beyond hf_dim*K*3 are set to 0. */
int selfc_gmm_sample_generic_bwd(const float* raw, const float* eps, const float* dv, float* draw, size_t npix, int hf_dim, int K,
                                 int raw_stride, int v_stride, float logsigma_scale, void* stream);
int selfc_lrelu_bwd(float* dx, const float* x, size_t n, void* stream);       /* dx *= (x > 0 ? 1 : 0.2), n % 4 == 0 */
/* dst_i[j] = beta_i * dst_i[j] + sum over rows_i of src_i[r][j] (src_i row-major [rows_i][len_i]) for n <= 8 small matrices in one
 * launch: the per-clip partial gradients of selfc_globalagg_bwd summed over the clips straight into their destination
 * (autograd.globalagg_bwd: the trainer's flat gradient buffer with beta = 1).  No reference counterpart (autograd does it). */
#define SELFC_ROWSUM_MAX 8
typedef struct {
  const float* src[SELFC_ROWSUM_MAX];
  float* dst[SELFC_ROWSUM_MAX];
  int len[SELFC_ROWSUM_MAX], rows[SELFC_ROWSUM_MAX];
  float beta[SELFC_ROWSUM_MAX];
  int n;
} selfc_rowsum;
int selfc_rowsum_accum(const selfc_rowsum* job, void* stream);
size_t selfc_globalagg_bwd_scratch_bytes(int N, int T, int H, int W);
/* Backward of selfc_globalagg_run (GlobalAgg.forward, SelfC_GMM_arch_inv.py:265-285): x, dy, dx fp32 [N][H*W][64];
 * w1t = pack_planes_generic(proj1.weight^T).  Parameter gradients: dw1 (64,64) complete; the others per clip
 * ([B][64], [B][64*64], [B], dwmap_clip [B][H*W]) - the caller sums over clips and folds dwmap through the pooling
 * map (packing.py:pool_weight_map) into fc.weight. */
int selfc_globalagg_bwd(const float* x, const float* dy, float* dx, const float* wmap, const float* fc_bias, const void* w1t,
                        const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                        float* dw1, float* db1_clip, float* dw2_clip, float* db2_clip, float* dw3_clip, float* db3_clip,
                        float* dfcb_clip, float* dwmap_clip, void* scratch, size_t scratch_bytes,
                        int N, int T, int H, int W, void* stream);
/* (abi 13) the same with the gradient maxima folded in (as selfc_subnet_bwd_phase_x does inside the block stacks): `dy_amax` = the finished
 * max|dy| of whoever produced dy (NULL: taken here), `dx_amax_out` = a zeroed float that receives max|dx| (NULL: not wanted) - the STP
 * chain alternates dense blocks and GlobalAgg blocks, each consuming the other's input gradient */
int selfc_globalagg_bwd_x(const float* x, const float* dy, float* dx, const float* wmap, const float* fc_bias, const void* w1t,
                        const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                        float* dw1, float* db1_clip, float* dw2_clip, float* db2_clip, float* dw3_clip, float* db3_clip,
                        float* dfcb_clip, float* dwmap_clip, void* scratch, size_t scratch_bytes,
                        int N, int T, int H, int W, const float* dy_amax, float* dx_amax_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SELFC_HIP_H */
