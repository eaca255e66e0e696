/*
 * fdsr.h -- C ABI of libfdsr_hip.so, the MI355X (gfx950) engine behind the
 * FastDiffSR 20-step sampling path.
 *
 * The reference (Meng-333/FastDiffSR) is pure Python and has no FFI of its
 * own; the drop-in boundary is the duck-typed netG surface that
 * FastDiffSR/model/model.py (class DDPM) calls.  Each entry point below names
 * the reference interface it stands behind (file:line under
 * /root/reference/FastDiffSR/).  The Python facade in fastdiffsr_amd/
 * (diffusion.py, unet.py) binds these with ctypes and re-exposes the
 * reference's class/method names; INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative FDSR_E_* code; nothing
 *    throws across the ABI; fdsr_last_error() gives the message.
 *  - tensors at the boundary are fp32, NCHW, contiguous, DEVICE pointers
 *    (the layout the reference's tensors have, LRHR_dataset.py:113-119);
 *    weights are handed over as HOST pointers in the reference checkpoint
 *    layout (Conv2d [Cout,Cin,kh,kw], Linear [out,in]) and repacked inside.
 *  - the caller owns every tensor and the workspace; the library owns its
 *    packed weights and captured graphs.  One handle per device; a handle is
 *    not thread-safe.  All work is stream-ordered on `stream` and asynchronous:
 *    the execution entry points do not synchronise the device, with one exception: the first
 *    fdsr_sample after the weights or the schedule changed builds the noise-embedding table for
 *    all T levels (one small launch + a stream synchronise).  Loading weights, fdsr_set_schedule
 *    and fdsr_set_seed are host-synchronous copies.
 *  - H and W must be multiples of 2^(n_mults-1) (three stride-2 stages => 8).
 */
#ifndef FDSR_H_
#define FDSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDSR_OK 0
#define FDSR_E_INVALID (-1)   /* bad argument / unsupported configuration  */
#define FDSR_E_KEY (-2)       /* unknown checkpoint key or shape mismatch  */
#define FDSR_E_STATE (-3)     /* weights / schedule missing                */
#define FDSR_E_WORKSPACE (-4) /* workspace too small or misaligned         */
#define FDSR_E_HIP (-5)       /* HIP runtime error (see fdsr_last_error)   */
#define FDSR_E_SATURATED (-6) /* f16x3: a raw conv input left the f16 range (fdsr_check_saturation) */

#define FDSR_MAX_MULTS 8

typedef struct fdsr_engine* fdsr_handle;

/* Hyper-parameters of unet.UNet(...) exactly as networks.define_G passes them
 * (model/networks.py:94-104; ctor model/fastdiffsr_modules/unet.py:224-297). */
typedef struct fdsr_config {
  int32_t in_channel;    /* 6 = cat(cond SR image, x_t)                    */
  int32_t out_channel;   /* 3                                              */
  int32_t inner_channel; /* 64                                             */
  int32_t norm_groups;   /* 32                                             */
  int32_t n_mults;
  int32_t channel_mults[FDSR_MAX_MULTS]; /* {1,2,4,4}                      */
  int32_t res_blocks;    /* 2                                              */
  float dropout;         /* 0.2; identity in eval (sampling) mode          */
  int32_t image_size;    /* FastDiffSR: informational.  SR3 variant: attention is placed where the
                            resolution (image_size halved per level) is in attn_res (ddpm_modules/unet.py:183) */
  int32_t variant;       /* FDSR_VARIANT_FASTDIFFSR (model/fastdiffsr_modules), _SR3 (model/ddpm_modules) or _TESR (model/tesr_modules) */
  int32_t n_attn_res;
  int32_t attn_res[FDSR_MAX_MULTS];
} fdsr_config;
#define FDSR_VARIANT_FASTDIFFSR 0
#define FDSR_VARIANT_SR3 1
#define FDSR_VARIANT_GDP 3   /* model/gdp_modules: the guided-diffusion UNet (scale-shift-norm ResBlocks, up/down ResBlocks, multi-head attention); inner_channel = model_channels, attn_res = attention_resolutions (downsample rates); predicts x_0; input cat[x, cond] */
#define FDSR_VARIANT_TESR 2  /* model/tesr_modules: FastDiffSR's blocks and noise-level embedding, SR3's SelfAttention placement, sampler returns x_0 */

/* Per-timestep scalars the reverse process reads (diffusion.py:109-155; only
 * these five buffers + the fp64 sqrt_alphas_cumprod_prev list are used by
 * p_sample, :157-190).  All arrays have n_timestep entries, index = t. */
typedef struct fdsr_schedule {
  int32_t n_timestep;
  const float* noise_level;      /* fp32(sqrt_alphas_cumprod_prev[t+1])  :169-170 */
  const float* sqrt_recip;       /* sqrt_recip_alphas_cumprod[t]         :157-159 */
  const float* sqrt_recipm1;     /* sqrt_recipm1_alphas_cumprod[t]                 */
  const float* coef1;            /* posterior_mean_coef1[t]              :161-165 */
  const float* coef2;            /* posterior_mean_coef2[t]                        */
  const float* sigma;            /* exp(0.5*posterior_log_variance_clipped[t]) :190 */
} fdsr_schedule;

/* -- lifetime ------------------------------------------------------------ */
/* unet.UNet.__init__ + GaussianDiffusion.__init__ (unet.py:224, diffusion.py:78). */
int fdsr_create(const fdsr_config* cfg, fdsr_handle* out);
void fdsr_destroy(fdsr_handle h);
/* Message of the last failing call on this handle (or the global one if h==NULL). */
const char* fdsr_last_error(fdsr_handle h);
/* "gfx950 f32-mfma ..." build string. */
const char* fdsr_version(void);

/* -- checkpoint schema: nn.Module.state_dict()/load_state_dict (model.py:135,159) */
int fdsr_num_weights(fdsr_handle h);
/* idx-th tensor of the UNet schema (keys without the 'denoise_fn.' prefix), in
 * state_dict order.  live=0 for the 22 never-executed `<blk>.conv` layers
 * (unet.py:212) that exist only in checkpoints. */
int fdsr_weight_info(fdsr_handle h, int idx, char* key, int key_cap,
                     int64_t shape[4], int* ndim, int* live);
/* Hand over one checkpoint tensor (host pointer, reference layout). */
int fdsr_load_weight(fdsr_handle h, const char* key, const float* host,
                     const int64_t* shape, int ndim);
/* 1 when every live tensor has been loaded. */
int fdsr_weights_complete(fdsr_handle h);

/* GaussianDiffusion.set_new_noise_schedule (diffusion.py:109-155). */
int fdsr_set_schedule(fdsr_handle h, const fdsr_schedule* s);

/* -- execution ------------------------------------------------------------ */
/* Bytes of device scratch fdsr_unet_forward / fdsr_sample need for this shape. */
int fdsr_workspace_bytes(fdsr_handle h, int batch, int height, int width, size_t* bytes);

/* UNet.forward(x, noise_level) (unet.py:299-323), eval mode.
 *   x_nchw      [B,in_channel,H,W]   noise_level [B]   eps_nchw [B,out_channel,H,W] */
int fdsr_unet_forward(fdsr_handle h, const float* x_nchw, const float* noise_level,
                      float* eps_nchw, int batch, int height, int width,
                      void* workspace, size_t workspace_bytes, void* hip_stream);

/* SR3 variant: same entry point; noise is [T+1,B,3,H,W] (a draw exists for t = 0 too and is masked, ddpm_modules/
 * diffusion.py:189-196), the network sees the integer time, and out is x_0 itself (no res2img, :226-227).
 *
 * GaussianDiffusion.p_sample_loop, conditional branch (diffusion.py:192-221),
 * batched as B independent B=1 runs (the reference crashes for B>=2, :215-216).
 *   cond_nchw [B,3,H,W]        the bicubic-upsampled LR image (x_in)
 *   noise     [T,B,3,H,W]      noise[0] = x_T (`randn(shape)` :207), noise[k] =
 *                              the `randn_like` of step t = T-k (:189), k=1..T-1;
 *                              or NULL: the engine draws the same planes itself inside the
 *                              loop (Philox4x32-10 + Box-Muller keyed by fdsr_set_seed and a
 *                              per-call counter) -- the throughput mode; parity runs pass noise
 *   out_nchw  [B,3,H,W]        res2img(x_0, cond) (:214, :275-281) == ret_img[-1]
 *   traj_nchw [T,B,3,H,W] or NULL: x_t after every step (t = T-1..0), for
 *                              continous=True frames and parity tests.
 * flags: FDSR_SAMPLE_GRAPH replays the 20-step loop as one captured hipGraph; hip_stream must then be a
 *        created stream (capture cannot run on the NULL stream: FDSR_E_INVALID). */
#define FDSR_SAMPLE_GRAPH 1
int fdsr_sample(fdsr_handle h, const float* cond_nchw, const float* noise,
                float* out_nchw, float* traj_nchw, int batch, int height, int width,
                void* workspace, size_t workspace_bytes, void* hip_stream, int flags);

/* Seed of the engine-side noise (resets the per-call counter): the same seed and call order give
 * the same images, whatever the batch split or launch geometry.  The reference draws from torch's
 * global generator (diffusion.py:189, :207); its stream is not reproduced, so parity is defined on
 * explicit noise only. */
int fdsr_set_seed(fdsr_handle h, uint64_t seed);
/* Plane `plane` ([B,3,H,W] fp32, device) of the noise drawn under the current call counter
 * (plane 0 = x_T, plane k = step t = T-k).  For tests of the generator. */
int fdsr_randn(fdsr_handle h, float* dst_nchw, int batch, int height, int width, int plane, void* hip_stream);

/* Arithmetic of the convolutions (everything else is fp32 in every mode):
 *   FDSR_PREC_F32    exact fp32 on v_mfma_f32_32x32x2_f32 (default)
 *   FDSR_PREC_F16X3  fp32-grade: operands split hi/lo into two f16, three f16 MFMAs per product,
 *                    fp32 accumulate (stays inside the 1e-3 parity bound)
 *   FDSR_PREC_BF16   one bf16 MFMA per product and bf16 activations in HBM (BASELINE config 3; judged on
 *                    PSNR delta).  Needs 16-aligned channel counts (FDSR_E_INVALID otherwise); the SR3 / TESR / GDP variants
 *                    run their attention on bf16 MFMA kernels (fp32 scores and softmax). */
#define FDSR_PREC_F32 0
#define FDSR_PREC_F16X3 1
#define FDSR_PREC_BF16 2
/*   FDSR_PREC_F16    (round 6) one f16 MFMA per product -- the hi plane of the f16x3 weight forms against un-split f16 activations --
 *                    and f16 activations in HBM: the bf16 mode's bytes and MFMA rate with 11 mantissa bits instead of 8.  Stores
 *                    saturate at +-65504.  Judged on PSNR like bf16 (|delta| stays above 1e-3); every variant (the siblings' attention
 *                    kernels have f16 twins of their bf16 forms); sampling only. */
#define FDSR_PREC_F16 3
int fdsr_set_precision(fdsr_handle h, int mode);

/* -- val-loop helper ------------------------------------------ */
/* Metrics.tensor2img (core/metrics.py:16-42): clamp to [lo,hi], map to [0,1], *255, round half to
 * even, uint8.  src [B,C,H,W] fp32 device, dst [B,H,W,C] uint8 device; saves the fp32 D2H copy of
 * DDPM.get_current_visuals (model/model.py:97-111).  h may be NULL. */
int fdsr_tensor2img_u8(fdsr_handle h, const float* src_nchw, uint8_t* dst_nhwc, int batch, int channels,
                       int height, int width, float lo, float hi, void* hip_stream);

/* The per-image metric sums of the evaluation loop (sr_mfe.py:313-345: skimage.measure compare_mse / compare_psnr /
 * compare_ssim(multichannel=True) and Metrics.calculate_ergas, core/metrics.py:147-152; FDSR_SSIM_GAUSS11 adds the 11x11
 * Gaussian-window SSIM of core/metrics.py:103-145), on uint8 images that are already on the device -- the SR batch never
 * crosses PCIe as fp32 and the host does no per-pixel work.
 *   test / truth [B,H,W,C] uint8 device (C = 1..4);  out_dev [B][FDSR_METRIC_FIELDS] fp64 device:
 *     [0] sum (test - truth)^2      [1] sum test            -- exact integers
 *     [2] sum of the SSIM map, uniform 7x7 window (skimage defaults: sample covariance, K1 .01, K2 .03, L 255), interior
 *         positions of every channel                          [3] number of those positions
 *     [4], [5] the same for the 11x11 Gaussian window (sigma 1.5, "valid" part)      [6], [7] zero
 * The caller forms MSE = [0]/(H*W*C), PSNR, ERGAS = 100*sqrt(MSE / mean(test)^2 / C) / scale and SSIM = [2]/[3] with the
 * reference's scalar formulas (fastdiffsr_amd/metrics.py does).  Fixed-order reductions: reruns are bitwise identical.
 * h may be NULL. */
#define FDSR_METRIC_FIELDS 8
#define FDSR_SSIM_UNIFORM7 1
#define FDSR_SSIM_GAUSS11 2
int fdsr_image_metrics_workspace_bytes(int batch, int height, int width, size_t* bytes);
int fdsr_image_metrics_u8(fdsr_handle h, const uint8_t* test_nhwc, const uint8_t* truth_nhwc, int batch, int height,
                          int width, int channels, int flags, double* out_dev, void* workspace, size_t workspace_bytes,
                          void* hip_stream);

/* -- input-pipeline helper ------------------------------------ */
/* The dataset's tensor transform on the device (data/util.py:66-75 transform_augment: ToTensor() = uint8 / 255 as fp32,
 * HWC -> CHW, then img * (hi - lo) + lo; LRHR_dataset.py:113-119 passes min_max = (-1, 1)): the loader threads hand over
 * the decoded uint8 batch, one byte per sample crosses PCIe.  src [B,H,W,C] uint8 device, dst [B,C,H,W] fp32 device,
 * bit-identical to the torch ops of the reference.  h may be NULL. */
int fdsr_u8_to_tensor(fdsr_handle h, const uint8_t* src_nhwc, float* dst_nchw, int batch, int channels, int height,
                      int width, float lo, float hi, void* hip_stream);

/* The conditioning image: LR uint8 RGB -> PIL-exact bicubic resize (Image.BICUBIC as used by
 * data/prepare_data_mfe_dm.py:17-40; Pillow's 8-bit fixed-point two-pass resample, bit for bit) ->
 * optionally the val-time tensor transform ToTensor()*2-1 (data/util.py:66-75).
 *   src [B,h,w,3] uint8 device; tmp: B*h*W*3 bytes of device scratch;
 *   dst_u8 [B,H,W,3] uint8 and/or dst_f32 [B,3,H,W] fp32 in [-1,1] (either may be NULL).  h may be NULL.
 * The first call for a new (in,out) size builds its coefficient tables (synchronous upload). */
int fdsr_resize_bicubic_u8(fdsr_handle h, const uint8_t* src_nhwc, int batch, int in_h, int in_w, int out_h,
                           int out_w, uint8_t* tmp, uint8_t* dst_u8_nhwc, float* dst_f32_nchw, void* hip_stream);

/* f16x3 range guard.  The split-f16 arithmetic clamps every operand to +-65504.  GroupNorm'ed conv inputs are re-scaled
 * before the split, but a RAW input (ResnetBlock res_conv, Down/Upsample convs: unet.py:66-83,112) beyond that range would be
 * clamped silently; the kernels raise a sticky device flag instead.  This call synchronises `hip_stream`, reads and clears the
 * flag: FDSR_OK, or FDSR_E_SATURATED if any fdsr_sample / fdsr_unet_forward since the last check clamped a raw input (their
 * outputs are then not fp32-grade: re-run them after fdsr_set_precision(FDSR_PREC_F32), which has no such limit). */
int fdsr_check_saturation(fdsr_handle h, void* hip_stream);

/* Bits of the "k32" and "strip" options of fdsr_debug_option (which kernel form a stride-1 3x3 launch lands on; every setting
 * computes the same function within the tested bounds).  fastdiffsr_amd/_lib.py mirrors the names for the tests and bench.py. */
enum fdsr_k32_bits {
  FDSR_K32_F16X3 = 1,              /* the 16x16x32-MFMA form in f16x3 */
  FDSR_K32_BF16 = 2,               /* ... and in bf16 */
  FDSR_K32_RIDER_16ROW = 4,        /* the 16-row tile with a res_conv rider */
  FDSR_K32_SMALL_GRID_2ROW = 8,    /* the 2-row-per-wave tiles of small grids */
  FDSR_K32_UP2 = 16,               /* the sub-pixel upsample convs */
  FDSR_K32_SMALL_WG_F16X3 = 32,    /* rider-less 64-cout launches of large grids on 4-wave workgroups, two per CU (f16x3) */
  FDSR_K32_SMALL_WG_RIDER_F16X3 = 64,   /* ... those with a rider too, rider chunks first (f16x3) */
  FDSR_K32_SMALL_WG_BF16 = 128,    /* the small-workgroup form in bf16 (8-row tiles) */
  FDSR_K32_SMALL_WG_RIDER_BF16 = 512,   /* ... with a rider in bf16 (off by default: slower) */
  FDSR_K32_RIDER_FIRST_8WAVE = 1024,    /* rider chunks first on the 8-wave rider kernels (launches without a K split) */
  FDSR_K32_DEFAULT = 1 | 2 | 8 | 16 | 32 | 64 | 128 | 1024     /* 1275 */
};
enum fdsr_strip_bits {
  FDSR_STRIP_BF16_64 = 1,          /* bf16 64 -> 64 launches on the column-strip kernel (two workgroups per CU) */
  FDSR_STRIP_F16X3_64 = 2,         /* the f16x3 64 -> 64 launches (one workgroup per CU: hi / lo weight planes) */
  FDSR_STRIP_BF16_ONE_WG = 4,      /* A/B: bf16 64 -> 64 on one workgroup per CU */
  FDSR_STRIP_BF16_CAT64 = 8,       /* bf16 (64 | 64) -> 64 */
  FDSR_STRIP_BF16_RIDER = 16,      /* bf16 64 -> 64 with a res_conv rider */
  FDSR_STRIP_BF16_CAT128 = 32,     /* bf16 (128 | 64) -> 64 (off by default: 216 weight registers spill) */
  FDSR_STRIP_BF16_COUT128 = 64,    /* bf16 128 -> 128 and 64 -> 128 as two workgroups of 64 couts per strip */
  FDSR_STRIP_DEFAULT = 1 | 2 | 8 | 16 | 64                     /* 91 */
};

/* -- introspection for parity tests and bench.py -------------------------- */
/* Debug / A-B options of the launchers (process-wide; nothing in the library reads the environment).  Names:
 * "rider" (0|1|2|3), "up2" (0|1), "th_min_wgs", "splitk" (0|1), "sk_target", "wgrad_form" (0 default | 1 four-wave | 2 eight-wave
 * plain), "wgrad_colsum" (0|1), "wgrad_f32" (0|1), "drop_stage" (0|1: f16x3 training forwards apply Dropout in the staging of the 16x16x32 kernels instead of materialising the dropped
 * activation; default 1), "gnb_fuse" (0|1: f16x3 training steps run the reduce half of the GroupNorm backward in the epilogue of the
 * input-gradient launch; default 1), "wgrad_big_bytes", "strip" (bits: 1 bf16 64 -> 64 launches on the column-strip kernel, 2 the f16x3 ones, 4 A/B: bf16 on one workgroup per CU, 8 bf16 (64|64) -> 64, 16 bf16 64 -> 64 with a res_conv rider, 32 bf16 (128|64) -> 64, 64 bf16 128 -> 128 and 64 -> 128; default 91), "strip_min_wgs" (from this many strip segments on; default 512),
 * "k32" (bits: 1 f16x3, 2 bf16, 4 16-row tiles with a rider, 8 2-row tiles of small grids, 16 the sub-pixel upsample convs, 32 the rider-less f16x3 64-cout
 * launches of large grids on 4-wave workgroups, two per CU, 64 those with a rider too, 128 in bf16 too, 512 the bf16 launches with a rider, 1024 rider chunks first on the 8-wave rider kernels too (launches without a K split); default 1275 -- the 16x16x32-MFMA form of the
 * stride-1 3x3 launches), "k32_sb_min_wgs" (bit 32 from this many workgroups on; default 1024), "k32_stagger" (start delay of a CU's odd
 * workgroup slot in that form, 64-cycle units per K chunk; default 0), "gn_consumer" (0|1: small grids form GroupNorm scale / shift in the consumer conv's prologue from the producers' fixed-point
 * channel-pair sums instead of a gn_finalize launch; default 1), "sat_guard" (0|1), "bf16_f16x3_steps" (probe: bf16 sampling runs the first n, or for n < 0 the last -n, reverse steps on the f16x3 kernels; default 0),
 * "tail" (0|1: the input / output convs of the 16-bit modes on their own bandwidth-shaped kernels),
 * "drop_image_offset" (the batch is images [k, k+N) of a larger one: its dropout masks are those images' masks).
 * Every setting computes the same function within the tested bounds; they exist so that tests can force each kernel
 * form and same-box A/B runs can price them.  Returns FDSR_E_INVALID for an unknown name.  Not for production use. */
int fdsr_debug_option(const char* name, long long value);
/* When on, the next plan keeps every layer output in its own buffer. */
int fdsr_set_debug(fdsr_handle h, int on);
/* Device pointer (NHWC fp32, inside the workspace of the last forward) and shape
 * of the output of reference module `name` ("downs.4", "mid.0", "ups.7", ...). */
int fdsr_debug_tensor(fdsr_handle h, const char* name, const float** dev_ptr,
                      int* n, int* hgt, int* wid, int* ch);
/* Element size of that tensor in the workspace: 4 (fp32), or 2 in bf16 mode, which keeps every
 * activation but the packed input and eps as bf16 in HBM. */
int fdsr_debug_tensor_elem_bytes(fdsr_handle h, const char* name, int* bytes);
/* Timing hooks: record hipEvents on `stream` around every launch of the
 * dominant kernel family (the 3x3 MFMA convolutions) during the next calls,
 * then read back count / total milliseconds / algorithmic FLOPs. */
int fdsr_profile_begin(fdsr_handle h);
int fdsr_profile_end(fdsr_handle h, int* launches, double* conv_ms, double* conv_flops,
                     double* conv_bytes);

/* nn.Module.train() / .eval() of the denoiser: in training mode the Dropout(p) in front of every block2 conv
 * (unet.py:89-101, p = fdsr_config.dropout) is live, in fdsr_unet_forward and in fdsr_train_grads alike (the two
 * fp32-grade precisions only).  The keep-mask of a forward is a pure function of (fdsr_set_seed, the count of training-mode
 * forwards so far, block, element) -- Philox4x32-10 -- and can be read back for parity checks:
 * fdsr_debug_dropout_mask gives its offset inside the workspace of the last forward, [N][H][W][C] bytes (1 = keep),
 * and the factor 1/(1-p) kept elements are multiplied by.  `block` is the reference module, e.g. "downs.1". */
int fdsr_set_training(fdsr_handle h, int on);
/* Key of the dropout masks alone (fdsr_set_seed sets it too) and restart of the forward count: the facade draws it from
 * torch's generator before every training-mode call, so runs repeat under torch.manual_seed like the reference's. */
int fdsr_set_dropout_seed(fdsr_handle h, uint64_t seed);
int fdsr_debug_dropout_mask(fdsr_handle h, const char* block, const unsigned char** dev_off, int* n, int* hgt, int* wid,
                            int* ch, float* scale);

/* ---- training step (every variant) ------------------------------------------------
 * DDPM.optimize_parameters (model/model.py:47-57): zero_grad, l_pix = netG(data) = p_losses
 * (fastdiffsr_modules/diffusion.py:242-270), l_pix.sum() / (b*c*h*w), backward, Adam.step.  The engine keeps
 * an fp32 master copy of every executed checkpoint tensor, its gradient and the two Adam moments on the
 * device; the step runs in FDSR_PREC_F32 (everything exact fp32) or FDSR_PREC_F16X3 (forward, input-gradient and
 * weight-gradient convolutions fp32-grade on split-f16 MFMA kernels; everything else fp32), every reduction in a fixed
 * order: a step is bitwise reproducible.  The 44 never-executed tensors of the schema
 * (unet.py:212) get no gradient and are not touched, as in torch. */

/* Workspace for fdsr_train_grads at this shape (the forward keeps every activation). */
int fdsr_train_workspace_bytes(fdsr_handle h, int batch, int height, int width, size_t* bytes);

/* Forward + loss + backward: gradients of  loss_scale * loss(target, UNet(x, noise_level))  w.r.t. every
 * executed parameter, left on the device (fdsr_get_grad).
 *   x_nchw       [B,6,H,W]  cat([SR, x_noisy]) (diffusion.py:265-266), x_noisy = q_sample(img2res(HR,SR), gamma, noise)
 *   noise_level  [B]        gamma, the continuous sqrt(alpha_bar) drawn per sample (:246-255)
 *   target_nchw  [B,3,H,W]  the noise that q_sample mixed in (:259)
 *   loss_l2      0: nn.L1Loss(reduction='sum') (loss_type 'l1', :101-103); 1: nn.MSELoss(reduction='sum'); 2: the SUM of the Charbonnier
 *                terms sqrt(d^2 + 1e-6) (TESR's 'l1' is their mean, tesr_modules/unet.py:956-967: put the 1 / (b*c*h*w) of the mean into
 *                loss_scale beside the one of model.py:50-52)
 *   (SR3 / TESR variants: x_nchw = cat[SR, q_sample(HR, ...)] as their p_losses forms it, noise_level = the integer time t as a float
 *    (SR3, ddpm_modules/diffusion.py:279-291) or gamma (TESR); the backward then includes the SelfAttention blocks.
 *    GDP variant, gdp_modules/diffusion.py:277-299: x_nchw = cat[q_sample(HR, t), SR], noise_level = t as a float, target_nchw = HR
 *    itself -- the network predicts x_0 -- and loss_l2 = 1 for both of its loss types; the backward then runs through the scale-shift
 *    GroupNorms (whose (scale, shift) gradient feeds each ResBlock's Linear and the time MLP), the average-pooled / nearest-upsampled
 *    ResBlocks and the heads of QKVAttentionLegacy, gdp_modules/unet.py:276-439, :461-488.)
 *   loss_scale   the reference divides the summed loss by b*c*h*w before backward (model.py:50-52)
 *   loss_host    optional: receives the UNSCALED summed loss (what netG(data) returns); synchronises the stream
 * All pointers but loss_host are device pointers. */
int fdsr_train_grads(fdsr_handle h, const float* x_nchw, const float* noise_level, const float* target_nchw,
                     int loss_l2, float loss_scale, float* loss_host, int batch, int height, int width,
                     void* workspace, size_t workspace_bytes, void* hip_stream);
/* The same step from the training pair itself: img2res (diffusion.py:283-289), q_sample (:233-241) and cat([SR, x_noisy]) (:257-263)
 * run in the kernel that writes the packed network input (the arithmetic of the tensor torch forms op by op: separately rounded products and sums).  hr / sr / noise:
 * [B,3,H,W] NCHW fp32 device pointers, gamma: [B] (the continuous sqrt(alpha_bar) per sample, :246-256).  noise == NULL: the
 * engine draws N(0,1) itself (Philox, fdsr_set_seed) and uses it as the target. */
int fdsr_train_grads_pairs(fdsr_handle h, const float* hr_nchw, const float* sr_nchw, const float* gamma, const float* noise_nchw,
                           int loss_l2, float loss_scale, float* loss_host, int batch, int height, int width, void* workspace,
                           size_t workspace_bytes, void* hip_stream);

/* torch.optim.Adam.step on every executed tensor (model.py:37-38, :56: lr from the config, betas (0.9, 0.999),
 * eps 1e-8), then the device-side re-packing of the fp32 kernel forms. */
int fdsr_adam_step(fdsr_handle h, float lr, float beta1, float beta2, float eps, void* hip_stream);

/* Copy one tensor of the master copy / of the last gradients to the host, in checkpoint layout
 * (state_dict() after training; tests).  Never-executed tensors: FDSR_E_KEY. */
int fdsr_get_weight(fdsr_handle h, const char* key, float* host);
int fdsr_get_grad(fdsr_handle h, const char* key, float* host);

/* torch.optim.Adam's state of one executed tensor (exp_avg, exp_avg_sq; either may be NULL on get) and the
 * common step count: what `I{iter}_E{epoch}_opt.pth` stores and load_network restores (model.py:139-146, :161-166). */
int fdsr_get_optimizer_state(fdsr_handle h, const char* key, float* exp_avg, float* exp_avg_sq, int* step);
int fdsr_set_optimizer_state(fdsr_handle h, const char* key, const float* exp_avg, const float* exp_avg_sq, int step);

/* Device pointer and length of the gradient arena (every executed tensor in schema order).  Data-parallel
 * training (the reference wraps netG in nn.DataParallel, networks.py:116-118) sums it over the ranks in place,
 * one RCCL all-reduce, between fdsr_train_grads and fdsr_adam_step. */
int fdsr_grad_arena(fdsr_handle h, float** dev_ptr, size_t* count);

/* After optimiser steps: rebuild the 16-bit weight forms (f16x3 / bf16 sampling) from the master copy.
 * fdsr_sample and the eval-mode fdsr_unet_forward do this by themselves when needed (one host re-pack after the last optimiser
 * step, not one per step). */
int fdsr_sync_weight_forms(fdsr_handle h);

#ifdef __cplusplus
}
#endif
#endif /* FDSR_H_ */
