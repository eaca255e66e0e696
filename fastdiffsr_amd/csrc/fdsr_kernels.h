// Launch interfaces of the gfx950 kernels (fdsr_kernels.hip) used by the engine.
// Internal header: not part of the C ABI (include/fdsr.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fdsr.h"   // fdsr_k32_bits / fdsr_strip_bits

namespace fdsr {

// One convolution launch.  Activations are NHWC fp32.  The input is the virtual
// concat of (x0: C0 channels, x1: C1 channels) -- reference unet.py:319
// torch.cat((x, feats.pop()), dim=1) -- optionally nearest-upsampled x2
// (unet.py:66-74) and optionally passed through GroupNorm+Swish (unet.py:89-101)
// while it is staged into LDS.
struct ConvParams {
  const float* x0;
  const float* x1;       // may be null (C1 == 0)
  const float* w;        // packed [tap][Cout_pad][Cin_pad]
  const float* bias;     // [Cout]
  const float* temb;     // [N][temb_stride] + temb_off, or null (FeatureWiseAffine shift, unet.py:38-54)
  const float* res;      // residual [N,Hout,Wout,Cout] or null (may alias out)
  float* out;            // [N,Hout,Wout,Cout]
  const float* gn_scale;   // [N][C0+C1] per-channel GroupNorm scale = rstd*gamma (null: no GN+Swish)
  const float* gn_shift;   // [N][C0+C1] beta - mean*scale        (gn_finalize_kernel writes both)
  int gn_plain;            // 1: GroupNorm only (SelfAttention.norm); 0: GroupNorm + Swish (Block)
  float* part_out;         // optional [N][tiles][Cout][2]: per-tile (sum, sumsq) of OUT per channel,
                           // the GroupNorm statistics of the next Block, fused into this epilogue
  int N, Hin, Win;       // source tensor dims (before upsample)
  int Hout, Wout;
  int C0, C1, Cout;
  int Cin_pad, Cout_pad;
  int temb_stride, temb_off;
  // 16-bit MFMA path (fdsr_conv_h.hip): weights in MFMA-fragment order, see pack_weights_h()
  const void* wq;
  float w_inv_scale;     // accumulator un-scaling (weights are stored multiplied by a power of two)
  const float* w_inv_scale_dev;   // if set, the un-scaling factor is read from here (weights re-packed on the device after optimiser steps)
  // split-K (small grids only, see conv_h_ksplit): slice s of the K loop writes its raw accumulators
  // to kscratch[s][N,Hout,Wout,Cout]; splitk_reduce_kernel sums the slices in order and applies the epilogue
  int ksplit;            // <= 1: off
  float* kscratch;
  // PREC_BF16 keeps activations in HBM as bf16 (half the traffic): the 16-bit kernels then read x0/x1/res
  // and write out as bf16 through the same pointers.  out_f32: this launch still writes fp32 (the final
  // conv: eps feeds the fp32 posterior update); the fp32 kernel sets out_bf16 for the 6-channel input conv.
  int out_f32;
  int out_bf16;
  // train-mode Dropout(p) between Swish and the conv of block2 (unet.py:89-101): a = swish(gn(x)) * keep * 1/(1-p).
  // keep is a byte per element of the input tensor (dropout_mask_kernel); the fp32 kernel and the f16x3 16x16x32 kernels
  // (conv_h_drop_ok) apply it in their staging.
  const unsigned char* drop_mask;   // [N][Hin][Win][C0] or null
  float drop_scale;
  // "rider" (16-bit 3x3 stride-1 kernels only): a 1x1 convolution over a second, raw input (xr0 | xr1) accumulated into the same
  // output tile -- ResnetBlock's out = block2(h) + res_conv(x) (unet.py:104-120) in one launch, without the round trip of the
  // res_conv output through HBM.  Its weights are the 1x1 conv's own fragments (same WN), its bias is added in the epilogue.
  const float* xr0;      // null: no rider
  const float* xr1;
  int Cr0, Cr1, nkr;     // nkr: 16-channel chunks of the rider (its fragment stride)
  const void* wq_r;
  const float* bias_r;
  float w_inv_scale_r;               // the rider's accumulator un-scaling; the accumulator is brought to THIS scale when the
  const float* w_inv_scale_r_dev;    // K loop passes from the main chunks to the rider chunks (powers of two: exact)
  int* sat_flag;                     // f16x3: set to 1 when a RAW input value exceeds the f16 range (null: no check)
  int stagger;                       // small-workgroup k32 form: workgroups in an odd slot of their CU start this many 64-cycle sleeps per K chunk late
  // Training, input-gradient launches of a GroupNorm'd convolution (f16x3 16x16x32 kernels only, fdsr_conv_k32.hip): the epilogue turns
  // the gradient w.r.t. the ACTIVATED input, dA = this convolution's output, into g = dA * keep/(1-p) * swish'(u) (u = x*scale + shift),
  // stores g instead of dA and emits the per-tile channel sums (sum g, sum g*xhat) through part_out -- the reduction pass of the
  // GroupNorm backward (gn_bwd_reduce_kernel: one more read of x, dA and the mask) inside the launch that produces dA.
  const float* gb_x0;                // null: off.  The forward GroupNorm's raw input (gb_x0: gb_C0 channels | gb_x1: Cout - gb_C0)
  const float* gb_x1;
  int gb_C0, gb_G, gb_plain;         // gb_G groups over Cout channels; gb_plain: GroupNorm only (no Swish)
  const float* gb_scale;             // [N][Cout] the forward's per-channel scale / shift (as gn_scale / gn_shift)
  const float* gb_shift;
  const float* gb_stats;             // [N][G][2] mean, rstd
  const unsigned char* gb_mask;      // [N][Hout][Wout][Cout] dropout keep bytes or null
  float gb_drop;                     // 1/(1-p)
  // GroupNorm statistics WITHOUT a finalisation launch (small grids, where a dependent launch costs ~5 us whatever it does -- 45 of the 139
  // launches of a B = 1 step were gn_finalize): a producer adds the (sum, sum of squares) of every channel PAIR of its output tile, as
  // fixed-point int64 (GSUM_BITS1 / GSUM_BITS2 fraction bits; integer adds commute, so a rerun is bitwise identical), to ITS XCD's
  // shard of the tensor's table gsum[n][xcd][pair][2] -- an atomic that the XCD's own L2 executes (every adder of a shard runs on that
  // XCD; the kernel boundary writes the lines back).  Agent-scope atomics, which execute at the memory side, serialised: 64 adders
  // per word took splitk_reduce from 7.4 to 17.2 us per launch (profiles/r06_b1_gn_consumer_agent_atomics_per_kernel.txt).  The consumer
  // (the MB = 2 instantiations of conv_k32_kernel, GNC) folds the shards and the pairs of the groups its K slice touches and forms
  // scale / shift itself, in an LDS table, in its prologue.  Pair granularity: a skip tensor is normalised under two groupings (alone,
  // and concatenated behind another tensor).
  unsigned long long* gsum_out;      // producer side: this launch's output tensor's table, or null
  const unsigned long long* gs0;     // consumer side: the tables of x0 / x1 (gs0 null: scale / shift come from gn_scale / gn_shift)
  const unsigned long long* gs1;
  const float* gs_gamma;             // [C0 + C1]
  const float* gs_beta;
  float gs_eps;
  int gs_G;                          // groups over the C0 + C1 concatenated channels
};
enum { GSUM_SHARDS = 8, GSUM_BITS1 = 16, GSUM_BITS2 = 12 };   // shards = XCDs
// producer epilogues: v = (sum, sumsq) of one channel pair over this workgroup's pixels -> the table (fdsr_act_io.h: gsum_add)

// PREC_F16 (round 6): ONE f16 product per multiply (the hi plane of the f16x3 weight forms, un-split activations) and f16
// activations in HBM -- the bf16 mode's kernels, bytes and MFMA rate with 11 mantissa bits instead of 8.
enum Precision { PREC_F32 = 0, PREC_F16X3 = 1, PREC_BF16 = 2, PREC_F16 = 3 };
constexpr bool prec_is16(int prec) { return prec == PREC_BF16 || prec == PREC_F16; }       // 2-byte activations, one MFMA per product
constexpr int prec_wform(int prec) { return prec == PREC_F16 ? PREC_F16X3 : prec; }         // the weight arena the mode reads (f16: the f16x3 form's hi plane)
constexpr int prec_act16(int prec) { return prec == PREC_BF16 ? 1 : (prec == PREC_F16 ? 2 : 0); }   // ConvParams::out_bf16 / act_bf16 codes: 0 fp32, 1 bf16, 2 f16

// Debug / A-B options of the launchers (fdsr_debug_option in include/fdsr.h).  Process-wide, never read from the
// environment: a library behind a C ABI must not change numerics paths because of a stray variable.  `epoch` moves with
// every change, so that cached workspace plans and captured graphs made under other settings are dropped.
struct Tunables {
  int rider = 2;            // ResnetBlock res_conv inside block2's launch: 0 never, 1 bandwidth-bound ones only, 2 always
  int up2 = 1;              // sub-pixel form of Upsample + conv3x3 (0: the generic folded-upsample kernel)
  long th_min_wgs = 256;    // conv tile rows: the largest tile that still gives this many workgroups
  int splitk = 1;           // split the K loop of small grids over workgroups
  int sk_target = 256;      // ... up to this many workgroups
  int wgrad_form = 0;       // f16x3 weight gradients: 0 default (8-wave in-row), 1 4-wave everywhere, 2 8-wave without the interleave
  int wgrad_colsum = 1;     // column sums of dy fused into the in-row weight-gradient kernel
  int wgrad_f32 = 0;        // f16x3 steps keep exact-fp32 weight gradients
  int drop_stage = 1;       // f16x3 training forwards: Dropout applied in the staging of the 16x16x32 kernels (0: the dropped activation is materialised first)
  int gnb_fuse = 1;         // f16x3 steps: the reduce half of the GroupNorm backward inside the input-gradient launch (ConvParams::gb_*)
  long long wgrad_big_bytes = 1ll << 32;   // tensors from this size on take the 4-wave weight-gradient kernel (64-bit offsets)
  int k32 = FDSR_K32_DEFAULT;           // v_mfma_f32_16x16x32 form (fdsr_conv_k32.hip) of the stride-1 3x3 launches that fit it; bits: 1 f16x3, 2 bf16, 4 the 16-row tile with a rider, 8 the 2-row-per-wave tiles of small grids, 16 the sub-pixel upsample convs, 32 the small-workgroup form (4 waves, two workgroups per CU; 6-row tiles in f16x3, 8-row tiles in bf16) of the rider-less 64-cout launches of large grids in f16x3, 128 in bf16 too, 64 the f16x3 launches with a rider too (rider chunks first), 512 the bf16 ones with a rider (off: slower), 1024 the 8-wave rider kernels with the rider chunks first (launches without a K split); 0 never
  int k32_stagger = 0;      // ... start delay of the CU's odd workgroup slot, in 64-cycle units per K chunk (0: none)
  long k32_sb_min_wgs = 1024;   // ... from this many workgroups on (a small grid wants all eight waves of a CU on its one tile)
  int strip = FDSR_STRIP_DEFAULT;           // column-strip form (fdsr_conv_strip.hip: weights in registers, one input row per step) of the 64-cout launches: bit 1 bf16 64 -> 64, 2 f16x3 64 -> 64, 4 (A/B) bf16 on one workgroup per CU, 8 bf16 (64 | 64) -> 64, 16 bf16 64 -> 64 with a res_conv rider, 32 bf16 (128 | 64) -> 64, 64 bf16 128 -> 128 / 64 -> 128
  long strip_min_wgs = 512; // ... from this many strip segments of >= 16 rows on (two workgroups per CU)
  int tail = 1;             // the input / output convs of the 16-bit modes on their own kernels (fdsr_conv_tail.hip); 0: the general ones
  int knockout = 0;         // TIMING-ONLY probes, results are garbage: bit 1 leaves the gn_finalize launches out, bit 2 the splitk_reduce launches
                            // (the upper bound of what fusing them into their producers could save; EXPERIMENTS round 4)
  int gn_consumer = 1;      // small grids: GroupNorm scale / shift formed in the consumer conv's prologue from the producers' fixed-point group sums (no gn_finalize launch)
  int bf16_f16x3_steps = 0; // PROBE (EXPERIMENTS R6): bf16 sampling runs the first n (n > 0) or the last -n (n < 0) reverse steps of the loop on the f16x3 kernels
  int sat_guard = 1;        // f16x3: sticky device flag when a RAW conv input exceeds the f16 range
  int drop_image_offset = 0;   // tests: the batch is images [offset, offset + N) of a larger one (its dropout masks follow)
  unsigned epoch = 0;
};
extern Tunables g_tun;
int set_tunable(const char* name, long long value);   // 0 ok, -1 unknown name

enum ConvKind { CONV3_S1 = 0, CONV3_S2 = 1, CONV3_UP = 2, CONV1 = 3 };

// KC (K-chunk) / BN (Cout tile) the launcher will use for this shape; the packer
// pads weights accordingly.
void conv_tile_config(ConvKind kind, int C0, int C1, int Cout, int* KC, int* BN);
hipError_t launch_conv(ConvKind kind, const ConvParams& p, hipStream_t s, int* tiles_per_image);
// Raise the dynamic-LDS limit of every kernel that needs more than 64 KB.  Must run
// once per process before any launch (and outside stream capture).
hipError_t kernels_init();

// 16-bit-operand MFMA convolutions (fp32-grade f16x3 split, or plain bf16).
void conv_h_config(ConvKind kind, int Cout, int* TH, int* WN);   // BN = 32*WN, K-chunk = 16
hipError_t launch_conv_h(ConvKind kind, int prec, const ConvParams& p, hipStream_t s, int* tiles_per_image);
bool conv_h_gnb_ok(ConvKind kind, int prec, const ConvParams& p);   // ConvParams::gb_* honoured by the kernel this launch lands on
bool conv_h_drop_ok(ConvKind kind, int prec, const ConvParams& p);  // ConvParams::drop_mask (train-mode dropout in the staging) honoured by a 16-bit launch
bool conv_k32_drop_ok(int prec, const ConvParams& q);
// K-loop split factor of a 16-bit conv launch (1 = none): a pure function of the shape, so the
// workspace planner and the launcher agree.  Only grids that would leave most of the 256 CUs idle split.
int conv_h_ksplit(ConvKind kind, int N, int Hout, int Wout, int Cout, int Cout_pad, int Cin_pad, int C0, int C1);
hipError_t kernels_h_init();
// the two ends of the UNet in the 16-bit modes (fdsr_conv_tail.hip): the 6(8)-channel input conv as a gather + 16x16x32 MFMA kernel
// without LDS, the <= 3-channel output conv as fp32 FMAs in scatter form; weights from the fp32 master copy [Cout][Cin][3][3]
bool conv_in8_ok(ConvKind kind, int prec, const ConvParams& p, int cin_real);
hipError_t launch_conv_in8(int prec, const ConvParams& p, const float* wmaster, int cin_real, hipStream_t s, int* tiles);
bool conv_out3_ok(ConvKind kind, int prec, const ConvParams& p);
hipError_t launch_conv_out3(int prec, const ConvParams& p, const float* wmaster, int cin_real, hipStream_t s, int* tiles);
hipError_t kernels_tail_init();
// K=32 MFMA form of the stride-1 3x3 launches (fdsr_conv_k32.hip): same ConvParams, same packed weights; launch_conv_h
// dispatches to it when conv_k32_ok() (wave tile of 4 x 32 pixels, whole 32-channel chunks on both sides of a concat seam).
bool conv_k32_ok(int TH, int WN, int prec, const ConvParams& p);
hipError_t launch_conv_k32(int TH, int WN, int prec, const ConvParams& q, int nwg, hipStream_t s);
// ... as 256-thread workgroups, two per CU, for the 64-cout launches of large grids (k32 bit 32): asked before a tile is picked
bool conv_k32_small_ok(ConvKind kind, int prec, const ConvParams& p);
// would launch_conv_h run this GroupNorm'd launch on a kernel that forms scale / shift itself (ConvParams::gs0)?  Asked by the engine
// before it decides to skip the gn_finalize launch; does this launch's kernel add its output's pair sums to ConvParams::gsum_out?
bool conv_h_gnc_ok(ConvKind kind, int prec, const ConvParams& p);
bool conv_h_gsum_ok(ConvKind kind, int prec, const ConvParams& p, bool sub_pixel_up2);
bool conv_k32_gnc_ok(int TH, int WN, int prec, const ConvParams& p);
hipError_t launch_conv_k32_small(int prec, const ConvParams& p, hipStream_t s, int* tiles);
hipError_t kernels_k32_init();
// Column-strip form of the 64-cout launches (fdsr_conv_strip.hip): same ConvParams, the same packed weights (wn_a: the WN they
// were packed for); asked before the small-workgroup form
bool conv_strip_ok(ConvKind kind, int prec, const ConvParams& p);
hipError_t launch_conv_strip(int prec, const ConvParams& p, int wn_a, hipStream_t s, int* tiles);
hipError_t kernels_strip_init();
// ... and of the sub-pixel upsample kernel (fdsr_conv_up2.hip): k32 bit 16
bool conv_up2_k32_ok(int prec, const ConvParams& p);
hipError_t launch_conv_up2_k32(int TH, int WN, int prec, const ConvParams& q, int nwg, hipStream_t s);
// Upsample(nearest x2)+Conv3x3 in sub-pixel form (fdsr_conv_up2.hip): four 2x2 convs on the source grid
// with pre-summed weights packed [cot][kc][wn][py][px*4+a*2+b][plane][lane] x 16 B.
hipError_t launch_conv_up2_h(int prec, const ConvParams& p, hipStream_t s, int* tiles_per_image);

// GroupNorm finalisation: per-tile per-channel partial sums (written by the producers' epilogues,
// fixed summation order => bitwise reproducible) of the virtual concat (x0: C0, x1: C1 channels)
// -> per-(n, channel) scale = rstd*gamma and shift = beta - mean*scale (biased variance, eps inside
// the sqrt: torch.nn.GroupNorm, reference unet.py:93).  Groups may straddle the concat seam.
struct GnFinalizeParams {
  const float* part0; int nt0, C0;
  const float* part1; int nt1, C1;
  const float* gamma; const float* beta;   // [C0+C1]
  float* scale; float* shift;              // [N][C0+C1]
  float* stats;                            // optional [N][G][2]: (mean, rstd) kept for the backward pass
  // optional FiLM of the guided-diffusion ResBlock (gdp_modules/unet.py:377-381, use_scale_shift_norm):
  // norm(h) * (1 + s) + t with s = film[n*film_stride + film_off + c], t = film[... + C + c], folded into scale / shift
  const float* film;
  int film_stride, film_off;
  int N, G, HW;
  float eps;
};
hipError_t launch_gn_finalize(const GnFinalizeParams& p, hipStream_t s);
// upper bound of tiles_per_image over every conv kernel variant (sizes the partial buffers)
int conv_max_tiles(int H, int W);

// noise-level embedding: PositionalEncoding -> Linear -> Swish -> Linear
// (unet.py:22-35, :242-248) and the per-ResnetBlock shift Linear(inner -> Cout)
// (unet.py:38-54) for all blocks at once.  nl_dev: [N] per-sample levels or null
// (then nl_scalar is used for every sample).  out temb[N][TE].
struct TembParams {
  const float* freq;  // [inner/2]   exp(-ln(1e4) * k/(inner/2))
  const float* w1;    // [4*inner][inner]
  const float* b1;
  const float* w2;    // [inner][4*inner]
  const float* b2;
  const float* wn;    // [TE][inner]   all noise_func weights concatenated
  const float* bn;    // [TE]
  const float* nl_dev;
  float nl_scalar;
  float* temb;        // [N][TE]
  int inner, TE, N;
  int swish_block;    // SR3 / GDP variants: per-block Linear applied to Swish(t)
  // dimensions (0: the FastDiffSR defaults enc = t = inner, hid = 4*inner); GDP: enc = model_channels, hid = t = 4*model_channels
  int enc_dim, hid_dim, t_dim;
  int cos_first;      // GDP timestep_embedding: cat([cos, sin]) (gdp_modules/unet.py:120-138) instead of [sin, cos]
};
hipError_t launch_temb(const TembParams& p, hipStream_t s);

// CLAM (unet.py:123-149): gate[N][C] = sigmoid(fc2(relu(fc1(avg))) + fc2(relu(fc1(max)))), pooled in
// FDSR_CLAM_SLICES pixel slices per image (phase 1) that phase 2 folds in order.
// scratch (clam_slam_scratch_floats): gate [N][C] | pool [N][SLICES][C][2] | map [N][2][HW]
#define FDSR_CLAM_SLICES 32
size_t clam_slam_scratch_floats(int N, int HW, int C);
hipError_t launch_clam_gate(const float* x, int N, int HW, int C, const float* fc1 /*[C/16][C]*/,
                            const float* fc2 /*[C][C/16]*/, int Cr, float* scratch, hipStream_t s, int act_bf16);
// SLAM applied to (x * gate) (unet.py:151-173): out = y * sigmoid(conv7x7([mean_c y, max_c y]))
// part_out (optional): [N][tiles][C][2] per-tile (2 x 32 pixels) partial (sum, sumsq) of out per channel.
hipError_t launch_slam(const float* x, float* scratch, const float* w7 /*[2][7][7]*/,
                       int N, int H, int W, int C, float* out, float* part_out, hipStream_t s, int* tiles_per_image,
                       int act_bf16 /* x and out are bf16 tensors (bf16 mode) */);

// Dropout keep-mask of one block: byte e = 1 with probability 1-p, a pure function of (seed, step, slot, e)
// (Philox4x32-10), so a run can be repeated and the mask inspected (fdsr_debug_dropout_mask).
hipError_t launch_dropout_mask(unsigned char* mask, size_t n, unsigned long long seed, unsigned step, unsigned slot, float p,
                               hipStream_t s, size_t first_elem = 0);

// layout changes at the boundary
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int N, int Csrc, int H, int W,
                               int Cdst, int c_off, int zero_rest, hipStream_t s);
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W,
                               int Csrc_stride, hipStream_t s);

// SelfAttention (n_head = 1) of the SR3 sibling on the NHWC qkv tensor [N][HW][3C]:
// S (scratch [N][HW][HW]) = softmax(Q K^T / sqrt(C)), O [N][HW][C] = S V.
// heads > 1 (GDP AttentionBlock, QKVAttentionLegacy, gdp_modules/unet.py:461-488): qkv channels are laid out
// [head][q | k | v][C/heads]; every head attends on its own, scores / sqrt(C/heads).  heads == 1: [q | k | v][C].
// act_bf16: qkv and O hold bf16 (the bf16 precision mode); scores / softmax stay fp32 either way
hipError_t launch_self_attention(const float* qkv, float* S, float* O, int N, int HW, int C, int heads, hipStream_t s, int act_bf16 = 0);
size_t attn_scratch_floats(int N, int HW, int heads);
// the first two thirds of the above (fp32 activations): S = softmax(Q K^T / sqrt(C / heads)), pitch round_up(HW, 16), pad columns zero --
// the backward recomputes the probabilities with the forward's own kernels
hipError_t launch_attn_probs(const float* qkv, float* S, int N, int HW, int C, int heads, hipStream_t s);
// 2x2 average pool of x (optionally of swish(x*scale + shift), the activated GroupNorm output) and nearest x2
// upsampling, materialised: the up/down ResBlocks of GDP resample h AND the skip input (gdp_modules/unet.py:369-376)
hipError_t launch_pool2(const float* x, const float* gn_scale, const float* gn_shift, float* out, int N, int H, int W, int C,
                        hipStream_t s, int act_bf16 = 0);   // x [N,H,W,C] -> out [N,H/2,W/2,C]
hipError_t launch_upsample2(const float* x, float* out, int N, int H, int W, int C, hipStream_t s, int act_bf16 = 0);   // -> [N,2H,2W,C]

// tensor2img of the val loop (core/metrics.py:16-42): NCHW fp32 -> HWC uint8
hipError_t launch_tensor2img_u8(const float* src, unsigned char* dst, int N, int C, int H, int W, float lo, float hi,
                                hipStream_t s);

// the val loop's per-image metric sums on uint8 images and the dataset's tensor transform (fdsr_val.hip)
size_t image_metrics_workspace_bytes(int N, int H, int W);
hipError_t launch_image_metrics_u8(const unsigned char* a, const unsigned char* b, int N, int H, int W, int C, int flags, double* out,
                                   void* ws, hipStream_t s);
hipError_t launch_u8_to_tensor(const unsigned char* src, float* dst, int N, int C, int H, int W, float lo, float hi, hipStream_t s);

// PIL-exact 8-bit bicubic resize (two passes) + uint8 -> model tensor; tables built on the host
hipError_t launch_resize_bicubic_u8(const unsigned char* src, unsigned char* tmp, unsigned char* dst_u8, float* dst_f32, int N,
                                    int h, int w, int H, int W, const int* bounds_x, const int* kk_x, int ksize_x,
                                    const int* bounds_y, const int* kk_y, int ksize_y, hipStream_t s);

// one reverse-diffusion update (diffusion.py:157-190) on the packed state tensor
// xin [N,H,W,CP]: channels [0,3) = cond, [3,6) = x_t.
struct PosteriorParams {
  const float* eps;    // [N,H,W,3]
  float* xin;          // [N,H,W,CP]
  const float* noise;  // [N,3,H,W] NCHW or null (t == 0, or engine RNG)
  const unsigned long long* rng;   // device {seed, call counter} or null: draw N(0,1) in the kernel (Philox4x32-10)
  int rng_plane;                   // which of the T noise planes this step draws (the k of noise[k])
  float* traj;         // [N,3,H,W] NCHW or null: x_{t-1}
  float* out;          // [N,3,H,W] NCHW or null: res2img(x_0, cond) at the last step
  int N, HW, CP;
  float c_recip, c_recipm1, coef1, coef2, sigma;
  int plain_out;       // SR3 variant: out = x_0 itself (no res2img)
  int x_off;           // first channel of x_t inside xin (3: cat[cond, x]; GDP: 0, cat[x, cond], gdp_modules/diffusion.py:191)
  int x0_pred;         // GDP: the network predicts x_0 itself (clamped), not the noise (:190-195)
};
hipError_t launch_posterior(const PosteriorParams& p, hipStream_t s);

// Engine-side noise (fdsr_sample with noise == NULL): counter-based Philox4x32-10 + Box-Muller, keyed by
// (seed, call counter) and indexed by (plane, pixel): the values do not depend on the launch geometry.
hipError_t launch_rng_advance(unsigned long long* rng, hipStream_t s);                      // ++call counter
// plane `plane` of the noise the engine would use, as [N,3,H,W] fp32 (tests) ...
hipError_t launch_randn_plane(const unsigned long long* rng, float* dst_nchw, int N, int HW, int plane, hipStream_t s);
// ... and x_T = plane 0 written straight into channels 3..5 of the packed UNet input
hipError_t launch_randn_xin(const unsigned long long* rng, float* xin, int N, int HW, int CP, hipStream_t s, int c_off = 3);

}  // namespace fdsr
