// Launch interfaces of the training-step kernels (fdsr_train.hip): the backward pass of the FastDiffSR UNet,
// the loss, Adam and the device-side re-packing of updated weights.  Internal header (not the C ABI).
//
// Reference: DDPM.optimize_parameters (model/model.py:47-57) = zero_grad, l_pix = netG(data) (p_losses,
// model/fastdiffsr_modules/diffusion.py:242-270), l_pix.sum() / (b*c*h*w), backward, Adam.step.
// All arithmetic here is fp32 (accumulations that cross many pixels in fp64), reductions in a fixed order:
// a training step is bitwise reproducible like the sampling path.
#pragma once
#include "fdsr_kernels.h"

namespace fdsr {

// ---- loss -------------------------------------------------------------------------------------------------
// eps [N,H,W,3] NHWC (the UNet output), target [N,3,H,W] NCHW (the noise q_sample mixed in):
//   loss = sum |target - eps| (l1, nn.L1Loss(reduction='sum')) or sum (target - eps)^2 (l2), one fp32 scalar
//   deps [N,H,W,8] (channels 3..7 zero): d(loss * scale) / d eps
// partial: scratch of >= loss_partial_floats(N*HW) doubles-as-floats pairs.
size_t loss_partial_count(size_t npix);
hipError_t launch_loss_grad(const float* eps, const float* target, float* deps8, double* partial, float* loss_out, int N, int HW,
                            int l2, float scale, hipStream_t s);

// ---- network input of a training step: xin [N,H,W,CP] = [SR | gamma * img2res(HR, SR) + sqrt(1 - gamma^2) * noise | 0 ...] from NCHW
// HR, SR, noise and the per-sample gamma [N] (diffusion.py:233-263); the arithmetic of the op-by-op torch tensor
hipError_t launch_qsample_pack(const float* hr, const float* sr, const float* gamma, const float* noise, float* xin, int N, int HW,
                               int CP, hipStream_t s);

// ---- small helpers -----------------------------------------------------------------------------------------
// S[n][c] = sum over the pixels of image n of dy[n][p][c]   (bias / noise-shift gradients); fixed order
hipError_t launch_colsum(const float* dy, float* S, double* scratch, int N, int HW, int C, hipStream_t s);
size_t colsum_scratch_doubles(int N, int HW, int C);
// x[i] *= f
hipError_t launch_scale_inplace(float* x, size_t n, float f, hipStream_t s);
// dst[tab[3e+1] + i] = src[tab[3e] + i] for i < tab[3e+2], e < entries (many small tensors in one launch)
hipError_t launch_copy_table(const float* src, float* dst, const unsigned long long* tab, int entries, hipStream_t s, size_t max_elems = 0);   // max_elems: the largest entry (sizes the grid)
// out[c] = sum_n S[n*stride + c], c < C, in image order
hipError_t launch_sum_rows(const float* S, int N, int stride, int C, float* out, hipStream_t s);
// z [N,2H,2W,C] = dy [N,H,W,C] at the even positions, zero elsewhere (the stride-2 conv's transpose)
hipError_t launch_zero_insert(const float* dy, float* z, int N, int H, int W, int C, hipStream_t s);
// dx [N,H,W,C] += (assign: =) sum of the 2x2 block of du [N,2H,2W,C]   (nearest x2 upsampling, transposed)
hipError_t launch_pool2_add(const float* du, float* dx, int N, int H, int W, int C, bool assign, hipStream_t s);
// dx [N,2H,2W,C] += (assign: =) scale * dy [N,H,W,C] at each of the four pixels of its 2x2 block   (2x2 average pooling, transposed:
// scale = 0.25; the GDP sibling's down ResBlocks, gdp_modules/unet.py:369-375)
hipError_t launch_unpool2_add(const float* dy, float* dx, int N, int H, int W, int C, float scale, bool assign, hipStream_t s);
// dst[n][p][0..C) += (assign: =) src[n][p][off .. off+C) of a tensor with Cs channels (routes a concat half)
hipError_t launch_add_slice(const float* src, float* dst, size_t npix, int Cs, int off, int C, bool assign, hipStream_t s);

// ---- GroupNorm + Swish backward ------------------------------------------------------------------------------
// a = swish(u), u = x * scale + shift  (scale = rstd*gamma, shift = beta - mean*scale; x = virtual concat x0|x1)
// Given dA [N,HW,C]: g = dA * swish'(u);  dgamma[c] = sum g*xhat, dbeta[c] = sum g,
// dx = rstd * (gamma*g - mean_grp(gamma*g) - xhat * mean_grp(gamma*g*xhat)), accumulated into dx0 / dx1.
struct GnBwdParams {
  const float* dA;                // [N][HW][C0+C1]
  const float* x0; const float* x1;
  int C0, C1;
  const float* scale; const float* shift;   // [N][C]
  const float* stats;             // [N][G][2] (mean, rstd)
  const float* gamma;             // [C]
  float* dx0; float* dx1;         // accumulated (+=); dx1 may be null when C1 == 0
  int assign0, assign1;           // 1: this is the first contribution to that gradient tensor, store instead of accumulate
  float* dgamma; float* dbeta;    // [C], written (=)
  double* scratch;                // gn_bwd_scratch_doubles()
  int N, HW, H, G;                // H: rows of the map (HW / H columns)
  int plain;                      // 1: GroupNorm only (no Swish)
  const unsigned char* drop_mask; // train-mode dropout after the Swish (block2): keep bytes [N][HW][C0] or null
  float drop_scale;               // 1 / (1 - p)
  // The input-gradient launch already did the first half (ConvParams::gb_*, fdsr_conv_k32.hip): dA then holds g itself and g_part the
  // per-tile channel sums [N][g_nt][C][2] (sum g, sum g*xhat) -- the reduce pass is left out, the apply pass reads neither the mask nor
  // scale / shift.  g_part lives at the head of `scratch` (gn_bwd_tile_part()).
  const float* g_part;            // null: dA is the raw gradient w.r.t. the activated input
  int g_nt;
  // scale-shift GroupNorm of the GDP sibling's ResBlocks (gdp_modules/unet.py:377-381): u = (xhat*gamma + beta) * (1 + s) + t with
  // (s, t) = film[n*film_stride + c], film[n*film_stride + C + c] (two runs of C columns of the step's [N][TE] embedding table; scale
  // and shift above already hold the folded form).  The effective gamma of image n is gamma*(1 + s); dgamma / dbeta sum (1 + s) times
  // the per-image sums; dfilm (same layout, stride dfilm_stride) receives ds = gamma*sum(g*xhat) + beta*sum(g), dt = sum(g).
  const float* film; int film_stride;
  const float* beta;              // [C], read with film only
  float* dfilm; int dfilm_stride;
};
size_t gn_bwd_scratch_doubles(int N, int H, int W, int C);
inline float* gn_bwd_tile_part(double* scratch) { return reinterpret_cast<float*>(scratch); }   // room for ceil(W/32) * ceil(H/2) tiles per image
hipError_t launch_gn_bwd(const GnBwdParams& p, hipStream_t s);

// ---- convolution weight gradient -----------------------------------------------------------------------------
// dW[co][ci][tap] = sum_{n,p} dy[n][p][co] * a[n][p (+) tap][ci], a = the conv's (virtually concatenated,
// optionally GroupNorm+Swish-activated, optionally nearest-x2 upsampled) input exactly as the forward staged it.
// Exact fp32 on v_mfma_f32_32x32x2_f32; the pixel range is split over workgroups, slices are summed in order.
struct WgradParams {
  const float* dy;                 // [N][Hout][Wout][Cout_s]  (Cout_s = channel stride of dy, >= Cout)
  const float* x0; const float* x1;
  const float* gn_scale; const float* gn_shift;   // [N][C0+C1] or null
  int gn_plain;
  float* dw;                       // [Cout][Cin][ks][ks] (checkpoint layout), written (=)
  float* scratch;                  // wgrad_scratch_floats()
  int N, Hin, Win, Hout, Wout;     // Hin/Win: source dims (before upsampling)
  int C0, C1, Cin_real;            // Cin_real: channels of the weight tensor (the packed input conv: 6 of 8)
  int Cout, Cout_s;
  const unsigned char* drop_mask;  // as ConvParams::drop_mask (the activated input is a * keep * drop_scale)
  float drop_scale;
  float* colsum;                   // launch_wgrad_h only, when wgrad_h_fuses_colsum(): S [N][Cout_s] = per-image column sums of dy
  float* colsum_part;              // (set by the launcher: per-slice partial sums in the scratch)
};
size_t wgrad_scratch_floats(ConvKind kind, int N, int Hout, int Wout, int Cin, int Cout);
// true when launch_wgrad_h(kind, p) will also write p.colsum (the bias / noise-shift gradient sums launch_colsum computes)
bool wgrad_h_fuses_colsum(ConvKind kind, const WgradParams& p);
hipError_t launch_wgrad(ConvKind kind, const WgradParams& p, hipStream_t s);
// the same in the split-f16 form (three f16 MFMAs per product, fp32 accumulate): FDSR_PREC_F16X3 training steps
hipError_t launch_wgrad_h(ConvKind kind, const WgradParams& p, hipStream_t s);
hipError_t train_kernels_init();

// ---- CLAM / SLAM backward (unet.py:123-173) -----------------------------------------------------------------
// forward: gate = sigmoid(fc2 relu(fc1 avg) + fc2 relu(fc1 max)), y = x*gate, m = [mean_c y, max_c y],
//          out = y * sigmoid(conv7x7(m))
struct ClamSlamBwdParams {
  const float* x;          // [N][HW][C] input of the pair (the ResnetBlock output)
  const float* dout;       // [N][HW][C]
  float* dx;               // [N][HW][C] accumulated (+=)
  const float* fc1; const float* fc2; const float* w7;   // [Cr][C], [C][Cr], [2][7][7]
  float* dfc1; float* dfc2; float* dw7;                  // written (=)
  float* scratch;          // clam_slam_bwd_scratch_floats()
  int N, H, W, C, Cr;
};
size_t clam_slam_bwd_scratch_floats(int N, int HW, int C, int Cr);
hipError_t launch_clam_slam_bwd(const ClamSlamBwdParams& p, hipStream_t s);

// ---- noise-level embedding backward (unet.py:22-54, :242-248) ---------------------------------------------------
// forward: enc = [sin, cos](nl * freq); hid = swish(W1 enc + b1); t = W2 hid + b2; temb = Wn t + bn
// dtemb [N][TE] = per-(image, channel) sums of the block1 output gradients.
// SR3 sibling (ddpm_modules/unet.py:19-34, :81-94, :163-170): the same chain on the integer time, with the per-block Linear applied to
// Swish(t): temb = Wn swish(t) + bn (swish_block).
struct TembBwdParams {
  const float* freq; const float* w1; const float* b1; const float* w2; const float* b2; const float* wn;
  const float* nl;         // [N]
  const float* dtemb;      // [N][TE]
  float* dw1; float* db1; float* dw2; float* db2; float* dwn; float* dbn;   // written (=)
  float* scratch;          // N * temb_bwd_scratch_floats_per_image() floats: [N] records, then the row blocks' partial sums [N][blocks][t_dim]
  int inner, TE, N;
  int swish_block;
  // GDP sibling (gdp_modules/unet.py:120-138, :588-593, :336-342): timestep_embedding(enc_dim) = cat([cos, sin]) -> Linear(enc_dim, hid_dim)
  // -> SiLU -> Linear(hid_dim, t_dim); every per-block Linear(t_dim, 2*Cout) sees SiLU(t).  Zero: the widths above (inner, 4 inner, inner).
  int enc_dim, hid_dim, t_dim, cos_first;
};
size_t temb_bwd_scratch_floats_per_image(int inner, int enc_dim, int hid_dim, int t_dim, int TE);
hipError_t launch_temb_bwd(const TembBwdParams& p, hipStream_t s);

// ---- SelfAttention backward (SR3 sibling: ddpm_modules/unet.py:99-127, n_head = 1; GDP sibling: QKVAttentionLegacy,
// gdp_modules/unet.py:461-488, heads of C / heads channels laid out [head][q | k | v] along the channel axis) ----------------------
// forward (fdsr_kernels.hip): S = Q K^T / sqrt(C), P = softmax_rows(S), O = P V on the NHWC qkv tensor [N][HW][q | k | v].
// backward: P is recomputed by the forward's kernels; dP = dO V^T; dS = P (dP - rowsum(dP P)); dQ = dS K / sqrt(C);
// dK = dS^T Q / sqrt(C); dV = P^T dO -- five fp32-MFMA products of a few MFLOP, every output element written by exactly one lane
// (no atomics: a rerun is bitwise identical).
struct AttnBwdParams {
  const float* qkv;        // [N][HW][3C]
  const float* dO;         // [N][HW][C]
  float* dqkv;             // [N][HW][3C], written (=)
  float* scratch;          // attn_bwd_scratch_floats(): P | dP -> dS
  int N, HW, C;
  int heads;               // 0 or 1: one head over all C channels
};
size_t attn_bwd_scratch_floats(int N, int HW, int heads = 1);
hipError_t launch_attn_bwd(const AttnBwdParams& p, hipStream_t s);

// ---- optimiser -----------------------------------------------------------------------------------------------
// torch.optim.Adam (defaults of model.py:37-38: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; w -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
hipError_t launch_adam(float* w, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step,
                       hipStream_t s);
// checkpoint layout [Cout][Cin][ks][ks] -> the fp32 kernel's [tap][Cout_pad][Cin_pad] (zero padded)
hipError_t launch_pack_conv_f32(const float* w, float* packed, int Cout, int Cin, int ks, int cout_pad, int cin_pad, hipStream_t s);
// ... and the transposed, tap-flipped form the input-gradient convolution runs on:
// packed_t[tap][ci][co] = w[co][c_off + ci][flip(tap)], ci < Csub: [tap][round_up(Csub,BN)][round_up(Cout,KC)]
hipError_t launch_pack_conv_f32_t(const float* w, float* packed_t, int Cout, int Cin, int ks, int c_off, int Csub, int rows_pad,
                                  int cols_pad, hipStream_t s);

// ---- f16x3 forms on the device (fp32-grade split-f16 MFMA convolutions, fdsr_conv_h.hip) ------------------------
// Per-tensor power-of-two scale of the f16x3 forms, e = min(12, floor(log2(32768 / max|w|))) (what pack_weights_h chooses on
// the host): launch_hamax folds max|w| of one tensor into amax_bits[slot] (bit pattern of a non-negative float; atomicMax
// is order-independent, so this stays deterministic), launch_hscale_all then writes scale2[slot] = {2^e, 2^-e} for every slot.
hipError_t launch_hamax(const float* w, size_t n, unsigned* amax_bits, hipStream_t s);
hipError_t launch_hscale_all(const unsigned* amax_bits, float* scale2, int nslots, hipStream_t s);
// checkpoint layout [Cout][Cin][ks][ks] -> MFMA B-fragment order [cot][kc][wn][tap][hi|lo][lane] x 16 B (see
// pack_weights_h in fdsr_engine.cpp), values multiplied by scale2[0].  transposed != 0: the input-gradient form, the
// conv weight  W'[co' = ci - c_off][ci' = co][tap'] = W[co][ci][T-1-tap'],  co' < rows (the Cin slice of one concat source)
hipError_t launch_pack_conv_h(const float* w, void* frags, const float* scale2, int Cout, int Cin, int ks, int WN, int cout_pad,
                              int cin_pad, int transposed, int c_off, int rows, hipStream_t s);
// the sub-pixel form of an upsample conv (fdsr_conv_up2.hip) from the fp32 master weights, scale = scale2[0] / 4; *inv_out = its inverse
hipError_t launch_pack_conv_up2_h(const float* w, void* frags, const float* scale2, float* inv_out, int Cout, int Cin, int WN, int cout_pad,
                                  int cin_pad, hipStream_t s);
// a[n][p][c] = swish(x*scale + shift) * keep * drop_scale: the dropped activation of block2 materialised, for the f16x3
// forward (the 16-bit conv kernel then reads it raw; its staging has no spare registers for the mask)
hipError_t launch_gn_silu_drop(const float* x, const float* gn_scale, const float* gn_shift, const unsigned char* mask, float drop_scale,
                               float* out, int N, int HW, int C, hipStream_t s);

}  // namespace fdsr
