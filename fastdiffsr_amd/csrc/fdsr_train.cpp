// Training step of the FastDiffSR UNet on the device: forward in "keep" mode, loss, backward
// pass, Adam, re-packing of the updated weights.  Reference: DDPM.optimize_parameters (model/model.py:47-57),
// GaussianDiffusion.p_losses (model/fastdiffsr_modules/diffusion.py:242-270), UNet.forward (unet.py:299-323).
// The siblings' plans (ddpm_modules, tesr_modules, gdp_modules) walk the same loop; their own ops -- the
// attention core, GDP's pooled / upsampled ResBlocks and scale-shift GroupNorms -- are cases of it below.
//
// The backward walks the forward plan in reverse.  For a convolution op  y = conv(a) + b + shift (+ res),
// a = swish(gn(x)) or a = x, x = cat(x0, x1):
//   * d res  += dy                                  (identity residual; a 1x1 res_conv shares the buffer)
//   * S[n][c] = sum_p dy                            -> bias gradient, and the noise-embedding gradient of the block
//   * dW      = wgrad(dy, a)                        (fdsr_train.hip, exact fp32 MFMA)
//   * dA      = conv(dy; W transposed, taps flipped) on the FORWARD kernel (stride 2: zero-inserted dy;
//               upsample: 2x2 sum-pool afterwards)
//   * dx0/dx1 += GroupNorm+Swish backward(dA)       or += dA where the conv reads its input raw
// The first contribution to a gradient tensor is a store, later ones accumulate, all in stream order: deterministic.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "fdsr_engine_int.h"
#include "fdsr_train.h"

using namespace fdsr;
using namespace fdsr_int;

namespace {

struct TrainPlan {
  size_t base = 0;                      // first byte after the forward plan
  std::vector<size_t> grad_off;         // per tensor: its gradient [N][HW][C] (eps: 8 channels)
  size_t grad_bytes = 0;                // all gradient tensors (one memset)
  size_t off_tmpA = 0, off_tmpZ = 0, off_S = 0, off_dtemb = 0, off_dwn = 0, off_dbn = 0, off_dbl = 0, off_wg = 0, off_csb = 0,
         off_tb = 0, off_loss = 0, off_deps = 0, off_noise = 0, off_attn = 0;
  size_t bytes = 0;                     // total workspace (forward plan + extras)
};

int tensor_channels(fdsr_handle h, int t) { return t == h->t_eps ? 8 : h->tensors[t].C; }

int make_train_plan(fdsr_handle h, int N, int H, int W, TrainPlan* tp) {
  const ShapePlan& sp = h->plan;
  size_t off = align_up(sp.bytes, 256);
  tp->base = off;
  tp->grad_off.assign(h->tensors.size(), 0);
  const size_t g0 = off;
  for (size_t t = 0; t < h->tensors.size(); ++t) {
    const TensorDesc& td = h->tensors[t];
    if (td.C < 0 || (int)t == h->t_in) continue;       // attention scratch; the network input needs no gradient
    const size_t hw = (size_t)(H >> td.level) * (W >> td.level);
    tp->grad_off[t] = off;
    off += align_up((size_t)N * hw * tensor_channels(h, (int)t) * sizeof(float), 256);
  }
  tp->grad_bytes = off - g0;
  size_t tmpA = 256, tmpZ = 256, wg = 256, csb = 256, dbl = 256, attn = 256;
  int maxC = 8;
  for (const Op& op : h->ops) {
    if (op.kind == Op::ATTN)
      attn = std::max(attn, attn_bwd_scratch_floats(N, (H >> op.lvl_in) * (W >> op.lvl_in), op.heads) * sizeof(float));
    if (op.kind == Op::POOL2 && op.gn_slot >= 0) {   // GDP down ResBlock: the pooled gradient spread back to the fine grid, then GroupNorm backward
      const int Hi = H >> op.lvl_in, Wi = W >> op.lvl_in;
      tmpA = std::max(tmpA, (size_t)N * Hi * Wi * op.C0 * sizeof(float));
      dbl = std::max(dbl, gn_bwd_scratch_doubles(N, Hi, Wi, op.C0) * sizeof(double));
      maxC = std::max(maxC, op.C0);
    }
    if (op.kind == Op::SLAM) {
      const int hw = (H >> op.lvl_in) * (W >> op.lvl_in);
      csb = std::max(csb, clam_slam_bwd_scratch_floats(N, hw, op.C0, op.C0 / 16) * sizeof(float));
    }
    if (op.kind != Op::CONV) continue;
    const int Ho = H >> op.lvl_out, Wo = W >> op.lvl_out, Hi = H >> op.lvl_in, Wi = W >> op.lvl_in;
    const int Cin = op.C0 + op.C1;
    maxC = std::max(maxC, std::max(Cin, op.Cout));
    // dA at the grid the taps walk on (fine grid for stride 2 and for the upsample conv)
    const size_t fine = (size_t)(op.ck == CONV3_UP ? Ho * Wo : Hi * Wi);
    tmpA = std::max(tmpA, (size_t)N * fine * Cin * sizeof(float));
    if (op.ck == CONV3_S2) tmpZ = std::max(tmpZ, (size_t)N * Hi * Wi * op.Cout * sizeof(float));
    if (op.ck == CONV3_UP && op.gn_slot >= 0) tmpZ = std::max(tmpZ, (size_t)N * Hi * Wi * Cin * sizeof(float));   // GDP up ResBlock
    wg = std::max(wg, wgrad_scratch_floats(op.ck, N, Ho, Wo, Cin, op.Cout) * sizeof(float));
    dbl = std::max(dbl, colsum_scratch_doubles(N, Ho * Wo, std::max(op.Cout, 8)) * sizeof(double));
    if (op.gn_slot >= 0) dbl = std::max(dbl, gn_bwd_scratch_doubles(N, Hi, Wi, Cin) * sizeof(double));
  }
  dbl = std::max(dbl, loss_partial_count((size_t)N * H * W) * sizeof(double));
  auto take = [&](size_t b) { const size_t r = off; off += align_up(b, 256); return r; };
  tp->off_tmpA = take(tmpA);
  tp->off_tmpZ = take(tmpZ);
  tp->off_S = take((size_t)N * maxC * sizeof(float));
  tp->off_dtemb = take((size_t)N * h->TE * sizeof(float));
  const int ic = h->cfg.inner_channel, enc_dim = h->gdp ? ic : 0, hid_dim = h->gdp ? 4 * ic : 0, t_dim = h->gdp ? 4 * ic : 0;
  tp->off_dwn = take((size_t)h->TE * (t_dim ? t_dim : ic) * sizeof(float));
  tp->off_dbn = take((size_t)h->TE * sizeof(float));
  tp->off_dbl = take(dbl);
  tp->off_wg = take(wg);
  tp->off_csb = take(csb);
  tp->off_tb = take((size_t)N * temb_bwd_scratch_floats_per_image(ic, enc_dim, hid_dim, t_dim, h->TE) * sizeof(float));
  tp->off_loss = take(256);
  tp->off_attn = take(attn);
  tp->off_noise = off;                  // the target noise of a step whose noise the engine draws itself (fdsr_train_grads_pairs)
  off += align_up((size_t)N * 3 * H * W * sizeof(float), 256);
  tp->bytes = off;
  return FDSR_OK;
}

// transposed weights: per conv weight, per concat source (fdsr_train.hip: pack_conv_f32_t)
struct WtDims { int rows_pad, cols_pad; };
WtDims wt_dims(ConvKind ck, int K, int Csub) {
  int KC, BN;
  const ConvKind k = ck == CONV1 ? CONV1 : CONV3_S1;     // the transposed conv always runs at stride 1
  conv_tile_config(k, K, 0, Csub, &KC, &BN);
  return WtDims{round_up(Csub, BN), round_up(K, KC)};
}

int conv_K(fdsr_handle h, const Op& op) { return op.dst == h->t_eps ? 8 : op.Cout; }   // channels of dy as stored

int train_prepare(fdsr_handle h) {
  if (h->train_ready) return FDSR_OK;
  // FastDiffSR; the two siblings built from the same blocks plus SelfAttention: SR3 (ddpm_modules: integer time, Swish in front of the
  // per-block Linear) and TESR; and GDP (gdp_modules: scale-shift GroupNorms, pooled / nearest-upsampled ResBlocks, heads of 64 channels).
  for (const Op& op : h->ops)
    if (op.kind == Op::CONV && op.src0 != h->t_in && (conv_K(h, op) % 8 || (op.C0 % 16) || (op.C1 % 16)))
      return fail(h, FDSR_E_INVALID, "training needs channel counts that are multiples of 16 (layer %s)", op.name.c_str());
  const size_t nb = std::max<size_t>(h->master_floats, 4) * sizeof(float);
  HIPCHK(h, hipMalloc((void**)&h->d_grad, nb));
  HIPCHK(h, hipMalloc((void**)&h->d_adam_m, nb));
  HIPCHK(h, hipMalloc((void**)&h->d_adam_v, nb));
  HIPCHK(h, hipMemset(h->d_grad, 0, nb));
  HIPCHK(h, hipMemset(h->d_adam_m, 0, nb));
  HIPCHK(h, hipMemset(h->d_adam_v, 0, nb));
  h->adam_t = 0;
  // transposed-weight arena
  h->wt_off0.assign(h->weights.size(), SIZE_MAX);
  h->wt_off1.assign(h->weights.size(), SIZE_MAX);
  size_t off = 0;
  int maxC = 8;
  for (const Op& op : h->ops) {
    if (op.kind != Op::CONV || op.src0 == h->t_in) continue;
    const int T = op.ck == CONV1 ? 1 : 9, K = conv_K(h, op);
    maxC = std::max(maxC, op.C0 + op.C1);
    const bool whole = op.gn_slot >= 0;                   // GroupNorm'ed input: one transposed conv over all input channels
    if (whole || op.C1 == 0) {
      const WtDims d = wt_dims(op.ck, K, op.C0 + op.C1);
      h->wt_off0[op.w] = off;
      off += align_up((size_t)T * d.rows_pad * d.cols_pad, 64);
    } else {                                              // raw concat input: one transposed conv per source
      const WtDims d0 = wt_dims(op.ck, K, op.C0), d1 = wt_dims(op.ck, K, op.C1);
      h->wt_off0[op.w] = off;  off += align_up((size_t)T * d0.rows_pad * d0.cols_pad, 64);
      h->wt_off1[op.w] = off;  off += align_up((size_t)T * d1.rows_pad * d1.cols_pad, 64);
    }
  }
  h->wt_floats = off;
  // the same forms as f16x3 MFMA fragments, for the convs whose transposed shape the 16-bit kernels take
  // (16-channel K chunks: K % 16 == 0 and the produced channel count % 16 == 0)
  h->wtq_off0.assign(h->weights.size(), SIZE_MAX);
  h->wtq_off1.assign(h->weights.size(), SIZE_MAX);
  size_t qoff = 0;
  for (const Op& op : h->ops) {
    if (op.kind != Op::CONV || op.src0 == h->t_in) continue;
    const int T = op.ck == CONV1 ? 1 : 9, K = conv_K(h, op);
    if (K % 16 || op.C0 % 16 || op.C1 % 16) continue;
    auto frag_bytes = [&](int rows) {
      int TH, WN;
      conv_h_config(op.ck == CONV1 ? CONV1 : CONV3_S1, rows, &TH, &WN);
      return align_up((size_t)(round_up(rows, 32 * WN) / 32) * (round_up(K, 16) / 16) * T * 64 * 16 * 2, 256);
    };
    if (op.gn_slot >= 0 || op.C1 == 0) {
      h->wtq_off0[op.w] = qoff;  qoff += frag_bytes(op.C0 + op.C1);
    } else {
      h->wtq_off0[op.w] = qoff;  qoff += frag_bytes(op.C0);
      h->wtq_off1[op.w] = qoff;  qoff += frag_bytes(op.C1);
    }
  }
  h->wtq_bytes = qoff;
  HIPCHK(h, hipMalloc((void**)&h->d_wtq, std::max<size_t>(qoff, 256)));
  HIPCHK(h, hipMalloc((void**)&h->d_hamax, h->weights.size() * sizeof(unsigned)));
  HIPCHK(h, hipMalloc((void**)&h->d_up2_inv, h->weights.size() * sizeof(float)));
  HIPCHK(h, hipMalloc((void**)&h->d_wt, std::max<size_t>(off, 4) * sizeof(float)));
  HIPCHK(h, hipMalloc((void**)&h->d_zero, (size_t)round_up(maxC, 64) * sizeof(float)));
  HIPCHK(h, hipMemset(h->d_zero, 0, (size_t)round_up(maxC, 64) * sizeof(float)));
  {   // the non-conv tensors (GroupNorm affine, biases, MLPs ...) follow the master copy through ONE table-driven copy
    std::vector<unsigned long long> tab;
    for (int i = 0; i < h->n_schema; ++i) {
      const WeightEntry& w = h->weights[i];
      if (!w.live || w.sink == WeightEntry::CONV_PACK) continue;
      tab.push_back(h->master_off[i]);
      tab.push_back(w.dev_off);
      tab.push_back(numel(w.shape));
      h->copy_tab_max = std::max<size_t>(h->copy_tab_max, numel(w.shape));
    }
    h->n_copy_tab = (int)(tab.size() / 3);
    HIPCHK(h, hipMalloc((void**)&h->d_copy_tab, std::max<size_t>(tab.size(), 3) * sizeof(unsigned long long)));
    if (!tab.empty()) HIPCHK(h, hipMemcpy(h->d_copy_tab, tab.data(), tab.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
  }
  HIPCHK(h, train_kernels_init());
  h->train_ready = true;
  return FDSR_OK;
}

// (re-)build every device form the fp32 kernels read from the master copy: after load and after each Adam step
int repack_from_master(fdsr_handle h, hipStream_t st, bool forward_forms, bool all_f32_forms) {
  // In f16x3 mode a step reads the fp32 conv forms only where the 16-bit kernels cannot run (the packed-input conv, odd
  // shapes): the others are refreshed when something asks for them (ensure_f32_forms: fp32 mode, fp32 sampling).
  const bool lazy = h->prec == PREC_F16X3 && !all_f32_forms;
  bool skipped = false;
  if (forward_forms) {
    for (int i = 0; i < h->n_schema; ++i) {
      WeightEntry& w = h->weights[i];
      if (!w.live || w.sink != WeightEntry::CONV_PACK) continue;
      if (lazy && w.h_ok) { skipped = true; continue; }
      HIPCHK(h, launch_pack_conv_f32(h->d_master + h->master_off[i], h->d_params + w.dev_off, (int)w.shape[0], (int)w.shape[1], w.ks,
                                     w.cout_pad, w.cin_pad, st));
    }
    HIPCHK(h, launch_copy_table(h->d_master, h->d_params, h->d_copy_tab, h->n_copy_tab, st, h->copy_tab_max));
  }
  for (const Op& op : h->ops) {
    if (op.kind != Op::CONV || op.src0 == h->t_in) continue;
    if (lazy && h->wtq_off0[op.w] != SIZE_MAX) { skipped = true; continue; }
    const WeightEntry& w = h->weights[op.w];
    const float* src = h->d_master + h->master_off[op.w];
    const int K = conv_K(h, op), Cout = (int)w.shape[0], Cin = (int)w.shape[1];
    if (h->wt_off1[op.w] == SIZE_MAX) {
      const WtDims d = wt_dims(op.ck, K, op.C0 + op.C1);
      HIPCHK(h, launch_pack_conv_f32_t(src, h->d_wt + h->wt_off0[op.w], Cout, Cin, w.ks, 0, op.C0 + op.C1, d.rows_pad, d.cols_pad, st));
    } else {
      const WtDims d0 = wt_dims(op.ck, K, op.C0), d1 = wt_dims(op.ck, K, op.C1);
      HIPCHK(h, launch_pack_conv_f32_t(src, h->d_wt + h->wt_off0[op.w], Cout, Cin, w.ks, 0, op.C0, d0.rows_pad, d0.cols_pad, st));
      HIPCHK(h, launch_pack_conv_f32_t(src, h->d_wt + h->wt_off1[op.w], Cout, Cin, w.ks, op.C0, op.C1, d1.rows_pad, d1.cols_pad, st));
    }
  }
  // f16x3 forms (forward + transposed), packed on the device with a per-tensor power-of-two scale
  HIPCHK(h, hipMemsetAsync(h->d_hamax, 0, h->weights.size() * sizeof(unsigned), st));
  for (int i = 0; i < h->n_schema; ++i) {
    const WeightEntry& w = h->weights[i];
    if (w.live && w.sink == WeightEntry::CONV_PACK && w.h_ok)
      HIPCHK(h, launch_hamax(h->d_master + h->master_off[i], numel(w.shape), h->d_hamax + i, st));
  }
  // every slot gets a scale; the slots that are not f16x3 conv weights (amax 0 -> e = 12) are never read
  HIPCHK(h, launch_hscale_all(h->d_hamax, h->d_hscale, (int)h->weights.size(), st));
  for (int i = 0; i < h->n_schema; ++i) {
    WeightEntry& w = h->weights[i];
    if (!w.live || w.sink != WeightEntry::CONV_PACK || !w.h_ok) continue;
    const float* src = h->d_master + h->master_off[i];
    float* sc2 = h->d_hscale + 2 * (size_t)i;
    if (forward_forms) {
      HIPCHK(h, launch_pack_conv_h(src, h->d_wq + w.hq_off[PREC_F16X3], sc2, (int)w.shape[0], (int)w.shape[1], w.ks, w.h_WN,
                                   w.h_cout_pad, w.h_cin_pad, 0, 0, 0, st));
      if (w.ck == CONV3_UP)   // and the sub-pixel form the upsample convs run on
        HIPCHK(h, launch_pack_conv_up2_h(src, h->d_wq + w.up2_off[PREC_F16X3], sc2, h->d_up2_inv + i, (int)w.shape[0], (int)w.shape[1],
                                         w.h_WN, w.h_cout_pad, w.h_cin_pad, st));
    }
  }
  if (forward_forms) {
    h->up2_dev_fresh = true;
    h->h_forms_stale = true;   // the sampling forms (bf16, the host-packed sub-pixel forms) now differ from what this pass wrote: the next
                               // eval-mode sample / forward, or a switch of the precision, re-packs them on the host
  }
  for (const Op& op : h->ops) {
    if (op.kind != Op::CONV || op.src0 == h->t_in || h->wtq_off0[op.w] == SIZE_MAX) continue;
    const WeightEntry& w = h->weights[op.w];
    const float* src = h->d_master + h->master_off[op.w];
    const float* sc2 = h->d_hscale + 2 * (size_t)op.w;
    const int K = conv_K(h, op), Cout = (int)w.shape[0], Cin = (int)w.shape[1];
    auto pack_t = [&](size_t qo, int c_off, int rows) -> hipError_t {
      int TH, WN;
      conv_h_config(op.ck == CONV1 ? CONV1 : CONV3_S1, rows, &TH, &WN);
      return launch_pack_conv_h(src, h->d_wtq + qo, sc2, Cout, Cin, w.ks, WN, round_up(rows, 32 * WN), round_up(K, 16), 1, c_off, rows, st);
    };
    if (h->wtq_off1[op.w] == SIZE_MAX) {
      HIPCHK(h, pack_t(h->wtq_off0[op.w], 0, op.C0 + op.C1));
    } else {
      HIPCHK(h, pack_t(h->wtq_off0[op.w], 0, op.C0));
      HIPCHK(h, pack_t(h->wtq_off1[op.w], op.C0, op.C1));
    }
  }
  h->temb_table_valid = false;
  h->f32_forms_stale = all_f32_forms ? false : (h->f32_forms_stale || skipped);
  return FDSR_OK;
}

// one transposed convolution on the forward fp32 kernel: out (+)= conv(dy; wt)
int launch_dgrad(fdsr_handle h, ConvKind ck, const float* dy, int K, int Hs, int Ws, const float* wt, int Csub, float* out,
                 bool accumulate, int N, hipStream_t st) {
  const ConvKind k = ck == CONV1 ? CONV1 : CONV3_S1;
  const WtDims d = wt_dims(ck, K, Csub);
  ConvParams p{};
  p.x0 = dy; p.w = wt; p.bias = h->d_zero; p.out = out; p.res = accumulate ? out : nullptr;
  p.N = N; p.Hin = Hs; p.Win = Ws; p.Hout = Hs; p.Wout = Ws;
  p.C0 = K; p.C1 = 0; p.Cout = Csub; p.Cin_pad = d.cols_pad; p.Cout_pad = d.rows_pad;
  HIPCHK(h, launch_conv(k, p, st, nullptr));
  return FDSR_OK;
}

// the same on the f16x3 kernels (fp32-grade: three f16 MFMAs per product), fragments from d_wtq
// gb (optional): the GroupNorm-backward fields (ConvParams::gb_*, part_out) of a launch whose output is the gradient w.r.t. a
// GroupNorm'd + activated input; *gb_tiles returns the tiles per image of the per-tile sums, 0 when the kernel this launch lands on
// has no such epilogue (the fields are then left out and the caller runs the reduce pass).
int launch_dgrad_h(fdsr_handle h, const Op& op, const float* dy, int K, int Hs, int Ws, size_t qoff, int Csub, float* out,
                   bool accumulate, int N, hipStream_t st, const ConvParams* gb = nullptr, int* gb_tiles = nullptr) {
  const ConvKind k = op.ck == CONV1 ? CONV1 : CONV3_S1;
  int TH, WN;
  conv_h_config(k, Csub, &TH, &WN);
  ConvParams p{};
  p.x0 = dy; p.bias = h->d_zero; p.out = out; p.res = accumulate ? out : nullptr;
  p.N = N; p.Hin = Hs; p.Win = Ws; p.Hout = Hs; p.Wout = Ws;
  p.C0 = K; p.C1 = 0; p.Cout = Csub; p.Cin_pad = round_up(K, 16); p.Cout_pad = round_up(Csub, 32 * WN);
  p.wq = h->d_wtq + qoff;
  p.w_inv_scale = 1.0f;
  p.w_inv_scale_dev = h->d_hscale + 2 * (size_t)op.w + 1;
  p.ksplit = 1;
  p.out_f32 = 1;
  if (gb_tiles) *gb_tiles = 0;
  // (gn_bwd_finalize_tiles_kernel holds at most 64 channels of a group: wider groups -- norm_groups small against the channel
  // count -- take the separate reduce pass)
  if (gb && gb_tiles && g_tun.gnb_fuse && gb->gb_G > 0 && Csub / gb->gb_G <= 64 && conv_h_gnb_ok(k, PREC_F16X3, p)) {
    p.gb_x0 = gb->gb_x0; p.gb_x1 = gb->gb_x1; p.gb_C0 = gb->gb_C0; p.gb_G = gb->gb_G; p.gb_plain = gb->gb_plain;
    p.gb_scale = gb->gb_scale; p.gb_shift = gb->gb_shift; p.gb_stats = gb->gb_stats; p.gb_mask = gb->gb_mask; p.gb_drop = gb->gb_drop;
    p.part_out = gb->part_out;
    HIPCHK(h, launch_conv_h(k, PREC_F16X3, p, st, gb_tiles));
    return FDSR_OK;
  }
  HIPCHK(h, launch_conv_h(k, PREC_F16X3, p, st, nullptr));
  return FDSR_OK;
}

}  // namespace

namespace fdsr_int {
int train_workspace_extra(fdsr_handle h, int N, int H, int W, size_t* bytes) {
  TrainPlan tp;
  int rc = make_train_plan(h, N, H, W, &tp);
  if (rc) return rc;
  *bytes = tp.bytes;
  return FDSR_OK;
}
// the fp32 conv forms a lazy f16x3 step left behind (see repack_from_master)
int ensure_f32_forms(fdsr_handle h, hipStream_t st) {
  if (!h->f32_forms_stale || !h->train_ready) return FDSR_OK;
  return repack_from_master(h, st, true, true);
}
}  // namespace fdsr_int

extern "C" {

int fdsr_train_workspace_bytes(fdsr_handle h, int batch, int height, int width, size_t* bytes) {
  if (!h || !bytes) return fail(h, FDSR_E_INVALID, "null argument");
  const bool dbg = h->debug;
  h->debug = true;                                        // keep mode: every activation has its own buffer
  int rc = get_plan(h, batch, height, width);
  h->debug = dbg;
  if (rc) return rc;
  return train_workspace_extra(h, batch, height, width, bytes);
}

}  // extern "C"

namespace {
// The step behind both entry points.  x_nchw != null: the packed network input is given ([B,6,H,W], cat([SR, x_noisy])).
// Otherwise (hr, sr): img2res + q_sample + cat (diffusion.py:233-241, :283-289, :257-263) run here, in the kernel that writes
// the packed NHWC input; target_nchw == null then means: the engine draws the noise itself.
int train_grads_impl(fdsr_handle h, const float* x_nchw, const float* hr_nchw, const float* sr_nchw, const float* noise_level,
                     const float* target_nchw, int loss_l2, float loss_scale, float* loss_host, int batch, int height, int width,
                     void* workspace, size_t workspace_bytes, void* hip_stream) {
  if (prec_is16(h->prec))
    return fail(h, FDSR_E_INVALID, "the training step runs the fp32-grade kernels: fdsr_set_precision(FDSR_PREC_F32 or FDSR_PREC_F16X3) first");
  if (h->cfg.in_channel != 6 || h->cfg.out_channel != 3) return fail(h, FDSR_E_INVALID, "training needs in_channel=6, out_channel=3");
  int rc = check_ready(h, false);
  if (rc) return rc;
  if ((rc = train_prepare(h))) return rc;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  const int N = batch, H = height, W = width, G = h->cfg.norm_groups;
  // keep mode for the forward of this call only: the caller's own fdsr_set_debug / statistics settings come back on every exit path
  struct KeepMode {
    fdsr_handle h; bool debug, keep;
    explicit KeepMode(fdsr_handle hh) : h(hh), debug(hh->debug), keep(hh->keep_stats) { h->debug = true; }
    void restore() { h->debug = debug; h->keep_stats = keep; }
    ~KeepMode() { restore(); }
  } keep_mode(h);
  rc = get_plan(h, N, H, W);
  if (rc) return rc;
  TrainPlan tp;
  if ((rc = make_train_plan(h, N, H, W, &tp))) return rc;
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) || workspace_bytes < tp.bytes)
    return fail(h, FDSR_E_WORKSPACE, "training workspace too small or misaligned: %zu < %zu bytes", workspace_bytes, tp.bytes);
  char* ws = reinterpret_cast<char*>(workspace);
  ShapePlan& sp = h->plan;
  // the transposed forms follow the master copy
  if (h->prec == PREC_F32 && (rc = ensure_f32_forms(h, st))) return rc;
  if (!h->wt_valid) {
    // f16x3: the forward forms are re-packed on the device here too, as every optimiser step will: a run resumed from a checkpoint
    // (host-packed at load) then steps on the same bits as the uninterrupted run (the two packers round the sub-pixel forms differently)
    if ((rc = repack_from_master(h, st, h->prec == PREC_F16X3, false))) return rc;
    h->wt_valid = true;
  }

  // ---- forward, keeping every activation and the GroupNorm statistics ----
  float* xin = reinterpret_cast<float*>(ws + sp.tensor_off[h->t_in]);
  if (x_nchw) {
    HIPCHK(h, launch_nchw_to_nhwc(x_nchw, xin, N, h->cfg.in_channel, H, W, h->CP, 0, 1, st));
  } else {
    if (!target_nchw) {   // nobody supplied the noise: draw it (Philox, keyed by fdsr_set_seed and the count of such steps)
      if ((rc = ensure_rng(h))) return rc;
      float* nz = reinterpret_cast<float*>(ws + tp.off_noise);
      HIPCHK(h, launch_rng_advance(h->d_rng, st));
      HIPCHK(h, launch_randn_plane(h->d_rng, nz, N, H * W, 0, st));
      target_nchw = nz;
    }
    HIPCHK(h, launch_qsample_pack(hr_nchw, sr_nchw, noise_level, target_nchw, xin, N, H * W, h->CP, st));
  }
  h->keep_stats = true;
  const bool sat_armed = h->prec == PREC_F16X3 && g_tun.sat_guard && h->d_sat;
  if (sat_armed) HIPCHK(h, hipMemsetAsync(h->d_sat, 0, sizeof(int), st));   // the range flag speaks for this step only
  rc = run_unet(h, N, H, W, ws, noise_level, 0.f, st);
  keep_mode.restore();
  if (rc) return rc;

  auto P = [&](int widx) -> const float* { return widx >= 0 ? h->d_params + h->weights[widx].dev_off : nullptr; };
  auto TP = [&](int t) -> float* { return t >= 0 ? reinterpret_cast<float*>(ws + sp.tensor_off[t]) : nullptr; };
  auto GT = [&](int t) -> float* { return t >= 0 ? reinterpret_cast<float*>(ws + tp.grad_off[t]) : nullptr; };
  auto DG = [&](int widx) -> float* { return h->d_grad + h->master_off[widx]; };
  float* tmpA = reinterpret_cast<float*>(ws + tp.off_tmpA);
  float* tmpZ = reinterpret_cast<float*>(ws + tp.off_tmpZ);
  float* S = reinterpret_cast<float*>(ws + tp.off_S);
  float* dtemb = reinterpret_cast<float*>(ws + tp.off_dtemb);
  double* dbl = reinterpret_cast<double*>(ws + tp.off_dbl);
  float* wg = reinterpret_cast<float*>(ws + tp.off_wg);
  float* loss_dev = reinterpret_cast<float*>(ws + tp.off_loss);

  // ---- loss and its gradient ----
  // No memset of the gradient tensors: the FIRST contribution to each (the host loop below is the order) is a plain store, later
  // ones accumulate.  `first(t)` answers that once per tensor; a gradient that is read before anything wrote it is a plan error.
  std::vector<char> touched(h->tensors.size(), 0);
  auto first = [&](int t) { const bool f = !touched[t]; touched[t] = 1; return f; };
  auto grad_bytes_of = [&](int t) {
    const TensorDesc& td = h->tensors[t];
    return (size_t)N * (size_t)(H >> td.level) * (W >> td.level) * tensor_channels(h, t) * sizeof(float);
  };
  HIPCHK(h, hipMemsetAsync(dtemb, 0, (size_t)N * h->TE * sizeof(float), st));
  const float* eps = TP(h->t_eps);
  // f16x3: the gradients that flow through the split-f16 convolutions are kept near 1 by a power-of-two factor
  // (the loss gradient is +-loss_scale, typically 1e-7; f16's hi/lo split keeps 22 bits only above ~6e-5) and the
  // arena is un-scaled at the end.  Powers of two: exact in fp32, so the fp32 parts of the backward do not see it.
  float gscale = 1.0f;
  if (h->prec == PREC_F16X3 && loss_scale > 0.f) gscale = std::ldexp(1.0f, -(int)std::floor(std::log2((double)loss_scale)));
  HIPCHK(h, launch_loss_grad(eps, target_nchw, GT(h->t_eps), dbl, loss_dev, N, H * W, loss_l2, loss_scale * gscale, st));
  touched[h->t_eps] = 1;

  // ---- backward over the plan ----
  std::vector<const Op*> gn_of_slot((size_t)std::max(h->n_gn_slots, 1), nullptr);   // the finalisation op of a slot: its gamma / beta, its scale-shift columns
  for (const Op& o : h->ops)
    if (o.kind == Op::GN_FINALIZE && o.gn_slot >= 0) gn_of_slot[o.gn_slot] = &o;
  float* temb_tab = reinterpret_cast<float*>(ws + sp.off_temb);   // [N][TE]: the step's per-block embedding rows (run_unet filled it)
  for (int oi = (int)h->ops.size() - 1; oi >= 0; --oi) {
    const Op& op = h->ops[oi];
    const int Hi = H >> op.lvl_in, Wi = W >> op.lvl_in;
    if (op.kind == Op::ATTN) {   // attention core (SR3 / TESR: one head; GDP: heads of 64 channels): d qkv from d O; the qkv tensor has no other reader
      if (!touched[op.dst]) return fail(h, FDSR_E_STATE, "internal: gradient read before it was written");
      AttnBwdParams a{};
      a.qkv = TP(op.src0); a.dO = GT(op.dst); a.dqkv = GT(op.src0);
      a.scratch = reinterpret_cast<float*>(ws + tp.off_attn);
      a.N = N; a.HW = Hi * Wi; a.C = op.C0; a.heads = op.heads;
      if (!first(op.src0)) return fail(h, FDSR_E_STATE, "internal: the qkv tensor has a second reader");
      HIPCHK(h, launch_attn_bwd(a, st));
      continue;
    }
    if (op.kind == Op::UP2X) {   // GDP up ResBlock, x' = nearest-x2(x): the 2x2 sums of d x'
      if (!touched[op.dst]) return fail(h, FDSR_E_STATE, "internal: gradient read before it was written");
      HIPCHK(h, launch_pool2_add(GT(op.dst), GT(op.src0), N, Hi, Wi, op.C0, first(op.src0), st));
      continue;
    }
    if (op.kind == Op::POOL2) {  // GDP down ResBlock: avg_pool2(x) (the skip path) or avg_pool2(silu(norm(x))) (the main path)
      if (!touched[op.dst]) return fail(h, FDSR_E_STATE, "internal: gradient read before it was written");
      if (op.gn_slot < 0) {
        HIPCHK(h, launch_unpool2_add(GT(op.dst), GT(op.src0), N, Hi >> 1, Wi >> 1, op.C0, 0.25f, first(op.src0), st));
        continue;
      }
      const Op* gn = gn_of_slot[op.gn_slot];
      if (!gn) return fail(h, FDSR_E_STATE, "internal: pooled GroupNorm without its finalisation op");
      HIPCHK(h, launch_unpool2_add(GT(op.dst), tmpA, N, Hi >> 1, Wi >> 1, op.C0, 0.25f, true, st));
      GnBwdParams g{};
      g.dA = tmpA; g.x0 = TP(op.src0); g.C0 = op.C0;
      g.scale = reinterpret_cast<const float*>(ws + sp.gn_off[op.gn_slot]);
      g.shift = g.scale + (size_t)N * op.C0;
      g.stats = reinterpret_cast<const float*>(ws + sp.gn_stats_off[op.gn_slot]);
      g.gamma = P(gn->gamma);
      g.dx0 = GT(op.src0);
      g.assign0 = first(op.src0) ? 1 : 0;
      g.dgamma = DG(gn->gamma); g.dbeta = DG(gn->beta);
      g.scratch = dbl;
      g.N = N; g.HW = Hi * Wi; g.H = Hi; g.G = G;
      HIPCHK(h, launch_gn_bwd(g, st));
      continue;
    }
    if (op.kind == Op::SLAM) {
      // the CLAM op right before shares src0: gate and map are recomputed from x
      const Op* ca = nullptr;
      for (int k = oi - 1; k >= 0; --k)
        if (h->ops[k].kind == Op::CLAM) { ca = &h->ops[k]; break; }
      if (!ca) return fail(h, FDSR_E_STATE, "internal: SLAM without CLAM");
      ClamSlamBwdParams c{};
      if (!touched[op.dst]) return fail(h, FDSR_E_STATE, "internal: gradient read before it was written");
      if (first(op.src0)) HIPCHK(h, hipMemsetAsync(GT(op.src0), 0, grad_bytes_of(op.src0), st));   // these kernels accumulate
      c.x = TP(op.src0); c.dout = GT(op.dst); c.dx = GT(op.src0);
      c.fc1 = P(ca->fc1); c.fc2 = P(ca->fc2); c.w7 = P(op.w);
      c.dfc1 = DG(ca->fc1); c.dfc2 = DG(ca->fc2); c.dw7 = DG(op.w);
      c.scratch = reinterpret_cast<float*>(ws + tp.off_csb);
      c.N = N; c.H = Hi; c.W = Wi; c.C = op.C0; c.Cr = op.C0 / 16;
      HIPCHK(h, launch_clam_slam_bwd(c, st));
      continue;
    }
    if (op.kind != Op::CONV) continue;
    const WeightEntry& w = h->weights[op.w];
    const int Ho = H >> op.lvl_out, Wo = W >> op.lvl_out, Cin = op.C0 + op.C1, K = conv_K(h, op);
    const float* dy = GT(op.dst);
    if (!touched[op.dst]) return fail(h, FDSR_E_STATE, "internal: gradient read before it was written");
    // residual (identity): the same gradient flows to the block input
    if (op.res >= 0 && op.res != op.dst)
      HIPCHK(h, launch_add_slice(dy, GT(op.res), (size_t)N * Ho * Wo, op.Cout, 0, op.Cout, first(op.res), st));
    // weight gradient; the per-image column sums of dy (bias and noise-embedding gradients) come out of the f16x3 weight-gradient
    // kernel's own staging of dy where it can produce them, else from a pass of their own
    {
      WgradParams q{};
      q.dy = dy; q.x0 = TP(op.src0); q.x1 = TP(op.src1);
      if (op.gn_slot >= 0) {
        q.gn_scale = reinterpret_cast<const float*>(ws + sp.gn_off[op.gn_slot]);
        q.gn_shift = q.gn_scale + (size_t)N * Cin;
        q.gn_plain = op.gn_plain ? 1 : 0;
      }
      q.dw = DG(op.w); q.scratch = wg;
      q.N = N; q.Hin = Hi; q.Win = Wi; q.Hout = Ho; q.Wout = Wo;
      q.C0 = op.C0; q.C1 = op.C1; q.Cin_real = (int)w.shape[1]; q.Cout = op.Cout; q.Cout_s = K;
      if (sp.training && op.drop_slot >= 0) {
        q.drop_mask = reinterpret_cast<const unsigned char*>(ws + sp.drop_off[op.drop_slot]);
        q.drop_scale = 1.0f / (1.0f - h->cfg.dropout);
      }
      const bool f32_wgrad = g_tun.wgrad_f32 != 0;   // A/B option: keep the weight gradients exact fp32
      const bool hw = h->prec == PREC_F16X3 && !f32_wgrad;
      q.colsum = S;
      if (!(hw && wgrad_h_fuses_colsum(op.ck, q))) {
        q.colsum = nullptr;
        HIPCHK(h, launch_colsum(dy, S, dbl, N, Ho * Wo, K, st));
      }
      if (hw) HIPCHK(h, launch_wgrad_h(op.ck, q, st));
      else HIPCHK(h, launch_wgrad(op.ck, q, st));
    }
    if (op.b >= 0) HIPCHK(h, launch_sum_rows(S, N, K, op.Cout, DG(op.b), st));   // db[c] = sum_n S[n][c]
    if (op.temb_off >= 0)   // rows of S (stride K) -> this block's columns of the [N][TE] table
      HIPCHK(h, hipMemcpy2DAsync(dtemb + op.temb_off, (size_t)h->TE * sizeof(float), S, (size_t)K * sizeof(float),
                                 (size_t)op.Cout * sizeof(float), N, hipMemcpyDeviceToDevice, st));
    if (op.src0 == h->t_in) continue;                     // no gradient w.r.t. the network input
    // input gradient: on the f16x3 kernels in that mode (where the transposed shape fits them), else exact fp32
    auto dgrad = [&](const float* dyp, int Hs, int Ws, int src, int Csub, float* outp, bool acc) -> int {
      const size_t qo = src == 0 ? h->wtq_off0[op.w] : h->wtq_off1[op.w];
      if (h->prec == PREC_F16X3 && qo != SIZE_MAX) return launch_dgrad_h(h, op, dyp, K, Hs, Ws, qo, Csub, outp, acc, N, st);
      return launch_dgrad(h, op.ck, dyp, K, Hs, Ws, h->d_wt + (src == 0 ? h->wt_off0[op.w] : h->wt_off1[op.w]), Csub, outp, acc, N, st);
    };
    if (op.gn_slot >= 0) {
      GnBwdParams g{};
      const Op* gn = gn_of_slot[op.gn_slot];
      const bool up = op.ck == CONV3_UP;     // GDP up ResBlock: conv over nearest-x2(silu(norm(x))): dA is the 2x2 sum of the fine-grid gradient
      // f16x3: the first half of the GroupNorm backward (g = dA * dropout * swish'(u) and its per-tile channel sums) runs in the epilogue
      // of the launch that produces dA, where that launch is a 16x16x32 kernel
      int gb_tiles = 0;
      {
        const size_t qo = h->wtq_off0[op.w];
        if (h->prec == PREC_F16X3 && qo != SIZE_MAX && !up) {
          ConvParams gb{};
          gb.gb_x0 = TP(op.src0); gb.gb_x1 = TP(op.src1); gb.gb_C0 = op.C0; gb.gb_G = G; gb.gb_plain = op.gn_plain ? 1 : 0;
          gb.gb_scale = reinterpret_cast<const float*>(ws + sp.gn_off[op.gn_slot]);
          gb.gb_shift = gb.gb_scale + (size_t)N * Cin;
          gb.gb_stats = reinterpret_cast<const float*>(ws + sp.gn_stats_off[op.gn_slot]);
          if (sp.training && op.drop_slot >= 0) {
            gb.gb_mask = reinterpret_cast<const unsigned char*>(ws + sp.drop_off[op.drop_slot]);
            gb.gb_drop = 1.0f / (1.0f - h->cfg.dropout);
          }
          gb.part_out = gn_bwd_tile_part(dbl);
          if ((rc = launch_dgrad_h(h, op, dy, K, Ho, Wo, qo, Cin, tmpA, false, N, st, &gb, &gb_tiles))) return rc;
        } else if ((rc = dgrad(dy, Ho, Wo, 0, Cin, tmpA, false))) return rc;
      }
      if (up) HIPCHK(h, launch_pool2_add(tmpA, tmpZ, N, Hi, Wi, Cin, true, st));
      if (gb_tiles > 0) { g.g_part = gn_bwd_tile_part(dbl); g.g_nt = gb_tiles; }
      g.dA = up ? tmpZ : tmpA; g.x0 = TP(op.src0); g.x1 = TP(op.src1); g.C0 = op.C0; g.C1 = op.C1;
      g.scale = reinterpret_cast<const float*>(ws + sp.gn_off[op.gn_slot]);
      g.shift = g.scale + (size_t)N * Cin;
      g.stats = reinterpret_cast<const float*>(ws + sp.gn_stats_off[op.gn_slot]);
      g.gamma = P(op.gamma);
      g.dx0 = GT(op.src0); g.dx1 = GT(op.src1);
      g.assign0 = first(op.src0) ? 1 : 0;
      g.assign1 = op.src1 >= 0 && first(op.src1) ? 1 : 0;
      g.dgamma = DG(op.gamma); g.dbeta = DG(op.beta);
      g.scratch = dbl;
      g.N = N; g.HW = Hi * Wi; g.H = Hi; g.G = G; g.plain = op.gn_plain ? 1 : 0;
      if (sp.training && op.drop_slot >= 0) {
        g.drop_mask = reinterpret_cast<const unsigned char*>(ws + sp.drop_off[op.drop_slot]);
        g.drop_scale = 1.0f / (1.0f - h->cfg.dropout);
      }
      if (gn && gn->film_off >= 0) {          // scale-shift norm: (s, t) are columns of the embedding table, their gradient goes to dtemb
        g.film = temb_tab + gn->film_off; g.film_stride = h->TE;
        g.beta = P(op.beta);
        g.dfilm = dtemb + gn->film_off; g.dfilm_stride = h->TE;
      }
      HIPCHK(h, launch_gn_bwd(g, st));
    } else if (op.ck == CONV3_S2) {
      HIPCHK(h, launch_zero_insert(dy, tmpZ, N, Ho, Wo, K, st));
      if ((rc = dgrad(tmpZ, Hi, Wi, 0, op.C0, GT(op.src0), !first(op.src0)))) return rc;
    } else if (op.ck == CONV3_UP) {
      if ((rc = dgrad(dy, Ho, Wo, 0, op.C0, tmpA, false))) return rc;
      HIPCHK(h, launch_pool2_add(tmpA, GT(op.src0), N, Hi, Wi, op.C0, first(op.src0), st));
    } else {
      if ((rc = dgrad(dy, Ho, Wo, 0, op.C0, GT(op.src0), !first(op.src0)))) return rc;
      if (op.C1 > 0 && (rc = dgrad(dy, Ho, Wo, 1, op.C1, GT(op.src1), !first(op.src1)))) return rc;
    }
  }

  // ---- noise-level embedding ----
  {
    TembBwdParams t{};
    t.freq = P(h->w_freq); t.w1 = P(h->w_mlp[0]); t.b1 = P(h->w_mlp[1]); t.w2 = P(h->w_mlp[2]); t.b2 = P(h->w_mlp[3]);
    t.wn = h->d_params + h->noise_w_off;
    t.nl = noise_level; t.dtemb = dtemb;
    t.dw1 = DG(h->w_mlp[0]); t.db1 = DG(h->w_mlp[1]); t.dw2 = DG(h->w_mlp[2]); t.db2 = DG(h->w_mlp[3]);
    t.dwn = reinterpret_cast<float*>(ws + tp.off_dwn); t.dbn = reinterpret_cast<float*>(ws + tp.off_dbn);
    t.scratch = reinterpret_cast<float*>(ws + tp.off_tb);
    t.inner = h->cfg.inner_channel; t.TE = h->TE; t.N = N;
    t.swish_block = (h->sr3 || h->gdp) ? 1 : 0;
    if (h->gdp) { t.enc_dim = t.inner; t.hid_dim = t.t_dim = 4 * t.inner; t.cos_first = 1; }
    const int t_dim = t.t_dim ? t.t_dim : t.inner;
    HIPCHK(h, launch_temb_bwd(t, st));
    for (int i = 0; i < h->n_schema; ++i) {               // scatter the concatenated tables back to the per-block tensors
      const WeightEntry& w = h->weights[i];
      if (!w.live) continue;
      if (w.sink == WeightEntry::NOISE_W)
        HIPCHK(h, hipMemcpyAsync(DG(i), t.dwn + (size_t)w.row_off * t_dim, numel(w.shape) * sizeof(float), hipMemcpyDeviceToDevice, st));
      else if (w.sink == WeightEntry::NOISE_B)
        HIPCHK(h, hipMemcpyAsync(DG(i), t.dbn + w.row_off, numel(w.shape) * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  }
  if (gscale != 1.0f) HIPCHK(h, launch_scale_inplace(h->d_grad, h->master_floats, 1.0f / gscale, st));
  if (loss_host) {
    // the step's forward armed the f16x3 range flag (run_unet): it is read in the same synchronisation as the loss, so a
    // clamped raw input fails THIS call instead of surfacing in an unrelated later fdsr_check_saturation
    if (sat_armed && !h->h_sat) HIPCHK(h, hipHostMalloc((void**)&h->h_sat, 64, hipHostMallocDefault));
    if (sat_armed) { *h->h_sat = 0; HIPCHK(h, hipMemcpyAsync(h->h_sat, h->d_sat, sizeof(int), hipMemcpyDeviceToHost, st)); }
    HIPCHK(h, hipMemcpyAsync(loss_host, loss_dev, sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (sat_armed && *h->h_sat) {
      HIPCHK(h, hipMemsetAsync(h->d_sat, 0, sizeof(int), st));
      return fail(h, FDSR_E_SATURATED, "f16x3 training step: a raw convolution input exceeded the f16 range (+-65504) and was clamped in "
                                       "the forward pass; the gradients of this call are not fp32-grade: re-run it with "
                                       "fdsr_set_precision(FDSR_PREC_F32)");
    }
  }
  return FDSR_OK;
}

}  // namespace

extern "C" {

int fdsr_train_grads(fdsr_handle h, const float* x_nchw, const float* noise_level, const float* target_nchw, int loss_l2,
                     float loss_scale, float* loss_host, int batch, int height, int width, void* workspace, size_t workspace_bytes,
                     void* hip_stream) {
  if (!h || !x_nchw || !noise_level || !target_nchw) return fail(h, FDSR_E_INVALID, "null argument");
  return train_grads_impl(h, x_nchw, nullptr, nullptr, noise_level, target_nchw, loss_l2, loss_scale, loss_host, batch, height, width,
                          workspace, workspace_bytes, hip_stream);
}

int fdsr_train_grads_pairs(fdsr_handle h, const float* hr_nchw, const float* sr_nchw, const float* gamma, const float* noise_nchw,
                           int loss_l2, float loss_scale, float* loss_host, int batch, int height, int width, void* workspace,
                           size_t workspace_bytes, void* hip_stream) {
  if (!h || !hr_nchw || !sr_nchw || !gamma) return fail(h, FDSR_E_INVALID, "null argument");
  return train_grads_impl(h, nullptr, hr_nchw, sr_nchw, gamma, noise_nchw, loss_l2, loss_scale, loss_host, batch, height, width,
                          workspace, workspace_bytes, hip_stream);
}

int fdsr_adam_step(fdsr_handle h, float lr, float beta1, float beta2, float eps, void* hip_stream) {
  if (!h) return FDSR_E_INVALID;
  if (!h->train_ready) return fail(h, FDSR_E_STATE, "fdsr_train_grads has not run yet");
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  h->adam_t += 1;
  HIPCHK(h, launch_adam(h->d_master, h->d_grad, h->d_adam_m, h->d_adam_v, h->master_floats, lr, beta1, beta2, eps, h->adam_t, st));
  int rc = repack_from_master(h, st, true, false);
  if (rc) return rc;
  h->wt_valid = true;
  h->h_forms_stale = true;
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
  h->graphs.clear();
  return FDSR_OK;
}

static int find_weight(fdsr_handle h, const char* key) {
  auto it = h->key2w.find(key);
  if (it == h->key2w.end() || it->second >= h->n_schema) return -1;
  return it->second;
}

int fdsr_get_weight(fdsr_handle h, const char* key, float* host) {
  if (!h || !key || !host) return fail(h, FDSR_E_INVALID, "null argument");
  const int i = find_weight(h, key);
  if (i < 0) return fail(h, FDSR_E_KEY, "unexpected key '%s'", key);
  if (!h->weights[i].live) return fail(h, FDSR_E_KEY, "'%s' is never executed (unet.py:212): the engine keeps no copy", key);
  if (!h->d_master) return fail(h, FDSR_E_STATE, "no weights loaded");
  HIPCHK(h, hipMemcpy(host, h->d_master + h->master_off[i], numel(h->weights[i].shape) * sizeof(float), hipMemcpyDeviceToHost));
  return FDSR_OK;
}

int fdsr_get_grad(fdsr_handle h, const char* key, float* host) {
  if (!h || !key || !host) return fail(h, FDSR_E_INVALID, "null argument");
  const int i = find_weight(h, key);
  if (i < 0) return fail(h, FDSR_E_KEY, "unexpected key '%s'", key);
  if (!h->weights[i].live) return fail(h, FDSR_E_KEY, "'%s' is never executed (unet.py:212): it has no gradient", key);
  if (!h->d_grad) return fail(h, FDSR_E_STATE, "fdsr_train_grads has not run yet");
  HIPCHK(h, hipMemcpy(host, h->d_grad + h->master_off[i], numel(h->weights[i].shape) * sizeof(float), hipMemcpyDeviceToHost));
  return FDSR_OK;
}

// torch.optim.Adam's per-parameter state (exp_avg, exp_avg_sq) and its step count, for `_opt.pth` (model.py:126-166)
int fdsr_get_optimizer_state(fdsr_handle h, const char* key, float* exp_avg, float* exp_avg_sq, int* step) {
  if (!h || !key) return fail(h, FDSR_E_INVALID, "null argument");
  const int i = find_weight(h, key);
  if (i < 0 || !h->weights[i].live) return fail(h, FDSR_E_KEY, "no optimiser state for '%s'", key);
  if (!h->train_ready) return fail(h, FDSR_E_STATE, "no optimiser step has run yet");
  const size_t n = numel(h->weights[i].shape) * sizeof(float);
  if (exp_avg) HIPCHK(h, hipMemcpy(exp_avg, h->d_adam_m + h->master_off[i], n, hipMemcpyDeviceToHost));
  if (exp_avg_sq) HIPCHK(h, hipMemcpy(exp_avg_sq, h->d_adam_v + h->master_off[i], n, hipMemcpyDeviceToHost));
  if (step) *step = h->adam_t;
  return FDSR_OK;
}

int fdsr_set_optimizer_state(fdsr_handle h, const char* key, const float* exp_avg, const float* exp_avg_sq, int step) {
  if (!h || !key || !exp_avg || !exp_avg_sq || step < 0) return fail(h, FDSR_E_INVALID, "bad optimiser state");
  const int i = find_weight(h, key);
  if (i < 0 || !h->weights[i].live) return fail(h, FDSR_E_KEY, "no optimiser state for '%s'", key);
  int rc = check_ready(h, false);
  if (rc) return rc;
  if ((rc = train_prepare(h))) return rc;
  const size_t n = numel(h->weights[i].shape) * sizeof(float);
  HIPCHK(h, hipMemcpy(h->d_adam_m + h->master_off[i], exp_avg, n, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_adam_v + h->master_off[i], exp_avg_sq, n, hipMemcpyHostToDevice));
  h->adam_t = step;
  return FDSR_OK;
}

// The gradient arena (every executed tensor, checkpoint layout, schema order): data-parallel training sums it
// across ranks in place (RCCL all-reduce over xGMI) between fdsr_train_grads and fdsr_adam_step.
int fdsr_grad_arena(fdsr_handle h, float** dev_ptr, size_t* count) {
  if (!h || !dev_ptr || !count) return fail(h, FDSR_E_INVALID, "null argument");
  if (!h->d_grad) {   // a data-parallel rank with an empty shard asks for the arena before it ever ran a step: allocate (zeroed)
    int rc = check_ready(h, false);
    if (rc) return rc;
    if ((rc = train_prepare(h))) return rc;
  }
  *dev_ptr = h->d_grad;
  *count = h->master_floats;
  return FDSR_OK;
}

// Bring the 16-bit weight forms (f16x3 / bf16 sampling) in line with the master copy after optimiser steps.
int fdsr_sync_weight_forms(fdsr_handle h) {
  if (!h) return FDSR_E_INVALID;
  if (!h->h_forms_stale) return FDSR_OK;
  HIPCHK(h, hipDeviceSynchronize());
  std::vector<float> host;
  for (int i = 0; i < h->n_schema; ++i) {
    WeightEntry& w = h->weights[i];
    if (!w.live || w.sink != WeightEntry::CONV_PACK || !w.h_ok) continue;
    host.resize(numel(w.shape));
    HIPCHK(h, hipMemcpy(host.data(), h->d_master + h->master_off[i], host.size() * sizeof(float), hipMemcpyDeviceToHost));
    int rc = pack_weights_h(h, w, host.data());
    if (rc) return rc;
  }
  h->h_forms_stale = false;
  h->up2_dev_fresh = false;   // the host-packed sub-pixel forms (own scale) are current again
  return FDSR_OK;
}

}  // extern "C"
