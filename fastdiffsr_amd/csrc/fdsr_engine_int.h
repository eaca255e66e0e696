// Internal types of the engine shared by fdsr_engine.cpp (plan, forward, sampling, C ABI) and
// fdsr_train.cpp (backward pass, Adam).  Not part of the C ABI (include/fdsr.h).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/fdsr.h"
#include "fdsr_kernels.h"

namespace fdsr_int {
using namespace fdsr;   // kernel-launch types (ConvKind, ConvParams, ...)

extern thread_local std::string g_global_error;

struct WeightEntry {
  std::string key;
  std::vector<int64_t> shape;
  bool live = true;
  bool loaded = false;
  // sink: how the tensor is stored on the device
  enum Sink { RAW, CONV_PACK, NOISE_W, NOISE_B } sink = RAW;
  size_t dev_off = 0;      // float offset into the parameter arena
  int ks = 1, cin_pad = 0, cout_pad = 0;   // CONV_PACK
  int row_off = 0;                          // NOISE_W / NOISE_B: first row inside the concatenated table
  // 16-bit MFMA forms (CONV_PACK entries that the h-kernels can run)
  bool h_ok = false;
  ConvKind ck = CONV3_S1;
  int h_WN = 0, h_cin_pad = 0, h_cout_pad = 0;
  size_t hq_off[3] = {0, 0, 0};             // byte offsets into the 16-bit weight arena, per Precision
  float h_inv_scale[3] = {1.f, 1.f, 1.f};
  size_t up2_off[3] = {0, 0, 0};            // CONV3_UP only: sub-pixel (4 x 2x2) form
  float up2_inv_scale[3] = {1.f, 1.f, 1.f};
};

struct TensorDesc {
  int C = 0;
  int level = 0;        // spatial = (H >> level, W >> level)
  bool persistent = false;
  bool need_part = false;  // feeds a GroupNorm: its producer also writes per-tile channel sums
  int first_def = -1, last_use = -1;
  size_t off = 0;       // byte offset inside the workspace (per plan)
  std::string name;     // reference module whose output this is ("" for temporaries)
};

struct Op {
  enum Kind { GN_FINALIZE, CONV, CLAM, SLAM, ATTN, POOL2, UP2X } kind;
  std::string name;
  int src0 = -1, src1 = -1, dst = -1, res = -1;
  ConvKind ck = CONV3_S1;
  int C0 = 0, C1 = 0, Cout = 0;
  int lvl_in = 0, lvl_out = 0;
  int gn_slot = -1;
  int w = -1, b = -1, gamma = -1, beta = -1;   // weight-entry indices
  int temb_off = -1;
  int fc1 = -1, fc2 = -1;
  bool no_part = false;   // dst is overwritten later by another producer (res_conv pre-fill)
  int aux = -1;           // ATTN: scratch tensor for the scores
  bool gn_plain = false;  // the GroupNorm in front of this conv has no Swish (SelfAttention.norm / AttentionBlock.norm)
  int film_off = -1;      // GN_FINALIZE: FiLM (1 + scale, shift) of the GDP ResBlock from the embedding table at this column
  int heads = 1;          // ATTN
  bool force_generic = false;   // CONV3_UP with a GroupNorm prologue: not the sub-pixel kernel
  int drop_slot = -1;     // block2 conv with Dropout(p > 0) in front of it (unet.py:89-101): index of its keep-mask
  int rider = -1;         // block2 conv: index of the res_conv op it can absorb as a 1x1 rider (ConvParams::xr0); that op: rider_of
  int rider_of = -1;
};

struct ShapePlan {
  int N = 0, H = 0, W = 0;
  bool debug = false;
  unsigned tun_epoch = 0;          // fdsr::g_tun.epoch the plan was made under
  size_t bytes = 0;
  size_t off_temb = 0, off_gate = 0, off_splitk = 0;
  std::vector<int> op_ksplit;      // per op: K-loop split factor of a 16-bit conv at this shape (1 = none)
  std::vector<size_t> tensor_off;
  std::vector<size_t> part_off;    // per tensor: per-tile channel sums [N][max_tiles][C][2] (0 = none)
  std::vector<size_t> gn_off;      // per GroupNorm slot: scale [N][C] then shift [N][C]
  std::vector<size_t> drop_off;    // per dropout slot: keep bytes [N][HW][C] (plans made in training mode only)
  size_t off_dropA = 0;            // f16x3 training forward: the dropped activation of the current block2, materialised
  bool training = false;
  std::vector<size_t> gn_stats_off;   // per GroupNorm slot: (mean, rstd) [N][G][2], written when the engine keeps statistics
  std::vector<int> tensor_nt;      // tiles per image its producer actually used (set at launch)
  // consumer-side GroupNorm (ConvParams::gsum_out / gs0): per tensor that feeds a GroupNorm, its table of fixed-point channel-pair sums
  // [N][C/2][GSUM_SHARDS][2] int64 in one arena that a single memset clears at the start of a forward
  std::vector<size_t> gsum_off;    // (0 = none)
  size_t off_gsum = 0, gsum_bytes = 0;
  std::vector<char> tensor_gsum;   // this forward's producer of the tensor filled its table (set at launch)
  std::vector<char> gsum_wanted[4];   // per precision, per tensor: some GroupNorm'd conv that reads it would form its statistics itself at this shape
};

struct GraphEntry {
  const void *cond, *noise, *out, *traj, *ws;
  int N, H, W;
  hipGraphExec_t exec;
};

}  // namespace fdsr_int

using namespace fdsr_int;

struct fdsr_engine {

  fdsr_config cfg{};
  std::string err;
  std::vector<WeightEntry> weights;
  std::map<std::string, int> key2w;
  std::vector<TensorDesc> tensors;
  std::vector<Op> ops;
  int t_in = -1, t_eps = -1, CP = 8;
  int n_gn_slots = 0, TE = 0, Cmid = 0, max_qkv = 0;
  int n_schema = 0;   // checkpoint tensors (the entries after them are synthetic)
  std::vector<int> gn_channels;   // per GroupNorm slot
  int w_freq = -1;   // synthetic entry: positional-encoding frequencies (SR3: the checkpoint's inv_freq buffer)
  int w_zero_bias = -1;   // synthetic zeros for bias-free 1x1 convs (attn.qkv)
  bool sr3 = false;          // SR3 sibling (ddpm_modules): integer-time embedding, noise [T+1]
  bool gdp = false;          // GDP sibling (gdp_modules): guided-diffusion UNet, predicts x_0, input cat[x, cond], integer time
  int temb_in = 0;           // row length of the per-block embedding table (inner_channel; GDP: 4 * model_channels)
  int freq_count = 0;        // sinusoid frequencies of the time / noise-level encoding
  bool attn_blocks = false;  // SR3 and TESR siblings: SelfAttention per attn_res + mid[0], no dead .conv, no CLAM/SLAM
  bool plain_out = false;    // SR3 and TESR: the sampler returns x_0 itself (no res2img)
  size_t param_floats = 0, noise_w_off = 0, noise_b_off = 0;
  int w_mlp[4] = {-1, -1, -1, -1};
  float* d_params = nullptr;
  unsigned char* d_wq = nullptr;
  // Sampling feeds every image of a batch the same noise level, and only T distinct ones ever
  // occur (diffusion.py:169-170), so the whole embedding table [T][TE] is evaluated once per
  // (weights, schedule) and the conv epilogues index it with batch stride 0.
  unsigned long long* d_rng = nullptr;   // {seed, call counter}: noise drawn by the engine (fdsr_sample, noise == NULL)
  unsigned long long rng_seed = 0;
  int* h_sat = nullptr;            // pinned landing word of fdsr_check_saturation / the training step's own check
  int* d_sat = nullptr;            // f16x3 range guard: sticky flag raised by the raw-input staging paths
  float* d_temb_table = nullptr;
  float* d_nl = nullptr;
  bool temb_table_valid = false;
  size_t wq_bytes = 0;
  int prec = PREC_F32;
  bool kernels_ready = false;
  // schedule
  int T = 0;
  std::vector<float> s_nl, s_recip, s_recipm1, s_c1, s_c2, s_sigma;
  // plan cache
  ShapePlan plan;
  bool debug = false;
  // profiling
  bool profiling = false;
  bool prof_step = true;     // in sampling only every 4th step is bracketed by events: <1 % overhead in the timed region
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  double prof_flops = 0, prof_bytes = 0;
  std::vector<GraphEntry> graphs;
  unsigned graphs_epoch = 0;
  // ---- training state (fdsr_train.cpp) ----
  float* d_master = nullptr;          // every live checkpoint tensor in checkpoint layout, concatenated in schema order
  std::vector<size_t> master_off;     // per weight entry: float offset into d_master (SIZE_MAX: dead / synthetic)
  size_t master_floats = 0;
  float* d_grad = nullptr;            // gradients, same layout as d_master
  float* d_adam_m = nullptr;
  float* d_adam_v = nullptr;
  int adam_t = 0;
  float* d_wt = nullptr;              // transposed, tap-flipped fp32 conv weights (input-gradient convolutions)
  size_t wt_floats = 0;
  std::vector<size_t> wt_off0, wt_off1;   // per weight entry: offsets into d_wt for concat source 0 / 1 (SIZE_MAX: none)
  float* d_hscale = nullptr;          // per weight entry: {2^e, 2^-e} of its f16x3 forms (read by the kernels after device re-packs)
  unsigned* d_hamax = nullptr;        // scratch of the scale computation: max|w| per weight entry (float bits)
  unsigned char* d_wtq = nullptr;     // transposed, tap-flipped f16x3 fragment forms (input-gradient convolutions in f16x3)
  size_t wtq_bytes = 0;
  std::vector<size_t> wtq_off0, wtq_off1;   // per weight entry (SIZE_MAX: this conv's input gradient stays on the fp32 kernel)
  float* d_zero = nullptr;            // zeros (bias of the input-gradient convolutions)
  bool train_ready = false;
  bool wt_valid = false;              // d_wt matches d_master
  bool training = false;              // .train(): Dropout(p) of block2 is live (unet.py:89-101); fp32 kernels only
  int n_drop_slots = 0;
  unsigned long long drop_seed = 0;   // key of the dropout masks (fdsr_set_seed / fdsr_set_dropout_seed)
  unsigned drop_step = 0;             // forward passes made in training mode: part of the mask's Philox counter
  bool keep_stats = false;            // forward also stores per-(image, group) mean / rstd of every GroupNorm
  bool h_forms_stale = false;         // 16-bit weight forms lag behind the master copy (after an optimiser step)
  float* d_up2_inv = nullptr;         // per weight entry: un-scaling of its device-packed sub-pixel (upsample) form
  bool up2_dev_fresh = false;         // the f16x3 sub-pixel forms were re-packed on the device after the last optimiser step
  bool f32_forms_stale = false;       // fp32 conv forms (d_params packs, d_wt) lag: f16x3 training steps refresh only what they read
  unsigned long long* d_copy_tab = nullptr;   // {src offset, dst offset, count} triples: master -> d_params for the non-conv tensors
  int n_copy_tab = 0;
  size_t copy_tab_max = 0;                // elements of the largest entry of that table
};

namespace fdsr_int {

int fail(fdsr_handle h, int code, const char* fmt, ...);

#define HIPCHK(h, expr)                                                                     \
  do {                                                                                      \
    hipError_t e__ = (expr);                                                                \
    if (e__ != hipSuccess)                                                                  \
      return fail(h, FDSR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

// fdsr_engine.cpp
int get_plan(fdsr_handle h, int N, int H, int W);
int check_ready(fdsr_handle h, bool need_schedule);
int check_ws(fdsr_handle h, void* ws, size_t bytes);
int ensure_rng(fdsr_handle h);
int run_unet(fdsr_handle h, int N, int H, int W, char* ws, const float* nl_dev, float nl_scalar, hipStream_t st,
             const float* temb_row = nullptr);
int pack_weights_h(fdsr_handle h, WeightEntry& w, const float* host);
int ensure_f32_forms(fdsr_handle h, hipStream_t st);   // fdsr_train.cpp: fp32 conv forms left behind by lazy f16x3 training steps
// fdsr_train.cpp
int train_workspace_extra(fdsr_handle h, int N, int H, int W, size_t* bytes);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int round_up(int v, int a) { return (v + a - 1) / a * a; }
inline size_t numel(const std::vector<int64_t>& s) {
  size_t n = 1;
  for (auto d : s) n *= (size_t)d;
  return n;
}

}  // namespace fdsr_int
