// Column-strip form of the 64-cout stride-1 3x3 convolutions (the 256 x 256 level of the UNet: reference unet.py:89-120, Block /
// ResnetBlock at inner_channel = 64) for gfx950: v_mfma_f32_16x16x32_{bf16,f16}, fp32 accumulate.
//
// The tile-per-workgroup kernels (fdsr_conv_k32.hip) run these launches as 2 - 6 K chunks per tile: prologue (a cold fetch of the
// first halo image), staging and epilogue never leave the critical path, and every input pixel is activated 1.33 x (halo).  Here a
// 256-thread workgroup walks DOWN a strip of SW = 16 NPH pixel columns:
//
//   * WEIGHTS LIVE IN REGISTERS for the whole strip: wave w owns output channels [16 w, 16 w + 16) and holds the 9 x KCH fragments
//     (16 couts x 32 input channels each; A operand) of all taps: 72 VGPRs for 64 -> 64 in bf16.  They are read once per workgroup,
//     from the arena pack_weights_h() fills, by the same 16-byte permutation as conv_k32_kernel.
//   * one input row per step: row iy is fetched (one step ahead, in registers), GroupNorm-applied, activated and split ONCE (halo
//     columns only: 1 + 2 / SW re-reads), written to one of two row slots in LDS (XOR-swizzled 16-byte units: every ds_read_b128 of a
//     fragment is conflict free, tests/test_k32_maps.py), and contributes to THREE output rows: its fragment of (kx, 32-channel
//     chunk, 16 pixels) is read once and multiplied with the ky = 0, 1, 2 weights into the accumulator sets of rows iy + 1, iy, iy - 1
//     (three fragment reads less per MFMA than the tile form).  After step iy row iy - 1 is complete: bias, noise shift, residual,
//     store, GroupNorm partial sums of the output, and its accumulator set becomes row iy + 2's.
//   * one barrier per step; no prologue / epilogue per tile: a strip segment of R rows costs R + 2 steps.
//
// Same ConvParams, same packed weights, same outputs and statistics layout ([N][tiles][C][2], one partial per strip segment) as the
// other 16-bit kernels; launch_conv_h asks conv_strip_ok() first.
#include "fdsr_kernels.h"
#include "fdsr_act_io.h"

#include <type_traits>

namespace fdsr {

typedef float s_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned s_u32x4 __attribute__((ext_vector_type(4)));   // (native vectors: arrays of HIP's uint4 struct were left in scratch memory)
typedef _Float16 s_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 s_b8 __attribute__((ext_vector_type(8)));

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N) -- the step's slots and items are indexed by true constants (a
// `#pragma unroll` over them gave up on the larger instantiations and left the weight fragments in scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ float silu_s(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

#ifndef STRIP_XS      // activation-fragment slots in registers: XS - 1 fragments ahead of their MFMAs
#define STRIP_XS 4
#endif
#ifndef STRIP_XS1     // ... of the one-workgroup-per-CU instantiations (512 registers: a single wave per SIMD has nobody to cover an LDS wait)
#define STRIP_XS1 4
#endif
#ifndef STRIP_DIAG    // timing-only diagnostic builds (results are garbage): 1 no residual loads, 2 no output stores, 4 no activation math
#define STRIP_DIAG 0
#endif
#ifndef STRIP_FENCE3  // fence behind each of a slot's three MFMA groups (the item's three stages stay between them)
#define STRIP_FENCE3 1
#endif

// One instantiation = (arithmetic, channels of the two concatenated input tensors C0 | C1, channels of the rider's two raw input
// tensors CR0 | CR1 or 0, 16-pixel blocks per strip row).  SW = 64 pixels: a staging pass of the 256 threads covers 32 channels of
// the row, so a tensor of C channels takes C / 32 passes.
template <int PREC, int C0_, int C1_, int CR0_, int CR1_, int NPH>
struct StripCfg {
  static constexpr int NP = PREC == PREC_F16X3 ? 2 : 1;
  static constexpr int ESZ = prec_is16(PREC) ? 2 : 4;     // bytes per activation element in HBM
  static constexpr int WNP = PREC == PREC_BF16 ? 1 : 2;   // planes per fragment of the weight arena the mode reads (PREC_F16: the f16x3 form, hi plane only)
  static constexpr int C0 = C0_, C1 = C1_, CIN = C0_ + C1_, KCH = CIN / 32, SW = 16 * NPH, HWD = SW + 2;
  static constexpr int OPP = CIN / 8;              // 8-channel units ("octs") per pixel and plane
  static constexpr int RB = 16 * OPP * NP;         // LDS bytes per staged pixel: [plane][oct] x 16 B, swizzled
  static constexpr int SLOT_BYTES = HWD * RB;
  static constexpr int NP0 = C0_ / 32, NP1 = C1_ / 32, NPM = NP0 + NP1;   // staging passes of the two input tensors
  static constexpr int NF = 3 * KCH * NPH;         // activation fragments per step (three MFMAs each)
  // rider: a 1x1 convolution over a second, raw input (the ResnetBlock's res_conv, unet.py:104-120) accumulated into the same rows
  static constexpr int CR0 = CR0_, CR1 = CR1_, CR = CR0_ + CR1_, KCR = CR / 32;
  static constexpr int RRB = CR * 2;               // bytes per pixel of a staged rider row (bf16 only, one plane, no halo)
  static constexpr int RSLOT_BYTES = SW * RRB;
  static constexpr int NR0 = CR0_ / 32, NR1 = CR1_ / 32, NPR = NR0 + NR1;
  static constexpr int NFR = KCR * NPH;            // rider fragments per step (one MFMA each)
  static constexpr int NT = NF + NFR;              // slots per step
  // LDS: two activation row slots | two rider row slots | GroupNorm table | two output-row tiles | two residual-row tiles
  static constexpr int RSLOT_OFF = 2 * SLOT_BYTES;
  static constexpr int GTAB_OFF = RSLOT_OFF + 2 * RSLOT_BYTES;
  static constexpr int OSZ = prec_is16(PREC) ? 2 : 4;     // bytes per output / residual element
  static constexpr int PB = 64 * OSZ;                  // bytes per pixel of a row tile
  static constexpr int UPP = PB / 16;                  // 16-byte units per pixel
  static constexpr int TILE_BYTES = SW * PB;           // one output (or residual) row of the strip, [pixel][64 couts], 16-byte units swizzled
  static constexpr int OUT_OFF = GTAB_OFF + 64 * OPP;
  static constexpr int RES_OFF = OUT_OFF + 2 * TILE_BYTES;
  static constexpr int NIT = SW * PB / 16 / 256;       // 16-byte units of a row tile per thread
  static_assert(NPH == 4, "a staging pass = 32 channels of a 64-pixel row");
  static_assert(C0_ % 32 == 0 && C1_ % 32 == 0 && CR0_ % 32 == 0 && CR1_ % 32 == 0 && C0_ > 0, "whole 32-channel chunks");
  static_assert(256 % (C0_ / 8) == 0 && (C1_ == 0 || 256 % (C1_ / 8) == 0) && (CR0_ == 0 || 256 % (CR0_ / 8) == 0) &&
                (CR1_ == 0 || 256 % (CR1_ / 8) == 0), "a thread keeps one oct index per tensor");
  static_assert(RB == 128 || RB == 256 || RB == 384 || RB == 512, "swizzles below are verified for these pixel strides");
  static_assert(CR == 0 || (prec_is16(PREC) && (RRB == 256 || RRB == 384)), "riders: the 2-byte modes (f16: launched only while the rider and the main conv share one weight scale, conv_strip_ok)");
  static_assert(NIT >= 1 && NIT <= 4 && 256 % UPP == 0, "row-tile map");
  static_assert(SLOT_BYTES + 16 * (NPH - 1) * RB < 65536 && RSLOT_BYTES < 65536, "fragment offsets must fit the ds_read immediate");
  // ---- the step's schedule: which item of vector work sits behind which slot's MFMAs ----
  // items 0 .. NIT-1: a flush unit; then per main pass 5 (four activation slices, write + re-fetch); a halo item per input tensor;
  // a write + re-fetch per rider pass; the residual item.  Spread evenly over the slots behind the epilogue quads (slots 0 .. NPH-1).
  static constexpr int NITEMS = NIT + 5 * NPM + (C1_ ? 2 : 1) + NPR + 1;
  static constexpr int slot_of(int j) { return NPH + (j * (NT - NPH)) / NITEMS; }
  static constexpr int first_item(int f) {       // the first item whose slot is >= f (items of slot f: [first_item(f), first_item(f + 1)))
    int j = 0;
    while (j < NITEMS && slot_of(j) < f) ++j;
    return j;
  }
};

// 16-byte unit XOR of a staged pixel (conflict-free ds_read_b128 / ds_write_b128 on the instruction's lane groups for all three
// kx shifts; replayed lane by lane in tests/test_k32_maps.py)
template <int RB>
__device__ __forceinline__ int strip_swz(int px) {
  return (RB == 128 || RB == 384) ? 2 * ((px >> 1) & 3) : 2 * (px & 7);
}

// what a thread fetches of one input row: an oct (eight channels of one pixel) per staging pass, two channels of a halo pixel
template <int PREC> struct StripRaw;
template <> struct StripRaw<PREC_BF16> {
  uint4 v;
  __device__ __forceinline__ void load(const unsigned char* ptr) { v = *reinterpret_cast<const uint4*>(ptr); }
  __device__ __forceinline__ void pair(int k, float& a, float& b) const {
    const unsigned u = k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
    a = __builtin_bit_cast(float, u << 16);
    b = __builtin_bit_cast(float, u & 0xffff0000u);
  }
};
template <> struct StripRaw<PREC_F16> {
  uint4 v;
  __device__ __forceinline__ void load(const unsigned char* ptr) { v = *reinterpret_cast<const uint4*>(ptr); }
  __device__ __forceinline__ void pair(int k, float& a, float& b) const {
    typedef _Float16 h2t __attribute__((ext_vector_type(2)));
    const h2t h = __builtin_bit_cast(h2t, k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)));
    a = (float)h[0];
    b = (float)h[1];
  }
};
template <> struct StripRaw<PREC_F16X3> {
  s_f32x4 a4, b4;
  __device__ __forceinline__ void load(const unsigned char* ptr) {
    a4 = *reinterpret_cast<const s_f32x4*>(ptr);
    b4 = *reinterpret_cast<const s_f32x4*>(ptr + 16);
  }
  __device__ __forceinline__ void pair(int k, float& a, float& b) const {
    a = k < 2 ? a4[2 * k] : b4[2 * k - 4];
    b = k < 2 ? a4[2 * k + 1] : b4[2 * k - 3];
  }
};
template <int PREC> struct StripRawPair;          // two channels (one slice) of a halo pixel
template <> struct StripRawPair<PREC_BF16> {
  unsigned v;
  __device__ __forceinline__ void load(const unsigned char* ptr) { v = *reinterpret_cast<const unsigned*>(ptr); }
  __device__ __forceinline__ void pair(float& a, float& b) const {
    a = __builtin_bit_cast(float, v << 16);
    b = __builtin_bit_cast(float, v & 0xffff0000u);
  }
};
template <> struct StripRawPair<PREC_F16> {
  unsigned v;
  __device__ __forceinline__ void load(const unsigned char* ptr) { v = *reinterpret_cast<const unsigned*>(ptr); }
  __device__ __forceinline__ void pair(float& a, float& b) const {
    typedef _Float16 h2t __attribute__((ext_vector_type(2)));
    const h2t h = __builtin_bit_cast(h2t, v);
    a = (float)h[0];
    b = (float)h[1];
  }
};
template <> struct StripRawPair<PREC_F16X3> {
  float2 v;
  __device__ __forceinline__ void load(const unsigned char* ptr) { v = *reinterpret_cast<const float2*>(ptr); }
  __device__ __forceinline__ void pair(float& a, float& b) const { a = v.x; b = v.y; }
};

// EVERY vector-memory instruction of the kernel is issued unconditionally (clamped addresses, results masked when used): after a
// branch that contains one the compiler's wait-count insertion no longer knows how many are in flight and falls back to
// s_waitcnt vmcnt(0) at the next use -- which here would wait for the row fetched three steps ahead.  (First versions of this
// kernel: 46 - 63 % of the wave cycles parked.)  Hence HAS_RES as a template parameter, the peeled first steps (no epilogue yet) and
// the halo columns split over all four waves (wave w activates slice w of both halo pixels of every row: no wave-dependent branch).
template <int PREC, int C0_, int C1_, int CR0_, int CR1_, int NPH, bool HAS_RES, int LB>   // LB: workgroups per CU the registers are budgeted for (256 / 512 VGPRs)
__global__ void __launch_bounds__(256, LB) conv_strip_kernel(const ConvParams p, const int seg_rows, const int wn_a) {
  using Cfg = StripCfg<PREC, C0_, C1_, CR0_, CR1_, NPH>;
  constexpr int NP = Cfg::NP, RB = Cfg::RB, SW = Cfg::SW, OPP = Cfg::OPP, KCH = Cfg::KCH, NF = Cfg::NF, ESZ = Cfg::ESZ;
  constexpr int NP0 = Cfg::NP0, NP1 = Cfg::NP1, NPM = Cfg::NPM, KCR = Cfg::KCR, NPR = Cfg::NPR, NR0 = Cfg::NR0, NFR = Cfg::NFR, NT = Cfg::NT;
  constexpr int RRB = Cfg::RRB;
  constexpr bool CAT = C1_ > 0, RIDER = Cfg::CR > 0;
  constexpr int XS = (LB == 1 ? STRIP_XS1 : STRIP_XS) < NF ? (LB == 1 ? STRIP_XS1 : STRIP_XS) : NF;
  using IO = ActIO<PREC>;
  typedef typename IO::Quad Quad;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c15 = lane & 15, g = lane >> 4;
  const int H = p.Hout, W = p.Wout;          // stride 1, padding 1: the input has the output's size (launcher)

  const int stripsX = W / SW, segs = (H + seg_rows - 1) / seg_rows;
  int bid;
  {   // consecutive strips on one XCD (as conv_k32_kernel)
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7, k = b >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int ncg = p.Cout / 64;               // groups of 64 output channels: one workgroup each (128-cout launches: two read the same rows)
  const int cg = bid % ncg;
  bid /= ncg;
  const int sx = bid % stripsX;
  bid /= stripsX;
  const int seg = bid % segs;
  const int n = bid / segs;
  // balanced segments: rows [seg H / segs, (seg + 1) H / segs) -- every segment has floor or ceil of H / segs rows (>= 4 for the
  // launcher's seg_rows >= 9; a remainder of ONE row, e.g. H = 136 at seg_rows = 9, would break the peeled steps and the tail below)
  const int ox0 = sx * SW, oy0 = (int)((long)seg * H / segs);
  const int oy1 = (int)((long)(seg + 1) * H / segs);          // this segment's output rows [oy0, oy1)

  // ---- staging identity per input tensor t (0: x0, 1: x1): thread -> oct ot[t] of that tensor's channels, pixel ppx[t] + i PPP[t] ----
  constexpr int OPPt[2] = {C0_ / 8, CAT ? C1_ / 8 : 1};
  const unsigned char* tb[2];      // this thread's channels of pixel (0, 0) of image n in tensor t
  int ot[2], ppx[2], og[2];        // og: the oct's index in the concatenated pixel row (GroupNorm table, LDS unit)
  ot[0] = tid % OPPt[0]; ppx[0] = tid / OPPt[0]; og[0] = ot[0];
  tb[0] = reinterpret_cast<const unsigned char*>(p.x0) + ((size_t)n * H * W * C0_ + 8 * ot[0]) * ESZ;
  if (CAT) {
    ot[1] = tid % OPPt[1]; ppx[1] = tid / OPPt[1]; og[1] = C0_ / 8 + ot[1];
    tb[1] = reinterpret_cast<const unsigned char*>(p.x1) + ((size_t)n * H * W * C1_ + 8 * ot[1]) * ESZ;
  } else { ot[1] = 0; ppx[1] = 0; og[1] = 0; tb[1] = tb[0]; }
  constexpr unsigned PIXB[2] = {C0_ * ESZ, (CAT ? C1_ : C0_) * ESZ};   // bytes per pixel of tensor t
  // GroupNorm scale / shift of image n as an LDS table [oct][slice] x (sc0, sc1, sh0, sh1): an activation slice reads its four values
  // when it runs instead of holding sixteen registers for the whole strip
  float* gtab = reinterpret_cast<float*>(smem_s + Cfg::GTAB_OFF);
  if (tid < 4 * OPP) {
    const int c = 8 * (tid >> 2) + 2 * (tid & 3);
    const float* ps = p.gn_scale + (size_t)n * Cfg::CIN + c;
    const float* ph = p.gn_shift + (size_t)n * Cfg::CIN + c;
    *reinterpret_cast<s_f32x4*>(gtab + 4 * tid) = s_f32x4{ps[0], ps[1], ph[0], ph[1]};
  }
  __syncthreads();
  // the two halo columns of a row (2 OPPt octs per tensor): wave w activates slice w (channels 2 w, 2 w + 1 of every oct) of both, lane
  // hl = lane % (2 OPPt) -> (column hl / OPPt, oct hl % OPPt == ot); the other lanes of the wave repeat them (same address, same value)
  int hdst[2];
  unsigned hcol[2];
  bool hok[2];
#pragma unroll
  for (int t = 0; t < (CAT ? 2 : 1); ++t) {
    const int hside = ((lane % (2 * OPPt[t])) / OPPt[t]) & 1;
    const int hix = hside ? ox0 + SW : ox0 - 1;
    hok[t] = hix >= 0 && hix < W;
    hcol[t] = (unsigned)(hok[t] ? hix : ox0) * PIXB[t] + 2 * w * ESZ;
    const int hpx = hside ? SW + 1 : 0;
    hdst[t] = hpx * RB + 16 * (og[t] ^ strip_swz<RB>(hpx)) + 4 * w;   // (the lo plane: ^ 16 OPP)
  }

  StripRaw<PREC> raw[3][NPM];        // three rows in flight: set = (row's step) % 3; passes of tensor 0, then of tensor 1
  StripRawPair<PREC> rawh[3][CAT ? 2 : 1];
  s_u32x4 rraw[3][RIDER ? NPR : 1];    // the rider's raw row, 16 bytes per pass
  // Row iy of the input, re-fetched into the register set of the row just staged (the row three steps on): the strip reads every
  // input byte exactly once, so its loads are bound by what is in flight per CU.  Rows outside the image or past the segment are
  // fetched from a clamped row (cache hits) and zeroed when staged.
  auto crow_in = [&](int iy) __attribute__((always_inline)) {
    const int hi = oy1 < H - 1 ? oy1 : H - 1;
    return iy < 0 ? 0 : (iy > hi ? hi : iy);
  };
  auto load_pass = [&](int iy, StripRaw<PREC>* set, int q) __attribute__((always_inline)) {   // pass q: tensor 0's NP0 passes, then tensor 1's
    const int t = q < NP0 ? 0 : 1, i = q < NP0 ? q : q - NP0;
    set[q].load(tb[t] + ((unsigned)(crow_in(iy) * W) + (unsigned)(ox0 + ppx[t] + i * (256 / OPPt[t]))) * PIXB[t]);
  };
  auto load_halo = [&](int iy, StripRawPair<PREC>* set, int t) __attribute__((always_inline)) {
    set[t].load(tb[t] + ((unsigned)(crow_in(iy) * W) * PIXB[t] + hcol[t]));
  };
  // rider rows (raw, no halo: a 1x1 convolution): pass r = 32 channels of the 64-pixel row, tensor xr0's passes first
  constexpr int OPRt[2] = {RIDER ? CR0_ / 8 : 1, RIDER && CR1_ ? CR1_ / 8 : 1};
  const unsigned char* rb[2] = {nullptr, nullptr};
  int rdst[RIDER ? NPR : 1];
  unsigned rsrc[RIDER ? NPR : 1];
  if (RIDER) {
#pragma unroll
    for (int r = 0; r < NPR; ++r) {
      const int t = r < NR0 ? 0 : 1, i = r < NR0 ? r : r - NR0;
      const int o = tid % OPRt[t], px = tid / OPRt[t] + i * (256 / OPRt[t]);
      const int ogr = (t ? CR0_ / 8 : 0) + o;
      rdst[r] = px * RRB + 16 * (ogr ^ strip_swz<RRB>(px));
      rsrc[r] = (unsigned)(ox0 + px) * (unsigned)((t ? CR1_ : CR0_) * 2) + 16 * o;
    }
    rb[0] = reinterpret_cast<const unsigned char*>(p.xr0) + (size_t)n * H * W * CR0_ * 2;
    rb[1] = CR1_ ? reinterpret_cast<const unsigned char*>(p.xr1) + (size_t)n * H * W * CR1_ * 2 : rb[0];
  }
  auto load_rider = [&](int iy, s_u32x4* set, int r) __attribute__((always_inline)) {
    const int t = r < NR0 ? 0 : 1;
    const int hi = oy1 - 1, rr = iy < oy0 ? oy0 : (iy > hi ? hi : iy);      // (rider rows are the segment's output rows)
    set[r] = *reinterpret_cast<const s_u32x4*>(rb[t] + ((unsigned)(rr * W) * (unsigned)((t ? CR1_ : CR0_) * 2) + rsrc[r]));
  };
  // GroupNorm apply + Swish (unet.py:89-101) + conversion, in SLICES of two channels (a slice is what the step schedule below places
  // between two MFMAs); a finished oct is written to its swizzled unit(s) of the slot
  struct Staged { unsigned hi[4]; unsigned lo[PREC == PREC_F16X3 ? 4 : 1]; };
  // ... in three stages, one behind each of a slot's three MFMA groups (an in-order wave issues the next MFMA only after the
  // vector instructions in front of it: one lump of ~17 behind three back-to-back MFMAs left the matrix pipe idle for their length).
  // Same operations in the same order as silu_s(a * sc + sh): bit-identical.
  struct ActTmp { float y0, y1, t0, t1; };
  auto act_s0 = [&](float a, float b, const float* gt4, ActTmp& m) __attribute__((always_inline)) {   // GroupNorm apply, exponent argument
    const s_f32x4 gt = *reinterpret_cast<const s_f32x4*>(gt4);
    m.y0 = a * gt[0] + gt[2];
    m.y1 = b * gt[1] + gt[3];
    m.t0 = m.y0 * -1.44269504088896340736f;
    m.t1 = m.y1 * -1.44269504088896340736f;
  };
  auto act_s1 = [&](ActTmp& m) __attribute__((always_inline)) {                                      // sigmoid: the four transcendentals
    if (!(STRIP_DIAG & 4)) {
      m.t0 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(m.t0));
      m.t1 = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(m.t1));
    }
  };
  auto act_s2 = [&](const ActTmp& m, unsigned& hi, unsigned& lo) __attribute__((always_inline)) {    // Swish product, conversion
    const float a = (STRIP_DIAG & 4) ? m.y0 : m.y0 * m.t0, b = (STRIP_DIAG & 4) ? m.y1 : m.y1 * m.t1;
    if (PREC == PREC_F16X3) {
      const float ca = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), cb = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
      typedef _Float16 h2t __attribute__((ext_vector_type(2)));
      const h2t h = {(_Float16)ca, (_Float16)cb};
      hi = __builtin_bit_cast(unsigned, h);
      // lo = f16(fma(hi, -1, v)), rounded once (as fdsr_conv_k32.hip)
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(ca));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(cb));
    } else if (PREC == PREC_F16) {
      typedef _Float16 h2t __attribute__((ext_vector_type(2)));
      const h2t h = {(_Float16)a, (_Float16)b};
      hi = __builtin_bit_cast(unsigned, h);
    } else {
      typedef __bf16 b2t __attribute__((ext_vector_type(2)));
      const b2t h = {(__bf16)a, (__bf16)b};
      hi = __builtin_bit_cast(unsigned, h);
    }
  };
  ActTmp atmp;
  auto act_slice = [&](const StripRaw<PREC>& r, Staged& d, int t, int k, int stage) __attribute__((always_inline)) {   // stage 0 .. 2, or -1: all
    if (stage <= 0) {
      float a, b;
      r.pair(k, a, b);
      act_s0(a, b, gtab + 16 * og[t] + 4 * k, atmp);
    }
    if (stage < 0 || stage == 1) act_s1(atmp);
    if (stage < 0 || stage == 2) act_s2(atmp, d.hi[k], d.lo[PREC == PREC_F16X3 ? k : 0]);
  };
  auto write_oct = [&](const Staged& d, bool ok, unsigned char* slot, int dofs) __attribute__((always_inline)) {
    const uint4 z = {0u, 0u, 0u, 0u};             // zero padding: rows outside the image
    *reinterpret_cast<uint4*>(slot + dofs) = ok ? uint4{d.hi[0], d.hi[1], d.hi[2], d.hi[3]} : z;
    if (PREC == PREC_F16X3)
      *reinterpret_cast<uint4*>(slot + (dofs ^ (16 * OPP))) = ok ? uint4{d.lo[0], d.lo[1], d.lo[2], d.lo[3]} : z;   // unit + OPP: the lo plane
  };
  auto stage_halo = [&](const StripRawPair<PREC>& r, int t, bool ok, unsigned char* slot, int stage) __attribute__((always_inline)) {
    if (stage <= 0) {
      float a, b;
      r.pair(a, b);
      act_s0(a, b, gtab + 16 * og[t] + 4 * w, atmp);
    }
    if (stage < 0 || stage == 1) act_s1(atmp);
    if (stage < 0 || stage == 2) {
      unsigned hi, lo = 0;
      act_s2(atmp, hi, lo);
      *reinterpret_cast<unsigned*>(slot + hdst[t]) = ok ? hi : 0u;
      if (PREC == PREC_F16X3) *reinterpret_cast<unsigned*>(slot + (hdst[t] ^ (16 * OPP))) = ok ? lo : 0u;
    }
  };
  int pdst[NPM];
#pragma unroll
  for (int q = 0; q < NPM; ++q) {
    const int t = q < NP0 ? 0 : 1, i = q < NP0 ? q : q - NP0;
    const int px = 1 + ppx[t] + i * (256 / OPPt[t]);
    pdst[q] = px * RB + 16 * (og[t] ^ strip_swz<RB>(px));
  }
  Staged stg;          // (one oct is in the making at a time: the schedule finishes a pass's four slices and its write before the next)

  // ---- weight fragments, resident: [ky][kx][32-channel chunk][plane]; lane (g, c15) = cout 16 w + c15, channels 32 kc + 8 g .. + 7 ----
  uint4 Wf[3][3][KCH][NP];
  uint4 Wr[RIDER ? KCR : 1];
  {
    const uint4* wq = reinterpret_cast<const uint4*>(p.wq);
    const int nk16 = p.Cin_pad / 16, co32 = 2 * cg + (w >> 1), cot = co32 / wn_a, wna = co32 % wn_a;
    const int wlane = 32 * (g & 1) + 16 * (w & 1) + c15;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const uint4* src = wq + ((((size_t)cot * nk16 + 2 * kc + (g >> 1)) * wn_a + wna) * 9 + tap) * (Cfg::WNP * 64) + wlane;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) Wf[tap / 3][tap % 3][kc][pl] = src[pl * 64];
      }
    if (RIDER) {   // the 1x1 conv's own fragments [cot][kc16][wn] (one tap)
      const uint4* wr = reinterpret_cast<const uint4*>(p.wq_r);
#pragma unroll
      for (int kc = 0; kc < KCR; ++kc) Wr[kc] = wr[(((size_t)cot * p.nkr + 2 * kc + (g >> 1)) * wn_a + wna) * (Cfg::WNP * 64) + wlane];
    }
  }

  // ---- activation fragment addresses inside a slot: lane -> pixel c15 + kx (+ 16 ph), channels 32 kc + 8 g .. + 7 ----
  int xa[3][KCH][NP];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) {
        const int px = c15 + kx;
        xa[kx][kc][pl] = px * RB + 16 * ((pl * OPP + 4 * kc + g) ^ strip_swz<RB>(px));
      }
  int xr[RIDER ? KCR : 1];     // rider fragments: pixel c15 (+ 16 ph) of the rider row, channels 32 kc + 8 g .. + 7
  if (RIDER) {
#pragma unroll
    for (int kc = 0; kc < KCR; ++kc) xr[kc] = c15 * RRB + 16 * ((4 * kc + g) ^ strip_swz<RRB>(c15));
  }

  s_f32x4 acc[3][NPH];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) acc[a][ph] = s_f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- epilogue: lane = pixel c15 (+ 16 ph) of the row, output channels cob .. cob + 3 ----
  const int cob = 64 * cg + 16 * w + 4 * g;     // (in the launch's Cout channels; inside this workgroup's 64: 16 w + 4 g)
  s_f32x4 add;
  {
    const float* tembp = p.temb ? p.temb + (size_t)n * p.temb_stride + p.temb_off : p.bias;   // (unconditional loads)
    const float tmul = p.temb ? 1.f : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = p.bias[cob + r];
      if (RIDER) a += p.bias_r[cob + r];
      add[r] = a + tmul * tembp[cob + r];
    }
  }
  const float winv = p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale;
  constexpr int OSZ = Cfg::OSZ, NIT = Cfg::NIT, TILE = Cfg::TILE_BYTES;
  s_f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  // The finished row leaves through LDS: a wave owns 16 of the 64 output channels, so its accumulator layout reaches memory in
  // 32-byte pieces (bf16) -- and the residual arrives in the same pieces.  Measured at B = 64: the residual launches took 400 us
  // against 300 us with EITHER their residual loads OR their stores left out, at exactly the algorithmic HBM traffic: the load /
  // store path was bound by the number of requests, not by bytes.  So: the accumulator lanes write their quads into a
  // [pixel][64 couts] row tile, and after the step's barrier every thread moves NIT whole 16-byte units of it (a pixel's 128 bytes =
  // 8 consecutive threads); the residual row comes the same way in reverse, two steps ahead.
  unsigned char* otile = smem_s + Cfg::OUT_OFF;
  unsigned char* rtile = smem_s + Cfg::RES_OFF;
  constexpr int PB = Cfg::PB, UPP = Cfg::UPP;
  // 16-byte unit u of pixel px sits at unit u ^ tsw(px): conflict free for the accumulator-side accesses (8 bytes per lane in bf16,
  // 16 in the fp32-output modes) and for the 16-byte row-side ones (replayed in tests/test_k32_maps.py)
  auto tsw = [](int px) { return OSZ == 2 ? ((px >> 1) & 7) : 2 * (px & 7); };
  const int aq = OSZ == 2 ? c15 * PB + 16 * (((4 * w + g) >> 1) ^ tsw(c15)) + 8 * (g & 1)      // + 16 PB ph: quad ph of this lane
                          : c15 * PB + 16 * ((4 * w + g) ^ tsw(c15));
  const unsigned GPB = (unsigned)p.Cout * OSZ;   // bytes per pixel of the output / residual tensors (the row tiles hold this workgroup's 64 channels: PB)
  int a128[NIT];
  unsigned gofs[NIT];            // byte offset of this thread's unit i inside an output row of the strip
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int px = tid / UPP + (256 / UPP) * i, u = tid % UPP;
    a128[i] = px * PB + 16 * (u ^ tsw(px));
    gofs[i] = (unsigned)((ox0 + px) * GPB + 64 * cg * OSZ + 16 * u);
  }
  const size_t img = (size_t)n * H * W * GPB;
  unsigned char* outn = reinterpret_cast<unsigned char*>(p.out) + img;
  const unsigned char* resn = reinterpret_cast<const unsigned char*>(HAS_RES ? p.res : p.out) + img;
  const uint4 z4 = {0u, 0u, 0u, 0u};
  uint4 rres0 = z4, rres1 = z4, rres2 = z4, rres3 = z4;   // (named register sets: an array here stayed in scratch memory)
  auto crow = [&](int oy) __attribute__((always_inline)) { return oy < oy0 ? oy0 : (oy >= oy1 ? oy1 - 1 : oy); };
  auto load_res = [&](int oy) __attribute__((always_inline)) {   // residual row oy (clamped into the segment), 16 bytes per unit
    if (HAS_RES && !(STRIP_DIAG & 1)) {
      const unsigned r0 = (unsigned)(crow(oy) * W) * GPB;
      rres0 = *reinterpret_cast<const uint4*>(resn + (r0 + gofs[0]));
      if (NIT > 1) rres1 = *reinterpret_cast<const uint4*>(resn + (r0 + gofs[NIT > 1 ? 1 : 0]));
      if (NIT > 2) rres2 = *reinterpret_cast<const uint4*>(resn + (r0 + gofs[NIT > 2 ? 2 : 0]));
      if (NIT > 3) rres3 = *reinterpret_cast<const uint4*>(resn + (r0 + gofs[NIT > 3 ? 3 : 0]));
    }
  };
  auto res_to_lds = [&](int oy) __attribute__((always_inline)) {
    if (HAS_RES) {
      unsigned char* t = rtile + (oy & 1) * TILE;
      *reinterpret_cast<uint4*>(t + a128[0]) = rres0;
      if (NIT > 1) *reinterpret_cast<uint4*>(t + a128[NIT > 1 ? 1 : 0]) = rres1;
      if (NIT > 2) *reinterpret_cast<uint4*>(t + a128[NIT > 2 ? 2 : 0]) = rres2;
      if (NIT > 3) *reinterpret_cast<uint4*>(t + a128[NIT > 3 ? 3 : 0]) = rres3;
    }
  };
  // one quad (pixel c15 + 16 ph of row oy, four output channels) of a finished row: residual from its tile, result into the row's tile
  auto finish = [&](int oy, int ph, s_f32x4 a) __attribute__((always_inline)) {
    s_f32x4 v = a * winv + add;
    if (HAS_RES) v += IO::widen(*reinterpret_cast<const Quad*>(rtile + (oy & 1) * TILE + aq + 16 * PB * ph));
    unsigned char* dst = otile + (oy & 1) * TILE + aq + 16 * PB * ph;
    if (OSZ == 2) {
      *reinterpret_cast<uint2*>(dst) = pack4_16<PREC>(v);   // (bf16: one v_cvt_pk_bf16_f32 per pair; f16: saturating)
    } else {
      *reinterpret_cast<s_f32x4*>(dst) = v;
    }
    s1 += v;
    s2 += v * v;
  };
  auto flush = [&](int oy, int i) __attribute__((always_inline)) {   // unit i of row oy's tile to memory (after the barrier that followed its epilogue)
    const uint4 v = *reinterpret_cast<const uint4*>(otile + (oy & 1) * TILE + a128[i]);
    if (!(STRIP_DIAG & 2)) *reinterpret_cast<uint4*>(outn + ((unsigned)(oy * W) * GPB + gofs[i])) = v;
  };

  // ---- one step: input row iy (staged in slot `cur`) into the three accumulator sets; row iy + 1 staged into `nxt`; row iy + 4
  // fetched; output row iy - 2 (finished by the step before) leaves its accumulators.  ROT = step % 3 names the accumulator sets and
  // the raw register sets statically: ky = 0 -> (ROT + 1) % 3 (a fresh row), ky = 1 -> ROT, ky = 2 -> (ROT + 2) % 3.
  // The step is NT slots of [next fragment read | the fragment's MFMAs (ky = 2, 1, 0; a rider fragment: one) | its items of vector
  // work], fenced: the activation slices of the next row, the epilogue quads of the row finished last step, the row tiles' traffic are
  // spread over the slots (Cfg::slot_of), so that no wave runs a long MFMA-free stretch.  EPI: the step runs the epilogue of a row
  // (steps >= 3 of a segment); FLUSH: it moves a row's tile to memory (steps >= 4).
  uint4 Xf[XS][NP];
  auto mfma1 = [&](const uint4* wf, const uint4* xf, s_f32x4 c) __attribute__((always_inline)) -> s_f32x4 {
    if (PREC == PREC_F16X3) {   // small terms first: lo(x) hi(w), hi(x) lo(w), hi(x) hi(w)
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s_h8, wf[0]), __builtin_bit_cast(s_h8, xf[NP - 1]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s_h8, wf[NP - 1]), __builtin_bit_cast(s_h8, xf[0]), c, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s_h8, wf[0]), __builtin_bit_cast(s_h8, xf[0]), c, 0, 0, 0);
    }
    return mfma16_k32<PREC>(wf[0], xf[0], c);
  };
  auto step = [&](int iy, int it, auto rot_tag, auto epi_tag, auto flush_tag) __attribute__((always_inline)) {
    constexpr int ROT = decltype(rot_tag)::value;
    constexpr bool EPI = decltype(epi_tag)::value, FLUSH = decltype(flush_tag)::value;
    constexpr int A0 = (ROT + 1) % 3, A1 = ROT, A2 = (ROT + 2) % 3;
    StripRaw<PREC>* rset = raw[(ROT + 1) % 3];             // holds row iy + 1; re-filled with row iy + 4
    StripRawPair<PREC>* hset = rawh[(ROT + 1) % 3];
    s_u32x4* rrs = rraw[(ROT + 1) % 3];
    unsigned char* cur = smem_s + (it & 1) * Cfg::SLOT_BYTES;
    unsigned char* nxt = smem_s + ((it & 1) ^ 1) * Cfg::SLOT_BYTES;
    unsigned char* rcur = smem_s + Cfg::RSLOT_OFF + (it & 1) * Cfg::RSLOT_BYTES;
    unsigned char* rnxt = smem_s + Cfg::RSLOT_OFF + ((it & 1) ^ 1) * Cfg::RSLOT_BYTES;
    const bool rok = iy + 1 >= 0 && iy + 1 < H;            // the row being staged lies inside the image
    auto load_x = [&](auto fc) __attribute__((always_inline)) {   // fragment f: (kx, kc, ph), ph fastest; f >= NF: the rider's (kc, ph)
      constexpr int f = decltype(fc)::value;
      if constexpr (f < NF) {
        constexpr int ph = f % NPH, kc = (f / NPH) % KCH, kx = f / (NPH * KCH);
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          Xf[f % XS][pl] = *reinterpret_cast<const uint4*>(cur + xa[kx][kc][pl] + ph * 16 * RB);
      } else if constexpr (f < NT) {
        constexpr int ph = (f - NF) % NPH, kc = (f - NF) / NPH;
        Xf[f % XS][0] = *reinterpret_cast<const uint4*>(rcur + xr[kc] + ph * 16 * RRB);
      }
    };
    auto item = [&](auto jc, auto sc) __attribute__((always_inline)) {   // item j of the step's vector work (order: Cfg::NITEMS), stage 0 .. 2 of its slot (-1: all)
      constexpr int j = decltype(jc)::value, stage = decltype(sc)::value;
      constexpr int J1 = NIT, J2 = J1 + 5 * NPM, J3 = J2 + (CAT ? 2 : 1), J4 = J3 + NPR;
      constexpr bool last = stage < 0 || stage == 2;       // items that are not activation math run behind the slot's last MFMA group
      if constexpr (j < J1) {
        if (FLUSH && last) flush(iy - 3, j);               // the row whose epilogue ran a step ago
      } else if constexpr (j < J2) {
        constexpr int q = (j - J1) / 5, k = (j - J1) % 5, t = q < NP0 ? 0 : 1;
        if constexpr (k < 4) act_slice(rset[q], stg, t, k, stage);
        else if constexpr (last) {
          write_oct(stg, rok, nxt, pdst[q]);
          load_pass(iy + 4, rset, q);
        }
      } else if constexpr (j < J3) {
        constexpr int t = j - J2;
        stage_halo(hset[t], t, hok[t] && rok, nxt, stage);
        if constexpr (last) load_halo(iy + 4, hset, t);
      } else if constexpr (j < J4) {
        constexpr int r = j - J3;
        if constexpr (last) {
          *reinterpret_cast<s_u32x4*>(rnxt + rdst[r]) = rrs[r];
          load_rider(iy + 4, rrs, r);
        }
      } else if constexpr (last) {
        res_to_lds(iy - 1);    // the residual of the row this step finishes goes into its tile; the next row's is fetched
        load_res(iy);
      }
    };
    static_for<0, XS - 1>(load_x);
    static_for<0, NT>([&](auto fc) __attribute__((always_inline)) {
      constexpr int f = decltype(fc)::value;
      load_x(std::integral_constant<int, f + XS - 1>{});
      // slots 0 .. NPH - 1: quad ph of the row the previous step finished leaves its accumulator (set A0: ky = 2 of the step
      // before) right before this slot's fresh product overwrites it
      if constexpr (EPI && f < NPH) finish(iy - 2, f, acc[A0][f]);
      constexpr int J0 = Cfg::first_item(f), J1e = Cfg::first_item(f + 1);
      static_assert(J1e - J0 <= 1, "one item per slot (the activation stages share one set of temporaries)");
      typedef std::integral_constant<int, 0> S0;
      typedef std::integral_constant<int, 1> S1;
      typedef std::integral_constant<int, 2> S2;
      typedef std::integral_constant<int, -1> SA;
      if constexpr (f < NF) {
        constexpr int ph = f % NPH, kc = (f / NPH) % KCH, kx = f / (NPH * KCH);
        constexpr bool fresh = kx == 0 && kc == 0;         // the first product of a new output row starts from zero
        const s_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        acc[A2][ph] = mfma1(Wf[2][kx][kc], Xf[f % XS], acc[A2][ph]);
        static_for<J0, J1e>([&](auto jc) __attribute__((always_inline)) { item(jc, S0{}); });
        if (STRIP_FENCE3) __builtin_amdgcn_sched_barrier(0);
        acc[A1][ph] = mfma1(Wf[1][kx][kc], Xf[f % XS], acc[A1][ph]);
        static_for<J0, J1e>([&](auto jc) __attribute__((always_inline)) { item(jc, S1{}); });
        if (STRIP_FENCE3) __builtin_amdgcn_sched_barrier(0);
        acc[A0][ph] = mfma1(Wf[0][kx][kc], Xf[f % XS], fresh ? zero : acc[A0][ph]);
        static_for<J0, J1e>([&](auto jc) __attribute__((always_inline)) { item(jc, S2{}); });
      } else {               // the rider's 1x1 product of input row iy belongs to output row iy
        constexpr int ph = (f - NF) % NPH, kc = (f - NF) / NPH;
        acc[A1][ph] = mfma16_k32<PREC>(Wr[kc], Xf[f % XS][0], acc[A1][ph]);
        static_for<J0, J1e>([&](auto jc) __attribute__((always_inline)) { item(jc, SA{}); });
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> R0;
  typedef std::integral_constant<int, 1> R1;
  typedef std::integral_constant<int, 2> R2;

  // ---- the strip segment: input rows oy0 - 1 .. oy1 ----
  {
    // rows oy0 - 1 (staged here), oy0, oy0 + 1, oy0 + 2 in flight before the first step; the set of a row = its step % 3
    StripRaw<PREC> first[NPM];
    StripRawPair<PREC> firsth[CAT ? 2 : 1];
    s_u32x4 firstr[RIDER ? NPR : 1];
#pragma unroll
    for (int q = 0; q < NPM; ++q) load_pass(oy0 - 1, first, q);
#pragma unroll
    for (int t = 0; t < (CAT ? 2 : 1); ++t) load_halo(oy0 - 1, firsth, t);
    if (RIDER) {
#pragma unroll
      for (int r = 0; r < NPR; ++r) load_rider(oy0 - 1, firstr, r);
    }
#pragma unroll
    for (int r = 1; r <= 3; ++r) {
#pragma unroll
      for (int q = 0; q < NPM; ++q) load_pass(oy0 - 1 + r, raw[r % 3], q);
#pragma unroll
      for (int t = 0; t < (CAT ? 2 : 1); ++t) load_halo(oy0 - 1 + r, rawh[r % 3], t);
      if (RIDER) {
#pragma unroll
        for (int q = 0; q < NPR; ++q) load_rider(oy0 - 1 + r, rraw[r % 3], q);
      }
    }
    const bool rok = oy0 - 1 >= 0;
#pragma unroll
    for (int q = 0; q < NPM; ++q) {
#pragma unroll
      for (int k = 0; k < 4; ++k) act_slice(first[q], stg, q < NP0 ? 0 : 1, k, -1);
      write_oct(stg, rok, smem_s, pdst[q]);
    }
#pragma unroll
    for (int t = 0; t < (CAT ? 2 : 1); ++t) stage_halo(firsth[t], t, hok[t] && rok, smem_s, -1);
    if (RIDER) {
#pragma unroll
      for (int r = 0; r < NPR; ++r) *reinterpret_cast<s_u32x4*>(smem_s + Cfg::RSLOT_OFF + rdst[r]) = firstr[r];
    }
  }
  __syncthreads();
  const int nsteps = oy1 - oy0 + 2;          // >= 5 (segments have at least 3 rows: balanced above, conv_strip_ok for the whole map)
  load_res(oy0);
  step(oy0 - 1, 0, R0{}, std::false_type{}, std::false_type{});
  step(oy0, 1, R1{}, std::false_type{}, std::false_type{});
  step(oy0 + 1, 2, R2{}, std::false_type{}, std::false_type{});
  step(oy0 + 2, 3, R0{}, std::true_type{}, std::false_type{});
  for (int it = 4; it < nsteps; it += 3) {
    step(oy0 - 1 + it, it, R1{}, std::true_type{}, std::true_type{});
    if (it + 1 < nsteps) step(oy0 + it, it + 1, R2{}, std::true_type{}, std::true_type{});
    if (it + 2 < nsteps) step(oy0 + 1 + it, it + 2, R0{}, std::true_type{}, std::true_type{});
  }
  // the last two rows: row oy1 - 2 waits in its tile; row oy1 - 1 sits in the last step's ky = 2 set (its residual tile was written there)
#pragma unroll
  for (int i = 0; i < NIT; ++i) flush(oy1 - 2, i);
  {
    const int last = ((nsteps - 1) % 3 + 2) % 3;
#pragma unroll
    for (int q = 0; q < NPH; ++q) finish(oy1 - 1, q, last == 0 ? acc[0][q] : (last == 1 ? acc[1][q] : acc[2][q]));
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NIT; ++i) flush(oy1 - 1, i);

  // ---- GroupNorm partial sums of this segment's outputs: the 16 pixel lanes of a k group fold into lane c15 == 0 ----
  if (p.part_out) {
    float vals[8] = {s1[0], s2[0], s1[1], s2[1], s1[2], s2[2], s1[3], s2[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int m = 8; m >= 1; m >>= 1) vals[e] += __shfl_xor(vals[e], m, 64);
    if (c15 == 0) {
      float* dst = p.part_out + (((size_t)n * (segs * stripsX) + seg * stripsX + sx) * p.Cout + cob) * 2;
      *reinterpret_cast<s_f32x4*>(dst) = s_f32x4{vals[0], vals[1], vals[2], vals[3]};
      *reinterpret_cast<s_f32x4*>(dst + 4) = s_f32x4{vals[4], vals[5], vals[6], vals[7]};
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The instantiations, and where they are compiled: each takes 15 - 30 s of hipcc, so the file is compiled THREE times in parallel
// (fastdiffsr_amd/build.py): -DSTRIP_PART=0 (default) the host side and the f16x3 kernels, =2 the bf16 kernels, =3 the f16 kernels;
// part 0 only declares the others (extern template: their host stubs and device code live in the other two objects).
#ifndef STRIP_PART
#define STRIP_PART 0
#endif
#define STRIP_INSTANCES_F16X3(X) X(PREC_F16X3, 64, 0, 0, 0, false, 1) X(PREC_F16X3, 64, 0, 0, 0, true, 1)
#define STRIP_INSTANCES_16(X, P)                                                                                  \
  X(P, 64, 0, 0, 0, false, 2) X(P, 64, 0, 0, 0, true, 2) X(P, 64, 0, 0, 0, false, 1) X(P, 64, 0, 0, 0, true, 1)      \
  X(P, 64, 64, 0, 0, false, 1) X(P, 128, 64, 0, 0, false, 1) X(P, 64, 0, 64, 64, false, 1) X(P, 64, 0, 128, 64, false, 1) \
  X(P, 128, 0, 0, 0, false, 1) X(P, 128, 0, 0, 0, true, 1)
#define STRIP_DEFINE(P, A, B, C, D, R, L) template __global__ void conv_strip_kernel<P, A, B, C, D, 4, R, L>(const ConvParams, const int, const int);
#define STRIP_DECLARE(P, A, B, C, D, R, L) extern template __global__ void conv_strip_kernel<P, A, B, C, D, 4, R, L>(const ConvParams, const int, const int);
#if STRIP_PART == 2
STRIP_INSTANCES_16(STRIP_DEFINE, PREC_BF16)
#elif STRIP_PART == 3
STRIP_INSTANCES_16(STRIP_DEFINE, PREC_F16)
#else
STRIP_INSTANCES_16(STRIP_DECLARE, PREC_BF16)
STRIP_INSTANCES_16(STRIP_DECLARE, PREC_F16)
#endif

#if STRIP_PART == 0
// ---------------------------------------------------------------------------------------------------------------------------
// g_tun.strip bits: 1 bf16 64 -> 64 launches (two workgroups per CU), 2 the f16x3 ones (one per CU: hi / lo weight planes = 144
// registers), 4 (A/B) bf16 64 -> 64 on one workgroup per CU, 8 the bf16 launches with a concatenated input (64 + 64 -> 64,
// 128 + 64 -> 64 under bit 32: 144 / 216 weight registers, one workgroup per CU), 16 the bf16 launches with a res_conv rider,
// 64 the bf16 128-cout launches 128 -> 128 and 64 -> 128 (two workgroups of 64 couts per strip)
static bool strip_wide(const ConvParams& p, int prec) { return !prec_is16(prec) || (g_tun.strip & FDSR_STRIP_BF16_ONE_WG) || p.C1 || p.xr0 || p.C0 > 64; }
static long strip_min_wgs(const ConvParams& p, int prec) { return strip_wide(p, prec) ? g_tun.strip_min_wgs / 2 : g_tun.strip_min_wgs; }

static int strip_seg_rows(const ConvParams& p, int SW, int prec) {
  // strips x segments: enough workgroups to fill the chip (two per CU where the registers allow), segments as long as that allows
  // (each costs two extra steps)
  const long strips = (long)p.N * ((p.Wout + SW - 1) / SW) * (p.Cout / 64);
  int rows = p.Hout;
  while (rows > 16 && strips * ((p.Hout + rows - 1) / rows) < strip_min_wgs(p, prec)) rows = (rows + 1) / 2;
  return rows;
}

bool conv_strip_ok(ConvKind kind, int prec, const ConvParams& p) {
  if (kind != CONV3_S1 || p.ksplit > 1 || (p.Cout != 64 && p.Cout != 128) || p.Cout_pad != p.Cout) return false;
  if (!p.gn_scale || p.gn_plain || p.drop_mask) return false;
  if (p.Hin != p.Hout || p.Win != p.Wout || p.Wout % 64 || p.Hout < 3 || (size_t)p.Hout * p.Wout * 192 * 4 >= (1ull << 31)) return false;
  if (prec_is16(prec) && p.out_f32) return false;
  const int Cin = p.C0 + p.C1;
  if (Cin != p.Cin_pad) return false;
  bool shape;
  if (p.Cout == 128) {   // the 128-cout level: two workgroups of 64 couts read the same rows; 128 -> 128 (144 weight registers) and 64 -> 128, bf16
    shape = prec_is16(prec) && (g_tun.strip & FDSR_STRIP_BF16_COUT128) && !p.xr0 && p.C1 == 0 && (p.C0 == 64 || p.C0 == 128);
  } else if (p.xr0) {        // 64 -> 64 with a rider over (64 | 64) or (128 | 64) raw channels
    // (f16: the two products share the accumulators, so the rider's power-of-two weight scale must be the main conv's -- both are 2^12 unless a
    // tensor holds weights beyond 8; otherwise the launch stays on the tile kernels, which re-scale between the two phases)
    if (prec == PREC_F16 && p.w_inv_scale > 0.f && p.w_inv_scale_r > 0.f && p.w_inv_scale_r != p.w_inv_scale) return false;   // (a shape query without scales: assume the form)
    shape = prec_is16(prec) && (g_tun.strip & FDSR_STRIP_BF16_RIDER) && !p.res && p.C0 == 64 && p.C1 == 0 && p.Cr1 == 64 && (p.Cr0 == 64 || p.Cr0 == 128) &&
            p.nkr * 16 == p.Cr0 + p.Cr1;
  } else if (p.C1) {  // concatenated input (64 | 64) or (128 | 64), no residual (block1 of the up path)
    shape = prec_is16(prec) && !p.res && p.C1 == 64 && ((p.C0 == 64 && (g_tun.strip & FDSR_STRIP_BF16_CAT64)) || (p.C0 == 128 && (g_tun.strip & FDSR_STRIP_BF16_CAT128)));   // (bit 32: 216 weight registers, spills)
  } else {
    shape = p.C0 == 64 && (g_tun.strip & (prec_is16(prec) ? FDSR_STRIP_BF16_64 : FDSR_STRIP_F16X3_64));
  }
  if (!shape) return false;
  const long wgs = (long)p.N * ((p.Wout + 63) / 64) * ((p.Hout + 15) / 16) * (p.Cout / 64);
  return wgs >= strip_min_wgs(p, prec);   // (a small grid keeps the split-K tile kernels)
}

template <int PREC, int C0_, int C1_, int CR0_, int CR1_, bool HAS_RES, int LB>
static hipError_t launch_strip_t(const ConvParams& p, int wn_a, hipStream_t s, int* tiles) {
  using Cfg = StripCfg<PREC, C0_, C1_, CR0_, CR1_, 4>;
  const int rows = strip_seg_rows(p, Cfg::SW, PREC);
  const int stripsX = (p.Wout + Cfg::SW - 1) / Cfg::SW, segs = (p.Hout + rows - 1) / rows;
  if (tiles) *tiles = stripsX * segs;
  const size_t lds = (size_t)Cfg::RES_OFF + (HAS_RES ? 2 * Cfg::TILE_BYTES : 0);
  hipLaunchKernelGGL((conv_strip_kernel<PREC, C0_, C1_, CR0_, CR1_, 4, HAS_RES, LB>), dim3(p.N * stripsX * segs * (p.Cout / 64)), dim3(256), lds, s, p, rows, wn_a);
  return hipGetLastError();
}

hipError_t launch_conv_strip(int prec, const ConvParams& p, int wn_a, hipStream_t s, int* tiles) {
  if (prec == PREC_F16X3) {
    return p.res ? launch_strip_t<PREC_F16X3, 64, 0, 0, 0, true, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_F16X3, 64, 0, 0, 0, false, 1>(p, wn_a, s, tiles);
  }
  if (prec == PREC_F16) {   // the bf16 mode's forms
    if (p.xr0)
      return p.Cr0 == 64 ? launch_strip_t<PREC_F16, 64, 0, 64, 64, false, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_F16, 64, 0, 128, 64, false, 1>(p, wn_a, s, tiles);
    if (p.C0 == 128 && !p.C1)
      return p.res ? launch_strip_t<PREC_F16, 128, 0, 0, 0, true, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_F16, 128, 0, 0, 0, false, 1>(p, wn_a, s, tiles);
    if (p.C1)
      return p.C0 == 64 ? launch_strip_t<PREC_F16, 64, 64, 0, 0, false, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_F16, 128, 64, 0, 0, false, 1>(p, wn_a, s, tiles);
    if (g_tun.strip & FDSR_STRIP_BF16_ONE_WG)
      return p.res ? launch_strip_t<PREC_F16, 64, 0, 0, 0, true, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_F16, 64, 0, 0, 0, false, 1>(p, wn_a, s, tiles);
    return p.res ? launch_strip_t<PREC_F16, 64, 0, 0, 0, true, 2>(p, wn_a, s, tiles) : launch_strip_t<PREC_F16, 64, 0, 0, 0, false, 2>(p, wn_a, s, tiles);
  }
  if (prec != PREC_BF16) return hipErrorInvalidValue;
  if (p.C0 == 128 && !p.C1)   // (Cout 128)
    return p.res ? launch_strip_t<PREC_BF16, 128, 0, 0, 0, true, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_BF16, 128, 0, 0, 0, false, 1>(p, wn_a, s, tiles);
  if (p.xr0) {
    return p.Cr0 == 64 ? launch_strip_t<PREC_BF16, 64, 0, 64, 64, false, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_BF16, 64, 0, 128, 64, false, 1>(p, wn_a, s, tiles);
  }
  if (p.C1) {
    return p.C0 == 64 ? launch_strip_t<PREC_BF16, 64, 64, 0, 0, false, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_BF16, 128, 64, 0, 0, false, 1>(p, wn_a, s, tiles);
  }
  if (g_tun.strip & FDSR_STRIP_BF16_ONE_WG)
    return p.res ? launch_strip_t<PREC_BF16, 64, 0, 0, 0, true, 1>(p, wn_a, s, tiles) : launch_strip_t<PREC_BF16, 64, 0, 0, 0, false, 1>(p, wn_a, s, tiles);
  return p.res ? launch_strip_t<PREC_BF16, 64, 0, 0, 0, true, 2>(p, wn_a, s, tiles) : launch_strip_t<PREC_BF16, 64, 0, 0, 0, false, 2>(p, wn_a, s, tiles);
}

template <int PREC, int C0_, int C1_, int CR0_, int CR1_, bool HAS_RES, int LB>
static hipError_t init_strip_t() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_strip_kernel<PREC, C0_, C1_, CR0_, CR1_, 4, HAS_RES, LB>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t kernels_strip_init() {
  hipError_t e;
#define X(P, A, B, C, D, R, L) if ((e = init_strip_t<P, A, B, C, D, R, L>()) != hipSuccess) return e;
  STRIP_INSTANCES_F16X3(X)
  STRIP_INSTANCES_16(X, PREC_BF16)
  STRIP_INSTANCES_16(X, PREC_F16)
#undef X
  return hipSuccess;
}
#endif   // STRIP_PART == 0

}  // namespace fdsr
