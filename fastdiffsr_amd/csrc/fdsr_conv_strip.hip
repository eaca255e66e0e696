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
typedef _Float16 s_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 s_b8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float silu_s(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

#ifndef STRIP_XS      // activation-fragment slots in registers: XS - 1 fragments ahead of their MFMAs
#define STRIP_XS 4
#endif
#ifndef STRIP_DIAG    // timing-only diagnostic builds (results are garbage): 1 no residual loads, 2 no output stores, 4 no activation math
#define STRIP_DIAG 0
#endif
#ifndef STRIP_PIN     // keep the fragment reads ahead of the MFMAs (sched_barrier that VALU / SALU may cross)
#define STRIP_PIN 1
#endif

template <int PREC, int KCH, int NPH>
struct StripCfg {
  static constexpr int NP = PREC == PREC_F16X3 ? 2 : 1;
  static constexpr int CIN = 32 * KCH, SW = 16 * NPH, HWD = SW + 2;
  static constexpr int OPP = CIN / 8;              // 8-channel units ("octs") per pixel and plane
  static constexpr int RB = 16 * OPP * NP;         // LDS bytes per staged pixel: [plane][oct] x 16 B, swizzled
  static constexpr int SLOT_BYTES = HWD * RB;
  static constexpr int GTAB_OFF = 2 * SLOT_BYTES;      // GroupNorm table: 64 OPP bytes
  static constexpr int OSZ = PREC == PREC_BF16 ? 2 : 4;   // bytes per output / residual element
  static constexpr int TILE_BYTES = SW * 64 * OSZ;     // one output (or residual) row of the strip, [pixel][64 couts], 16-byte units swizzled
  static constexpr int OUT_OFF = GTAB_OFF + 64 * OPP;  // two output-row tiles, then two residual-row tiles
  static constexpr int RES_OFF = OUT_OFF + 2 * TILE_BYTES;
  static constexpr int LDS_BYTES = RES_OFF + 2 * TILE_BYTES;
  static constexpr int NIT = SW * 64 * OSZ / 16 / 256; // 16-byte units of a row tile per thread
  static constexpr int PB = 64 * OSZ;                  // bytes per pixel of a row tile
  static constexpr int UPP = PB / 16;                  // 16-byte units per pixel
  static_assert(NIT >= 1 && NIT <= 4 && 256 % UPP == 0, "row-tile map");
  static constexpr int PPP = 256 / OPP;            // pixels staged per pass
  static constexpr int NPASS = SW / PPP;
  static constexpr int NF = 3 * KCH * NPH;         // activation fragments per step
  static_assert(256 % OPP == 0 && SW % PPP == 0 && 64 % OPP == 0, "staging map");
  static_assert(RB == 128 || RB == 256 || RB == 512, "swizzles below are verified for these pixel strides");
  static_assert(SLOT_BYTES + 16 * (NPH - 1) * RB < 65536, "fragment offsets must fit the ds_read immediate");
};

// 16-byte unit XOR of a staged pixel (conflict-free ds_read_b128 / ds_write_b128 on the instruction's lane groups for all three
// kx shifts; replayed lane by lane in tests/test_k32_maps.py)
template <int RB>
__device__ __forceinline__ int strip_swz(int px) {
  return RB == 128 ? 2 * ((px >> 1) & 3) : 2 * (px & 7);
}

// what a thread fetches of one input row: an oct (eight channels of one pixel) per staging pass, two channels of a halo pixel
template <int PREC> struct StripRaw;
template <> struct StripRaw<PREC_BF16> {
  uint4 v;
  static constexpr int ESZ = 2;
  __device__ __forceinline__ void load(const unsigned char* ptr) { v = *reinterpret_cast<const uint4*>(ptr); }
  __device__ __forceinline__ void pair(int k, float& a, float& b) const {
    const unsigned u = k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w));
    a = __builtin_bit_cast(float, u << 16);
    b = __builtin_bit_cast(float, u & 0xffff0000u);
  }
};
template <> struct StripRaw<PREC_F16X3> {
  s_f32x4 a4, b4;
  static constexpr int ESZ = 4;
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
template <int PREC, int KCH, int NPH, bool HAS_RES, int LB = 2>   // LB: workgroups per CU the registers are budgeted for (256 / 512 VGPRs)
__global__ void __launch_bounds__(256, LB) conv_strip_kernel(const ConvParams p, const int seg_rows, const int wn_a) {
  using Cfg = StripCfg<PREC, KCH, NPH>;
  constexpr int NP = Cfg::NP, RB = Cfg::RB, SW = Cfg::SW, OPP = Cfg::OPP, PPP = Cfg::PPP, NPASS = Cfg::NPASS, NF = Cfg::NF;
  constexpr int XS = STRIP_XS < NF ? STRIP_XS : NF;
  constexpr int ESZ = StripRaw<PREC>::ESZ;
  using IO = ActIO<PREC>;
  typedef typename IO::Quad Quad;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, c15 = lane & 15, g = lane >> 4;
  const int Cin = p.C0 + p.C1;
  const int H = p.Hout, W = p.Wout;          // stride 1, padding 1: the input has the output's size (launcher)

  const int stripsX = W / SW, segs = (H + seg_rows - 1) / seg_rows;
  int bid;
  {   // consecutive strips on one XCD (as conv_k32_kernel)
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7, k = b >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int sx = bid % stripsX;
  bid /= stripsX;
  const int seg = bid % segs;
  const int n = bid / segs;
  const int ox0 = sx * SW, oy0 = seg * seg_rows;
  const int oy1 = oy0 + seg_rows < H ? oy0 + seg_rows : H;   // this segment's output rows [oy0, oy1)

  // ---- staging identity: thread -> (oct o of the input channels, pixel pp + i PPP of the strip row) ----
  const int o = tid % OPP, pp = tid / OPP;
  const unsigned char* tb;     // this thread's channels of pixel (0, 0) of image n
  unsigned pixb;               // bytes per pixel of the tensor it reads
  if (8 * o < p.C0) {
    tb = reinterpret_cast<const unsigned char*>(p.x0) + ((size_t)n * H * W * p.C0 + 8 * o) * ESZ;
    pixb = p.C0 * ESZ;
  } else {
    tb = reinterpret_cast<const unsigned char*>(p.x1) + ((size_t)n * H * W * p.C1 + 8 * o - p.C0) * ESZ;
    pixb = p.C1 * ESZ;
  }
  // GroupNorm scale / shift of image n as an LDS table [oct][slice] x (sc0, sc1, sh0, sh1): an activation slice reads its four values
  // when it runs instead of holding sixteen registers for the whole strip
  float* gtab = reinterpret_cast<float*>(smem_s + Cfg::GTAB_OFF);
  if (tid < 4 * OPP) {
    const int c = 8 * (tid >> 2) + 2 * (tid & 3);
    const float* ps = p.gn_scale + (size_t)n * Cin + c;
    const float* ph = p.gn_shift + (size_t)n * Cin + c;
    *reinterpret_cast<s_f32x4*>(gtab + 4 * tid) = s_f32x4{ps[0], ps[1], ph[0], ph[1]};
  }
  __syncthreads();
  const float* gmine = gtab + 16 * o;
  // the two halo columns of a row (2 OPP octs): wave w activates slice w (channels 2 w, 2 w + 1 of every oct) of both, lane hl =
  // lane % (2 OPP) -> (column hl / OPP, oct hl % OPP == o); the other lanes of the wave repeat them (same address, same value)
  const int hside = ((lane % (2 * OPP)) / OPP) & 1;
  const int hix = hside ? ox0 + SW : ox0 - 1;
  const bool hok = hix >= 0 && hix < W;
  const unsigned hcol = (unsigned)(hok ? hix : ox0) * pixb + 2 * w * ESZ;
  const int hpx = hside ? SW + 1 : 0;
  const int hdst = hpx * RB + 16 * (o ^ strip_swz<RB>(hpx)) + 4 * w;   // (the lo plane: ^ 16 OPP)

  StripRaw<PREC> raw[3][NPASS];      // three rows in flight: set = (row's step) % 3
  StripRawPair<PREC> rawh[3];
  // Row iy of the input, re-fetched into the register set of the row just staged (the row three steps on): the strip reads every
  // input byte exactly once, so its loads are bound by what is in flight per CU.  Rows outside the image or past the segment are
  // fetched from a clamped row (cache hits) and zeroed when staged.
  auto row_off = [&](int iy) __attribute__((always_inline)) {
    const int hi = oy1 < H - 1 ? oy1 : H - 1;
    const int r = iy < 0 ? 0 : (iy > hi ? hi : iy);
    return (unsigned)(r * W) * pixb;
  };
  auto load_pass = [&](int iy, StripRaw<PREC>* set, int i) __attribute__((always_inline)) {
    set[i].load(tb + (row_off(iy) + (unsigned)(ox0 + pp + i * PPP) * pixb));
  };
  auto load_halo = [&](int iy, StripRawPair<PREC>& d) __attribute__((always_inline)) { d.load(tb + (row_off(iy) + hcol)); };
  // GroupNorm apply + Swish (unet.py:89-101) + conversion, in SLICES of two channels (a slice is what the step schedule below places
  // between two MFMAs); a finished oct is written to its swizzled unit(s) of the slot
  struct Staged { unsigned hi[4]; unsigned lo[PREC == PREC_F16X3 ? 4 : 1]; };
  auto act_pair = [&](float a, float b, int k, unsigned& hi, unsigned& lo) __attribute__((always_inline)) {
    const s_f32x4 gt = *reinterpret_cast<const s_f32x4*>(gmine + 4 * k);
    if (STRIP_DIAG & 4) {
      a = a * gt[0] + gt[2];
      b = b * gt[1] + gt[3];
    } else {
      a = silu_s(a * gt[0] + gt[2]);
      b = silu_s(b * gt[1] + gt[3]);
    }
    if (PREC == PREC_F16X3) {
      const float ca = __builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), cb = __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
      typedef _Float16 h2t __attribute__((ext_vector_type(2)));
      const h2t h = {(_Float16)ca, (_Float16)cb};
      hi = __builtin_bit_cast(unsigned, h);
      // lo = f16(fma(hi, -1, v)), rounded once (as fdsr_conv_k32.hip)
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(ca));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(cb));
    } else {
      typedef __bf16 b2t __attribute__((ext_vector_type(2)));
      const b2t h = {(__bf16)a, (__bf16)b};
      hi = __builtin_bit_cast(unsigned, h);
    }
  };
  auto act_slice = [&](const StripRaw<PREC>& r, Staged& d, int k) __attribute__((always_inline)) {
    float a, b;
    r.pair(k, a, b);
    act_pair(a, b, k, d.hi[k], d.lo[PREC == PREC_F16X3 ? k : 0]);
  };
  auto write_oct = [&](const Staged& d, bool ok, unsigned char* slot, int dofs) __attribute__((always_inline)) {
    const uint4 z = {0u, 0u, 0u, 0u};             // zero padding: rows outside the image
    *reinterpret_cast<uint4*>(slot + dofs) = ok ? uint4{d.hi[0], d.hi[1], d.hi[2], d.hi[3]} : z;
    if (PREC == PREC_F16X3)
      *reinterpret_cast<uint4*>(slot + (dofs ^ (16 * OPP))) = ok ? uint4{d.lo[0], d.lo[1], d.lo[2], d.lo[3]} : z;   // unit + OPP: the lo plane
  };
  auto stage_halo = [&](const StripRawPair<PREC>& r, bool ok, unsigned char* slot) __attribute__((always_inline)) {
    float a, b;
    r.pair(a, b);
    unsigned hi, lo = 0;
    act_pair(a, b, w, hi, lo);
    *reinterpret_cast<unsigned*>(slot + hdst) = ok ? hi : 0u;
    if (PREC == PREC_F16X3) *reinterpret_cast<unsigned*>(slot + (hdst ^ (16 * OPP))) = ok ? lo : 0u;
  };
  int pdst[NPASS];
#pragma unroll
  for (int i = 0; i < NPASS; ++i) {
    const int px = 1 + pp + i * PPP;
    pdst[i] = px * RB + 16 * (o ^ strip_swz<RB>(px));
  }
  Staged stg[NPASS];

  // ---- weight fragments, resident: [ky][kx][32-channel chunk][plane]; lane (g, c15) = cout 16 w + c15, channels 32 kc + 8 g .. + 7 ----
  uint4 Wf[3][3][KCH][NP];
  {
    const uint4* wq = reinterpret_cast<const uint4*>(p.wq);
    const int nk16 = p.Cin_pad / 16, co32 = w >> 1, cot = co32 / wn_a, wna = co32 % wn_a;
    const int wlane = 32 * (g & 1) + 16 * (w & 1) + c15;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const uint4* src = wq + ((((size_t)cot * nk16 + 2 * kc + (g >> 1)) * wn_a + wna) * 9 + tap) * (NP * 64) + wlane;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) Wf[tap / 3][tap % 3][kc][pl] = src[pl * 64];
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

  s_f32x4 acc[3][NPH];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) acc[a][ph] = s_f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- epilogue: lane = pixel c15 (+ 16 ph) of the row, output channels cob .. cob + 3 ----
  const int cob = 16 * w + 4 * g;
  s_f32x4 add;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float a = p.bias[cob + r];
    if (p.temb) a += p.temb[(size_t)n * p.temb_stride + p.temb_off + cob + r];
    add[r] = a;
  }
  const float winv = p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale;
  constexpr int OSZ = Cfg::OSZ, NIT = Cfg::NIT, TILE = Cfg::TILE_BYTES;
  s_f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  // The finished row leaves through LDS: a wave owns 16 of the 64 output channels, so its accumulator layout reaches memory in
  // 32-byte pieces (bf16) -- and the residual arrives in the same pieces.  Measured at B = 64: the residual launches took 400 us
  // against 300 us with EITHER their residual loads OR their stores left out, at exactly the algorithmic HBM traffic: the load /
  // store path was bound by the number of requests, not by bytes.  So: the accumulator lanes write their quads (8 bytes) into a
  // [pixel][64 couts] row tile, and after the step's barrier every thread moves NIT whole 16-byte units of it (a pixel's 128 bytes =
  // 8 consecutive threads); the residual row comes the same way in reverse, two steps ahead.  16-byte unit u of pixel px sits at
  // unit u ^ ((px >> 1) & 7): conflict free for the 8-byte accumulator-side accesses and the 16-byte row-side ones.
  unsigned char* otile = smem_s + Cfg::OUT_OFF;
  unsigned char* rtile = smem_s + Cfg::RES_OFF;
  constexpr int PB = Cfg::PB, UPP = Cfg::UPP;
  // 16-byte unit u of pixel px sits at unit u ^ tsw(px): conflict free for the accumulator-side accesses (8 bytes per lane in bf16,
  // 16 in the fp32-output modes) and for the 16-byte row-side ones (replayed in tests/test_k32_maps.py)
  auto tsw = [](int px) { return OSZ == 2 ? ((px >> 1) & 7) : 2 * (px & 7); };
  const int aq = OSZ == 2 ? c15 * PB + 16 * (((4 * w + g) >> 1) ^ tsw(c15)) + 8 * (g & 1)      // + 16 PB ph: quad ph of this lane
                          : c15 * PB + 16 * ((4 * w + g) ^ tsw(c15));
  int a128[NIT];
  unsigned gofs[NIT];            // byte offset of this thread's unit i inside an output row of the strip
#pragma unroll
  for (int i = 0; i < NIT; ++i) {
    const int px = tid / UPP + (256 / UPP) * i, u = tid % UPP;
    a128[i] = px * PB + 16 * (u ^ tsw(px));
    gofs[i] = (unsigned)((ox0 + px) * PB + 16 * u);
  }
  const size_t img = (size_t)n * H * W * PB;
  unsigned char* outn = reinterpret_cast<unsigned char*>(p.out) + img;
  const unsigned char* resn = reinterpret_cast<const unsigned char*>(HAS_RES ? p.res : p.out) + img;
  const uint4 z4 = {0u, 0u, 0u, 0u};
  uint4 rres0 = z4, rres1 = z4, rres2 = z4, rres3 = z4;   // (named register sets: an array here stayed in scratch memory)
  auto crow = [&](int oy) __attribute__((always_inline)) { return oy < oy0 ? oy0 : (oy >= oy1 ? oy1 - 1 : oy); };
  auto load_res = [&](int oy) __attribute__((always_inline)) {   // residual row oy (clamped into the segment), 16 bytes per unit
    if (HAS_RES && !(STRIP_DIAG & 1)) {
      const unsigned r0 = (unsigned)(crow(oy) * W) * PB;
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
      typedef __bf16 b2t __attribute__((ext_vector_type(2)));
      const b2t lo = {(__bf16)v[0], (__bf16)v[1]}, hi = {(__bf16)v[2], (__bf16)v[3]};   // one v_cvt_pk_bf16_f32 each
      *reinterpret_cast<uint2*>(dst) = uint2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    } else {
      *reinterpret_cast<s_f32x4*>(dst) = v;
    }
    s1 += v;
    s2 += v * v;
  };
  auto flush = [&](int oy, int i) __attribute__((always_inline)) {   // unit i of row oy's tile to memory (after the barrier that followed its epilogue)
    const uint4 v = *reinterpret_cast<const uint4*>(otile + (oy & 1) * TILE + a128[i]);
    if (!(STRIP_DIAG & 2)) *reinterpret_cast<uint4*>(outn + ((unsigned)(oy * W) * PB + gofs[i])) = v;
  };

  // ---- one step: input row iy (staged in slot `cur`) into the three accumulator sets; row iy + 1 staged into `nxt`; row iy + 4
  // fetched; output row iy - 2 (finished by the step before) written.  ROT = step % 3 names the accumulator sets and the raw
  // register sets statically: ky = 0 -> (ROT + 1) % 3 (a fresh row), ky = 1 -> ROT, ky = 2 -> (ROT + 2) % 3.
  // The step is NF slots of [next fragment read | three MFMAs (ky = 2, 1, 0 on one fragment) | ONE item of vector work], fenced: the
  // 4 + 4 + 1 activation slices of the next row and the four epilogue quads of the row finished last step are spread over the
  // slots, so that neither wave of a SIMD runs a long MFMA-free stretch.  EPI: the step writes a row (steps >= 3 of a segment);
  // RESLD: it fetches the residual of the row it finishes (steps >= 2).
  static_assert(NPASS == 2 && NF == 24, "the slot schedule below is written for two staging passes and 24 fragments per row");
  uint4 Xf[XS][NP];
  auto mfma1 = [&](const uint4* wf, const uint4* xf, s_f32x4 c) __attribute__((always_inline)) -> s_f32x4 {
    if (PREC == PREC_F16X3) {   // small terms first: lo(x) hi(w), hi(x) lo(w), hi(x) hi(w)
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s_h8, wf[0]), __builtin_bit_cast(s_h8, xf[NP - 1]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s_h8, wf[NP - 1]), __builtin_bit_cast(s_h8, xf[0]), c, 0, 0, 0);
      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(s_h8, wf[0]), __builtin_bit_cast(s_h8, xf[0]), c, 0, 0, 0);
    }
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s_b8, wf[0]), __builtin_bit_cast(s_b8, xf[0]), c, 0, 0, 0);
  };
  auto step = [&](int iy, int it, auto rot_tag, auto epi_tag, auto flush_tag) __attribute__((always_inline)) {
    constexpr int ROT = decltype(rot_tag)::value;
    constexpr bool EPI = decltype(epi_tag)::value, FLUSH = decltype(flush_tag)::value;
    constexpr int A0 = (ROT + 1) % 3, A1 = ROT, A2 = (ROT + 2) % 3;
    StripRaw<PREC>* rset = raw[(ROT + 1) % 3];             // holds row iy + 1; re-filled with row iy + 4
    StripRawPair<PREC>& hset = rawh[(ROT + 1) % 3];
    unsigned char* cur = smem_s + (it & 1) * Cfg::SLOT_BYTES;
    unsigned char* nxt = smem_s + ((it & 1) ^ 1) * Cfg::SLOT_BYTES;
    const bool rok = iy + 1 >= 0 && iy + 1 < H;            // the row being staged lies inside the image
    auto load_x = [&](int f) __attribute__((always_inline)) {   // fragment f = (kx, kc, ph), ph fastest
      const int ph = f % NPH, kc = (f / NPH) % KCH, kx = f / (NPH * KCH);
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        Xf[f % XS][pl] = *reinterpret_cast<const uint4*>(cur + xa[kx][kc][pl] + ph * 16 * RB);
    };
#pragma unroll
    for (int f = 0; f < XS - 1; ++f) load_x(f);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int ph = f % NPH, kc = (f / NPH) % KCH, kx = f / (NPH * KCH);
      if (f + XS - 1 < NF) load_x(f + XS - 1);
      const bool fresh = kx == 0 && kc == 0;               // the first product of a new output row starts from zero
      const s_f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      // slots 0 .. NPH - 1: quad ph of the row the previous step finished leaves its accumulator (set A0: ky = 2 of the step
      // before) right before this slot's fresh product overwrites it
      if (EPI && f < NPH) finish(iy - 2, f, acc[A0][f]);
      acc[A2][ph] = mfma1(Wf[2][kx][kc], Xf[f % XS], acc[A2][ph]);
      acc[A1][ph] = mfma1(Wf[1][kx][kc], Xf[f % XS], acc[A1][ph]);
      acc[A0][ph] = mfma1(Wf[0][kx][kc], Xf[f % XS], fresh ? zero : acc[A0][ph]);
      // ---- this slot's item of vector work ----
      if (f >= 4 && f <= 10 && (f & 1) == 0) act_slice(rset[0], stg[0], (f - 4) >> 1);   // slots 4, 6, 8, 10
      if (f == 11) {
        write_oct(stg[0], rok, nxt, pdst[0]);
        load_pass(iy + 4, rset, 0);
      }
      if (f >= 12 && f <= 18 && (f & 1) == 0) act_slice(rset[1], stg[1], (f - 12) >> 1);   // slots 12, 14, 16, 18
      if (f == 19) {
        write_oct(stg[1], rok, nxt, pdst[1]);
        load_pass(iy + 4, rset, 1);
      }
      if (f == 20) {
        stage_halo(hset, hok && rok, nxt);
        load_halo(iy + 4, hset);
      }
      if (FLUSH && f == 5) flush(iy - 3, 0);               // the row whose epilogue ran a step ago
      if (FLUSH && f == 7 && NIT > 1) flush(iy - 3, NIT > 1 ? 1 : 0);
      if (FLUSH && f == 9 && NIT > 2) flush(iy - 3, NIT > 2 ? 2 : 0);
      if (FLUSH && f == 13 && NIT > 3) flush(iy - 3, NIT > 3 ? 3 : 0);
      if (f == 22) {          // the residual of the row this step finishes goes into its tile; the next row's is fetched
        res_to_lds(iy - 1);
        load_res(iy);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> R0;
  typedef std::integral_constant<int, 1> R1;
  typedef std::integral_constant<int, 2> R2;

  // ---- the strip segment: input rows oy0 - 1 .. oy1 ----
  {
    // rows oy0 - 1 (staged here), oy0, oy0 + 1, oy0 + 2 in flight before the first step; the set of a row = its step % 3
    StripRaw<PREC> first[NPASS];
    StripRawPair<PREC> firsth;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) load_pass(oy0 - 1, first, i);
    load_halo(oy0 - 1, firsth);
#pragma unroll
    for (int r = 1; r <= 3; ++r) {
#pragma unroll
      for (int i = 0; i < NPASS; ++i) load_pass(oy0 - 1 + r, raw[r % 3], i);
      load_halo(oy0 - 1 + r, rawh[r % 3]);
    }
    const bool rok = oy0 - 1 >= 0;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
#pragma unroll
      for (int k = 0; k < 4; ++k) act_slice(first[i], stg[i], k);
      write_oct(stg[i], rok, smem_s, pdst[i]);
    }
    stage_halo(firsth, hok && rok, smem_s);
  }
  __syncthreads();
  const int nsteps = oy1 - oy0 + 2;          // >= 5 (segments have at least 3 rows: launcher)
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
// g_tun.strip bits: 1 bf16 launches (two workgroups per CU), 2 f16x3 launches (one per CU: hi / lo weight planes = 144 registers)
static long strip_min_wgs(int prec) { return prec == PREC_BF16 ? g_tun.strip_min_wgs : g_tun.strip_min_wgs / 2; }

static int strip_seg_rows(const ConvParams& p, int SW, int prec) {
  // strips x segments: enough workgroups to fill the chip (bf16: two per CU), segments as long as that allows (each costs two extra steps)
  const long strips = (long)p.N * ((p.Wout + SW - 1) / SW);
  int rows = p.Hout;
  while (rows > 16 && strips * ((p.Hout + rows - 1) / rows) < strip_min_wgs(prec)) rows = (rows + 1) / 2;
  return rows;
}

bool conv_strip_ok(ConvKind kind, int prec, const ConvParams& p) {
  if (!(g_tun.strip & (prec == PREC_BF16 ? 1 : 2))) return false;
  if (kind != CONV3_S1 || p.ksplit > 1 || p.Cout != 64 || p.Cout_pad != 64) return false;
  if (p.xr0 || !p.gn_scale || p.gn_plain || p.drop_mask) return false;
  if (p.Hin != p.Hout || p.Win != p.Wout || p.Wout % 64 || p.Hout < 3 || (size_t)p.Hout * p.Wout * 64 * 4 >= (1ull << 31)) return false;
  if (prec == PREC_BF16 && p.out_f32) return false;
  const int Cin = p.C0 + p.C1;
  if (Cin != p.Cin_pad || Cin != 64) return false;
  if (p.C0 % 8 || p.C1 % 8) return false;
  const long wgs = (long)p.N * ((p.Wout + 63) / 64) * ((p.Hout + 15) / 16);
  return wgs >= strip_min_wgs(prec);   // (a small grid keeps the split-K tile kernels)
}

template <int PREC, int KCH, int NPH, int LB>
static hipError_t launch_strip_t(const ConvParams& p, int wn_a, hipStream_t s, int* tiles) {
  using Cfg = StripCfg<PREC, KCH, NPH>;
  const int rows = strip_seg_rows(p, Cfg::SW, PREC);
  const int stripsX = (p.Wout + Cfg::SW - 1) / Cfg::SW, segs = (p.Hout + rows - 1) / rows;
  if (tiles) *tiles = stripsX * segs;
  const dim3 grid(p.N * stripsX * segs);
  const size_t lds = (size_t)Cfg::LDS_BYTES;
  if (p.res) hipLaunchKernelGGL((conv_strip_kernel<PREC, KCH, NPH, true, LB>), grid, dim3(256), lds, s, p, rows, wn_a);
  else hipLaunchKernelGGL((conv_strip_kernel<PREC, KCH, NPH, false, LB>), grid, dim3(256), lds, s, p, rows, wn_a);
  return hipGetLastError();
}

hipError_t launch_conv_strip(int prec, const ConvParams& p, int wn_a, hipStream_t s, int* tiles) {
  if (prec == PREC_BF16) return launch_strip_t<PREC_BF16, 2, 4, 2>(p, wn_a, s, tiles);
  if (prec == PREC_F16X3) return launch_strip_t<PREC_F16X3, 2, 4, 1>(p, wn_a, s, tiles);
  return hipErrorInvalidValue;
}

template <int PREC, int KCH, int NPH, int LB>
static hipError_t init_strip_t() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_strip_kernel<PREC, KCH, NPH, false, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_strip_kernel<PREC, KCH, NPH, true, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t kernels_strip_init() {
  hipError_t e = init_strip_t<PREC_BF16, 2, 4, 2>();
  if (e != hipSuccess) return e;
  return init_strip_t<PREC_F16X3, 2, 4, 1>();
}

}  // namespace fdsr
