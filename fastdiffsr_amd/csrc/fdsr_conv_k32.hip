// K=32 form of the stride-1 3x3 convolutions for gfx950: v_mfma_f32_16x16x32_{f16,bf16}, fp32 accumulate.
//
// Same arithmetic modes, same ConvParams, same packed weights and the same outputs layout as conv_mfma_h_kernel
// (fdsr_conv_h.hip): this kernel is a drop-in for its <3, 1, false, TH, WN, PREC, 1, RIDER> instantiations whose wave
// tile is four rows of 32 pixels (MB = 4) when the input splits into whole 32-channel chunks.  What changes is the matrix
// instruction: 16x16x32 moves half the accumulator bytes per FLOP through the register file (the
// sampling loop runs at the package power cap, so throughput follows energy per image; the guide measures 1.12-1.15x for
// this shape in power-limited loops).
//
//   * K chunk = 32 input channels x one tap.  The MFMA's "A" operand is the WEIGHT fragment (rows = 16 output channels),
//     its "B" operand the ACTIVATION fragment (columns = 16 pixels), so a lane's four accumulator registers are four
//     consecutive output channels of one pixel: 16-byte stores / residual loads in the NHWC epilogue.
//   * Weights are read from the arena pack_weights_h() already fills ([cot][kc16][wn][tap][plane][lane] x 16 B, the
//     32x32x16 B-operand order): the 16x16x32 fragment of (32-channel chunk, tap, cout half ch, plane) is a permutation of
//     16-byte units of two of those blocks -- lane (g = l >> 4, c = l & 15) takes unit 32 (g & 1) + 16 ch + c of 16-channel
//     block 2 kc + (g >> 1).  No second weight form, so optimiser re-packs and the transposed (input-gradient) forms serve
//     this kernel as they are.
//   * Only a ring of three taps of weight fragments is live (48 VGPRs in f16x3 instead of 144; 24 in bf16): tap t + 3 is
//     fetched from L2 right after tap t's last MFMA, while taps t + 1, t + 2 are multiplied (>= 3000 cycles of MFMAs in f16x3).
//   * Halo image in LDS: 128-byte pixel rows [4 x 16 B hi | 4 x 16 B lo] (bf16: 64 bytes), no padding -- the 18 x 34 halo
//     of a 16-row tile, double-buffered, is 153 KB of the CU's 160 -- with the 16-byte slot XOR-swizzled by the pixel
//     column (f16x3: slot ^ (hx & 7); bf16: slot ^ ((hx >> 1) & 3)): every ds_read_b128 of an activation fragment is
//     bank-conflict free on the instruction's four 16-lane groups for all three kx shifts (tests/test_k32_maps.py replays
//     the maps lane by lane in numpy).
//   * Activation fragments are read one (tap, row) step ahead of their MFMAs and the reads are pinned there (sched_barrier) in
//     the 8- / 4-row tiles; f16x3 rider-less kernels fetch and stage the next chunk in two halves and run their last chunk
//     as a code copy that already fetches the residual tile.  The K32_* / XS_* macros below are the A/B switches of those
//     choices (tools/k32_variant.sh builds a variant library; every measured pair is in profiles/r03_k32_ab.txt).
#include "fdsr_kernels.h"
#include "fdsr_act_io.h"

#include <type_traits>

namespace fdsr {

typedef float k_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 k_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 k_h4 __attribute__((ext_vector_type(4)));
typedef __bf16 k_b8 __attribute__((ext_vector_type(8)));
typedef __bf16 k_b4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float silu_k(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

#ifndef XS_F16X3
#define XS_F16X3 2
#endif
#ifndef XS_BF16
#define XS_BF16 4
#endif
#ifndef K32_PIN      // pin the fragment reads ahead of the MFMAs (sched_barrier) in the 8- and 4-row tiles
#define K32_PIN 1
#endif
#ifndef K32_RIDER2   // rider chunks fetch their input two chunks ahead (two register sets) instead of one
#define K32_RIDER2 1
#endif
#ifndef K32_SPLIT     // f16x3: the next chunk's input is fetched and staged in two halves (taps 0-3, 3-7): half the prefetch registers
#define K32_SPLIT 1
#endif
#ifndef K32_SPLIT_BF16   // ... in bf16 mode too
#define K32_SPLIT_BF16 1
#endif
#ifndef K32_MIXSPLIT  // hi/lo split of the staging as v_cvt_pk_f16_f32 + v_fma_mix (fewer VALU instructions, same bits)
#define K32_MIXSPLIT 1
#endif
#ifndef K32_PEEL      // rider-less kernels: the last chunk is a code copy of its own that also fetches the residual tile
#define K32_PEEL 1
#endif
#ifndef K32_GNB_DIAG    // timing probes of the GroupNorm-backward epilogue (results are garbage): 1 no swish' arithmetic, 2 no x / mask loads
#define K32_GNB_DIAG 0
#endif
#ifndef K32_DIAG        // timing-only probes of the staging (results are garbage; tools/pmc_variants.sh gives the clock beside the time -- a probe that
#define K32_DIAG 0      // changes the operands changes the clock the chip holds): 1 no GroupNorm / Swish arithmetic, 2 nothing staged inside the K
#endif                  // loop, 4 no statistics in the epilogue, 8 everything but the LDS store, 16 the staged values do not depend on the fetches,
                        // 32 every chunk reads chunk 0's weight fragments
#ifndef K32_RFIRST   // small-workgroup rider kernels: rider chunks first
#define K32_RFIRST 1
#endif
#ifndef K32_PIN16    // ... and in the 16-row tile (its ten staging quads leave no registers for it: spills)
#define K32_PIN16 0
#endif

// NW_ = 8: one 512-thread workgroup per CU (two waves per SIMD in ONE barrier domain).
// NW_ = 4 ("small workgroups", round 4): 256-thread workgroups, TWO per CU -- the same two waves per SIMD, but in two barrier
// domains, so that one workgroup's prologue / epilogue can run beside the other's main loop.  Built for the 64-channel level, whose
// tiles have 2 - 6 chunks: there prologue + epilogue of a one-workgroup-per-CU tile are ~40 % of its time.  Two
// double-buffered halo images must fit 80 KB: 6-row tiles in f16x3 (2 x 8 x 34 pixels x 128 B = 69.6 KB), 8-row tiles in bf16 (43.5 KB;
// three activation-fragment slots instead of four keep that form clear of spills: 256 VGPRs).
// LDS behind the halo buffers of a GNC launch: pair sums (double2) of the groups the K slice touches, then the scale and shift tables
constexpr int K32_GNC_MAXC = 1024, K32_GNC_MAXP = K32_GNC_MAXC / 2 + 32, K32_GNC_LDS = K32_GNC_MAXP * 32 + K32_GNC_MAXC * 8;

template <int TH, int WN, int PREC, int NW_ = 8>
struct ConvK32Cfg {
  static constexpr int TW = 32, NW = NW_, KC = 32;
  static constexpr int NP = PREC == PREC_F16X3 ? 2 : 1;
  static constexpr int WNP = PREC == PREC_BF16 ? 1 : 2;   // planes per fragment in the weight arena the mode reads (PREC_F16: the f16x3 form, hi plane only)
  static constexpr int ROWB = 64 * NP;           // LDS bytes per halo pixel (no pad: swizzled)
  static constexpr int HH = TH + 2, HWD = TW + 2, NPIX = HH * HWD;
  static constexpr int WM = NW / WN, BN = 32 * WN, MB = TH / WM;
  static constexpr int NT = 64 * NW;
  static constexpr int Q4 = KC / 4;              // float4 slots per halo pixel
  static constexpr int RPP = NT / Q4;            // halo pixels filled per pass
  static constexpr int NIN = (NPIX + RPP - 1) / RPP;
  static constexpr int BUF_BYTES = NPIX * ROWB;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES;
  static constexpr int R = 3;                     // taps of weight fragments in registers (a ring: 9 % R == 0)
  static constexpr int XS = PREC == PREC_F16X3 ? XS_F16X3 : (NW_ == 4 && TH == 8 ? 3 : XS_BF16);   // activation-fragment slots: XS - 1 (tap, row) steps ahead (the bf16 8-row small-workgroup tile spills 24 VGPRs with four)
  static_assert(MB == 4 || MB == 3 || MB == 2, "wave tile = 4 (small grids: 2; small workgroups in f16x3: 3) rows of 32 pixels");
  static_assert(((TH - WM + 2) * HWD + 16) * ROWB < 65536, "fragment offsets must fit the ds_read immediate");
};

// 16-byte slot of (k group | plane) within a pixel row, swizzled by the halo column
template <int PREC>
__device__ __forceinline__ int k32_slot(int slot, int hx) {
  return PREC == PREC_F16X3 ? (slot ^ (hx & 7)) : (slot ^ ((hx >> 1) & 3));
}

// GNB: the GroupNorm-backward epilogue of the training step's input-gradient launches (ConvParams::gb_*)
// DROP: train-mode Dropout between the Swish and the convolution applied in the staging (ConvParams::drop_mask; f16x3 training forwards)
// GNC: GroupNorm scale / shift formed HERE, in the prologue, from the producers' fixed-point pair sums (ConvParams::gs0; small grids)
template <int TH, int WN, int PREC, bool RIDER, int NW = 8, bool RF = (NW == 4), bool GNB = false, bool DROP = false, bool GNC = false>   // RF: rider chunks first (launches without a K split)
__global__ void __launch_bounds__(64 * NW, 2) conv_k32_kernel(const ConvParams p) {
  using Cfg = ConvK32Cfg<TH, WN, PREC, NW>;
  constexpr int TW = Cfg::TW, KC = Cfg::KC, NP = Cfg::NP, ROWB = Cfg::ROWB, HWD = Cfg::HWD, NPIX = Cfg::NPIX;
  constexpr int WM = Cfg::WM, BN = Cfg::BN, MB = Cfg::MB, RPP = Cfg::RPP, NIN = Cfg::NIN, R = Cfg::R, XS = Cfg::XS;
  constexpr bool PIN = TH == 16 ? K32_PIN16 != 0 : K32_PIN != 0;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_k[];
  unsigned char* sBuf0 = smem_k;
  unsigned char* sBuf1 = smem_k + Cfg::BUF_BYTES;
  const int Cin = p.C0 + p.C1;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int c15 = lane & 15, g = lane >> 4;

  const int nco = p.Cout_pad / BN;
  const int tilesX = (p.Wout + TW - 1) / TW, tilesY = (p.Hout + TH - 1) / TH;
  int bid;
  {   // consecutive tiles on one XCD (as conv_mfma_h_kernel)
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7, k = b >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int SK = p.ksplit > 1 ? p.ksplit : 1;
  const int ksi = bid % SK;
  bid /= SK;
  const int cot = bid % nco;
  int pt = bid / nco;
  const int tx = pt % tilesX;
  pt /= tilesX;
  const int ty = pt % tilesY;
  const int n = pt / tilesY;
  const int oy0 = ty * TH, ox0 = tx * TW, co0 = cot * BN;

  const bool gn = GNC ? true : p.gn_scale != nullptr;

  if (NW == 4 && p.stagger > 0) {
    // Two of these workgroups share a CU.  Dispatched together and running the same program they would stay in phase: both in
    // their prologue, both in their MFMAs, both in their epilogue.  The one in an odd workgroup slot of its CU (HW_ID.TG_ID) starts
    // late by a share of a tile's time, so that its prologue / epilogue fall beside the other's main loop.  Timing only: nothing
    // is computed from it.
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    if ((hwid >> 16) & 1u) {
      const int n64 = p.stagger * (p.Cin_pad / KC + (RIDER ? p.nkr / 2 : 0));
      for (int i = 0; i < n64; i += 8) __builtin_amdgcn_s_sleep(8);     // s_sleep 8 = 8 x 64 cycles
    }
  }

  // ---- staging indices (chunk invariant): thread -> (halo pixel row0 + i*64, float4 slot q of the 32 channels) ----
  const int q = tid & 7, row0 = tid >> 3;
  int in_pix[NIN];
  unsigned long long skeys = 0;   // 3 bits per pass: the swizzle key of this thread's halo column (the shifts are compile-time)
  static_assert(NIN <= 21, "swizzle keys are packed into one register pair");
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int pix = row0 + i * RPP;
    int v = -2;
    if (pix < NPIX) {
      const int hy = pix / HWD, hx = pix % HWD;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
      v = ok ? (n * p.Hin + iy) * p.Win + ix : -1;
      skeys |= (unsigned long long)k32_slot<PREC>(0, hx) << (3 * i);
    }
    in_pix[i] = v;
  }
  const int sbase = row0 * ROWB + 8 * (q & 1);   // + i * RPP * ROWB + 16 * ((q >> 1) ^ key_i)

  using IO = ActIO<PREC>;
  typedef typename IO::Quad Quad;
  struct QM { Quad q; unsigned m; };   // a staged quad and (DROP) the four dropout bytes of its channels; `m` is never materialised otherwise
  // RFIRST (the small-workgroup rider kernels): the rider's chunks run BEFORE the main chunks, so that their whole-chunk prefetch set is
  // dead when the main loop starts and that loop is the rider-less one (two-half staging, peeled last chunk) -- the round-3 plan.
  constexpr bool RFIRST = RIDER && RF && K32_RFIRST != 0;
  constexpr bool SPLIT = K32_SPLIT != 0 && (PREC == PREC_F16X3 || K32_SPLIT_BF16 != 0) && (!RIDER || RFIRST);   // (main chunks first: the rider's whole-chunk sets would stay live across the main loop)
  static_assert(!RFIRST || SPLIT, "the rider-first path writes whole chunks into rr1, which is sized for them only when SPLIT");
  constexpr int NA = SPLIT ? (NIN + 1) / 2 : NIN;   // quads in flight in the main loop
  typedef std::integral_constant<int, 0> I_0;
  typedef std::integral_constant<int, NA> I_A;
  typedef std::integral_constant<int, NIN> I_N;
  QM rin[NA];
  QM rr1[RIDER && SPLIT ? NIN : 1];          // rider chunks are fetched whole: first set (rin itself when not SPLIT)
  QM rin2[RIDER && K32_RIDER2 && NW == 8 ? NIN : 1];   // (small workgroups: one set -- the CU's other workgroup covers the latency)   // second prefetch set of the rider chunks
  k_f32x4 rsc = {1.f, 1.f, 1.f, 1.f}, rsh = {0.f, 0.f, 0.f, 0.f};
  const int nk = p.Cin_pad / KC;               // main chunks; nk .. nk + nkr - 1 are the rider's (raw second input, centre tap)
  const int nk16 = p.Cin_pad / 16;
  const int nkr = RIDER ? p.nkr / 2 : 0;
  // ---- GNC: the scale / shift table of this workgroup's main chunks [gnc_lo, gnc_hi) in LDS behind the halo buffers ----
  double* gnc_pr = reinterpret_cast<double*>(smem_k + Cfg::LDS_BYTES);                  // [pairs][shard half][2]
  float* gnc_sc = reinterpret_cast<float*>(smem_k + Cfg::LDS_BYTES + K32_GNC_MAXP * 32);   // [channels]
  float* gnc_sh = gnc_sc + K32_GNC_MAXC;
  int gnc_lo = 0, gnc_hi = 0, gnc_cpg = 1, gnc_q0 = 0, gnc_q1 = 0;
  constexpr int GNC_PT = (2 * K32_GNC_MAXP + Cfg::NT - 1) / Cfg::NT;   // units per thread: a unit = one pair x four of its eight shards
  uint4 gnc_ld[GNC ? GNC_PT : 1][GSUM_SHARDS / 2];
  constexpr int GNC_CT = (K32_GNC_MAXC + Cfg::NT - 1) / Cfg::NT;   // channels per thread
  float gnc_gam[GNC ? GNC_CT : 1], gnc_bet[GNC ? GNC_CT : 1];      // (fetched with the table: one round trip, not two)
  auto gnc_issue = [&](int c_lo, int c_hi) __attribute__((always_inline)) {   // every load of the table in flight (beside the weights and the first chunk)
    gnc_lo = c_lo; gnc_hi = c_hi;
    gnc_cpg = Cin / p.gs_G;
    gnc_q0 = (c_lo / gnc_cpg) * gnc_cpg / 2;                       // pairs of every group the range touches (groups may cross chunk
    gnc_q1 = ((c_hi - 1) / gnc_cpg + 1) * gnc_cpg / 2;             // boundaries and the concat seam: 384 channels = 32 groups of 12)
#pragma unroll
    for (int j = 0; j < GNC_PT; ++j) {
      const int u = tid + j * Cfg::NT, qc = min(gnc_q0 + (u >> 1), gnc_q1 - 1);  // (clamped: unconditional loads)
      const bool first = qc < (p.C0 >> 1);
      const unsigned long long* tab = first ? p.gs0 : p.gs1;
      const int np = (first ? p.C0 : p.C1) >> 1, ql = first ? qc : qc - (p.C0 >> 1);
#pragma unroll
      for (int sdx = 0; sdx < GSUM_SHARDS / 2; ++sdx)
        gnc_ld[j][sdx] = *reinterpret_cast<const uint4*>(tab + (((size_t)n * GSUM_SHARDS + (u & 1) * (GSUM_SHARDS / 2) + sdx) * np + ql) * 2);
    }
#pragma unroll
    for (int j = 0; j < GNC_CT; ++j) {
      const int cc = min(c_lo + tid + j * Cfg::NT, c_hi - 1);
      gnc_gam[j] = p.gs_gamma[cc];
      gnc_bet[j] = p.gs_beta[cc];
    }
  };
  auto gnc_finish = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < GNC_PT; ++j) {
      const int u = tid + j * Cfg::NT, qi = gnc_q0 + (u >> 1);
      long long a = 0, b = 0;
#pragma unroll
      for (int sdx = 0; sdx < GSUM_SHARDS / 2; ++sdx) {
        a += (long long)(((unsigned long long)gnc_ld[j][sdx].y << 32) | gnc_ld[j][sdx].x);
        b += (long long)(((unsigned long long)gnc_ld[j][sdx].w << 32) | gnc_ld[j][sdx].z);
      }
      if (qi < gnc_q1) {   // (integer sums of four shards: exact; the two halves of a pair meet as doubles below)
        gnc_pr[2 * u] = (double)a * (1.0 / (double)(1 << GSUM_BITS1));
        gnc_pr[2 * u + 1] = (double)b * (1.0 / (double)(1 << GSUM_BITS2));
      }
    }
    __syncthreads();
    const double inv = 1.0 / ((double)gnc_cpg * (double)p.Hin * (double)p.Win);
#pragma unroll
    for (int j = 0; j < GNC_CT; ++j) {   // (as gn_finalize_kernel: fp64 statistics, fp32 scale / shift)
      const int c = gnc_lo + tid + j * Cfg::NT;
      if (c >= gnc_hi) break;
      const int g = c / gnc_cpg, hp = gnc_cpg >> 1;
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < 2 * hp; ++k) { s1 += gnc_pr[2 * (2 * (g * hp - gnc_q0) + k)]; s2 += gnc_pr[2 * (2 * (g * hp - gnc_q0) + k) + 1]; }   // (pair, shard half) units
      const double mean = s1 * inv;
      double var = s2 * inv - mean * mean;
      var = var < 0.0 ? 0.0 : var;
      const double rstd = 1.0 / sqrt(var + (double)p.gs_eps);
      const float scv = (float)rstd * gnc_gam[j];
      gnc_sc[c - gnc_lo] = scv;
      gnc_sh[c - gnc_lo] = gnc_bet[j] - (float)mean * scv;
    }
    __syncthreads();
  };
  auto prefetch_rng = [&](int kc, QM* rin, auto i0_tag, auto i1_tag, bool load_gn) __attribute__((always_inline)) {
    constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value;
    int cbase = kc * KC;
    const float* base;
    int Cs, cc;
    bool gg = gn;
    if (RIDER && kc >= nk) {
      cbase -= nk * KC;
      gg = false;
      if (cbase < p.Cr0) { base = p.xr0; Cs = p.Cr0; cc = cbase + q * 4; }
      else { base = p.xr1; Cs = p.Cr1; cc = cbase - p.Cr0 + q * 4; }
    } else if (cbase < p.C0) { base = p.x0; Cs = p.C0; cc = cbase + q * 4; }
    else { base = p.x1; Cs = p.C1; cc = cbase - p.C0 + q * 4; }
    if (!GNC && gg && load_gn) {
      rsc = *reinterpret_cast<const k_f32x4*>(p.gn_scale + (size_t)n * Cin + cbase + q * 4);
      rsh = *reinterpret_cast<const k_f32x4*>(p.gn_shift + (size_t)n * Cin + cbase + q * 4);
    }
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const size_t pi = (size_t)(in_pix[i] < 0 ? 0 : in_pix[i]);
      rin[i - I0].q = IO::load4(base, pi * Cs + cc);
      if (DROP)   // the dropout bytes of these four channels ([N][H][W][C0], single input); a rider chunk reads element 0 (unused): a select, not a branch
        rin[i - I0].m = *reinterpret_cast<const unsigned*>(p.drop_mask + ((RIDER && kc >= nk) ? (size_t)0 : pi * Cs + cc));
    }
  };
  auto prefetch_to = [&](int kc, QM* rin) __attribute__((always_inline)) { prefetch_rng(kc, rin, I_0{}, I_N{}, true); };
  auto stage_rng = [&](int kc, unsigned char* buf, const QM* rin, auto i0_tag, auto i1_tag) __attribute__((always_inline)) {
    constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value;
    k_f32x4 sc = rsc, sh = rsh;
    if (GNC && !(RIDER && kc >= nk)) {   // this chunk's four channels of the LDS table the prologue built
      sc = *reinterpret_cast<const k_f32x4*>(gnc_sc + (kc * KC - gnc_lo + q * 4));
      sh = *reinterpret_cast<const k_f32x4*>(gnc_sh + (kc * KC - gnc_lo + q * 4));
    }
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      if (NIN * RPP > NPIX && i == NIN - 1 && row0 + i * RPP >= NPIX) continue;   // only the last pass can overrun
      k_f32x4 v = IO::widen(rin[i - I0].q);
      if (K32_DIAG & 16) v = sc;
      if ((K32_DIAG & 1) && gn) {
        v = v + sh;
      } else if (gn && !(RIDER && kc >= nk)) {
        v = v * sc + sh;
        if (!p.gn_plain) { v.x = silu_k(v.x); v.y = silu_k(v.y); v.z = silu_k(v.z); v.w = silu_k(v.w); }
        if (DROP) {   // train-mode Dropout(p) behind the Swish: keep bytes are 0 / 1
          const unsigned m = rin[i - I0].m;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= (float)((m >> (8 * e)) & 0xffu) * p.drop_scale;
        }
      } else if (PREC == PREC_F16X3 && p.sat_flag) {
        sat_check(p.sat_flag, v, 65504.f);
      }
      const int sd = sbase + i * (RPP * ROWB) + 16 * ((q >> 1) ^ (int)((skeys >> (3 * i)) & 7u));
      unsigned char* dst = buf + sd;
      if (PREC == PREC_F16X3) {
        const float lim = in_pix[i] >= 0 ? 65504.f : 0.f;   // f16 range clamp and zero padding in one med3
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -lim, lim);
#if K32_MIXSPLIT
        // one packed convert and two v_fma_mix per pair: lo = f16(fma(hi, -1, v)), rounded once -- bit-identical to the two-step
        // form below (as in fdsr_train.hip).  With 16x16x32 MFMAs holding the issue port half of their time
        // the staging VALU count matters more than it did beside 32x32x16.
        typedef _Float16 h2t __attribute__((ext_vector_type(2)));
        uint2 hi, lo;
        {
          const h2t h0 = {(_Float16)v[0], (_Float16)v[1]}, h1 = {(_Float16)v[2], (_Float16)v[3]};
          hi.x = __builtin_bit_cast(unsigned, h0);
          hi.y = __builtin_bit_cast(unsigned, h1);
        }
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo.x) : "v"(hi.x), "v"(v[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo.x) : "v"(hi.x), "v"(v[1]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo.y) : "v"(hi.y), "v"(v[2]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo.y) : "v"(hi.y), "v"(v[3]));
        *reinterpret_cast<uint2*>(dst) = hi;
        *reinterpret_cast<uint2*>(buf + (sd ^ 64)) = lo;   // slot + 4 of the swizzled row
#else
        k_h4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        k_h4 lo = {(_Float16)(v.x - (float)hi.x), (_Float16)(v.y - (float)hi.y), (_Float16)(v.z - (float)hi.z),
                   (_Float16)(v.w - (float)hi.w)};
        *reinterpret_cast<k_h4*>(dst) = hi;
        *reinterpret_cast<k_h4*>(buf + (sd ^ 64)) = lo;   // slot + 4 of the swizzled row
#endif
      } else {
        const float keep = in_pix[i] >= 0 ? 1.f : 0.f;
        v = v * keep;
        const uint2 hb = stage4_16<PREC>(v);
        if (K32_DIAG & 8) asm volatile("" ::"v"(hb.x), "v"(hb.y));
        else *reinterpret_cast<uint2*>(dst) = hb;
      }
    }
  };
  auto stage_from = [&](int kc, unsigned char* buf, const QM* rin) __attribute__((always_inline)) { stage_rng(kc, buf, rin, I_0{}, I_N{}); };

  // ---- weight fragments: ring slot s holds tap t with t % R == s; [cout half][plane] ----
  const uint4* wq = reinterpret_cast<const uint4*>(p.wq);
  const int nkt = RIDER ? nk + nkr : nk;
  uint4 Wf[R][2][NP];
  const int wlane = 32 * (g & 1) + c15;          // + 16 ch: unit within the 32x32x16-order block
  auto load_w = [&](int kc, int tap, int slot) __attribute__((always_inline)) {
    if (K32_DIAG & 32) kc = 0;   // (timing probe: every chunk multiplies chunk 0's fragments -- what a weight fetch that never misses would buy)
    const uint4* src = wq + ((((size_t)cot * nk16 + 2 * kc + (g >> 1)) * WN + wn) * 9 + tap) * (Cfg::WNP * 64) + wlane;
    if (RIDER && kc >= nk)   // the 1x1 conv's own fragments [cot][kc16][wn]
      src = reinterpret_cast<const uint4*>(p.wq_r) + (((size_t)cot * p.nkr + 2 * (kc - nk) + (g >> 1)) * WN + wn) * (Cfg::WNP * 64) + wlane;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) Wf[slot][ch][pl] = src[pl * 64 + 16 * ch];
  };

  // ---- activation fragment addresses: lane -> pixel column c15 (+ 16 ph) of a 32-pixel row, k group g ----
  int xoff[3][NP];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int hx = c15 + kx;
    xoff[kx][0] = (wm * HWD + hx) * ROWB + 16 * k32_slot<PREC>(g, hx);
    if (NP == 2) xoff[kx][NP - 1] = xoff[kx][0] ^ 64;
  }

  k_f32x4 acc[MB][2][2];   // [row][pixel half][cout half]
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) acc[mb][ph][ch] = k_f32x4{0.f, 0.f, 0.f, 0.f};

  const int kc0 = ksi * nkt / SK, kc1 = (ksi + 1) * nkt / SK;   // this slice's chunks
  if (RFIRST || (RIDER && kc0 >= nk)) load_w(RFIRST ? nk : kc0, 0, 0);
  else {
#pragma unroll
    for (int t = 0; t < R; ++t) load_w(kc0, t, t);
  }
  // GNC: the table's loads go out here, beside the weights and the first chunk's input; the fold and the two barriers follow the first fetch
  const int gm0 = RFIRST ? 0 : (kc0 < nk ? kc0 : nk), gm1 = RIDER ? (kc1 < nk ? kc1 : nk) : kc1;   // this workgroup's main chunks
  const bool gnc_on = GNC && gm1 > gm0;
  if (gnc_on) gnc_issue(gm0 * KC, gm1 * KC < Cin ? gm1 * KC : Cin);
  if (RFIRST) {   // (never split in K: kc0 = 0, kc1 = nk + nkr)
    {
      QM r0[NIN];
      prefetch_to(nk, r0);
      if (gnc_on) gnc_finish();
      stage_from(nk, sBuf0, r0);
    }
    prefetch_to(nkr > 1 ? nk + 1 : 0, rr1);   // the second rider chunk -- or, after a single one, the first main chunk (fetched whole)
  } else if (SPLIT) {
    {   // the first chunk is fetched whole (the accumulators are not live yet)
      QM r0[NIN];
      prefetch_to(kc0, r0);
      if (gnc_on) gnc_finish();
      stage_from(kc0, sBuf0, r0);
    }
    if (RIDER && kc0 >= nk && kc0 + 1 < kc1) prefetch_to(kc0 + 1, rr1);   // (a slice that starts among the rider chunks)
  } else {
    prefetch_to(kc0, rin);
    if (gnc_on) gnc_finish();
    stage_from(kc0, sBuf0, rin);
    if (kc0 + 1 < kc1) prefetch_to(kc0 + 1, rin);
  }
  __syncthreads();

  uint4 Xf[XS][2][NP];   // [slot][pixel half][plane]: a ring over (tap, row) steps, XS - 1 steps ahead of the MFMAs
  const unsigned char* xptr[3][NP];
  auto load_x = [&](int slot, int ky, int kx, int mb) __attribute__((always_inline)) {
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        Xf[slot][ph][pl] = *reinterpret_cast<const uint4*>(xptr[kx][pl] + ((mb * WM + ky) * HWD + 16 * ph) * ROWB);
  };
  auto mfma_step = [&](int xs, int ws, int mb) __attribute__((always_inline)) {
    if (PREC == PREC_F16X3) {
      // small terms first: lo(x) hi(w), hi(x) lo(w), hi(x) hi(w); four independent accumulators between dependent MFMAs
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            const uint4 xv = Xf[xs][ph][term == 0 ? NP - 1 : 0];
            const uint4 wv = Wf[ws][ch][term == 1 ? NP - 1 : 0];
            acc[mb][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(k_h8, wv), __builtin_bit_cast(k_h8, xv),
                                                                    acc[mb][ph][ch], 0, 0, 0);
          }
    } else {
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
          acc[mb][ph][ch] = mfma16_k32<PREC>(Wf[ws][ch][0], Xf[xs][ph][0], acc[mb][ph][ch]);
    }
  };

  // epilogue geometry (needed early: the last chunk already fetches the residual tile)
  const int cob = co0 + wn * 32 + 4 * g;
  const bool interior = (oy0 + TH <= p.Hout) && (ox0 + TW <= p.Wout) && (co0 + BN <= p.Cout);
  const size_t obase = ((size_t)(n * p.Hout + oy0 + wm) * p.Wout + ox0 + c15) * p.Cout + cob;
  const size_t rstride = (size_t)WM * p.Wout * p.Cout, pstride = (size_t)16 * p.Cout;
  constexpr bool PEEL = K32_PEEL != 0 && (!RIDER || RFIRST);
  Quad rv[MB][2][2];   // residual tile [row][pixel half][cout half]
  const bool res_early = PEEL && p.res && interior && SK == 1;   // ... fetched during the last chunk, into registers the loop no longer needs
  auto load_res = [&](auto q0_tag, auto q1_tag) __attribute__((always_inline)) {
    constexpr int Q0 = decltype(q0_tag)::value, Q1 = decltype(q1_tag)::value;
#pragma unroll
    for (int qi = Q0; qi < Q1; ++qi) {
      const int mb = qi >> 2, ph = (qi >> 1) & 1, ch = qi & 1;
      rv[mb][ph][ch] = IO::load4(p.res, obase + mb * rstride + ph * pstride + 16 * ch);
    }
  };
  constexpr int NQ = 4 * MB;   // residual quads per lane
  typedef std::integral_constant<int, (NQ < 7 ? NQ : 7)> I_7;
  typedef std::integral_constant<int, (NQ < 11 ? NQ : 11)> I_11;
  typedef std::integral_constant<int, (NQ < 15 ? NQ : 15)> I_15;
  typedef std::integral_constant<int, NQ> I_16;

  // main chunks: all nine taps of 32 GroupNorm'ed channels
  const int kcm = RIDER ? (kc1 < nk ? kc1 : nk) : kc1;
  const int par0 = RFIRST ? nkr : 0;          // rider-first: the main chunks take the halo buffers in turn after nkr rider chunks
  auto main_chunk = [&](int kc, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;   // the peeled final chunk: nothing to fetch or stage for a next one
    unsigned char* cur = ((kc - kc0 + par0) & 1) ? sBuf1 : sBuf0;
    unsigned char* nxt = ((kc - kc0 + par0) & 1) ? sBuf0 : sBuf1;
    const bool more = LAST ? false : (kc + 1 < (RFIRST ? nk : kc1));
    const bool next_rider = RIDER && !RFIRST && kc + 1 >= nk;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) xptr[kx][pl] = cur + xoff[kx][pl];
#pragma unroll
    for (int s = 0; s < XS - 1; ++s) load_x(s, (s / MB) / 3, (s / MB) % 3, s % MB);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int s = tap * MB + mb, s2 = s + XS - 1;
        if (s2 < 9 * MB) {
          const int t2 = s2 / MB, m2 = s2 % MB;
          load_x(s2 % XS, t2 / 3, t2 % 3, m2);
        }
        if (PIN) __builtin_amdgcn_sched_barrier(0);   // keep the fragment reads AHEAD of the MFMAs (the scheduler otherwise sinks them to their use)
        mfma_step(s % XS, tap % R, mb);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
      }
      // this tap's ring slot is free: fetch the tap that will use it next
      if (tap + R < 9) load_w(kc, tap + R, tap % R);
      else if (more && (tap + R - 9 == 0 || !next_rider)) load_w(kc + 1, tap + R - 9, tap % R);
      if (SPLIT) {   // the other halo buffer is filled in two halves, each fetched four taps before it is staged
        if (tap == 0 && more) prefetch_rng(kc + 1, rin, I_0{}, I_A{}, true);
        if (tap == 3 && more) {
          if (!(K32_DIAG & 2)) stage_rng(kc + 1, nxt, rin, I_0{}, I_A{});
          prefetch_rng(kc + 1, rin, I_A{}, I_N{}, false);
        }
        if (tap == 7 && more) {
          if (!(K32_DIAG & 2)) stage_rng(kc + 1, nxt, rin, I_A{}, I_N{});
          if (next_rider && kc + 2 < kc1) prefetch_to(kc + 2, rr1);   // the rider loop expects its second chunk in flight
        }
      } else if (tap == 4 && more) {   // mid-chunk: fill the other halo buffer
        stage_from(kc + 1, nxt, rin);
        if (kc + 2 < kc1) prefetch_to(kc + 2, rin);
      }
      if (LAST && res_early) {   // the staging registers are free from the start, a weight-ring slot after each of taps 6, 7, 8
        if (tap == 0) load_res(I_0{}, I_7{});
        if (tap == 6) load_res(I_7{}, I_11{});
        if (tap == 7) load_res(I_11{}, I_15{});
        if (tap == 8) load_res(I_15{}, I_16{});
      }
    }
    __syncthreads();
  };
  if (RFIRST) {
    // rider chunks first: centre tap only, the 1x1 conv's fragments in ring slot 0, the next chunk fetched whole into rr1 one chunk
    // ahead (the CU's other workgroup covers the latency); the last one stages the first MAIN chunk and primes the weight ring
    for (int v = 0; v < nkr; ++v) {
      unsigned char* cur = (v & 1) ? sBuf1 : sBuf0;
      unsigned char* nxt = (v & 1) ? sBuf0 : sBuf1;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) xptr[1][pl] = cur + xoff[1][pl];
#pragma unroll
      for (int mb = 0; mb < XS - 1 && mb < MB; ++mb) load_x(mb, 1, 1, mb);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        if (mb + XS - 1 < MB) load_x((mb + XS - 1) % XS, 1, 1, mb + XS - 1);
        mfma_step(mb % XS, 0, mb);
      }
      if (v + 1 < nkr) {
        load_w(nk + v + 1, 0, 0);
        stage_from(nk + v + 1, nxt, rr1);
        prefetch_to(v + 2 < nkr ? nk + v + 2 : 0, rr1);
      } else {
#pragma unroll
        for (int t = 0; t < R; ++t) load_w(0, t, t);
        stage_from(0, nxt, rr1);
      }
      __syncthreads();
    }
    {   // the accumulators leave the rider's weight scale for the main conv's (a power of two: exact)
      const float ratio = (p.w_inv_scale_r_dev ? *p.w_inv_scale_r_dev : p.w_inv_scale_r) / (p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) acc[mb][ph][ch] *= ratio;
    }
  }
  if (PEEL) {
    for (int kc = kc0; kc < kcm - 1; ++kc) main_chunk(kc, std::false_type{});
    if (kcm > kc0) main_chunk(kcm - 1, std::true_type{});
  } else {
    for (int kc = kc0; kc < kcm; ++kc) main_chunk(kc, std::false_type{});
  }
  if (RIDER && !RFIRST && kc1 > nk) {
    // rider chunks: 32 raw channels of the second input each, centre tap only, the 1x1 conv's fragments in ring slot 0
    if (kc0 < nk) {   // the accumulators leave the main conv's weight scale for the rider's (a power of two: exact)
      const float ratio = (p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale) / (p.w_inv_scale_r_dev ? *p.w_inv_scale_r_dev : p.w_inv_scale_r);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) acc[mb][ph][ch] *= ratio;
    }
    // A rider chunk is short (48 MFMAs per wave): its input is fetched TWO chunks ahead, into two register sets that alternate
    // (on entry the first set holds chunk k0 + 1, as the main loop leaves it; two of the three weight-ring slots are free by now).
    const int k0 = kc0 > nk ? kc0 : nk;
    constexpr int AH = K32_RIDER2 && NW == 8 ? 2 : 1;
    if (AH == 2 && k0 + 2 < kc1) prefetch_to(k0 + 2, rin2);
    auto rider_chunk = [&](int kc, QM* r1) __attribute__((always_inline)) {   // r1 holds chunk kc + 1
      unsigned char* cur = ((kc - kc0) & 1) ? sBuf1 : sBuf0;
      unsigned char* nxt = ((kc - kc0) & 1) ? sBuf0 : sBuf1;
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) xptr[1][pl] = cur + xoff[1][pl];
#pragma unroll
      for (int mb = 0; mb < XS - 1 && mb < MB; ++mb) load_x(mb, 1, 1, mb);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        if (mb + XS - 1 < MB) load_x((mb + XS - 1) % XS, 1, 1, mb + XS - 1);
        mfma_step(mb % XS, 0, mb);
      }
      if (kc + 1 < kc1) {
        load_w(kc + 1, 0, 0);
        stage_from(kc + 1, nxt, r1);
        if (kc + 1 + AH < kc1) prefetch_to(kc + 1 + AH, r1);
      }
      __syncthreads();
    };
    QM* rfirst = SPLIT ? rr1 : rin;
    if (AH == 2) {
      for (int kc = k0; kc < kc1; kc += 2) {
        rider_chunk(kc, rfirst);
        if (kc + 1 < kc1) rider_chunk(kc + 1, rin2);
      }
    } else {
      for (int kc = k0; kc < kc1; ++kc) rider_chunk(kc, rfirst);
    }
  }

  // ---- epilogue: lane = pixel c15 (+ 16 ph) of row wm + mb*WM, output channels cob + 16 ch + 0..3 ----
  // bias (+ the rider's) + noise shift of this lane's output channels.  Every load is issued unconditionally (a clamped channel, the
  // noise shift through a 0 / 1 factor): with `if (p.temb)` / `if (cok)` around them the compiler put an s_waitcnt vmcnt(0) behind each
  // of the eight pairs -- eight exposed round trips at the end of every tile, with nothing else running on the CU.
  k_f32x4 add[2];
  bool cok[2];
  const float* tembp = p.temb ? p.temb + (size_t)n * p.temb_stride + p.temb_off : p.bias;
  const float tmul = p.temb ? 1.f : 0.f;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const int co = cob + 16 * ch;
    cok[ch] = co < p.Cout;   // Cout % 4 == 0 (launcher)
    const int cs = cok[ch] ? co : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float a = p.bias[cs + r];
      if (RIDER) a += p.bias_r[cs + r];
      a += tmul * tembp[cs + r];
      add[ch][r] = a;
    }
  }
  const float winv_m = p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale;
  const float winv = RIDER && !RFIRST ? (p.w_inv_scale_r_dev ? *p.w_inv_scale_r_dev : p.w_inv_scale_r) : winv_m;
  if (RIDER && !RFIRST && kc1 <= nk) {   // (split K) a slice that never reached the rider chunks: bring it to the rider's scale too
    const float ratio = winv_m / winv;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) acc[mb][ph][ch] *= ratio;
  }
  if (SK > 1) {   // raw partial accumulators; bias, shift, residual and statistics happen in the reduce
    float* sb = p.kscratch + (size_t)ksi * p.N * p.Hout * p.Wout * p.Cout;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const int oy = oy0 + wm + mb * WM, ox = ox0 + 16 * ph + c15;
          if (interior || (cok[ch] && oy < p.Hout && ox < p.Wout))
            *reinterpret_cast<k_f32x4*>(sb + obase + mb * rstride + ph * pstride + 16 * ch) = acc[mb][ph][ch];
        }
    return;
  }
  k_f32x4 s1[2], s2[2];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) s1[ch] = s2[ch] = k_f32x4{0.f, 0.f, 0.f, 0.f};
  auto epilogue = [&](auto out16_tag) __attribute__((always_inline)) {
    constexpr bool OUT16 = decltype(out16_tag)::value;
    auto put4 = [&](size_t idx, k_f32x4 v) {
      if (OUT16) {
        *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + idx) = pack4_16<PREC>(v);
      } else {
        *reinterpret_cast<k_f32x4*>(p.out + idx) = v;
      }
    };
    if (interior) {
      if (p.res && !res_early) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int ph = 0; ph < 2; ++ph)
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) rv[mb][ph][ch] = IO::load4(p.res, obase + mb * rstride + ph * pstride + 16 * ch);
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            k_f32x4 v = acc[mb][ph][ch] * winv + add[ch];
            if (p.res) v += IO::widen(rv[mb][ph][ch]);
            put4(obase + mb * rstride + ph * pstride + 16 * ch, v);
            if (!(K32_DIAG & 4)) {
              s1[ch] += v;
              s2[ch] += v * v;
            }
          }
    } else {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            const int oy = oy0 + wm + mb * WM, ox = ox0 + 16 * ph + c15;
            if (cok[ch] && oy < p.Hout && ox < p.Wout) {
              const size_t idx = obase + mb * rstride + ph * pstride + 16 * ch;
              k_f32x4 v = acc[mb][ph][ch] * winv + add[ch];
              if (p.res) v += IO::widen(IO::load4(p.res, idx));
              put4(idx, v);
              s1[ch] += v;
              s2[ch] += v * v;
            }
          }
    }
  };
  // GroupNorm backward, first half (fp32 output, no residual, no K split: the launcher checks): per output quad the raw forward input
  // x and the dropout bytes are fetched one tile row ahead of the arithmetic; every load is unconditional on a clamped address.
  auto epilogue_gnb = [&](auto int_tag) __attribute__((always_inline)) {
    constexpr bool INTERIOR = decltype(int_tag)::value;   // (a compile-time copy of `interior`: the whole-tile path has no branch at all)
    const int Ct = p.Cout, cpg = Ct / p.gb_G;
    const bool hm = p.gb_mask != nullptr;
    const unsigned char* mk = hm ? p.gb_mask : reinterpret_cast<const unsigned char*>(p.gb_scale);
    const float ds = p.gb_drop;
    // one half of the lane's output channels at a time: the per-channel constants of both halves, the x tile and the accumulators
    // together do not fit the register file
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int cs = cok[ch] ? cob + 16 * ch : 0;
      const k_f32x4 gsc = *reinterpret_cast<const k_f32x4*>(p.gb_scale + (size_t)n * Ct + cs);
      const k_f32x4 gsh = *reinterpret_cast<const k_f32x4*>(p.gb_shift + (size_t)n * Ct + cs);
      k_f32x4 gme, grs;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int grp = (cs + r) / cpg;
        gme[r] = p.gb_stats[((size_t)n * p.gb_G + grp) * 2];
        grs[r] = p.gb_stats[((size_t)n * p.gb_G + grp) * 2 + 1];
      }
      const float* xb;
      int xcs;
      if (cs < p.gb_C0) { xb = p.gb_x0 + cs; xcs = p.gb_C0; }
      else { xb = p.gb_x1 + (cs - p.gb_C0); xcs = Ct - p.gb_C0; }
      k_f32x4 xq[MB][2];
      unsigned mq[MB][2];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int oy = min(oy0 + wm + mb * WM, p.Hout - 1);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const size_t pix = (size_t)(n * p.Hout + oy) * p.Wout + min(ox0 + 16 * ph + c15, p.Wout - 1);
          if (K32_GNB_DIAG & 2) { xq[mb][ph] = gsc; mq[mb][ph] = 0x01010101u; continue; }
          xq[mb][ph] = *reinterpret_cast<const k_f32x4*>(xb + pix * xcs);
          mq[mb][ph] = *reinterpret_cast<const unsigned*>(mk + (hm ? pix * Ct + cs : (size_t)0));
        }
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int oy = oy0 + wm + mb * WM;
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
          const int ox = ox0 + 16 * ph + c15;
          const bool ok = INTERIOR || (cok[ch] && oy < p.Hout && ox < p.Wout);
          const k_f32x4 v = acc[mb][ph][ch] * winv + add[ch];
          const k_f32x4 x = xq[mb][ph];
          const unsigned m = mq[mb][ph];
          k_f32x4 gq, gx;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float d = v[r];
            if (hm) d = ((m >> (8 * r)) & 0xffu) ? d * ds : 0.f;
            const float xh = (x[r] - gme[r]) * grs[r];
            float gg = d;
            if (!p.gb_plain && !(K32_GNB_DIAG & 1)) {
              const float u = fmaf(x[r], gsc[r], gsh[r]);
              const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-u));   // (the forward's own sigmoid: silu_k)
              gg = d * (sg * (1.0f + u * (1.0f - sg)));     // d/du [u * sigmoid(u)]
            }
            gq[r] = gg;
            gx[r] = gg * xh;
          }
          if (ok) {
            *reinterpret_cast<k_f32x4*>(p.out + obase + mb * rstride + ph * pstride + 16 * ch) = gq;
            s1[ch] += gq;
            s2[ch] += gx;
          }
        }
      }
    }
  };
  if constexpr (GNB) {
    if (interior) epilogue_gnb(std::true_type{});
    else epilogue_gnb(std::false_type{});
  } else {
    if (prec_is16(PREC) && !p.out_f32) epilogue(std::true_type{});
    else epilogue(std::false_type{});
  }
  if (p.part_out) {
    // Per-channel (sum, sumsq) of this wave's 4 x 32 pixels: the 16 lanes of a k group hold 16 values each
    // (value v = ch*8 + r*2 + stat); a halving butterfly over the pixel lanes leaves lane c15 with the total of value c15.
    float vals[16];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int r = 0; r < 4; ++r) { vals[ch * 8 + r * 2] = s1[ch][r]; vals[ch * 8 + r * 2 + 1] = s2[ch][r]; }
#pragma unroll
    for (int half = 8; half >= 1; half >>= 1) {
      const bool up = (c15 & half) != 0;
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const float keep = up ? vals[i + half] : vals[i];
        const float send = up ? vals[i] : vals[i + half];
        vals[i] = keep + __shfl_xor(send, half, 64);
      }
    }
    // the main loop ended with a barrier: the halo buffers are free
    float* sp = reinterpret_cast<float*>(smem_k);   // [WM][BN][2]
    {
      const int ch = c15 >> 3, r = (c15 >> 1) & 3, st = c15 & 1;
      sp[(wm * BN + wn * 32 + 16 * ch + 4 * g + r) * 2 + st] = vals[0];
    }
    __syncthreads();
    float a_keep = 0.f, b_keep = 0.f;
    if (tid < BN && co0 + tid < p.Cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { a += sp[(w * BN + tid) * 2 + 0]; b += sp[(w * BN + tid) * 2 + 1]; }
      a_keep = a; b_keep = b;
      float* dst = p.part_out + (((size_t)n * (tilesX * tilesY) + ty * tilesX + tx) * p.Cout + co0 + tid) * 2;
      dst[0] = a;
      dst[1] = b;
    }
    if (p.gsum_out) {   // the consumer-side GroupNorm: even threads add their channel pair's sums (Cout is even: launcher)
      const float a2 = a_keep + __shfl_xor(a_keep, 1, 64), b2 = b_keep + __shfl_xor(b_keep, 1, 64);
      if (tid < BN && !(tid & 1) && co0 + tid < p.Cout) gsum_add(p.gsum_out, n, p.Cout >> 1, (co0 + tid) >> 1, xcc_id(), a2, b2);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The sub-pixel form of Upsample(nearest x2) + Conv3x3 (fdsr_conv_up2.hip: four 2x2 convolutions on the source grid, one per output
// parity, with pre-summed weights) on the same 16x16x32 machinery: same ConvParams, same grid (source tile x row parity py x cout
// block), the same packed weights ([cot][kc16][wn][py][slot = px*4 + a*2 + b][plane][lane], read by the same 16-byte permutation).
// A wave owns 2 source rows x 32 source pixels x 32 couts for both column parities: acc[row][px][pixel half][cout half].  The eight
// (px, a, b) weight slots of a 32-channel chunk would be 128 VGPRs resident, so the loop runs SLOT-outer / row-inner with a ring of
// four slots: slot s + 4 is fetched right after slot s's 24 MFMAs.  The activation fragment of (row, a, shift c = px + b) is re-read
// per slot (8 instead of 6 fragment reads per row: the LDS has the room), one (slot, row) step ahead, pinned.
template <int TH, int WN, int PREC>
struct ConvUp2K32Cfg {
  static constexpr int TW = 32, KC = 32;
  static constexpr int NP = PREC == PREC_F16X3 ? 2 : 1;
  static constexpr int WNP = PREC == PREC_BF16 ? 1 : 2;   // (as ConvK32Cfg)
  static constexpr int ROWB = 64 * NP;
  static constexpr int HH = TH + 1, HWD = TW + 2, NPIX = HH * HWD;
  static constexpr int WM = 8 / WN, BN = 32 * WN, MB = TH / WM;
  static constexpr int RPP = 512 / 8, NIN = (NPIX + RPP - 1) / RPP;
  static constexpr int BUF_BYTES = NPIX * ROWB;
  static constexpr int R = 4;   // weight slots in registers (a ring: 8 % R == 0)
  static_assert(MB == 2, "two source rows per wave");
};

template <int TH, int WN, int PREC>
__global__ void __launch_bounds__(512, 2) conv_up2_k32_kernel(const ConvParams p) {
  using Cfg = ConvUp2K32Cfg<TH, WN, PREC>;
  constexpr int TW = Cfg::TW, KC = Cfg::KC, NP = Cfg::NP, ROWB = Cfg::ROWB, HWD = Cfg::HWD, NPIX = Cfg::NPIX;
  constexpr int WM = Cfg::WM, BN = Cfg::BN, MB = Cfg::MB, RPP = Cfg::RPP, NIN = Cfg::NIN, R = Cfg::R;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_uk[];
  unsigned char* sBuf0 = smem_uk;
  unsigned char* sBuf1 = smem_uk + Cfg::BUF_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int c15 = lane & 15, g = lane >> 4;

  // source-resolution tiling; p.Hin/Win = source dims, p.Hout/Wout = 2x (as conv_up2_h_kernel)
  const int nco = p.Cout_pad / BN;
  const int tilesX = (p.Win + TW - 1) / TW, tilesY = (p.Hin + TH - 1) / TH;
  int bid;
  {
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = b & 7, k = b >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + k;
  }
  const int cot = bid % nco;
  int pt = bid / nco;
  const int py = pt & 1;
  pt >>= 1;
  const int tx = pt % tilesX;
  pt /= tilesX;
  const int ty = pt % tilesY;
  const int n = pt / tilesY;
  const int oy0 = ty * TH, ox0 = tx * TW, co0 = cot * BN;

  // ---- staging: halo rows oy0-1+py .. oy0+TH-1+py, cols ox0-1 .. ox0+32 of the SOURCE; raw input (no GroupNorm before an upsample conv)
  const int q = tid & 7, row0 = tid >> 3;
  int in_pix[NIN];
  unsigned skeys = 0;
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int pix = row0 + i * RPP;
    int v = -2;
    if (pix < NPIX) {
      const int hy = pix / HWD, hx = pix % HWD;
      const int iy = oy0 - 1 + py + hy, ix = ox0 - 1 + hx;
      const bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
      v = ok ? (n * p.Hin + iy) * p.Win + ix : -1;
      skeys |= (unsigned)k32_slot<PREC>(0, hx) << (3 * i);
    }
    in_pix[i] = v;
  }
  const int sbase = row0 * ROWB + 8 * (q & 1);
  using IO = ActIO<PREC>;
  typedef typename IO::Quad Quad;
  Quad rin[NIN];
  auto prefetch = [&](int kc) {
    const size_t coff = (size_t)kc * KC + q * 4;
#pragma unroll
    for (int i = 0; i < NIN; ++i) rin[i] = IO::load4(p.x0, (size_t)(in_pix[i] < 0 ? 0 : in_pix[i]) * p.C0 + coff);
  };
  auto stage = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      if (NIN * RPP > NPIX && i == NIN - 1 && row0 + i * RPP >= NPIX) continue;
      k_f32x4 v = IO::widen(rin[i]);
      const int sd = sbase + i * (RPP * ROWB) + 16 * ((q >> 1) ^ (int)((skeys >> (3 * i)) & 7u));
      if (PREC == PREC_F16X3) {
        if (p.sat_flag) sat_check(p.sat_flag, v, 65504.f);
        const float lim = in_pix[i] >= 0 ? 65504.f : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -lim, lim);
        typedef _Float16 h2t __attribute__((ext_vector_type(2)));
        uint2 hi, lo;
        {
          const h2t h0 = {(_Float16)v[0], (_Float16)v[1]}, h1 = {(_Float16)v[2], (_Float16)v[3]};
          hi.x = __builtin_bit_cast(unsigned, h0);
          hi.y = __builtin_bit_cast(unsigned, h1);
        }
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo.x) : "v"(hi.x), "v"(v[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo.x) : "v"(hi.x), "v"(v[1]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo.y) : "v"(hi.y), "v"(v[2]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo.y) : "v"(hi.y), "v"(v[3]));
        *reinterpret_cast<uint2*>(buf + sd) = hi;
        *reinterpret_cast<uint2*>(buf + (sd ^ 64)) = lo;
      } else {
        const float keep = in_pix[i] >= 0 ? 1.f : 0.f;
        v = v * keep;
        *reinterpret_cast<uint2*>(buf + sd) = stage4_16<PREC>(v);
      }
    }
  };

  // ---- weight fragments: ring slot s % R; [cout half][plane] ----
  const uint4* wq = reinterpret_cast<const uint4*>(p.wq);
  const int nk = p.Cin_pad / KC, nk16 = p.Cin_pad / 16;
  uint4 Wf[R][2][NP];
  const int wlane = 32 * (g & 1) + c15;
  auto load_w = [&](int kc, int slot) {
    const uint4* src = wq + ((((((size_t)cot * nk16 + 2 * kc + (g >> 1)) * WN + wn) * 2 + py) * 8 + slot) * (Cfg::WNP * 64)) + wlane;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) Wf[slot % R][ch][pl] = src[pl * 64 + 16 * ch];
  };

  int xoff[3][NP];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int hx = c15 + c;
    xoff[c][0] = (wm * HWD + hx) * ROWB + 16 * k32_slot<PREC>(g, hx);
    if (NP == 2) xoff[c][NP - 1] = xoff[c][0] ^ 64;
  }

  k_f32x4 acc[MB][2][2][2];   // [row][px][pixel half][cout half]
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) acc[mb][px][ph][ch] = k_f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int sl = 0; sl < R; ++sl) load_w(0, sl);
  prefetch(0);
  stage(sBuf0);
  if (nk > 1) prefetch(1);
  __syncthreads();

  uint4 Xf[2][2][NP];   // [ring slot][pixel half][plane]
  const unsigned char* xptr[3][NP];
  auto load_x = [&](int xs, int slot, int mb) {
    const int a = (slot >> 1) & 1, c = (slot >> 2) + (slot & 1);   // halo row a, column shift px + b
#pragma unroll
    for (int ph = 0; ph < 2; ++ph)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl)
        Xf[xs][ph][pl] = *reinterpret_cast<const uint4*>(xptr[c][pl] + ((mb * WM + a) * HWD + 16 * ph) * ROWB);
  };
  auto mfma_step = [&](int xs, int slot, int mb) {
    const int px = slot >> 2, ws = slot % R;
    if (PREC == PREC_F16X3) {
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int ch = 0; ch < 2; ++ch) {
            const uint4 xv = Xf[xs][ph][term == 0 ? NP - 1 : 0];
            const uint4 wv = Wf[ws][ch][term == 1 ? NP - 1 : 0];
            acc[mb][px][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(k_h8, wv), __builtin_bit_cast(k_h8, xv),
                                                                        acc[mb][px][ph][ch], 0, 0, 0);
          }
    } else {
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch)
          acc[mb][px][ph][ch] = mfma16_k32<PREC>(Wf[ws][ch][0], Xf[xs][ph][0], acc[mb][px][ph][ch]);
    }
  };

  for (int kc = 0; kc < nk; ++kc) {
    unsigned char* cur = (kc & 1) ? sBuf1 : sBuf0;
    unsigned char* nxt = (kc & 1) ? sBuf0 : sBuf1;
    const bool more = kc + 1 < nk;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) xptr[c][pl] = cur + xoff[c][pl];
    load_x(0, 0, 0);
#pragma unroll
    for (int slot = 0; slot < 8; ++slot) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int s = slot * MB + mb;
        if (s + 1 < 8 * MB) load_x((s + 1) & 1, (s + 1) / MB, (s + 1) % MB);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(s & 1, slot, mb);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (slot + R < 8) load_w(kc, slot + R);
      else if (more) load_w(kc + 1, slot + R - 8);
      if (slot == 3 && more) {   // mid-chunk: fill the other halo buffer
        stage(nxt);
        if (kc + 2 < nk) prefetch(kc + 2);
      }
    }
    __syncthreads();
  }

  // ---- epilogue: out[2 (oy0 + row) + py][2 (ox0 + col) + px], + bias (+ temb), GroupNorm partials of the output ----
  const int cob = co0 + wn * 32 + 4 * g;
  const float winv = p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale;
  k_f32x4 add[2];
  bool cok[2];
  const float* tembp = p.temb ? p.temb + (size_t)n * p.temb_stride + p.temb_off : p.bias;   // (all loads unconditional: see conv_k32_kernel)
  const float tmul = p.temb ? 1.f : 0.f;
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) {
    const int co = cob + 16 * ch;
    cok[ch] = co < p.Cout;
    const int cs = cok[ch] ? co : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) add[ch][r] = p.bias[cs + r] + tmul * tembp[cs + r];
  }
  k_f32x4 s1[2], s2[2];
#pragma unroll
  for (int ch = 0; ch < 2; ++ch) s1[ch] = s2[ch] = k_f32x4{0.f, 0.f, 0.f, 0.f};
  const bool out16 = prec_is16(PREC);     // bf16 / f16 mode: the upsample conv's output is a 16-bit activation
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          const int sy = oy0 + wm + mb * WM, sx = ox0 + 16 * ph + c15;
          if (cok[ch] && sy < p.Hin && sx < p.Win) {
            const size_t idx = ((size_t)(n * p.Hout + 2 * sy + py) * p.Wout + 2 * sx + px) * p.Cout + cob + 16 * ch;
            const k_f32x4 v = acc[mb][px][ph][ch] * winv + add[ch];
            if (out16) {
              *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + idx) = pack4_16<PREC>(v);
            } else {
              *reinterpret_cast<k_f32x4*>(p.out + idx) = v;
            }
            s1[ch] += v;
            s2[ch] += v * v;
          }
        }
  if (p.part_out) {
    float vals[16];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch)
#pragma unroll
      for (int r = 0; r < 4; ++r) { vals[ch * 8 + r * 2] = s1[ch][r]; vals[ch * 8 + r * 2 + 1] = s2[ch][r]; }
#pragma unroll
    for (int half = 8; half >= 1; half >>= 1) {
      const bool up = (c15 & half) != 0;
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const float keep = up ? vals[i + half] : vals[i];
        const float send = up ? vals[i] : vals[i + half];
        vals[i] = keep + __shfl_xor(send, half, 64);
      }
    }
    float* sp = reinterpret_cast<float*>(smem_uk);   // halo buffers are free after the last barrier
    {
      const int ch = c15 >> 3, r = (c15 >> 1) & 3, st = c15 & 1;
      sp[(wm * BN + wn * 32 + 16 * ch + 4 * g + r) * 2 + st] = vals[0];
    }
    __syncthreads();
    float a_keep = 0.f, b_keep = 0.f;
    if (tid < BN && co0 + tid < p.Cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { a += sp[(w * BN + tid) * 2 + 0]; b += sp[(w * BN + tid) * 2 + 1]; }
      a_keep = a; b_keep = b;
      float* dst = p.part_out + (((size_t)n * (tilesX * tilesY * 2) + (ty * tilesX + tx) * 2 + py) * p.Cout + co0 + tid) * 2;
      dst[0] = a;
      dst[1] = b;
    }
    if (p.gsum_out) {   // the consumer-side GroupNorm (as conv_k32_kernel)
      const float a2 = a_keep + __shfl_xor(a_keep, 1, 64), b2 = b_keep + __shfl_xor(b_keep, 1, 64);
      if (tid < BN && !(tid & 1) && co0 + tid < p.Cout) gsum_add(p.gsum_out, n, p.Cout >> 1, (co0 + tid) >> 1, xcc_id(), a2, b2);
    }
  }
}

bool conv_up2_k32_ok(int prec, const ConvParams& p) {
  if (!(g_tun.k32 & FDSR_K32_UP2) || !(g_tun.k32 & (prec_is16(prec) ? FDSR_K32_BF16 : FDSR_K32_F16X3))) return false;
  return p.C1 == 0 && p.C0 % 32 == 0 && p.C0 == p.Cin_pad && p.Cout % 4 == 0 && !p.gn_scale && !p.res;
}

template <int TH, int WN, int PREC>
static hipError_t launch_up2_k32_t(const ConvParams& q, int nwg, hipStream_t s) {
  using Cfg = ConvUp2K32Cfg<TH, WN, PREC>;
  hipLaunchKernelGGL((conv_up2_k32_kernel<TH, WN, PREC>), dim3(nwg), dim3(512), (size_t)2 * Cfg::BUF_BYTES, s, q);
  return hipGetLastError();
}

#define FDSR_UP2_K32_SHAPES(X) X(8, 2) X(4, 4) X(2, 8)

hipError_t launch_conv_up2_k32(int TH, int WN, int prec, const ConvParams& q, int nwg, hipStream_t s) {
#define X(TH_, WN_)                                                                                    \
  if (TH == TH_ && WN == WN_)                                                                          \
    return prec == PREC_F16X3 ? launch_up2_k32_t<TH_, WN_, PREC_F16X3>(q, nwg, s) : prec == PREC_F16 ? launch_up2_k32_t<TH_, WN_, PREC_F16>(q, nwg, s) : launch_up2_k32_t<TH_, WN_, PREC_BF16>(q, nwg, s);
  FDSR_UP2_K32_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}

// (TH, WN) pairs with MB = TH / (8 / WN) = 4, and the 2-row-per-wave tiles of small grids (MB = 2)
#define FDSR_K32_SHAPES(X) X(16, 2) X(8, 4) X(4, 8) X(8, 2) X(4, 4) X(2, 8)

bool conv_k32_ok(int TH, int WN, int prec, const ConvParams& p) {
  // g_tun.k32 bits (default 27 = 1|2|8|16): 1 f16x3, 2 bf16 (+2.7 % at B=64 once the ring / pinning / peeled last chunk were in), 4 the 16-row tile with a rider (measured slower: off)
  if (!(g_tun.k32 & (prec_is16(prec) ? FDSR_K32_BF16 : FDSR_K32_F16X3))) return false;
  if (TH == 16 && p.xr0 && !(g_tun.k32 & FDSR_K32_RIDER_16ROW)) return false;
  if (TH * WN != 32 && !(TH * WN == 16 && (g_tun.k32 & FDSR_K32_SMALL_GRID_2ROW))) return false;   // MB == 4 (bit 8: the 2-row tiles of small grids too)
  if (WN != 2 && WN != 4 && WN != 8) return false;
  if (p.Cin_pad % 32 || (p.C0 + p.C1) != p.Cin_pad) return false;    // whole 32-channel chunks
  if (p.C1 != 0 && p.C0 % 32) return false;                          // the concat seam on a chunk boundary
  if (p.Cout % 4) return false;                                      // 16-byte stores
  if (p.xr0) {
    if (p.nkr % 2 || (p.Cr0 + p.Cr1) != p.nkr * 16) return false;
    if (p.Cr1 != 0 && p.Cr0 % 32) return false;
  }
  return true;
}

template <int TH, int WN, int PREC, bool RIDER, int NW = 8, bool RF = (NW == 4), bool GNB = false, bool DROP = false, bool GNC = false>
static hipError_t launch_k32_t(const ConvParams& q, int nwg, hipStream_t s) {
  using Cfg = ConvK32Cfg<TH, WN, PREC, NW>;
  hipLaunchKernelGGL((conv_k32_kernel<TH, WN, PREC, RIDER, NW, RF, GNB, DROP, GNC>), dim3(nwg), dim3(Cfg::NT), (size_t)Cfg::LDS_BYTES + (GNC ? K32_GNC_LDS : 0), s, q);
  return hipGetLastError();
}

// train-mode dropout in the staging: f16x3, one GroupNorm'd + Swish'd input, no K split, a rider only in the rider-first form
bool conv_k32_drop_ok(int prec, const ConvParams& q) {
  if (prec != PREC_F16X3 || !q.drop_mask || !q.gn_scale || q.gn_plain || q.C1 != 0 || q.ksplit > 1 || q.gb_x0) return false;
  return !q.xr0 || (g_tun.k32 & FDSR_K32_RIDER_FIRST_8WAVE);
}

// the GroupNorm-backward epilogue: f16x3, fp32 output, no rider, no residual, no K split, whole groups of the Cout channels
static bool k32_gnb_ok(int prec, const ConvParams& q) {
  if (prec != PREC_F16X3 || !q.out_f32 || q.xr0 || q.res || q.ksplit > 1 || !q.gb_scale || !q.gb_shift || !q.gb_stats) return false;
  if (q.gb_G <= 0 || q.Cout % q.gb_G || q.Cout % 4) return false;
  return q.gb_x1 ? (q.gb_C0 % 4 == 0 && q.gb_C0 > 0 && q.gb_C0 < q.Cout) : q.gb_C0 == q.Cout;
}

// The small-workgroup form (NW = 4, two workgroups per CU) of the 64-cout launches: 6-row tiles (what two double-buffered f16x3 halo
// images per CU leave room for).  Large grids only: a small grid wants all eight waves of a CU on its one
// tile.  launch_conv_h asks before it picks a tile of its own; tiles / nwg are computed here.
static int k32_small_rows(int prec, bool rider) { return prec_is16(prec) && !rider ? 8 : 6; }   // (the bf16 8-row tile with a rider spills 27 VGPRs)

bool conv_k32_small_ok(ConvKind kind, int prec, const ConvParams& p) {
  if (!(g_tun.k32 & FDSR_K32_SMALL_WG_F16X3) || kind != CONV3_S1 || p.ksplit > 1 || p.Cout_pad != 64) return false;
  if (prec_is16(prec) && !(g_tun.k32 & FDSR_K32_SMALL_WG_BF16)) return false;   // bf16: bit 128 (its 6-row tile measured -0.4 % end to end, its 8-row tile +1.0 %: on)
  // launches with a rider: bit 64 in f16x3 (on since the rider chunks run FIRST on this form: +0.7 % end to end over the 32x32x16 rider
  // kernel; main chunks first it measured 4 - 7 % slower on those launches), bit 512 in bf16 (off: -0.4 %)
  if (p.xr0 && !(g_tun.k32 & (prec_is16(prec) ? FDSR_K32_SMALL_WG_RIDER_BF16 : FDSR_K32_SMALL_WG_RIDER_F16X3))) return false;
  if (!conv_k32_ok(8, 4, prec, p)) return false;          // the form's own conditions (an MB = 4 shape: no tile-size bits involved)
  const int th = k32_small_rows(prec, p.xr0 != nullptr);
  const long wgs = (long)p.N * ((p.Wout + 31) / 32) * ((p.Hout + th - 1) / th);
  return wgs >= g_tun.k32_sb_min_wgs;
}

hipError_t launch_conv_k32_small(int prec, const ConvParams& p, hipStream_t s, int* tiles) {
  const int th = k32_small_rows(prec, p.xr0 != nullptr);
  const int tilesX = (p.Wout + 31) / 32, tilesY = (p.Hout + th - 1) / th;
  if (tiles) *tiles = tilesX * tilesY;
  const int nwg = p.N * tilesX * tilesY;
  ConvParams q = p;
  q.out_bf16 = p.out_f32 ? 0 : prec_act16(prec);
  q.stagger = g_tun.k32_stagger;
  if (q.gb_x0) return k32_gnb_ok(prec, q) ? launch_k32_t<6, 2, PREC_F16X3, false, 4, true, true>(q, nwg, s) : hipErrorInvalidValue;
  if (q.drop_mask) {
    if (!conv_k32_drop_ok(prec, q)) return hipErrorInvalidValue;
    return q.xr0 ? launch_k32_t<6, 2, PREC_F16X3, true, 4, true, false, true>(q, nwg, s) : launch_k32_t<6, 2, PREC_F16X3, false, 4, true, false, true>(q, nwg, s);
  }
  if (prec == PREC_F16X3)
    return q.xr0 ? launch_k32_t<6, 2, PREC_F16X3, true, 4>(q, nwg, s) : launch_k32_t<6, 2, PREC_F16X3, false, 4>(q, nwg, s);
  if (prec == PREC_F16) return q.xr0 ? launch_k32_t<6, 2, PREC_F16, true, 4>(q, nwg, s) : launch_k32_t<8, 2, PREC_F16, false, 4>(q, nwg, s);
  return q.xr0 ? launch_k32_t<6, 2, PREC_BF16, true, 4>(q, nwg, s) : launch_k32_t<8, 2, PREC_BF16, false, 4>(q, nwg, s);
}

// the consumer-side GroupNorm (ConvParams::gs0): the 2-row-per-wave tiles of small grids (MB = 2: 32 accumulator registers leave the room),
// sampling launches only (no GroupNorm-backward epilogue, no dropout staging)
#define FDSR_K32_GNC_SHAPES(X) X(8, 2) X(4, 4) X(2, 8)
bool conv_k32_gnc_ok(int TH, int WN, int prec, const ConvParams& p) {
  if (TH * WN != 16 || !conv_k32_ok(TH, WN, prec, p)) return false;
  if (p.gb_x0 || p.drop_mask || p.gn_plain || (p.C0 & 1) || (p.C1 & 1)) return false;
  const int Cin = p.C0 + p.C1;
  return Cin <= K32_GNC_MAXC && p.gs_G > 0 && Cin % p.gs_G == 0 && ((Cin / p.gs_G) & 1) == 0;
}

hipError_t launch_conv_k32(int TH, int WN, int prec, const ConvParams& q, int nwg, hipStream_t s) {
  if (q.gs0) {
    if (!conv_k32_gnc_ok(TH, WN, prec, q)) return hipErrorInvalidValue;
    const bool rf = q.xr0 && q.ksplit <= 1 && (g_tun.k32 & FDSR_K32_RIDER_FIRST_8WAVE);
#define X(TH_, WN_)                                                                                                      \
    if (TH == TH_ && WN == WN_) {                                                                                        \
      if (rf) return prec == PREC_F16X3 ? launch_k32_t<TH_, WN_, PREC_F16X3, true, 8, true, false, false, true>(q, nwg, s)   \
                                        : prec == PREC_F16 ? launch_k32_t<TH_, WN_, PREC_F16, true, 8, true, false, false, true>(q, nwg, s) : launch_k32_t<TH_, WN_, PREC_BF16, true, 8, true, false, false, true>(q, nwg, s);    \
      if (q.xr0) return prec == PREC_F16X3 ? launch_k32_t<TH_, WN_, PREC_F16X3, true, 8, false, false, false, true>(q, nwg, s)  \
                                           : prec == PREC_F16 ? launch_k32_t<TH_, WN_, PREC_F16, true, 8, false, false, false, true>(q, nwg, s) : launch_k32_t<TH_, WN_, PREC_BF16, true, 8, false, false, false, true>(q, nwg, s);   \
      return prec == PREC_F16X3 ? launch_k32_t<TH_, WN_, PREC_F16X3, false, 8, false, false, false, true>(q, nwg, s)      \
                                : prec == PREC_F16 ? launch_k32_t<TH_, WN_, PREC_F16, false, 8, false, false, false, true>(q, nwg, s) : launch_k32_t<TH_, WN_, PREC_BF16, false, 8, false, false, false, true>(q, nwg, s);       \
    }
    FDSR_K32_GNC_SHAPES(X)
#undef X
    return hipErrorInvalidValue;
  }
#define X(TH_, WN_)                                                                                             \
  if (TH == TH_ && WN == WN_) {                                                                                 \
    if (q.gb_x0) return k32_gnb_ok(prec, q) ? launch_k32_t<TH_, WN_, PREC_F16X3, false, 8, false, true>(q, nwg, s) : hipErrorInvalidValue; \
    if (q.drop_mask) {                                                                                          \
      if (!conv_k32_drop_ok(prec, q)) return hipErrorInvalidValue;                                              \
      return q.xr0 ? launch_k32_t<TH_, WN_, PREC_F16X3, true, 8, true, false, true>(q, nwg, s)                  \
                   : launch_k32_t<TH_, WN_, PREC_F16X3, false, 8, false, false, true>(q, nwg, s);               \
    }                                                                                                           \
    if (q.xr0 && q.ksplit <= 1 && (g_tun.k32 & FDSR_K32_RIDER_FIRST_8WAVE))                                                           \
      return prec == PREC_F16X3 ? launch_k32_t<TH_, WN_, PREC_F16X3, true, 8, true>(q, nwg, s)                  \
                                : prec == PREC_F16 ? launch_k32_t<TH_, WN_, PREC_F16, true, 8, true>(q, nwg, s) : launch_k32_t<TH_, WN_, PREC_BF16, true, 8, true>(q, nwg, s);                  \
    if (q.xr0) return prec == PREC_F16X3 ? launch_k32_t<TH_, WN_, PREC_F16X3, true>(q, nwg, s)                  \
                                         : prec == PREC_F16 ? launch_k32_t<TH_, WN_, PREC_F16, true>(q, nwg, s) : launch_k32_t<TH_, WN_, PREC_BF16, true>(q, nwg, s);                  \
    return prec == PREC_F16X3 ? launch_k32_t<TH_, WN_, PREC_F16X3, false>(q, nwg, s)                            \
                              : prec == PREC_F16 ? launch_k32_t<TH_, WN_, PREC_F16, false>(q, nwg, s) : launch_k32_t<TH_, WN_, PREC_BF16, false>(q, nwg, s);                            \
  }
  FDSR_K32_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}

template <int TH, int WN, int PREC, bool RIDER, int NW = 8, bool RF = (NW == 4), bool GNB = false, bool DROP = false, bool GNC = false>
static hipError_t init_k32_t() {
  auto kfn = conv_k32_kernel<TH, WN, PREC, RIDER, NW, RF, GNB, DROP, GNC>;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int TH, int WN, int PREC>
static hipError_t init_up2_k32_t() {
  auto kfn = conv_up2_k32_kernel<TH, WN, PREC>;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

hipError_t kernels_k32_init() {
  hipError_t e;
#define X(TH_, WN_)                                                                 \
  if ((e = init_up2_k32_t<TH_, WN_, PREC_F16X3>()) != hipSuccess) return e;         \
  if ((e = init_up2_k32_t<TH_, WN_, PREC_BF16>()) != hipSuccess) return e; \
  if ((e = init_up2_k32_t<TH_, WN_, PREC_F16>()) != hipSuccess) return e;
  FDSR_UP2_K32_SHAPES(X)
#undef X
#define X(TH_, WN_)                                                                        \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, false>()) != hipSuccess) return e;             \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, true>()) != hipSuccess) return e;              \
  if ((e = init_k32_t<TH_, WN_, PREC_BF16, false>()) != hipSuccess) return e;              \
  if ((e = init_k32_t<TH_, WN_, PREC_F16, false>()) != hipSuccess) return e;              \
  if ((e = init_k32_t<TH_, WN_, PREC_BF16, true>()) != hipSuccess) return e;                \
  if ((e = init_k32_t<TH_, WN_, PREC_F16, true>()) != hipSuccess) return e;                \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, true, 8, true>()) != hipSuccess) return e;     \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, false, 8, false, true>()) != hipSuccess) return e; \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, false, 8, false, false, true>()) != hipSuccess) return e; \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, true, 8, true, false, true>()) != hipSuccess) return e; \
  if ((e = init_k32_t<TH_, WN_, PREC_BF16, true, 8, true>()) != hipSuccess) return e; \
  if ((e = init_k32_t<TH_, WN_, PREC_F16, true, 8, true>()) != hipSuccess) return e;
  FDSR_K32_SHAPES(X)
#undef X
#define X(TH_, WN_)                                                                                                     \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, false, 8, false, false, false, true>()) != hipSuccess) return e;            \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, true, 8, false, false, false, true>()) != hipSuccess) return e;             \
  if ((e = init_k32_t<TH_, WN_, PREC_F16X3, true, 8, true, false, false, true>()) != hipSuccess) return e;              \
  if ((e = init_k32_t<TH_, WN_, PREC_BF16, false, 8, false, false, false, true>()) != hipSuccess) return e;             \
  if ((e = init_k32_t<TH_, WN_, PREC_F16, false, 8, false, false, false, true>()) != hipSuccess) return e;             \
  if ((e = init_k32_t<TH_, WN_, PREC_BF16, true, 8, false, false, false, true>()) != hipSuccess) return e;              \
  if ((e = init_k32_t<TH_, WN_, PREC_F16, true, 8, false, false, false, true>()) != hipSuccess) return e;              \
  if ((e = init_k32_t<TH_, WN_, PREC_BF16, true, 8, true, false, false, true>()) != hipSuccess) return e; \
  if ((e = init_k32_t<TH_, WN_, PREC_F16, true, 8, true, false, false, true>()) != hipSuccess) return e;
  FDSR_K32_GNC_SHAPES(X)
#undef X
  if ((e = init_k32_t<6, 2, PREC_F16X3, false, 4>()) != hipSuccess) return e;
  if ((e = init_k32_t<6, 2, PREC_F16X3, true, 4>()) != hipSuccess) return e;
  if ((e = init_k32_t<6, 2, PREC_F16X3, false, 4, true, true>()) != hipSuccess) return e;
  if ((e = init_k32_t<6, 2, PREC_F16X3, false, 4, true, false, true>()) != hipSuccess) return e;
  if ((e = init_k32_t<6, 2, PREC_F16X3, true, 4, true, false, true>()) != hipSuccess) return e;
  if ((e = init_k32_t<6, 2, PREC_BF16, true, 4>()) != hipSuccess) return e;
  if ((e = init_k32_t<6, 2, PREC_F16, true, 4>()) != hipSuccess) return e;
  return hipSuccess;
}

}  // namespace fdsr
