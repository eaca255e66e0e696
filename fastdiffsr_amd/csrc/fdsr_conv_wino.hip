// Winograd F(2x2, 3x3) form of the stride-1 3x3 convolutions on the split-f16 MFMA path (PREC_F16X3), gfx950.
//
// Block convs of the UNet (reference unet.py:89-101: GroupNorm -> Swish -> Conv2d 3x3, padding 1) cost 9 MFMA
// products per (pixel, cin, cout) in direct form; in the Winograd domain a 2x2 output tile needs 16 products
// instead of 36 (2.25x fewer):        Y = A^T [ sum_cin (G g G^T) . (B^T d B) ] A
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   (4x4 input patch d, rows/cols 2ty-1 .. 2ty+2)
//   G   = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]       (3x3 filter g; U = G g G^T packed at load, fp64 -> hi/lo f16)
//   A^T = [1 1 1 0; 0 1 -1 -1]
// The input transform runs in fp32 on the ACTIVATED tensor (GroupNorm-apply + Swish fused into the staging as in the
// direct kernel) BEFORE the hi/lo split, so operands keep 22 mantissa bits: x = hi + lo, product = hi*hi + hi*lo + lo*hi.
//
// OFF by default (fdsr_debug_option("wino", 2) turns it on): the 16x16x32 direct kernel (fdsr_conv_k32.hip) overtook it on every
// layer set in round 3; what stays is the half-tile pipeline form, as the documented option and A/B partner.
//
// Workgroup = 8 wave64 = one 16x16-pixel output block (8x8 tiles) x 64 output channels, all 16 Winograd positions:
//   * accumulators: 16 positions x 64 tiles x 64 couts fp32 = 256 KB = half the CU's register file.  Wave w owns
//     position row xi = w & 3 (4 positions), cout half w >> 2 and all 64 tiles: 8 accumulator tiles (128 VGPRs).
//   * per 16-channel chunk: halo 18x18 pixels -> registers -> GroupNorm + Swish -> LDS (fp32, double-buffered);
//     barrier; every thread transforms (tile, 4 channels, position-row pair) -> 8 positions -> hi/lo split -> LDS
//     image V[pos][tile][16 hi | 16 lo | pad] (80-byte rows: the fragment reads are bank-conflict free);
//     barrier; 24 MFMAs per wave on A fragments from V and B fragments (transformed weights) streamed straight from
//     L2 in fragment order, one chunk ahead, while the next chunk's halo is activated and stored.
//   * epilogue: each wave folds its position row over nu (Z[xi][j]), the four rows meet in LDS, every thread
//     finishes 8 (pixel, 4-cout) outputs: bias + noise-embedding shift + residual, coalesced 16-byte stores, and the
//     per-tile channel sums (GroupNorm statistics of the next Block), reduced in a fixed order.
// blockIdx -> (cout block, tile): workgroups of one XCD (blockIdx & 7) share ONE 64-cout block, so an XCD's 4 MiB L2
// holds Cin x 4 KB of transformed weights (2 MB at Cin = 512) instead of the whole layer (8 MB).
#include "fdsr_kernels.h"
#include "fdsr_act_io.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace fdsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int W_TT = 64;                       // tiles per workgroup: 8 x 8 tiles of 2 x 2 outputs
constexpr int W_HW = 18;                       // halo width / height in pixels
constexpr int W_NPIX = W_HW * W_HW;            // 324
constexpr int W_ROWB = 80;                     // V row of one (position, tile): [16 hi | 16 lo | 16 B pad]
constexpr int W_RAWB = 80;                     // raw halo pixel: 16 fp32 = 64 B + 16 B pad (odd number of 16-byte slots)
constexpr int W_V_BYTES = 16 * W_TT * W_ROWB;  // 81920
constexpr int W_RAW_BYTES = W_NPIX * W_RAWB;   // 25920
constexpr int W_ZROWB = 272;                   // exchange row: 64 couts fp32 + 16 B pad
constexpr int W_Z_BYTES = 8 * W_TT * W_ZROWB;  // 139264: Z[xi][j][tile][cout]
constexpr int W_NIN = 3;                       // halo (pixel, quad) items per thread: 324 * 4 / 512 -> 3 passes

__device__ __forceinline__ float silu_w(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

}  // namespace

// Perf-only knock-out builds (tools/build_wino_variant.sh -DWINO_KO_...): remove one part of the kernel to price it; results are garbage.
#define WINO_SINK4(u) asm volatile("" ::"v"((u).x), "v"((u).y), "v"((u).z), "v"((u).w))

// (The first, serial-phase form of this kernel and a 256-thread two-workgroups-per-CU form were built, measured and removed in
// round 4: they never beat the half-tile pipeline below -- DESIGN / EXPERIMENTS, "Winograd".)


// ---------------------------------------------------------------------------------------------------------------------
// Second form of the same kernel ("half-tile pipeline", the default): the measured first form spends its chunk time in
// SERIAL phases -- transform (VALU + LDS), barrier, MFMAs with the halo staging in the middle, barrier: 17 % MFMA busy,
// 17 VALU per MFMA (profiles/r03_wino_knockouts.txt).  Here every barrier interval holds BOTH kinds of work and the two
// waves of a SIMD run them in opposite order, so one wave's MFMAs cover its partner's VALU / LDS work:
//   phase A(k): MFMAs on tile block 0 of chunk k  |  transform tile block 1 of chunk k -> V1; stage items 1, 2 of chunk k+1
//   phase B(k): MFMAs on tile block 1 of chunk k  |  transform tile block 0 of chunk k+1 -> V0; stage item 0 of chunk k+2
// (waves 0-3: transform / staging first, then MFMAs; waves 4-7: MFMAs first).  V is two 32-tile halves, each written in
// one phase and read in the next; the fp32 halo stays double-buffered.  Fewer VALU per element: hi/lo split as one
// packed convert + two v_fma_mix per pair, wave-uniform roles in SGPRs (scalar branches), no range check (the input is
// GroupNorm'ed), halo rows padded to 1536 B so that the transform reads of a (tile row, tile row + 1) lane pair fall on
// the same banks modulo 64 words -- every ds_read_b128 group of 16 lanes covers 8 tile columns x 2 channel quads.
namespace {
constexpr int W2_RAWROW = 1536;                          // bytes per halo row (18 pixels x 80 B = 1440, padded)
constexpr int W2_RAW_BYTES = W_HW * W2_RAWROW;           // 27648
constexpr int W2_MAXCIN = 1024;                          // GroupNorm scale / shift of the whole input live in LDS: 8 B per channel
constexpr int W2_SS_OFF = W_V_BYTES + 2 * W2_RAW_BYTES;  // 137216
constexpr int W2_LDS = W2_SS_OFF + W2_MAXCIN * 8;        // 145408 (> the epilogue's exchange image, 139264)

// hi = rn_f16(v), lo = rn_f16(v - hi): one packed convert and two v_fma_mix per pair (lo = f16(fma(hi, -1, v)), rounded once:
// bit-identical to the two-step form); hi at dst, lo at dst + 32
__device__ __forceinline__ void split_store2(unsigned char* dst, f32x4 v) {
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
  *reinterpret_cast<uint2*>(dst + 32) = lo;
}
}  // namespace

#ifdef WINO_STAMPS
// Diagnostic build only (tools/build_wino_variant.sh -DWINO_STAMPS): s_memtime stamps of waves 0 and 4 of the first 256 workgroups
// of ONE layer (Cin = WINO_STAMP_CIN, Cout = WINO_STAMP_COUT) at the phase boundaries; read back by fdsr_diag_wino_stamps.
__device__ unsigned long long g_w2_stamps[256][2][128];
#define W2_STAMP()                                                                                                   \
  do {                                                                                                               \
    if (stamp_on && si < 128) g_w2_stamps[blockIdx.x][wave >> 2][si] = __builtin_amdgcn_s_memtime();                 \
    ++si;                                                                                                            \
  } while (0)
#ifdef WINO_STAMPS_FINE
#define W2_FSTAMP() W2_STAMP()
#else
#define W2_FSTAMP() do {} while (0)
#endif
#else
#define W2_STAMP() do {} while (0)
#define W2_FSTAMP() do {} while (0)
#endif

__global__ void __launch_bounds__(512, 2) conv_wino2_h_kernel(const ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
  unsigned char* sV = smem_w;
  unsigned char* sRaw0 = smem_w + W_V_BYTES;
  unsigned char* sRaw1 = sRaw0 + W2_RAW_BYTES;
  float* sSS = reinterpret_cast<float*>(smem_w + W2_SS_OFF);   // scale[Cin] | shift[Cin] of image n

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: roles below live in SGPRs
  const int Cin = p.C0 + p.C1;
  const int nk = p.Cin_pad >> 4;
  const int nco = p.Cout_pad >> 6;
  const int tilesX = p.Wout >> 4, ntile = tilesX * (p.Hout >> 4);
#ifdef WINO_STAMPS
  int si = 0;
  const bool stamp_on = lane == 0 && (wave & 3) == 0 && blockIdx.x < 256 && Cin == WINO_STAMP_CIN && p.Cout == WINO_STAMP_COUT;
#endif
  W2_STAMP();
  int cot, sp;
  {
    const int b = blockIdx.x;
    if (nco <= 8 && (8 % nco) == 0 && (gridDim.x & 7) == 0) {   // one cout block per XCD (round-robin dispatch: XCD = b & 7)
      const int xcd = b & 7, k = b >> 3, per = 8 / nco;
      cot = xcd % nco;
      sp = k * per + xcd / nco;
    } else {
      cot = b % nco;
      sp = b / nco;
    }
  }
  const int n = sp / ntile, tl = sp % ntile;
  const int oy0 = (tl / tilesX) * 16, ox0 = (tl % tilesX) * 16, co0 = cot * 64;

  // ---- halo staging: thread -> (pixel row0 + 128 i, channel quad q), i = 0..2 ----
  const int q = tid & 3, row0 = tid >> 2;
  int in_pix[W_NIN], raw_off[W_NIN];
#pragma unroll
  for (int i = 0; i < W_NIN; ++i) {
    const int pix = row0 + i * 128;
    int v = -2, ro = 0;
    if (pix < W_NPIX) {
      const int hy = pix / W_HW, hx = pix % W_HW;
      const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
      const bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
      v = ok ? (n * p.Hin + iy) * p.Win + ix : -1;
      ro = hy * W2_RAWROW + hx * W_RAWB + q * 16;
    }
    if (pix >= W_NPIX) ro = 18 * W_RAWB + q * 16;   // no third item on this lane: a dummy slot in the padding of halo row 0 (bytes 1440..1503)
    in_pix[i] = v;
    raw_off[i] = ro;
  }
  f32x4 rin[W_NIN];
  // per-(image, channel) GroupNorm scale / shift: read from LDS where they are used (no registers held across phases, and no
  // vector-memory wait that would drag the in-flight halo prefetch with it: vmcnt counts in issue order)
  for (int c4 = tid; c4 < (Cin >> 2); c4 += 512) {
    *reinterpret_cast<f32x4*>(sSS + c4 * 4) = *reinterpret_cast<const f32x4*>(p.gn_scale + (size_t)n * Cin + c4 * 4);
    *reinterpret_cast<f32x4*>(sSS + Cin + c4 * 4) = *reinterpret_cast<const f32x4*>(p.gn_shift + (size_t)n * Cin + c4 * 4);
  }
  auto prefetch_item = [&](int i, int kc) {
    const int cbase = kc * 16;
    const float* base;
    int Cs, cc;
    if (cbase < p.C0) { base = p.x0; Cs = p.C0; cc = cbase + q * 4; }
    else { base = p.x1; Cs = p.C1; cc = cbase - p.C0 + q * 4; }
    // (hand-placed s_waitcnt vmcnt over inline-asm loads was tried here: no faster -- the phases are not waiting for memory --
    // and wrong at B = 16, 256 x 256: the compiler moves asm outputs before the data has landed.  Plain loads, counted by the compiler.)
    rin[i] = *reinterpret_cast<const f32x4*>(base + (size_t)(in_pix[i] < 0 ? 0 : in_pix[i]) * Cs + cc);   // padding reads pixel 0, zeroed below
  };
  f32x4 ssc, ssh;
  auto load_ss = [&](int kc) {
    ssc = *reinterpret_cast<const f32x4*>(sSS + kc * 16 + q * 4);
    ssh = *reinterpret_cast<const f32x4*>(sSS + Cin + kc * 16 + q * 4);
  };
  auto stage_item = [&](int i, int kc, unsigned char* buf) {
    (void)kc;
    f32x4 v = rin[i] * ssc + ssh;
    v.x = silu_w(v.x); v.y = silu_w(v.y); v.z = silu_w(v.z); v.w = silu_w(v.w);
    const float lim = in_pix[i] >= 0 ? 16376.f : 0.f;   // zero padding of the ACTIVATED tensor; |V| <= 4 max|a| stays inside f16
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -lim, lim);
    *reinterpret_cast<f32x4*>(buf + raw_off[i]) = v;
  };

  // ---- transformed-weight fragments: [cot][kc][wave][nu][plane][lane] x 16 B ----
  const u32x4* wq = reinterpret_cast<const u32x4*>(p.wq) + ((size_t)cot * nk * 8 + wave) * (4 * 2 * 64) + lane;
  u32x4 Bf[4][2];
  auto load_b = [&](int kc, int nu) {
    const u32x4* src = wq + (size_t)kc * (8 * 4 * 2 * 64) + nu * (2 * 64);
    Bf[nu][0] = src[0];
    Bf[nu][1] = src[64];
  };

  // ---- input transform roles: lane = {ty[1:0], cq[0], tx[2:0]}, wave = {xi_t[1:0], cq[1]}; one item = (tile of a 32-tile half,
  // 4 channels, position row xi_t): 2 halo rows x 4 columns in, 4 positions out ----
  const int t_tx = lane & 7, t_ty = lane >> 4, t_cq = ((lane >> 3) & 1) | ((wave & 1) << 1);
  const int xi_t = wave >> 1;
  // row combination of B^T d as ONE fma, R = a + s b: (a, b, s) = (d0, d2, -1) (d1, d2, +1) (d2, d1, -1) (d1, d3, -1) for xi_t = 0..3
  const int ra_off = (xi_t == 0 ? 0 : (xi_t == 2 ? 2 : 1)) * W2_RAWROW, rb_off = (xi_t == 3 ? 3 : (xi_t == 2 ? 1 : 2)) * W2_RAWROW;
  const float t_s = xi_t == 1 ? 1.f : -1.f;
  const int t_rd = (2 * t_ty) * W2_RAWROW + (2 * t_tx) * W_RAWB + t_cq * 16;                       // + tb * 8 rows
  const int t_wr = ((xi_t * 4) * W_TT + t_ty * 8 + t_tx) * W_ROWB + t_cq * 8;                      // + tb * 32 tiles
  // One item in two steps, so that every LDS read of a phase's transform / staging part goes out before anything waits: ONE LDS
  // round trip per part (column by column the compiler waited out three, plus one per staged item for scale / shift).
  f32x4 ta[4], tb_[4];
  auto t_load = [&](int tb, const unsigned char* raw) {
    const unsigned char* ra = raw + t_rd + tb * (8 * W2_RAWROW) + ra_off;
    const unsigned char* rb = raw + t_rd + tb * (8 * W2_RAWROW) + rb_off;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ta[c] = *reinterpret_cast<const f32x4*>(ra + c * W_RAWB);
      tb_[c] = *reinterpret_cast<const f32x4*>(rb + c * W_RAWB);
    }
  };
  auto t_compute = [&](int tb) {
    unsigned char* dst = sV + t_wr + tb * (32 * W_ROWB);
    const f32x4 R0 = tb_[0] * t_s + ta[0], R2 = tb_[2] * t_s + ta[2];
    split_store2(dst + 0 * (W_TT * W_ROWB), R0 - R2);
    const f32x4 R1 = tb_[1] * t_s + ta[1];
    split_store2(dst + 1 * (W_TT * W_ROWB), R1 + R2);
    split_store2(dst + 2 * (W_TT * W_ROWB), R2 - R1);
    const f32x4 R3 = tb_[3] * t_s + ta[3];
    split_store2(dst + 3 * (W_TT * W_ROWB), R1 - R3);
  };
  auto transform = [&](int tb, const unsigned char* raw) {
    t_load(tb, raw);
    t_compute(tb);
  };

  // ---- MFMA roles: position row xi = wave & 3, cout half = wave >> 2 (also the stagger group) ----
  const int xi = wave & 3, chalf = wave >> 2;
  const int r31 = lane & 31, kh = lane >> 5;
  const unsigned char* abase = sV + ((xi * 4) * W_TT + r31) * W_ROWB + 16 * kh;

  f32x16 acc[4][2];
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nu][tb][i] = 0.f;

  // 12 MFMAs on tile block tb of the current chunk: all eight A-fragment reads go out first (the compiler otherwise re-uses one
  // register quad and waits out the LDS latency in front of every MFMA group); RELOAD: the next chunk's weight fragments into
  // the same registers right after their last use -- unconditionally (the last chunk re-reads its own), so that every
  // vector-memory operation of the loop sits in straight-line code and the waits the compiler inserts are counted, not vmcnt(0)
  u32x4 Af[4][2] = {};
  auto mfma_part = [&](auto tb_tag, auto reload_tag, int kc_next) {
    constexpr int tb = decltype(tb_tag)::value;
    constexpr bool RELOAD = decltype(reload_tag)::value;
#ifndef WINO_KO_A
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
        Af[nu][pl] = *reinterpret_cast<const u32x4*>(abase + (nu * W_TT + tb * 32) * W_ROWB + 32 * pl);
#endif
    W2_FSTAMP();      // A reads issued
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const u32x4 ahi = Af[nu][0], alo = Af[nu][1];
#ifndef WINO_KO_M
      acc[nu][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, alo), __builtin_bit_cast(h8, Bf[nu][0]), acc[nu][tb], 0, 0, 0);
      acc[nu][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ahi), __builtin_bit_cast(h8, Bf[nu][1]), acc[nu][tb], 0, 0, 0);
      acc[nu][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ahi), __builtin_bit_cast(h8, Bf[nu][0]), acc[nu][tb], 0, 0, 0);
#else
      WINO_SINK4(ahi); WINO_SINK4(alo); WINO_SINK4(Bf[nu][0]); WINO_SINK4(Bf[nu][1]);
#endif
#ifndef WINO_KO_B
      if (RELOAD) load_b(kc_next, nu);
#endif
    }
  };
  using TB0 = std::integral_constant<int, 0>;
  using TB1 = std::integral_constant<int, 1>;

  // ---- prologue: chunk 0 staged whole, its tile block 0 transformed; chunk 1 fetched, its item 0 staged.  The vector-memory
  // operations go out in the order the loop keeps (items 1, 2 | weight fragments | item 0), so that the wait counts the compiler
  // derives at the loop head are those of the steady state ----
#pragma unroll
  for (int i = 0; i < W_NIN; ++i) prefetch_item(i, 0);
  __syncthreads();          // scale / shift are in LDS
  load_ss(0);
#pragma unroll
  for (int i = 0; i < W_NIN; ++i) stage_item(i, 0, sRaw0);
  {
    const int k1 = nk > 1 ? 1 : 0, k2 = nk > 2 ? 2 : nk - 1;
    prefetch_item(0, k1);
    prefetch_item(1, k1);
    prefetch_item(2, k1);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) load_b(0, nu);
    __syncthreads();
    load_ss(k1);
    transform(0, sRaw0);
    stage_item(0, k1, sRaw1);
    prefetch_item(0, k2);
  }
  __syncthreads();
  W2_STAMP();      // [1] prologue done

  // The loop body is branch-free: past the end, chunk indices are clamped (the tail re-stages and re-fetches the last chunk into
  // buffers nobody reads any more), and every vector-memory operation sits in straight-line code, so the waits the compiler
  // inserts are counted (vmcnt completes in issue order: a wait for a weight fragment must not drag a younger HBM fetch along).
  // Two copies of the loop, one per wave group: the MFMA part and the transform / staging part of a phase in opposite order.
  auto run_loop = [&](auto first_tag) {
    // MODE 1 / 0: this copy of the loop runs the transform part first / last; MODE 2: one copy of the loop for both groups, the
    // transform part guarded in front of and behind the MFMA part
    constexpr int MODE = (int)decltype(first_tag)::value;
    for (int kc = 0; kc < nk; ++kc) {
      unsigned char* cur = (kc & 1) ? sRaw1 : sRaw0;
      unsigned char* nxt = (kc & 1) ? sRaw0 : sRaw1;
      const int kc1 = kc + 1 < nk ? kc + 1 : nk - 1, kc2 = kc + 2 < nk ? kc + 2 : nk - 1, kc3 = kc + 3 < nk ? kc + 3 : nk - 1;
      // ---- phase A: MFMAs on tile block 0 | tile block 1 transformed, items 1, 2 of the next chunk staged ----
      auto t_a = [&]() {
#ifndef WINO_KO_T
        t_load(1, cur);
#endif
#ifndef WINO_KO_STAGE
        load_ss(kc1);
#endif
        W2_FSTAMP();      // LDS reads issued
        __builtin_amdgcn_sched_barrier(0);
#ifndef WINO_KO_T
        t_compute(1);
#endif
        W2_FSTAMP();      // transform written
        __builtin_amdgcn_sched_barrier(0);
#ifndef WINO_KO_STAGE
        stage_item(1, kc1, nxt);
        W2_FSTAMP();
        __builtin_amdgcn_sched_barrier(0);
        stage_item(2, kc1, nxt);
#endif
      };
      if (MODE == 1 || (MODE == 2 && chalf == 0)) t_a();
      W2_STAMP();
      __builtin_amdgcn_sched_barrier(0);          // the two parts stay apart in a wave's own stream (register pressure)
      mfma_part(TB0{}, std::false_type{}, 0);
      W2_STAMP();
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 0 || (MODE == 2 && chalf != 0)) t_a();
#ifndef WINO_KO_STAGE
      prefetch_item(1, kc2);
      prefetch_item(2, kc2);
#endif
      W2_STAMP();
      __syncthreads();
      W2_STAMP();
      // ---- phase B: MFMAs on tile block 1 | tile block 0 of the next chunk transformed, item 0 of the one after staged ----
      auto t_b = [&](auto after_reload) {
#ifndef WINO_KO_T
        t_load(0, nxt);
#endif
#ifndef WINO_KO_STAGE
        load_ss(kc2);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifndef WINO_KO_T
        t_compute(0);
#endif
#ifndef WINO_KO_STAGE
        (void)after_reload;
        stage_item(0, kc2, cur);
#endif
      };
      if (MODE == 1 || (MODE == 2 && chalf == 0)) t_b(std::false_type{});
      W2_STAMP();
      __builtin_amdgcn_sched_barrier(0);
      mfma_part(TB1{}, std::true_type{}, kc1);
      W2_STAMP();
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 0 || (MODE == 2 && chalf != 0)) t_b(std::true_type{});
#ifndef WINO_KO_STAGE
      prefetch_item(0, kc3);                      // after this phase's weight-fragment loads (see above)
#endif
      W2_STAMP();
      __syncthreads();
      W2_STAMP();
    }
  };
#if defined(WINO2_TWO_LOOPS)
  if (chalf == 0) run_loop(std::integral_constant<int, 1>{});
  else run_loop(std::integral_constant<int, 0>{});
#elif !defined(WINO2_INTERLEAVED)
  run_loop(std::integral_constant<int, 2>{});      // the form used: one loop, the transform part guarded in front of / behind the MFMA part
#else
  // Third shape of the loop (-DWINO2_INTERLEAVED; measured 3 % SLOWER than the guarded form, 10.13 vs 9.80 ms for the 44 launches of
  // a B=16 forward, insensitive to the VALU count per gap: kept for reference only).  Both waves of a SIMD run the SAME stream, and
  // inside it every MFMA is followed by its share of the phase's VALU work.  The timeline of the guarded form (profiles/r03_wino_phase_timeline.txt) shows why the stagger
  // is not enough: the transform / staging part is ~3x longer than the MFMA part, so for most of a phase BOTH waves of a SIMD are in
  // it and the matrix pipe idles -- the loop is VALU-issue-bound (~140 instructions per wave and phase), and an MFMA costs the
  // issuing wave only 8 of its 32 cycles.  One basic block per phase (branch-free: lanes without a third halo item stage into a
  // dummy slot); sched_group_barrier asks for 1 MFMA : W2_IL_VALU VALU per gap, with the LDS reads up front and the writes trailing.
#ifndef W2_IL_VALU
#define W2_IL_VALU 11
#endif
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);            // LDS reads run a gap or two ahead of their use
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);            // 1 MFMA
      __builtin_amdgcn_sched_group_barrier(0x002, W2_IL_VALU, 0);   // its share of the VALU work
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);            // an LDS write when one is ready
    }
  };
  u32x4 Ai[2][2];   // A fragments of position nu in slot nu & 1: read one position ahead
  auto read_a = [&](int nu, int tb) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
      Ai[nu & 1][pl] = *reinterpret_cast<const u32x4*>(abase + (nu * W_TT + tb * 32) * W_ROWB + 32 * pl);
  };
  auto mfma3 = [&](int nu, auto tb_tag) {
    constexpr int tb = decltype(tb_tag)::value;
    const u32x4 ahi = Ai[nu & 1][0], alo = Ai[nu & 1][1];
    acc[nu][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, alo), __builtin_bit_cast(h8, Bf[nu][0]), acc[nu][tb], 0, 0, 0);
    acc[nu][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ahi), __builtin_bit_cast(h8, Bf[nu][1]), acc[nu][tb], 0, 0, 0);
    acc[nu][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ahi), __builtin_bit_cast(h8, Bf[nu][0]), acc[nu][tb], 0, 0, 0);
  };
  // one phase in the order its values are wanted: A(0), columns 0 and 2 -> V0 | A(1), column 1 -> V1, V2 | A(2), column 3 -> V3 |
  // A(3), staging; the MFMAs of position nu sit between the reads of nu + 1 and the VALU work that follows
  auto phase = [&](auto tb_tag, int ttb, const unsigned char* traw, int kss, auto second_stage) {
    constexpr int tb = decltype(tb_tag)::value;
    const unsigned char* ra = traw + t_rd + ttb * (8 * W2_RAWROW) + ra_off;
    const unsigned char* rb = traw + t_rd + ttb * (8 * W2_RAWROW) + rb_off;
    unsigned char* dst = sV + t_wr + ttb * (32 * W_ROWB);
    auto col = [&](int c) -> f32x4 {
      const f32x4 a = *reinterpret_cast<const f32x4*>(ra + c * W_RAWB);
      const f32x4 b = *reinterpret_cast<const f32x4*>(rb + c * W_RAWB);
      return b * t_s + a;
    };
    read_a(0, tb);
    const f32x4 R0 = col(0), R2 = col(2);
    load_ss(kss);
    read_a(1, tb);
    mfma3(0, tb_tag);
    split_store2(dst + 0 * (W_TT * W_ROWB), R0 - R2);
    const f32x4 R1 = col(1);
    read_a(2, tb);
    mfma3(1, tb_tag);
    split_store2(dst + 1 * (W_TT * W_ROWB), R1 + R2);
    split_store2(dst + 2 * (W_TT * W_ROWB), R2 - R1);
    const f32x4 R3 = col(3);
    read_a(3, tb);
    mfma3(2, tb_tag);
    split_store2(dst + 3 * (W_TT * W_ROWB), R1 - R3);
    second_stage(0);
    mfma3(3, tb_tag);
    second_stage(1);
    interleave();
  };
  for (int kc = 0; kc < nk; ++kc) {
    unsigned char* cur = (kc & 1) ? sRaw1 : sRaw0;
    unsigned char* nxt = (kc & 1) ? sRaw0 : sRaw1;
    const int kc1 = kc + 1 < nk ? kc + 1 : nk - 1, kc2 = kc + 2 < nk ? kc + 2 : nk - 1, kc3 = kc + 3 < nk ? kc + 3 : nk - 1;
    // ---- phase A: 12 MFMAs on tile block 0 | tile block 1 transformed, items 1, 2 of the next chunk staged ----
    phase(TB0{}, 1, cur, kc1, [&](int part) { stage_item(part ? 2 : 1, kc1, nxt); });
    __builtin_amdgcn_sched_barrier(0);
    prefetch_item(1, kc2);
    prefetch_item(2, kc2);
    __syncthreads();
    // ---- phase B: 12 MFMAs on tile block 1 | tile block 0 of the next chunk transformed, item 0 of the one after staged ----
    phase(TB1{}, 0, nxt, kc2, [&](int part) { if (part == 0) stage_item(0, kc2, cur); });
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) load_b(kc1, nu);    // same registers, next chunk (the last chunk re-reads its own)
    prefetch_item(0, kc3);
    __syncthreads();
  }
#endif

  // ---- epilogue (as the first form): fold the position row over nu, rows meet in LDS ----
  {
    float* z = reinterpret_cast<float*>(smem_w);
    const int cz = chalf * 32 + r31;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int tile = tb * 32 + (i & 3) + 8 * (i >> 2) + 4 * kh;
        const float m0 = acc[0][tb][i], m1 = acc[1][tb][i], m2 = acc[2][tb][i], m3 = acc[3][tb][i];
        z[(((xi * 2 + 0) * W_TT + tile) * W_ZROWB >> 2) + cz] = m0 + m1 + m2;
        z[(((xi * 2 + 1) * W_TT + tile) * W_ZROWB >> 2) + cz] = m1 - m2 - m3;
      }
  }
  W2_STAMP();      // Z written
  __syncthreads();
  W2_STAMP();
  const float winv = p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale;
  const int cqo = tid & 15, pp0 = tid >> 4;
  const int co = co0 + cqo * 4;
  f32x4 add = *reinterpret_cast<const f32x4*>(p.bias + co);
  if (p.temb) add += *reinterpret_cast<const f32x4*>(p.temb + (size_t)n * p.temb_stride + p.temb_off + co);
  f32x4 rv[8], yv[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pp = pp0 + it * 32, py = pp >> 4, px = pp & 15;
    const size_t o = ((size_t)(n * p.Hout + oy0 + py) * p.Wout + ox0 + px) * p.Cout + co;
    rv[it] = p.res ? *reinterpret_cast<const f32x4*>(p.res + o) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pp = pp0 + it * 32, py = pp >> 4, px = pp & 15;
    const int tile = (py >> 1) * 8 + (px >> 1), i = py & 1, j = px & 1;
    const unsigned char* zb = smem_w + ((size_t)j * W_TT + tile) * W_ZROWB + cqo * 16;
    const f32x4 z0 = *reinterpret_cast<const f32x4*>(zb + (size_t)(0 + i) * 2 * W_TT * W_ZROWB);
    const f32x4 z1 = *reinterpret_cast<const f32x4*>(zb + (size_t)(1 + i) * 2 * W_TT * W_ZROWB);
    const f32x4 z2 = *reinterpret_cast<const f32x4*>(zb + (size_t)(2 + i) * 2 * W_TT * W_ZROWB);
    const f32x4 y = i == 0 ? (z0 + z1) + z2 : (z0 - z1) - z2;
    yv[it] = y * winv + add + rv[it];
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int pp = pp0 + it * 32, py = pp >> 4, px = pp & 15;
    const size_t o = ((size_t)(n * p.Hout + oy0 + py) * p.Wout + ox0 + px) * p.Cout + co;
    *reinterpret_cast<f32x4*>(p.out + o) = yv[it];
    s1 += yv[it];
    s2 += yv[it] * yv[it];
  }
  if (p.part_out) {
    // lanes l, l ^ 16, l ^ 32 of a wave hold the same cout quad (4 pixel groups per wave): fold them in a fixed order, then the 8 waves
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s1[e] += __shfl_xor(s1[e], 16, 64); s2[e] += __shfl_xor(s2[e], 16, 64);
      s1[e] += __shfl_xor(s1[e], 32, 64); s2[e] += __shfl_xor(s2[e], 32, 64);
    }
    __syncthreads();
    f32x4* sred = reinterpret_cast<f32x4*>(smem_w);    // [8 waves][16 quads][2]
    if (lane < 16) {
      sred[(wave * 16 + lane) * 2 + 0] = s1;
      sred[(wave * 16 + lane) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 16) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = a;
#pragma unroll
      for (int g = 0; g < 8; ++g) { a += sred[(g * 16 + tid) * 2 + 0]; b += sred[(g * 16 + tid) * 2 + 1]; }
      float* dst = p.part_out + (((size_t)n * ntile + tl) * p.Cout + co0 + tid * 4) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { dst[2 * e] = a[e]; dst[2 * e + 1] = b[e]; }
    }
  }
  W2_STAMP();      // end
}



// Which launches take the Winograd form: stride-1 3x3, f16x3, 16-pixel-aligned maps, whole 64-cout blocks, 16-aligned concat
// halves, and a grid that fills the chip (small grids keep the direct kernel with its K split).
bool conv_wino_ok(ConvKind kind, int prec, const ConvParams& p, bool has_rider) {
  if (!g_tun.wino || kind != CONV3_S1 || prec != PREC_F16X3) return false;
  // Where the form pays (same-box per-layer timings at B = 16, profiles/r03_wino_per_layer.txt): the kernel is LDS-bound at about
  // the direct kernel's speed, so it wins only where the direct one is at its worst -- the 32 x 32 maps (-18 %) and the widest
  // concatenated inputs (-3 .. -7 %) -- and never where it would cost a ResnetBlock its res_conv rider (the 1x1 comes back as a
  // launch of its own).  Since the 16x16x32 direct form (fdsr_conv_k32.hip) the wide-input launches are faster direct again
  // (wino_wide_cin, default off); the 32 x 32 maps, whose 2-row tiles that form does not cover, keep the Winograd kernel.
  // Debug option wino_all = 1 lifts the rule (tests run every layer through the form).
  if (!g_tun.wino_all) {
    const bool small_map = (long)p.Hout * p.Wout <= 1024;
    if (!(small_map || (!has_rider && p.C0 + p.C1 >= g_tun.wino_wide_cin))) return false;
  }
  if (p.xr0 || p.drop_mask || p.ksplit > 1 || p.gn_plain) return false;   // (raw inputs -- no GroupNorm in front -- are turned away by the engine: the kernel has no range check)
  if ((p.Hout & 15) || (p.Wout & 15) || p.Hin != p.Hout || p.Win != p.Wout) return false;
  if ((p.Cout & 63) || (p.C0 & 15) || (p.C1 & 15) || p.C0 + p.C1 < 16 || p.C0 + p.C1 > 1024) return false;
  const long wgs = (long)p.N * (p.Hout >> 4) * (p.Wout >> 4) * (p.Cout >> 6);
  return wgs >= g_tun.wino_min_wgs;
}

hipError_t launch_conv_wino_h(const ConvParams& p, hipStream_t s, int* tiles) {
  const int ntile = (p.Hout >> 4) * (p.Wout >> 4);
  if (tiles) *tiles = ntile;
  ConvParams q = p;
  q.Cin_pad = p.C0 + p.C1;     // 16-aligned halves: no channel padding in this form
  q.Cout_pad = p.Cout;
  const int nwg = p.N * ntile * (p.Cout >> 6);
  hipLaunchKernelGGL(conv_wino2_h_kernel, dim3(nwg), dim3(512), (size_t)W2_LDS, s, q);
  return hipGetLastError();
}

#ifdef WINO_STAMPS
extern "C" int fdsr_diag_wino_stamps(unsigned long long* dst, size_t count) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_w2_stamps), count * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

hipError_t kernels_wino_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino2_h_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace fdsr
