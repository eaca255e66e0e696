// The two convolutions at the ends of the UNet, in the 16-bit precision modes (gfx950).  Both move a 64-channel full-resolution
// tensor for almost no arithmetic, so they are priced on HBM bytes, and the general implicit-GEMM kernels served them badly:
//
//   conv_in8_kernel   downs.0 (reference unet.py:243: Conv2d(in_channel = 6, inner, 3, padding 1)) on the packed NHWC-8 sampler
//                     state [cond | x_t | 0 0].  It ran on the exact-fp32 MFMA kernel in every mode (129 us at B = 16 = 2.1 TB/s
//                     of output; 578 us at B = 64 in bf16).  Here: K = 9 taps x 8 channels = 72 (padded to 96 = three 16x16x32
//                     steps), the im2col row of a pixel gathered straight from L1 / L2 by the MFMA's own operand lanes (lane (pixel,
//                     k group g) reads the 32 bytes of tap 4 i + g) -- no LDS on the data path; the 64 x 72 weights live in
//                     registers, split / scaled once per workgroup of a persistent grid (two workgroups per CU); f16x3 = three MFMAs per product like every
//                     other conv of that mode, bf16 = one.  Epilogue: bias, 16-byte stores, the GroupNorm partials of the next Block.
//   conv_out3_kernel  final_conv (unet.py:293: Block(pre_channel, out_channel = 3): GroupNorm -> Swish -> Conv3x3) -- it padded its
//                     3 output channels to a 32-wide MFMA tile (201 us at B = 16 to read 268 MB = 1.3 TB/s, 12.6 % MFMA busy).
//                     Here in "scatter" form: every input pixel is read ONCE (halo 1.2x) by the four lanes of its MFMA column, activated,
//                     and its 27 partial dot products (3 couts x 9 taps, one 64 x 27 matrix product per 16 pixels: 4 MFMAs, 12 in f16x3)
//                     go to LDS; every output pixel then adds its 9 partials per cout.  (A first version did those dot products as
//                     fp32 FMAs against LDS-broadcast weights: 2 400 VALU per input pixel, no faster than the padded MFMA tile.)
// Arithmetic: both round like the other convs of their mode (f16x3: three MFMAs per product, fp32 accumulate).
// Reductions (GroupNorm partials) in a fixed order: reruns are bitwise identical.
#include "fdsr_kernels.h"
#include "fdsr_act_io.h"

namespace fdsr {

typedef float t_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 t_h8 __attribute__((ext_vector_type(8)));
typedef __bf16 t_b8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ float silu_t(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// 8 fp32 -> the MFMA operand planes: f16x3: hi = f16(v), lo = f16(v - hi); bf16: one plane
template <int PREC>
__device__ __forceinline__ void to_planes(const float (&v)[8], uint4 (&pl)[PREC == PREC_F16X3 ? 2 : 1]) {
  if (PREC == PREC_F16) {          // one f16 plane (saturating)
    t_h8 hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) hi[j] = (_Float16)__builtin_amdgcn_fmed3f(v[j], -65504.f, 65504.f);
    pl[0] = __builtin_bit_cast(uint4, hi);
  } else if (PREC == PREC_F16X3) {
    t_h8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float c = __builtin_amdgcn_fmed3f(v[j], -65504.f, 65504.f);
      hi[j] = (_Float16)c;
      lo[j] = (_Float16)(c - (float)hi[j]);
    }
    pl[0] = __builtin_bit_cast(uint4, hi);
    pl[PREC == PREC_F16X3 ? 1 : 0] = __builtin_bit_cast(uint4, lo);
  } else {
    t_b8 b;
#pragma unroll
    for (int j = 0; j < 8; ++j) b[j] = (__bf16)v[j];
    pl[0] = __builtin_bit_cast(uint4, b);
  }
}

constexpr int IN_TH = 8, IN_TW = 32;     // workgroup tile of the input conv: 4 waves x 2 rows x 32 pixels

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
// downs.0: x [N,H,W,8] fp32 (6 real channels) -> out [N,H,W,Cout] (fp32, or bf16 in bf16 mode), Cout = 16 NCB <= 64.
// wm: the fp32 master copy of the weight in checkpoint layout [Cout][Cin][3][3].
// ---------------------------------------------------------------------------------------------------------------------------
template <int PREC, int NCB>
__global__ void __launch_bounds__(256, 2) conv_in8_kernel(const ConvParams p, const float* __restrict__ wm, int Cin, int ntiles) {
  constexpr int NP = PREC == PREC_F16X3 ? 2 : 1;
  // (the weight fragments live in registers: 96 VGPRs in f16x3 at 64 couts.  A form with them in LDS and three / four workgroups per CU
  // was tried: the compiler hoists the fragment reads of a whole pixel group and spills 79 - 148 VGPRs under the smaller caps)
  __shared__ __attribute__((aligned(16))) unsigned char sraw[4 * 16 * NCB * 2 * 4];     // the statistics exchange [4 waves][Cout][2]
  __shared__ float samax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c15 = lane & 15, g = lane >> 4;
  const int Cout = 16 * NCB;

  // ---- weights: lane (c15 = cout within its 16-block, g = k group) of fragment (cb, i) holds the 8 channels of tap 4 i + g of output
  // channel 16 cb + c15, scaled by a power of two so that the lo plane stays a normal f16 number ----
  float amax = 0.f;
  {   // Cout * Cin * 9 <= 64 * 8 * 9 values: 18 per thread, every load in flight before the first max (indices past the end re-read the last value)
    const int nw = Cout * Cin * 9;
    float wv[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) wv[k] = wm[min(tid + 256 * k, nw - 1)];
#pragma unroll
    for (int k = 0; k < 18; ++k) amax = fmaxf(amax, fabsf(wv[k]));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  if (lane == 0) samax[wave] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(samax[0], samax[1]), fmaxf(samax[2], samax[3]));
  float wscale = 1.0f, winv = 1.0f;
  if (PREC == PREC_F16X3) {
    int e = 12;
    if (amax > 0.f) e = min(12, (int)floorf(log2f(32768.0f / amax)));
    wscale = ldexpf(1.0f, e);
    winv = ldexpf(1.0f, -e);
  }
  uint4 Wf[NCB][3][NP];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int tap = 4 * i + g, co = 16 * cb + c15;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // (unconditional on a clamped index: a load under `j < Cin` waits for its own round trip)
        const float w = wm[((size_t)co * Cin + min(j, Cin - 1)) * 9 + min(tap, 8)];
        v[j] = (tap < 9 && j < Cin) ? w * wscale : 0.f;
      }
      to_planes<PREC>(v, Wf[cb][i]);
    }
  t_f32x4 bias4[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[cb][r] = p.bias[16 * cb + 4 * g + r];
  __syncthreads();

  const int tilesX = (p.Wout + IN_TW - 1) / IN_TW, tilesY = (p.Hout + IN_TH - 1) / IN_TH;
  const int per_img = tilesX * tilesY;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / per_img, tt = tile % per_img;
    const int ty = tt / tilesX, tx = tt % tilesX;
    const int oy0 = ty * IN_TH + 2 * wave, ox0 = tx * IN_TW;
    t_f32x4 s1[NCB], s2[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) s1[cb] = s2[cb] = t_f32x4{0.f, 0.f, 0.f, 0.f};
    // the six 32-byte gathers of two pixel groups (one row) first, then group by group: convert, multiply, store
    for (int half = 0; half < 2; ++half) {
    t_f32x4 raw[2][3][2];
    bool okm[2][3];
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      const int oy = oy0 + half, ox = ox0 + 16 * gq + c15;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int tap = 4 * i + g;
        const int iy = oy + tap / 3 - 1, ix = ox + tap % 3 - 1;
        const bool ok = tap < 9 && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        const t_f32x4* src = reinterpret_cast<const t_f32x4*>(p.x0 + ((size_t)(n * p.Hin + (ok ? iy : 0)) * p.Win + (ok ? ix : 0)) * 8);
        raw[gq][i][0] = src[0];
        raw[gq][i][1] = src[1];
        okm[gq][i] = ok;
      }
    }
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {               // the two 16-pixel groups of row `half`
      const int grp = gq;
      const int oy = oy0 + half, ox = ox0 + 16 * gq + c15;
      uint4 Xf[3][NP];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = okm[grp][i] ? raw[grp][i][0][j] : 0.f; v[4 + j] = okm[grp][i] ? raw[grp][i][1][j] : 0.f; }
        if (PREC == PREC_F16X3 && p.sat_flag) { sat_check(p.sat_flag, raw[grp][i][0], 65504.f); sat_check(p.sat_flag, raw[grp][i][1], 65504.f); }
        to_planes<PREC>(v, Xf[i]);
      }
      t_f32x4 acc[NCB];
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = t_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const uint4 w0 = Wf[cb][i][0];
          if (PREC == PREC_F16X3) {
            const uint4 w1 = Wf[cb][i][NP - 1];
            // small terms first: lo(x) hi(w), hi(x) lo(w), hi(x) hi(w)
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t_h8, w0), __builtin_bit_cast(t_h8, Xf[i][NP - 1]), acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t_h8, w1), __builtin_bit_cast(t_h8, Xf[i][0]), acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t_h8, w0), __builtin_bit_cast(t_h8, Xf[i][0]), acc[cb], 0, 0, 0);
          } else {
            acc[cb] = mfma16_k32<PREC>(w0, Xf[i][0], acc[cb]);
          }
        }
      // lane = pixel c15 of the group, output channels 16 cb + 4 g + 0..3
      if (oy < p.Hout && ox < p.Wout) {
        const size_t o = ((size_t)(n * p.Hout + oy) * p.Wout + ox) * Cout + 4 * g;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          const t_f32x4 v = acc[cb] * winv + bias4[cb];
          if (p.out_bf16) {
            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(p.out) + o + 16 * cb) = PREC == PREC_F16 ? pack4_16<PREC_F16>(v) : pack4_16<PREC_BF16>(v);
          } else {
            *reinterpret_cast<t_f32x4*>(p.out + o + 16 * cb) = v;
          }
          s1[cb] += v;
          s2[cb] += v * v;
        }
      }
    }
    }   // half
    if (p.part_out) {
      // per-tile (sum, sumsq) per channel, the GroupNorm statistics of the next Block: each wave folds its 2 rows x 32 pixels over the 16
      // pixel lanes by a fixed xor butterfly, the four waves meet in LDS and are added in order (one partial tile per workgroup tile:
      // a partial per WAVE made gn_finalize read 1 024 tiles per image)
      float* sst = reinterpret_cast<float*>(sraw);             // [4 waves][Cout][2]; the weight fragments live in registers by now
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a = s1[cb][r], b = s2[cb][r];
#pragma unroll
          for (int off = 8; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
          if (c15 == 0) {
            sst[(wave * Cout + 16 * cb + 4 * g + r) * 2] = a;
            sst[(wave * Cout + 16 * cb + 4 * g + r) * 2 + 1] = b;
          }
        }
      __syncthreads();
      if (tid < 2 * Cout) {
        const float v = ((sst[tid] + sst[2 * Cout + tid]) + sst[4 * Cout + tid]) + sst[6 * Cout + tid];
        p.part_out[((size_t)n * per_img + tt) * Cout * 2 + tid] = v;
      }
      __syncthreads();                                         // (before the next tile's waves write their sums)
    }
  }
}

bool conv_in8_ok(ConvKind kind, int prec, const ConvParams& p, int cin_real) {
  if (!g_tun.tail || kind != CONV3_S1 || (prec != PREC_F16X3 && !prec_is16(prec))) return false;
  if (p.C0 != 8 || p.C1 != 0 || cin_real > 8 || p.gn_scale || p.res || p.temb || p.xr0 || p.drop_mask || p.ksplit > 1) return false;
  return p.Cout == 16 || p.Cout == 32 || p.Cout == 48 || p.Cout == 64;
}

// [f16x3 | bf16][couts / 16 - 1]: workgroups of conv_in8_kernel resident on the whole device.  Queried once (kernels_tail_init) on the
// device that is current then and used for every device of the process: one process drives one GPU here, and the GPUs of a node are
// the same part.  The kernel has static LDS only (the launch passes 0 dynamic bytes, as the query does).  A value that is too large only
// repeats a workgroup's weight set-up, one that is too small leaves CUs idle: performance, never correctness; the grid is clamped to the
// tile count below.
static int g_in8_resident[8];

hipError_t launch_conv_in8(int prec, const ConvParams& p, const float* wmaster, int cin_real, hipStream_t s, int* tiles) {
  const int per_img = ((p.Wout + IN_TW - 1) / IN_TW) * ((p.Hout + IN_TH - 1) / IN_TH);
  if (tiles) *tiles = per_img;
  const int ntiles = p.N * per_img;
  // persistent: as many workgroups as are resident at once (the register count of the instantiation decides: two per CU at 64 couts, three
  // or four below), so that each sets its weights up and scans the range of its input exactly once
  const int slot = (prec == PREC_F16X3 ? 0 : 4) + p.Cout / 16 - 1;
  const int resident = (slot >= 0 && slot < 8 && g_in8_resident[slot] > 0) ? g_in8_resident[slot] : 512;
  const int grid = ntiles < resident ? ntiles : resident;
  ConvParams q = p;
  q.out_bf16 = p.out_f32 ? 0 : prec_act16(prec);
#define IN8_CASE(NCB_)                                                                                                     \
  if (p.Cout == 16 * NCB_) {                                                                                               \
    if (prec == PREC_F16X3) hipLaunchKernelGGL((conv_in8_kernel<PREC_F16X3, NCB_>), dim3(grid), dim3(256), 0, s, q, wmaster, cin_real, ntiles); \
    else if (prec == PREC_F16) hipLaunchKernelGGL((conv_in8_kernel<PREC_F16, NCB_>), dim3(grid), dim3(256), 0, s, q, wmaster, cin_real, ntiles);  \
    else hipLaunchKernelGGL((conv_in8_kernel<PREC_BF16, NCB_>), dim3(grid), dim3(256), 0, s, q, wmaster, cin_real, ntiles);  \
    return hipGetLastError();                                                                                              \
  }
  IN8_CASE(1) IN8_CASE(2) IN8_CASE(3) IN8_CASE(4)
#undef IN8_CASE
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------------------
// final_conv: x [N,H,W,C] (fp32 or bf16), GroupNorm scale / shift per (image, channel), Swish, Conv3x3 C -> Cout <= 3, fp32 out.
// Workgroup = 8 rows x 32 pixels of output = 340 halo pixels = 22 groups of 16 pixels, dealt to the four waves (a 16-row tile has
// less halo but its 66 KB of partials leave two workgroups per CU: this kernel lives on loads in flight).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int OT_TH = 8, OT_TW = 32, OT_HW = OT_TW + 2, OT_NPIX = (OT_TH + 2) * OT_HW;   // 340 halo pixels: 36.7 KB of partials, four workgroups per CU
}

// The K order inside a 32-channel step is free as long as both MFMA operands agree: lane group g holds the channel quads 4 g and 16 + 4 g,
// so that each of its two 16-byte loads, over the four lanes of a pixel, covers 64 contiguous bytes (fp32; 32 in bf16).
__device__ __forceinline__ int kq(int g, int half) { return 16 * half + 4 * g; }

template <int PREC, int NKS>     // NKS = C / 32 k steps
__global__ void __launch_bounds__(256, NKS <= 2 ? 4 : 2) conv_out3_kernel(const ConvParams p, const float* __restrict__ wm, int Cin_real, int ntiles) {
  constexpr int NP = PREC == PREC_F16X3 ? 2 : 1;
  using IO = ActIO<PREC>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
  const int C = p.C0, nj = 9 * p.Cout;                       // partial sums per input pixel: j = (cout, tap) < 27, padded to 32
  float* gss = reinterpret_cast<float*>(smem_t);             // [2][C]  GroupNorm scale | shift of this image
  float* part = gss + 2 * C;                                 // [27][OT_NPIX]
  __shared__ float samax[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c15 = lane & 15, g = lane >> 4;
  const int tilesX = (p.Wout + OT_TW - 1) / OT_TW, tilesY = (p.Hout + OT_TH - 1) / OT_TH;
  // ---- weights as the MFMA's A operand: row = j (16 per block nb), k = channel; lane (c15 = j - 16 nb, g) of k step i holds the channels
  // 32 i + 8 g .. + 7 of (cout, tap) = j, scaled by a power of two (f16x3) so that the lo plane stays a normal f16 number ----
  float amax = 0.f;
  {   // Cout * Cin * 9 <= 3 * 128 * 9 values: 14 per thread, every load in flight before the first max (as conv_in8_kernel)
    const int nw = p.Cout * Cin_real * 9;
    float wv[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) wv[k] = wm[min(tid + 256 * k, nw - 1)];
#pragma unroll
    for (int k = 0; k < 14; ++k) amax = fmaxf(amax, fabsf(wv[k]));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
  if (lane == 0) samax[wave] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(samax[0], samax[1]), fmaxf(samax[2], samax[3]));
  float wscale = 1.0f, winv = 1.0f;
  if (PREC == PREC_F16X3) {
    int e = 12;
    if (amax > 0.f) e = min(12, (int)floorf(log2f(32768.0f / amax)));
    wscale = ldexpf(1.0f, e);
    winv = ldexpf(1.0f, -e);
  }
  uint4 Wf[2][NKS][NP];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int i = 0; i < NKS; ++i) {
      const int j = 16 * nb + c15, co = j / 9, tap = j % 9;
      float v[8];
#pragma unroll
      for (int e8 = 0; e8 < 8; ++e8) {
        const int ci = 32 * i + kq(g, e8 >> 2) + (e8 & 3);
        const float w = wm[((size_t)min(co, p.Cout - 1) * Cin_real + min(ci, Cin_real - 1)) * 9 + tap];   // (unconditional, clamped)
        v[e8] = (j < nj && ci < Cin_real) ? w * wscale : 0.f;
      }
      to_planes<PREC>(v, Wf[nb][i]);
    }
  const bool act = p.gn_scale != nullptr && !p.gn_plain;
  // persistent grid: the weight fragments above are set up once per workgroup, then it walks its tiles
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int tx = tile % tilesX, ty = (tile / tilesX) % tilesY, n = tile / (tilesX * tilesY);
  const int oy0 = ty * OT_TH, ox0 = tx * OT_TW;
  __syncthreads();                                           // (the previous tile's partials and scale / shift have been read)
  if (tid < C) {   // C <= 128 (conv_out3_ok); the GroupNorm'd input is a condition of this kernel
    gss[tid] = p.gn_scale[(size_t)n * C + tid];
    gss[C + tid] = p.gn_shift[(size_t)n * C + tid];
  }
  __syncthreads();
  // ---- every halo pixel once: lane (pixel c15 of its 16-pixel group, k group g) reads its 8 channels of each k step, activates them,
  // and the group's 27 partial dot products come out of 2 NKS MFMAs (x 3 in f16x3) with lane (pixel, g) holding j = 16 nb + 4 g + 0..3 ----
  constexpr int NGRP = (OT_NPIX + 15) / 16;
  typename IO::Quad rq[NKS][2], rn[NKS][2];
  bool ok = false, okn = false;
  auto fetch = [&](int grp, typename IO::Quad (&dst)[NKS][2]) __attribute__((always_inline)) {
    const int hp = grp * 16 + c15;
    const int hy = hp / OT_HW, hx = hp % OT_HW;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool o = grp < NGRP && hp < OT_NPIX && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
    const size_t base = o ? ((size_t)(n * p.Hin + iy) * p.Win + ix) * C : 0;
#pragma unroll
    for (int i = 0; i < NKS; ++i) {
      dst[i][0] = IO::load4(p.x0, base + 32 * i + kq(g, 0));
      dst[i][1] = IO::load4(p.x0, base + 32 * i + kq(g, 1));
    }
    return o;
  };
  okn = fetch(wave, rn);
  for (int grp = wave; grp < NGRP; grp += 4) {
    const int hp = grp * 16 + c15;
#pragma unroll
    for (int i = 0; i < NKS; ++i) { rq[i][0] = rn[i][0]; rq[i][1] = rn[i][1]; }
    ok = okn;
    okn = fetch(grp + 4, rn);                                // the next group's rows are in flight while this one is multiplied
    t_f32x4 acc[2] = {t_f32x4{0.f, 0.f, 0.f, 0.f}, t_f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < NKS; ++i) {
      float v[8];
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        const t_f32x4 sc = *reinterpret_cast<const t_f32x4*>(gss + 32 * i + kq(g, hq));
        const t_f32x4 sh = *reinterpret_cast<const t_f32x4*>(gss + C + 32 * i + kq(g, hq));
        t_f32x4 x = IO::widen(rq[i][hq]) * sc + sh;
        if (act) { x.x = silu_t(x.x); x.y = silu_t(x.y); x.z = silu_t(x.z); x.w = silu_t(x.w); }
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) v[4 * hq + e4] = ok ? x[e4] : 0.f;      // zero padding applies to the ACTIVATED tensor
      }
      uint4 Xf[NP];
      to_planes<PREC>(v, Xf);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        if (PREC == PREC_F16X3) {
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t_h8, Wf[nb][i][0]), __builtin_bit_cast(t_h8, Xf[NP - 1]), acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t_h8, Wf[nb][i][NP - 1]), __builtin_bit_cast(t_h8, Xf[0]), acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(t_h8, Wf[nb][i][0]), __builtin_bit_cast(t_h8, Xf[0]), acc[nb], 0, 0, 0);
        } else {
          acc[nb] = mfma16_k32<PREC>(Wf[nb][i][0], Xf[0], acc[nb]);
        }
      }
    }
    if (hp < OT_NPIX) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * nb + 4 * g + r;
          if (j < nj) part[j * OT_NPIX + hp] = acc[nb][r] * winv;
        }
    }
  }
  __syncthreads();
  // output pixel (oy0 + r, ox0 + c): its nine partials per cout sit at halo pixels (r + ky, c + kx)
  for (int o = tid; o < OT_TH * OT_TW; o += 256) {
    const int r = o / OT_TW, c = o % OT_TW;
    const int oy = oy0 + r, ox = ox0 + c;
    if (oy >= p.Hout || ox >= p.Wout) continue;
    for (int co = 0; co < p.Cout; ++co) {
      float s = p.bias[co];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) s += part[(co * 9 + tap) * OT_NPIX + (r + tap / 3) * OT_HW + c + tap % 3];
      p.out[((size_t)(n * p.Hout + oy) * p.Wout + ox) * p.Cout + co] = s;
    }
  }
  }   // tiles
}

static size_t conv_out3_lds(int C) { return (size_t)(2 * C + 27 * OT_NPIX) * sizeof(float); }

bool conv_out3_ok(ConvKind kind, int prec, const ConvParams& p) {
  if (!g_tun.tail || kind != CONV3_S1 || (prec != PREC_F16X3 && !prec_is16(prec))) return false;
  if (p.Cout < 1 || p.Cout > 3 || p.C1 != 0 || (p.C0 != 32 && p.C0 != 64 && p.C0 != 96 && p.C0 != 128)) return false;   // whole 32-channel k steps
  if (p.res || p.temb || p.xr0 || p.drop_mask || p.ksplit > 1 || p.part_out || !p.out_f32) return false;
  if (!p.gn_scale) return false;   // (a raw input would be clamped to the f16 range without the range flag: this kernel is the GroupNorm'd final conv's)
  return conv_out3_lds(p.C0) <= 80 * 1024;
}

hipError_t launch_conv_out3(int prec, const ConvParams& p, const float* wmaster, int cin_real, hipStream_t s, int* tiles) {
  const int per_img = ((p.Wout + OT_TW - 1) / OT_TW) * ((p.Hout + OT_TH - 1) / OT_TH);
  if (tiles) *tiles = per_img;
  const size_t lds = conv_out3_lds(p.C0);
  const int ntiles = p.N * per_img, grid = ntiles < 1024 ? ntiles : 1024;      // persistent: four workgroups per CU
#define OUT3_CASE(NKS_)                                                                                                                  \
  if (p.C0 == 32 * NKS_) {                                                                                                               \
    if (prec == PREC_F16X3) hipLaunchKernelGGL((conv_out3_kernel<PREC_F16X3, NKS_>), dim3(grid), dim3(256), lds, s, p, wmaster, cin_real, ntiles); \
    else if (prec == PREC_F16) hipLaunchKernelGGL((conv_out3_kernel<PREC_F16, NKS_>), dim3(grid), dim3(256), lds, s, p, wmaster, cin_real, ntiles); \
    else hipLaunchKernelGGL((conv_out3_kernel<PREC_BF16, NKS_>), dim3(grid), dim3(256), lds, s, p, wmaster, cin_real, ntiles);           \
    return hipGetLastError();                                                                                                            \
  }
  OUT3_CASE(1) OUT3_CASE(2) OUT3_CASE(3) OUT3_CASE(4)
#undef OUT3_CASE
  return hipErrorInvalidValue;
}

hipError_t kernels_tail_init() {
  hipError_t e;
#define OUT3_INIT(NKS_)                                                                                                                                        \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_out3_kernel<PREC_F16X3, NKS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)) != hipSuccess) return e; \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_out3_kernel<PREC_BF16, NKS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)) != hipSuccess) return e; \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_out3_kernel<PREC_F16, NKS_>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)) != hipSuccess) return e;
  OUT3_INIT(1) OUT3_INIT(2) OUT3_INIT(3) OUT3_INIT(4)
#undef OUT3_INIT
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
#define IN8_OCC(NCB_)                                                                                                              \
  {                                                                                                                                \
    int nb = 0;                                                                                                                    \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_in8_kernel<PREC_F16X3, NCB_>, 256, 0) == hipSuccess && nb > 0)      \
      g_in8_resident[NCB_ - 1] = nb * cus;                                                                                         \
    nb = 0;                                                                                                                        \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, conv_in8_kernel<PREC_BF16, NCB_>, 256, 0) == hipSuccess && nb > 0)       \
      g_in8_resident[4 + NCB_ - 1] = nb * cus;                                                                                     \
  }
  IN8_OCC(1) IN8_OCC(2) IN8_OCC(3) IN8_OCC(4)
#undef IN8_OCC
  return hipSuccess;
}

}  // namespace fdsr
