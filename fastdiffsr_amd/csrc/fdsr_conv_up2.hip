// Upsample(nearest x2) -> Conv3x3 (reference unet.py:66-74) in sub-pixel form, 16-bit MFMA path.
//
// With U[R][C] = src[R>>1][C>>1] (zero outside), the output pixel (2y+py, 2x+px) of the 3x3 conv
// over U only ever touches the 2x2 source neighbourhood rows {y-1+py, y+py} x cols {x-1+px, x+px}:
//   out[2y+py][2x+px] = sum_{a,b in {0,1}} W2[py][px][a][b] . src[y-1+py+a][x-1+px+b]
//   W2[py][px][a][b]  = sum_{ky in R(py,a)} sum_{kx in R(px,b)} W[ky][kx],
//   R(0,0)={0}, R(0,1)={1,2}, R(1,0)={0,1}, R(1,1)={2}
// (the zero padding of U maps onto the zero padding of src, so the identity is exact up to the fp32
// rounding of the pre-summed weights).  16 tap-products per source pixel instead of 36: 2.25x fewer
// MFMAs, and the staged halo per output pixel halves.
//
// Same structure as conv_mfma_h_kernel ("weights in registers, pixels from LDS"): a workgroup =
// 8 waves = TH x 32 SOURCE pixels x one row parity py x both column parities x 32*WN output channels;
// a wave owns 32 channels and TH/WM source rows and holds the 8 (px,a,b) weight fragments of the
// current 16-channel chunk (hi|lo) in 64 VGPRs; per source row it reads 6 A fragments (2 rows x 3
// column shifts) that feed the 8 tap-products; accumulators: [row][px].
#include "fdsr_kernels.h"
#include "fdsr_act_io.h"

namespace fdsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));

template <int TH, int WN, int PREC>
struct ConvUp2Cfg {
  static constexpr int TW = 32, KC = 16;
  static constexpr int NP = PREC == PREC_F16X3 ? 2 : 1;
  static constexpr int WNP = PREC == PREC_BF16 ? 1 : 2;   // planes per fragment of the weight arena the mode reads (PREC_F16: the f16x3 form, hi plane only)
  static constexpr int ROWB = NP * 32 + 16;
  static constexpr int HH = TH + 1, HWD = TW + 2, NPIX = HH * HWD;
  static constexpr int WM = 8 / WN, BN = 32 * WN, MB = TH / WM;
  static constexpr int RPP = 512 / 4, NIN = (NPIX + RPP - 1) / RPP;
  static constexpr int BUF_BYTES = (NPIX * ROWB + 15) / 16 * 16;
  static_assert(TH % WM == 0 && MB <= 2, "at most 2 source rows x 2 column parities per wave");
};

template <int TH, int WN, int PREC>
__global__ void __launch_bounds__(512, 2) conv_up2_h_kernel(const ConvParams p) {
  using Cfg = ConvUp2Cfg<TH, WN, PREC>;
  constexpr int TW = Cfg::TW, KC = Cfg::KC, NP = Cfg::NP, ROWB = Cfg::ROWB, HWD = Cfg::HWD, NPIX = Cfg::NPIX;
  constexpr int WM = Cfg::WM, BN = Cfg::BN, MB = Cfg::MB, RPP = Cfg::RPP, NIN = Cfg::NIN;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_u[];
  unsigned char* sBuf0 = smem_u;
  unsigned char* sBuf1 = smem_u + Cfg::BUF_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave % WN, wm = wave / WN;
  const int r31 = lane & 31, h = lane >> 5;

  // source-resolution tiling; p.Hin/Win = source dims, p.Hout/Wout = 2x
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
  const int py = pt & 1;          // the two row parities of a source tile are neighbours (shared input in L2)
  pt >>= 1;
  const int tx = pt % tilesX;
  pt /= tilesX;
  const int ty = pt % tilesY;
  const int n = pt / tilesY;
  const int oy0 = ty * TH, ox0 = tx * TW, co0 = cot * BN;

  // ---- staging: halo rows oy0-1+py .. oy0+TH-1+py, cols ox0-1 .. ox0+32 of the SOURCE ----
  const int q = tid & 3, row0 = tid >> 2;
  int in_pix[NIN];
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int pix = row0 + i * RPP;
    const int hy = pix / HWD, hx = pix % HWD;
    const int iy = oy0 - 1 + py + hy, ix = ox0 - 1 + hx;
    const bool ok = pix < NPIX && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
    in_pix[i] = ok ? (n * p.Hin + iy) * p.Win + ix : -1;
  }
  using IO = ActIO<PREC>;
  typename IO::Quad rin[NIN];
  auto prefetch = [&](int kc) {
    const size_t coff = (size_t)kc * KC + q * 4;
#pragma unroll
    for (int i = 0; i < NIN; ++i)
      rin[i] = IO::load4(p.x0, (size_t)(in_pix[i] < 0 ? 0 : in_pix[i]) * p.C0 + coff);
  };
  auto stage = [&](unsigned char* buf) {
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      if (NIN * RPP > NPIX && i == NIN - 1 && row0 + i * RPP >= NPIX) continue;
      f32x4 v = IO::widen(rin[i]);
      const float keep = in_pix[i] >= 0 ? 1.f : 0.f;
      const float lim = in_pix[i] >= 0 ? 65504.f : 0.f;
      unsigned char* dst = buf + (row0 + i * RPP) * ROWB + q * 8;
      if (PREC == PREC_F16X3) {
        if (p.sat_flag) sat_check(p.sat_flag, v, 65504.f);   // raw input (no GroupNorm in front of the upsample conv)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -lim, lim);   // padding: lim = 0
        h4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        h4 lo = {(_Float16)(v.x - (float)hi.x), (_Float16)(v.y - (float)hi.y), (_Float16)(v.z - (float)hi.z),
                 (_Float16)(v.w - (float)hi.w)};
        *reinterpret_cast<h4*>(dst) = hi;
        *reinterpret_cast<h4*>(dst + 32) = lo;
      } else {
        v = v * keep;
        *reinterpret_cast<uint2*>(dst) = stage4_16<PREC>(v);
      }
    }
  };

  // ---- weight fragments: [cot][kc][wn][py][slot = px*4 + a*2 + b][plane][lane] x 16 B ----
  const uint4* wq = reinterpret_cast<const uint4*>(p.wq);
  const int nk = p.Cin_pad / KC;
  uint4 Bf[8][NP];
  auto load_b_slot = [&](int kc, int slot) {
    const uint4* src = wq + (((((size_t)cot * nk + kc) * WN + wn) * 2 + py) * 8 + slot) * (Cfg::WNP * 64) + lane;
#pragma unroll
    for (int pl = 0; pl < NP; ++pl) Bf[slot][pl] = src[pl * 64];
  };

  int abase[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) abase[mb] = ((wm + mb * WM) * HWD + r31) * ROWB + 16 * h;

  f32x16 acc[MB][2];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int px = 0; px < 2; ++px)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][px][i] = 0.f;

#pragma unroll
  for (int sl = 0; sl < 8; ++sl) load_b_slot(0, sl);
  prefetch(0);
  stage(sBuf0);
  if (nk > 1) prefetch(1);
  __syncthreads();

  auto mma = [&](f32x16& c, const uint4 (&a)[NP], const uint4 (&b)[NP]) {
    if (PREC == PREC_F16X3) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[NP - 1]), __builtin_bit_cast(h8, b[0]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[0]), __builtin_bit_cast(h8, b[NP - 1]), c, 0, 0, 0);
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a[0]), __builtin_bit_cast(h8, b[0]), c, 0, 0, 0);
    } else {
      c = mfma32_k16<PREC>(a[0], b[0], c);
    }
  };

  for (int kc = 0; kc < nk; ++kc) {
    unsigned char* cur = (kc & 1) ? sBuf1 : sBuf0;
    unsigned char* nxt = (kc & 1) ? sBuf0 : sBuf1;
    const bool more = kc + 1 < nk;
    // per source row: 6 A fragments (halo row a in {0,1} x column shift c in {0,1,2}); (px,b) uses
    // shift px+b.  Weight slots are re-loaded for the next chunk after their last use (last row).
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      uint4 Af[2][3][NP];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
          for (int pl = 0; pl < NP; ++pl)
            Af[a][c][pl] = *reinterpret_cast<const uint4*>(cur + abase[mb] + (a * HWD + c) * ROWB + 32 * pl);
#pragma unroll
      for (int slot = 0; slot < 8; ++slot) {
        const int px = slot >> 2, a = (slot >> 1) & 1, b = slot & 1;
        mma(acc[mb][px], Af[a][px + b], Bf[slot]);
        if (mb == MB - 1 && more) load_b_slot(kc + 1, slot);
        if (mb == MB - 1 && slot == 1 && more) {
          stage(nxt);
          if (kc + 2 < nk) prefetch(kc + 2);
        }
      }
    }
    __syncthreads();
  }

  // ---- epilogue: out[(2(oy0+row)+py)][2(ox0+col)+px], + bias (+ temb), GN partials of the output ----
  const int co = co0 + wn * 32 + r31;
  const bool cok = co < p.Cout;
  const float winv = p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale;   // device value: forms re-packed after optimiser steps
  float add;
  {   // unconditional loads (clamped channel, noise shift through a 0 / 1 factor)
    const int cs = cok ? co : 0;
    const float* tembp = p.temb ? p.temb + (size_t)n * p.temb_stride + p.temb_off : p.bias;
    add = p.bias[cs] + (p.temb ? 1.f : 0.f) * tembp[cs];
  }
  float s1 = 0.f, s2 = 0.f;
  const bool interior = (oy0 + TH <= p.Hin) && (ox0 + TW <= p.Win) && (co0 + BN <= p.Cout);
  if (interior) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int Y = 2 * (oy0 + wm + mb * WM) + py;
      const size_t rowp = ((size_t)(n * p.Hout + Y) * p.Wout + 2 * (ox0 + 4 * h)) * p.Cout + co;
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float v = acc[mb][px][i] * winv + add;
          IO::store1(p.out, rowp + (size_t)(2 * ((i & 3) + 8 * (i >> 2)) + px) * p.Cout, v);   // bf16 mode: bf16 activation
          s1 += v;
          s2 += v * v;
        }
    }
  } else {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int sy = oy0 + wm + mb * WM, sx = ox0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (cok && sy < p.Hin && sx < p.Win) {
            const float v = acc[mb][px][i] * winv + add;
            IO::store1(p.out, ((size_t)(n * p.Hout + 2 * sy + py) * p.Wout + 2 * sx + px) * p.Cout + co, v);
            s1 += v;
            s2 += v * v;
          }
        }
  }
  if (p.part_out) {
    float* sp = reinterpret_cast<float*>(smem_u);   // halo buffers are free after the last barrier
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (h == 0) {
      sp[(wm * BN + wn * 32 + r31) * 2 + 0] = s1;
      sp[(wm * BN + wn * 32 + r31) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < BN && co0 + tid < p.Cout) {
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { a += sp[(w * BN + tid) * 2 + 0]; b += sp[(w * BN + tid) * 2 + 1]; }
      float* dst = p.part_out + (((size_t)n * (tilesX * tilesY * 2) + (ty * tilesX + tx) * 2 + py) * p.Cout + co0 + tid) * 2;
      dst[0] = a;
      dst[1] = b;
    }
  }
}

template <int TH, int WN, int PREC>
static hipError_t launch_up2_t(const ConvParams& p, hipStream_t s, int* tiles) {
  using Cfg = ConvUp2Cfg<TH, WN, PREC>;
  auto kfn = conv_up2_h_kernel<TH, WN, PREC>;
  const int tilesX = (p.Win + Cfg::TW - 1) / Cfg::TW, tilesY = (p.Hin + TH - 1) / TH;
  if (tiles) *tiles = tilesX * tilesY * 2;
  const int nwg = p.N * tilesX * tilesY * 2 * (p.Cout_pad / Cfg::BN);
  if (conv_up2_k32_ok(PREC, p)) return launch_conv_up2_k32(TH, WN, PREC, p, nwg, s);   // the 16x16x32 form (fdsr_conv_k32.hip): same grid
  hipLaunchKernelGGL(kfn, dim3(nwg), dim3(512), (size_t)2 * Cfg::BUF_BYTES, s, p);
  return hipGetLastError();
}

// TH = 2 * (8 / WN): two source rows per wave
hipError_t launch_conv_up2_h(int prec, const ConvParams& p, hipStream_t s, int* tiles) {
  int TH, WN;
  conv_h_config(CONV3_UP, p.Cout, &TH, &WN);
#define X(TH_, WN_)                                                                           \
  if (WN == WN_) return prec == PREC_F16X3 ? launch_up2_t<TH_, WN_, PREC_F16X3>(p, s, tiles)  \
                                           : prec == PREC_F16 ? launch_up2_t<TH_, WN_, PREC_F16>(p, s, tiles) : launch_up2_t<TH_, WN_, PREC_BF16>(p, s, tiles);
  X(8, 2) X(4, 4) X(2, 8)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace fdsr
