// Small-map form of the 256-cout stride-1 3x3 convolutions (the 32 x 32 and 64 x 64 levels of the UNet at small batches: reference
// unet.py:89-120 at 4 x inner_channel; the reference's own val loop runs B = 1, sr_mfe.py:274-284) for gfx950:
// v_mfma_f32_16x16x32_f16, fp32 accumulate, the fp32-grade f16x3 arithmetic.
//
// At B = 1 a 32 x 32 map gives the tile kernels 16 workgroups; they split K over workgroups (partial sums through HBM) and a second
// launch (splitk_reduce_kernel) adds the slices, applies the epilogue and emits the GroupNorm partial sums: 16.6 + 7.4 us per layer,
// 24 layers per forward.  Here the K split stays INSIDE a workgroup:
//
//   * workgroup = 8 waves = one tile of 2 x 32 pixels x 32 output channels (16 pixel tiles x 8 cout blocks = 128 workgroups per
//     32 x 32 image);
//   * wave k owns the 32-channel K slice(s) k, k + 8, ...: it stages ITS channels of the tile's 4 x 34 halo (GroupNorm apply + Swish
//     + hi / lo split fused, as everywhere) into a wave-private LDS image -- no barrier in the main loop -- reads its weight
//     fragments straight from the packed arena (the 16-byte permutation of conv_k32_kernel), and accumulates the whole
//     64 x 32 output tile over its slice(s): 216 MFMAs per slice;
//   * the eight partial tiles meet in LDS (over the halo images), are added in wave order (bitwise reproducible), and the epilogue --
//     bias, noise shift, residual, 16-byte coalesced stores, GroupNorm partial sums of the output -- runs once, in the same launch.
//
// Same ConvParams, same packed weights, same outputs and statistics layout ([N][tiles][C][2], tile = 2 x 32 pixels) as the other
// 16-bit kernels; launch_conv_h asks conv_small_ok() first.
#include "fdsr_kernels.h"
#include "fdsr_act_io.h"

namespace fdsr {

typedef float m_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 m_h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float silu_m(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

struct SmallCfg {
  static constexpr int NWV = 8, KC = 32, TH = 2, TW = 32, BN = 32;
  static constexpr int HH = TH + 2, HWD = TW + 2, NPIX = HH * HWD;     // 4 x 34 halo pixels
  static constexpr int ROWB = 128;                                     // LDS bytes per halo pixel: [4 x 16 B hi | 4 x 16 B lo], units XOR hx & 7
  static constexpr int HALO_BYTES = NPIX * ROWB;                       // per wave
  static constexpr int NIN = NPIX / 8;                                 // staging passes of a wave: 8 pixels x 8 channel quads per pass
  static constexpr int RED_BYTES = NWV * TH * TW * BN * 4;             // the eight partial tiles (over the halo images)
  static constexpr int LDS_BYTES = NWV * HALO_BYTES;
  static_assert(NPIX % 8 == 0 && RED_BYTES <= LDS_BYTES, "staging map / reduction overlay");
};

__global__ void __launch_bounds__(512, 2) conv_small_kernel(const ConvParams p) {
  using Cfg = SmallCfg;
  constexpr int KC = Cfg::KC, TH = Cfg::TH, TW = Cfg::TW, BN = Cfg::BN, HWD = Cfg::HWD, NPIX = Cfg::NPIX, ROWB = Cfg::ROWB, NIN = Cfg::NIN;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_m[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c15 = lane & 15, g = lane >> 4;
  const int Cin = p.C0 + p.C1;
  const int tilesX = p.Wout / TW, tilesY = p.Hout / TH, ncb = p.Cout / BN;
  int bid = blockIdx.x;
  const int cb = bid % ncb;           // the cout blocks of one pixel tile are neighbours: they read the same input
  bid /= ncb;
  const int tx = bid % tilesX;
  bid /= tilesX;
  const int ty = bid % tilesY;
  const int n = bid / tilesY;
  const int oy0 = ty * TH, ox0 = tx * TW, co0 = cb * BN;

  unsigned char* halo = smem_m + wave * Cfg::HALO_BYTES;

  // ---- staging map of a wave: lane -> (channel quad q of the slice, halo pixel (lane >> 3) + 8 i) ----
  const int q = lane & 7, row0 = lane >> 3;
  unsigned in_off[NIN];               // pixel index inside image n (clamped), bit 31: outside the image
#pragma unroll
  for (int i = 0; i < NIN; ++i) {
    const int pix = row0 + 8 * i, hy = pix / HWD, hx = pix % HWD;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
    in_off[i] = ok ? (unsigned)(iy * p.Win + ix) : 0x80000000u;
  }

  // ---- fragment addresses inside the halo image: lane -> pixel column c15 + kx (+ 16 col half), k group g; hi plane (+ 64: lo) ----
  int xoff[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int hx = c15 + kx;
    xoff[kx] = hx * ROWB + 16 * (g ^ (hx & 7));
  }

  m_f32x4 acc[4][2];                  // [pixel tile: row * 2 + column half][cout half]
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) acc[pt][ch] = m_f32x4{0.f, 0.f, 0.f, 0.f};

  const uint4* wq = reinterpret_cast<const uint4*>(p.wq);
  const int nk16 = p.Cin_pad / 16, WNA = p.Cout_pad / 32;       // the arena's cout split (BN = 32 WN, one cot for Cout <= 256)
  const int wlane = 32 * (g & 1) + c15;
  const int nsl = Cin / KC;

  for (int sl = wave; sl < nsl; sl += Cfg::NWV) {
    // ---- this wave's 32 channels of the halo: fetch everything, then GroupNorm + Swish + split into the LDS image ----
    const int cbase = sl * KC;
    const float* xb;
    int Cs, cc;
    if (cbase < p.C0) { xb = p.x0 + (size_t)n * p.Hin * p.Win * p.C0; Cs = p.C0; cc = cbase + 4 * q; }
    else { xb = p.x1 + (size_t)n * p.Hin * p.Win * p.C1; Cs = p.C1; cc = cbase - p.C0 + 4 * q; }
    const m_f32x4 gsc = *reinterpret_cast<const m_f32x4*>(p.gn_scale + (size_t)n * Cin + cbase + 4 * q);
    const m_f32x4 gsh = *reinterpret_cast<const m_f32x4*>(p.gn_shift + (size_t)n * Cin + cbase + 4 * q);
    m_f32x4 rin[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i)
      rin[i] = *reinterpret_cast<const m_f32x4*>(xb + (size_t)(in_off[i] & 0x7fffffffu) * Cs + cc);
    // the first taps' weight fragments travel meanwhile: [tap ring slot][cout half][plane]
    uint4 Wf[3][2][2];
    auto load_w = [&](int tap, int slot) __attribute__((always_inline)) {
      const uint4* src = wq + (((size_t)(2 * sl + (g >> 1)) * WNA + cb) * 9 + tap) * 128 + wlane;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) Wf[slot][ch][pl] = src[pl * 64 + 16 * ch];
    };
#pragma unroll
    for (int t = 0; t < 3; ++t) load_w(t, t);
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      m_f32x4 v = rin[i] * gsc + gsh;
      const bool ok = !(in_off[i] & 0x80000000u);
      const float lim = ok ? 65504.f : 0.f;      // f16 range clamp and zero padding in one med3
      const int pix = row0 + 8 * i, hx = pix % HWD;
      unsigned char* dst = halo + pix * ROWB + 16 * ((q >> 1) ^ (hx & 7)) + 8 * (q & 1);
      uint2 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(p.gn_plain ? v[e] : silu_m(v[e]), -lim, lim);
      typedef _Float16 h2t __attribute__((ext_vector_type(2)));
      {
        const h2t h0 = {(_Float16)v[0], (_Float16)v[1]}, h1 = {(_Float16)v[2], (_Float16)v[3]};
        hi.x = __builtin_bit_cast(unsigned, h0);
        hi.y = __builtin_bit_cast(unsigned, h1);
      }
      // lo = f16(fma(hi, -1, v)), rounded once (as fdsr_conv_k32.hip)
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo.x) : "v"(hi.x), "v"(v[0]));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo.x) : "v"(hi.x), "v"(v[1]));
      asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lo.y) : "v"(hi.y), "v"(v[2]));
      asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo.y) : "v"(hi.y), "v"(v[3]));
      *reinterpret_cast<uint2*>(dst) = hi;
      *reinterpret_cast<uint2*>(halo + ((int)(dst - halo) ^ 64)) = lo;      // unit + 4 of the swizzled row
    }
    // ---- nine taps x four pixel tiles x two cout halves x three terms (small terms first) ----
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap % 3;
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        const unsigned char* xp = halo + xoff[kx] + (((pt >> 1) + ky) * HWD + 16 * (pt & 1)) * ROWB;
        const uint4 xh = *reinterpret_cast<const uint4*>(xp);
        const uint4 xl = *reinterpret_cast<const uint4*>(halo + ((int)(xp - halo) ^ 64));
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) {
          m_f32x4 c = acc[pt][ch];
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(m_h8, Wf[tap % 3][ch][0]), __builtin_bit_cast(m_h8, xl), c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(m_h8, Wf[tap % 3][ch][1]), __builtin_bit_cast(m_h8, xh), c, 0, 0, 0);
          acc[pt][ch] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(m_h8, Wf[tap % 3][ch][0]), __builtin_bit_cast(m_h8, xh), c, 0, 0, 0);
        }
      }
      if (tap + 3 < 9) load_w(tap + 3, tap % 3);       // this tap's ring slot is free
    }
  }

  // ---- the eight partial tiles meet in LDS: red[wave][pixel][cout quad] x 16 B (over the halo images) ----
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem_m);
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const int px = 32 * (pt >> 1) + 16 * (pt & 1) + c15, cq = 4 * ch + g;        // couts 16 ch + 4 g .. + 3
      *reinterpret_cast<m_f32x4*>(red + ((wave * (TH * TW) + px) * (BN / 4) + (cq ^ (px & 7))) * 4) = acc[pt][ch];
    }
  __syncthreads();
  // ---- epilogue: thread -> (pixel tid >> 3, cout quad tid & 7): the slices in wave order, bias + noise shift + residual, store ----
  const int px = tid >> 3, cq = tid & 7, co = co0 + 4 * cq;
  m_f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int wv = 0; wv < Cfg::NWV; ++wv)
    a += *reinterpret_cast<const m_f32x4*>(red + ((wv * (TH * TW) + px) * (BN / 4) + (cq ^ (px & 7))) * 4);
  const float* tembp = p.temb ? p.temb + (size_t)n * p.temb_stride + p.temb_off : p.bias;   // (unconditional loads)
  const float tmul = p.temb ? 1.f : 0.f;
  m_f32x4 add;
#pragma unroll
  for (int r = 0; r < 4; ++r) add[r] = p.bias[co + r] + tmul * tembp[co + r];
  const float winv = p.w_inv_scale_dev ? *p.w_inv_scale_dev : p.w_inv_scale;
  const size_t o = ((size_t)(n * p.Hout + oy0 + (px >> 5)) * p.Wout + ox0 + (px & 31)) * p.Cout + co;
  a = a * winv + add;
  if (p.res) a += *reinterpret_cast<const m_f32x4*>(p.res + o);
  *reinterpret_cast<m_f32x4*>(p.out + o) = a;
  // ---- GroupNorm partial sums of the tile: per channel over its 64 pixels (lanes of equal cout quad, then the 8 waves in order) ----
  if (p.part_out) {
    float v[8] = {a[0], a[0] * a[0], a[1], a[1] * a[1], a[2], a[2] * a[2], a[3], a[3] * a[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int m = 8; m < 64; m <<= 1) v[e] += __shfl_xor(v[e], m, 64);
    __syncthreads();                               // (the partial tiles have been read)
    float* sst = reinterpret_cast<float*>(smem_m);  // [wave][cout quad][8]
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) sst[(wave * 8 + lane) * 8 + e] = v[e];
    }
    __syncthreads();
    if (tid < BN * 2) {                            // thread -> (channel tid >> 1, statistic tid & 1)
      const int c = tid >> 1, st = tid & 1;
      float t = 0.f;
#pragma unroll
      for (int wv = 0; wv < Cfg::NWV; ++wv) t += sst[(wv * 8 + (c >> 2)) * 8 + 2 * (c & 3) + st];
      p.part_out[(((size_t)n * (tilesX * tilesY) + ty * tilesX + tx) * p.Cout + co0 + c) * 2 + st] = t;
    }
  }
}

// Launches of small grids (g_tun.small: on) in f16x3: 256 couts, 256 or 256 | 256 input channels, GroupNorm'ed input, no rider.
bool conv_small_ok(ConvKind kind, int prec, const ConvParams& p) {
  if (!g_tun.small || prec != PREC_F16X3 || kind != CONV3_S1) return false;
  if (p.xr0 || !p.gn_scale || p.drop_mask || p.out_bf16) return false;
  if (p.Cout != 256 || p.Cout_pad != 256 || p.C0 != 256 || (p.C1 != 0 && p.C1 != 256) || p.C0 + p.C1 != p.Cin_pad) return false;
  if (p.Hin != p.Hout || p.Win != p.Wout || p.Wout % 32 || p.Hout % 2) return false;
  const long wgs = (long)p.N * (p.Wout / 32) * (p.Hout / 2) * 8;
  return wgs <= g_tun.small_max_wgs;    // (large grids: the tile kernels, which re-stage nothing)
}

hipError_t launch_conv_small(const ConvParams& p, hipStream_t s, int* tiles) {
  const int tilesX = p.Wout / 32, tilesY = p.Hout / 2;
  if (tiles) *tiles = tilesX * tilesY;
  hipLaunchKernelGGL(conv_small_kernel, dim3(p.N * tilesX * tilesY * (p.Cout / 32)), dim3(512), (size_t)SmallCfg::LDS_BYTES, s, p);
  return hipGetLastError();
}

hipError_t kernels_small_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace fdsr
