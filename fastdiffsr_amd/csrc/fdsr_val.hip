// Val-loop kernels on uint8 images (gfx950): the per-image metrics of the reference's evaluation loop
// (FastDiffSR/sr_mfe.py:313-345 -- skimage's compare_mse / compare_psnr / compare_ssim(multichannel=True) and
// core/metrics.py:147-152 calculate_ergas; core/metrics.py:103-145 ssim / calculate_ssim) and the dataset's tensor
// transform (data/util.py:66-75: ToTensor() then x * (max - min) + min).
//
// Everything an integer can carry is carried as one: the squared-error sum, the pixel sum and -- for the uniform 7x7
// window of skimage's compare_ssim -- the five window sums (x, y, x^2, y^2, xy) are EXACT int32 / uint64 values; only the
// SSIM quotient itself and the 11x11 Gaussian window of core/metrics.ssim are fp64.  Reductions run in a fixed order (no
// floating-point atomics): a rerun is bitwise identical.  The host turns the per-image sums into MSE / PSNR / ERGAS with
// the reference's own scalar formulas (fastdiffsr_amd/metrics.py), so those three match the host path bit for bit.
#include "fdsr_kernels.h"

#include <cmath>
#include <type_traits>

namespace fdsr {

namespace {

constexpr int VT = 256;     // threads = input columns a workgroup stages
constexpr int VTR = 16;     // output rows per workgroup

struct GaussTaps { double g[11]; };

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);     // fixed butterfly: reruns are bitwise
  return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// One workgroup = one image x one band of VTR valid rows x one chunk of VT - 2R valid columns.  Thread t stages input column
// xs + t (all C channels) of the band's VTR + 2R rows into LDS, forms the vertical window sums of its column for one output
// row at a time, hands them to its neighbours through LDS and -- if it is a valid output column -- adds up the horizontal
// window and the SSIM quotient.  The workgroup also owns a rectangle of input pixels for the squared-error / pixel sums
// (bands and chunks tile the image; the borders belong to the first / last band and chunk).
// partial [N][tiles][4] = (ssim sum, ssim positions, squared-error sum, sum of `a`), tiles = bands * chunks.
template <int R, bool GAUSS>
__global__ void __launch_bounds__(VT) ssim_u8_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b,
                                                     int H, int W, int C, int bands, int chunks, GaussTaps taps,
                                                     double* __restrict__ partial, int want_sse) {
  constexpr int K = 2 * R + 1;
  constexpr int OC = VT - 2 * R;
  constexpr int ROWS = VTR + 2 * R;
  using acc_t = typename std::conditional<GAUSS, double, int>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = threadIdx.x;
  const int cx = blockIdx.x % chunks, by = (blockIdx.x / chunks) % bands, n = blockIdx.x / (chunks * bands);
  const int xs = cx * OC, ys = by * VTR;                      // first staged input column / row
  unsigned char* sa = smem;                                   // [ROWS][C][VT]
  unsigned char* sb = sa + ROWS * C * VT;
  acc_t* ex = reinterpret_cast<acc_t*>(sb + ROWS * C * VT);   // [5][C][VT]   (ROWS*C*VT*2 is a multiple of 16)
  const int xin = xs + t;
  const size_t img = (size_t)n * H * W * C;

  // stage + the owned rectangle's integer sums
  const int own_y0 = by == 0 ? 0 : R + ys, own_y1 = by == bands - 1 ? H : R + ys + VTR;
  const int own_t0 = cx == 0 ? 0 : R, own_t1 = cx == chunks - 1 ? W - xs : VT - R;
  unsigned int sse = 0, suma = 0;
  for (int r = 0; r < ROWS; ++r) {
    const int y = ys + r;
    const bool in = y < H && xin < W;
    const bool own = want_sse && in && y >= own_y0 && y < own_y1 && t >= own_t0 && t < own_t1;
    for (int c = 0; c < C; ++c) {
      int va = 0, vb = 0;
      if (in) {
        const size_t o = img + ((size_t)y * W + xin) * C + c;
        va = a[o];
        vb = b[o];
      }
      sa[(r * C + c) * VT + t] = (unsigned char)va;
      sb[(r * C + c) * VT + t] = (unsigned char)vb;
      if (own) { const int d = va - vb; sse += (unsigned)(d * d); suma += (unsigned)va; }
    }
  }
  __syncthreads();

  const bool out_col = t >= R && t < VT - R && xin < W - R;
  double acc = 0.0;
  unsigned int cnt = 0;
  const int rows_here = min(VTR, H - 2 * R - ys);             // valid output rows of this band
  for (int r = 0; r < rows_here; ++r) {
    for (int c = 0; c < C; ++c) {
      acc_t s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int va = sa[((r + k) * C + c) * VT + t], vb = sb[((r + k) * C + c) * VT + t];
        if (GAUSS) {
          const double w = taps.g[k], fa = (double)va, fb = (double)vb;
          s0 += w * fa; s1 += w * fb; s2 += w * (fa * fa); s3 += w * (fb * fb); s4 += w * (fa * fb);
        } else {
          s0 += va; s1 += vb; s2 += va * va; s3 += vb * vb; s4 += va * vb;
        }
      }
      ex[(0 * C + c) * VT + t] = s0;
      ex[(1 * C + c) * VT + t] = s1;
      ex[(2 * C + c) * VT + t] = s2;
      ex[(3 * C + c) * VT + t] = s3;
      ex[(4 * C + c) * VT + t] = s4;
    }
    __syncthreads();
    if (out_col) {
      for (int c = 0; c < C; ++c) {
        acc_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const int j = t - R + k;
          if (GAUSS) {
            const double w = taps.g[k];
            h0 += w * ex[(0 * C + c) * VT + j]; h1 += w * ex[(1 * C + c) * VT + j]; h2 += w * ex[(2 * C + c) * VT + j];
            h3 += w * ex[(3 * C + c) * VT + j]; h4 += w * ex[(4 * C + c) * VT + j];
          } else {
            h0 += ex[(0 * C + c) * VT + j]; h1 += ex[(1 * C + c) * VT + j]; h2 += ex[(2 * C + c) * VT + j];
            h3 += ex[(3 * C + c) * VT + j]; h4 += ex[(4 * C + c) * VT + j];
          }
        }
        const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
        double S;
        if (GAUSS) {
          // core/metrics.py:111-122: mu = filter(img), sigma = filter(img^2) - mu^2, ...
          const double mu1 = h0, mu2 = h1, mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
          const double s1 = h2 - mu1_sq, s2 = h3 - mu2_sq, s12 = h4 - mu12;
          S = ((2.0 * mu12 + C1) * (2.0 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2));
        } else {
          // skimage 0.16 compare_ssim defaults: uniform 7x7 window, sample covariance (NP / (NP - 1)), K1 = 0.01, K2 = 0.03
          const double NP = (double)(K * K), cov_norm = NP / (NP - 1.0);
          const double ux = (double)h0 / NP, uy = (double)h1 / NP, uxx = (double)h2 / NP, uyy = (double)h3 / NP,
                       uxy = (double)h4 / NP;
          const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
          S = ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
        }
        acc += S;
        ++cnt;
      }
    }
    __syncthreads();
  }

  // workgroup reduction in a fixed order: lanes (butterfly), then waves 0..3 (scratch: the exchange buffer, free after the
  // loop's last barrier; no static LDS, so the dynamic region starts 16-byte aligned)
  double* red_f = reinterpret_cast<double*>(ex);
  unsigned long long (*red_u)[4] = reinterpret_cast<unsigned long long (*)[4]>(red_f + 4);
  const int wave = t >> 6, lane = t & 63;
  const double wacc = wave_sum_f64(acc);
  const unsigned long long wcnt = wave_sum_u64(cnt), wsse = wave_sum_u64(sse), wsum = wave_sum_u64(suma);
  if (lane == 0) { red_f[wave] = wacc; red_u[0][wave] = wcnt; red_u[1][wave] = wsse; red_u[2][wave] = wsum; }
  __syncthreads();
  if (t == 0) {
    double* p = partial + ((size_t)n * bands * chunks + (size_t)by * chunks + cx) * 4;
    p[0] = ((red_f[0] + red_f[1]) + red_f[2]) + red_f[3];
    p[1] = (double)(red_u[0][0] + red_u[0][1] + red_u[0][2] + red_u[0][3]);
    p[2] = (double)(red_u[1][0] + red_u[1][1] + red_u[1][2] + red_u[1][3]);      // < 2^53: exact
    p[3] = (double)(red_u[2][0] + red_u[2][1] + red_u[2][2] + red_u[2][3]);
  }
}

// out [N][8] = (squared-error sum, sum of a, uniform-7 ssim sum, its positions, gauss-11 ssim sum, its positions, 0, 0):
// tiles summed in index order by one thread per field
__global__ void __launch_bounds__(64) metrics_finalize_kernel(const double* __restrict__ pu, int tiles_u, const double* __restrict__ pg,
                                                              int tiles_g, double* __restrict__ out) {
  const int n = blockIdx.x, f = threadIdx.x;
  if (f >= 8) return;
  double v = 0.0;
  if (f < 4 && pu) {          // fields 0..3 from the uniform pass: sse (2), suma (3), ssim (0), count (1)
    const int src = f == 0 ? 2 : f == 1 ? 3 : f == 2 ? 0 : 1;
    for (int i = 0; i < tiles_u; ++i) v += pu[((size_t)n * tiles_u + i) * 4 + src];
  } else if (f < 4 && pg && f < 2) {   // no uniform pass: the integer sums come from the Gaussian pass
    const int src = f == 0 ? 2 : 3;
    for (int i = 0; i < tiles_g; ++i) v += pg[((size_t)n * tiles_g + i) * 4 + src];
  } else if ((f == 4 || f == 5) && pg) {
    for (int i = 0; i < tiles_g; ++i) v += pg[((size_t)n * tiles_g + i) * 4 + (f - 4)];
  }
  out[(size_t)n * 8 + f] = v;
}

// data/util.py:66-75: ToTensor() (uint8 / 255 -> fp32, HWC -> CHW) then img * (max - min) + min
__global__ void __launch_bounds__(256) u8_to_tensor_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int C, int HW,
                                                           float lo, float hi, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over N*H*W pixels
  if (i >= total) return;
  const size_t n = i / HW, pix = i % HW;
  const float span = __fsub_rn(hi, lo);
  for (int c = 0; c < C; ++c)
    dst[(n * C + c) * HW + pix] = __fadd_rn(__fmul_rn(__fdiv_rn((float)src[i * C + c], 255.0f), span), lo);
}

template <int R>
void tile_counts(int H, int W, int* bands, int* chunks) {
  *bands = (H - 2 * R + VTR - 1) / VTR;
  *chunks = (W - 2 * R + (VT - 2 * R) - 1) / (VT - 2 * R);
}

template <int R, bool GAUSS>
size_t ssim_lds_bytes(int C) {
  return (size_t)(VTR + 2 * R) * C * VT * 2 + (size_t)5 * C * VT * (GAUSS ? sizeof(double) : sizeof(int));
}

}  // namespace

size_t image_metrics_workspace_bytes(int N, int H, int W) {
  int bu = 0, cu = 0, bg = 0, cg = 0;
  if (H > 6 && W > 6) tile_counts<3>(H, W, &bu, &cu);
  if (H > 10 && W > 10) tile_counts<5>(H, W, &bg, &cg);
  return ((size_t)N * bu * cu + (size_t)N * bg * cg) * 4 * sizeof(double) + 64;
}

hipError_t launch_image_metrics_u8(const unsigned char* a, const unsigned char* b, int N, int H, int W, int C, int flags, double* out,
                                   void* ws, hipStream_t s) {
  int bu = 0, cu = 0, bg = 0, cg = 0;
  const bool uni = (flags & 1) != 0, gau = (flags & 2) != 0;
  if (uni) tile_counts<3>(H, W, &bu, &cu);
  if (gau) tile_counts<5>(H, W, &bg, &cg);
  double* pu = reinterpret_cast<double*>(ws);
  double* pg = pu + (size_t)N * bu * cu * 4;
  GaussTaps taps{};
  {   // cv2.getGaussianKernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)), normalised to sum 1
    double sum = 0.0;
    for (int i = 0; i < 11; ++i) { taps.g[i] = std::exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); sum += taps.g[i]; }
    for (int i = 0; i < 11; ++i) taps.g[i] /= sum;
  }
  if (uni) {
    auto k = ssim_u8_kernel<3, false>;
    const size_t lds = ssim_lds_bytes<3, false>(C);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)(N * bu * cu)), dim3(VT), lds, s, a, b, H, W, C, bu, cu, taps, pu, 1);
  }
  if (gau) {
    auto k = ssim_u8_kernel<5, true>;
    const size_t lds = ssim_lds_bytes<5, true>(C);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)(N * bg * cg)), dim3(VT), lds, s, a, b, H, W, C, bg, cg, taps, pg, uni ? 0 : 1);
  }
  hipLaunchKernelGGL(metrics_finalize_kernel, dim3((unsigned)N), dim3(64), 0, s, uni ? pu : nullptr, bu * cu, gau ? pg : nullptr,
                     bg * cg, out);
  return hipGetLastError();
}

hipError_t launch_u8_to_tensor(const unsigned char* src, float* dst, int N, int C, int H, int W, float lo, float hi, hipStream_t s) {
  const size_t total = (size_t)N * H * W;
  hipLaunchKernelGGL(u8_to_tensor_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, C, H * W, lo, hi, total);
  return hipGetLastError();
}

}  // namespace fdsr
