// Training-step kernels for gfx950 (fdsr_train.h): loss, backward of GroupNorm+Swish, convolution weight
// gradients (exact fp32 MFMA), CLAM / SLAM / noise-embedding backward, Adam, weight re-packing.
// Input gradients of the convolutions run on the forward kernels with transposed, tap-flipped weights
// (fdsr_train.cpp).  Every reduction is ordered: a step is bitwise reproducible.
#include "fdsr_train.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace fdsr {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// fixed-order sum of one double per thread over a 256-thread block; result valid in thread 0
__device__ __forceinline__ double block_sum_256(double v, double* sh /* [4] */) {
  v = wave_sum_d(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

}  // namespace

// ---------------------------------------------------------------------------
// network input of a training step, formed on the device: img2res (diffusion.py:283-289), q_sample (:233-241) and
// cat([SR, x_noisy], dim=1) (:257-263) in the kernel that writes the packed NHWC input.  Separately rounded products and
// sums (no contraction), the same arithmetic as the tensor torch forms op by op.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) qsample_pack_kernel(const float* __restrict__ hr, const float* __restrict__ sr,
                                                           const float* __restrict__ gamma, const float* __restrict__ noise,
                                                           float* __restrict__ xin, int HW, int CP, size_t npix) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const size_t n = p / HW, hw = p % HW;
  const float g = gamma[n];
  const float sg = __fsqrt_rn(__fsub_rn(1.0f, __fmul_rn(g, g)));   // (1 - gamma ** 2).sqrt()
  float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const size_t i = (n * 3 + c) * HW + hw;
    const float s = sr[i];
    const float xs = fminf(fmaxf(__fmul_rn(__fsub_rn(hr[i], s), 2.0f), -1.0f), 1.0f);   // img2res: ((hr - sr) * 2).clamp(-1, 1)
    o[c] = s;
    o[3 + c] = __fadd_rn(__fmul_rn(g, xs), __fmul_rn(sg, noise[i]));                    // gamma * x_start + sqrt(1 - gamma^2) * noise
  }
  float* dst = xin + p * CP;
  for (int c = 0; c < CP; ++c) dst[c] = c < 8 ? o[c] : 0.f;
}

hipError_t launch_qsample_pack(const float* hr, const float* sr, const float* gamma, const float* noise, float* xin, int N, int HW,
                               int CP, hipStream_t s) {
  const size_t npix = (size_t)N * HW;
  hipLaunchKernelGGL(qsample_pack_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, hr, sr, gamma, noise, xin, HW, CP, npix);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// loss
// ---------------------------------------------------------------------------
size_t loss_partial_count(size_t npix) { return (npix + 255) / 256; }

__global__ void __launch_bounds__(256) loss_grad_kernel(const float* __restrict__ eps, const float* __restrict__ target,
                                                        float* __restrict__ deps8, double* __restrict__ partial, int HW, size_t npix,
                                                        int l2, float scale) {
  __shared__ double sh[4];
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  if (p < npix) {
    const size_t n = p / HW, hw = p % HW;
    f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float e = eps[p * 3 + c], t = target[(n * 3 + c) * HW + hw];
      const float d = e - t;
      if (l2 == 2) {   // Charbonnier, eps = 1e-3 (TESR's 'l1': tesr_modules/unet.py:956-967, diffusion.py:85-90); the caller's scale carries the mean
        const float r = sqrtf(d * d + 1e-6f);
        acc += (double)r;
        lo[c] = d / r * scale;
      } else if (l2) { acc += (double)d * (double)d; lo[c] = 2.0f * d * scale; }
      else { acc += (double)fabsf(d); lo[c] = (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.0f)) * scale; }   // d|x|/dx, sign(0) = 0 like torch
    }
    *reinterpret_cast<f32x4*>(deps8 + p * 8) = lo;
    *reinterpret_cast<f32x4*>(deps8 + p * 8 + 4) = hi;
  }
  const double s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256) sum_partials_kernel(const double* __restrict__ partial, size_t n, float* __restrict__ out) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < n; i += 256) acc += partial[i];
  const double s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) out[0] = (float)s;
}

hipError_t launch_loss_grad(const float* eps, const float* target, float* deps8, double* partial, float* loss_out, int N, int HW,
                            int l2, float scale, hipStream_t s) {
  const size_t npix = (size_t)N * HW, nb = loss_partial_count(npix);
  hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)nb), dim3(256), 0, s, eps, target, deps8, partial, HW, npix, l2, scale);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partial, nb, loss_out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// per-(image, channel) pixel sums
// ---------------------------------------------------------------------------
#define FDSR_COLSUM_SLICES 64
size_t colsum_scratch_doubles(int N, int HW, int C) { (void)HW; return (size_t)N * FDSR_COLSUM_SLICES * C; }

// grid (slices, N): thread (channel quad c4, row r) sums its pixels of the slice; rows folded in order
__global__ void __launch_bounds__(256) colsum_part_kernel(const float* __restrict__ dy, int HW, int C, double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) double ps[256 * 4];
  const int tid = threadIdx.x, sl = blockIdx.x, n = blockIdx.y;
  const int cq = C >> 2;
  const int p0 = (int)((long)sl * HW / FDSR_COLSUM_SLICES), p1 = (int)((long)(sl + 1) * HW / FDSR_COLSUM_SLICES);
  for (int cb = 0; cb < cq; cb += 256) {           // channel-quad blocks (C <= 1024: one pass)
    const int nq = min(cq - cb, 256), rows = 256 / nq, c4 = tid % nq, r = tid / nq;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < rows) {
      const size_t base = (size_t)n * HW * C + (size_t)(cb + c4) * 4;
      for (int pix = p0 + r; pix < p1; pix += rows) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(dy + base + (size_t)pix * C);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += (double)v[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) ps[tid * 4 + e] = a[e];
    __syncthreads();
    if (tid < nq) {
      double t[4] = {0.0, 0.0, 0.0, 0.0};
      for (int rr = 0; rr < rows; ++rr)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += ps[(rr * nq + tid) * 4 + e];
      double* dst = part + ((size_t)n * FDSR_COLSUM_SLICES + sl) * C + (size_t)(cb + tid) * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[e] = t[e];
    }
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) colsum_fold_kernel(const double* __restrict__ part, int C, float* __restrict__ S) {
  const int n = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0;
  for (int sl = 0; sl < FDSR_COLSUM_SLICES; ++sl) a += part[((size_t)n * FDSR_COLSUM_SLICES + sl) * C + c];
  S[(size_t)n * C + c] = (float)a;
}

hipError_t launch_colsum(const float* dy, float* S, double* scratch, int N, int HW, int C, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colsum_part_kernel, dim3(FDSR_COLSUM_SLICES, N), dim3(256), 0, s, dy, HW, C, scratch);
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((C + 255) / 256, N), dim3(256), 0, s, scratch, C, S);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) scale_inplace_kernel(float* __restrict__ x, size_t n, float f) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] *= f;
}

// dst[tab[3e+1] + i] = src[tab[3e] + i], i < tab[3e+2]: one workgroup per table entry (the small tensors of a model in one launch)
__global__ void __launch_bounds__(256) copy_table_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         const unsigned long long* __restrict__ tab) {
  // grid (entries, split): a tensor's elements are dealt over the workgroups of its row (most entries are a few hundred
  // floats; the GDP sibling's per-block Linears are 4 MB each -- one workgroup per tensor moved them at 76 GB/s)
  const unsigned long long so = tab[3 * blockIdx.x], dof = tab[3 * blockIdx.x + 1], n = tab[3 * blockIdx.x + 2];
  for (unsigned long long i = (unsigned long long)blockIdx.y * 256 + threadIdx.x; i < n; i += 256ull * gridDim.y) dst[dof + i] = src[so + i];
}
hipError_t launch_copy_table(const float* src, float* dst, const unsigned long long* tab, int entries, hipStream_t s, size_t max_elems) {
  const unsigned split = (unsigned)std::min<size_t>(64, std::max<size_t>(1, (max_elems + 32767) / 32768));   // >= 32 K floats per workgroup of the largest entry
  if (entries > 0) hipLaunchKernelGGL(copy_table_kernel, dim3(entries, split), dim3(256), 0, s, src, dst, tab);
  return hipGetLastError();
}

hipError_t launch_scale_inplace(float* x, size_t n, float f, hipStream_t s) {
  hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n, f);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) sum_rows_kernel(const float* __restrict__ S, int N, int stride, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  // eight rows per trip, their loads in flight together (rows past the end re-read the last one and add 0: the same sum in the same
  // order); one load per trip of a loop with a runtime bound is one memory round trip per image -- 9.4 us for 32 images
  float a = 0.f;
  for (int n0 = 0; n0 < N; n0 += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = S[(size_t)min(n0 + j, N - 1) * stride + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) a += n0 + j < N ? v[j] : 0.f;
  }
  out[c] = a;
}

hipError_t launch_sum_rows(const float* S, int N, int stride, int C, float* out, hipStream_t s) {
  hipLaunchKernelGGL(sum_rows_kernel, dim3((C + 255) / 256), dim3(256), 0, s, S, N, stride, C, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// layout helpers of the strided / upsampled convolutions' transposes
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) zero_insert_kernel(const float* __restrict__ dy, float* __restrict__ z, int H, int W, int cq,
                                                          size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [N][2H][2W][cq] quads
  if (i >= total) return;
  const int c4 = (int)(i % cq);
  size_t r = i / cq;
  const int x = (int)(r % (2 * W));
  r /= (2 * W);
  const int y = (int)(r % (2 * H));
  const size_t n = r / (2 * H);
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (!(x & 1) && !(y & 1)) v = *reinterpret_cast<const f32x4*>(dy + (((n * H + (y >> 1)) * W + (x >> 1)) * cq + c4) * 4);
  *reinterpret_cast<f32x4*>(z + i * 4) = v;
}

hipError_t launch_zero_insert(const float* dy, float* z, int N, int H, int W, int C, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  const size_t total = (size_t)N * 2 * H * 2 * W * (C >> 2);
  hipLaunchKernelGGL(zero_insert_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dy, z, H, W, C >> 2, total);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) pool2_add_kernel(const float* __restrict__ du, float* __restrict__ dx, int H, int W, int cq,
                                                        size_t total, int assign) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [N][H][W][cq]
  if (i >= total) return;
  const int c4 = (int)(i % cq);
  size_t r = i / cq;
  const int x = (int)(r % W);
  r /= W;
  const int y = (int)(r % H);
  const size_t n = r / H;
  const size_t W2 = 2 * (size_t)W;
  const float* b = du + (((n * 2 * H + 2 * y) * W2 + 2 * x) * cq + c4) * 4;
  const f32x4 a0 = *reinterpret_cast<const f32x4*>(b), a1 = *reinterpret_cast<const f32x4*>(b + (size_t)cq * 4);
  const f32x4 a2 = *reinterpret_cast<const f32x4*>(b + W2 * cq * 4), a3 = *reinterpret_cast<const f32x4*>(b + (W2 + 1) * cq * 4);
  f32x4 v = (a0 + a1) + (a2 + a3);
  if (!assign) v += *reinterpret_cast<const f32x4*>(dx + i * 4);
  *reinterpret_cast<f32x4*>(dx + i * 4) = v;
}

hipError_t launch_pool2_add(const float* du, float* dx, int N, int H, int W, int C, bool assign, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  const size_t total = (size_t)N * H * W * (C >> 2);
  hipLaunchKernelGGL(pool2_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, du, dx, H, W, C >> 2, total, assign ? 1 : 0);
  return hipGetLastError();
}

// 2x2 average pooling, transposed: every fine pixel takes scale * dy of its block
__global__ void __launch_bounds__(256) unpool2_add_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int cq,
                                                          float scale, int assign, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [N][2H][2W][cq]
  if (i >= total) return;
  const int c4 = (int)(i % cq);
  size_t r = i / cq;
  const int x = (int)(r % (2 * W));
  r /= 2 * W;
  const int y = (int)(r % (2 * H));
  const size_t n = r / (2 * H);
  const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (((n * H + (y >> 1)) * W + (x >> 1)) * cq + c4) * 4);
  f32x4 o = v * scale;
  if (!assign) o += *reinterpret_cast<const f32x4*>(dx + i * 4);
  *reinterpret_cast<f32x4*>(dx + i * 4) = o;
}
hipError_t launch_unpool2_add(const float* dy, float* dx, int N, int H, int W, int C, float scale, bool assign, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  const size_t total = (size_t)N * 2 * H * 2 * W * (C >> 2);
  hipLaunchKernelGGL(unpool2_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dy, dx, H, W, C >> 2, scale, assign ? 1 : 0,
                     total);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) add_slice_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cs4, int off4,
                                                        int C4, size_t total, int assign) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [npix][C4]
  if (i >= total) return;
  const size_t p = i / C4;
  const int c = (int)(i % C4);
  f32x4 v = *reinterpret_cast<const f32x4*>(src + (p * Cs4 + off4 + c) * 4);
  if (!assign) v += *reinterpret_cast<const f32x4*>(dst + i * 4);
  *reinterpret_cast<f32x4*>(dst + i * 4) = v;
}

hipError_t launch_add_slice(const float* src, float* dst, size_t npix, int Cs, int off, int C, bool assign, hipStream_t s) {
  if ((Cs | off | C) & 3) return hipErrorInvalidValue;
  const size_t total = npix * (C >> 2);
  hipLaunchKernelGGL(add_slice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, Cs >> 2, off >> 2, C >> 2, total, assign ? 1 : 0);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// GroupNorm + Swish backward
// ---------------------------------------------------------------------------
#define FDSR_GNB_SLICES 64
// scratch: part [N][SLICES][C][2] | tot [N][C][2] | gm [N][G][2]   (doubles)
// (the partial region also holds the fused form's fp32 tile sums: tiles of >= 2 rows x 32 pixels)
static size_t gn_bwd_max_tiles(int H, int W) { return (size_t)((W + 31) / 32) * ((H + 1) / 2); }
static size_t gn_bwd_part_doubles(int N, int H, int W, int C) {
  const size_t slices = std::max<size_t>(FDSR_GNB_SLICES, (gn_bwd_max_tiles(H, W) + 1) / 2);
  return (size_t)N * slices * C * 2;
}
size_t gn_bwd_scratch_doubles(int N, int H, int W, int C) {
  return gn_bwd_part_doubles(N, H, W, C) + (size_t)N * C * 2 + (size_t)N * C * 2;
}

// dA arrives already multiplied by the dropout factor (keep / (1-p)) of this element when dropout is on
__device__ __forceinline__ void gnb_elem(float x, float dA, float sc, float sh, float mean, float rstd, int plain, float& g, float& xhat) {
  xhat = (x - mean) * rstd;
  if (plain) { g = dA; return; }
  const float u = fmaf(x, sc, sh);
  const float sg = sigmoid_f(u);
  g = dA * (sg * (1.0f + u * (1.0f - sg)));     // d/du [u * sigmoid(u)]
}

// gamma as image n sees it: the scale-shift form (GnBwdParams::film) multiplies the affine output by (1 + s[n][c])
__device__ __forceinline__ float gnb_gamma(const GnBwdParams& p, size_t n, int c) {
  const float gam = p.gamma[c];
  return p.film ? gam * (1.0f + p.film[n * p.film_stride + c]) : gam;
}

// grid (SLICES, N): per-(image, channel) partial sums of g and g*xhat over a pixel slice
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const GnBwdParams p, double* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) double ps[256 * 8];
  const int tid = threadIdx.x, sl = blockIdx.x, n = blockIdx.y;
  const int C = p.C0 + p.C1, cq = C >> 2, cpg = C / p.G;
  const int p0 = (int)((long)sl * p.HW / FDSR_GNB_SLICES), p1 = (int)((long)(sl + 1) * p.HW / FDSR_GNB_SLICES);
  for (int cb = 0; cb < cq; cb += 256) {
    const int nq = min(cq - cb, 256), rows = 256 / nq, c4 = tid % nq, r = tid / nq;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (r < rows) {
      const int c = (cb + c4) * 4;
      const float* xs; int Cs, cc;
      if (c < p.C0) { xs = p.x0; Cs = p.C0; cc = c; } else { xs = p.x1; Cs = p.C1; cc = c - p.C0; }
      const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + (size_t)n * C + c);
      const f32x4 sh = *reinterpret_cast<const f32x4*>(p.shift + (size_t)n * C + c);
      float mean[4], rstd[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int g = (c + e) / cpg;
        mean[e] = p.stats[((size_t)n * p.G + g) * 2];
        rstd[e] = p.stats[((size_t)n * p.G + g) * 2 + 1];
      }
      for (int pix = p0 + r; pix < p1; pix += rows) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(xs + ((size_t)n * p.HW + pix) * Cs + cc);
        f32x4 d = *reinterpret_cast<const f32x4*>(p.dA + ((size_t)n * p.HW + pix) * C + c);
        if (p.drop_mask) {
          const unsigned m = *reinterpret_cast<const unsigned*>(p.drop_mask + ((size_t)n * p.HW + pix) * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) d[e] = ((m >> (8 * e)) & 0xffu) ? d[e] * p.drop_scale : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float g, xh;
          gnb_elem(x[e], d[e], sc[e], sh[e], mean[e], rstd[e], p.plain, g, xh);
          a[e] += (double)g;
          b[e] += (double)g * (double)xh;
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { ps[tid * 8 + e] = a[e]; ps[tid * 8 + 4 + e] = b[e]; }
    __syncthreads();
    if (tid < nq) {
      double ta[4] = {0, 0, 0, 0}, tb[4] = {0, 0, 0, 0};
      for (int rr = 0; rr < rows; ++rr)
#pragma unroll
        for (int e = 0; e < 4; ++e) { ta[e] += ps[(rr * nq + tid) * 8 + e]; tb[e] += ps[(rr * nq + tid) * 8 + 4 + e]; }
      double* dst = part + (((size_t)n * FDSR_GNB_SLICES + sl) * C + (size_t)(cb + tid) * 4) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { dst[2 * e] = ta[e]; dst[2 * e + 1] = tb[e]; }
    }
    __syncthreads();
  }
}

// grid (N): fold the slices per channel, then per group the two means the apply pass needs
__global__ void __launch_bounds__(256) gn_bwd_finalize_kernel(const GnBwdParams p, const double* __restrict__ part, double* __restrict__ tot,
                                                              double* __restrict__ gm) {
  const int n = blockIdx.x, tid = threadIdx.x, C = p.C0 + p.C1, cpg = C / p.G;
  for (int c = tid; c < C; c += 256) {
    double a = 0.0, b = 0.0;
    for (int sl = 0; sl < FDSR_GNB_SLICES; ++sl) {
      const double* src = part + (((size_t)n * FDSR_GNB_SLICES + sl) * C + c) * 2;
      a += src[0];
      b += src[1];
    }
    tot[((size_t)n * C + c) * 2] = a;
    tot[((size_t)n * C + c) * 2 + 1] = b;
  }
  __syncthreads();
  for (int g = tid; g < p.G; g += 256) {
    double m1 = 0.0, m2 = 0.0;
    for (int k = 0; k < cpg; ++k) {
      const int c = g * cpg + k;
      const double gam = (double)gnb_gamma(p, n, c);
      m1 += gam * tot[((size_t)n * C + c) * 2];
      m2 += gam * tot[((size_t)n * C + c) * 2 + 1];
    }
    const double inv = 1.0 / ((double)cpg * (double)p.HW);
    gm[((size_t)n * p.G + g) * 2] = m1 * inv;
    gm[((size_t)n * p.G + g) * 2 + 1] = m2 * inv;
  }
}

// The fused form's finalisation (GnBwdParams::g_part): grid (N, G), one workgroup per group of one image.  Thread (ch, slice) adds the
// tiles slice, slice + S, ... of channel ch in double, the S (a power of two) slices are folded pairwise in LDS -- a fixed order.
__global__ void __launch_bounds__(256) gn_bwd_finalize_tiles_kernel(const GnBwdParams p, double* __restrict__ tot, double* __restrict__ gm) {
  __shared__ double sd[256][2];
  const int n = blockIdx.x, g = blockIdx.y, tid = threadIdx.x, C = p.C0 + p.C1, cpg = C / p.G;   // cpg <= 64 (launcher)
  int S = 1;
  while (S * 2 * cpg <= 256) S *= 2;
  const int ch = tid % cpg, slice = tid / cpg, c = g * cpg + ch;
  double a = 0.0, b = 0.0;
  if (slice < S) {
    const float* src = p.g_part + ((size_t)n * p.g_nt * C + c) * 2;
#pragma unroll 4
    for (int t = slice; t < p.g_nt; t += S) {
      const float2 v = *reinterpret_cast<const float2*>(src + (size_t)t * C * 2);
      a += (double)v.x;
      b += (double)v.y;
    }
  }
  sd[tid][0] = a;
  sd[tid][1] = b;
  __syncthreads();
  for (int h = S >> 1; h >= 1; h >>= 1) {
    if (slice < h) { sd[tid][0] += sd[tid + h * cpg][0]; sd[tid][1] += sd[tid + h * cpg][1]; }
    __syncthreads();
  }
  if (tid < cpg) {
    tot[((size_t)n * C + c) * 2] = sd[tid][0];
    tot[((size_t)n * C + c) * 2 + 1] = sd[tid][1];
  }
  if (tid == 0) {
    double m1 = 0.0, m2 = 0.0;
    for (int k = 0; k < cpg; ++k) {
      const double gam = (double)gnb_gamma(p, n, g * cpg + k);
      m1 += gam * sd[k][0];
      m2 += gam * sd[k][1];
    }
    const double inv = 1.0 / ((double)cpg * (double)p.HW);
    gm[((size_t)n * p.G + g) * 2] = m1 * inv;
    gm[((size_t)n * p.G + g) * 2 + 1] = m2 * inv;
  }
}

// dgamma[c] = sum_n s2[n][c], dbeta[c] = sum_n s1[n][c] (in image order)
__global__ void __launch_bounds__(256) gn_bwd_affine_kernel(const double* __restrict__ tot, int N, int C, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double a = 0.0, b = 0.0;
  for (int n0 = 0; n0 < N; n0 += 8) {   // (eight images per trip: as sum_rows_kernel)
    double va[8], vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const size_t o = ((size_t)min(n0 + j, N - 1) * C + c) * 2;
      va[j] = tot[o];
      vb[j] = tot[o + 1];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a += n0 + j < N ? va[j] : 0.0;
      b += n0 + j < N ? vb[j] : 0.0;
    }
  }
  dbeta[c] = (float)a;
  dgamma[c] = (float)b;
}
// the scale-shift form: the per-image sums weighted by (1 + s), and the gradient of (s, t) themselves
__global__ void __launch_bounds__(256) gn_bwd_affine_film_kernel(const GnBwdParams p, const double* __restrict__ tot) {
  const int C = p.C0 + p.C1, c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const double gam = (double)p.gamma[c], bet = (double)p.beta[c];
  double a = 0.0, b = 0.0;
  for (int n = 0; n < p.N; ++n) {
    const double sa = tot[((size_t)n * C + c) * 2], sb = tot[((size_t)n * C + c) * 2 + 1];   // sum g, sum g*xhat
    const double one_s = 1.0 + (double)p.film[(size_t)n * p.film_stride + c];
    a += one_s * sa;
    b += one_s * sb;
    p.dfilm[(size_t)n * p.dfilm_stride + c] = (float)(gam * sb + bet * sa);
    p.dfilm[(size_t)n * p.dfilm_stride + C + c] = (float)sa;
  }
  p.dbeta[c] = (float)a;
  p.dgamma[c] = (float)b;
}

// elementwise: dx += rstd * (gamma*g - m1 - xhat*m2)      (GREADY: dA holds g already, see GnBwdParams::g_part)
template <bool GREADY>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const GnBwdParams p, const double* __restrict__ gm, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [N][HW][C/4]
  if (i >= total) return;
  const int C = p.C0 + p.C1, cq = C >> 2, cpg = C / p.G;
  const int c = (int)(i % cq) * 4;
  const size_t pix = i / cq;                     // n*HW + p
  const size_t n = pix / p.HW;
  const float* xs; float* dxs; int Cs, cc, assign;
  if (c < p.C0) { xs = p.x0; dxs = p.dx0; Cs = p.C0; cc = c; assign = p.assign0; } else { xs = p.x1; dxs = p.dx1; Cs = p.C1; cc = c - p.C0; assign = p.assign1; }
  const f32x4 x = *reinterpret_cast<const f32x4*>(xs + pix * Cs + cc);
  f32x4 d = *reinterpret_cast<const f32x4*>(p.dA + pix * C + c);
  if (!GREADY && p.drop_mask) {
    const unsigned m = *reinterpret_cast<const unsigned*>(p.drop_mask + pix * C + c);
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = ((m >> (8 * e)) & 0xffu) ? d[e] * p.drop_scale : 0.f;
  }
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
  if (!GREADY) {
    sc = *reinterpret_cast<const f32x4*>(p.scale + n * C + c);
    sh = *reinterpret_cast<const f32x4*>(p.shift + n * C + c);
  }
  f32x4 o = {0.f, 0.f, 0.f, 0.f};
  if (!assign) o = *reinterpret_cast<const f32x4*>(dxs + pix * Cs + cc);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int g = (c + e) / cpg;
    const float mean = p.stats[(n * p.G + g) * 2], rstd = p.stats[(n * p.G + g) * 2 + 1];
    float gg, xh;
    gnb_elem(x[e], d[e], sc[e], sh[e], mean, rstd, GREADY ? 1 : p.plain, gg, xh);
    const float m1 = (float)gm[(n * p.G + g) * 2], m2 = (float)gm[(n * p.G + g) * 2 + 1];
    o[e] += rstd * (gnb_gamma(p, n, c + e) * gg - m1 - xh * m2);
  }
  *reinterpret_cast<f32x4*>(dxs + pix * Cs + cc) = o;
}

hipError_t launch_gn_bwd(const GnBwdParams& p, hipStream_t s) {
  const int C = p.C0 + p.C1;
  if ((p.C0 & 3) || (p.C1 & 3) || C % p.G || p.H <= 0 || p.HW % p.H) return hipErrorInvalidValue;
  double* part = p.scratch;
  double* tot = part + gn_bwd_part_doubles(p.N, p.H, p.HW / p.H, C);
  double* gm = tot + (size_t)p.N * C * 2;
  const size_t total = (size_t)p.N * p.HW * (C >> 2);
  if (p.film && (!p.beta || !p.dfilm || p.C1)) return hipErrorInvalidValue;
  auto affine = [&]() {
    if (p.film) hipLaunchKernelGGL(gn_bwd_affine_film_kernel, dim3((C + 255) / 256), dim3(256), 0, s, p, tot);
    else hipLaunchKernelGGL(gn_bwd_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, s, tot, p.N, C, p.dgamma, p.dbeta);
  };
  if (p.g_part) {   // the input-gradient launch did the reduction (per-tile sums) and left g in dA
    if (C / p.G > 64 || p.g_nt <= 0 || (size_t)p.g_nt > gn_bwd_max_tiles(p.H, p.HW / p.H) || p.g_part != gn_bwd_tile_part(p.scratch)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gn_bwd_finalize_tiles_kernel, dim3(p.N, p.G), dim3(256), 0, s, p, tot, gm);
    affine();
    hipLaunchKernelGGL(gn_bwd_apply_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, gm, total);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(FDSR_GNB_SLICES, p.N), dim3(256), 0, s, p, part);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(p.N), dim3(256), 0, s, p, part, tot, gm);
  affine();
  hipLaunchKernelGGL(gn_bwd_apply_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, gm, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// convolution weight gradient: exact fp32 on v_mfma_f32_32x32x2_f32
// ---------------------------------------------------------------------------
// Workgroup = 4 waves = one (64 output channels x 64 input channels) block of dW, all taps, over a slice of
// 4x16-pixel output tiles.  Per tile the dy tile [64 px][64 co] and the activated input halo
// [(4-1)*S+KS x (16-1)*S+KS px][64 ci] are staged in LDS (GroupNorm-apply + Swish fused, as the forward does);
// wave (wc, wi) then accumulates, for every tap, dW[32 co][32 ci] += dy^T (32 x 2 px) * a (2 px x 32) over the
// 32 pixel pairs: 9 accumulator tiles (144 VGPRs) per wave.  Slices write their blocks to scratch; a second
// kernel sums the slices in order and scatters into the checkpoint layout [Cout][Cin][ks][ks].
template <int KS, int STRIDE, bool UP>
struct WgCfg {
  // 4x16-pixel tiles (2x16 at stride 2): 47 KB of LDS per workgroup, so two workgroups share a CU and one's
  // staging runs beside the other's MFMAs (an 8x16 tile needs 84 KB: one workgroup per CU, nothing overlaps)
  static constexpr int TH = STRIDE == 2 ? 2 : 4, TW = 16, T = KS * KS;
  static constexpr int HH = (TH - 1) * STRIDE + KS, HWD = (TW - 1) * STRIDE + KS, NPIX = HH * HWD;
  static constexpr int ROW = 64 + 4;                       // floats per staged pixel (pad: conflict-free 32-lane rows)
  static constexpr int LDS_FLOATS = (TH * TW + NPIX) * ROW;
};

template <int KS, int STRIDE, bool UP>
__global__ void __launch_bounds__(256) wgrad_kernel(const WgradParams p, const int nslices, const int ncb, const int nib) {
  using Cfg = WgCfg<KS, STRIDE, UP>;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, T = Cfg::T, HWD = Cfg::HWD, NPIX = Cfg::NPIX, ROW = Cfg::ROW, PAD = KS / 2;
  extern __shared__ __attribute__((aligned(16))) float wsm[];
  float* sDy = wsm;                       // [TH*TW][ROW]
  float* sIn = wsm + TH * TW * ROW;       // [NPIX][ROW]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave & 1, wi = wave >> 1;                 // 32-channel halves of the (co, ci) block
  int b = blockIdx.x;
  const int sl = b % nslices;  b /= nslices;
  const int ib = b % nib;  b /= nib;
  const int cb = b;                                        // < ncb
  const int co0 = cb * 64, ci0 = ib * 64;
  const int Cin = p.C0 + p.C1;
  const int tilesX = (p.Wout + TW - 1) / TW, tilesY = (p.Hout + TH - 1) / TH;
  const int ntiles = p.N * tilesX * tilesY;
  const int t0 = (int)((long)sl * ntiles / nslices), t1 = (int)((long)(sl + 1) * ntiles / nslices);
  const int Hsrc = UP ? p.Hout : p.Hin, Wsrc = UP ? p.Wout : p.Win;    // grid the conv taps walk on

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int r31 = lane & 31, kh = lane >> 5;
  // Register prefetch of the NEXT tile's dy tile and raw input halo while the MFMAs of the current one run
  // (the activation is applied when the registers are stored to LDS): thread -> (channel quad q, pixels i*16 + tid/16).
  constexpr int NDY = TH * TW * 16 / 256, NIN = (NPIX * 16 + 255) / 256;
  const int q = tid & 15, prow = tid >> 4;
  f32x4 rdy[NDY], rin[NIN], rsc = {1.f, 1.f, 1.f, 1.f}, rsh = {0.f, 0.f, 0.f, 0.f};
  unsigned rmask[NIN];
  bool rok[NIN];
  auto prefetch = [&](int tile) {
    int tt = tile;
    const int tx = tt % tilesX;  tt /= tilesX;
    const int ty = tt % tilesY;
    const int n = tt / tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const int px = i * 16 + prow;
      const int oy = oy0 + px / TW, ox = ox0 + px % TW, co = co0 + q * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy < p.Hout && ox < p.Wout && co < p.Cout_s)
        v = *reinterpret_cast<const f32x4*>(p.dy + ((size_t)(n * p.Hout + oy) * p.Wout + ox) * p.Cout_s + co);
      rdy[i] = v;
    }
    const int c = ci0 + q * 4;
    if (p.gn_scale && c < Cin) {
      rsc = *reinterpret_cast<const f32x4*>(p.gn_scale + (size_t)n * Cin + c);
      rsh = *reinterpret_cast<const f32x4*>(p.gn_shift + (size_t)n * Cin + c);
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int hp = i * 16 + prow;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      unsigned m = 0x01010101u;
      bool ok = false;
      if (hp < NPIX) {
        const int hy = hp / HWD, hx = hp % HWD;
        const int iy = oy0 * STRIDE - PAD + hy, ix = ox0 * STRIDE - PAD + hx;
        if (iy >= 0 && iy < Hsrc && ix >= 0 && ix < Wsrc && c < Cin) {
          const int sy = UP ? (iy >> 1) : iy, sx = UP ? (ix >> 1) : ix;
          const float* xs; int Cs, cc;
          if (c < p.C0) { xs = p.x0; Cs = p.C0; cc = c; } else { xs = p.x1; Cs = p.C1; cc = c - p.C0; }
          const size_t o = ((size_t)(n * p.Hin + sy) * p.Win + sx) * Cs + cc;
          v = *reinterpret_cast<const f32x4*>(xs + o);
          if (p.drop_mask) m = *reinterpret_cast<const unsigned*>(p.drop_mask + o);   // dropout sits between the Swish and the conv (C1 == 0 there)
          ok = true;
        }
      }
      rin[i] = v;
      rmask[i] = m;
      rok[i] = ok;
    }
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NDY; ++i) *reinterpret_cast<f32x4*>(sDy + (i * 16 + prow) * ROW + q * 4) = rdy[i];
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int hp = i * 16 + prow;
      if (hp >= NPIX) continue;
      f32x4 v = rin[i];
      if (rok[i] && p.gn_scale) {            // the conv zero-pads the ACTIVATED tensor
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u = fmaf(v[e], rsc[e], rsh[e]);
          v[e] = p.gn_plain ? u : u * sigmoid_f(u);
        }
        if (p.drop_mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ((rmask[i] >> (8 * e)) & 0xffu) ? v[e] * p.drop_scale : 0.f;
        }
      }
      *reinterpret_cast<f32x4*>(sIn + hp * ROW + q * 4) = v;
    }
  };
  if (t0 < t1) prefetch(t0);
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();                                       // the previous tile's reads are done
    store();
    __syncthreads();
    if (tile + 1 < t1) prefetch(tile + 1);                 // in flight under the MFMAs below
    // ---- pixel pairs: A = dy^T (lane: co = r31, pixel kh of the pair), B = a shifted by the tap ----
#pragma unroll 2
    for (int pp = 0; pp < TH * TW / 2; ++pp) {
      // the pair = pixels (x, x + 8) of one tile row: their LDS rows are 8*ROW floats apart = 32 banks,
      // so the two half-waves read disjoint banks
      const int py = pp >> 3, pxx = (pp & 7) + 8 * kh;
      const int px = py * TW + pxx;                         // this lane's pixel of the pair
      const float av = sDy[px * ROW + wc * 32 + r31];
      const float* brow = sIn + ((py * STRIDE) * HWD + pxx * STRIDE) * ROW + wi * 32 + r31;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const float bv = brow[((t / KS) * HWD + (t % KS)) * ROW];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
      }
    }
  }
  // ---- write this slice's block: scratch [sl][cb][ib][t][64 co][64 ci]; D layout: col n = r31 (ci), rows 8*(i/4) + 4*kh + i%4 (co)
  float* dst = p.scratch + ((((size_t)sl * ncb + cb) * nib + ib) * T) * 4096;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;      // co within the wave's 32
      dst[(size_t)t * 4096 + (wc * 32 + row) * 64 + wi * 32 + r31] = acc[t][i];
    }
}

// dw[co][ci][t] = sum over slices (in order) of scratch[sl][cb][ib][t][co%64][ci%64]
__global__ void __launch_bounds__(256) wgrad_fold_kernel(const float* __restrict__ scratch, float* __restrict__ dw, int Cout, int Cin_real,
                                                         int T, int nslices, int ncb, int nib, size_t total) {
  // One workgroup folds 64 consecutive scratch elements (one row of a 64x64 (co, ci) block of one tap): wave w sums the w-th
  // quarter of the slices in ascending order (256 contiguous bytes per slice and wave), the four partial sums are added in
  // wave order -- a fixed association, fp64 -- and the row goes to the checkpoint layout [Cout][Cin][tap].
  __shared__ double part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + lane;              // over [cb][ib][t][co & 63][ci & 63]
  const size_t stride = (size_t)ncb * nib * T * 4096;
  const int s0 = w * nslices / 4, s1 = (w + 1) * nslices / 4;
  const float* src = scratch + e;
  double a = 0.0;
  int sl = s0;
  for (; sl + 8 <= s1; sl += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(sl + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += (double)v[u];
  }
  for (; sl < s1; ++sl) a += (double)src[(size_t)sl * stride];
  part[w][lane] = a;
  __syncthreads();
  if (w == 0) {
    const double r = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
    size_t q = e;
    const int ci = (int)(q & 63);  q >>= 6;
    const int co = (int)(q & 63);  q >>= 6;
    const int t = (int)(q % T);  q /= T;
    const int ib = (int)(q % nib), cb = (int)(q / nib);
    const int gco = cb * 64 + co, gci = ib * 64 + ci;
    if (gco < Cout && gci < Cin_real) dw[((size_t)gco * Cin_real + gci) * T + t] = (float)r;
  }
}

static int wgrad_slices(int ntiles, int ncb, int nib) {
  const int blocks = ncb * nib;
  int s = (1024 + blocks - 1) / blocks;      // ~4 workgroups per CU in total
  if (s > ntiles) s = ntiles;
  if (s > 512) s = 512;
  return s < 1 ? 1 : s;
}

size_t wgrad_scratch_floats(ConvKind kind, int N, int Hout, int Wout, int Cin, int Cout) {
  const int T = kind == CONV1 ? 1 : 9, TH = kind == CONV3_S2 ? 2 : 4;
  const int ncb = (Cout + 63) / 64, nib = (Cin + 63) / 64, blocks = ncb * nib;
  const int ntiles = N * ((Wout + 15) / 16) * ((Hout + TH - 1) / TH);
  size_t need = (size_t)wgrad_slices(ntiles, ncb, nib) * blocks * T * 4096;
  if (kind == CONV3_S1 || kind == CONV3_UP) {
    // the 8-wave f16x3 forms (wgrad_h8_plan): up to N * (k0 + 1) <= 512 image-aligned slices, plus their dy column sums
    int k0 = 512 / (blocks * N);
    if (k0 < 1) k0 = 1;
    long ns = (long)N * (k0 + 1);
    if (ns > 512) ns = (long)N * k0;
    if (ns > 512) ns = 512;
    if (ns < 512 / blocks) ns = 512 / blocks;
    need = std::max(need, (size_t)ns * blocks * T * 4096 + (size_t)ns * ncb * 64);
  }
  return need;
}

template <int KS, int STRIDE, bool UP>
static hipError_t launch_wgrad_t(const WgradParams& p, hipStream_t s) {
  using Cfg = WgCfg<KS, STRIDE, UP>;
  const int Cin = p.C0 + p.C1;
  const int ncb = (p.Cout + 63) / 64, nib = (Cin + 63) / 64;
  const int ntiles = p.N * ((p.Wout + Cfg::TW - 1) / Cfg::TW) * ((p.Hout + Cfg::TH - 1) / Cfg::TH);
  const int ns = wgrad_slices(ntiles, ncb, nib);
  hipLaunchKernelGGL((wgrad_kernel<KS, STRIDE, UP>), dim3(ns * ncb * nib), dim3(256), (size_t)Cfg::LDS_FLOATS * sizeof(float), s, p, ns,
                     ncb, nib);
  const size_t total = (size_t)p.Cout * p.Cin_real * Cfg::T;
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)(ncb * nib * Cfg::T * 64)), dim3(256), 0, s, p.scratch, p.dw, p.Cout, p.Cin_real,
                     Cfg::T, ns, ncb, nib, total);
  return hipGetLastError();
}

hipError_t launch_wgrad(ConvKind kind, const WgradParams& p, hipStream_t s) {
  if ((p.C0 & 3) || (p.C1 & 3) || (p.Cout_s & 3)) return hipErrorInvalidValue;
  switch (kind) {
    case CONV3_S1: return launch_wgrad_t<3, 1, false>(p, s);
    case CONV3_S2: return launch_wgrad_t<3, 2, false>(p, s);
    case CONV3_UP: return launch_wgrad_t<3, 1, true>(p, s);
    case CONV1: return launch_wgrad_t<1, 1, false>(p, s);
  }
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------
// convolution weight gradient, split-f16 ("f16x3") form: v_mfma_f32_32x32x16_f16, three MFMAs per product
// (hi*hi + hi*lo + lo*hi of dy = dh + dl and a = ah + al), fp32 accumulation, fp64 fold of the slices.
// ---------------------------------------------------------------------------
// Pixels are the K dimension and both operands live in HBM pixel-major (NHWC): the MFMA wants, per lane, 8 consecutive
// PIXELS of one channel.  The tiles are staged as plain [pixel][64 channels] f16 images (hi plane, lo plane; coalesced
// loads, 8-byte LDS writes) and read with gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group fetches a
// 4-pixel x 16-channel block and every lane receives one channel's 4 pixels), two reads per operand half.  A tap only
// changes the FIRST ROW of the B block, so there is no alignment problem and no shifted copy.  Rows are 128 B; the two
// 64-byte halves of a row are swapped on rows with bit 1 set, which makes the 4-row blocks bank-conflict free.
typedef _Float16 th8 __attribute__((ext_vector_type(8)));
typedef short ts4 __attribute__((ext_vector_type(4)));

template <int KS, int STRIDE, bool UP>
struct WgHCfg {
  static constexpr int TH = STRIDE == 2 ? 2 : 4, TW = 16, T = KS * KS;
  static constexpr int HH = (TH - 1) * STRIDE + KS, HWD = (TW - 1) * STRIDE + KS, NPIX = HH * HWD;
  static constexpr int PLANE_DY = TH * TW * 128, PLANE_IN = NPIX * 128;     // bytes per plane
  static constexpr int LDS_BYTES = 2 * PLANE_DY + 2 * PLANE_IN;
};

__device__ __forceinline__ int tr_img_off(int row, int col) {   // byte offset of (pixel row, channel col) inside a plane
  return row * 128 + ((col ^ (((row >> 1) & 1) << 5)) << 1);
}

__device__ __forceinline__ th8 tr_frag(const unsigned char* plane, int row0, int row_step, int col) {
  // 8 k-values of this lane: rows row0 + {0..3} * row_step (first read) and row0 + {4..7} * row_step (second read);
  // lane 4q+p of its 16-lane group supplies the address of row q, columns col .. col+3 (col already includes 4p)
  typedef ts4 __attribute__((address_space(3))) * lds_ts4;
  const int q = (threadIdx.x >> 2) & 3;
  const ts4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ts4)(plane + tr_img_off(row0 + q * row_step, col)));
  const ts4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ts4)(plane + tr_img_off(row0 + (4 + q) * row_step, col)));
  typedef short ts8 __attribute__((ext_vector_type(8)));
  const ts8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(th8, v);
}

template <int KS, int STRIDE, bool UP>
__global__ void __launch_bounds__(256, 2) wgrad_h_kernel(const WgradParams p, const int nslices, const int ncb, const int nib) {
  using Cfg = WgHCfg<KS, STRIDE, UP>;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, T = Cfg::T, HWD = Cfg::HWD, NPIX = Cfg::NPIX, PAD = KS / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char wsh[];
  unsigned char* sDyH = wsh;
  unsigned char* sDyL = wsh + Cfg::PLANE_DY;
  unsigned char* sInH = wsh + 2 * Cfg::PLANE_DY;
  unsigned char* sInL = sInH + Cfg::PLANE_IN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wc = wave & 1, wi = wave >> 1;
  int b = blockIdx.x;
  const int sl = b % nslices;  b /= nslices;
  const int ib = b % nib;  b /= nib;
  const int cb = b;
  const int co0 = cb * 64, ci0 = ib * 64;
  const int Cin = p.C0 + p.C1;
  const int tilesX = (p.Wout + TW - 1) / TW, tilesY = (p.Hout + TH - 1) / TH;
  const int ntiles = p.N * tilesX * tilesY;
  const int t0 = (int)((long)sl * ntiles / nslices), t1 = (int)((long)(sl + 1) * ntiles / nslices);
  const int Hsrc = UP ? p.Hout : p.Hin, Wsrc = UP ? p.Wout : p.Win;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // ---- register prefetch of the next tile (as in wgrad_kernel) ----
  constexpr int NDY = TH * TW * 16 / 256, NIN = (NPIX * 16 + 255) / 256;
  const int q4 = tid & 15, prow = tid >> 4;
  f32x4 rdy[NDY], rin[NIN], rsc = {1.f, 1.f, 1.f, 1.f}, rsh = {0.f, 0.f, 0.f, 0.f};
  unsigned rmask[NIN];
  bool rok[NIN];
  auto prefetch = [&](int tile) {
    int tt = tile;
    const int tx = tt % tilesX;  tt /= tilesX;
    const int ty = tt % tilesY;
    const int n = tt / tilesY;
    const int oy0 = ty * TH, ox0 = tx * TW;
#pragma unroll
    for (int i = 0; i < NDY; ++i) {
      const int px = i * 16 + prow;
      const int oy = oy0 + px / TW, ox = ox0 + px % TW, co = co0 + q4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy < p.Hout && ox < p.Wout && co < p.Cout_s)
        v = *reinterpret_cast<const f32x4*>(p.dy + ((size_t)(n * p.Hout + oy) * p.Wout + ox) * p.Cout_s + co);
      rdy[i] = v;
    }
    const int c = ci0 + q4 * 4;
    if (p.gn_scale && c < Cin) {
      rsc = *reinterpret_cast<const f32x4*>(p.gn_scale + (size_t)n * Cin + c);
      rsh = *reinterpret_cast<const f32x4*>(p.gn_shift + (size_t)n * Cin + c);
    }
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int hp = i * 16 + prow;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      unsigned m = 0x01010101u;
      bool ok = false;
      if (hp < NPIX) {
        const int hy = hp / HWD, hx = hp % HWD;
        const int iy = oy0 * STRIDE - PAD + hy, ix = ox0 * STRIDE - PAD + hx;
        if (iy >= 0 && iy < Hsrc && ix >= 0 && ix < Wsrc && c < Cin) {
          const int sy = UP ? (iy >> 1) : iy, sx = UP ? (ix >> 1) : ix;
          const float* xs; int Cs, cc;
          if (c < p.C0) { xs = p.x0; Cs = p.C0; cc = c; } else { xs = p.x1; Cs = p.C1; cc = c - p.C0; }
          const size_t o = ((size_t)(n * p.Hin + sy) * p.Win + sx) * Cs + cc;
          v = *reinterpret_cast<const f32x4*>(xs + o);
          if (p.drop_mask) m = *reinterpret_cast<const unsigned*>(p.drop_mask + o);
          ok = true;
        }
      }
      rin[i] = v;
      rmask[i] = m;
      rok[i] = ok;
    }
  };
  auto put_split = [&](unsigned char* ph, unsigned char* pl, int row, f32x4 v) {
    // hi = rn_f16(clamp(v)), lo = rn_f16(v - hi): 22 mantissa bits (fdsr_conv_h.hip)
    typedef _Float16 h4t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
    const h4t hi = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    const h4t lo = {(_Float16)(v[0] - (float)hi[0]), (_Float16)(v[1] - (float)hi[1]), (_Float16)(v[2] - (float)hi[2]),
                    (_Float16)(v[3] - (float)hi[3])};
    const int off = tr_img_off(row, q4 * 4);
    *reinterpret_cast<h4t*>(ph + off) = hi;
    *reinterpret_cast<h4t*>(pl + off) = lo;
  };
  auto store = [&]() {
#pragma unroll
    for (int i = 0; i < NDY; ++i) put_split(sDyH, sDyL, i * 16 + prow, rdy[i]);
#pragma unroll
    for (int i = 0; i < NIN; ++i) {
      const int hp = i * 16 + prow;
      if (hp >= NPIX) continue;
      f32x4 v = rin[i];
      if (rok[i] && p.gn_scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u = fmaf(v[e], rsc[e], rsh[e]);
          v[e] = p.gn_plain ? u : u * sigmoid_f(u);
        }
        if (p.drop_mask) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ((rmask[i] >> (8 * e)) & 0xffu) ? v[e] * p.drop_scale : 0.f;
        }
      }
      put_split(sInH, sInL, hp, v);
    }
  };

  // ---- operand addressing: lane -> (16-lane group g, i = 4q + p); channel column = 32 * wave half + 16 * (g & 1) + 4p,
  //      first pixel of the 8-pixel run = 8 * (g >> 1)
  const int g = lane >> 4, pp4 = lane & 3;
  const int colA = wc * 32 + 16 * (g & 1) + 4 * pp4, colB = wi * 32 + 16 * (g & 1) + 4 * pp4;
  const int k0 = 8 * (g >> 1);

  if (t0 < t1) prefetch(t0);
  for (int tile = t0; tile < t1; ++tile) {
    __syncthreads();
    store();
    __syncthreads();
    if (tile + 1 < t1) prefetch(tile + 1);
#pragma unroll 1
    for (int y = 0; y < TH; ++y) {                          // one K-step = one 16-pixel tile row
      const th8 ah = tr_frag(sDyH, y * TW + k0, 1, colA);
      const th8 al = tr_frag(sDyL, y * TW + k0, 1, colA);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const int rb = (y * STRIDE + t / KS) * HWD + k0 * STRIDE + (t % KS);
        const th8 bh = tr_frag(sInH, rb, STRIDE, colB);
        const th8 bl = tr_frag(sInL, rb, STRIDE, colB);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
      }
    }
  }
  const int r31 = lane & 31, kh = lane >> 5;
  float* dst = p.scratch + ((((size_t)sl * ncb + cb) * nib + ib) * T) * 4096;
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
      dst[(size_t)t * 4096 + (wc * 32 + row) * 64 + wi * 32 + r31] = acc[t][i];
    }
}

// ---- the stride-1 form used for almost all of the work: 8 waves, double-buffered tiles ---------------------------------
// One workgroup per CU (two waves per SIMD): waves 0-3 and 4-7 take the upper and lower four rows of an 8x16-pixel tile of the
// same 64 (co) x 64 (ci) block, so a tile is 128 K-values per barrier.  The NEXT tile is fetched in four pieces, one per tile
// row: its global loads are issued before the 27 MFMAs of the row and split / written to the other LDS buffer after them (12
// staging registers instead of 44, one barrier per tile, the staging VALU of one wave under the MFMAs of its SIMD partner).
// The B fragments of tap t+1 are read while the MFMAs of tap t run.  The two half-tile accumulators are added through LDS at
// the end (rows 0-3 + rows 4-7, a fixed order), so the scratch slices and wgrad_fold_kernel are those of the other forms.
template <int KS, bool UP>
struct WgH8Cfg {
  static constexpr int TH = 8, TW = 16, T = KS * KS;
  static constexpr int HH = TH - 1 + KS, HWD = TW - 1 + KS, NPIX = HH * HWD;
  static constexpr int PLANE_DY = TH * TW * 128, PLANE_IN = NPIX * 128;     // bytes per plane
  static constexpr int BUF = 2 * PLANE_DY + 2 * PLANE_IN;
  static constexpr int RED = T * 4096 * 4;                                  // the final half-tile reduction
  static constexpr int LDS_BYTES = 2 * BUF > RED ? 2 * BUF : RED;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ th8 tr_frag_at(const unsigned char* a) {   // two transposing reads: rows +0..3 and +4..7 of this lane's block
  typedef ts4 __attribute__((address_space(3))) * lds_ts4;
  typedef short ts8 __attribute__((ext_vector_type(8)));
  const ts4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ts4)(a));
  const ts4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ts4)(a + 4 * 128));   // row + 4: same swizzle
  const ts8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(th8, v);
}

template <int KS, bool UP>
__global__ void __launch_bounds__(512, 2) wgrad_h8_kernel(const WgradParams p, const int nslices, const int ncb, const int nib) {
  using Cfg = WgH8Cfg<KS, UP>;
  constexpr int TH = Cfg::TH, TW = Cfg::TW, T = Cfg::T, HH = Cfg::HH, HWD = Cfg::HWD, PAD = KS / 2;
  static_assert(KS == 3, "the half-tile staging below is laid out for the 3x3 halo");
  static_assert((2 * HWD) % 4 == 0 && (4 * HWD) % 4 == 0, "the LDS swizzle must not depend on the tile row");
  extern __shared__ __attribute__((aligned(16))) unsigned char wsh8[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), wc = wave & 1, wi = (wave >> 1) & 1;
  int b = blockIdx.x;
  const int sl = b % nslices;  b /= nslices;
  const int ib = b % nib;  b /= nib;
  const int cb = b;
  const int co0 = cb * 64, ci0 = ib * 64;
  const int Cin = p.C0 + p.C1;
  const int tilesX = (p.Wout + TW - 1) / TW, tilesY = (p.Hout + TH - 1) / TH;
  const int ntiles = p.N * tilesX * tilesY;
  const int t0 = (int)((long)sl * ntiles / nslices), t1 = (int)((long)(sl + 1) * ntiles / nslices);
  const int Hsrc = UP ? p.Hout : p.Hin, Wsrc = UP ? p.Wout : p.Win;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // ---- staging: thread -> channel quad q4 and pixel (ry, rx) of a 2-row x 16-column strip.  The next tile is fetched in two
  // halves; half h = dy rows 4h+ry and 4h+2+ry, halo rows 4h+ry and 4h+2+ry (columns 0..15), plus the halo rows 8+ry (h = 0) or
  // the two right-hand halo columns (h = 1, threads with prow < 2*HH): every index is a shift or a mask and every LDS address is
  // a lane base + an immediate.  Global addresses are tensor base (uniform) + a 32-bit byte offset = uniform tile offset + the
  // lane's offset inside a tile (modular: the halo origin may lie before the image); lanes outside the image read the tensor's
  // first element instead and are zeroed by the clamp (lim = 0).  The launcher guarantees < 4 GiB tensors.
  const int q4 = tid & 15, prow = tid >> 4, ry = prow >> 4, rx = prow & 15;
  const int cdy = co0 + q4 * 4, cin = ci0 + q4 * 4;
  const bool src0 = ci0 < p.C0;                                      // the 64-channel block lies in one concat source (launcher)
  const char* xs = reinterpret_cast<const char*>(src0 ? p.x0 : p.x1);
  const char* dys = reinterpret_cast<const char*>(p.dy);
  const int Cs = src0 ? p.C0 : p.C1, cc = (src0 ? ci0 : ci0 - p.C0) + q4 * 4;
  const bool gn = p.gn_scale != nullptr, cin_ok = cin < Cin, cdy_ok = cdy < p.Cout_s;
  const bool edge_lane = prow < 2 * HH;
  const int ehy = prow >> 1, ehx = TW + (prow & 1);
  auto lsrc = [&](int h) { return UP ? ((h - 1) >> 1) : h; };         // halo coordinate -> source coordinate, lane part
  const unsigned o_dy = (unsigned)((ry * p.Wout + rx) * p.Cout_s + cdy) * 4u;
  const unsigned o_main = (unsigned)((lsrc(ry) * p.Win + lsrc(rx)) * Cs + cc) * 4u;
  const unsigned o_edge = (unsigned)((lsrc(ehy) * p.Win + lsrc(ehx)) * Cs + cc) * 4u;
  const int st_dy = tr_img_off(prow, q4 * 4);                        // + k * 32 rows
  const int st_in = tr_img_off(ry * HWD + rx, q4 * 4);               // + kk * 2 * HWD rows
  const int st_edge = tr_img_off(ehy * HWD + ehx, q4 * 4);
  const unsigned row_dy = (unsigned)(p.Wout * p.Cout_s) * 4u, row_in = (unsigned)(p.Win * Cs) * 4u;   // bytes per image row
  constexpr int NPV = 5;            // registers of one half: 2 dy, 2 halo strips, 1 extra
  f32x4 pv[NPV], nsc = {1.f, 1.f, 1.f, 1.f}, nsh = {0.f, 0.f, 0.f, 0.f};
  unsigned pm[NPV];
  bool pok[NPV];
  int toy = 0, tox = 0;             // the tile being fetched
  unsigned ub_dy = 0, ub_in = 0;    // its uniform byte offsets: dy tile origin, halo origin
  auto set_tile = [&](int tile) {
    int tt = tile;
    tox = (tt % tilesX) * TW;  tt /= tilesX;
    toy = (tt % tilesY) * TH;
    const int tn = tt / tilesY;
    ub_dy = (unsigned)(((tn * p.Hout + toy) * p.Wout + tox) * p.Cout_s) * 4u;
    ub_in = (unsigned)(((tn * p.Hin + (UP ? toy / 2 : toy - PAD)) * p.Win + (UP ? tox / 2 : tox - PAD)) * Cs) * 4u;
    if (gn && cin_ok) {
      nsc = *reinterpret_cast<const f32x4*>(p.gn_scale + (size_t)tn * Cin + cin);
      nsh = *reinterpret_cast<const f32x4*>(p.gn_shift + (size_t)tn * Cin + cin);
    }
  };
  auto load_dy = [&](int u, int k) {                                  // dy rows 2k + ry
    const bool ok = toy + 2 * k + ry < p.Hout && tox + rx < p.Wout && cdy_ok;
    const unsigned off = ok ? ub_dy + o_dy + 2 * k * row_dy : 0u;
    pv[u] = *reinterpret_cast<const f32x4*>(dys + off);
    pok[u] = ok;
  };
  auto load_in = [&](int u, unsigned o, int rows, int hy, int hx, bool lane_ok) {
    const bool ok = lane_ok && (unsigned)(toy - PAD + hy) < (unsigned)Hsrc && (unsigned)(tox - PAD + hx) < (unsigned)Wsrc && cin_ok;
    const unsigned off = ok ? ub_in + o + rows * row_in : 0u;
    pv[u] = *reinterpret_cast<const f32x4*>(xs + off);
    if (p.drop_mask) pm[u] = *reinterpret_cast<const unsigned*>(p.drop_mask + (off >> 2));
    pok[u] = ok;
  };
  auto load_half = [&](int h) {
    load_dy(0, 2 * h);
    load_dy(1, 2 * h + 1);
    load_in(2, o_main, UP ? 2 * h : 4 * h, 4 * h + ry, rx, true);
    load_in(3, o_main, UP ? 2 * h + 1 : 4 * h + 2, 4 * h + 2 + ry, rx, true);
    if (h == 0) load_in(4, o_main, UP ? 4 : 8, 8 + ry, rx, true);
    else load_in(4, o_edge, 0, ehy, ehx, edge_lane);
  };
  // hi = rn_f16(clamp(v)), lo = rn_f16(v - hi) (22 mantissa bits, as fdsr_conv_h.hip): one packed convert and two
  // v_fma_mix per pair, lo = f16(fma(hi, -1, v)) rounded once -- bit-identical to the two-step form.  lim = 0 zero-pads.
  auto put_split = [&](unsigned char* ph, unsigned char* pl, f32x4 v, float lim) {
    typedef _Float16 h2t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -lim, lim);
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
    *reinterpret_cast<uint2*>(ph) = hi;
    *reinterpret_cast<uint2*>(pl) = lo;
  };
  auto put_in = [&](int u, unsigned char* inH, int off) {
    f32x4 v = pv[u];
    if (gn) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], nsc[e], nsh[e]);
      if (!p.gn_plain) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[e]));
      }
      if (p.drop_mask) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((pm[u] >> (8 * e)) & 0xffu) ? v[e] * p.drop_scale : 0.f;
      }
    }
    put_split(inH + off, inH + Cfg::PLANE_IN + off, v, pok[u] ? 65504.f : 0.f);   // lim = 0: the conv zero-pads the ACTIVATED tensor
  };
  auto store_half = [&](int h, unsigned char* buf) {
    unsigned char* inH = buf + 2 * Cfg::PLANE_DY;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int off = st_dy + (2 * h + u) * 32 * 128;
      put_split(buf + off, buf + Cfg::PLANE_DY + off, pv[u], pok[u] ? 65504.f : 0.f);
    }
    put_in(2, inH, st_in + (2 * h) * 2 * HWD * 128);
    put_in(3, inH, st_in + (2 * h + 1) * 2 * HWD * 128);
    if (h == 0) put_in(4, inH, st_in + 4 * 2 * HWD * 128);
    else if (edge_lane) put_in(4, inH, st_edge);
  };

  // ---- operand addressing: lane -> (16-lane group g, q = row inside the 4-row block, p = column quad); channel column =
  // 32 * wave half + 16 * (g & 1) + 4p, first pixel of the 8-pixel run = 8 * (g >> 1).  The swizzle bit of a read is bit 1 of
  // its LDS row = bit 1 of (c2 + q), c2 = the row's compile-time part mod 4: one lane base per c2, immediates for the rest.
  const int g = lane >> 4, pp4 = lane & 3, q = (lane >> 2) & 3;
  const int k0 = 8 * (g >> 1);
  const int colAb = (wc * 32 + 16 * (g & 1) + 4 * pp4) * 2, colBb = (wi * 32 + 16 * (g & 1) + 4 * pp4) * 2;
  const int baseA = (grp * 4 * TW + k0 + q) * 128 + (colAb ^ (((q >> 1) & 1) << 6));
  int baseB[4];
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) baseB[c2] = 2 * Cfg::PLANE_DY + (grp * 4 * HWD + k0 + q) * 128 + (colBb ^ ((((c2 + q) >> 1) & 1) << 6));

  auto mfma_row = [&](const unsigned char* cur, int y) {     // one K-step = one 16-pixel row of this half tile: 27 MFMAs
    const th8 ah = tr_frag_at(cur + baseA + y * TW * 128);
    const th8 al = tr_frag_at(cur + baseA + y * TW * 128 + Cfg::PLANE_DY);
    th8 bh[2], bl[2];
    bh[0] = tr_frag_at(cur + baseB[(y * HWD) & 3] + y * HWD * 128);
    bl[0] = tr_frag_at(cur + baseB[(y * HWD) & 3] + y * HWD * 128 + Cfg::PLANE_IN);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) {
        const int rr = (y + (t + 1) / KS) * HWD + (t + 1) % KS;      // compile-time part of the LDS row
        bh[(t + 1) & 1] = tr_frag_at(cur + baseB[rr & 3] + rr * 128);
        bl[(t + 1) & 1] = tr_frag_at(cur + baseB[rr & 3] + rr * 128 + Cfg::PLANE_IN);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t & 1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t & 1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t & 1], acc[t], 0, 0, 0);
    }
  };

  if (t0 < t1) {
    set_tile(t0);
#pragma unroll
    for (int h = 0; h < 2; ++h) { load_half(h); store_half(h, wsh8); }
  }
  __syncthreads();
  // The two waves of a SIMD (w and w + 4) run the same loop half a phase apart: waves 0-3 issue the MFMAs of two rows and then
  // split the half tile they fetched before them; waves 4-7 split first (a half fetched two rows earlier -- their first half of
  // the tile after next is fetched under the last two rows) and issue their MFMAs after, so one wave's VALU runs under the
  // other's MFMAs and every fetch has two MFMA rows of both waves to land (40 KB in flight per CU).
  if (grp == 0) {
    for (int tile = t0; tile < t1; ++tile) {
      const int curoff = ((tile - t0) & 1) ? Cfg::BUF : 0;
      const unsigned char* cur = wsh8 + curoff;
      unsigned char* nxt = wsh8 + (Cfg::BUF - curoff);
      const bool more = tile + 1 < t1;
      if (more) set_tile(tile + 1);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (more) load_half(h);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(cur, 2 * h);
        __builtin_amdgcn_sched_barrier(0);
        mfma_row(cur, 2 * h + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_half(h, nxt);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  } else {
    if (t0 + 1 < t1) { set_tile(t0 + 1); load_half(0); }
    for (int tile = t0; tile < t1; ++tile) {
      const int curoff = ((tile - t0) & 1) ? Cfg::BUF : 0;
      const unsigned char* cur = wsh8 + curoff;
      unsigned char* nxt = wsh8 + (Cfg::BUF - curoff);
      const bool more = tile + 1 < t1;
      if (more) { store_half(0, nxt); load_half(1); }
      __builtin_amdgcn_sched_barrier(0);
      mfma_row(cur, 0);
      __builtin_amdgcn_sched_barrier(0);
      mfma_row(cur, 1);
      __builtin_amdgcn_sched_barrier(0);
      if (more) store_half(1, nxt);
      if (tile + 2 < t1) { set_tile(tile + 2); load_half(0); }
      __builtin_amdgcn_sched_barrier(0);
      mfma_row(cur, 2);
      __builtin_amdgcn_sched_barrier(0);
      mfma_row(cur, 3);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
    }
  }

  // ---- rows 4-7 (waves 4-7) are added to rows 0-3 through LDS, then one slice goes to the scratch ----
  const int r31 = lane & 31, kh = lane >> 5;
  float* red = reinterpret_cast<float*>(wsh8);
  if (grp == 1) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
        red[t * 4096 + (wc * 32 + row) * 64 + wi * 32 + r31] = acc[t][i];
      }
  }
  __syncthreads();
  if (grp == 0) {
    float* dst = p.scratch + ((((size_t)sl * ncb + cb) * nib + ib) * T) * 4096;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
        const int o = t * 4096 + (wc * 32 + row) * 64 + wi * 32 + r31;
        dst[o] = acc[t][i] + red[o];
      }
  }
}

// ---- the same with the staging INSIDE the MFMA rows --------------------------------------------------------------------
// wgrad_h8_kernel's knock-outs say its split / activate / LDS-write stream costs as much as its MFMAs and overlaps them only
// partly: an in-order wave cannot issue its own VALU while it is blocked in a run of 27 MFMAs.  Here a tile row is ONE basic
// block: the loads of quarter-piece y+1, the 27 MFMAs of row y with their fragment reads, and the split of piece y (fetched a
// whole row earlier, so no wait) -- branch-free (GN / DROP are template parameters, the last tile re-fetches itself, the two
// right-hand halo columns go to a dummy LDS slot on the lanes that do not own one) -- and sched_group_barrier asks for
// 1 MFMA : 5 VALU : 2 LDS reads per gap.  Two piece register sets (y & 1); both wave groups run the same code.
#ifndef WG_IL_VALU
#define WG_IL_VALU 4
#endif
#ifndef WG_IL_DSR
#define WG_IL_DSR 0
#endif
template <bool UP, bool GN, bool DROP>
__global__ void __launch_bounds__(512, 2) wgrad_h8i_kernel(const WgradParams p, const int nslices, const int ncb, const int nib) {
  using Cfg = WgH8Cfg<3, UP>;
  constexpr int KS = 3, TH = Cfg::TH, TW = Cfg::TW, T = Cfg::T, HH = Cfg::HH, HWD = Cfg::HWD, PAD = 1;
  constexpr int DUMMY = 2 * Cfg::BUF;                      // 8 spare bytes per lane-quad above the two buffers (hi), +2 KB (lo)
  static_assert(2 * Cfg::BUF + 4096 <= 160 * 1024, "room for the dummy slot");
  extern __shared__ __attribute__((aligned(16))) unsigned char wsh8[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), wc = wave & 1, wi = (wave >> 1) & 1;
  int b = blockIdx.x;
  const int sl = b % nslices;  b /= nslices;
  const int ib = b % nib;  b /= nib;
  const int cb = b;
  const int co0 = cb * 64, ci0 = ib * 64;
  const int Cin = p.C0 + p.C1;
  const int tilesX = (p.Wout + TW - 1) / TW, tilesY = (p.Hout + TH - 1) / TH;
  const int ntiles = p.N * tilesX * tilesY;
  const int t0 = (int)((long)sl * ntiles / nslices), t1 = (int)((long)(sl + 1) * ntiles / nslices);
  const int Hsrc = UP ? p.Hout : p.Hin, Wsrc = UP ? p.Wout : p.Win;

  f32x16 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  if (t0 >= t1) return;                                    // (never: slices <= tiles)

  const int q4 = tid & 15, prow = tid >> 4, ry = prow >> 4, rx = prow & 15;
  const int cdy = co0 + q4 * 4, cin = ci0 + q4 * 4;
  const bool src0 = ci0 < p.C0;
  const char* xs = reinterpret_cast<const char*>(src0 ? p.x0 : p.x1);
  const char* dys = reinterpret_cast<const char*>(p.dy);
  const int Cs = src0 ? p.C0 : p.C1, cc = (src0 ? ci0 : ci0 - p.C0) + q4 * 4;
  const bool cin_ok = cin < Cin, cdy_ok = cdy < p.Cout_s;
  const bool edge_lane = prow < 2 * HH;
  const int ehy = prow >> 1, ehx = TW + (prow & 1);
  auto lsrc = [&](int h) { return UP ? ((h - 1) >> 1) : h; };
  const unsigned o_dy = (unsigned)((ry * p.Wout + rx) * p.Cout_s + cdy) * 4u;
  const unsigned o_main = (unsigned)((lsrc(ry) * p.Win + lsrc(rx)) * Cs + cc) * 4u;
  const unsigned o_edge = (unsigned)((lsrc(ehy) * p.Win + lsrc(ehx)) * Cs + cc) * 4u;
  const int st_dy = tr_img_off(prow, q4 * 4);
  const int st_in = 2 * Cfg::PLANE_DY + tr_img_off(ry * HWD + rx, q4 * 4);
  const int st_edge = 2 * Cfg::PLANE_DY + tr_img_off(ehy * HWD + ehx, q4 * 4);
  const unsigned row_dy = (unsigned)(p.Wout * p.Cout_s) * 4u, row_in = (unsigned)(p.Win * Cs) * 4u;
  f32x4 pv[2][3], nsc = {1.f, 1.f, 1.f, 1.f}, nsh = {0.f, 0.f, 0.f, 0.f};
  f32x4 cs = {0.f, 0.f, 0.f, 0.f};   // this thread's share of the column sums of dy (bias / noise-shift gradients), real tiles only
  bool cs_real = true;
  unsigned pm[2][3];
  bool pok[2][3];
  int toy = 0, tox = 0, tn = 0;
  unsigned ub_dy = 0, ub_in = 0;
  auto set_tile = [&](int tile) {
    int tt = tile;
    tox = (tt % tilesX) * TW;  tt /= tilesX;
    toy = (tt % tilesY) * TH;
    tn = tt / tilesY;
    ub_dy = (unsigned)(((tn * p.Hout + toy) * p.Wout + tox) * p.Cout_s) * 4u;
    ub_in = (unsigned)(((tn * p.Hin + (UP ? toy / 2 : toy - PAD)) * p.Win + (UP ? tox / 2 : tox - PAD)) * Cs) * 4u;
  };
  auto load_gn = [&]() {
    if (GN) {
      const int c = cin_ok ? cin : 0;
      nsc = *reinterpret_cast<const f32x4*>(p.gn_scale + (size_t)tn * Cin + c);
      nsh = *reinterpret_cast<const f32x4*>(p.gn_shift + (size_t)tn * Cin + c);
    }
  };
  auto load_in = [&](int set, int u, unsigned o, int rows, int hy, int hx, bool lane_ok) {
    const bool ok = lane_ok && (unsigned)(toy - PAD + hy) < (unsigned)Hsrc && (unsigned)(tox - PAD + hx) < (unsigned)Wsrc && cin_ok;
    const unsigned off = ok ? ub_in + o + rows * row_in : 0u;
    pv[set][u] = *reinterpret_cast<const f32x4*>(xs + off);
    if (DROP) pm[set][u] = *reinterpret_cast<const unsigned*>(p.drop_mask + (off >> 2));
    pok[set][u] = ok;
  };
  auto load_piece = [&](int set, int j) {                  // quarter j: dy rows 2j+ry, halo rows 2j+ry, + rows 8+ry (j=0) / edge (j=1)
    {
      const bool ok = toy + 2 * j + ry < p.Hout && tox + rx < p.Wout && cdy_ok;
      const unsigned off = ok ? ub_dy + o_dy + 2 * j * row_dy : 0u;
      pv[set][0] = *reinterpret_cast<const f32x4*>(dys + off);
      pok[set][0] = ok;
    }
    load_in(set, 1, o_main, UP ? j : 2 * j, 2 * j + ry, rx, true);
    if (j == 0) load_in(set, 2, o_main, UP ? 4 : 8, 8 + ry, rx, true);
    if (j == 1) load_in(set, 2, o_edge, 0, ehy, ehx, edge_lane);
  };
  auto put_split = [&](unsigned char* ph, unsigned char* pl, f32x4 v, float lim) {
    typedef _Float16 h2t __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(v[e], -lim, lim);
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
    *reinterpret_cast<uint2*>(ph) = hi;
    *reinterpret_cast<uint2*>(pl) = lo;
  };
  auto put_in = [&](int set, int u, unsigned char* buf, int off, int lo_off) {
    f32x4 v = pv[set][u];
    if (GN) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], nsc[e], nsh[e]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = v[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-v[e]));
      if (DROP) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((pm[set][u] >> (8 * e)) & 0xffu) ? v[e] * p.drop_scale : 0.f;
      }
    }
    put_split(buf + off, buf + off + lo_off, v, pok[set][u] ? 65504.f : 0.f);
  };
  auto store_piece = [&](int set, int j, unsigned char* buf) {
    {
      const bool add = pok[set][0] && cs_real;
#pragma unroll
      for (int e = 0; e < 4; ++e) cs[e] += add ? pv[set][0][e] : 0.f;
    }
    put_split(buf + st_dy + j * 32 * 128, buf + Cfg::PLANE_DY + st_dy + j * 32 * 128, pv[set][0], pok[set][0] ? 65504.f : 0.f);
    put_in(set, 1, buf, st_in + j * 2 * HWD * 128, Cfg::PLANE_IN);
    if (j == 0) put_in(set, 2, buf, st_in + 4 * 2 * HWD * 128, Cfg::PLANE_IN);
    if (j == 1) {                                           // lanes without a halo-edge pixel write a dummy slot (no branch)
      unsigned char* base = edge_lane ? buf + st_edge : wsh8 + DUMMY + (tid & 255) * 8;
      put_in(set, 2, base, 0, edge_lane ? Cfg::PLANE_IN : 2048);
    }
  };

  const int g = lane >> 4, pp4 = lane & 3, q = (lane >> 2) & 3;
  const int k0 = 8 * (g >> 1);
  const int colAb = (wc * 32 + 16 * (g & 1) + 4 * pp4) * 2, colBb = (wi * 32 + 16 * (g & 1) + 4 * pp4) * 2;
  const int baseA = (grp * 4 * TW + k0 + q) * 128 + (colAb ^ (((q >> 1) & 1) << 6));
  int baseB[4];
#pragma unroll
  for (int c2 = 0; c2 < 4; ++c2) baseB[c2] = 2 * Cfg::PLANE_DY + (grp * 4 * HWD + k0 + q) * 128 + (colBb ^ ((((c2 + q) >> 1) & 1) << 6));

  // one tile row: MFMAs + fragment reads; the caller puts the staging of a piece into the same block
  auto mfma_row = [&](const unsigned char* cur, int y) {
    const th8 ah = tr_frag_at(cur + baseA + y * TW * 128);
    const th8 al = tr_frag_at(cur + baseA + y * TW * 128 + Cfg::PLANE_DY);
    th8 bh[2], bl[2];
    bh[0] = tr_frag_at(cur + baseB[(y * HWD) & 3] + y * HWD * 128);
    bl[0] = tr_frag_at(cur + baseB[(y * HWD) & 3] + y * HWD * 128 + Cfg::PLANE_IN);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) {
        const int rr = (y + (t + 1) / KS) * HWD + (t + 1) % KS;
        bh[(t + 1) & 1] = tr_frag_at(cur + baseB[rr & 3] + rr * 128);
        bl[(t + 1) & 1] = tr_frag_at(cur + baseB[rr & 3] + rr * 128 + Cfg::PLANE_IN);
      }
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t & 1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t & 1], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t & 1], acc[t], 0, 0, 0);
    }
  };
  auto interleave = [&]() {                                 // the shape asked of the scheduler for a row block
#pragma unroll
    for (int i = 0; i < 27; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // 1 MFMA
#if WG_IL_DSR
      __builtin_amdgcn_sched_group_barrier(0x100, WG_IL_DSR, 0);
#endif
      __builtin_amdgcn_sched_group_barrier(0x002, WG_IL_VALU, 0);    // VALU per MFMA gap
    }
  };

  // prologue: tile t0 staged directly, piece 0 of the next fetch tile in flight
  set_tile(t0);
  load_gn();
#pragma unroll
  for (int j = 0; j < 4; ++j) { load_piece(j & 1, j); store_piece(j & 1, j, wsh8); }
  {
    const int f = t0 + 1 < t1 ? t0 + 1 : t1 - 1;
    set_tile(f);
    load_gn();
    load_piece(0, 0);
  }
  __syncthreads();
  for (int tile = t0; tile < t1; ++tile) {
    const int curoff = ((tile - t0) & 1) ? Cfg::BUF : 0;
    const unsigned char* cur = wsh8 + curoff;
    unsigned char* nxt = wsh8 + (Cfg::BUF - curoff);
    cs_real = tile + 1 < t1;                                 // the last tile re-fetches itself: not summed twice
#pragma unroll
    for (int y = 0; y < 4; ++y) {
      // fetch: piece y+1 of tile+1, or (y = 3) piece 0 of tile+2; convert: piece y of tile+1 (fetched during the previous row)
      if (y == 3) set_tile(tile + 2 < t1 ? tile + 2 : t1 - 1);
      load_piece((y + 1) & 1, (y + 1) & 3);
      mfma_row(cur, y);
      store_piece(y & 1, y, nxt);
      if (y == 3) load_gn();                                 // after the last use of this tile's scale / shift
      interleave();
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- column sums of dy over this slice (one image or part of one): lanes l, l^16, l^32, l^48 share the channel quad ----
  if (p.colsum_part && ib == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cs[e] += __shfl_xor(cs[e], 16, 64);
      cs[e] += __shfl_xor(cs[e], 32, 64);
    }
    float* cred = reinterpret_cast<float*>(wsh8);           // [wave][64 channels]
    if (lane < 16) *reinterpret_cast<f32x4*>(cred + wave * 64 + lane * 4) = cs;
    __syncthreads();
    if (tid < 64) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += cred[w * 64 + tid];
      p.colsum_part[((size_t)sl * ncb + cb) * 64 + tid] = a;
    }
    __syncthreads();
  }
  const int r31 = lane & 31, kh = lane >> 5;
  float* red = reinterpret_cast<float*>(wsh8);
  if (grp == 1) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
        red[t * 4096 + (wc * 32 + row) * 64 + wi * 32 + r31] = acc[t][i];
      }
  }
  __syncthreads();
  if (grp == 0) {
    float* dst = p.scratch + ((((size_t)sl * ncb + cb) * nib + ib) * T) * 4096;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
        const int o = t * 4096 + (wc * 32 + row) * 64 + wi * 32 + r31;
        dst[o] = acc[t][i] + red[o];
      }
  }
}

// How a stride-1 3x3 (or folded-upsample) f16x3 weight gradient runs.  `colsum`: the in-row kernel also produces the column sums
// of dy; that wants every slice inside ONE image (ns = N * k), so that S[n][c] is a sum of whole slices.
struct H8Plan { bool h8, inrow, colsum; int ns, k; };
static H8Plan wgrad_h8_plan(ConvKind kind, const WgradParams& p) {
  const bool four_wave = g_tun.wgrad_form == 1;   // A/B options (fdsr_debug_option): the 4-wave single-buffer form everywhere,
  const bool plain8 = g_tun.wgrad_form == 2;      // the 8-wave form without the in-row interleave
  const bool no_colsum = !g_tun.wgrad_colsum;
  H8Plan r{false, false, false, 1, 0};
  if (kind != CONV3_S1 && kind != CONV3_UP) return r;
  // the 8-wave forms want a 64-channel block inside one concat source and 32-bit byte offsets; the 4-wave form takes the rest
  const size_t in_px = (size_t)p.N * p.Hin * p.Win, out_px = (size_t)p.N * p.Hout * p.Wout;
  const bool seam = (p.C1 > 0 && (p.C0 & 63) != 0) || in_px * (size_t)(p.C0 > p.C1 ? p.C0 : p.C1) * 4 >= (size_t)g_tun.wgrad_big_bytes ||
                    out_px * (size_t)p.Cout_s * 4 >= (size_t)g_tun.wgrad_big_bytes;
  if (four_wave || seam) return r;
  r.h8 = true;
  r.inrow = !plain8 && !p.gn_plain;
  const int ncb = (p.Cout + 63) / 64, nib = (p.C0 + p.C1 + 63) / 64, blocks = ncb * nib;
  const int tpi = ((p.Wout + 15) / 16) * ((p.Hout + 7) / 8), ntiles = p.N * tpi;
  int s = 512 / blocks;                       // <= two rounds of one workgroup per CU, never a straggler third
  if (s > ntiles) s = ntiles;
  r.ns = s < 1 ? 1 : s;
  if (r.inrow && !no_colsum && p.Cout_s <= ncb * 64) {
    // slices per image: k0 or k0 + 1, whichever fills whole rounds of 256 workgroups better; within [1, tiles per image]
    int k0 = 512 / (blocks * p.N);
    if (k0 < 1) k0 = 1;
    int best = 0;
    double beff = -1.0;
    for (int k = k0; k <= k0 + 1; ++k) {
      if (k > tpi || p.N * k > 512) continue;
      const long w = (long)blocks * p.N * k;
      const double eff = (double)w / (double)(((w + 255) / 256) * 256);
      if (eff > beff + 1e-9) { beff = eff; best = k; }
    }
    if (best > 0) { r.colsum = true; r.k = best; r.ns = p.N * best; }
  }
  return r;
}

bool wgrad_h_fuses_colsum(ConvKind kind, const WgradParams& p) { return wgrad_h8_plan(kind, p).colsum; }

// S[n][c] = the slices of image n added in slice order
__global__ void __launch_bounds__(256) colsum_slices_kernel(const float* __restrict__ part, float* __restrict__ S, int K, int ncb, int k) {
  const int c = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (c >= K) return;
  double a = 0.0;
  for (int j0 = 0; j0 < k; j0 += 8) {   // (eight slices per trip, in flight together: as sum_rows_kernel)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = part[((size_t)(n * k + min(j0 + u, k - 1)) * ncb + (c >> 6)) * 64 + (c & 63)];
#pragma unroll
    for (int u = 0; u < 8; ++u) a += j0 + u < k ? (double)v[u] : 0.0;
  }
  S[(size_t)n * K + c] = (float)a;
}

template <int KS, bool UP>
static hipError_t launch_wgrad_h8_t(const WgradParams& p0, const H8Plan& pl, hipStream_t s) {
  using Cfg = WgH8Cfg<KS, UP>;
  WgradParams p = p0;
  const int Cin = p.C0 + p.C1;
  const int ncb = (p.Cout + 63) / 64, nib = (Cin + 63) / 64;
  const int ns = pl.ns;
  p.colsum_part = pl.colsum && p.colsum ? p.scratch + (size_t)ns * ncb * nib * Cfg::T * 4096 : nullptr;
  if (pl.inrow) {
    const size_t lds = (size_t)2 * Cfg::BUF + 4096;
    const dim3 grid(ns * ncb * nib), block(512);
    if (!p.gn_scale) hipLaunchKernelGGL((wgrad_h8i_kernel<UP, false, false>), grid, block, lds, s, p, ns, ncb, nib);
    else if (!p.drop_mask) hipLaunchKernelGGL((wgrad_h8i_kernel<UP, true, false>), grid, block, lds, s, p, ns, ncb, nib);
    else hipLaunchKernelGGL((wgrad_h8i_kernel<UP, true, true>), grid, block, lds, s, p, ns, ncb, nib);
  } else {
    hipLaunchKernelGGL((wgrad_h8_kernel<KS, UP>), dim3(ns * ncb * nib), dim3(512), (size_t)Cfg::LDS_BYTES, s, p, ns, ncb, nib);
  }
  const size_t total = (size_t)p.Cout * p.Cin_real * Cfg::T;
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)(ncb * nib * Cfg::T * 64)), dim3(256), 0, s, p.scratch, p.dw, p.Cout, p.Cin_real,
                     Cfg::T, ns, ncb, nib, total);
  if (p.colsum_part)
    hipLaunchKernelGGL(colsum_slices_kernel, dim3((p.Cout_s + 255) / 256, p.N), dim3(256), 0, s, p.colsum_part, p.colsum, p.Cout_s, ncb, pl.k);
  return hipGetLastError();
}

template <int KS, int STRIDE, bool UP>
static hipError_t launch_wgrad_h_t(const WgradParams& p, hipStream_t s) {
  using Cfg = WgHCfg<KS, STRIDE, UP>;
  const int Cin = p.C0 + p.C1;
  const int ncb = (p.Cout + 63) / 64, nib = (Cin + 63) / 64;
  const int ntiles = p.N * ((p.Wout + Cfg::TW - 1) / Cfg::TW) * ((p.Hout + Cfg::TH - 1) / Cfg::TH);
  const int ns = wgrad_slices(ntiles, ncb, nib);
  hipLaunchKernelGGL((wgrad_h_kernel<KS, STRIDE, UP>), dim3(ns * ncb * nib), dim3(256), (size_t)Cfg::LDS_BYTES, s, p, ns, ncb, nib);
  const size_t total = (size_t)p.Cout * p.Cin_real * Cfg::T;
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)(ncb * nib * Cfg::T * 64)), dim3(256), 0, s, p.scratch, p.dw, p.Cout, p.Cin_real,
                     Cfg::T, ns, ncb, nib, total);
  return hipGetLastError();
}

// the tiles (4x16, stride 2: 2x16) equal wgrad_kernel's, so the scratch sizing (wgrad_scratch_floats) is shared
hipError_t launch_wgrad_h(ConvKind kind, const WgradParams& p, hipStream_t s) {
  if ((p.C0 & 3) || (p.C1 & 3) || (p.Cout_s & 3)) return hipErrorInvalidValue;
  const H8Plan pl = wgrad_h8_plan(kind, p);
  switch (kind) {
    case CONV3_S1: return pl.h8 ? launch_wgrad_h8_t<3, false>(p, pl, s) : launch_wgrad_h_t<3, 1, false>(p, s);
    case CONV3_S2: return launch_wgrad_h_t<3, 2, false>(p, s);
    case CONV3_UP: return pl.h8 ? launch_wgrad_h8_t<3, true>(p, pl, s) : launch_wgrad_h_t<3, 1, true>(p, s);
    case CONV1: return launch_wgrad_h_t<1, 1, false>(p, s);   // bandwidth-bound: two small workgroups per CU keep more loads in flight
  }
  return hipErrorInvalidValue;
}

hipError_t train_kernels_init() {
  hipError_t e;
#define FDSR_WG_INIT(KS, ST, UP)                                                                                      \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<KS, ST, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)(WgCfg<KS, ST, UP>::LDS_FLOATS * sizeof(float)))) != hipSuccess)                      \
    return e;
  FDSR_WG_INIT(3, 1, false) FDSR_WG_INIT(3, 2, false) FDSR_WG_INIT(3, 1, true) FDSR_WG_INIT(1, 1, false)
#undef FDSR_WG_INIT
#define FDSR_WGH_INIT(KS, ST, UP)                                                                                      \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_h_kernel<KS, ST, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)WgHCfg<KS, ST, UP>::LDS_BYTES)) != hipSuccess)                                        \
    return e;
  FDSR_WGH_INIT(3, 1, false) FDSR_WGH_INIT(3, 2, false) FDSR_WGH_INIT(3, 1, true) FDSR_WGH_INIT(1, 1, false)
#undef FDSR_WGH_INIT
#define FDSR_WGH8_INIT(KS, UP)                                                                                         \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_h8_kernel<KS, UP>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               (int)WgH8Cfg<KS, UP>::LDS_BYTES)) != hipSuccess)                                           \
    return e;
  FDSR_WGH8_INIT(3, false) FDSR_WGH8_INIT(3, true)
#undef FDSR_WGH8_INIT
#define FDSR_WGH8I_INIT(UP, GN, DROP)                                                                                  \
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_h8i_kernel<UP, GN, DROP>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                               2 * WgH8Cfg<3, UP>::BUF + 4096)) != hipSuccess)                                           \
    return e;
  FDSR_WGH8I_INIT(false, false, false) FDSR_WGH8I_INIT(false, true, false) FDSR_WGH8I_INIT(false, true, true)
  FDSR_WGH8I_INIT(true, false, false) FDSR_WGH8I_INIT(true, true, false) FDSR_WGH8I_INIT(true, true, true)
#undef FDSR_WGH8I_INIT
  return hipSuccess;
}

// ---------------------------------------------------------------------------
// CLAM / SLAM backward
// ---------------------------------------------------------------------------
// scratch layout (floats), per launch_clam_slam_bwd:
//   pool [N][32][C][2] | avg [N][C] | mx [N][C] | ha [N][Cr] | hm [N][Cr] | gate [N][C] | map [N][2][HW] |
//   sg [N][HW] | dz [N][HW] | dm [N][2][HW] | dgp [N][32][C] | dgate [N][C] | davg [N][C] | dmx [N][C] |
//   pfc1 [N][Cr][C] | pfc2 [N][C][Cr]
#define FDSR_CSB_SLICES 32
struct CsbOff { size_t pool, avg, mx, ha, hm, gate, map, sg, dz, dm, dgp, dgate, davg, dmx, pfc1, pfc2, total; };
static CsbOff csb_offsets(int N, int HW, int C, int Cr) {
  CsbOff o{};
  size_t off = 0;
  auto take = [&](size_t n) { const size_t r = off; off += (n + 3) / 4 * 4; return r; };
  o.pool = take((size_t)N * FDSR_CSB_SLICES * C * 2);
  o.avg = take((size_t)N * C); o.mx = take((size_t)N * C);
  o.ha = take((size_t)N * Cr); o.hm = take((size_t)N * Cr);
  o.gate = take((size_t)N * C);
  o.map = take((size_t)N * 2 * HW);
  o.sg = take((size_t)N * HW); o.dz = take((size_t)N * HW);
  o.dm = take((size_t)N * 2 * HW);
  o.dgp = take((size_t)N * FDSR_CSB_SLICES * C);
  o.dgate = take((size_t)N * C); o.davg = take((size_t)N * C); o.dmx = take((size_t)N * C);
  o.pfc1 = take((size_t)N * Cr * C); o.pfc2 = take((size_t)N * C * Cr);
  o.total = off;
  return o;
}
size_t clam_slam_bwd_scratch_floats(int N, int HW, int C, int Cr) { return csb_offsets(N, HW, C, Cr).total; }

// (1) per-(image, channel) sum and max over a pixel slice: grid (SLICES, N)
__global__ void __launch_bounds__(256) csb_pool_kernel(const float* __restrict__ x, int HW, int C, float* __restrict__ pool) {
  const int sl = blockIdx.x, n = blockIdx.y;
  const int p0 = (int)((long)sl * HW / FDSR_CSB_SLICES), p1 = (int)((long)(sl + 1) * HW / FDSR_CSB_SLICES);
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f, m = -INFINITY;
    for (int pix = p0; pix < p1; ++pix) {
      const float v = x[((size_t)n * HW + pix) * C + c];
      s += v;
      m = fmaxf(m, v);
    }
    float* dst = pool + (((size_t)n * FDSR_CSB_SLICES + sl) * C + c) * 2;
    dst[0] = s;
    dst[1] = m;
  }
}

// (2) the gate MLP again, keeping its intermediates: grid (N)
__global__ void __launch_bounds__(256) csb_gate_fwd_kernel(const float* __restrict__ pool, int HW, int C, int Cr, const float* __restrict__ fc1,
                                                           const float* __restrict__ fc2, float* avg, float* mx, float* ha, float* hm,
                                                           float* gate) {
  const int n = blockIdx.x, tid = threadIdx.x;
  avg += (size_t)n * C; mx += (size_t)n * C; ha += (size_t)n * Cr; hm += (size_t)n * Cr; gate += (size_t)n * C;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f, m = -INFINITY;
    for (int sl = 0; sl < FDSR_CSB_SLICES; ++sl) {
      const float* src = pool + (((size_t)n * FDSR_CSB_SLICES + sl) * C + c) * 2;
      s += src[0];
      m = fmaxf(m, src[1]);
    }
    avg[c] = s / (float)HW;
    mx[c] = m;
  }
  __syncthreads();
  for (int j = tid; j < 2 * Cr; j += 256) {
    const int jj = j % Cr;
    const float* v = j < Cr ? avg : mx;
    float a = 0.f;
    for (int k = 0; k < C; ++k) a = fmaf(fc1[(size_t)jj * C + k], v[k], a);
    (j < Cr ? ha : hm)[jj] = fmaxf(a, 0.f);
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int j = 0; j < Cr; ++j) a = fmaf(fc2[(size_t)c * Cr + j], ha[j] + hm[j], a);
    gate[c] = sigmoid_f(a);
  }
}

// (3) map = [mean_c y, max_c y], y = x * gate: one wave per pixel, grid (ceil(HW/4), N)
__global__ void __launch_bounds__(256) csb_map_kernel(const float* __restrict__ x, const float* __restrict__ gate, int HW, int C,
                                                      float* __restrict__ map) {
  const int n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pix = blockIdx.x * 4 + wave;
  if (pix >= HW) return;
  float s = 0.f, m = -INFINITY;
  for (int c = lane; c < C; c += 64) {
    const float y = x[((size_t)n * HW + pix) * C + c] * gate[(size_t)n * C + c];
    s += y;
    m = fmaxf(m, y);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); m = fmaxf(m, __shfl_xor(m, o, 64)); }
  if (lane == 0) { map[(size_t)n * 2 * HW + pix] = s / (float)C; map[(size_t)n * 2 * HW + HW + pix] = m; }
}

// (4) sg = sigmoid(conv7x7(map)); dz = (sum_c dout*y) * sg*(1-sg): one wave per pixel
__global__ void __launch_bounds__(256) csb_dz_kernel(const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ gate,
                                                     const float* __restrict__ map, const float* __restrict__ w7, int H, int W, int C,
                                                     float* __restrict__ sg, float* __restrict__ dz) {
  const int HW = H * W, n = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pix = blockIdx.x * 4 + wave;
  if (pix >= HW) return;
  const int y0 = pix / W, x0 = pix % W;
  float z = 0.f;
  for (int k = lane; k < 98; k += 64) {
    const int ch = k / 49, ky = (k % 49) / 7, kx = k % 7;
    const int iy = y0 + ky - 3, ix = x0 + kx - 3;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) z += w7[k] * map[(size_t)n * 2 * HW + (size_t)ch * HW + iy * W + ix];
  }
  float ds = 0.f;
  for (int c = lane; c < C; c += 64) {
    const size_t o = ((size_t)n * HW + pix) * C + c;
    ds += dout[o] * (x[o] * gate[(size_t)n * C + c]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { z += __shfl_xor(z, o, 64); ds += __shfl_xor(ds, o, 64); }
  if (lane == 0) {
    const float s = sigmoid_f(z);
    sg[(size_t)n * HW + pix] = s;
    dz[(size_t)n * HW + pix] = ds * s * (1.0f - s);
  }
}

// (5) dm[ch][q] = sum_k dz[q - (k - 3)] * w7[ch][k]: thread per (n, ch, q)
__global__ void __launch_bounds__(256) csb_dm_kernel(const float* __restrict__ dz, const float* __restrict__ w7, int H, int W, size_t total,
                                                     float* __restrict__ dm) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // [N][2][HW]
  if (i >= total) return;
  const int HW = H * W;
  const int q = (int)(i % HW), ch = (int)((i / HW) % 2);
  const size_t n = i / (2 * (size_t)HW);
  const int qy = q / W, qx = q % W;
  float a = 0.f;
  for (int ky = 0; ky < 7; ++ky) {
    const int py = qy - (ky - 3);
    if (py < 0 || py >= H) continue;
    for (int kx = 0; kx < 7; ++kx) {
      const int px = qx - (kx - 3);
      if (px < 0 || px >= W) continue;
      a = fmaf(dz[n * HW + (size_t)py * W + px], w7[(ch * 7 + ky) * 7 + kx], a);
    }
  }
  dm[i] = a;
}

// (6) dw7[ch][ky][kx] = sum_{n,p} dz[n][p] * map[n][ch][p + k - 3]: one block per tap, ordered reduction
__global__ void __launch_bounds__(256) csb_dw7_kernel(const float* __restrict__ dz, const float* __restrict__ map, int N, int H, int W,
                                                      float* __restrict__ dw7) {
  __shared__ double sh[4];
  const int k = blockIdx.x, ch = k / 49, ky = (k % 49) / 7, kx = k % 7, HW = H * W;
  double acc = 0.0;
  for (size_t i = threadIdx.x; i < (size_t)N * HW; i += 256) {
    const size_t n = i / HW;
    const int p = (int)(i % HW), iy = p / W + ky - 3, ix = p % W + kx - 3;
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) acc += (double)dz[i] * (double)map[n * 2 * HW + (size_t)ch * HW + iy * W + ix];
  }
  const double s = block_sum_256(acc, sh);
  if (threadIdx.x == 0) dw7[k] = (float)s;
}

// (7) dy = dout*sg + dm0/C + dm1*[y == max_c y]; dx += dy*gate; per-slice sums of dy*x: grid (SLICES, N)
__global__ void __launch_bounds__(256) csb_dy_kernel(const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ gate,
                                                     const float* __restrict__ map, const float* __restrict__ sg, const float* __restrict__ dm,
                                                     int HW, int C, float* __restrict__ dx, float* __restrict__ dgp) {
  const int sl = blockIdx.x, n = blockIdx.y;
  const int p0 = (int)((long)sl * HW / FDSR_CSB_SLICES), p1 = (int)((long)(sl + 1) * HW / FDSR_CSB_SLICES);
  for (int c = threadIdx.x; c < C; c += 256) {
    const float g = gate[(size_t)n * C + c];
    float acc = 0.f;
    for (int pix = p0; pix < p1; ++pix) {
      const size_t o = ((size_t)n * HW + pix) * C + c;
      const float xv = x[o], y = xv * g;
      const float m1 = map[(size_t)n * 2 * HW + HW + pix];
      float dy = dout[o] * sg[(size_t)n * HW + pix] + dm[(size_t)n * 2 * HW + pix] / (float)C;
      if (y == m1) dy += dm[(size_t)n * 2 * HW + HW + pix];
      dx[o] += dy * g;
      acc += dy * xv;
    }
    dgp[((size_t)n * FDSR_CSB_SLICES + sl) * C + c] = acc;
  }
}

// (8) gate MLP backward per image: grid (N)
__global__ void __launch_bounds__(256) csb_gate_bwd_kernel(const float* __restrict__ dgp, const float* __restrict__ gate,
                                                           const float* __restrict__ avg, const float* __restrict__ mx,
                                                           const float* __restrict__ ha, const float* __restrict__ hm,
                                                           const float* __restrict__ fc1, const float* __restrict__ fc2, int C, int Cr,
                                                           float* __restrict__ dgate, float* __restrict__ davg, float* __restrict__ dmx,
                                                           float* __restrict__ pfc1, float* __restrict__ pfc2) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dq[C] | dha[Cr] | dhm[Cr]
  float* dq = sm;
  float* dha = sm + C;
  float* dhm = dha + Cr;
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) {
    float a = 0.f;
    for (int sl = 0; sl < FDSR_CSB_SLICES; ++sl) a += dgp[((size_t)n * FDSR_CSB_SLICES + sl) * C + c];
    dgate[(size_t)n * C + c] = a;
    const float g = gate[(size_t)n * C + c];
    dq[c] = a * g * (1.0f - g);
  }
  __syncthreads();
  for (int i = tid; i < C * Cr; i += 256) {                 // d fc2[c][j] (this image's term)
    const int c = i / Cr, j = i % Cr;
    pfc2[(size_t)n * C * Cr + i] = dq[c] * (ha[(size_t)n * Cr + j] + hm[(size_t)n * Cr + j]);
  }
  for (int j = tid; j < 2 * Cr; j += 256) {
    const int jj = j % Cr;
    const float hv = (j < Cr ? ha : hm)[(size_t)n * Cr + jj];
    float a = 0.f;
    for (int c = 0; c < C; ++c) a = fmaf(dq[c], fc2[(size_t)c * Cr + jj], a);
    (j < Cr ? dha : dhm)[jj] = hv > 0.f ? a : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < Cr * C; i += 256) {                 // d fc1[j][c]
    const int j = i / C, c = i % C;
    pfc1[(size_t)n * Cr * C + i] = dha[j] * avg[(size_t)n * C + c] + dhm[j] * mx[(size_t)n * C + c];
  }
  for (int c = tid; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int j = 0; j < Cr; ++j) { a = fmaf(dha[j], fc1[(size_t)j * C + c], a); b = fmaf(dhm[j], fc1[(size_t)j * C + c], b); }
    davg[(size_t)n * C + c] = a;
    dmx[(size_t)n * C + c] = b;
  }
}

// (9) sum the per-image terms in image order
__global__ void __launch_bounds__(256) sum_over_images_kernel(const float* __restrict__ per, int N, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int k = 0; k < N; ++k) a += per[(size_t)k * n + i];
  out[i] = a;
}

// (10) dx += davg/HW + dmx*[x == max_p x]
__global__ void __launch_bounds__(256) csb_pool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ davg,
                                                           const float* __restrict__ dmx, const float* __restrict__ mx, int HW, int C,
                                                           size_t total, float* __restrict__ dx) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // [N][HW][C]
  if (i >= total) return;
  const int c = (int)(i % C);
  const size_t n = i / ((size_t)HW * C);
  float v = davg[n * C + c] / (float)HW;
  if (x[i] == mx[n * C + c]) v += dmx[n * C + c];
  dx[i] += v;
}

hipError_t launch_clam_slam_bwd(const ClamSlamBwdParams& p, hipStream_t s) {
  const int HW = p.H * p.W, N = p.N, C = p.C, Cr = p.Cr;
  const CsbOff o = csb_offsets(N, HW, C, Cr);
  float* S = p.scratch;
  hipLaunchKernelGGL(csb_pool_kernel, dim3(FDSR_CSB_SLICES, N), dim3(256), 0, s, p.x, HW, C, S + o.pool);
  hipLaunchKernelGGL(csb_gate_fwd_kernel, dim3(N), dim3(256), 0, s, S + o.pool, HW, C, Cr, p.fc1, p.fc2, S + o.avg, S + o.mx, S + o.ha,
                     S + o.hm, S + o.gate);
  hipLaunchKernelGGL(csb_map_kernel, dim3((HW + 3) / 4, N), dim3(256), 0, s, p.x, S + o.gate, HW, C, S + o.map);
  hipLaunchKernelGGL(csb_dz_kernel, dim3((HW + 3) / 4, N), dim3(256), 0, s, p.x, p.dout, S + o.gate, S + o.map, p.w7, p.H, p.W, C,
                     S + o.sg, S + o.dz);
  const size_t tm = (size_t)N * 2 * HW;
  hipLaunchKernelGGL(csb_dm_kernel, dim3((unsigned)((tm + 255) / 256)), dim3(256), 0, s, S + o.dz, p.w7, p.H, p.W, tm, S + o.dm);
  hipLaunchKernelGGL(csb_dw7_kernel, dim3(98), dim3(256), 0, s, S + o.dz, S + o.map, N, p.H, p.W, p.dw7);
  hipLaunchKernelGGL(csb_dy_kernel, dim3(FDSR_CSB_SLICES, N), dim3(256), 0, s, p.x, p.dout, S + o.gate, S + o.map, S + o.sg, S + o.dm, HW,
                     C, p.dx, S + o.dgp);
  hipLaunchKernelGGL(csb_gate_bwd_kernel, dim3(N), dim3(256), (size_t)(C + 2 * Cr) * sizeof(float), s, S + o.dgp, S + o.gate, S + o.avg,
                     S + o.mx, S + o.ha, S + o.hm, p.fc1, p.fc2, C, Cr, S + o.dgate, S + o.davg, S + o.dmx, S + o.pfc1, S + o.pfc2);
  const size_t nf = (size_t)C * Cr;
  hipLaunchKernelGGL(sum_over_images_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, S + o.pfc1, N, nf, p.dfc1);
  hipLaunchKernelGGL(sum_over_images_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, s, S + o.pfc2, N, nf, p.dfc2);
  const size_t tx = (size_t)N * HW * C;
  hipLaunchKernelGGL(csb_pool_bwd_kernel, dim3((unsigned)((tx + 255) / 256)), dim3(256), 0, s, p.x, S + o.davg, S + o.dmx, S + o.mx, HW, C,
                     tx, p.dx);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// noise-level embedding backward
// ---------------------------------------------------------------------------
// Widths: enc E, hidden Hd, t Td (FastDiffSR / SR3: inner, 4 inner, inner; GDP: mc, 4 mc, 4 mc -- TembBwdParams::enc_dim ...).
// (1) per image: recompute enc / pre-activation / hid / t, then dt, dhid, dpre.  scratch per image:
//     enc[E] | hid[Hd] | dt[Td] | dpre[Hd] | t[Td]
struct TembDims { int E, Hd, Td; };
__host__ __device__ __forceinline__ TembDims temb_dims(const TembBwdParams& p) {
  return TembDims{p.enc_dim ? p.enc_dim : p.inner, p.hid_dim ? p.hid_dim : 4 * p.inner, p.t_dim ? p.t_dim : p.inner};
}
// (0) dt[k] = sum_o dtemb[n][o] * wn[o][k], the one large product of this backward (GDP at the reference's width: 30 000 x 512 per
//     image): grid (N, blocks of FDSR_TEMB_BWD_ROWS rows o).  Within a block the rows split over 256 / Td thread groups; the groups'
//     sums are added in order, and the image kernel adds the blocks in order: a fixed order whatever the grid.
#define FDSR_TEMB_BWD_ROWS 512
__host__ __device__ __forceinline__ int temb_bwd_blocks(int TE) { return (TE + FDSR_TEMB_BWD_ROWS - 1) / FDSR_TEMB_BWD_ROWS; }
__global__ void __launch_bounds__(256) temb_bwd_dt_kernel(const TembBwdParams p, float* __restrict__ dt_blocks) {
  extern __shared__ __attribute__((aligned(16))) float st[];   // [parts][Td]
  const TembDims d = temb_dims(p);
  const int Td = d.Td, tid = threadIdx.x, n = blockIdx.x, blk = blockIdx.y;
  const int r0 = blk * FDSR_TEMB_BWD_ROWS, r1 = min(p.TE, r0 + FDSR_TEMB_BWD_ROWS);
  const int parts = 256 / Td > 0 ? 256 / Td : 1, part = tid / Td, k = tid % Td;
  const float* dte = p.dtemb + (size_t)n * p.TE;
  if (Td > 256) {                                            // wider than the workgroup: one part, threads stride over k
    for (int kk = tid; kk < Td; kk += 256) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int o = r0;
      for (; o + 4 <= r1; o += 4) {
        a0 = fmaf(dte[o], p.wn[(size_t)o * Td + kk], a0);
        a1 = fmaf(dte[o + 1], p.wn[(size_t)(o + 1) * Td + kk], a1);
        a2 = fmaf(dte[o + 2], p.wn[(size_t)(o + 2) * Td + kk], a2);
        a3 = fmaf(dte[o + 3], p.wn[(size_t)(o + 3) * Td + kk], a3);
      }
      for (; o < r1; ++o) a0 = fmaf(dte[o], p.wn[(size_t)o * Td + kk], a0);
      st[kk] = (a0 + a1) + (a2 + a3);
    }
  } else if (part < parts) {
    const int o0 = r0 + (int)((long)part * (r1 - r0) / parts), o1 = r0 + (int)((long)(part + 1) * (r1 - r0) / parts);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int o = o0;
    for (; o + 4 <= o1; o += 4) {
      a0 = fmaf(dte[o], p.wn[(size_t)o * Td + k], a0);
      a1 = fmaf(dte[o + 1], p.wn[(size_t)(o + 1) * Td + k], a1);
      a2 = fmaf(dte[o + 2], p.wn[(size_t)(o + 2) * Td + k], a2);
      a3 = fmaf(dte[o + 3], p.wn[(size_t)(o + 3) * Td + k], a3);
    }
    for (; o < o1; ++o) a0 = fmaf(dte[o], p.wn[(size_t)o * Td + k], a0);
    st[part * Td + k] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  float* dst = dt_blocks + ((size_t)n * gridDim.y + blk) * Td;
  for (int kk = tid; kk < Td; kk += 256) {
    float a = 0.f;
    for (int q = 0; q < parts; ++q) a += st[q * Td + kk];
    dst[kk] = a;
  }
}

__global__ void __launch_bounds__(256) temb_bwd_image_kernel(const TembBwdParams p, const float* __restrict__ dt_blocks, int nblk) {
  extern __shared__ __attribute__((aligned(16))) float st[];   // enc[E] | pre[Hd] | hidv[Hd] | dt[Td]
  const TembDims d = temb_dims(p);
  const int E = d.E, hid = d.Hd, Td = d.Td, tid = threadIdx.x, n = blockIdx.x, half = E / 2;
  float* enc = st;
  float* pre = enc + E;
  float* hv = pre + hid;
  float* dt = hv + hid;
  const float nl = p.nl[n];
  for (int k = tid; k < half; k += 256) {
    const float e = nl * p.freq[k];
    enc[p.cos_first ? half + k : k] = sinf(e);
    enc[p.cos_first ? k : half + k] = cosf(e);
  }
  __syncthreads();
  for (int j = tid; j < hid; j += 256) {
    float a = p.b1[j];
    const float* w = p.w1 + (size_t)j * E;
    for (int k = 0; k < E; ++k) a = fmaf(w[k], enc[k], a);
    pre[j] = a;
    hv[j] = a / (1.0f + expf(-a));
  }
  for (int kk = tid; kk < Td; kk += 256) {                   // the row blocks of temb_bwd_dt_kernel, in order
    float a = 0.f;
    for (int b = 0; b < nblk; ++b) a += dt_blocks[((size_t)n * nblk + b) * Td + kk];
    dt[kk] = a;
  }
  __syncthreads();
  float* out = p.scratch + (size_t)n * (E + 2 * hid + 2 * Td);
  for (int k = tid; k < E; k += 256) out[k] = enc[k];
  for (int k = tid; k < Td; k += 256) {
    float t = p.b2[k];                                       // t[k] = b2[k] + sum_q w2[k][q] hid[q]: the input of the per-block Linear
    for (int q = 0; q < hid; ++q) t = fmaf(p.w2[(size_t)k * hid + q], hv[q], t);
    if (p.swish_block) {                                     // SR3 / GDP: the per-block Linear sees swish(t); what arrived in dt is d swish(t)
      const float sg = 1.0f / (1.0f + expf(-t));
      dt[k] *= sg * (1.0f + t * (1.0f - sg));
      t *= sg;
    }
    out[E + hid + k] = dt[k];
    out[E + 2 * hid + Td + k] = t;
  }
  __syncthreads();
  for (int j = tid; j < hid; j += 256) {
    float a = 0.f;                                           // dhid[j] = sum_k dt[k] * w2[k][j]
    for (int k = 0; k < Td; ++k) a = fmaf(dt[k], p.w2[(size_t)k * hid + j], a);
    const float u = pre[j], sg = 1.0f / (1.0f + expf(-u));
    out[E + j] = hv[j];
    out[E + hid + Td + j] = a * (sg * (1.0f + u * (1.0f - sg)));
  }
}

// (2) parameter gradients: sums over the images in order.  One thread per output element.
__global__ void __launch_bounds__(256) temb_bwd_param_kernel(const TembBwdParams p, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const TembDims d = temb_dims(p);
  const int E = d.E, hid = d.Hd, Td = d.Td, TE = p.TE, N = p.N;
  const size_t n_wn = (size_t)TE * Td, n_w2 = (size_t)Td * hid, n_w1 = (size_t)hid * E, stride = (size_t)E + 2 * hid + 2 * Td;
  const int o_hid = E, o_dt = E + hid, o_dpre = E + hid + Td, o_t = E + 2 * hid + Td;
  size_t j = i;
  auto S = [&](int n, int off) { return p.scratch + (size_t)n * stride + off; };
  if (j < n_wn) {                       // dwn[o][k] = sum_n dtemb[n][o] * t[n][k]
    const int o = (int)(j / Td), k = (int)(j % Td);
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += p.dtemb[(size_t)n * TE + o] * S(n, o_t)[k];
    p.dwn[j] = a;
    return;
  }
  j -= n_wn;
  if (j < (size_t)TE) {                 // dbn[o]
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += p.dtemb[(size_t)n * TE + j];
    p.dbn[j] = a;
    return;
  }
  j -= TE;
  if (j < n_w2) {                       // dw2[k][q] = sum_n dt[n][k] * hid[n][q]
    const int k = (int)(j / hid), q = (int)(j % hid);
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += S(n, o_dt)[k] * S(n, o_hid)[q];
    p.dw2[j] = a;
    return;
  }
  j -= n_w2;
  if (j < (size_t)Td) {                 // db2[k]
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += S(n, o_dt)[j];
    p.db2[j] = a;
    return;
  }
  j -= Td;
  if (j < n_w1) {                       // dw1[q][k] = sum_n dpre[n][q] * enc[n][k]
    const int q = (int)(j / E), k = (int)(j % E);
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += S(n, o_dpre)[q] * S(n, 0)[k];
    p.dw1[j] = a;
    return;
  }
  j -= n_w1;
  if (j < (size_t)hid) {                // db1[q]
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += S(n, o_dpre)[j];
    p.db1[j] = a;
  }
}

size_t temb_bwd_scratch_floats_per_image(int inner, int enc_dim, int hid_dim, int t_dim, int TE) {
  const size_t E = enc_dim ? enc_dim : inner, Hd = hid_dim ? hid_dim : 4 * inner, Td = t_dim ? t_dim : inner;
  return E + 2 * Hd + 2 * Td + (size_t)temb_bwd_blocks(TE) * Td;   // the image kernel's record | the row blocks' partial dt
}

hipError_t launch_temb_bwd(const TembBwdParams& p, hipStream_t s) {
  const TembDims d = temb_dims(p);
  const int parts = 256 / d.Td > 0 ? 256 / d.Td : 1, nblk = temb_bwd_blocks(p.TE);
  float* dt_blocks = p.scratch + (size_t)p.N * (d.E + 2 * d.Hd + 2 * d.Td);   // [N][nblk][Td], behind the per-image records
  hipLaunchKernelGGL(temb_bwd_dt_kernel, dim3(p.N, nblk), dim3(256), (size_t)parts * d.Td * sizeof(float), s, p, dt_blocks);
  hipLaunchKernelGGL(temb_bwd_image_kernel, dim3(p.N), dim3(256), (size_t)(d.E + 2 * d.Hd + d.Td) * sizeof(float), s, p, dt_blocks, nblk);
  const size_t total = (size_t)p.TE * d.Td + p.TE + (size_t)d.Td * d.Hd + d.Td + (size_t)d.Hd * d.E + d.Hd;
  hipLaunchKernelGGL(temb_bwd_param_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// SelfAttention backward (SR3 / TESR siblings: one head; GDP sibling: heads of C / heads channels)
// ---------------------------------------------------------------------------
// C[b][m][n] = alpha * sum_k A[b](m, k) B[b](k, n) with element strides on every operand, on v_mfma_f32_32x32x2_f32: one wave per
// 32 x 32 tile of C (the forward's attn_scores_kernel with the strides made arguments).  The products of the attention backward are a
// few MFLOP each; the strides let P^T, dS^T and the q / k / v channel slices of the NHWC qkv tensor be read in place.  The batch index
// is (image, head): operand X of batch b starts at (b / heads) * x_sb + (b % heads) * x_sh.
struct SGemm {
  const float* A; long a_sm, a_sk, a_sb, a_sh;
  const float* B; long b_sk, b_sn, b_sb, b_sh;
  float* C; long c_sm, c_sn, c_sb, c_sh;
  int M, Nn, K, heads;
  float alpha;
};
typedef float t_f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(64) sgemm_strided_kernel(const SGemm g) {
  const int lane = threadIdx.x, r31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32, bi = blockIdx.z / g.heads, hd = blockIdx.z % g.heads;
  const float* A = g.A + (size_t)bi * g.a_sb + (size_t)hd * g.a_sh + (size_t)min(m0 + r31, g.M - 1) * g.a_sm;
  const float* B = g.B + (size_t)bi * g.b_sb + (size_t)hd * g.b_sh + (size_t)min(n0 + r31, g.Nn - 1) * g.b_sn;
  t_f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int k = 0; k < g.K; k += 8) {
    float av[4], bv[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {   // (every load of the trip in flight before the first product)
      const int kk = k + 4 * h + s2, kc = min(kk, g.K - 1);
      const float a = A[(size_t)kc * g.a_sk], bb = B[(size_t)kc * g.b_sk];
      av[s2] = kk < g.K ? a : 0.f;
      bv[s2] = kk < g.K ? bb : 0.f;
    }
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], bv[s2], acc, 0, 0, 0);
  }
  const int col = n0 + r31;
  float* C = g.C + (size_t)bi * g.c_sb + (size_t)hd * g.c_sh;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (row < g.M && col < g.Nn) C[(size_t)row * g.c_sm + (size_t)col * g.c_sn] = acc[i] * g.alpha;
  }
}

static hipError_t launch_sgemm(const SGemm& g, int images, hipStream_t s) {
  hipLaunchKernelGGL(sgemm_strided_kernel, dim3((g.Nn + 31) / 32, (g.M + 31) / 32, images * g.heads), dim3(64), 0, s, g);
  return hipGetLastError();
}

// dS[r][j] = P[r][j] (dP[r][j] - sum_j' dP[r][j'] P[r][j']), in place over dP; one wave per row, the row sum folded in a fixed order
__global__ void __launch_bounds__(256) attn_dsoftmax_rows_kernel(const float* __restrict__ P, float* __restrict__ D, int HW, int HWp, size_t rows) {
  const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* pr = P + row * HWp;
  float* dr = D + row * HWp;
  float a = 0.f;
  for (int i = lane; i < HW; i += 64) a = fmaf(dr[i], pr[i], a);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
  for (int i = lane; i < HW; i += 64) dr[i] = pr[i] * (dr[i] - a);
}

size_t attn_bwd_scratch_floats(int N, int HW, int heads) { return 2 * attn_scratch_floats(N, HW, heads < 1 ? 1 : heads); }

hipError_t launch_attn_bwd(const AttnBwdParams& p, hipStream_t s) {
  const int N = p.N, HW = p.HW, C = p.C, HWp = (HW + 15) / 16 * 16, heads = p.heads < 1 ? 1 : p.heads;
  if (C % heads || (C / heads) % 32) return hipErrorInvalidValue;
  const int ch = C / heads;
  float* P = p.scratch;
  float* D = p.scratch + attn_scratch_floats(N, HW, heads);
  hipError_t e = launch_attn_probs(p.qkv, P, N, HW, C, heads, s);
  if (e != hipSuccess) return e;
  const float inv = 1.0f / sqrtf((float)ch);
  // channel offsets of q / k / v of head 0 and the step from head to head (fdsr_kernels.hip: attn_offsets)
  const long ko = heads > 1 ? ch : C, vo = heads > 1 ? 2L * ch : 2L * C, hq = 3L * ch;
  const long sq = (long)HW * 3 * C, so = (long)HW * C, sp = (long)HW * HWp, spn = sp * heads;
  // dP[i][j] = sum_c dO[i][c] V[j][c]
  SGemm g{p.dO, C, 1, so, ch, p.qkv + vo, 1, 3L * C, sq, hq, D, HWp, 1, spn, sp, HW, HW, ch, heads, 1.0f};
  if ((e = launch_sgemm(g, N, s)) != hipSuccess) return e;
  const size_t rows = (size_t)N * heads * HW;
  hipLaunchKernelGGL(attn_dsoftmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, P, D, HW, HWp, rows);
  // dQ[i][c] = inv sum_j dS[i][j] K[j][c]
  g = SGemm{D, HWp, 1, spn, sp, p.qkv + ko, 3L * C, 1, sq, hq, p.dqkv, 3L * C, 1, sq, hq, HW, ch, HW, heads, inv};
  if ((e = launch_sgemm(g, N, s)) != hipSuccess) return e;
  // dK[j][c] = inv sum_i dS[i][j] Q[i][c]
  g = SGemm{D, 1, HWp, spn, sp, p.qkv, 3L * C, 1, sq, hq, p.dqkv + ko, 3L * C, 1, sq, hq, HW, ch, HW, heads, inv};
  if ((e = launch_sgemm(g, N, s)) != hipSuccess) return e;
  // dV[j][c] = sum_i P[i][j] dO[i][c]
  g = SGemm{P, 1, HWp, spn, sp, p.dO, C, 1, so, ch, p.dqkv + vo, 3L * C, 1, sq, hq, HW, ch, HW, heads, 1.0f};
  if ((e = launch_sgemm(g, N, s)) != hipSuccess) return e;
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Adam and weight re-packing
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                                                   float bc1, float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  // torch.optim.Adam (single-tensor path): exp_avg.lerp_(grad, 1-b1); exp_avg_sq.mul_(b2).addcmul_(grad, grad, 1-b2)
  const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
  const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  w[i] = w[i] - (lr / bc1) * (mi / denom);
}

hipError_t launch_adam(float* w, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps, int step,
                       hipStream_t s) {
  const double bc1 = 1.0 - std::pow((double)b1, step), bc2 = 1.0 - std::pow((double)b2, step);
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, g, m, v, n, lr, b1, b2, eps, (float)bc1,
                     (float)std::sqrt(bc2));
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) pack_conv_f32_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int T,
                                                            int cout_pad, int cin_pad, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [T][cout_pad][cin_pad]
  if (i >= total) return;
  const int ci = (int)(i % cin_pad);
  size_t r = i / cin_pad;
  const int co = (int)(r % cout_pad), t = (int)(r / cout_pad);
  packed[i] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * T + t] : 0.f;
}

hipError_t launch_pack_conv_f32(const float* w, float* packed, int Cout, int Cin, int ks, int cout_pad, int cin_pad, hipStream_t s) {
  const int T = ks * ks;
  const size_t total = (size_t)T * cout_pad * cin_pad;
  hipLaunchKernelGGL(pack_conv_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, packed, Cout, Cin, T, cout_pad, cin_pad,
                     total);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) pack_conv_f32_t_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin, int T,
                                                              int c_off, int Csub, int rows_pad, int cols_pad, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [T][rows_pad (ci)][cols_pad (co)]
  if (i >= total) return;
  const int co = (int)(i % cols_pad);
  size_t r = i / cols_pad;
  const int ci = (int)(r % rows_pad), t = (int)(r / rows_pad);
  packed[i] = (co < Cout && ci < Csub) ? w[((size_t)co * Cin + c_off + ci) * T + (T - 1 - t)] : 0.f;   // tap flip: 8 - t
}

hipError_t launch_pack_conv_f32_t(const float* w, float* packed_t, int Cout, int Cin, int ks, int c_off, int Csub, int rows_pad,
                                  int cols_pad, hipStream_t s) {
  const int T = ks * ks;
  const size_t total = (size_t)T * rows_pad * cols_pad;
  hipLaunchKernelGGL(pack_conv_f32_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, packed_t, Cout, Cin, T, c_off, Csub,
                     rows_pad, cols_pad, total);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// f16x3 weight forms, packed on the device
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hamax_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ amax_bits) {
  __shared__ float sm[4];
  float m = 0.f;
  const size_t step = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 16 * step) {   // 16 loads in flight per trip (a max: re-reading the last element is harmless)
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = w[min(i0 + j * step, n - 1)];
#pragma unroll
    for (int j = 0; j < 16; ++j) m = fmaxf(m, fabsf(v[j]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(amax_bits, __builtin_bit_cast(unsigned, fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]))));
}

hipError_t launch_hamax(const float* w, size_t n, unsigned* amax_bits, hipStream_t s) {
  const unsigned blocks = (unsigned)std::min<size_t>((n + 256 * 16 - 1) / (256 * 16), 512);
  hipLaunchKernelGGL(hamax_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, s, w, n, amax_bits);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) hscale_all_kernel(const unsigned* __restrict__ amax_bits, float* __restrict__ scale2, int nslots) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nslots) return;
  const float amax = __builtin_bit_cast(float, amax_bits[i]);
  int e = 12;
  if (amax > 0.f) {
    const int f = (int)floor(log2(32768.0 / (double)amax));
    e = f < 12 ? f : 12;
  }
  scale2[2 * i] = ldexpf(1.0f, e);
  scale2[2 * i + 1] = ldexpf(1.0f, -e);
}

hipError_t launch_hscale_all(const unsigned* amax_bits, float* scale2, int nslots, hipStream_t s) {
  hipLaunchKernelGGL(hscale_all_kernel, dim3((nslots + 255) / 256), dim3(256), 0, s, amax_bits, scale2, nslots);
  return hipGetLastError();
}

// thread = (fragment, lane): 8 hi and 8 lo halves
__global__ void __launch_bounds__(256) pack_conv_h_kernel(const float* __restrict__ w, uint4* __restrict__ frags,
                                                          const float* __restrict__ scale2, int Cout, int Cin, int T, int WN, int nk,
                                                          int transposed, int c_off, int rows, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [cot][kc][wn][t][lane]
  if (i >= total) return;
  const int l = (int)(i & 63);
  size_t f = i >> 6;
  const int t = (int)(f % T);  f /= T;
  const int wn = (int)(f % WN);  f /= WN;
  const int kc = (int)(f % nk);
  const int cot = (int)(f / nk);
  const int BN = 32 * WN;
  const int co = cot * BN + wn * 32 + (l & 31);
  const float scale = scale2[0];
  unsigned short hi[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kc * 16 + 8 * (l >> 5) + j;
    float v = 0.f;
    if (!transposed) {
      if (co < Cout && k < Cin) v = w[((size_t)co * Cin + k) * T + t];
    } else {   // rows = the slice of input channels that become output channels; k runs over the forward Cout
      if (co < rows && k < Cout) v = w[((size_t)k * Cin + c_off + co) * T + (T - 1 - t)];
    }
    const float vs = v * scale;
    const _Float16 h = (_Float16)vs;
    const _Float16 lw = (_Float16)(vs - (float)h);
    hi[j] = __builtin_bit_cast(unsigned short, h);
    lo[j] = __builtin_bit_cast(unsigned short, lw);
  }
  const size_t fidx = (i >> 6);                   // fragment index (cot, kc, wn, t)
  uint4 a, b;
  a.x = hi[0] | ((unsigned)hi[1] << 16); a.y = hi[2] | ((unsigned)hi[3] << 16); a.z = hi[4] | ((unsigned)hi[5] << 16); a.w = hi[6] | ((unsigned)hi[7] << 16);
  b.x = lo[0] | ((unsigned)lo[1] << 16); b.y = lo[2] | ((unsigned)lo[3] << 16); b.z = lo[4] | ((unsigned)lo[5] << 16); b.w = lo[6] | ((unsigned)lo[7] << 16);
  frags[(fidx * 2) * 64 + l] = a;                 // plane 0 = hi
  frags[(fidx * 2 + 1) * 64 + l] = b;             // plane 1 = lo
}

// Sub-pixel form of Upsample(nearest x2) + Conv3x3 (pack_weights_h in fdsr_engine.cpp): per output parity (py, px) and source
// offset (a, b) the 3x3 taps that land there are pre-summed; fragments [cot][kc][wn][py][px*4+a*2+b][plane][lane] x 16 B.  A sum
// of up to four taps is at most 4 max|w|, so the scale is the 3x3 form's divided by 4 (inv_out = its inverse, for the epilogue).
__global__ void __launch_bounds__(256) pack_conv_up2_h_kernel(const float* __restrict__ w, uint4* __restrict__ frags,
                                                              const float* __restrict__ scale2, float* __restrict__ inv_out, int Cout,
                                                              int Cin, int WN, int nk, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [cot][kc][wn][py][slot][lane]
  if (i == 0) *inv_out = scale2[1] * 4.0f;
  if (i >= total) return;
  const int l = (int)(i & 63);
  size_t f = i >> 6;
  const int slot = (int)(f & 7);  f >>= 3;
  const int py = (int)(f & 1);  f >>= 1;
  const int wn = (int)(f % WN);  f /= WN;
  const int kc = (int)(f % nk);
  const int cot = (int)(f / nk);
  const int px = slot >> 2, a = (slot >> 1) & 1, b = slot & 1;
  // taps of parity par that land on source offset o: par 0: {0} | {1,2}; par 1: {0,1} | {2}
  const int y0 = py == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2), y1 = py == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2);
  const int x0 = px == 0 ? (b == 0 ? 0 : 1) : (b == 0 ? 0 : 2), x1 = px == 0 ? (b == 0 ? 0 : 2) : (b == 0 ? 1 : 2);
  const int co = cot * 32 * WN + wn * 32 + (l & 31);
  const float scale = scale2[0] * 0.25f;
  unsigned short hi[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = kc * 16 + 8 * (l >> 5) + j;
    float v = 0.f;
    if (co < Cout && k < Cin) {
      const float* wp = w + ((size_t)co * Cin + k) * 9;
      for (int ky = y0; ky <= y1; ++ky)
        for (int kx = x0; kx <= x1; ++kx) v += wp[ky * 3 + kx];
    }
    const float vs = v * scale;
    const _Float16 h = (_Float16)vs;
    const _Float16 lw = (_Float16)(vs - (float)h);
    hi[j] = __builtin_bit_cast(unsigned short, h);
    lo[j] = __builtin_bit_cast(unsigned short, lw);
  }
  const size_t fidx = (i >> 6);
  uint4 ua, ub;
  ua.x = hi[0] | ((unsigned)hi[1] << 16); ua.y = hi[2] | ((unsigned)hi[3] << 16); ua.z = hi[4] | ((unsigned)hi[5] << 16); ua.w = hi[6] | ((unsigned)hi[7] << 16);
  ub.x = lo[0] | ((unsigned)lo[1] << 16); ub.y = lo[2] | ((unsigned)lo[3] << 16); ub.z = lo[4] | ((unsigned)lo[5] << 16); ub.w = lo[6] | ((unsigned)lo[7] << 16);
  frags[(fidx * 2) * 64 + l] = ua;
  frags[(fidx * 2 + 1) * 64 + l] = ub;
}

hipError_t launch_pack_conv_up2_h(const float* w, void* frags, const float* scale2, float* inv_out, int Cout, int Cin, int WN, int cout_pad,
                                  int cin_pad, hipStream_t s) {
  const int ncot = cout_pad / (32 * WN), nk = cin_pad / 16;
  const size_t total = (size_t)ncot * nk * WN * 16 * 64;
  hipLaunchKernelGGL(pack_conv_up2_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<uint4*>(frags), scale2,
                     inv_out, Cout, Cin, WN, nk, total);
  return hipGetLastError();
}

hipError_t launch_pack_conv_h(const float* w, void* frags, const float* scale2, int Cout, int Cin, int ks, int WN, int cout_pad,
                              int cin_pad, int transposed, int c_off, int rows, hipStream_t s) {
  const int T = ks * ks, BN = 32 * WN, ncot = cout_pad / BN, nk = cin_pad / 16;
  const size_t total = (size_t)ncot * nk * WN * T * 64;
  hipLaunchKernelGGL(pack_conv_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, reinterpret_cast<uint4*>(frags), scale2,
                     Cout, Cin, T, WN, nk, transposed, c_off, rows, total);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) gn_silu_drop_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                                           const unsigned char* __restrict__ mask, float drop_scale, float* __restrict__ out,
                                                           int HW, int cq, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [N][HW][cq]
  if (i >= total) return;
  const int c4 = (int)(i % cq);
  const size_t n = i / ((size_t)HW * cq);
  f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
  const f32x4 a = *reinterpret_cast<const f32x4*>(sc + (n * cq + c4) * 4), b = *reinterpret_cast<const f32x4*>(sh + (n * cq + c4) * 4);
  const unsigned m = *reinterpret_cast<const unsigned*>(mask + i * 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float u = fmaf(v[e], a[e], b[e]);
    v[e] = ((m >> (8 * e)) & 0xffu) ? u * sigmoid_f(u) * drop_scale : 0.f;
  }
  *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

hipError_t launch_gn_silu_drop(const float* x, const float* gn_scale, const float* gn_shift, const unsigned char* mask, float drop_scale,
                               float* out, int N, int HW, int C, hipStream_t s) {
  if (C & 3) return hipErrorInvalidValue;
  const size_t total = (size_t)N * HW * (C >> 2);
  hipLaunchKernelGGL(gn_silu_drop_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, gn_scale, gn_shift, mask, drop_scale,
                     out, HW, C >> 2, total);
  return hipGetLastError();
}

}  // namespace fdsr
