"""TEST INFRASTRUCTURE -- CPU restatement of the image metrics the reference's evaluation loop computes.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product (fastdiffsr_amd/)
never does (tests/test_abi_symbols.py checks).  It is the CHECKER of the HIP metric kernels (csrc/fdsr_val.hip).

What the reference calls, and where:
  * sr_mfe.py:165-173 (and :313-333 in the val branch): `compare_mse`, `compare_psnr`, `compare_ssim(.., multichannel=True)` imported
    from `skimage.measure` (sr_mfe.py:15-17) -- scikit-image 0.16's functions with their defaults;
  * core/metrics.py:94-101 `calculate_psnr`, :104-125 `ssim` (cv2.getGaussianKernel(11, 1.5), cv2.filter2D, crop [5:-5, 5:-5]),
    :128-145 `calculate_ssim`, :147-152 `calculate_ergas` (skimage `compare_mse`).
scikit-image and OpenCV are third-party dependencies that are NOT vendored in /root/reference and NOT installed in this image
(the reference pins no versions for them; `compare_*` left skimage.measure in 0.18, so <= 0.17).  Their algorithms are restated
here from the published definitions with the primitives they themselves are built on:
  * skimage.measure.compare_ssim (0.16, `structural_similarity`): `scipy.ndimage.uniform_filter(size=win_size)` (scipy's default
    border mode 'reflect'), win_size 7, K1 = 0.01, K2 = 0.03, sample covariance (cov_norm = NP / (NP - 1)), data range 255 for
    uint8, the mean over the map cropped by (win_size - 1) // 2, multichannel = mean of the per-channel values;
  * cv2.getGaussianKernel(ksize, sigma): exp(-(i - (ksize - 1) / 2)^2 / (2 sigma^2)) normalised to sum 1;
  * cv2.filter2D(img, -1, k): correlation with the kernel anchored at its centre, BORDER_REFLECT_101 = scipy's mode 'mirror'.
PINNING: the core/metrics.py functions are pinned by tests/golden/metrics_ssim.npz (the reference's OWN functions executed in the
build container by oracle/make_goldens.py `metrics`, around restated cv2 primitives) and by brute-force window loops in
tests/test_oracle_golden.py; `compare_ssim` is pinned only against its brute-force definition: **unpinned against skimage itself**
(DESIGN.md section 2)."""
import math

import numpy as np
from scipy import ndimage


# ---- skimage.measure (0.16) as the val loop calls it: sr_mfe.py:165-173, :313-333 ----
def compare_mse(im1, im2):
    """skimage.measure.compare_mse: mean over ALL elements of the squared difference, in float64."""
    a, b = np.asarray(im1, dtype=np.float64), np.asarray(im2, dtype=np.float64)
    return float(np.mean(np.square(a - b), dtype=np.float64))


def compare_psnr(im_true, im_test, data_range=255.0):
    """skimage.measure.compare_psnr: 10 log10(data_range^2 / mse); uint8 images: data_range = 255."""
    err = compare_mse(im_true, im_test)
    return float('inf') if err == 0 else 10.0 * math.log10((data_range ** 2) / err)


def compare_ssim(X, Y, multichannel=False, win_size=7, data_range=255.0):
    """skimage.measure.compare_ssim with its defaults (gaussian_weights=False, use_sample_covariance=True, K1 = 0.01, K2 = 0.03)."""
    X, Y = np.asarray(X), np.asarray(Y)
    if X.shape != Y.shape:
        raise ValueError('Input images must have the same dimensions.')
    if multichannel:
        return float(np.mean([compare_ssim(X[..., c], Y[..., c], False, win_size, data_range) for c in range(X.shape[-1])]))
    if min(X.shape) < win_size:
        raise ValueError('win_size exceeds image extent.')
    X, Y = X.astype(np.float64), Y.astype(np.float64)
    NP = win_size ** X.ndim
    cov_norm = NP / (NP - 1.0)
    filt = lambda a: ndimage.uniform_filter(a, size=win_size)
    ux, uy = filt(X), filt(Y)
    uxx, uyy, uxy = filt(X * X), filt(Y * Y), filt(X * Y)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return float(S[tuple(slice(pad, -pad) for _ in range(S.ndim))].mean(dtype=np.float64))


# ---- core/metrics.py ----
def calculate_psnr(img1, img2):
    """core/metrics.py:94-101."""
    mse = np.mean((np.asarray(img1, dtype=np.float64) - np.asarray(img2, dtype=np.float64)) ** 2)
    return float('inf') if mse == 0 else 20.0 * math.log10(255.0 / math.sqrt(mse))


def gaussian_kernel(ksize=11, sigma=1.5):
    """cv2.getGaussianKernel(ksize, sigma) as a 1-D float64 vector."""
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(x * x) / (2.0 * sigma * sigma))
    return k / k.sum()


def filter2d(img, window):
    """cv2.filter2D(img, -1, window): correlation, anchor at the centre, BORDER_REFLECT_101; every channel on its own."""
    img = np.asarray(img, dtype=np.float64)
    if img.ndim == 3:
        return np.stack([ndimage.correlate(img[..., c], window, mode='mirror') for c in range(img.shape[2])], axis=-1)
    return ndimage.correlate(img, window, mode='mirror')


def ssim(img1, img2):
    """core/metrics.py:104-125 (the crop [5:-5, 5:-5] keeps the border-independent part; a 3-D array is filtered per channel and the
    mean runs over all of them)."""
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = np.asarray(img1, dtype=np.float64), np.asarray(img2, dtype=np.float64)
    k = gaussian_kernel(11, 1.5)
    window = np.outer(k, k)
    crop = lambda m: m[5:-5, 5:-5]
    mu1, mu2 = crop(filter2d(a, window)), crop(filter2d(b, window))
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    sigma1_sq = crop(filter2d(a ** 2, window)) - mu1_sq
    sigma2_sq = crop(filter2d(b ** 2, window)) - mu2_sq
    sigma12 = crop(filter2d(a * b, window)) - mu1_mu2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return float(ssim_map.mean())


def calculate_ssim(img1, img2):
    """core/metrics.py:128-145: the 3-channel branch scores the WHOLE array three times (`ssim(img1, img2)` inside the loop)."""
    if not img1.shape == img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return float(np.array([ssim(img1, img2) for _ in range(3)]).mean())
        if img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError('Wrong input image dimensions.')


def calculate_ergas(img1, img2, scale=4):
    """core/metrics.py:147-152."""
    channel = img1.shape[2]
    mse = compare_mse(img1, img2)
    mean2 = np.mean(img1, dtype=np.float64) ** 2
    return float(100.0 * np.sqrt(mse / mean2 / channel) / scale)
