"""Image metrics of the reference's val loop (FastDiffSR/core/metrics.py), host side, plus the
GPU `tensor2img` (clamp -> uint8 on the device, so only 1 byte/channel crosses PCIe instead of the
fp32 copy `DDPM.get_current_visuals` makes, model/model.py:97-111).

tensor2img and PSNR are pinned by goldens generated from the reference's own functions
(tests/golden/metrics.npz).  SSIM / ERGAS follow the reference formulas (metrics.py:103-152) but the
reference needs cv2 / skimage to run them, which this image lacks: those two are unpinned.
LPIPS (AlexNet weights) is out of scope.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib


def tensor2img(tensor, out_type=np.uint8, min_max=(-1, 1)):
    """metrics.py:16-42 for 3D (C,H,W) / 2D (H,W) tensors (after squeeze) -> HWC / HW numpy array.
    CUDA tensors are converted on the device (uint8 output only)."""
    t = tensor.squeeze()
    if t.dim() not in (2, 3):
        raise TypeError('Only support 3D and 2D tensor. But received with dimension: {:d}'.format(t.dim()))
    if t.is_cuda and out_type == np.uint8:
        t3 = (t if t.dim() == 3 else t[None]).float().contiguous()
        c, h, w = t3.shape
        dst = torch.empty(h, w, c, dtype=torch.uint8, device=t3.device)
        lib = _lib.load()
        st = torch.cuda.current_stream(t3.device).cuda_stream
        _lib.check(None, lib.fdsr_tensor2img_u8(None, C.c_void_p(t3.data_ptr()), C.c_void_p(dst.data_ptr()), 1, c, h, w,
                                                float(min_max[0]), float(min_max[1]), C.c_void_p(st)))
        out = dst.cpu().numpy()
        return out if t.dim() == 3 else out[:, :, 0]
    t = t.float().cpu().clamp_(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    img = t.numpy()
    if t.dim() == 3:
        img = np.transpose(img, (1, 2, 0))
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def calculate_mse(img1, img2):                       # skimage.measure.compare_mse
    return float(np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2))


def calculate_psnr(img1, img2):                      # metrics.py:94-101
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _gauss_window(size=11, sigma=1.5):               # cv2.getGaussianKernel(11, 1.5) outer product
    x = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    k = np.exp(-(x ** 2) / (2 * sigma ** 2))
    k /= k.sum()
    return np.outer(k, k)


def _filter_valid(img, win):
    """cv2.filter2D(img, -1, win)[5:-5, 5:-5]: the border-independent ('valid') part, per channel."""
    from numpy.lib.stride_tricks import sliding_window_view
    if img.ndim == 3:
        return np.stack([_filter_valid(img[..., c], win) for c in range(img.shape[2])], axis=-1)
    v = sliding_window_view(img, win.shape)
    return np.einsum('ijkl,kl->ij', v, win)


def ssim(img1, img2):                                # metrics.py:103-123
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    win = _gauss_window()
    mu1, mu2 = _filter_valid(a, win), _filter_valid(b, win)
    mu1_sq, mu2_sq, mu12 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _filter_valid(a ** 2, win) - mu1_sq
    s2 = _filter_valid(b ** 2, win) - mu2_sq
    s12 = _filter_valid(a * b, win) - mu12
    m = ((2 * mu12 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return float(m.mean())


def calculate_ssim(img1, img2):                      # metrics.py:126-145 (3-channel: ssim of the whole array, as there)
    if img1.shape != img2.shape:
        raise ValueError('Input images must have the same dimensions.')
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return float(np.array([ssim(img1, img2) for _ in range(3)]).mean())
        if img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError('Wrong input image dimensions.')


def calculate_ergas(img1, img2, scale=4):            # metrics.py:147-152
    channel = img1.shape[2]
    mse = calculate_mse(img1, img2)
    mean2 = np.mean(img1, dtype=np.float64) ** 2
    return float(100.0 * np.sqrt(mse / mean2 / channel) / scale)


# --- what sr_mfe.py's val loop actually calls for MSE / PSNR / SSIM (sr_mfe.py:313-333): the
# skimage.measure functions of skimage 0.16 with their defaults.  skimage is not in this image, so these are
# restated from its published algorithm (compare_ssim: uniform 7x7 window via scipy.ndimage.uniform_filter,
# sample covariance, K1 = 0.01, K2 = 0.03, data range of the dtype, borders cropped, mean over channels) and
# are unpinned.
def compare_mse(im1, im2):
    return float(np.mean(np.square(im1.astype(np.float64) - im2.astype(np.float64)), dtype=np.float64))


def compare_psnr(im_true, im_test):
    err = compare_mse(im_true, im_test)
    return float('inf') if err == 0 else 10 * math.log10((255.0 ** 2) / err)


def compare_ssim(X, Y, multichannel=True, win_size=7):
    from scipy.ndimage import uniform_filter
    if multichannel:
        return float(np.mean([compare_ssim(X[..., c], Y[..., c], multichannel=False, win_size=win_size)
                              for c in range(X.shape[-1])]))
    K1, K2, R = 0.01, 0.03, 255.0
    X, Y = X.astype(np.float64), Y.astype(np.float64)
    NP = win_size ** X.ndim
    cov_norm = NP / (NP - 1)
    ux, uy = uniform_filter(X, size=win_size), uniform_filter(Y, size=win_size)
    uxx, uyy, uxy = uniform_filter(X * X, size=win_size), uniform_filter(Y * Y, size=win_size), uniform_filter(X * Y, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * R) ** 2, (K2 * R) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return float(S[pad:-pad, pad:-pad].mean())
