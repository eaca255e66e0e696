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


# --- device-side forms for the evaluation loop (val.py): the batch stays on the GPU as uint8, the per-pixel work of the
# metrics runs in libfdsr_hip.so (csrc/fdsr_val.hip) and only B x 8 doubles come back ---------------------------------
def tensor2img_batch(t4, min_max=(-1, 1)):
    """tensor2img of every image of a [B,C,H,W] CUDA batch in ONE launch -> [B,H,W,C] uint8 CUDA tensor (stays on the device)."""
    if not t4.is_cuda or t4.dim() != 4:
        raise ValueError('tensor2img_batch takes a [B,C,H,W] CUDA tensor')
    t4 = t4.float().contiguous()
    b, c, h, w = t4.shape
    dst = torch.empty(b, h, w, c, dtype=torch.uint8, device=t4.device)
    st = torch.cuda.current_stream(t4.device).cuda_stream
    _lib.check(None, _lib.load().fdsr_tensor2img_u8(None, C.c_void_p(t4.data_ptr()), C.c_void_p(dst.data_ptr()), b, c, h, w,
                                                    float(min_max[0]), float(min_max[1]), C.c_void_p(st)))
    return dst


def u8_to_tensor(u8, min_max=(-1, 1)):
    """The dataset transform (data/util.py:66-75: ToTensor() then * (max - min) + min) of a decoded [B,H,W,C] uint8 CUDA batch
    -> [B,C,H,W] fp32, bit-identical to dataset.to_tensor on the host."""
    if not u8.is_cuda or u8.dtype != torch.uint8 or u8.dim() != 4:
        raise ValueError('u8_to_tensor takes a [B,H,W,C] uint8 CUDA tensor')
    u8 = u8.contiguous()
    b, h, w, c = u8.shape
    dst = torch.empty(b, c, h, w, dtype=torch.float32, device=u8.device)
    st = torch.cuda.current_stream(u8.device).cuda_stream
    _lib.check(None, _lib.load().fdsr_u8_to_tensor(None, C.c_void_p(u8.data_ptr()), C.c_void_p(dst.data_ptr()), b, c, h, w,
                                                   float(min_max[0]), float(min_max[1]), C.c_void_p(st)))
    return dst


_metric_ws = {}


def image_metric_sums(test_u8, truth_u8, gauss=False, out=None):
    """include/fdsr.h: fdsr_image_metrics_u8 -> [B,8] fp64 CUDA tensor of per-image sums (asynchronous on the current stream)."""
    if (not test_u8.is_cuda or test_u8.dtype != torch.uint8 or test_u8.dim() != 4 or truth_u8.shape != test_u8.shape or
            truth_u8.dtype != torch.uint8 or truth_u8.device != test_u8.device):
        raise ValueError('image_metric_sums takes two [B,H,W,C] uint8 CUDA tensors of one shape')
    test_u8, truth_u8 = test_u8.contiguous(), truth_u8.contiguous()
    b, h, w, c = test_u8.shape
    lib = _lib.load()
    need = C.c_size_t()
    _lib.check(None, lib.fdsr_image_metrics_workspace_bytes(b, h, w, C.byref(need)))
    key = (test_u8.device, torch.cuda.current_stream(test_u8.device).cuda_stream)
    ws = _metric_ws.get(key)
    if ws is None or ws.numel() < need.value:
        ws = _metric_ws[key] = torch.empty(int(need.value), dtype=torch.uint8, device=test_u8.device)
    if out is None:
        out = torch.empty(b, _lib.FDSR_METRIC_FIELDS, dtype=torch.float64, device=test_u8.device)
    flags = _lib.FDSR_SSIM_UNIFORM7 | (_lib.FDSR_SSIM_GAUSS11 if gauss else 0)
    _lib.check(None, lib.fdsr_image_metrics_u8(None, C.c_void_p(test_u8.data_ptr()), C.c_void_p(truth_u8.data_ptr()), b, h, w, c, flags,
                                               C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(key[1])))
    return out


def metrics_from_sums(sums, shape_hwc, scale=4):
    """Per-image (mse, psnr, ssim, ergas[, ssim_gauss]) from one row of image_metric_sums, with the scalar formulas of
    compare_mse / compare_psnr / calculate_ergas above (the sums are exact integers, so these three equal the host path
    bit for bit) and SSIM = map sum / positions."""
    h, w, c = shape_hwc
    n = float(h * w * c)
    sse, s_test, ss7, n7, ss11, n11 = (float(x) for x in sums[:6])
    mse = sse / n
    psnr = float('inf') if mse == 0 else 10 * math.log10((255.0 ** 2) / mse)
    mean2 = (s_test / n) ** 2
    ergas = float(100.0 * np.sqrt(np.float64(mse) / mean2 / c) / scale)
    out = {'mse': mse, 'psnr': psnr, 'ssim': ss7 / n7 if n7 else float('nan'), 'ergas': ergas}
    if n11:
        out['ssim_gauss'] = ss11 / n11
    return out


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
