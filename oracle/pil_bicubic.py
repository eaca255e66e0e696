"""ORACLE -- TEST INFRASTRUCTURE ONLY.

Integer restatement of Pillow's 8-bit bicubic resize (`Image.resize(size, Image.BICUBIC)`,
libImaging/Resample.c: bicubic_filter a=-0.5, precompute_coeffs, normalize_coeffs_8bpc with
PRECISION_BITS = 22, horizontal pass then vertical pass with clip8 after each), which is how the
reference builds the conditioning "SR" image from the LR image
(FastDiffSR/data/prepare_data_mfe_dm.py:17-40 via torchvision's functional.resize on PIL images), and
of the val-time tensor transform (data/util.py:66-75: ToTensor() = uint8/255 in fp32, then *2 - 1).

Third-party algorithm: Pillow (the reference pins none; this image has Pillow 12.2.0).  Pinned by
outputs of PIL itself run in the build container (tests/golden/bicubic.npz, oracle/make_goldens.py).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _bicubic(x, a=-0.5):
    x = -x if x < 0.0 else x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """-> (bounds int [out,2] = (xmin, count), kk int32 [out,ksize]) as Pillow computes them."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)          # C cast: truncation toward zero
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):                   # normalize_coeffs_8bpc
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _pass(img, bounds, kk, axis):
    """One resample pass along `axis` of an HWC uint8 image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + src.shape[1:], np.int64)
    for xx in range(bounds.shape[0]):
        xmin, cnt = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out, 0, axis).astype(np.uint8)


def resize_bicubic_u8(img, out_h, out_w):
    """HWC uint8 -> HWC uint8, == np.asarray(Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC))."""
    h, w = img.shape[:2]
    if w != out_w:
        img = _pass(img, *precompute_coeffs(w, out_w), axis=1)      # horizontal first
    if h != out_h:
        img = _pass(img, *precompute_coeffs(h, out_h), axis=0)
    return img


def u8_to_model_tensor(img):
    """data/util.py:66-75 with min_max=(-1,1): ToTensor() (uint8 -> fp32 / 255, CHW) then x*2 + (-1)."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(np.transpose(img, (2, 0, 1)))).to(torch.float32).div(255)
    return t * 2.0 + (-1.0)
