"""Host side of the val harness: the image-folder dataset (reference data/LRHR_dataset.py, data/util.py)
and the skimage-style metrics the reference's val loop calls (sr_mfe.py:313-333)."""
import os

import numpy as np
import pytest
import torch

from fastdiffsr_amd import metrics as M
from fastdiffsr_amd.dataset import LRHRDataset, create_dataset, to_tensor


def make_dataset(root, n=3, l=16, r=64, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    for sub in ('hr_%d' % r, 'lr_%d' % l, 'sr_%d_%d' % (l, r)):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i in range(n):
        yy, xx = np.mgrid[0:r, 0:r]
        hr = np.stack([127 + 100 * np.sin(xx / (3.0 + i) + c) * np.cos(yy / (5.0 + c)) for c in range(3)], -1)
        hr = np.clip(hr + rng.normal(0, 4, hr.shape), 0, 255).astype(np.uint8)
        him = Image.fromarray(hr)
        lim = him.resize((l, l), Image.BICUBIC)
        sim = lim.resize((r, r), Image.BICUBIC)
        name = '%05d.png' % (n - i)                 # written in reverse: pairing is by SORTED name
        him.save(os.path.join(root, 'hr_%d' % r, name))
        lim.save(os.path.join(root, 'lr_%d' % l, name))
        sim.save(os.path.join(root, 'sr_%d_%d' % (l, r), name))
    return root


def test_dataset_matches_reference_conventions(tmp_path):
    from PIL import Image
    root = make_dataset(str(tmp_path))
    ds = LRHRDataset(root, 'img', l_resolution=16, r_resolution=64, split='val', data_len=-1, need_LR=True)
    assert len(ds) == 3
    it = ds[0]
    assert set(it) == {'HR', 'SR', 'LR', 'Index'} and it['Index'] == 0
    assert it['HR'].shape == (3, 64, 64) and it['SR'].shape == (3, 64, 64) and it['LR'].shape == (3, 16, 16)
    raw = np.asarray(Image.open(os.path.join(root, 'hr_64', '00001.png')).convert('RGB'))     # sorted first
    want = torch.from_numpy(raw.transpose(2, 0, 1).copy()).float().div(255) * 2 - 1          # ToTensor()*2-1
    assert torch.equal(it['HR'], want) and it['HR'].min() >= -1 and it['HR'].max() <= 1
    assert len(LRHRDataset(root, 'img', 16, 64, data_len=2)) == 2
    assert len(LRHRDataset(root, 'img', 16, 64, data_len=99)) == 3
    with pytest.raises(NotImplementedError):
        LRHRDataset(root, 'lmdb', 16, 64)
    with pytest.raises(AssertionError):
        LRHRDataset(os.path.join(root, 'nope'), 'img', 16, 64)
    ds2 = create_dataset({'dataroot': root, 'datatype': 'img', 'l_resolution': 16, 'r_resolution': 64, 'data_len': -1,
                          'mode': 'HR'}, 'val', cond_from_lr=True)
    it2 = ds2[1]
    assert 'SR' not in it2 and it2['LR_u8'].dtype == torch.uint8 and it2['LR_u8'].shape == (16, 16, 3)
    # the pipelined loops ship the decoded bytes and finish the transform on the device: same pixels
    raw_item = ds.load_u8(0)
    assert set(raw_item) == {'HR', 'SR', 'LR', 'Index'} and raw_item['HR'].dtype == np.uint8 and raw_item['HR'].shape == (64, 64, 3)
    for k in ('HR', 'SR', 'LR'):
        assert torch.equal(to_tensor(Image.fromarray(raw_item[k])), it[k])
    tr = LRHRDataset(root, 'img', l_resolution=16, r_resolution=64, split='train', data_len=-1, need_LR=True)
    torch.manual_seed(4)
    a = [tr[i] for i in range(3)]
    torch.manual_seed(4)
    b = [tr.load_u8(i) for i in range(3)]                    # same RNG consumption, same flips
    for x, y in zip(a, b):
        for k in ('HR', 'SR', 'LR'):
            assert torch.equal(to_tensor(Image.fromarray(y[k])), x[k])
    g = to_tensor(Image.fromarray(np.full((4, 4), 255, np.uint8)))
    assert g.shape == (1, 4, 4) and float(g.max()) == 1.0


def test_skimage_style_metrics():
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (32, 40, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    assert M.compare_mse(a, a) == 0 and M.compare_psnr(a, a) == float('inf')
    mse = np.mean((a.astype(np.float64) - b) ** 2)
    assert abs(M.compare_mse(a, b) - mse) < 1e-9
    assert abs(M.compare_psnr(a, b) - 10 * np.log10(255 ** 2 / mse)) < 1e-9
    assert abs(M.compare_psnr(a, b) - M.calculate_psnr(a, b)) < 1e-9
    assert abs(M.compare_ssim(a, a) - 1.0) < 1e-12
    # brute-force SSIM (uniform 7x7 window, sample covariance, interior pixels) on one channel
    X, Y = a[..., 0].astype(np.float64), b[..., 0].astype(np.float64)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for y in range(3, X.shape[0] - 3):
        for x in range(3, X.shape[1] - 3):
            wx, wy = X[y - 3:y + 4, x - 3:x + 4].ravel(), Y[y - 3:y + 4, x - 3:x + 4].ravel()
            ux, uy = wx.mean(), wy.mean()
            vx, vy = wx.var(ddof=1), wy.var(ddof=1)
            vxy = ((wx - ux) * (wy - uy)).sum() / 48
            vals.append((2 * ux * uy + C1) * (2 * vxy + C2) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
    assert abs(M.compare_ssim(a[..., 0], b[..., 0], multichannel=False) - np.mean(vals)) < 1e-9
    s = M.compare_ssim(a, b)
    assert 0 < s < 1 and abs(s - np.mean([M.compare_ssim(a[..., c], b[..., c], multichannel=False) for c in range(3)])) < 1e-12
