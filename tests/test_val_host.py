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
        LRHRDataset(root, 'zip', 16, 64)
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


def test_prepare_folder_tool_matches_pillow_and_feeds_the_dataset(tmp_path):
    """data/prepare_data_mfe_dm.py: lr = crop(resize(img, l)), hr = crop(resize(img, r)), sr = crop(resize(lr, r)) with torchvision's
    shorter-edge resize; square and non-square sources; the folders are what LRHRDataset reads."""
    from PIL import Image
    from fastdiffsr_amd import prepare as P
    rng = np.random.default_rng(2)
    src = tmp_path / 'src'
    src.mkdir()
    for name, (h, w) in (('7', (96, 96)), ('12', (80, 120)), ('3', (130, 72))):
        Image.fromarray(rng.integers(0, 256, (h, w, 3)).astype(np.uint8)).save(src / f'{name}.png')
    n = P.main(['-p', str(src), '-o', str(tmp_path / 'set'), '--size', '16,64', '--n_worker', '2'])
    out = tmp_path / 'set_16_64'
    assert n == 3 and sorted(os.listdir(out)) == ['hr_64', 'lr_16', 'sr_16_64']
    assert sorted(os.listdir(out / 'hr_64')) == ['00003.tif', '00007.tif', '00012.tif']
    # square source: plain Pillow resizes
    img = Image.open(src / '7.png').convert('RGB')
    lr = img.resize((16, 16), Image.BICUBIC)
    assert np.array_equal(np.asarray(Image.open(out / 'lr_16' / '00007.tif')), np.asarray(lr))
    assert np.array_equal(np.asarray(Image.open(out / 'hr_64' / '00007.tif')), np.asarray(img.resize((64, 64), Image.BICUBIC)))
    assert np.array_equal(np.asarray(Image.open(out / 'sr_16_64' / '00007.tif')), np.asarray(lr.resize((64, 64), Image.BICUBIC)))
    # 80 x 120 (h x w): shorter edge 80 -> 64, long edge int(64 * 120 / 80) = 96, centre crop of 64 columns
    img = Image.open(src / '12.png').convert('RGB')
    hr = img.resize((96, 64), Image.BICUBIC).crop((16, 0, 80, 64))
    assert np.array_equal(np.asarray(Image.open(out / 'hr_64' / '00012.tif')), np.asarray(hr))
    ds = LRHRDataset(str(out), 'img', l_resolution=16, r_resolution=64, split='val', data_len=-1, need_LR=True)
    assert len(ds) == 3 and ds[0]['HR'].shape == (3, 64, 64) and ds[0]['LR'].shape == (3, 16, 16)


def test_hr_mask_folder_rides_the_sr_hr_stack(tmp_path):
    """LRHR_dataset.py:33-40,100-121: any `img_mask` but 'no' adds `hr_mask_{r}`; 'HR_Mask' shares the [SR, HR] flip decision in the
    train split, and DDPM.get_current_visuals hands it through (model.py:105-106)."""
    from PIL import Image
    import shutil
    root = make_dataset(str(tmp_path))
    shutil.copytree(os.path.join(root, 'hr_64'), os.path.join(root, 'hr_mask_64'))
    ds = LRHRDataset(root, 'img', 16, 64, split='val', img_mask='yes')
    it = ds[1]
    assert sorted(it) == ['HR', 'HR_Mask', 'Index', 'SR'] and torch.equal(it['HR_Mask'], it['HR'])
    tr = LRHRDataset(root, 'img', 16, 64, split='train', need_LR=True, img_mask='yes')
    torch.manual_seed(3)
    seen = set()
    for _ in range(8):
        a = tr[0]
        assert torch.equal(a['HR_Mask'], a['HR'])                      # mirrored together, whatever the draw
        seen.add(bool(torch.equal(a['HR'], ds[0]['HR'])))
        u = tr.load_u8(0)
        assert np.array_equal(u['HR_Mask'], u['HR'])
    assert seen == {True, False}
    assert 'HR_Mask' not in LRHRDataset(root, 'img', 16, 64, split='val')[0]


class _DictLmdb:
    """A dict-backed stand-in for the `lmdb` module (absent from this image), with the calls the reference makes:
    lmdb.open(path, ...) -> env; env.begin(write=) as txn; txn.get / txn.put (prepare_data_mfe_dm.py:82-92,113; LRHR_dataset.py:19-23,61-92)."""
    stores = {}

    class _Txn:
        def __init__(self, d, write):
            self.d, self.write = d, write

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def get(self, k):
            assert isinstance(k, bytes)
            return self.d.get(k)

        def put(self, k, v):
            assert self.write and isinstance(k, bytes) and isinstance(v, bytes)
            self.d[k] = v

    class _Env:
        def __init__(self, d, readonly):
            self.d, self.readonly = d, readonly

        def begin(self, write=False):
            assert not (write and self.readonly)
            return _DictLmdb._Txn(self.d, write)

        def close(self):
            pass

    @classmethod
    def open(cls, path, readonly=False, **kw):
        return cls._Env(cls.stores.setdefault(str(path), {}), readonly)


def test_lmdb_container_written_by_the_tool_and_read_by_the_dataset(tmp_path, monkeypatch):
    """`--lmdb` (prepare_data_mfe_dm.py:24-27,82-92) and LRHRDataset(datatype='lmdb') (LRHR_dataset.py:18-27,61-92) over a stand-in
    lmdb module: same keys, encoded tif entries, `length`; items equal the image-folder dataset's; a missing index is re-drawn."""
    import sys
    import random
    from PIL import Image
    from fastdiffsr_amd import prepare as P
    monkeypatch.setitem(sys.modules, 'lmdb', _DictLmdb)
    _DictLmdb.stores.clear()
    rng = np.random.default_rng(5)
    src = tmp_path / 'src'
    src.mkdir()
    for name in ('0', '1', '2'):
        Image.fromarray(rng.integers(0, 256, (40, 40, 3)).astype(np.uint8)).save(src / f'{name}.png')
    assert P.main(['-p', str(src), '-o', str(tmp_path / 'db'), '--size', '8,32', '--lmdb']) == 3
    assert P.main(['-p', str(src), '-o', str(tmp_path / 'dir'), '--size', '8,32']) == 3
    store = _DictLmdb.stores[str(tmp_path / 'db_8_32')]
    assert store[b'length'] == b'3'
    assert sorted(store) == sorted([b'length'] + [f'{k}_{i:05d}'.encode() for k in ('lr_8', 'hr_32', 'sr_8_32') for i in range(3)])
    a = LRHRDataset(str(tmp_path / 'db_8_32'), 'lmdb', l_resolution=8, r_resolution=32, split='val', need_LR=True)
    b = LRHRDataset(str(tmp_path / 'dir_8_32'), 'img', l_resolution=8, r_resolution=32, split='val', need_LR=True)
    assert len(a) == len(b) == 3
    for i in range(3):
        x, y = a[i], b[i]
        assert sorted(x) == sorted(y) == ['HR', 'Index', 'LR', 'SR']
        for k in ('HR', 'SR', 'LR'):
            assert torch.equal(x[k], y[k])
        u, v = a.load_u8(i), b.load_u8(i)
        for k in ('HR', 'SR', 'LR'):
            assert np.array_equal(u[k], v[k])
    # an index without an HR entry: the reference draws random indices until one is valid (LRHR_dataset.py:77-91)
    del store[b'hr_32_00001']
    random.seed(0)
    got = a[1]['HR']
    assert any(torch.equal(got, b[i]['HR']) for i in (0, 2))
    assert len(LRHRDataset(str(tmp_path / 'db_8_32'), 'lmdb', l_resolution=8, r_resolution=32, data_len=2)) == 2
    with pytest.raises(NotImplementedError):
        LRHRDataset(str(tmp_path / 'dir_8_32'), 'zip')
