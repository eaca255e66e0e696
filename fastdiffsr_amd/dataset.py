"""The reference's image-folder dataset (FastDiffSR/data/LRHR_dataset.py, data/util.py, data/__init__.py)
`{dataroot}/hr_{r}`, `sr_{l}_{r}` (the bicubic conditioning image), `lr_{l}`, files paired by sorted order.
Tensors are what `transform_augment(split, min_max=(-1, 1))` gives: ToTensor (uint8 / 255, CHW), in the train split
ONE RandomHorizontalFlip(p=0.5) decision for the stacked [SR, HR] (and one of its own for LR, util.py:66-88), then
* 2 - 1.  `datatype='lmdb'` reads the container the reference's prepare tool writes (LRHR_dataset.py:18-27,61-92: keys
`hr_{r}_{NNNNN}`, `sr_{l}_{r}_{NNNNN}`, `lr_{l}_{NNNNN}` holding encoded image files, `length`; an index whose HR or SR entry is
missing is replaced by `random.randint` draws until one is found); the `lmdb` module is imported when such a dataset is opened.

`cond_from_lr=True` builds SR on the GPU from the LR image instead of reading `sr_*` (bit-identical to the
reference's offline PIL bicubic, see data.lr_to_sr): the val loop then needs only LR + HR folders."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP', 'tif']


def is_image_file(filename):                                   # util.py:12-13 (note: 'tif' has no dot there either)
    return any(filename.endswith(ext) for ext in IMG_EXTENSIONS)


def get_paths_from_images(path):                               # util.py:16-25
    assert os.path.isdir(path), '{:s} is not a valid directory'.format(path)
    images = []
    for dirpath, _, fnames in sorted(os.walk(path)):
        for fname in sorted(fnames):
            if is_image_file(fname):
                images.append(os.path.join(dirpath, fname))
    assert images, '{:s} has no valid image file'.format(path)
    return sorted(images)


def to_tensor(pil_img, min_max=(-1, 1)):
    """torchvision ToTensor() then `img * (max - min) + min` (util.py:64-75), for 8-bit RGB images."""
    a = np.asarray(pil_img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)
    return t * (min_max[1] - min_max[0]) + min_max[0]


class LRHRDataset(Dataset):
    def __init__(self, dataroot, datatype='img', l_resolution=64, r_resolution=256, split='val', data_len=-1,
                 need_LR=False, img_mask='no', cond_from_lr=False):
        if datatype not in ('img', 'lmdb'):
            raise NotImplementedError('data_type [{:s}] is not recognized.'.format(str(datatype)))
        self.datatype = datatype
        self.l_res, self.r_res, self.split = l_resolution, r_resolution, split
        self.need_LR = need_LR or cond_from_lr
        self.cond_from_lr = cond_from_lr
        self.env = None
        if datatype == 'lmdb':                                     # LRHR_dataset.py:18-27
            import lmdb
            self.env = lmdb.open(dataroot, readonly=True, lock=False, readahead=False, meminit=False)
            with self.env.begin(write=False) as txn:
                self.dataset_len = int(txn.get('length'.encode('utf-8')))
            self.hr_path = self.sr_path = self.lr_path = self.hr_mask_path = None
        else:
            self.hr_path = get_paths_from_images('{}/hr_{}'.format(dataroot, r_resolution))
            self.sr_path = None if cond_from_lr else get_paths_from_images('{}/sr_{}_{}'.format(dataroot, l_resolution, r_resolution))
            self.lr_path = get_paths_from_images('{}/lr_{}'.format(dataroot, l_resolution)) if self.need_LR else None
            # LRHR_dataset.py:33-40: any img_mask but 'no' adds the folder hr_mask_{r}; its image rides the [SR, HR] stack ('HR_Mask')
            self.hr_mask_path = get_paths_from_images('{}/hr_mask_{}'.format(dataroot, r_resolution)) if img_mask != 'no' else None
            self.dataset_len = len(self.hr_path)
        self.data_len = self.dataset_len if data_len is None or data_len <= 0 else min(data_len, self.dataset_len)

    def _open(self, index):
        """The item's PIL images {'HR', 'SR'?, 'LR'?} from the folders or from the lmdb container."""
        from PIL import Image
        want_sr = not self.cond_from_lr
        if self.env is None:
            out = {'HR': Image.open(self.hr_path[index]).convert('RGB')}
            if want_sr:
                out['SR'] = Image.open(self.sr_path[index]).convert('RGB')
            if self.hr_mask_path:
                out['HR_Mask'] = Image.open(self.hr_mask_path[index]).convert('RGB')
            if self.need_LR:
                out['LR'] = Image.open(self.lr_path[index]).convert('RGB')
            return out
        import random
        from io import BytesIO
        with self.env.begin(write=False) as txn:                   # LRHR_dataset.py:61-92

            def entries(i):
                n = str(i).zfill(5)
                hr = txn.get('hr_{}_{}'.format(self.r_res, n).encode('utf-8'))
                sr = txn.get('sr_{}_{}_{}'.format(self.l_res, self.r_res, n).encode('utf-8'))
                lr = txn.get('lr_{}_{}'.format(self.l_res, n).encode('utf-8')) if self.need_LR else None
                return hr, sr, lr

            hr, sr, lr = entries(index)
            while hr is None or (want_sr and sr is None):          # "skip the invalid index" (cond_from_lr: no sr_* entry needed)
                hr, sr, lr = entries(random.randint(0, self.data_len - 1))
            out = {'HR': Image.open(BytesIO(hr)).convert('RGB')}
            if want_sr:
                out['SR'] = Image.open(BytesIO(sr)).convert('RGB')
            if self.need_LR:
                out['LR'] = Image.open(BytesIO(lr)).convert('RGB')
        return out

    def __len__(self):
        return self.data_len

    def draw_flips(self):
        """The train split's two RandomHorizontalFlip decisions of one item (util.py:66-88), drawn from torch's generator in the
        order __getitem__ draws them; (False, False) elsewhere.  ThreadedBatchLoader draws them on the consumer thread, in item
        order, so that a run repeats under torch.manual_seed however the decode threads are scheduled."""
        if self.split != 'train':
            return False, False
        flip_hr = bool(torch.rand(1) < 0.5)
        flip_lr = bool(torch.rand(1) < 0.5) if self.need_LR else False
        return flip_hr, flip_lr

    def load_u8(self, index, flips=None):
        """The same item as decoded uint8 HWC arrays ('HR', 'SR', 'LR'), the train-split flips applied: what the pipelined
        loops (val.py, train.py) ship to the GPU, where metrics.u8_to_tensor finishes the transform (one byte per sample over
        PCIe, and the decode -- which releases the GIL -- is all a loader thread does).  Same RNG consumption as __getitem__
        (flips=None), or the decisions handed in (draw_flips)."""
        out = {k: np.array(v, dtype=np.uint8) for k, v in self._open(index).items()}
        out['Index'] = index
        if self.split == 'train':
            flip_hr, flip_lr = self.draw_flips() if flips is None else flips
            if self.cond_from_lr:
                flip_lr = flip_hr          # the conditioning image built from LR stands in for SR: it mirrors with HR
            if flip_hr:
                for k in ('SR', 'HR', 'HR_Mask'):
                    if k in out:
                        out[k] = np.ascontiguousarray(out[k][:, ::-1])
            if flip_lr:
                out['LR'] = np.ascontiguousarray(out['LR'][:, ::-1])
        return out

    def __getitem__(self, index):
        imgs = self._open(index)
        out = {'HR': to_tensor(imgs['HR']), 'Index': index}
        if 'SR' in imgs:
            out['SR'] = to_tensor(imgs['SR'])
        if 'HR_Mask' in imgs:
            out['HR_Mask'] = to_tensor(imgs['HR_Mask'])
        if self.need_LR:
            out['LR'] = to_tensor(imgs['LR'])
            if self.cond_from_lr:
                out['LR_u8'] = torch.from_numpy(np.array(imgs['LR'], dtype=np.uint8))
        if self.split == 'train':
            # util.py:66-75: hflip(torch.stack([SR, HR])) -- torchvision's RandomHorizontalFlip draws torch.rand(1) once
            # for the whole stack; util.py:77-88 does the same, separately, for [LR]
            flip_hr = bool(torch.rand(1) < 0.5)
            if flip_hr:
                for k in ('SR', 'HR', 'HR_Mask'):
                    if k in out:
                        out[k] = out[k].flip(-1)
            if 'LR' in out and torch.rand(1) < 0.5:
                out['LR'] = out['LR'].flip(-1)
            # the conditioning image built later from LR_u8 stands in for SR: it must mirror with HR, not with LR's own draw
            if 'LR_u8' in out and flip_hr:
                out['LR_u8'] = out['LR_u8'].flip(1)
        return out


def create_dataloader(dataset, dataset_opt, phase):             # data/__init__.py:7-21
    from torch.utils.data import DataLoader
    if phase == 'train':
        return DataLoader(dataset, batch_size=dataset_opt['batch_size'], shuffle=dataset_opt['use_shuffle'],
                          num_workers=dataset_opt['num_workers'], pin_memory=True)
    if phase == 'val':
        return DataLoader(dataset, batch_size=1, shuffle=False, num_workers=1, pin_memory=True)
    raise NotImplementedError('Dataloader [{:s}] is not found.'.format(phase))


class ThreadedBatchLoader:
    """The training loop's loader (the reference: torch DataLoader with `num_workers` worker processes, data/__init__.py:9-15)
    as worker THREADS of this process: PIL decodes outside the GIL, nothing is forked off a process that holds the GPU, and a
    batch is handed over as stacked uint8 arrays {'HR','SR','LR': [B,H,W,3], 'Index': [...]} -- the caller ships the bytes and
    finishes ToTensor()*2-1 on the device (metrics.u8_to_tensor).  Order: torch.randperm(len, generator) per epoch when
    `shuffle` (RandomSampler's draw), else sequential; the flips are drawn on the consumer thread in item order."""

    def __init__(self, dataset, batch_size, shuffle=False, workers=4, generator=None, depth=3, stage=None):
        from concurrent.futures import ThreadPoolExecutor
        self.ds, self.bs, self.shuffle, self.gen, self.depth = dataset, int(batch_size), bool(shuffle), generator, depth
        # stage(key, [arrays], owner=) -> what the batch carries for that key: default the stacked array; the GPU loops pass val.HipOps.stage_host,
        # which stacks straight into pinned memory ON THE LOADER THREAD (the consumer then only issues the asynchronous copy)
        ops = getattr(stage, '__self__', None)           # a bound HipOps.stage_host: rings named by a token that is never reused
        self._ops = ops if hasattr(ops, 'release') else None
        self.owner = ops.new_owner() if hasattr(ops, 'new_owner') else id(self)
        self.stage = (lambda key, arrays: stage(key, arrays, owner=self.owner)) if stage else (lambda key, arrays: np.stack(arrays))
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(workers)))

    def __len__(self):
        return (len(self.ds) + self.bs - 1) // self.bs

    def close(self):
        """Give the pinned staging rings back (the loader itself lives for the whole run: one ring set, reused every epoch)."""
        if self._ops is not None:
            self._ops.release(self.owner)
        self.pool.shutdown(wait=False)

    def __iter__(self):
        n = len(self.ds)
        order = torch.randperm(n, generator=self.gen).tolist() if self.shuffle else list(range(n))
        batches = [order[i:i + self.bs] for i in range(0, n, self.bs)]
        futs, nxt = {}, 0

        def collate(item_futs):          # loader thread; its item jobs were queued first, so they are running or done
            items = [f.result() for f in item_futs]
            out = {key: self.stage(key, [it[key] for it in items]) for key in ('HR', 'SR', 'LR') if key in items[0]}
            out['Index'] = [it['Index'] for it in items]
            return out

        def submit_until(k):
            nonlocal nxt
            while nxt < len(batches) and nxt <= k:
                item_futs = [self.pool.submit(self.ds.load_u8, i, self.ds.draw_flips()) for i in batches[nxt]]
                futs[nxt] = self.pool.submit(collate, item_futs)
                nxt += 1

        # The consumer hands the GIL over around every launch / synchronisation; with Python's default 5 ms switch interval every
        # hand-back can cost it up to 5 ms while a decode thread is in a Python section (measured on the training loop: 16 ms per
        # step): 0.2 ms while this loader is being iterated.
        import sys
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(2e-4)
        try:
            yield from self._batches(batches, futs, submit_until)
        finally:
            sys.setswitchinterval(old_switch)

    def _batches(self, batches, futs, submit_until):
        submit_until(self.depth - 1)
        for k in range(len(batches)):
            submit_until(k)                    # (only if depth == 0)
            out = futs.pop(k).result()
            # the next decode jobs go to the pool right before the consumer takes this batch into its GPU call: the workers' Python
            # sections then run while the consumer sits in C without the GIL, not against this thread's own code
            submit_until(k + self.depth)
            yield out


def create_dataset(dataset_opt, phase, cond_from_lr=False):     # data/__init__.py:24-40
    return LRHRDataset(dataroot=dataset_opt['dataroot'], datatype=dataset_opt['datatype'],
                       l_resolution=dataset_opt['l_resolution'], r_resolution=dataset_opt['r_resolution'],
                       split=phase, data_len=dataset_opt['data_len'], need_LR=(dataset_opt['mode'] == 'LRHR'),
                       img_mask=dataset_opt.get('img_mask', 'no') or 'no', cond_from_lr=cond_from_lr)
