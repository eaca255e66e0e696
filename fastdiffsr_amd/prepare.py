"""Offline pair preparation: the folder tool of the reference (`python data/prepare_data_mfe_dm.py -p <images> -o <out> --size 64,256`,
FastDiffSR/data/prepare_data_mfe_dm.py:17-40,100-187).  Every image under `--path` becomes three files of the dataset layout
`LRHRDataset` reads: `<out>_<l>_<r>/lr_<l>/NNNNN.tif`, `hr_<r>/NNNNN.tif`, `sr_<l>_<r>/NNNNN.tif` with

    lr = center_crop(resize(img, l)),  hr = center_crop(resize(img, r)),  sr = center_crop(resize(lr, r))      (:30-40)

`resize(img, size)` is torchvision's: the SHORTER edge becomes `size`, aspect kept (long edge = int(size * long / short)), with Pillow's
resampler (bicubic by default) -- the same Pillow calls as the reference, so the files are the reference's files.  Decoding, resizing
and writing run on worker threads (Pillow releases the GIL).  `--lmdb` writes the reference's container instead (:24-27,82-92: the three
images as encoded tif files under `lr_<l>_<NNNNN>`, `hr_<r>_<NNNNN>`, `sr_<l>_<r>_<NNNNN>`, the running count under `length`; the
`lmdb` module is imported only then -- `LRHRDataset(datatype='lmdb')` reads it back).  On-line, `val.py --cond-from-lr` / `data.lr_to_sr` build the `sr_*` member on the GPU instead, bit-identical to this."""
import argparse
import os
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path


def _resize_shorter(img, size, resample):
    """torchvision.transforms.functional.resize(img, int size): shorter edge -> size (no-op when it already is)."""
    w, h = img.size
    short, long_ = (w, h) if w <= h else (h, w)
    if short == size:
        return img
    new_short, new_long = size, int(size * long_ / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    return img.resize((nw, nh), resample)


def _center_crop(img, size):
    """torchvision center_crop(img, size) for images at least `size` in both directions."""
    w, h = img.size
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def resize_and_convert(img, size, resample):                      # prepare_data_mfe_dm.py:17-21
    if img.size[0] != size:
        img = _resize_shorter(img, size, resample)
        img = _center_crop(img, size)
    return img


def resize_multiple(img, sizes=(64, 256), resample=None):         # :30-40
    from PIL import Image
    resample = Image.BICUBIC if resample is None else resample
    lr_img = resize_and_convert(img, sizes[0], resample)
    hr_img = resize_and_convert(img, sizes[1], resample)
    sr_img = resize_and_convert(lr_img, sizes[1], resample)
    return lr_img, hr_img, sr_img


def image_convert_bytes(img):                                      # :24-27
    from io import BytesIO
    buffer = BytesIO()
    img.save(buffer, format='tiff')
    return buffer.getvalue()


def prepare(img_path, out_path, n_worker=4, sizes=(64, 256), resample=None, ext='tif', lmdb_save=False):
    """:100-160.  Returns the number of images written."""
    from PIL import Image
    files = sorted(p for p in Path(str(img_path)).glob('**/*') if p.is_file())
    dirs = ['{}/lr_{}'.format(out_path, sizes[0]), '{}/hr_{}'.format(out_path, sizes[1]), '{}/sr_{}_{}'.format(out_path, sizes[0], sizes[1])]
    keys = ['lr_{}_{{}}'.format(sizes[0]), 'hr_{}_{{}}'.format(sizes[1]), 'sr_{}_{}_{{}}'.format(sizes[0], sizes[1])]
    env = None
    if lmdb_save:
        import lmdb
        env = lmdb.open(str(out_path), map_size=1024 ** 4, readahead=False)     # :113
    else:
        for d in dirs:
            os.makedirs(d, exist_ok=True)

    def one(f):
        img = Image.open(f).convert('RGB')                         # :43-44
        name = f.name.split('.')[0].zfill(5)                       # :49, :146
        imgs = resize_multiple(img, sizes, resample)
        if env is None:
            for im, d in zip(imgs, dirs):
                im.save('{}/{}.{}'.format(d, name, ext))
            return None
        return name, [image_convert_bytes(im) for im in imgs]      # encoded on the worker thread, written by the caller

    total = 0
    with ThreadPoolExecutor(max_workers=max(1, int(n_worker))) as ex:
        for res in ex.map(one, files):
            total += 1
            if env is not None:
                name, blobs = res
                with env.begin(write=True) as txn:                 # :82-92
                    for k, b in zip(keys, blobs):
                        txn.put(k.format(name).encode('utf-8'), b)
                    txn.put('length'.encode('utf-8'), str(total).encode('utf-8'))
    if env is not None and hasattr(env, 'close'):
        env.close()
    return total


def main(argv=None):
    from PIL import Image
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--path', '-p', type=str, default='../dataset/Train')
    ap.add_argument('--out', '-o', type=str, default='../dataset/Train')
    ap.add_argument('--size', type=str, default='64,256')
    ap.add_argument('--n_worker', type=int, default=4)
    ap.add_argument('--resample', type=str, default='bicubic', choices=['bilinear', 'bicubic'])
    ap.add_argument('--lmdb', '-l', action='store_true')
    a = ap.parse_args(argv)
    sizes = [int(s.strip()) for s in a.size.split(',')]
    out = '{}_{}_{}'.format(a.out, sizes[0], sizes[1])              # :182
    n = prepare(a.path, out, a.n_worker, sizes=sizes, resample={'bilinear': Image.BILINEAR, 'bicubic': Image.BICUBIC}[a.resample],
                lmdb_save=a.lmdb)
    print('{} images -> {}'.format(n, out))
    return n


if __name__ == '__main__':
    main()
