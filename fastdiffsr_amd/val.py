"""The val phase of the reference's driver (`python sr_mfe.py -p val -c config/sr_fastdiffsr_test_64_256.json`,
FastDiffSR/sr_mfe.py:257-378) on the HIP engine:

    python -m fastdiffsr_amd.val -c config/sr_fastdiffsr_test_64_256.json [--batch 16] [--cond-from-lr]
    python -m fastdiffsr_amd.val -c config/sr_fastdiffsr_infer_x4.json --infer        # infer.py

Same config files, same dataset folders, same per-image metrics and log lines (MSE / PSNR / SSIM as
skimage.measure computes them, ERGAS as core/metrics.py:147-152; LPIPS needs AlexNet weights and is left out),
same `{results}/{step}_{idx}_sr.tif` outputs.  Differences, all opt-in or harmless:
  * `--batch N` samples N images per loop (the reference's val loader is batch 1 and its sampler crashes for
    more); every image is still its own independent chain
  * `continous=False`: the reference asks for the 8 intermediate frames and keeps only the last
  * `--cond-from-lr`: the conditioning image is resized from `lr_*` on the GPU (bit-identical to the offline PIL
    bicubic) instead of being read from `sr_*`
  * under `torch.distributed.run` the images are sharded over the ranks and the metric sums all-reduced
"""
import argparse
import logging
import os
import time

import numpy as np
import torch

from . import metrics as M
from .config import load_config
from .data import lr_to_sr
from .dataset import create_dataset
from .model import create_model
from .parallel import shard_range

logger = logging.getLogger('base')


def _collate(items):
    out = {k: torch.stack([it[k] for it in items]) for k in items[0] if k != 'Index'}
    out['Index'] = [it['Index'] for it in items]
    return out


def run(opt, batch=1, cond_from_lr=False, precision='f16x3', results=None, max_images=None, rank=0, world=1,
        save_images=True, log=print, infer=False, diffusion=None, step=None, epoch=None):
    """infer=True is the reference's infer.py (:62-110): the same loop, `{step}_{idx}_sr.png` outputs, timing, no metrics.
    diffusion: an existing model (the validation pass inside the training loop, sr_mfe.py:122-244); else one is created."""
    val_opt = opt['datasets']['val']
    dataset = create_dataset(val_opt, 'val', cond_from_lr=cond_from_lr)
    n_total = len(dataset) if max_images is None else min(len(dataset), max_images)
    lo, hi = shard_range(n_total, rank, world)
    scale = int(val_opt['r_resolution']) // int(val_opt['l_resolution'])
    if diffusion is None:
        diffusion = create_model(opt)                                                 # sr_mfe.py:60
    diffusion.netG.precision = precision
    diffusion.set_new_noise_schedule(opt['model']['beta_schedule']['val'], schedule_phase='val')   # sr_mfe.py:66-67, :131-132
    current_step = diffusion.begin_step if step is None else step
    current_epoch = diffusion.begin_epoch if epoch is None else epoch
    result_path = results or (opt.get('path') or {}).get('results') or 'results'
    if save_images:
        os.makedirs(result_path, exist_ok=True)
    sums = np.zeros(9, dtype=np.float64)       # bic mse/psnr/ssim/ergas, sr mse/psnr/ssim/ergas, count
    t_sample = 0.0
    for b0 in range(lo, hi, batch):
        items = [dataset[i] for i in range(b0, min(b0 + batch, hi))]
        data = _collate(items)
        if cond_from_lr:
            data['SR'] = lr_to_sr(data.pop('LR_u8').cuda(), int(val_opt['r_resolution']), int(val_opt['r_resolution']))
        idxs = data.pop('Index')
        diffusion.feed_data(data)
        torch.cuda.synchronize()
        t0 = time.time()
        diffusion.test(continous=False)
        torch.cuda.synchronize()
        t_sample += time.time() - t0
        logger.info('inference time (s): {:.4f} for {} image(s)'.format(time.time() - t0, len(idxs)))   # sr_mfe.py:279-284
        sr_batch = diffusion.SR
        if sr_batch.dim() == 3:       # the ddpm / tesr siblings return ret_img[-1]: one image (their own convention)
            if len(idxs) != 1:
                raise ValueError("which_model_G in ('ddpm', 'tesr') returns one image per call: use --batch 1")
            sr_batch = sr_batch[None]
        for j, index in enumerate(idxs):
            idx = index + 1                                                           # sr_mfe.py:274 counts from 1
            hr_img = M.tensor2img(diffusion.data['HR'][j])
            fake_img = M.tensor2img(diffusion.data['SR'][j])                          # the bicubic image ('INF')
            sr_img = M.tensor2img(sr_batch[j])
            if save_images:
                from PIL import Image
                Image.fromarray(sr_img).save('{}/{}_{}_sr.{}'.format(result_path, current_step, idx, 'png' if infer else 'tif'))
            if infer:
                sums[8] += 1.0
                continue
            sums += np.array([M.compare_mse(fake_img, hr_img), M.compare_psnr(fake_img, hr_img), M.compare_ssim(fake_img, hr_img),
                              M.calculate_ergas(fake_img, hr_img, scale=scale),
                              M.compare_mse(sr_img, hr_img), M.compare_psnr(sr_img, hr_img), M.compare_ssim(sr_img, hr_img),
                              M.calculate_ergas(sr_img, hr_img, scale=scale), 1.0])
    if world > 1:
        import torch.distributed as dist
        t = torch.from_numpy(sums).cuda() if dist.get_backend() == 'nccl' else torch.from_numpy(sums)
        dist.all_reduce(t)
        sums = t.cpu().numpy()
    n = max(sums[8], 1.0)
    avg = sums[:8] / n
    res = dict(images=int(sums[8]), bic_mse=avg[0], bic_psnr=avg[1], bic_ssim=avg[2], bic_ergas=avg[3],
               sr_mse=avg[4], sr_psnr=avg[5], sr_ssim=avg[6], sr_ergas=avg[7],
               sample_seconds_this_rank=t_sample, result_path=result_path)
    if rank == 0 and infer:
        log('inference: {} images, {:.4f} s per image on this rank (batch {})'.format(int(sums[8]), t_sample / max(hi - lo, 1), batch))
    elif rank == 0:
        log('<epoch:{:3d}, iter:{:8,d}> bic_mse: {:.5e}, bic_psnr: {:.5e}, bic_ssim: {:.5e}, bic_ergas: {:.5e}'.format(
            current_epoch, current_step, *avg[:4]))
        log('<epoch:{:3d}, iter:{:8,d}> sr_mse: {:.5e}, sr_psnr: {:.5e}, sr_ssim: {:.5e}, sr_ergas: {:.5e}'.format(
            current_epoch, current_step, *avg[4:8]))
    return res


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('-c', '--config', required=True, help='JSON file for configuration (the reference\'s own)')
    ap.add_argument('-p', '--phase', choices=['val'], default='val')
    ap.add_argument('-gpu', '--gpu_ids', default=None)
    ap.add_argument('-debug', '-d', action='store_true')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--cond-from-lr', action='store_true')
    ap.add_argument('--precision', default='f16x3', choices=['f32', 'f16x3', 'bf16'])
    ap.add_argument('--results', default=None)
    ap.add_argument('--max-images', type=int, default=None)
    ap.add_argument('--no-save', action='store_true')
    ap.add_argument('--infer', action='store_true', help="the reference's infer.py: png outputs and timing, no metrics")
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')
    opt = load_config(a.config, phase=a.phase, gpu_ids=a.gpu_ids, debug=a.debug)
    log = print
    if not a.no_save and rank == 0 and (opt.get('path') or {}).get('log'):
        # sr_mfe.py:50-54: experiments/<name>_<timestamp>/logs/{train,val}.log; the two summary lines go to val.log
        from .config import setup_logger
        setup_logger(None, opt['path']['log'], 'train', screen=True)
        val_logger = setup_logger('val', opt['path']['log'], 'val')
        log = lambda msg: (print(msg), val_logger.info(msg))     # noqa: E731
        if a.results is None:
            a.results = opt['path'].get('results')
    res = run(opt, batch=a.batch, cond_from_lr=a.cond_from_lr, precision=a.precision, results=a.results,
              max_images=a.max_images, rank=rank, world=world, save_images=not a.no_save, infer=a.infer, log=log)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return res


if __name__ == '__main__':
    main()
