"""The train phase of the reference's driver (`python sr_mfe.py -p train -c config/sr_fastdiffsr_train_64_256.json`,
FastDiffSR/sr_mfe.py:69-251) on the HIP engine:

    python -m fastdiffsr_amd.train -c config/sr_fastdiffsr_train_64_256.json [--precision f16x3|f32]
    python -m torch.distributed.run --nproc-per-node 8 -m fastdiffsr_amd.train -c ...     # data parallel

Same config files and dataset folders; per iteration `feed_data` + `optimize_parameters` (forward, loss / (b*c*h*w),
backward and Adam in the engine), the `<epoch, iter> l_pix` log line every `print_freq`, a validation pass every
`val_freq` (the val schedule, then back to the train schedule, sr_mfe.py:122-244) and `I{iter}_E{epoch}_{gen,opt}.pth`
every `save_checkpoint_freq`.  tensorboard / wandb writers are not reproduced (the scalars go to the log).
Under torch.distributed every rank takes its slice of each batch and the gradient arena is all-reduced: the global
batch is the config's batch_size, as with the reference's nn.DataParallel (networks.py:116-118)."""
import argparse
import logging
import os

import torch

from . import val as V
from .config import load_config
from .dataset import create_dataset
from .model import create_model
from .parallel import host_threads_per_rank, shard_range

logger = logging.getLogger('base')


def loader_workers(num_workers):
    """Loader threads of this rank: the config's `num_workers` (the reference's DataLoader workers, data/__init__.py:9-15; per
    process there, one process in all) but never more than this rank's share of the usable cores."""
    share = host_threads_per_rank(cap=16, floor=2)
    return max(2, min(int(num_workers or 0) or share, share))


def run(opt, precision='f16x3', rank=0, world=1, log=print, val_batch=1, max_val_images=None, diffusion=None, ops=None):
    """diffusion / ops: a ready model and the device side of the loaders (val.HipOps) -- injection points of the tests."""
    train_opt = opt['datasets']['train']
    train_set = create_dataset(train_opt, 'train')
    # every rank draws the same batches (same shuffle seed) and keeps its own slice of each
    gen = torch.Generator().manual_seed(int(opt.get('seed', 0) or 0))
    # the reference's DataLoader(batch_size, shuffle, num_workers, pin_memory) (data/__init__.py:9-15) as loader THREADS that
    # hand over uint8 batches; the tensor transform runs on the device (val.HipOps)
    from .dataset import ThreadedBatchLoader
    ops = ops or V.HipOps('cuda')
    # one rank: the loader threads stack every batch straight into pinned memory (ops.stage_host); several ranks: every rank decodes the
    # whole batch (same shuffle seed), keeps its slice and stages that
    loader = ThreadedBatchLoader(train_set, train_opt['batch_size'], shuffle=train_opt['use_shuffle'],
                                 workers=loader_workers(train_opt.get('num_workers')), generator=gen if world > 1 else None,
                                 stage=ops.stage_host if world == 1 else None)
    own_model = diffusion is None
    if own_model:
        diffusion = create_model(opt)                                                  # sr_mfe.py:81
    diffusion.netG.precision = precision
    if world > 1 and own_model:
        # every rank ran init_weights / default inits on its OWN torch RNG: start the replicas from rank 0's weights
        # (resumed runs load the same checkpoint everywhere; the broadcast is then a no-op in value)
        from .parallel import broadcast_module_
        broadcast_module_(diffusion.netG, src=0)
        diffusion.netG.denoise_fn.sync_weights(force=True)
    current_step, current_epoch = diffusion.begin_step, diffusion.begin_epoch
    n_iter = opt['train']['n_iter']
    if (opt.get('path') or {}).get('resume_state'):
        log('Resuming training from epoch: {}, iter: {}.'.format(current_epoch, current_step))
    diffusion.set_new_noise_schedule(opt['model']['beta_schedule']['train'], schedule_phase='train')   # sr_mfe.py:93-94
    history = []
    while current_step < n_iter:                                                       # sr_mfe.py:96-251
        current_epoch += 1
        for train_data in loader:
            current_step += 1
            if current_step > n_iter:
                break
            train_data.pop('Index', None)
            if world > 1:
                b = train_data['HR'].shape[0]
                lo, hi = shard_range(b, rank, world)
                train_data = {k: v[lo:hi] for k, v in train_data.items()}
            # an empty shard (ragged last batch) still takes part in the step's collectives: zero-sized tensors
            if world == 1:
                train_data = {k: ops.to_tensor(ops.to_device(v)) for k, v in train_data.items()}
            else:
                train_data = {k: (ops.to_tensor(ops.upload(k, v)) if v.shape[0] else
                                  torch.empty((0, v.shape[3], v.shape[1], v.shape[2]), device=ops.device)) for k, v in train_data.items()}
            diffusion.feed_data(train_data)
            diffusion.optimize_parameters()
            if current_step % opt['train']['print_freq'] == 0:
                logs = diffusion.get_current_log()
                message = '<epoch:{:3d}, iter:{:8,d}> '.format(current_epoch, current_step)
                for k, v in logs.items():
                    message += '{:s}: {:.4e} '.format(k, v)
                history.append((current_step, dict(logs)))
                if rank == 0:
                    log(message)
            if current_step % opt['train']['val_freq'] == 0:
                res = V.run(opt, batch=val_batch, precision=precision, results=(opt.get('path') or {}).get('results'),
                            max_images=max_val_images, rank=rank, world=world, save_images=rank == 0 or world > 1, log=log,
                            diffusion=diffusion, step=current_step, epoch=current_epoch, ops=ops)
                history.append((current_step, {'val_psnr': res['sr_psnr']}))
                diffusion.set_new_noise_schedule(opt['model']['beta_schedule']['train'], schedule_phase='train')   # :233-234
            if current_step % opt['train']['save_checkpoint_freq'] == 0 and rank == 0:
                log('Saving models and training states.')
                diffusion.save_network(current_epoch, current_step)
    loader.close()
    if rank == 0:
        log('End of training.')
    return diffusion, history


def main(argv=None, diffusion=None, ops=None):
    """diffusion / ops: injection points of tests/test_val_cli_gloo.py (a stand-in model and device side on a GPU-less box)."""
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('-c', '--config', required=True)
    ap.add_argument('-p', '--phase', choices=['train'], default='train')
    ap.add_argument('-gpu', '--gpu_ids', default=None)
    ap.add_argument('-debug', '-d', action='store_true')
    ap.add_argument('--precision', default='f16x3', choices=['f32', 'f16x3'])
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        from .parallel import init_process_group
        init_process_group()
    opt = load_config(a.config, phase='train', gpu_ids=a.gpu_ids, debug=a.debug)
    log = print
    if rank == 0 and (opt.get('path') or {}).get('log'):
        from .config import setup_logger
        tl = setup_logger(None, opt['path']['log'], 'train', screen=True)
        setup_logger('val', opt['path']['log'], 'val')
        log = logging.getLogger('base').info if tl is None else tl.info
    out = run(opt, precision=a.precision, rank=rank, world=world, log=log, diffusion=diffusion, ops=ops)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return out


if __name__ == '__main__':
    main()
