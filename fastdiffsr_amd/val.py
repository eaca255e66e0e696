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
import itertools
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
from .parallel import host_threads_per_rank, shard_range

logger = logging.getLogger('base')


class HipOps:
    """The device side of the loop: uint8 batches in, model tensors / uint8 images / metric sums out, all on the GPU
    (csrc/fdsr_val.hip, fdsr_kernels.hip) -- there is no host implementation behind it.  run() takes the object as a parameter so
    that the sharding / reduction logic of the driver can be exercised on a GPU-less box with a stand-in the TESTS provide
    (tests/test_dist_gloo.py); the product always runs this one."""

    def __init__(self, device):
        if torch.device(device).type != 'cuda':
            raise RuntimeError('fastdiffsr_amd.val runs on the GPU only')
        import threading
        self.device = torch.device(device)
        self._cond = threading.Condition()
        self._owners = itertools.count(1)
        self._up = {}       # (owner, key, shape) -> ring of pinned staging buffers (H2D)
        self._down = {}     # (tag, shape, dtype) -> ring of pinned landing buffers (D2H)

    RING = 6            # pinned staging buffers per (owner, key, shape): more than any loader keeps staged ahead of its consumer (depth + 1 <= 4)
    STAGE_TIMEOUT = 120.0   # seconds a loader thread waits for a free staging buffer before it raises (a consumer that died with batches staged)

    def new_owner(self):
        """A token naming one loader's rings: monotonically increasing, never reused (id() of a dead loader can come back)."""
        with self._cond:
            return next(self._owners)

    def release(self, owner):
        """Drop the staging rings of a finished loader (pinned memory; a validation pass inside a training run makes a new loader
        every time).  Copies still in flight keep their buffers alive through the caching host allocator's stream bookkeeping."""
        with self._cond:
            for rkey in [k for k in self._up if k[0] == owner]:
                del self._up[rkey]
            self._cond.notify_all()

    def stage_host(self, key, arrays, owner=None):
        """Stack a batch's uint8 images (list of numpy arrays, or one stacked array) into a pinned staging buffer -- called on a
        LOADER thread, so that the sampling thread's share of the staging is one asynchronous copy.  A ring of RING buffers per
        (owner, key, shape); a buffer is OWNED by the batch staged into it until to_device() has issued its copy (batches may be
        staged out of order and stay staged for a while: a slot is never handed out again on call order alone), and rewritten only
        after that copy has completed.  `owner` names the loader: two loaders with the same batch shape never share a ring."""
        arr0 = arrays if isinstance(arrays, np.ndarray) else arrays[0]
        shape = tuple(arrays.shape) if isinstance(arrays, np.ndarray) else (len(arrays),) + tuple(arr0.shape)
        rkey = (owner, key, shape)
        with self._cond:
            ring = self._up.get(rkey)
            if ring is None:
                ring = self._up[rkey] = {'buf': [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(self.RING)],
                                         'ev': [None] * self.RING, 'held': [False] * self.RING, 'next': 0}
            while True:
                free = [(ring['next'] + i) % self.RING for i in range(self.RING) if not ring['held'][(ring['next'] + i) % self.RING]]
                if free:
                    break
                # every buffer holds a staged batch that has not been uploaded yet
                if not self._cond.wait(self.STAGE_TIMEOUT):
                    raise RuntimeError('no free staging buffer for %r after %.0f s: its %d staged batches were never uploaded '
                                       '(did the consumer stop?)' % (rkey, self.STAGE_TIMEOUT, self.RING))
            slot = free[0]
            ring['held'][slot] = True
            ring['next'] = (slot + 1) % self.RING
            ev = ring['ev'][slot]
        if ev is not None:
            ev.synchronize()
        dst = ring['buf'][slot].numpy()
        if isinstance(arrays, np.ndarray):
            dst[...] = arrays
        else:
            for j, a in enumerate(arrays):
                dst[j] = a
        return (rkey, slot)

    def to_device(self, staged):
        """The asynchronous H2D copy of a staged batch (sampling thread, current stream); releases the batch's hold on its buffer."""
        rkey, slot = staged
        ring = self._up[rkey]
        out = ring['buf'][slot].to(self.device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        with self._cond:
            ring['ev'][slot] = ev
            ring['held'][slot] = False
            self._cond.notify_all()
        return out

    def upload(self, key, arr):
        return self.to_device(self.stage_host(key, arr))

    to_tensor = staticmethod(M.u8_to_tensor)              # data/util.py:66-75 on the device
    tensor2img_batch = staticmethod(M.tensor2img_batch)   # core/metrics.py:16-42 on the device
    metric_sums = staticmethod(M.image_metric_sums)       # sr_mfe.py:313-345's per-pixel work on the device

    def lr_to_sr(self, lr_u8, h, w):
        return lr_to_sr(lr_u8, h, w)

    def new_sums(self, b):
        return torch.empty(2, b, 8, dtype=torch.float64, device=self.device)

    def land(self, tag, slot, t):
        """Asynchronous D2H of `t` into pinned buffer `slot` of its ring; valid once the event of mark() has passed."""
        key = (tag, tuple(t.shape), t.dtype)
        ring = self._down.get(key)
        if ring is None:      # (setdefault would build -- and pin -- three buffers on EVERY call: 1.6 ms each, round 6)
            ring = self._down[key] = [torch.empty(t.shape, dtype=t.dtype).pin_memory() for _ in range(3)]
        ring[slot].copy_(t, non_blocking=True)
        return ring[slot]

    def mark(self):
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def sync(self):
        torch.cuda.synchronize()


class _Loader:
    """The reference feeds its loop from a DataLoader (data/__init__.py:7-21: batch 1, one worker).  Here worker THREADS decode
    the image files of the next `depth` batches (PIL releases the GIL while it decodes) into uint8 arrays; the batch is stacked,
    crosses PCIe as bytes and becomes the model tensors on the device (ops.to_tensor, bit-identical to the dataset's own host
    transform)."""

    def __init__(self, dataset, lo, hi, batch, pool, ops, depth=2):
        self.ds, self.pool, self.depth, self.ops = dataset, pool, depth, ops
        self.owner = ops.new_owner() if hasattr(ops, 'new_owner') else id(self)
        self.batches = [list(range(b0, min(b0 + batch, hi))) for b0 in range(lo, hi, batch)]
        self.futs = {}
        self.next = 0

    def _collate(self, item_futs):
        """Loader thread: the batch's decoded images -> pinned staging buffers (its item jobs were queued before this job, so they are
        running or done: the wait cannot starve the pool)."""
        items = [f.result() for f in item_futs]
        out = {'Index': [it['Index'] for it in items]}
        for key in ('HR', 'SR', 'LR'):
            if key in items[0]:
                out[key] = self.ops.stage_host(key, [it[key] for it in items], owner=self.owner)
        return out

    def _submit_until(self, k):
        while self.next < len(self.batches) and self.next <= k:
            item_futs = [self.pool.submit(self.ds.load_u8, i) for i in self.batches[self.next]]
            self.futs[self.next] = self.pool.submit(self._collate, item_futs)
            self.next += 1

    def __len__(self):
        return len(self.batches)

    def close(self):
        """End of the pass (also an aborted one): wait for the collate jobs still queued, then give the staging rings back."""
        for f in list(self.futs.values()):
            try:
                f.result()
            except Exception:
                pass
        self.futs.clear()
        if hasattr(self.ops, 'release'):
            self.ops.release(self.owner)

    def prefetch(self, k):
        """Hand the decode + collate jobs of batches <= k to the pool.  Called by the loop right BEFORE it enters a long GPU call: the
        workers then run their Python sections while this thread sits in C without the GIL, instead of time-slicing it nine ways
        with this thread's own code (measured: 29 ms instead of 4 ms per batch on the sampling thread)."""
        self._submit_until(k)

    def get(self, k):
        """Batch k as {'HR','SR','LR': [B,H,W,3] uint8 device tensors (those the dataset has), 'Index': [...]}."""
        self._submit_until(k)
        staged = self.futs.pop(k).result()
        return {key: (v if key == 'Index' else self.ops.to_device(v)) for key, v in staged.items()}


def run(opt, batch=1, cond_from_lr=False, precision='f16x3', results=None, max_images=None, rank=0, world=1,
        save_images=True, log=print, infer=False, diffusion=None, step=None, epoch=None, workers=None, rng=None, graph=None,
        host_metrics=False, ops=None):
    """infer=True is the reference's infer.py (:62-110): the same loop, `{step}_{idx}_sr.png` outputs, timing, no metrics.
    diffusion: an existing model (the validation pass inside the training loop, sr_mfe.py:122-244); else one is created.

    The loop is a pipeline: loader threads decode the next batches while the GPU samples this one; tensor2img, MSE / PSNR /
    SSIM / ERGAS run on the device on the uint8 images (csrc/fdsr_val.hip; 2 x B x 8 doubles and the uint8 SR images come back
    through pinned memory), and a finisher thread turns the sums into the reference's per-image numbers and hands the images
    to the pool to be written -- none of which the sampling loop waits for.  host_metrics=True scores the landed uint8 images
    with metrics.compare_* instead (what the tests hold the device kernels against).
    rng: 'torch' (default; the reference's draws, reproducible under torch.manual_seed) or 'engine' (Philox inside the loop);
    graph: 'auto' | 'on' | 'off' -- replay the 20-step loop as a captured hipGraph (None: the model's default, 'auto').
    ops: the device side (HipOps; see there)."""
    import queue
    import sys
    import threading
    from concurrent.futures import ThreadPoolExecutor
    t_run0 = time.perf_counter()
    clock = {'stage': 0.0, 'post': 0.0}      # host seconds of this thread outside the sampling calls (returned in res['host_seconds'])
    val_opt = opt['datasets']['val']
    dataset = create_dataset(val_opt, 'val', cond_from_lr=cond_from_lr)
    n_total = len(dataset) if max_images is None else min(len(dataset), max_images)
    lo, hi = shard_range(n_total, rank, world)
    scale = int(val_opt['r_resolution']) // int(val_opt['l_resolution'])
    if diffusion is None:
        diffusion = create_model(opt)                                                 # sr_mfe.py:60
    diffusion.netG.precision = precision
    if rng is not None:
        diffusion.netG.rng = rng
    if graph is not None:
        diffusion.netG.graph = graph
    diffusion.set_new_noise_schedule(opt['model']['beta_schedule']['val'], schedule_phase='val')   # sr_mfe.py:66-67, :131-132
    current_step = diffusion.begin_step if step is None else step
    current_epoch = diffusion.begin_epoch if epoch is None else epoch
    result_path = results or (opt.get('path') or {}).get('results') or 'results'
    if save_images:
        os.makedirs(result_path, exist_ok=True)
    if ops is None:
        ops = HipOps(diffusion.device)
    rres = int(val_opt['r_resolution'])
    per_image = {}                     # index -> 8 numbers (bic mse/psnr/ssim/ergas, sr mse/psnr/ssim/ergas); summed in index order
    t_sample = 0.0
    # loader / writer threads of THIS rank: its share of the cores the job may use (affinity, cgroup quota, ranks on the host)
    n_workers = workers if workers else host_threads_per_rank(cap=8, floor=2)
    pool = ThreadPoolExecutor(max_workers=n_workers)
    jobs = queue.Queue()
    errors = []
    saves = []

    def _save(img, path):
        from PIL import Image
        Image.fromarray(img).save(path)

    def _finish():
        while True:
            job = jobs.get()
            if job is None:
                return
            try:
                ev, idxs, sr_host, sums_host, hr_host, inf_host, _ = job
                ev.synchronize()
                sr_np = sr_host.numpy()
                for j, index in enumerate(idxs):
                    idx = index + 1                                                   # sr_mfe.py:274 counts from 1
                    if save_images:
                        path = '{}/{}_{}_sr.{}'.format(result_path, current_step, idx, 'png' if infer else 'tif')
                        saves.append(pool.submit(_save, sr_np[j].copy(), path))
                    if infer:
                        per_image[index] = None
                    elif host_metrics:
                        h_, f_, s_ = hr_host.numpy()[j], inf_host.numpy()[j], sr_np[j]
                        per_image[index] = [M.compare_mse(f_, h_), M.compare_psnr(f_, h_), M.compare_ssim(f_, h_),
                                            M.calculate_ergas(f_, h_, scale=scale),
                                            M.compare_mse(s_, h_), M.compare_psnr(s_, h_), M.compare_ssim(s_, h_),
                                            M.calculate_ergas(s_, h_, scale=scale)]
                    else:
                        sm = sums_host.numpy()
                        b_ = M.metrics_from_sums(sm[0, j], sr_np[j].shape, scale)
                        s_ = M.metrics_from_sums(sm[1, j], sr_np[j].shape, scale)
                        per_image[index] = [b_['mse'], b_['psnr'], b_['ssim'], b_['ergas'], s_['mse'], s_['psnr'], s_['ssim'], s_['ergas']]
            except Exception as e:       # surfaced by run() after the loop
                errors.append(e)
            finally:
                job[-1].set()            # this slot's landing buffers may be refilled

    finisher = threading.Thread(target=_finish, daemon=True)
    finisher.start()
    # the sampling thread hands the GIL over around every launch / synchronisation; with Python's default 5 ms switch interval each
    # hand-back can cost it up to 5 ms while a loader / writer thread is in a Python section: ask for 0.2 ms while the loop runs
    old_switch = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)
    clock['setup'] = time.perf_counter() - t_run0
    loader = None
    slot_free = [None] * 3
    try:
        loader = _Loader(dataset, lo, hi, batch, pool, ops)

        def stage(k):
            """Batch k on the device as the model tensors: issued BEFORE the previous batch is sampled, so the bytes are
            there when its loop ends (the copies and the two small kernels sit in front of that loop on the stream)."""
            if k >= len(loader):
                return None
            ts = time.perf_counter()
            raw = loader.get(k)
            idxs = raw.pop('Index')
            data = {key: ops.to_tensor(v) for key, v in raw.items()}
            if cond_from_lr:
                data['SR'] = ops.lr_to_sr(raw['LR'], rres, rres)
            clock['stage'] += time.perf_counter() - ts
            return idxs, data

        loader.prefetch(1)
        staged = stage(0)
        for k in range(len(loader)):
            idxs, data = staged
            diffusion.feed_data(data)
            staged = stage(k + 1)
            loader.prefetch(k + 1 + loader.depth)
            ops.sync()
            t0 = time.time()
            diffusion.test(continous=False)
            ops.sync()
            t_sample += time.time() - t0
            logger.info('inference time (s): {:.4f} for {} image(s)'.format(time.time() - t0, len(idxs)))   # sr_mfe.py:279-284
            tp = time.perf_counter()
            sr_batch = diffusion.SR
            if sr_batch.dim() == 3:       # the ddpm / tesr siblings return ret_img[-1]: one image (their own convention)
                if len(idxs) != 1:
                    raise ValueError("which_model_G in ('ddpm', 'tesr') returns one image per call: use --batch 1")
                sr_batch = sr_batch[None]
            b = len(idxs)
            sr_u8 = ops.tensor2img_batch(sr_batch)                                    # Metrics.tensor2img, on the device
            slot = k % 3
            if slot_free[slot] is not None:
                slot_free[slot].wait()           # the finisher is done with this slot's previous contents
            sr_host = ops.land('sr', slot, sr_u8)
            sums_host = hr_host = inf_host = None
            if not infer:
                hr_u8 = ops.tensor2img_batch(diffusion.data['HR'])
                inf_u8 = ops.tensor2img_batch(diffusion.data['SR'])                   # the bicubic image ('INF')
                if host_metrics:
                    hr_host, inf_host = ops.land('hr', slot, hr_u8), ops.land('inf', slot, inf_u8)
                else:
                    sums = ops.new_sums(b)
                    ops.metric_sums(inf_u8, hr_u8, out=sums[0])
                    ops.metric_sums(sr_u8, hr_u8, out=sums[1])
                    sums_host = ops.land('sums', slot, sums)
            done = threading.Event()
            slot_free[slot] = done
            jobs.put((ops.mark(), idxs, sr_host, sums_host, hr_host, inf_host, done))
            clock['post'] += time.perf_counter() - tp
    finally:
        tt = time.perf_counter()
        jobs.put(None)
        finisher.join()
        if loader is not None:
            loader.close()
        for f in saves:
            f.result()
        pool.shutdown(wait=True)
        sys.setswitchinterval(old_switch)
        clock['tail'] = time.perf_counter() - tt
    if errors:
        raise errors[0]
    sums = np.zeros(9, dtype=np.float64)       # bic mse/psnr/ssim/ergas, sr mse/psnr/ssim/ergas, count
    for index in sorted(per_image):
        if per_image[index] is not None:
            sums[:8] += np.array(per_image[index])
        sums[8] += 1.0
    if world > 1:
        import torch.distributed as dist
        t = torch.from_numpy(sums).cuda() if dist.get_backend() == 'nccl' else torch.from_numpy(sums)
        dist.all_reduce(t)
        sums = t.cpu().numpy()
    n = max(sums[8], 1.0)
    avg = sums[:8] / n
    res = dict(images=int(sums[8]), bic_mse=avg[0], bic_psnr=avg[1], bic_ssim=avg[2], bic_ergas=avg[3],
               sr_mse=avg[4], sr_psnr=avg[5], sr_ssim=avg[6], sr_ergas=avg[7],
               sample_seconds_this_rank=t_sample, result_path=result_path,
               host_seconds=dict(clock, total=time.perf_counter() - t_run0, workers=n_workers))
    if rank == 0 and infer:
        log('inference: {} images, {:.4f} s per image on this rank (batch {})'.format(int(sums[8]), t_sample / max(hi - lo, 1), batch))
    elif rank == 0:
        log('<epoch:{:3d}, iter:{:8,d}> bic_mse: {:.5e}, bic_psnr: {:.5e}, bic_ssim: {:.5e}, bic_ergas: {:.5e}'.format(
            current_epoch, current_step, *avg[:4]))
        log('<epoch:{:3d}, iter:{:8,d}> sr_mse: {:.5e}, sr_psnr: {:.5e}, sr_ssim: {:.5e}, sr_ergas: {:.5e}'.format(
            current_epoch, current_step, *avg[4:8]))
    return res


def main(argv=None, diffusion=None, ops=None):
    """diffusion / ops: injection points of tests/test_dist_gloo.py (a stand-in model and device side on a GPU-less box)."""
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('-c', '--config', required=True, help='JSON file for configuration (the reference\'s own)')
    ap.add_argument('-p', '--phase', choices=['val'], default='val')
    ap.add_argument('-gpu', '--gpu_ids', default=None)
    ap.add_argument('-debug', '-d', action='store_true')
    ap.add_argument('--batch', type=int, default=1)
    ap.add_argument('--cond-from-lr', action='store_true')
    ap.add_argument('--precision', default='f16x3', choices=['f32', 'f16x3', 'bf16', 'f16'])
    ap.add_argument('--results', default=None)
    ap.add_argument('--max-images', type=int, default=None)
    ap.add_argument('--no-save', action='store_true')
    ap.add_argument('--infer', action='store_true', help="the reference's infer.py: png outputs and timing, no metrics")
    ap.add_argument('--workers', type=int, default=None, help='loader / writer threads of this rank (default: its share of the usable cores, 2 .. 8)')
    ap.add_argument('--rng', default=None, choices=['torch', 'engine'],
                    help="sampling noise: 'torch' (default; torch.randn in the reference's order, reproducible under torch.manual_seed) "
                         "or 'engine' (Philox inside the HIP loop: nothing pre-drawn)")
    ap.add_argument('--graph', default=None, choices=['auto', 'on', 'off'],
                    help='replay the T-step loop as a captured hipGraph (default auto: from the second call of a shape on)')
    ap.add_argument('--host-metrics', action='store_true', help='score on the host (numpy) instead of the device kernels')
    a = ap.parse_args(argv)
    rank, world = int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))
    if world > 1:
        from .parallel import init_process_group
        init_process_group()
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
              max_images=a.max_images, rank=rank, world=world, save_images=not a.no_save, infer=a.infer, log=log,
              workers=a.workers, rng=a.rng, graph=a.graph, host_metrics=a.host_metrics, diffusion=diffusion, ops=ops)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    return res


if __name__ == '__main__':
    main()
