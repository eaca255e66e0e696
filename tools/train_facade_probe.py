"""Where a façade training step's host time goes (GPU box): python tools/train_facade_probe.py
The loop of bench.py's `train_facade_b32` with host timers around its phases and around the two engine calls of optimize_step."""
import os, sys, time, shutil, tempfile
import numpy as np
import torch
sys.path.insert(0, '.')
import bench
from fastdiffsr_amd import val as V
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.dataset import ThreadedBatchLoader, create_dataset
from fastdiffsr_amd.model import create_model
from fastdiffsr_amd.synth import synth_state_dict

cfg = UNetConfig(**FASTDIFFSR_UNET)
sd = synth_state_dict(cfg, 0)
root = tempfile.mkdtemp(prefix='fdsr_probe_')
try:
    bench.synth_folder(root, 256)
    B = 32
    opt = bench.facade_opt(root, 'train', batch_size=B)
    torch.manual_seed(7); np.random.seed(7)
    model = create_model(opt)
    model.netG.denoise_fn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    ops = V.HipOps('cuda')
    loader = ThreadedBatchLoader(create_dataset(opt['datasets']['train'], 'train'), B, shuffle=True, workers=8, stage=ops.stage_host)
    model.set_new_noise_schedule(opt['model']['beta_schedule']['train'], schedule_phase='train')
    T = {'next': [], 'feed': [], 'opt': [], 'grads': [], 'adam': [], 'step': []}
    eng = model.netG._engine_for_training()
    g0, a0 = eng.train_grads_pairs, eng.adam_step
    def grads(*a, **k):
        t = time.perf_counter(); r = g0(*a, **k); T['grads'].append(time.perf_counter() - t); return r
    def adam(*a, **k):
        t = time.perf_counter(); r = a0(*a, **k); T['adam'].append(time.perf_counter() - t); return r
    eng.train_grads_pairs, eng.adam_step = grads, adam
    it = iter(loader)
    for k in range(9):
        t0 = time.perf_counter()
        try:
            data = next(it)
        except StopIteration:
            it = iter(loader); data = next(it)
        t1 = time.perf_counter()
        data.pop('Index')
        model.feed_data({key: ops.to_tensor(ops.to_device(v)) for key, v in data.items()})
        t2 = time.perf_counter()
        model.optimize_parameters()
        t3 = time.perf_counter()
        T['next'].append(t1 - t0); T['feed'].append(t2 - t1); T['opt'].append(t3 - t2); T['step'].append(t3 - t0)
    torch.cuda.synchronize()
    for k, v in T.items():
        print('%-6s ms: %s' % (k, ' '.join('%6.1f' % (1e3 * x) for x in v[3:])))
finally:
    shutil.rmtree(root, ignore_errors=True)
