"""`python -m fastdiffsr_amd.val` under two ranks on a GPU-less box (gloo): the images are sharded by parallel.shard_range,
every rank writes its own results, the metric sums are all-reduced -- union of the shards == all images, reduced averages ==
the single-rank run's.  The model is an oracle-backed stand-in and the device side of the loop (val.HipOps: HIP kernels, GPU
only) is replaced by a numpy stand-in defined HERE; the driver code under test (val.main / val.run, the loader threads, the
finisher, the sharding and the reduction) is the product's."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from fastdiffsr_amd import metrics as M
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.synth import synth_state_dict

CFG = dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2), attn_res=(16,),
           res_blocks=1, dropout=0.0, image_size=32)


class _Done:
    def synchronize(self):
        pass


class HostOps:
    """numpy stand-in of val.HipOps (test infrastructure): same methods, host arithmetic of fastdiffsr_amd.metrics."""
    device = torch.device('cpu')

    def stage_host(self, key, arrays, owner=None):
        return torch.from_numpy(np.stack(arrays) if not isinstance(arrays, np.ndarray) else np.ascontiguousarray(arrays))

    def to_device(self, staged):
        return staged

    def upload(self, key, arr):
        return self.to_device(self.stage_host(key, arr))

    def to_tensor(self, u8):
        return u8.permute(0, 3, 1, 2).to(torch.float32).div(255) * 2 + (-1)

    def tensor2img_batch(self, t4):
        return torch.from_numpy(np.stack([M.tensor2img(t.clone()) for t in t4]))

    def metric_sums(self, test, truth, out=None):
        a, b = test.numpy(), truth.numpy()
        for j in range(a.shape[0]):
            h, w, c = a[j].shape
            n7 = (h - 6) * (w - 6) * c
            out[j] = torch.tensor([float(((a[j].astype(np.int64) - b[j]) ** 2).sum()), float(a[j].astype(np.int64).sum()),
                                   M.compare_ssim(a[j], b[j]) * n7, n7, 0, 0, 0, 0], dtype=torch.float64)
        return out

    def new_sums(self, b):
        return torch.empty(2, b, 8, dtype=torch.float64)

    def land(self, tag, slot, t):
        return t.clone()

    def mark(self):
        return _Done()

    def sync(self):
        pass


class _NetG:
    precision, rng, graph = 'f16x3', 'torch', 'auto'


class OracleDDPM:
    """What val.run asks of the DDPM wrapper (model/model.py), on the CPU restatement: every image is its own chain whose noise
    is keyed by the image content, so a sharded run and a single-rank run sample the same thing."""

    def __init__(self):
        from oracle import fdsr_oracle as O
        self.O = O
        self.cfg = UNetConfig(**CFG)
        self.sd = O.to_torch_sd(synth_state_dict(self.cfg, 5))
        self.tab = None
        self.netG = _NetG()
        self.device = torch.device('cpu')
        self.begin_step = self.begin_epoch = 0

    def set_new_noise_schedule(self, schedule_opt, schedule_phase='train'):
        self.tab = self.O.schedule_tables(schedule_opt)

    def feed_data(self, data):
        self.data = data

    def test(self, continous=False):
        cond = self.data['SR']
        outs = []
        for j in range(cond.shape[0]):
            g = torch.Generator().manual_seed(int(cond[j].abs().sum().item() * 1000) % (2 ** 31))
            noise = torch.randn(20, 1, 3, cond.shape[2], cond.shape[3], generator=g)
            outs.append(self.O.p_sample_loop(self.sd, self.cfg, self.tab, cond[j:j + 1], noise))
        self.SR = torch.cat(outs)


def _config(root, l, r):
    return {"name": "sr_fastdiffsr_gloo", "phase": "val", "gpu_ids": [0],
            "path": {"log": "logs", "tb_logger": "tb_logger", "results": "results", "checkpoint": "checkpoint", "resume_state": None},
            "datasets": {"train": {"name": "t", "mode": "HR", "dataroot": root, "datatype": "img", "l_resolution": l, "r_resolution": r,
                                   "batch_size": 2, "num_workers": 1, "use_shuffle": True, "data_len": -1},
                         "val": {"name": "v", "mode": "LRHR", "dataroot": root, "datatype": "img", "l_resolution": l, "r_resolution": r,
                                 "data_len": -1}},
            "model": {"which_model_G": "fastdiffsr", "finetune_norm": False,
                      "unet": {"in_channel": 6, "out_channel": 3, "inner_channel": 32, "channel_multiplier": [1, 2], "attn_res": [16],
                               "res_blocks": 1, "dropout": 0.0},
                      "beta_schedule": {"train": dict(FASTDIFFSR_SCHEDULE_VAL), "val": dict(FASTDIFFSR_SCHEDULE_VAL)},
                      "diffusion": {"image_size": r, "channels": 3, "conditional": True}},
            "train": {"n_iter": 1, "val_freq": 1, "save_checkpoint_freq": 1, "print_freq": 1, "optimizer": {"type": "adam", "lr": 1e-4}},
            "wandb": {"project": "x"}}


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, cpath, out_dir, cwd, q):
    from fastdiffsr_amd import val
    os.chdir(cwd)
    torch.set_num_threads(2)
    os.environ.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'MASTER_ADDR': '127.0.0.1',
                       'MASTER_PORT': str(port)})
    os.environ.pop('FDSR_DIST_BACKEND', None)        # no GPU here: parallel.dist_backend() picks gloo by itself
    res = val.main(['-c', cpath, '--batch', '2', '--results', out_dir, '--workers', '2'], diffusion=OracleDDPM(), ops=HostOps())
    q.put((rank, {k: v for k, v in res.items() if k not in ('result_path', 'host_seconds')}))


@pytest.mark.timeout(600)
def test_val_cli_two_ranks_gloo(tmp_path):
    from test_val_host import make_dataset
    from fastdiffsr_amd import val
    from fastdiffsr_amd.config import load_config
    root = make_dataset(str(tmp_path / 'data'), n=5, l=8, r=32, seed=9)
    cpath = str(tmp_path / 'cfg.json')
    with open(cpath, 'w') as f:
        json.dump(_config(root, 8, 32), f)
    # single rank, in this process
    lines = []
    one = val.run(load_config(cpath, phase='val'), batch=2, results=str(tmp_path / 'one'), log=lines.append, diffusion=OracleDDPM(),
                  ops=HostOps(), workers=2)
    assert one['images'] == 5 and len(lines) == 2 and sorted(os.listdir(tmp_path / 'one')) == ['0_%d_sr.tif' % i for i in range(1, 6)]
    # the loop's own numbers against the metrics recomputed from the files it wrote
    from PIL import Image
    hr = [np.asarray(Image.open(os.path.join(root, 'hr_32', '%05d.png' % (i + 1)))) for i in range(5)]
    sr = [np.asarray(Image.open(tmp_path / 'one' / ('0_%d_sr.tif' % (i + 1)))) for i in range(5)]
    assert abs(one['sr_psnr'] - np.mean([M.compare_psnr(s, h) for s, h in zip(sr, hr)])) < 1e-9
    assert abs(one['sr_ssim'] - np.mean([M.compare_ssim(s, h) for s, h in zip(sr, hr)])) < 1e-9
    # two ranks through the CLI entry point
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, cpath, str(tmp_path / 'two'), str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=500) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert sorted(os.listdir(tmp_path / 'two')) == ['0_%d_sr.tif' % i for i in range(1, 6)]      # shard union == all images
    for i in range(1, 6):       # and every image is the one the single-rank run produced
        assert np.array_equal(np.asarray(Image.open(tmp_path / 'two' / ('0_%d_sr.tif' % i))), sr[i - 1])
    for r in (0, 1):            # the all-reduced averages are the single-rank averages on EVERY rank
        assert got[r]['images'] == 5
        for k in ('bic_mse', 'bic_psnr', 'bic_ssim', 'bic_ergas', 'sr_mse', 'sr_psnr', 'sr_ssim', 'sr_ergas'):
            assert abs(got[r][k] - one[k]) <= 1e-12 * max(1.0, abs(one[k])), (k, got[r][k], one[k])


# ---------------------------------------------------------------------------------------------------------------------------
# `python -m fastdiffsr_amd.train` under two ranks (gloo): same shuffle on every rank, each keeps its slice of every batch, the
# gradient arena is all-reduced with the GLOBAL divisor -- replicas end bitwise equal and log the same l_pix.
# ---------------------------------------------------------------------------------------------------------------------------
def _train_rank(rank, world, port, cpath, cwd, q):
    from test_dist_gloo import _OracleEngine
    from fastdiffsr_amd import train
    from fastdiffsr_amd.diffusion import GaussianDiffusion
    from fastdiffsr_amd.model import DDPM
    from fastdiffsr_amd.unet import UNet
    from fastdiffsr_amd.schedule import schedule_buffers
    os.chdir(cwd)
    torch.set_num_threads(2)
    os.environ.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'MASTER_ADDR': '127.0.0.1',
                       'MASTER_PORT': str(port)})
    cfg = UNetConfig(**CFG)
    torch.manual_seed(0)                                # (resumed-from-the-same-checkpoint situation: equal replicas at the start)
    unet = UNet(**CFG)
    netG = GaussianDiffusion(unet, image_size=32)
    sd = {k: v.detach().numpy().copy() for k, v in super(UNet, unet).state_dict().items()}
    eng = _OracleEngine(cfg, sd)
    probe = _OracleEngine(cfg, sd)
    g0 = torch.Generator().manual_seed(1)
    probe.train_grads(torch.rand(1, 6, 32, 32, generator=g0), torch.tensor([0.5]), torch.rand(1, 3, 32, 32, generator=g0), 'l1', 1.0)
    _OracleEngine._live_keys = probe.keys
    bufs, sqrt_prev = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    netG.sqrt_alphas_cumprod_prev, netG.num_timesteps = sqrt_prev, 20
    netG._engine_for_training = lambda: eng
    ddpm = object.__new__(DDPM)
    ddpm.netG, ddpm.lr, ddpm.betas, ddpm.adam_eps, ddpm.log_dict = netG, 1e-3, (0.9, 0.999), 1e-8, {}
    ddpm.device, ddpm.begin_step, ddpm.begin_epoch, ddpm.schedule_phase = torch.device('cpu'), 0, 0, 'train'
    ddpm.set_new_noise_schedule = lambda *a, **k: None          # (the stand-in engine has no device-side schedule)
    np.random.seed(11)                                  # t and gamma come from numpy's global RNG, the noise from torch's: same on both ranks
    torch.manual_seed(12)
    _, hist = train.main(['-c', cpath], diffusion=ddpm, ops=HostOps())
    end = torch.cat([eng.leaves[k].detach().reshape(-1) for k in eng.keys])
    logs = []
    for d, _, files in os.walk(os.path.join(cwd, 'experiments')):       # rank 0 writes experiments/<name>_<stamp>/logs/train.log
        if 'train.log' in files:
            logs += [ln for ln in open(os.path.join(d, 'train.log')) if '<epoch' in ln]
    q.put((rank, [v['l_pix'] for s, v in hist if 'l_pix' in v], end.numpy().tobytes(), logs if rank == 0 else []))


@pytest.mark.timeout(600)
def test_train_cli_two_ranks_gloo(tmp_path):
    from test_val_host import make_dataset
    root = make_dataset(str(tmp_path / 'data'), n=5, l=8, r=32, seed=9)
    cfg = _config(root, 8, 32)
    cfg['phase'] = 'train'
    cfg['datasets']['train'].update(batch_size=3, use_shuffle=True, num_workers=2)      # 5 images: batches of 3 and 2 -> ragged shards
    cfg['train'].update(n_iter=3, val_freq=100, save_checkpoint_freq=100, print_freq=1)
    cpath = str(tmp_path / 'train.json')
    with open(cpath, 'w') as f:
        json.dump(cfg, f)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_rank, args=(r, 2, port, cpath, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, l0, end0, lines0), (r1, l1, end1, lines1) = res
    assert len(l0) == 3 and l0 == l1 and all(np.isfinite(l0))       # the logged l_pix is the GLOBAL one on every rank
    assert end0 == end1                                             # replicas bitwise equal after three data-parallel steps
    assert len(lines0) == 3 and not lines1                          # rank 0 alone writes the iteration lines (logs/train.log)
