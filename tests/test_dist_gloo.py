"""The N>1 path on CPU: world_size-2 gloo.  Rank 0 owns the weights, ONE broadcast replicates
them, images are sharded with no data-path collective, results gathered in order.  The oracle
stands in for the per-rank sampler (no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fastdiffsr_amd import parallel
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs, state_dict_sha256

CFG = dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2), attn_res=(16,),
           res_blocks=1, dropout=0.0, image_size=16)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    from oracle import fdsr_oracle as O
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 3) if rank == 0 else None
    sd = parallel.broadcast_state_dict(sd, cfg, src=0)
    sha = state_dict_sha256(sd)
    cond, noise = synth_inputs(total, 16, 16, 20)
    lo, hi = parallel.shard_range(total, rank, world)
    tab = O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL)
    local = O.p_sample_loop(O.to_torch_sd(sd), cfg, tab, cond[lo:hi], noise[:, lo:hi])
    full = parallel.gather_images(local, total, world, rank, dst=0)
    if rank == 0:
        ref = O.p_sample_loop(O.to_torch_sd(sd), cfg, tab, cond, noise)
        q.put((sha, float((full - ref).abs().max()), tuple(full.shape)))
    else:
        q.put((sha, None, None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_broadcast_shard_gather():
    world, total = 2, 3          # ragged: rank 0 gets 2 images, rank 1 gets 1
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    shas = {r[0] for r in res}
    assert shas == {state_dict_sha256(synth_state_dict(UNetConfig(**CFG), 3))}     # both ranks hold rank 0's weights
    d = [r for r in res if r[1] is not None][0]
    assert d[2] == (3, 3, 16, 16)
    assert d[1] <= 1e-5      # sharded == unsharded (B independent runs)


# ---------------------------------------------------------------------------------------------------------------
# data-parallel training step (SURVEY 8e, configs[4]): every rank back-propagates its shard with the loss divided by
# the GLOBAL element count; ONE all-reduce(sum) of the flat gradient arena gives every rank the full-batch gradients
# ---------------------------------------------------------------------------------------------------------------
class _ArenaStandIn:
    """What parallel.allreduce_grads needs from an engine: the flat gradient arena as one tensor."""

    def __init__(self, flat):
        self.flat = flat

    def grad_arena(self):
        return self.flat


def _train_worker(rank, world, port, q):
    from oracle import fdsr_oracle as O
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 3) if rank == 0 else None
    sd = parallel.broadcast_state_dict(sd, cfg, src=0)
    g = torch.Generator().manual_seed(5)
    B = 4
    hr = torch.rand(B, 3, 16, 16, generator=g) * 2 - 1
    sr = (hr + 0.2 * torch.randn(B, 3, 16, 16, generator=g)).clamp(-1, 1)
    noise = torch.randn(B, 3, 16, 16, generator=g)
    gamma = torch.rand(B, generator=g) * 0.5 + 0.4
    lo, hi = parallel.shard_range(B, rank, world)

    def grads_of(sl):
        leaves = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd.items()}
        loss = O.p_losses(leaves, cfg, hr[sl], sr[sl], gamma[sl], noise[sl], 'l1') / hr.numel()      # global b*c*h*w
        loss.backward()
        keys = [k for k in leaves if leaves[k].grad is not None]
        return keys, torch.cat([leaves[k].grad.reshape(-1) for k in keys]), loss.item()

    keys, flat, l_local = grads_of(slice(lo, hi))
    parallel.allreduce_grads(_ArenaStandIn(flat))
    if rank == 0:
        _, full, l_full = grads_of(slice(0, B))
        q.put((float((flat - full).abs().max() / full.abs().max()), l_local, l_full, parallel.world_size()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_equals_full_batch():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    rel, l_local, l_full, ws = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ws == 2
    assert rel <= 1e-5, rel          # summed shard gradients == full-batch gradients (fp32 summation order only)
    assert l_local < l_full          # each rank holds its share of the globally normalised loss


# ---------------------------------------------------------------------------------------------------------------
# data-parallel TRAINING DRIVER logic (fastdiffsr_amd.train / DDPM.optimize_parameters / GaussianDiffusion.optimize_step):
# replicas start from rank 0's weights, the divisor is the GLOBAL sample count, ragged and EMPTY shards are fine
# ---------------------------------------------------------------------------------------------------------------
class _OracleEngine:
    """Stand-in for the HIP engine on CPU: the oracle's autograd fills a flat gradient arena, torch.optim.Adam steps the
    leaves.  Only what optimize_step touches."""

    def __init__(self, cfg, sd):
        from oracle import fdsr_oracle as O
        self.O, self.cfg = O, cfg
        self.leaves = {k: torch.from_numpy(np.array(v)).clone().requires_grad_(True) for k, v in sd.items()}
        self.keys = None
        self.flat = None
        self.opt = None

    def _probe_keys(self, x6, gamma, noise):
        for p in self.leaves.values():
            p.grad = None
        cond, x_noisy = x6[:, :3], x6[:, 3:]
        eps = self.O.unet_forward(self.leaves, self.cfg, torch.cat([cond, x_noisy], 1), gamma.view(-1, 1))
        return (noise - eps).abs().sum()

    def train_grads(self, x6, gamma, noise, loss_type, scale):
        loss = self._probe_keys(x6, gamma, noise)
        (loss * scale).backward()
        if self.keys is None:
            self.keys = [k for k, p in self.leaves.items() if p.grad is not None]
        self.flat = torch.cat([self.leaves[k].grad.reshape(-1) for k in self.keys])
        return float(loss)

    def train_grads_pairs(self, hr, sr, gamma, noise, loss_type, scale):
        """What fdsr_train_grads_pairs does on the device: img2res + q_sample + cat (diffusion.py:233-263), then the step."""
        x_start = ((hr - sr) * 2.0).clamp(-1, 1)
        g = gamma.view(-1, 1, 1, 1)
        x6 = torch.cat([sr, g * x_start + (1 - g ** 2).sqrt() * noise], 1)
        return self.train_grads(x6, gamma, noise, loss_type, scale)

    def zero_grads(self, device=None):
        if self.keys is None:       # the key set does not depend on the data: derive it from a dry run
            self.keys = _OracleEngine._live_keys
        self.flat = torch.zeros(sum(self.leaves[k].numel() for k in self.keys))

    def grad_arena(self):
        return self.flat

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        if self.opt is None:
            self.opt = torch.optim.Adam([self.leaves[k] for k in self.keys], lr=lr, betas=betas, eps=eps)
        off = 0
        for k in self.keys:
            n = self.leaves[k].numel()
            self.leaves[k].grad = self.flat[off:off + n].view_as(self.leaves[k]).clone()
            off += n
        self.opt.step()


def _driver_worker(rank, world, port, batches, q):
    from fastdiffsr_amd.diffusion import GaussianDiffusion
    from fastdiffsr_amd.model import DDPM
    from fastdiffsr_amd.unet import UNet
    from fastdiffsr_amd.schedule import schedule_buffers
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = UNetConfig(**CFG)
    torch.manual_seed(100 + rank)                       # every rank initialises on its OWN RNG, as define_G does
    unet = UNet(**CFG)
    before = torch.cat([p.detach().reshape(-1) for p in unet.parameters()]).clone()
    netG = GaussianDiffusion(unet, image_size=16)
    parallel.broadcast_module_(netG, src=0)             # what fastdiffsr_amd.train.run does after create_model
    start = torch.cat([p.detach().reshape(-1) for p in unet.parameters()]).clone()
    sd = {k: v.detach().numpy().copy() for k, v in super(UNet, unet).state_dict().items()}
    eng = _OracleEngine(cfg, sd)
    # live keys (for a rank that starts with an empty shard): everything the forward reads, from a dry run on a dummy sample
    probe = _OracleEngine(cfg, sd)
    g0 = torch.Generator().manual_seed(1)
    probe.train_grads(torch.rand(1, 6, 16, 16, generator=g0), torch.tensor([0.5]), torch.rand(1, 3, 16, 16, generator=g0), 'l1', 1.0)
    _OracleEngine._live_keys = probe.keys
    bufs, sqrt_prev = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    netG.sqrt_alphas_cumprod_prev, netG.num_timesteps = sqrt_prev, 20
    netG._engine_for_training = lambda: eng
    ddpm = object.__new__(DDPM)
    ddpm.netG, ddpm.lr, ddpm.betas, ddpm.adam_eps, ddpm.log_dict = netG, 1e-3, (0.9, 0.999), 1e-8, {}
    logs = []
    for step, B in enumerate(batches):
        g = torch.Generator().manual_seed(50 + step)    # every rank draws the same global batch and keeps its slice
        hr = torch.rand(B, 3, 16, 16, generator=g) * 2 - 1
        sr = (hr + 0.2 * torch.randn(B, 3, 16, 16, generator=g)).clamp(-1, 1)
        lo, hi = parallel.shard_range(B, rank, world)
        np.random.seed(7 + step)                        # t and gamma come from numpy's global RNG (diffusion.py:246-256)
        torch.manual_seed(9 + step)
        full_noise = torch.randn(B, 3, 16, 16)
        t = np.random.randint(1, 21)
        gam_full = np.random.uniform(sqrt_prev[t - 1], sqrt_prev[t], size=B)

        # optimize_step draws t and gamma with numpy and the noise with torch.randn_like (the reference's RNG, diffusion.py:246-259):
        # here every rank must see ITS SLICE of the global draws, so the three calls are patched for the step
        from unittest import mock
        patches = [mock.patch.object(np.random, 'randint', lambda a, b, t=t: t),
                   mock.patch.object(np.random, 'uniform', lambda lo_, hi_, size=None, g=gam_full[lo:hi]: g.copy()),
                   mock.patch.object(torch, 'randn_like', lambda x, nz=full_noise[lo:hi]: nz.clone())]
        for pt in patches:
            pt.start()
        ddpm.data = {'HR': hr[lo:hi], 'SR': sr[lo:hi]}
        try:
            ddpm.optimize_parameters()
        finally:
            for pt in patches:
                pt.stop()
        logs.append(ddpm.log_dict['l_pix'])
        if step == 0:
            first = dict(B=B, hr=hr, sr=sr, gam=gam_full, noise=full_noise, arena=eng.flat.clone())
    end = torch.cat([eng.leaves[k].detach().reshape(-1) for k in eng.keys])
    # reference for step 0: the full batch on one process, loss / (B*c*h*w)
    ref = _OracleEngine(cfg, sd)
    x_start = netG.img2res(first['hr'], first['sr'])
    gamma = torch.FloatTensor(first['gam']).view(-1, 1)
    x6 = torch.cat([first['sr'], netG.q_sample(x_start, gamma.view(-1, 1, 1, 1), first['noise'])], 1)
    l_full = ref.train_grads(x6, gamma, first['noise'], 'l1', 1.0 / (first['B'] * 3 * 16 * 16))
    rel = float((first['arena'] - ref.flat).abs().max() / ref.flat.abs().max())
    q.put((rank, float((before - start).abs().max()), start.numpy().tobytes(), end.numpy().tobytes(), rel, logs[0],
           l_full / (first['B'] * 3 * 16 * 16)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('batches', [(1, 3, 4)])      # batch < world (rank 1 empty), ragged, even
def test_two_rank_training_driver_lockstep_ragged_and_empty_shards(batches):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_driver_worker, args=(r, world, port, batches, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=500) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, moved0, start0, end0, rel0, l0, lf0), (r1, moved1, start1, end1, rel1, l1, lf1) = res
    assert moved0 == 0.0 and moved1 > 1e-3          # rank 1's own init was replaced by rank 0's
    assert start0 == start1                         # replicas start equal ...
    assert end0 == end1                             # ... and are bitwise equal after three optimiser steps
    assert rel0 <= 1e-5 and rel1 <= 1e-5            # all-reduced arena == full-batch gradient with the GLOBAL divisor (B=1 < world)
    assert abs(l0 - lf0) <= 1e-6 * abs(lf0) and l0 == l1   # logged l_pix = global loss sum / global element count
