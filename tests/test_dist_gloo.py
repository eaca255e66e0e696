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
