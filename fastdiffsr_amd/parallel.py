"""Multi-GPU layout of the sampling path: one process per GPU, images sharded
across ranks, weights replicated by ONE broadcast from rank 0 (RCCL over xGMI when
the backend is "nccl"; the same code runs on gloo for the CPU tests).  Sampling
itself needs no collective: no operator of the path mixes batch elements
(GroupNorm, noise level, CLAM/SLAM pools are all per sample)."""
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist

from .arch import UNetConfig, param_schema


def dist_backend():
    """'nccl' (= RCCL over xGMI) on a GPU box; 'gloo' where there is no GPU or FDSR_DIST_BACKEND says so (the CPU tests of the
    sharded drivers)."""
    import os
    forced = os.environ.get('FDSR_DIST_BACKEND')
    if forced:
        return forced
    return 'nccl' if torch.cuda.is_available() else 'gloo'


def init_process_group():
    """What `python -m torch.distributed.run ... -m fastdiffsr_amd.val|train` needs at start-up: one process per GPU."""
    import os
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    backend = dist_backend()
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
    dist.init_process_group(backend)
    return backend


def host_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (v2 `cpu.max`, v1 cfs quota)."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def local_world_size():
    """Ranks that share this host's cores: LOCAL_WORLD_SIZE (torch.distributed.run sets it), else WORLD_SIZE (one node), else 1."""
    import os
    for k in ('LOCAL_WORLD_SIZE', 'WORLD_SIZE'):
        v = os.environ.get(k)
        if v and v.isdigit() and int(v) > 0:
            return int(v)
    return 1


def host_threads_per_rank(cap=8, floor=2, cores=None, ranks=None):
    """Loader / writer threads ONE rank starts (the reference: DataLoader workers, data/__init__.py:9-18): this rank's share of
    the cores the job may use -- host_cores() / local ranks, minus the rank's own sampling thread -- within [floor, cap].  On a
    16-core lease an 8-rank run gets 2 per rank (16 loader threads in all), not 8 x 8."""
    cores = host_cores() if cores is None else int(cores)
    ranks = local_world_size() if ranks is None else max(1, int(ranks))
    return max(int(floor), min(int(cap), cores // ranks - 1))


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) slice of `total` images for `rank`."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def flatten_state_dict(sd, cfg: UNetConfig):
    keys = list(param_schema(cfg).keys())
    return np.concatenate([np.asarray(sd[k], dtype=np.float32).reshape(-1) for k in keys])


def unflatten_state_dict(flat, cfg: UNetConfig):
    out, off = OrderedDict(), 0
    for k, shp in param_schema(cfg).items():
        n = int(np.prod(shp))
        out[k] = np.asarray(flat[off:off + n]).reshape(shp)
        off += n
    assert off == len(flat)
    return out


def broadcast_state_dict(sd, cfg: UNetConfig, src=0, device='cpu'):
    """Rank `src` passes its state dict (others pass None); everyone returns the same dict.
    One packed fp32 message (95.2 MB for the FastDiffSR UNet)."""
    n = sum(int(np.prod(s)) for s in param_schema(cfg).values())
    if dist.get_rank() == src:
        buf = torch.from_numpy(flatten_state_dict(sd, cfg)).to(device)
    else:
        buf = torch.empty(n, dtype=torch.float32, device=device)
    dist.broadcast(buf, src=src)
    return unflatten_state_dict(buf.cpu().numpy(), cfg)


def gather_images(local, total, world, rank, dst=0):
    """Collect per-rank [b_r,3,H,W] results on `dst` in global image order (optional; the
    reference's val loop consumes images one at a time)."""
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    bmax = max(sizes)
    pad = torch.zeros((bmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


def allreduce_grads(engine):
    """Data-parallel training step: every rank ran fdsr_train_grads on its shard with the loss divided
    by the GLOBAL b*c*h*w; ONE all-reduce(sum) of the engine's gradient arena (91.6 MB fp32 for the FastDiffSR
    UNet; RCCL over xGMI) makes every rank's gradients the full-batch ones, then identical Adam steps keep the
    replicas in lockstep.  The reference's analogue is nn.DataParallel (networks.py:116-118)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(engine.grad_arena(), op=dist.ReduceOp.SUM)


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _coll_device():
    return 'cuda' if dist.get_backend() == 'nccl' else 'cpu'


def sum_over_ranks(value):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def broadcast_module_(module, src=0):
    """Make every rank's copy of `module` (parameters AND buffers) rank `src`'s, in place: ONE packed broadcast.
    Data-parallel training from scratch needs it -- each rank's init_weights draws from its own RNG, and the replicas
    only stay in lockstep if they start equal (the reference's nn.DataParallel re-replicates from device 0 every
    forward, networks.py:116-118)."""
    ts = [t for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
    if not ts:
        return module
    dev = _coll_device()
    flat = torch.cat([t.detach().reshape(-1).to(device=dev, dtype=torch.float32) for t in ts])
    dist.broadcast(flat, src=src)
    off = 0
    with torch.no_grad():
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t).to(device=t.device, dtype=t.dtype))
            off += n
    return module


def mean_over_ranks(value):
    t = torch.tensor([float(value)], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()
