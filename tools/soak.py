"""Soak test on the GPU: engines created and destroyed, shapes / precisions / graph mode cycled; device memory
must return to its starting level (no leaks in the engine's own allocations: packed weights, noise-embedding
table, RNG state, graphs, event pool)."""
import gc
import sys

import torch

sys.path.insert(0, '.')
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20


def main(rounds=12):
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4), res_blocks=1)
    sd = synth_state_dict(cfg, 1)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    sc = sampling_scalars(bufs, sp)
    torch.zeros(1).cuda()
    base = None
    for r in range(rounds):
        eng = Engine(cfg)
        eng.load_state_dict(sd)
        eng.set_schedule(sc)
        for prec in ('f16x3', 'bf16', 'f32'):
            eng.set_precision(prec)
            for (B, H, W) in [(1, 32, 32), (3, 64, 40), (2, 128, 128)]:
                cond, noise = synth_inputs(B, H, W, 20)
                cond, noise = cond.cuda(), noise.cuda()
                for graph in (False, True):
                    out = eng.sample(cond, noise, graph=graph)
                    out2 = eng.sample(cond, None, graph=graph)
                    assert torch.isfinite(out).all() and torch.isfinite(out2).all()
                eng.profile_begin(); eng.sample(cond, noise); eng.profile_end()
        del eng, cond, noise, out, out2
        gc.collect()
        torch.cuda.empty_cache()
        f = free_mb()
        if r == 1:
            base = f          # after the first rounds the allocator / runtime pools have settled
        print('round %2d free %.1f MB' % (r, f), flush=True)
    assert base is not None and abs(free_mb() - base) < 64, ('device memory drifted', base, free_mb())
    print('soak ok: free memory stable within %.1f MB' % abs(free_mb() - base))


if __name__ == '__main__':
    main()
