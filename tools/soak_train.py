"""Training soak on the GPU: engines created, stepped (both precisions, dropout live, changing batch shapes) and destroyed;
device memory must return to its starting level (master copy, gradient arena, Adam state, packed forms, copy table)."""
import gc
import sys

import torch

sys.path.insert(0, '.')
from fastdiffsr_amd.arch import UNetConfig
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.synth import synth_state_dict


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2 ** 20


def main(rounds=10):
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4), res_blocks=1, dropout=0.2)
    sd = synth_state_dict(cfg, 1)
    torch.zeros(1).cuda()
    base = None
    g = torch.Generator().manual_seed(0)
    for r in range(rounds):
        eng = Engine(cfg)
        eng.load_state_dict(sd)
        eng.set_training(True)
        for prec in ('f16x3', 'f32', 'f16x3'):
            eng.set_precision(prec)
            for (B, H, W) in [(2, 32, 32), (3, 64, 40), (1, 128, 128)]:
                x = torch.randn(B, 6, H, W, generator=g).cuda()
                nl = (torch.rand(B, generator=g) * 0.5 + 0.4).cuda()
                t = torch.randn(B, 3, H, W, generator=g).cuda()
                for _ in range(2):
                    loss = eng.train_grads(x, nl, t, 'l1', 1.0 / t.numel())
                    eng.adam_step(1e-4)
                    assert loss == loss
        eng.set_training(False)
        del eng
        gc.collect()
        torch.cuda.empty_cache()
        f = free_mb()
        if r == 1:
            base = f
        print(f'round {r:2d} free {f:.1f} MB', flush=True)
    assert base is not None and abs(free_mb() - base) <= 16.0, (free_mb(), base)
    print('training soak ok: free memory stable')


if __name__ == '__main__':
    main()
