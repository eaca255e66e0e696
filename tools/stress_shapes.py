"""Shape sweep on the GPU: workspace size, finiteness, graph == eager, and (small cases) batch invariance.
Run on an MI355X: python tools/stress_shapes.py"""
import sys
import time

import torch

sys.path.insert(0, '.')
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs


def main():
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    eng = Engine(cfg)
    eng.load_state_dict(synth_state_dict(cfg, 0))
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    dev = torch.device('cuda')
    for prec in ('f16x3', 'bf16', 'f32'):
        eng.set_precision(prec)
        for (B, H, W) in [(1, 1024, 1024), (128, 256, 256), (2, 264, 200), (3, 8, 8), (1, 8, 512), (5, 72, 40), (1, 2048, 2048)]:
            if prec == 'f32' and B * H * W > 16 * 256 * 256:
                continue
            if prec != 'f16x3' and H * W >= 2048 * 2048:
                continue
            cond, noise = synth_inputs(B, H, W, 20)
            cond, noise = cond.to(dev), noise.to(dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a = eng.sample(cond, noise)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            b = eng.sample(cond, noise, graph=True)
            torch.cuda.synchronize()
            ok = bool(torch.isfinite(a).all()) and torch.equal(a, b)
            extra = ''
            if B > 1 and B * H * W <= 8 * 264 * 200:
                one = eng.sample(cond[:1].contiguous(), noise[:, :1].contiguous())
                extra = ' |batch - single| = %.2e' % (a[:1] - one).abs().max().item()
            print('%-6s B=%-3d %4dx%-4d ws=%7.1f MB  %.2fs  finite&graph==eager: %s  range [%.3f, %.3f]%s' % (
                prec, B, H, W, eng.workspace_bytes(B, H, W) / 1e6, dt, ok, a.min().item(), a.max().item(), extra), flush=True)
            del cond, noise, a, b
            torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
