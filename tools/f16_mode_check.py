"""First light of the f16 mode (FDSR_PREC_F16): the 20-step loop against the f16x3 result at several batch sizes (all kernel selections),
graph == eager, and the B = 64 rate beside bf16.   python tools/f16_mode_check.py"""
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from fastdiffsr_amd.synth import synth_state_dict, synth_inputs
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    eng = Engine(cfg)
    eng.load_state_dict(synth_state_dict(cfg, 0))
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    for B, S in ((1, 256), (2, 256), (5, 128), (16, 256), (1, 512)):
        cond, noise = synth_inputs(B, S, S, 20)
        c, n = cond.cuda(), noise.cuda()
        eng.set_precision('f16x3')
        ref = eng.sample(c, n).clone()
        res = {}
        for prec in ('bf16', 'f16'):
            eng.set_precision(prec)
            out = eng.sample(c, n).clone()
            g = eng.sample(c, n, graph=True).clone()
            rm = (out - ref).pow(2).mean().sqrt().item()
            res[prec] = (20 * math.log10(2.0 / max(rm, 1e-12)), (out - ref).abs().max().item(), torch.equal(out, g), bool(torch.isfinite(out).all()))
        print(f'B={B} {S}x{S}: ' + ' | '.join(f'{k}: PSNR vs f16x3 {v[0]:6.2f} dB, max|d| {v[1]:.3e}, graph==eager {v[2]}, finite {v[3]}' for k, v in res.items()), flush=True)

    def rate(B, prec, reps=5):
        eng.set_precision(prec)
        cond, _ = synth_inputs(B, 256, 256, 1)
        cond = cond.cuda()
        eng.set_seed(1)
        o = torch.empty(B, 3, 256, 256, device='cuda')
        for _ in range(2):
            eng.sample(cond, None, graph=True, out=o)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.sample(cond, None, graph=True, out=o)
        torch.cuda.synchronize()
        return B * reps / (time.perf_counter() - t0)

    for rnd in range(2):
        print(f'B=64 graph: bf16 {rate(64, "bf16"):7.2f} img/s | f16 {rate(64, "f16"):7.2f} img/s | B=1: bf16 {rate(1, "bf16", 20):6.2f} f16 {rate(1, "f16", 20):6.2f}', flush=True)


if __name__ == '__main__':
    main()
