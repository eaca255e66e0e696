"""bf16 sampling with some reverse steps on the fp32-grade kernels (debug option "bf16_f16x3_steps"), measured on TRAINED-LIKE weights
(the fixture of tests/test_gpu_trained_weights.py: orthogonal init + 240 optimisation steps on the engine):

  * per count n (the first n loop iterations in f16x3, the rest bf16; n < 0: the last -n): PSNR(image, oracle image), rmse, and the
    PSNR difference against HR -- beside coef1[t] * sqrt_recipm1[t] of the schedule (the factor by which a step's network-output error
    enters x_{t-1}; NOT what it costs in the image: the high-noise steps' errors are denoised away by the steps after them);
  * the throughput of the 20-step loop at B = 64 (hipGraph, engine-drawn noise) for n = 0, 6, 10 and pure f16x3 at B = 16.

Usage (GPU box):  python tools/step_precision_probe.py > gpurun_out/step_precision_probe.txt"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import test_gpu_trained_weights as tw
    from conftest import oracle_loop_image
    from fastdiffsr_amd.arch import FASTDIFFSR_SCHEDULE_VAL
    from fastdiffsr_amd import _lib
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from fastdiffsr_amd.synth import synth_inputs
    from oracle import fdsr_oracle as O
    cfg, sd, losses = tw.make_trained()
    print(f'trained-like weights: l_pix {np.mean(losses[:20]):.4f} -> {np.mean(losses[-20:]):.4f}')
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    sc = sampling_scalars(bufs, sp)
    w = (np.asarray(sc['coef1'], dtype=np.float64) * np.asarray(sc['sqrt_recipm1'], dtype=np.float64))[::-1]
    print('error weight per loop iteration (k = 0 is t = T-1):', ' '.join(f'{x:.3g}' for x in w))
    hr, sr = tw._pairs(1, 256, 77)
    _, noise = synth_inputs(1, 256, 256, 20)
    ref = oracle_loop_image(sd, cfg, sr, noise)
    u8 = lambda t: O.tensor2img_u8(t[0].clone())
    p_ref = O.psnr_u8(u8(ref), u8(hr))
    print(f'oracle image against HR: {p_ref:.3f} dB')
    eng.set_precision('bf16')
    for n in [0, 1, 2, 3, 4, 5, 6, 10, 15, -3, -10]:
        _lib.check(eng.h, eng.lib.fdsr_debug_option(b'bf16_f16x3_steps', n))
        tag = f'the first {n} steps in f16x3' if n >= 0 else f'the LAST {-n} steps in f16x3'
        out = eng.sample(sr.cuda(), noise.cuda()).cpu()
        g = eng.sample(sr.cuda(), noise.cuda(), graph=True).cpu()
        rmse = (out - ref).pow(2).mean().sqrt().item()
        dps = O.psnr_u8(u8(out), u8(hr)) - p_ref
        print(f'bf16, {tag:32s}: PSNR(out, oracle) {20 * math.log10(2.0 / max(rmse, 1e-12)):6.2f} dB  rmse {rmse:.3e}  max|d| '
              f'{(out - ref).abs().max().item():.3e}  PSNR delta vs HR {dps:+.5f} dB  graph == eager {torch.equal(g, out)}')
    _lib.check(eng.h, eng.lib.fdsr_debug_option(b'bf16_f16x3_steps', 0))
    eng.set_precision('f16x3')
    out = eng.sample(sr.cuda(), noise.cuda()).cpu()
    print(f'f16x3: max|d| {(out - ref).abs().max().item():.3e}')

    def rate(B, prec, steps, reps=6):
        eng.set_precision(prec)
        _lib.check(eng.h, eng.lib.fdsr_debug_option(b'bf16_f16x3_steps', steps))
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
        print(f'B=64 bf16: {rate(64, "bf16", 0):7.2f} img/s | bf16 with 6 / 10 steps in f16x3: {rate(64, "bf16", 6):7.2f} / {rate(64, "bf16", 10):7.2f} img/s | '
              f'B=16 f16x3: {rate(16, "f16x3", 0):7.2f} img/s')


if __name__ == '__main__':
    main()
