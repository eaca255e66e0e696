"""The GDP sibling at the reference's own width (model/networks.py:88-104 + gdp_modules/unet.py:530-570: model_channels 128, mults
(1, 2, 4, 8) -> up to 1 024 channels and 16 attention heads, two ResBlocks per level, attention at downsample rates 8 / 16 / 32), which
the test suite's small networks do not reach: one optimisation step at 128 x 128 (B = 2) in exact fp32 and in f16x3 -- two disjoint
sets of convolution kernels -- compared tensor by tensor, a short sampling loop at 256 x 256 in f16x3 against bf16, and the step rate.

Usage (GPU box):  python tools/gdp_default_probe.py > gpurun_out/gdp_default_probe.txt"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from fastdiffsr_amd.arch import UNetConfig
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from fastdiffsr_amd.synth import synth_state_dict, synth_inputs
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=128, norm_groups=32, channel_mults=(1, 2, 4, 8), attn_res=(32, 16, 8),
                     res_blocks=2, dropout=0.0, image_size=128, variant='gdp')
    sd = synth_state_dict(cfg, 3)
    print(f'GDP at the reference width: {sum(v.size for v in sd.values()) / 1e6:.1f} M parameters in {len(sd)} tensors')
    sched = dict(schedule='linear', n_timestep=6, linear_start=1e-4, linear_end=2e-2)
    bufs, sp = schedule_buffers(sched)
    gen = torch.Generator().manual_seed(3)
    hr = torch.rand(2, 3, 128, 128, generator=gen) * 2 - 1
    sr = (hr + 0.1 * torch.randn(2, 3, 128, 128, generator=gen)).clamp(-1, 1)
    nz = torch.randn(2, 3, 128, 128, generator=gen)
    t = torch.tensor([1, 4])
    a = torch.from_numpy(np.asarray(bufs['sqrt_alphas_cumprod'], dtype=np.float32))[t].view(-1, 1, 1, 1)
    s = torch.from_numpy(np.asarray(bufs['sqrt_one_minus_alphas_cumprod'], dtype=np.float32))[t].view(-1, 1, 1, 1)
    x6 = torch.cat([a * hr + s * nz, sr], 1).cuda()
    grads, losses = {}, {}
    for prec in ('f32', 'f16x3'):
        eng = Engine(cfg)
        eng.load_state_dict(sd)
        eng.set_precision(prec)
        losses[prec] = eng.train_grads(x6, t.float().cuda(), hr.cuda(), 'l2', 1.0 / hr.numel())
        grads[prec] = {k: eng.get_grad(k) for k in sd}
        if prec == 'f16x3':
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                eng.train_grads(x6, t.float().cuda(), hr.cuda(), 'l2', 1.0 / hr.numel())
                eng.adam_step(1e-5)
            torch.cuda.synchronize()
            print(f'f16x3 optimisation step at 128 x 128, B = 2: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms')
        del eng
        torch.cuda.empty_cache()
    print(f'loss f32 {losses["f32"] / hr.numel():.6f}  f16x3 {losses["f16x3"] / hr.numel():.6f}')
    typical = float(np.median([np.abs(g).max() for g in grads['f32'].values()]))
    worst = (0.0, '')
    for k, g32 in grads['f32'].items():
        scale = float(np.abs(g32).max())
        if scale < 1e-4 * typical:
            continue
        d = float(np.abs(grads['f16x3'][k] - g32).max()) / scale
        worst = max(worst, (d, k))
    print(f'{len(grads["f32"])} gradients, f16x3 against exact fp32: worst {worst[1]} at {worst[0]:.3e} x max|g|')
    assert worst[0] <= 1e-4 and abs(losses['f32'] - losses['f16x3']) <= 1e-5 * abs(losses['f32'])
    # sampling at 256 x 256 (attention over 32 x 32 = 1 024 tokens with 16 heads at the deepest level)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_schedule(sampling_scalars(bufs, sp))
    cond, noise = synth_inputs(1, 256, 256, 7)
    outs = {}
    for prec in ('f32', 'f16x3', 'bf16'):
        eng.set_precision(prec)
        outs[prec] = eng.sample(cond.cuda(), noise.cuda()).cpu()
        assert torch.isfinite(outs[prec]).all()
    d = (outs['f16x3'] - outs['f32']).abs().max().item()
    rm = (outs['bf16'] - outs['f32']).pow(2).mean().sqrt().item()
    print(f'6-step sample at 256 x 256: f16x3 against exact fp32 max|d| {d:.3e}; bf16 against exact fp32 PSNR {20 * math.log10(2.0 / max(rm, 1e-12)):.2f} dB')
    assert d <= 1e-3


if __name__ == '__main__':
    main()
