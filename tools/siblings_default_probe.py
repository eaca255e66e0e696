"""The three sibling denoisers at the sizes the reference's own configs give them (config/sr_{ddpm,tesr,gdp}_train_64_256.json +
model/networks.py:82-104), which the test suite's small networks do not reach:

  ddpm  inner 64, mults (1, 1, 2, 2, 4, 4), attention at 16 x 16, two ResBlocks per level            (SR3)
  tesr  inner 64, mults (1, 2, 4, 8, 8), attention at 16 x 16
  gdp   model_channels 128 (define_G does not pass `inner_channel` on), mults (1, 2, 4, 8) -> 1 024 channels and 16 heads,
        attention at downsample rates 8 / 16 / 32

For each: one optimisation step at 256 x 256 (B = 2; gdp: 128 x 128) in exact fp32 and in f16x3 -- two disjoint sets of convolution
kernels -- compared tensor by tensor, the step rate, and a 6-step sampling loop at 256 x 256 in f16x3 and bf16 against exact fp32.

Usage (GPU box):  python tools/siblings_default_probe.py [ddpm tesr gdp] > gpurun_out/siblings_default_probe.txt"""
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = {
    'ddpm': dict(inner_channel=64, channel_mults=(1, 1, 2, 2, 4, 4), attn_res=(16,), size=256),
    'tesr': dict(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), attn_res=(16,), size=256),
    'gdp': dict(inner_channel=128, channel_mults=(1, 2, 4, 8), attn_res=(32, 16, 8), size=128),
}


def probe(variant):
    from fastdiffsr_amd.arch import UNetConfig
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from fastdiffsr_amd.synth import synth_state_dict, synth_inputs
    c = CONFIGS[variant]
    S = c['size']
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=c['inner_channel'], norm_groups=32, channel_mults=c['channel_mults'],
                     attn_res=c['attn_res'], res_blocks=2, dropout=0.0, image_size=256, variant=variant)
    sd = synth_state_dict(cfg, 3)
    print(f'{variant} at the reference size: {sum(v.size for v in sd.values()) / 1e6:.1f} M parameters in {len(sd)} tensors')
    sched = dict(schedule='linear', n_timestep=6, linear_start=1e-4, linear_end=2e-2)
    bufs, sp = schedule_buffers(sched)
    gen = torch.Generator().manual_seed(3)
    x6 = (torch.rand(2, 6, S, S, generator=gen) * 2 - 1).cuda()
    target = torch.randn(2, 3, S, S, generator=gen).cuda()
    nl = (torch.tensor([0.35, 0.8]) if variant == 'tesr' else torch.tensor([1.0, 4.0])).cuda()   # gamma (tesr) / integer time
    loss_type = 'l2' if variant == 'gdp' else 'l1'
    numel = target.numel()
    grads, losses = {}, {}
    for prec in ('f32', 'f16x3'):
        eng = Engine(cfg)
        eng.load_state_dict(sd)
        eng.set_precision(prec)
        losses[prec] = eng.train_grads(x6, nl, target, loss_type, 1.0 / numel)
        live = {k for k, _, lv in eng.schema() if lv}
        grads[prec] = {k: eng.get_grad(k) for k in sd if k in live}
        if prec == 'f16x3':
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                eng.train_grads(x6, nl, target, loss_type, 1.0 / numel)
                eng.adam_step(1e-5)
            torch.cuda.synchronize()
            print(f'{variant}: f16x3 optimisation step at {S} x {S}, B = 2: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms')
        del eng
        torch.cuda.empty_cache()
    print(f'{variant}: loss f32 {losses["f32"] / numel:.6f}  f16x3 {losses["f16x3"] / numel:.6f}')
    typical = float(np.median([np.abs(g).max() for g in grads['f32'].values()]))
    worst = (0.0, '')
    for k, g32 in grads['f32'].items():
        scale = float(np.abs(g32).max())
        if scale < 1e-4 * typical:
            continue
        d = float(np.abs(grads['f16x3'][k] - g32).max()) / scale
        worst = max(worst, (d, k))
    print(f'{variant}: {len(grads["f32"])} gradients, f16x3 against exact fp32: worst {worst[1]} at {worst[0]:.3e} x max|g|')
    assert worst[0] <= 1e-4 and abs(losses['f32'] - losses['f16x3']) <= 1e-5 * abs(losses['f32'])
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_schedule(sampling_scalars(bufs, sp))
    cond, noise = synth_inputs(1, 256, 256, 7 if variant in ('ddpm', 'gdp') else 6)
    outs = {}
    for prec in ('f32', 'f16x3', 'bf16'):
        eng.set_precision(prec)
        outs[prec] = eng.sample(cond.cuda(), noise.cuda()).cpu()
        assert torch.isfinite(outs[prec]).all()
    d = (outs['f16x3'] - outs['f32']).abs().max().item()
    rm = (outs['bf16'] - outs['f32']).pow(2).mean().sqrt().item()
    print(f'{variant}: 6-step sample at 256 x 256: f16x3 against exact fp32 max|d| {d:.3e}; bf16 against exact fp32 PSNR '
          f'{20 * math.log10(2.0 / max(rm, 1e-12)):.2f} dB')
    assert d <= 1e-3
    del eng
    torch.cuda.empty_cache()


if __name__ == '__main__':
    for v in (sys.argv[1:] or ['ddpm', 'tesr', 'gdp']):
        probe(v)
