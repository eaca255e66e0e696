"""The plugin surface (`networks.define_G` -> `GaussianDiffusion`) at the UNet settings of the reference's own config files
(config/sr_{fastdiffsr,ddpm,tesr,gdp}_train_64_256.json: batch 4, 256 x 256), for every `which_model_G`: the train phase's orthogonal
init, three `optimize_step`s (DDPM.optimize_parameters on the engine, Dropout(0.2) live), then the val phase's `super_resolution` on one
image (a 20-step schedule instead of the configs' 2 000 steps for the three siblings -- the loop is the same), `continous=True` included.
Prints parameter counts, the step time, the sampling time and the value ranges.

Usage (GPU box):  python tools/facade_reference_configs_probe.py > gpurun_out/facade_reference_configs_probe.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

UNETS = {   # "model.unet" of config/sr_<which>_train_64_256.json
    'fastdiffsr': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'channel_multiplier': [1, 2, 4, 4], 'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
    'ddpm': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'channel_multiplier': [1, 1, 2, 2, 4, 4], 'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
    'tesr': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'channel_multiplier': [1, 2, 4, 8, 8], 'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
    'gdp': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'channel_multiplier': [1, 2, 4, 8], 'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
}


def main():
    from fastdiffsr_amd import networks
    from fastdiffsr_amd.config import dict_to_nonedict
    train_sched = dict(schedule='linear', n_timestep=2000, linear_start=1e-6, linear_end=1e-2)
    val_sched = dict(schedule='linear_cosine', n_timestep=20, linear_start=1e-6, linear_end=1e-2)
    gen = torch.Generator().manual_seed(4)
    hr = (torch.rand(4, 3, 256, 256, generator=gen) * 2 - 1).cuda()
    sr = (hr + 0.1 * torch.randn(4, 3, 256, 256, generator=gen).cuda()).clamp(-1, 1)
    for which in (sys.argv[1:] or list(UNETS)):
        torch.manual_seed(7)
        np.random.seed(7)
        opt = dict_to_nonedict({
            'phase': 'train', 'gpu_ids': [0], 'distributed': False,
            'datasets': {'train': {'l_resolution': 64, 'r_resolution': 256}},
            'model': {'which_model_G': which, 'finetune_norm': False, 'unet': dict(UNETS[which]),
                      'beta_schedule': {'train': dict(train_sched), 'val': dict(val_sched)},
                      'diffusion': {'image_size': 256, 'channels': 3, 'conditional': True}}})
        netG = networks.define_G(opt).cuda()
        netG.set_loss('cuda')
        netG.set_new_noise_schedule(dict(train_sched), 'cuda')
        netG.train()
        n_par = sum(p.numel() for p in netG.denoise_fn.parameters())
        data = {'HR': hr, 'SR': sr, 'LR': sr}
        losses = [netG.optimize_step(data, lr=1e-4)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses += [netG.optimize_step(data, lr=1e-4) for _ in range(3)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        assert all(np.isfinite(losses)), losses
        netG.set_new_noise_schedule(dict(val_sched), 'cuda')
        netG.eval()
        with torch.no_grad():
            out = netG.super_resolution(sr[:1], continous=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = netG.super_resolution(sr[:1], continous=False)
            torch.cuda.synchronize()
            ts = time.perf_counter() - t0
            frames = netG.super_resolution(sr[:1], continous=True)
        assert torch.isfinite(out).all() and torch.isfinite(frames).all()
        print(f'{which:10s} {n_par / 1e6:7.1f} M parameters | optimize_step B=4 256x256: {dt * 1e3:7.1f} ms ({4 / dt:6.1f} img/s), l_pix '
              f'{losses[0]:.4g} -> {losses[-1]:.4g} | super_resolution 1 x 256x256, T=20: {ts * 1e3:6.1f} ms, out {tuple(out.shape)} in '
              f'[{out.min().item():+.3f}, {out.max().item():+.3f}], continous frames {tuple(frames.shape)}')
        del netG
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
