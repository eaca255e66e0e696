"""GDP sibling (`which_model_G == 'gdp'`, FastDiffSR/model/gdp_modules: the guided-diffusion UNet) on the HIP engine:
block-by-block and end-to-end against the oracle and the reference's own outputs (tests/golden/gdp.npz)."""
import os
from unittest import mock

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, GDP_UNET
from fastdiffsr_amd.gdp.arch import gdp_layers
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu

CFG = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 2), attn_res=(2, 4), res_blocks=1,
           dropout=0.1, image_size=32, variant='gdp')
SCHED = dict(schedule='linear', n_timestep=8, linear_start=1e-4, linear_end=2e-2)


def _engine(cfg, seed, sched=None):
    from fastdiffsr_amd.engine import Engine
    eng = Engine(cfg)
    sd = synth_state_dict(cfg, seed)
    eng.load_state_dict(sd)
    if sched:
        bufs, sp = schedule_buffers(sched)
        eng.set_schedule(sampling_scalars(bufs, sp))
    return eng, sd


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_gdp_blocks_and_forward_vs_reference(golden_dir, prec):
    from oracle import fdsr_oracle as O, gdp_oracle as GO
    g = np.load(os.path.join(golden_dir, 'gdp.npz'))
    cfg = UNetConfig(**CFG)
    eng, sd = _engine(cfg, 13)
    eng.set_precision(prec)
    x = torch.from_numpy(g['x'])
    tsd = O.to_torch_sd(sd)
    for i in range(3):
        t = torch.from_numpy(g[f't/{i}'])
        cap = {}
        with torch.no_grad():
            GO.unet_forward(tsd, cfg, x, t, capture=cap)
        eng.set_debug(True)
        out = eng.unet_forward(x.cuda(), t.float().cuda()).cpu().numpy()
        torch.cuda.synchronize()
        for L in gdp_layers(cfg):
            if not L.block or L.block == 'out':
                continue
            got = eng.debug_tensor(L.block).cpu()
            d = (got - cap[L.block]).abs().max().item()
            scale = max(1.0, cap[L.block].abs().max().item())
            assert d <= 1e-4 * scale, f'{L.block} (t={t.tolist()}): {d:.3e} at |ref| {scale:.2f} [{prec}]'
        eng.set_debug(False)
        ref = g[f'rec/{i}']
        assert np.abs(out - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), i


def test_gdp_bf16_mode_vs_reference(golden_dir):
    """bf16 mode on the guided-diffusion UNet: bf16 activations through the convs, the 64-channel-head attention on bf16 MFMA and the
    bf16 forms of the materialised avg-pool / nearest-x2 kernels.  Judged like the flagship's bf16 mode: block outputs inside a
    quarter of their range, the sampled x_0 on PSNR against the reference's own frames."""
    from oracle import fdsr_oracle as O, gdp_oracle as GO
    from test_gpu_parity import report
    g = np.load(os.path.join(golden_dir, 'gdp.npz'))
    cfg = UNetConfig(**CFG)
    eng, sd = _engine(cfg, 13, SCHED)
    eng.set_precision('bf16')
    x = torch.from_numpy(g['x'])
    t = torch.from_numpy(g['t/1'])
    cap = {}
    with torch.no_grad():
        GO.unet_forward(O.to_torch_sd(sd), cfg, x, t, capture=cap)
    eng.set_debug(True)
    out = eng.unet_forward(x.cuda(), t.float().cuda()).cpu().numpy()
    worst = 0.0
    for L in gdp_layers(cfg):
        if not L.block or L.block == 'out':
            continue
        scale = max(1.0, cap[L.block].abs().max().item())
        d = (eng.debug_tensor(L.block).cpu() - cap[L.block]).abs().max().item()
        worst = max(worst, d / scale)
        assert d <= 0.04 * scale, (L.block, d)
    eng.set_debug(False)
    ref = g['rec/1']
    d_out = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
    cond, noise = torch.from_numpy(g['cond']).cuda(), torch.from_numpy(g['noise']).cuda()
    img = eng.sample(cond, noise).cpu()
    x0 = torch.from_numpy(g['frames'][-1])
    d_img = (img[0] - x0).abs().max().item()
    psnr = O.psnr_u8(O.tensor2img_u8(img[0]), O.tensor2img_u8(x0))
    report(f'gdp bf16: worst block {worst:.3e} of range, forward {d_out:.3e} of range, x_0 (T=8) max|d|={d_img:.3e}, '
           f'PSNR(x_0 bf16, x_0 reference)={psnr:.2f} dB')
    # (measured 41.3 dB: the x_0-prediction posterior of this sibling re-amplifies eps errors by 1/sqrt(alpha_bar) at every step)
    assert d_out <= 0.05 and psnr >= 35.0


def test_gdp_sampler_vs_reference_frames(golden_dir):
    """p_sample_loop(continous=True) of the reference at T=8: x_0-prediction posterior, cat([x_t, cond]), noise at every
    step (the last masked)."""
    g = np.load(os.path.join(golden_dir, 'gdp.npz'))
    cfg = UNetConfig(**CFG)
    eng, sd = _engine(cfg, 13, SCHED)
    cond, noise = torch.from_numpy(g['cond']).cuda(), torch.from_numpy(g['noise']).cuda()
    for prec in ('f32', 'f16x3'):
        eng.set_precision(prec)
        out, traj = eng.sample(cond, noise, want_traj=True)
        frames = g['frames']
        per_step = np.abs(traj.cpu().numpy() - frames[1:, None]).reshape(8, -1).max(axis=1)
        print(f'gdp loop [{prec}] per-step max|d|: ' + ' '.join(f'{v:.1e}' for v in per_step))
        assert per_step.max() <= 1e-3
        assert np.abs(out.cpu().numpy()[0] - frames[-1]).max() <= 1e-3
    with pytest.raises(Exception):
        eng.sample(cond, noise[:8].contiguous())               # GDP wants T+1 noise planes


def test_gdp_facade_define_g_and_loss(golden_dir):
    from fastdiffsr_amd import networks
    g = np.load(os.path.join(golden_dir, 'gdp.npz'))
    opt = {'phase': 'val', 'gpu_ids': [0], 'distributed': False, 'datasets': {'train': {'l_resolution': 64}},
           'model': {'which_model_G': 'gdp', 'finetune_norm': False,
                     'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'norm_groups': 32,
                              'channel_multiplier': [1, 2, 4, 8], 'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
                     'beta_schedule': {'train': dict(SCHED), 'val': dict(SCHED)},
                     'diffusion': {'image_size': 256, 'channels': 3, 'conditional': True}}}
    big = networks.define_G(opt)                               # the reference's config: model_channels stays 128
    assert big.denoise_fn.cfg.inner_channel == 128 and big.denoise_fn.cfg.attn_res == (32, 16, 8)
    assert sum(p.numel() for p in big.parameters()) == 271417731          # the guided-diffusion UNet at model_channels 128, mults 1-2-4-8
    # the small network of the golden, through the facade classes
    from fastdiffsr_amd.gdp import diffusion, unet
    dev = torch.device('cuda')
    net = unet.UNet(image_size=32, in_channel=6, model_channels=64, out_channel=3, res_blocks=1, attention_resolutions=(2, 4),
                    dropout=0.1, channel_mults=(1, 2, 2), inner_channel=64, norm_groups=32, attn_res=(16,))
    netG = diffusion.GaussianDiffusion(net, image_size=32, channels=3, loss_type='l1', conditional=True, schedule_opt=SCHED).to(dev)
    netG.set_loss(dev)
    netG.set_new_noise_schedule(SCHED, dev)
    sd = synth_state_dict(UNetConfig(**CFG), 13, prefix='denoise_fn.')
    ck = {k: torch.from_numpy(v) for k, v in sd.items()}
    ck.update({k: v.cpu() for k, v in netG.state_dict().items() if not k.startswith('denoise_fn.')})
    assert [k[len('denoise_fn.'):] for k in netG.state_dict() if k.startswith('denoise_fn.')] == [str(k) for k in g['keys']]
    netG.load_state_dict(ck, strict=True)
    netG.eval()
    cond, noise = torch.from_numpy(g['cond']).to(dev), torch.from_numpy(g['noise']).to(dev)
    frames = netG.p_sample_loop(cond, continous=True, noise=noise)
    assert np.abs(frames.cpu().numpy() - g['frames']).max() <= 1e-3
    out = netG.p_sample_loop(cond, continous=False, noise=noise)
    assert out.shape == (3, 32, 32) and np.abs(out.cpu().numpy() - g['frames'][-1]).max() <= 1e-3
    hr, sr, nz = (torch.from_numpy(g[k]).to(dev) for k in ('hr', 'sr', 'loss_noise'))
    tt = torch.from_numpy(g['loss_t']).to(dev)
    with mock.patch.object(torch, 'randint', lambda *a, **k: tt):
        loss = netG({'HR': hr, 'SR': sr, 'LR': sr}, noise=nz)
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))


def test_gdp_reference_config_forward_vs_oracle():
    """The reference's x4 config (model_channels 128, mults 1-2-4-8, 16-head attention on 1024 channels at the deepest
    level and in the middle block) at 64x64: one forward against the oracle."""
    from oracle import fdsr_oracle as O, gdp_oracle as GO
    cfg = UNetConfig(**GDP_UNET)
    eng, sd = _engine(cfg, 2)
    gen = torch.Generator().manual_seed(31)
    x = torch.randn(1, 6, 64, 64, generator=gen)
    t = torch.tensor([417], dtype=torch.long)
    with torch.no_grad():
        ref = GO.unet_forward(O.to_torch_sd(sd), cfg, x, t)
    for prec in ('f32', 'f16x3'):
        eng.set_precision(prec)
        got = eng.unet_forward(x.cuda(), t.float().cuda()).cpu()
        d = (got - ref).abs().max().item()
        print(f'gdp reference config 64x64 [{prec}]: max|d| {d:.3e} (max|ref| {ref.abs().max().item():.2f})')
        assert d <= 1e-4 * max(1.0, ref.abs().max().item())
