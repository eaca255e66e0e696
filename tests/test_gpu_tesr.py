"""SURVEY 8f-4, second sibling: TESR (reference model/tesr_modules) through the HIP engine, against goldens produced
by the reference modules themselves (tests/golden/tesr.npz) and the oracle restatement."""
import os
from unittest import mock

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, build_layers, TESR_UNET, TESR_SCHEDULE_VAL
from fastdiffsr_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
CFG = dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
           attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='tesr')
SCHED = dict(schedule='linear', n_timestep=10, linear_start=1e-4, linear_end=2e-2)


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_tesr_unet_and_loop_vs_reference_goldens(golden_dir, prec):
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from oracle import fdsr_oracle as O, tesr_oracle as TO
    g = np.load(os.path.join(golden_dir, 'tesr.npz'))
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 9)
    eng = Engine(cfg)
    assert [k for k, _, _ in eng.schema()] == list(sd.keys())            # the engine's schema is the reference's, in order
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    x = torch.from_numpy(g['x'])
    cap = {}
    with torch.no_grad():
        TO.unet_forward(O.to_torch_sd(sd), cfg, x, torch.full((2, 1), 0.5), capture=cap)
    eng.set_debug(True)
    out = eng.unet_forward(x.cuda(), torch.full((2, 1), 0.5).cuda()).cpu().numpy()
    for L in build_layers(cfg):                                           # layer by layer, attention blocks included
        d = (eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item()
        assert d <= 1e-4 * max(1.0, cap[L.name].abs().max().item()), (L.name, d)
    eng.set_debug(False)
    assert np.abs(out - g['eps/0']).max() <= 1e-4
    for i in (1, 2):
        o = eng.unet_forward(x.cuda(), torch.full((2, 1), float(g[f'nl/{i}'])).cuda()).cpu().numpy()
        assert np.abs(o - g[f'eps/{i}']).max() <= 1e-4
    # the reference's own p_sample_loop(continous=True), T = 10: every x_t is kept
    bufs, sp = schedule_buffers(SCHED)
    eng.set_schedule(sampling_scalars(bufs, sp))
    cond, noise = torch.from_numpy(g['cond']).cuda(), torch.from_numpy(g['noise']).cuda()
    img, traj = eng.sample(cond, noise, want_traj=True)
    got = torch.cat([cond] + [traj[k] for k in range(10)]).cpu().numpy()
    assert np.abs(got - g['frames']).max() <= 1e-3
    assert np.abs(img.cpu().numpy()[0] - g['frames'][-1]).max() <= 1e-3   # x_0 itself: no res2img in this sibling


def test_tesr_bf16_mode_vs_reference_goldens(golden_dir):
    """bf16 mode (bf16 activations in HBM, bf16 MFMA convolutions and the bf16 QK^T / PV attention kernels): judged like the
    flagship's bf16 mode -- layer-wise inside a quarter of each layer's range, the sampled image on PSNR against the
    reference's own x_0."""
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from oracle import fdsr_oracle as O, tesr_oracle as TO
    from test_gpu_parity import report
    g = np.load(os.path.join(golden_dir, 'tesr.npz'))
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 9)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision('bf16')
    x = torch.from_numpy(g['x'])
    cap = {}
    with torch.no_grad():
        TO.unet_forward(O.to_torch_sd(sd), cfg, x, torch.full((2, 1), 0.5), capture=cap)
    eng.set_debug(True)
    out = eng.unet_forward(x.cuda(), torch.full((2, 1), 0.5).cuda()).cpu().numpy()
    worst = 0.0
    for L in build_layers(cfg):
        scale = max(1.0, cap[L.name].abs().max().item())
        d = (eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item()
        worst = max(worst, d / scale)
        assert d <= 0.04 * scale, (L.name, d)
    eng.set_debug(False)
    d_eps = np.abs(out - g['eps/0']).max()
    bufs, sp = schedule_buffers(SCHED)
    eng.set_schedule(sampling_scalars(bufs, sp))
    cond, noise = torch.from_numpy(g['cond']).cuda(), torch.from_numpy(g['noise']).cuda()
    img = eng.sample(cond, noise).cpu()
    ref = torch.from_numpy(g['frames'][-1])
    d_img = (img[0] - ref).abs().max().item()
    psnr = O.psnr_u8(O.tensor2img_u8(img[0]), O.tensor2img_u8(ref))
    report(f'tesr bf16: worst layer {worst:.3e} of range, eps max|d|={d_eps:.3e}, x_0 max|d|={d_img:.3e}, '
           f'PSNR(x_0 bf16, x_0 reference)={psnr:.2f} dB')
    assert d_eps <= 0.05 and psnr >= 40.0


def test_tesr_facade_reference_config_and_loss(golden_dir):
    """define_G(which_model_G='tesr'): strict checkpoint exchange with the reference's key set, the sampler's return
    convention, the Charbonnier loss value against the reference's own, and the reference's x4 config (5 levels up to
    512 channels, 256-token attention at 16x16) for one forward against the oracle."""
    from fastdiffsr_amd import networks
    from oracle import fdsr_oracle as O, tesr_oracle as TO
    g = np.load(os.path.join(golden_dir, 'tesr.npz'))

    def opt_for(unet_kw, sched, image_size):
        return {'phase': 'val', 'gpu_ids': [0], 'distributed': False, 'datasets': {'train': {'l_resolution': 64}},
                'model': {'which_model_G': 'tesr', 'finetune_norm': False,
                          'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': unet_kw['inner_channel'],
                                   'norm_groups': 32, 'channel_multiplier': list(unet_kw['channel_mults']),
                                   'attn_res': list(unet_kw['attn_res']), 'res_blocks': unet_kw['res_blocks'], 'dropout': 0.2},
                          'beta_schedule': {'train': dict(sched), 'val': dict(sched)},
                          'diffusion': {'image_size': image_size, 'channels': 3, 'conditional': True}}}
    dev = torch.device('cuda')
    netG = networks.define_G(opt_for(CFG, SCHED, 32)).to(dev)
    netG.set_loss(dev)
    netG.set_new_noise_schedule(SCHED, dev)
    sd = synth_state_dict(UNetConfig(**CFG), 9, prefix='denoise_fn.')
    ck = {k: torch.from_numpy(v) for k, v in sd.items()}
    ck.update({k: v.cpu() for k, v in netG.state_dict().items() if not k.startswith('denoise_fn.')})
    netG.load_state_dict(ck, strict=True)
    netG.eval()
    cond, noise = torch.from_numpy(g['cond']).to(dev), torch.from_numpy(g['noise']).to(dev)
    out = netG.p_sample_loop(cond, continous=False, noise=noise)
    assert out.shape == (3, 32, 32) and np.abs(out.cpu().numpy() - g['frames'][-1]).max() <= 1e-3   # ret_img[-1]
    frames = netG.p_sample_loop(cond, continous=True, noise=noise)
    assert np.abs(frames.cpu().numpy() - g['frames']).max() <= 1e-3
    hr, sr, nz = (torch.from_numpy(g[k]).to(dev) for k in ('hr', 'sr', 'loss_noise'))
    gam = g['gamma']
    with mock.patch.object(np.random, 'randint', lambda a, b: 4), mock.patch.object(np.random, 'uniform', lambda a, b, size: gam):
        loss = netG({'HR': hr, 'SR': sr}, noise=nz)
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))
    # the reference's TESR x4 config at 64x64 (attention where image_size / 2^level == 16 and in mid[0])
    big = networks.define_G(opt_for(TESR_UNET, TESR_SCHEDULE_VAL, 256)).to(dev)
    big.eval()                                       # train mode has live dropout (refused until the mask kernel exists)
    cfg = UNetConfig(**TESR_UNET)
    sdb = synth_state_dict(cfg, 4)
    big.denoise_fn.load_state_dict({k: torch.from_numpy(v) for k, v in sdb.items()}, strict=True)
    gen = torch.Generator().manual_seed(29)
    x = torch.randn(1, 6, 256, 256, generator=gen)
    nl = torch.tensor([[0.37]])
    with torch.no_grad():
        ref = TO.unet_forward(O.to_torch_sd(sdb), cfg, x, nl)
        got = big.denoise_fn(x.to(dev), nl.to(dev)).cpu()
    assert (got - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
