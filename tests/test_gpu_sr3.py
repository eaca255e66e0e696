"""the SR3 sibling (reference model/ddpm_modules) through the HIP engine, against goldens
produced by the reference modules themselves (tests/golden/sr3.npz) and the oracle restatement."""
import os

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, build_layers, SR3_UNET
from fastdiffsr_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
CFG = dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
           attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='ddpm')
SCHED = dict(schedule='linear', n_timestep=12, linear_start=1e-4, linear_end=2e-2)


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_sr3_unet_and_loop_vs_reference_goldens(golden_dir, prec):
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from oracle import fdsr_oracle as O, sr3_oracle as S
    g = np.load(os.path.join(golden_dir, 'sr3.npz'))
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 5)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    x = torch.from_numpy(g['x'])
    # layer by layer (incl. the attention blocks) against the oracle, then the reference goldens
    cap = {}
    with torch.no_grad():
        S.unet_forward(O.to_torch_sd(sd), cfg, x, torch.tensor([3, 999]), capture=cap)
    eng.set_debug(True)
    out = eng.unet_forward(x.cuda(), torch.tensor([3., 999.]).cuda()).cpu().numpy()
    for L in build_layers(cfg):
        d = (eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item()
        assert d <= 1e-4 * max(1.0, cap[L.name].abs().max().item()), (L.name, d)
    eng.set_debug(False)
    assert np.abs(out - g['eps_t']).max() <= 1e-4
    out0 = eng.unet_forward(x.cuda(), torch.zeros(2).cuda()).cpu().numpy()
    assert np.abs(out0 - g['eps_t0']).max() <= 1e-4
    # the reference's own p_sample_loop(continous=True), T = 12
    bufs, sp = schedule_buffers(SCHED)
    eng.set_schedule(sampling_scalars(bufs, sp))
    cond, noise = torch.from_numpy(g['cond']).cuda(), torch.from_numpy(g['noise']).cuda()
    img, traj = eng.sample(cond, noise, want_traj=True)
    ref = g['continous']
    got = torch.cat([cond] + [traj[k] for k in range(12)]).cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-3
    assert np.abs(img.cpu().numpy() - ref[-2:]).max() <= 1e-3


def test_sr3_bf16_mode_vs_reference_goldens(golden_dir):
    """bf16 mode for the attending sibling: bf16 activations, bf16 MFMA convolutions, QK^T and PV on
    v_mfma_f32_32x32x16_bf16 with fp32 scores / softmax.  Judged like the flagship's bf16 mode."""
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from oracle import fdsr_oracle as O, sr3_oracle as S
    from test_gpu_parity import report
    g = np.load(os.path.join(golden_dir, 'sr3.npz'))
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 5)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision('bf16')
    x = torch.from_numpy(g['x'])
    cap = {}
    with torch.no_grad():
        S.unet_forward(O.to_torch_sd(sd), cfg, x, torch.tensor([3, 999]), capture=cap)
    eng.set_debug(True)
    out = eng.unet_forward(x.cuda(), torch.tensor([3., 999.]).cuda()).cpu().numpy()
    worst = 0.0
    for L in build_layers(cfg):
        scale = max(1.0, cap[L.name].abs().max().item())
        d = (eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item()
        worst = max(worst, d / scale)
        assert d <= 0.04 * scale, (L.name, d)
    eng.set_debug(False)
    d_eps = np.abs(out - g['eps_t']).max()
    bufs, sp = schedule_buffers(SCHED)
    eng.set_schedule(sampling_scalars(bufs, sp))
    cond, noise = torch.from_numpy(g['cond']).cuda(), torch.from_numpy(g['noise']).cuda()
    img = eng.sample(cond, noise).cpu()
    ref = torch.from_numpy(g['continous'][-2:])
    d_img = (img - ref).abs().max().item()
    psnr = min(O.psnr_u8(O.tensor2img_u8(img[i]), O.tensor2img_u8(ref[i])) for i in range(2))
    report(f'sr3 bf16: worst layer {worst:.3e} of range, eps max|d|={d_eps:.3e}, x_0 max|d|={d_img:.3e}, '
           f'PSNR(x_0 bf16, x_0 reference)={psnr:.2f} dB')
    assert d_eps <= 0.05 and psnr >= 40.0


@pytest.mark.parametrize('size', [(24, 24), (24, 40)])
def test_attention_token_counts_off_the_tile_grid(size):
    """Token counts that are not multiples of the 32-query tiles or of the 16-key MFMA step (36 / 60 tokens at the attn_res level,
    9 / 15 in `mid`): the clamped loads, the zero pad keys and the masked stores of both attention kernel pairs, against the oracle --
    f32 and f16x3 at 1e-4, bf16 inside its range bound."""
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O, sr3_oracle as S
    H, W = size
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                     attn_res=(H // 4,), res_blocks=1, dropout=0.0, image_size=H, variant='ddpm')
    sd = synth_state_dict(cfg, 21)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    x = torch.randn(2, 6, H, W, generator=torch.Generator().manual_seed(4))
    t = torch.tensor([7, 640])
    cap = {}
    with torch.no_grad():
        ref = S.unet_forward(O.to_torch_sd(sd), cfg, x, t, capture=cap)
    attn_layers = [L.name for L in build_layers(cfg) if L.with_attn]
    assert attn_layers
    for prec, tol in (('f32', 1e-4), ('f16x3', 1e-4), ('bf16', 0.04)):
        eng.set_precision(prec)
        eng.set_debug(True)
        out = eng.unet_forward(x.cuda(), t.float().cuda()).cpu()
        for name in attn_layers:
            scale = max(1.0, cap[name].abs().max().item())
            d = (eng.debug_tensor(name).cpu() - cap[name]).abs().max().item()
            assert d <= tol * scale, (prec, name, d)
        eng.set_debug(False)
        assert (out - ref).abs().max().item() <= (1e-4 if tol < 0.01 else 0.05) * max(1.0, ref.abs().max().item()), prec


def test_sr3_facade_and_reference_config():
    """define_G(which_model_G='ddpm') with the reference's SR3 config (6 levels, attention at 16x16 and in mid),
    strict checkpoint exchange, one forward at 64x64 against the oracle."""
    from fastdiffsr_amd import networks
    from oracle import fdsr_oracle as O, sr3_oracle as S
    opt = {'phase': 'val', 'gpu_ids': [0], 'distributed': False, 'datasets': {'train': {'l_resolution': 64}},
           'model': {'which_model_G': 'ddpm', 'finetune_norm': False,
                     'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'norm_groups': 32,
                              'channel_multiplier': [1, 1, 2, 2, 4, 4], 'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
                     'beta_schedule': {'train': dict(SCHED), 'val': dict(SCHED)},
                     'diffusion': {'image_size': 256, 'channels': 3, 'conditional': True}}}
    netG = networks.define_G(opt).cuda()
    netG.set_loss('cuda')
    netG.set_new_noise_schedule(SCHED, 'cuda')
    cfg = netG.denoise_fn.cfg
    assert cfg.variant == 'ddpm' and sum(L.with_attn for L in build_layers(cfg)) == 2 + 3 + 1
    sd = synth_state_dict(cfg, 9)
    ck = {('denoise_fn.' + k): torch.from_numpy(v) for k, v in sd.items()}
    ck.update({k: v for k, v in netG.state_dict().items() if not k.startswith('denoise_fn.')})
    netG.load_state_dict(ck, strict=True)
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(1, 6, 256, 256, generator=gen)      # attention at 16x16 (256 tokens, C=256) and mid 8x8
    t = torch.tensor([417])
    with torch.no_grad():
        ref = S.unet_forward(O.to_torch_sd(sd), cfg, x, t)
        netG.eval()
        got = netG.denoise_fn(x.cuda(), t.cuda()).cpu()
    assert (got - ref).abs().max().item() <= 1e-4
    cond = torch.rand(1, 3, 64, 64, generator=gen) * 2 - 1
    sr = netG.super_resolution(cond.cuda(), continous=False)
    assert tuple(sr.shape) == (3, 64, 64) and torch.isfinite(sr).all()
