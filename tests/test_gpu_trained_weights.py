"""The sampling path on TRAINED-LIKE weights (verdict r5, missing #3 / weak #1).  Real checkpoints are external to the reference
repository (README.md:7, config/sr_fastdiffsr_test_64_256.json:12), so every other GPU number here is taken on closed-form synthetic
weights (synth.py: kaiming-uniform scaled).  This module manufactures the weight / activation distributions a trained network has
with the repository's own training path: `define_G` in the train phase (orthogonal init, model/networks.py:113-115), then a few
hundred optimisation steps of `DDPM.optimize_parameters` (model/model.py:47-57: q_sample + L1 + backward + Adam lr 1e-4) on synthetic
HR / bicubic-SR pairs, Dropout(0.2) live.  On those weights:

  (a) the 20-step loop at 256 x 256 in f16x3 with the range guard set to RAISE (no silent exact-fp32 re-run) against the oracle on
      the SAME trained weights: |delta| <= 1e-3 per pixel (north_star), and no saturation fallback was taken;
  (b) the same in bf16: PSNR(bf16 image, oracle image) and the PSNR difference against the HR image, reported and bounded.  MEASURED
      (round 6): 47.5 dB between the images (rmse 8.4e-3: three times the synthetic weights' 2.6e-3 / 57.6 dB) and +0.19 dB against
      HR at 25.4 dB -- on trained-like weights the bf16 mode is NOT within north_star's 0.01 dB, which the random-init checks
      (13 dB against any HR) could not show; the fp32-grade f16x3 mode is (|delta| 6e-7).  PyTorch-CPU under bf16 autocast -- the
      oracle's own arithmetic with bf16 convolutions -- is run beside it to tell the storage format from the implementation.
  (c) the same in the f16 mode (FDSR_PREC_F16, built in round 6 because of (b)): PSNR(f16 image, oracle image) >= 70 dB and the PSNR
      difference against HR within north_star's 0.01 dB.  MEASURED: 82.0 dB (rmse 1.6e-4, max|d| 5.3e-4) and +0.0009 dB.
"""
import math

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_inputs

pytestmark = pytest.mark.gpu
STEPS, BATCH, SIZE = 240, 8, 128          # the network is fully convolutional: trained at 128 x 128, sampled at 256 x 256


def _pairs(n, size, seed):
    """Smooth-plus-texture HR images in [-1, 1] and their x4 bicubic round trip (the SR conditioning image), as tensors."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(size).float(), torch.arange(size).float(), indexing='ij')
    hr = []
    for i in range(n):
        f = 4.0 + 9.0 * torch.rand(6, generator=g)
        img = torch.stack([torch.sin(xx / f[c] + c) * torch.cos(yy / f[3 + c] + i) for c in range(3)])
        hr.append((0.75 * img + 0.08 * torch.randn(3, size, size, generator=g)).clamp(-1, 1))
    hr = torch.stack(hr)
    lr = torch.nn.functional.interpolate(hr, scale_factor=0.25, mode='bicubic', align_corners=False, antialias=True)
    sr = torch.nn.functional.interpolate(lr, size=(size, size), mode='bicubic', align_corners=False).clamp(-1, 1)
    return hr, sr


def make_trained():
    """(cfg, trained state dict as numpy, loss history): define_G's orthogonal init + STEPS optimisation steps on the engine."""
    from fastdiffsr_amd import networks
    from fastdiffsr_amd.config import dict_to_nonedict
    torch.manual_seed(11)
    np.random.seed(11)
    sched = dict(FASTDIFFSR_SCHEDULE_VAL)
    opt = dict_to_nonedict({
        'phase': 'train', 'gpu_ids': [0], 'distributed': False,
        'datasets': {'train': {'l_resolution': 64, 'r_resolution': 256}},
        'model': {'which_model_G': 'fastdiffsr', 'finetune_norm': False,
                  'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'norm_groups': 32, 'channel_multiplier': [1, 2, 4, 4],
                           'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
                  'beta_schedule': {'train': sched, 'val': sched},
                  'diffusion': {'image_size': 256, 'channels': 3, 'conditional': True}}})
    netG = networks.define_G(opt).cuda()
    netG.set_loss('cuda')
    netG.set_new_noise_schedule(sched, 'cuda')
    netG.precision = 'f16x3'
    netG.train()
    hr, sr = _pairs(64, SIZE, 5)
    hr, sr = hr.cuda(), sr.cuda()
    losses = []
    for k in range(STEPS):
        idx = torch.randint(0, hr.shape[0], (BATCH,))
        losses.append(netG.optimize_step({'HR': hr[idx], 'SR': sr[idx]}, lr=1e-4))
    netG.eval()
    unet = netG.denoise_fn
    unet.pull_weights()                                    # the engine's master copy -> the module's Parameters
    sd = {k: v.detach().cpu().numpy().copy() for k, v in unet.state_dict().items()}
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    del netG
    torch.cuda.empty_cache()
    return cfg, sd, losses


@pytest.fixture(scope='module')
def trained():
    return make_trained()


def test_training_moved_the_weights_and_the_loss(trained):
    cfg, sd, losses = trained
    first, last = float(np.mean(losses[:20])), float(np.mean(losses[-20:]))
    print(f'trained-like weights: l_pix {first:.4f} -> {last:.4f} over {STEPS} steps (B={BATCH}, {SIZE}x{SIZE})')
    assert all(np.isfinite(losses)) and last < 0.8 * first
    w = sd['downs.1.res_block.block2.block.3.weight']
    # orthogonal rows have norm 1; 240 Adam steps of 1e-4 move every element by up to ~0.02 against |w| ~ 0.04
    assert abs(float(np.linalg.norm(w.reshape(w.shape[0], -1), axis=1).mean()) - 1.0) > 1e-3


@pytest.mark.timeout(1200)
def test_trained_weights_f16x3_raise_and_bf16_vs_oracle(trained):
    from fastdiffsr_amd.engine import Engine
    from conftest import oracle_loop_image
    cfg, sd, _ = trained
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    # an image-like conditioning input (what the val loop feeds), the parity runs' noise planes
    _, sr = _pairs(1, 256, 77)
    _, noise = synth_inputs(1, 256, 256, 20)
    ref = oracle_loop_image(sd, cfg, sr, noise)
    before = Engine.saturation_fallbacks
    eng.on_saturation = 'raise'                        # a raw conv input beyond the f16 range is an ERROR here, not a re-run
    eng.set_precision('f16x3')
    out = eng.sample(sr.cuda(), noise.cuda()).cpu()
    g = eng.sample(sr.cuda(), noise.cuda(), graph=True).cpu()
    d = (out - ref).abs().max().item()
    print(f'trained-like weights, f16x3 (on_saturation=raise), 256x256 20 steps: max|HIP - oracle| = {d:.3e}; graph == eager: {torch.equal(g, out)}')
    assert d <= 1e-3 and torch.equal(g, out)
    assert Engine.saturation_fallbacks == before
    # the exact-fp32 kernels on the same weights, for the record
    eng.set_precision('f32')
    d32 = (eng.sample(sr.cuda(), noise.cuda()).cpu() - ref).abs().max().item()
    # bf16 storage + bf16 MFMA
    eng.set_precision('bf16')
    ob = eng.sample(sr.cuda(), noise.cuda()).cpu()
    rmse = (ob - ref).pow(2).mean().sqrt().item()
    psnr_vs_oracle = 20 * math.log10(2.0 / max(rmse, 1e-12))
    from oracle import fdsr_oracle as O
    hr, _ = _pairs(1, 256, 77)
    u8 = lambda t: O.tensor2img_u8(t[0].clone())
    dps = O.psnr_u8(u8(ob), u8(hr)) - O.psnr_u8(u8(ref), u8(hr))
    print(f'trained-like weights: exact f32 max|d| = {d32:.3e}; bf16 PSNR(out, oracle out) = {psnr_vs_oracle:.2f} dB (rmse {rmse:.3e}, '
          f'max|d| {(ob - ref).abs().max().item():.3e}), PSNR(out, HR) - PSNR(oracle, HR) = {dps:+.5f} dB at {O.psnr_u8(u8(ref), u8(hr)):.2f} dB')
    assert d32 <= 1e-3
    # f16 storage + one f16 MFMA per product (FDSR_PREC_F16): the bf16 mode's speed with 11 mantissa bits
    eng.set_precision('f16')
    oh = eng.sample(sr.cuda(), noise.cuda()).cpu()
    rm_h = (oh - ref).pow(2).mean().sqrt().item()
    dps_h = O.psnr_u8(u8(oh), u8(hr)) - O.psnr_u8(u8(ref), u8(hr))
    print(f'trained-like weights: f16 PSNR(out, oracle out) = {20 * math.log10(2.0 / max(rm_h, 1e-12)):.2f} dB (rmse {rm_h:.3e}, max|d| '
          f'{(oh - ref).abs().max().item():.3e}), PSNR(out, HR) - PSNR(oracle, HR) = {dps_h:+.5f} dB')
    assert 20 * math.log10(2.0 / max(rm_h, 1e-12)) >= 70.0 and abs(dps_h) <= 0.01      # north_star's bound, on trained-like weights (measured 82.0 dB, +0.0009 dB)
    # the same loop on the CPU with PyTorch's bf16 autocast (convolutions and linears in bf16, everything else fp32): how far does
    # the number format alone move the image?
    with torch.autocast('cpu', dtype=torch.bfloat16):
        ref_bf16 = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), sr, noise).float()
    rm_a = (ref_bf16 - ref).pow(2).mean().sqrt().item()
    dps_a = O.psnr_u8(u8(ref_bf16), u8(hr)) - O.psnr_u8(u8(ref), u8(hr))
    print(f'trained-like weights: PyTorch-CPU bf16 autocast of the oracle vs the fp32 oracle: PSNR {20 * math.log10(2.0 / max(rm_a, 1e-12)):.2f} dB '
          f'(rmse {rm_a:.3e}), PSNR delta against HR {dps_a:+.5f} dB')
    assert psnr_vs_oracle >= 45.0 and abs(dps) <= 0.5
