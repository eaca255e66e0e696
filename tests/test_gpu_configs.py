"""BASELINE.json configs[2..4] at their own workloads, as far as one GPU can run them:

  configs[2]  x4 64->256, batch=64, T=20, bf16 + hipGraph loop, 1 GPU             -> as is
  configs[3]  x4 64->256, batch=512 over 8 GPUs, bf16                             -> the per-GPU slice (B=64 of 512)
  configs[4]  x8 32->256, batch=256 over 8 GPUs (+ training step)                 -> the per-GPU slice (B=32 of 256),
              conditioning image built from a 32x32 LR image by the on-device PIL-exact bicubic x8 (lr_to_sr)

At these sizes the oracle is affordable for one image only (~10 s of CPU), so each test checks that image against
the oracle and the rest of the batch through size-independent properties: finite, range, bitwise rerun, bitwise
permutation-equivariance over the batch (no operator of the path mixes batch elements).  bf16 is judged on PSNR
(north_star: within 0.01 dB of the reference), never on the 1e-3 bound."""
import math

import numpy as np
import pytest
import torch

from fastdiffsr_amd import parallel
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def full():
    from fastdiffsr_amd.engine import Engine
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    eng = Engine(cfg)
    sd = synth_state_dict(cfg, 0)
    eng.load_state_dict(sd)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    return cfg, eng, sd


def _synthetic_hr(cond):
    """cond + a fixed smooth residual: makes PSNR finite so that its DIFFERENCE is meaningful."""
    H, W = cond.shape[-2:]
    yy, xx = torch.meshgrid(torch.arange(float(H)), torch.arange(float(W)), indexing='ij')
    r = torch.stack([torch.sin(2 * math.pi * (yy / 64 + ch / 3)) * torch.cos(2 * math.pi * xx / 48) for ch in range(3)])[None]
    return (cond + 0.5 * r).clamp(-1, 1)


def _psnr_delta(out, ref, cond):
    from oracle import fdsr_oracle as O
    hr = O.tensor2img_u8(_synthetic_hr(cond)[0])
    # tensor2img clamps its argument in place (like the reference's): work on copies
    return O.psnr_u8(O.tensor2img_u8(out[0].clone()), hr) - O.psnr_u8(O.tensor2img_u8(ref[0].clone()), hr)


def _oracle_image(sd, cfg, cond1, noise1):
    from conftest import oracle_loop_image      # session-wide cache, shared with the other GPU test modules
    return oracle_loop_image(sd, cfg, cond1, noise1)


def _batch_properties(eng, c, n, graph):
    """finite / range / rerun bitwise / permutation over the batch bitwise; returns the first result."""
    B = c.shape[0]
    out = torch.empty(B, 3, *c.shape[-2:], device=c.device)
    eng.sample(c, n, graph=graph, out=out)
    first = out.clone()
    assert torch.isfinite(first).all()
    assert first.abs().max().item() <= 1.5 + 1e-6               # clamp(r)/2 + cond (diffusion.py:275-281)
    eng.sample(c, n, graph=graph, out=out)                      # graph: this one is a replay
    assert torch.equal(first, out), 'rerun is not bitwise identical'
    perm = torch.roll(torch.arange(B), B // 3 + 1)
    cp, npm = c[perm].contiguous(), n[:, perm].contiguous()
    outp = eng.sample(cp, npm)                                  # eager, other buffers: still the same bits per image
    assert torch.equal(outp, first[perm.to(first.device)]), 'images of a batch are not independent'
    return first


def test_config2_bf16_b64_hipgraph_256(full):
    """configs[2]: batch 64, 256x256, bf16 activations + bf16 MFMA, the 20-step loop replayed as a hipGraph."""
    cfg, eng, sd = full
    from conftest import plant_standard_pair
    cond, noise = synth_inputs(64, 256, 256, 20)
    plant_standard_pair(cond, noise, 41)       # the image compared with the oracle: the session's shared oracle image
    c, n = cond.cuda(), noise.cuda()
    eng.set_precision('bf16')
    try:
        s = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            out = _batch_properties(eng, c, n, graph=True)
        s.synchronize()
        i = 41
        ref = _oracle_image(sd, cfg, cond[i:i + 1], noise[:, i:i + 1])
        dps = _psnr_delta(out[i:i + 1].cpu(), ref, cond[i:i + 1])
        diff = out[i:i + 1].cpu() - ref
        d, rmse = diff.abs().max().item(), diff.pow(2).mean().sqrt().item()
        print(f'configs[2] bf16 B=64 graph: image {i} vs oracle PSNR delta {dps:+.5f} dB, max|d| {d:.3e}, rmse {rmse:.3e}')
        assert abs(dps) <= 0.01
        # bf16 is outside the 1e-3 bound by design (CPU bf16 autocast differs by rmse 1.9e-3 = 60.65 dB, max 4.9e-2;
        # isolated pixels can flip the x_0 clamp).  The PSNR delta above is taken against a synthetic HR where both images sit at
        # ~13 dB (random-init network) and cannot fail; THIS bound can: the bf16 image against the oracle's own image, data range 2
        psnr_vs_oracle = 20 * math.log10(2.0 / max(rmse, 1e-12))
        print(f'configs[2] bf16 B=64 graph: PSNR(bf16 out, oracle out) = {psnr_vs_oracle:.2f} dB')
        assert psnr_vs_oracle >= 50.0
    finally:
        eng.set_precision('f32')


def test_config3_per_gpu_slice_bf16_b64(full):
    """configs[3]: batch 512 over 8 GPUs = 64 images per rank.  Rank 5's shard of the global batch, bf16, eager."""
    cfg, eng, sd = full
    lo, hi = parallel.shard_range(512, 5, 8)
    assert hi - lo == 64
    # rank r draws its own shard (bench.py seeds per rank); the global batch is never materialised on one GPU
    from conftest import plant_standard_pair
    cond, noise = synth_inputs(hi - lo, 256, 256, 20, cond_seed=1234 + 5, noise_seed=4321 + 5)
    plant_standard_pair(cond, noise, 63)       # the image compared with the oracle: the session's shared oracle image
    c, n = cond.cuda(), noise.cuda()
    eng.set_precision('bf16')
    try:
        out = _batch_properties(eng, c, n, graph=False)
        i = 63
        ref = _oracle_image(sd, cfg, cond[i:i + 1], noise[:, i:i + 1])
        dps = _psnr_delta(out[i:i + 1].cpu(), ref, cond[i:i + 1])
        rmse = (out[i:i + 1].cpu() - ref).pow(2).mean().sqrt().item()
        psnr_vs_oracle = 20 * math.log10(2.0 / max(rmse, 1e-12))
        print(f'configs[3] slice bf16 B=64: image {i} vs oracle PSNR delta {dps:+.5f} dB, PSNR(bf16 out, oracle out) = {psnr_vs_oracle:.2f} dB')
        assert abs(dps) <= 0.01 and psnr_vs_oracle >= 50.0
    finally:
        eng.set_precision('f32')


@pytest.mark.parametrize('prec', ['f16x3', 'bf16'])
def test_config4_per_gpu_slice_x8_b32(full, prec):
    """configs[4] (sampling half): x8 32->256, batch 256 over 8 GPUs = 32 images per rank; the conditioning image
    comes from a 32x32 uint8 LR image through the on-device PIL-exact bicubic x8 (data.lr_to_sr)."""
    from fastdiffsr_amd.data import lr_to_sr
    cfg, eng, sd = full
    lo, hi = parallel.shard_range(256, 2, 8)
    assert hi - lo == 32
    g = torch.Generator().manual_seed(99)
    # smooth-ish LR content so the bicubic does real interpolation work, not noise
    base = torch.rand(hi - lo, 8, 8, 3, generator=g)
    lr = torch.nn.functional.interpolate(base.permute(0, 3, 1, 2), size=(32, 32), mode='bilinear', align_corners=False)
    lr_u8 = (lr.permute(0, 2, 3, 1) * 255).round().clamp(0, 255).to(torch.uint8).contiguous()
    c = lr_to_sr(lr_u8.cuda(), 256, 256)
    assert c.shape == (32, 3, 256, 256) and c.min().item() >= -1.0 and c.max().item() <= 1.0
    _, noise = synth_inputs(32, 256, 256, 20, noise_seed=4321 + 2)
    n = noise.cuda()
    eng.set_precision(prec)
    try:
        out = _batch_properties(eng, c, n, graph=False)
        i = 7
        cond_i = c[i:i + 1].cpu()
        ref = _oracle_image(sd, cfg, cond_i, noise[:, i:i + 1])
        d = (out[i:i + 1].cpu() - ref).abs().max().item()
        dps = _psnr_delta(out[i:i + 1].cpu(), ref, cond_i)
        print(f'configs[4] slice x8 B=32 [{prec}]: image {i} vs oracle max|d| {d:.3e}, PSNR delta {dps:+.5f} dB')
        if prec == 'f16x3':
            assert d <= 1e-3
        assert abs(dps) <= 0.01
    finally:
        eng.set_precision('f32')
