"""The training step at BASELINE configs[4]'s OWN workload (256x256; B=32 per GPU): the kernels take other paths there than
at the 32x32 goldens -- image-aligned weight-gradient slices (ns = N*k), the device-side amax / power-of-two gradient
scale, the fused column sums, larger conv tiles -- so the step is checked there too:
  (a) 256x256, B=2: loss and all 273 gradients against autograd over the oracle (reference: model/model.py:47-57,
      fastdiffsr_modules/diffusion.py:233-270), exact f32 and f16x3, and the ">= 4 GiB tensor" weight-gradient fallback
      (threshold lowered) against the default path;
  (b) 256x256, B=32, Dropout(0.2) live: bitwise rerun, and grad(B=32) == grad(images 0..15) + grad(images 16..31) with the
      very same masks (a size-independent property: the loss is a sum over samples)."""
import os

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
S = 256


def _batch(B, seed=31):
    g = torch.Generator().manual_seed(seed)
    hr = torch.rand(B, 3, S, S, generator=g) * 2 - 1
    sr = (hr + 0.1 * torch.randn(B, 3, S, S, generator=g)).clamp(-1, 1)
    nz = torch.randn(B, 3, S, S, generator=g)
    gamma = torch.rand(B, generator=g) * 0.5 + 0.4
    x_start = ((hr - sr) * 2.0).clamp(-1, 1)                      # img2res (diffusion.py:283-289)
    gg = gamma.view(-1, 1, 1, 1)
    x_noisy = gg * x_start + (1 - gg ** 2).sqrt() * nz            # q_sample (:233-241)
    return hr, sr, nz, gamma, torch.cat([sr, x_noisy], 1).contiguous()


def _live_grads(eng):
    return {k: eng.get_grad(k).copy() for k, _, live in eng.schema() if live}


@pytest.fixture(scope='module')
def oracle_step_256():
    """One oracle step at 256x256, B=2 (CPU autograd over the restatement; tens of seconds), shared by both precisions."""
    from oracle import fdsr_oracle as O
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, 0)
    hr, sr, nz, gamma, x = _batch(2)
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    l_ref, grads_ref, _ = O.train_step(O.to_torch_sd(sd), cfg, hr, sr, gamma, nz, lr=1e-4)
    return cfg, sd, (hr, sr, nz, gamma, x), float(l_ref), {k: v.numpy() for k, v in grads_ref.items()}


@pytest.mark.timeout(1200)
@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_step_at_256_matches_oracle(oracle_step_256, prec):
    from fastdiffsr_amd import _lib
    from fastdiffsr_amd.engine import Engine
    cfg, sd, (hr, sr, nz, gamma, x), l_ref, grads_ref = oracle_step_256
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    numel = hr.numel()
    loss = eng.train_grads(x.cuda(), gamma.cuda(), nz.cuda(), 'l1', 1.0 / numel)
    assert abs(loss / numel - l_ref) <= 1e-5 * abs(l_ref), (loss / numel, l_ref)
    got = _live_grads(eng)
    assert len(got) == 273 and sorted(got) == sorted(grads_ref)
    worst = (0.0, '')
    for k, ref in grads_ref.items():
        scale = max(float(np.abs(ref).max()), 1e-12)
        d = float(np.abs(got[k] - ref).max())
        worst = max(worst, (d / scale, k))
        assert d <= 1e-4 * scale, f'{k}: max|d| {d:.3e} vs max|g| {scale:.3e}'
    print(f'256x256 B=2 [{prec}]: worst gradient {worst[1]} at {worst[0]:.3e} x max|g|')
    if prec == 'f16x3':
        # the fallback for tensors >= 4 GiB (4-wave weight-gradient kernel, 64-bit offsets), forced by a 1 MiB threshold: at this
        # size every 3x3 layer takes it; same function as the default path, different kernel => fp32-grade, not bitwise
        _lib.debug_option('wgrad_big_bytes', 1 << 20)
        try:
            loss2 = eng.train_grads(x.cuda(), gamma.cuda(), nz.cuda(), 'l1', 1.0 / numel)
            got2 = _live_grads(eng)
        finally:
            _lib.debug_option('wgrad_big_bytes', 1 << 32)
        assert loss2 == loss
        differs = 0
        for k, ref in grads_ref.items():
            scale = max(float(np.abs(ref).max()), 1e-12)
            assert float(np.abs(got2[k] - ref).max()) <= 1e-4 * scale, k
            assert float(np.abs(got2[k] - got[k]).max()) <= 2e-5 * scale, k
            differs += int(not np.array_equal(got2[k], got[k]))
        assert differs > 20          # it really was another kernel


@pytest.mark.timeout(1200)
def test_b32_step_reruns_bitwise_and_splits_over_the_batch():
    """configs[4]'s per-GPU slice: B=32, 256x256, f16x3, Dropout(0.2) live."""
    from fastdiffsr_amd import _lib
    from fastdiffsr_amd.engine import Engine
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, 0)
    B = 32
    hr, sr, nz, gamma, x = _batch(B, seed=77)
    xg, gg, ng = x.cuda(), gamma.cuda(), nz.cuda()
    scale = 1.0 / hr.numel()
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision('f16x3')
    eng.set_training(True)

    def run(lo, hi):
        eng.set_seed(4242)                                   # same key, forward count back to zero: the same masks again
        _lib.debug_option('drop_image_offset', lo)
        try:
            loss = eng.train_grads(xg[lo:hi].contiguous(), gg[lo:hi].contiguous(), ng[lo:hi].contiguous(), 'l1', scale)
            return loss, _live_grads(eng)
        finally:
            _lib.debug_option('drop_image_offset', 0)
    l_all, g_all = run(0, B)
    keep = eng.dropout_mask('downs.1').cpu()
    assert abs((keep > 0).float().mean().item() - 0.8) < 5e-3      # dropout really was live
    l_again, g_again = run(0, B)
    assert l_all == l_again
    for k in g_all:
        assert np.array_equal(g_all[k], g_again[k]), k             # bitwise rerun (ordered reductions only)
    l_a, g_a = run(0, 16)
    m_a = eng.dropout_mask('downs.1').cpu()
    assert torch.equal(m_a, keep[:16])                             # the half-batch drew the first 16 images' masks
    l_b, g_b = run(16, 32)
    assert torch.equal(eng.dropout_mask('downs.1').cpu(), keep[16:])
    assert abs(l_all - (l_a + l_b)) <= 1e-5 * abs(l_all)
    worst = (0.0, '')
    for k in g_all:
        s = max(float(np.abs(g_all[k]).max()), 1e-12)
        d = float(np.abs(g_all[k] - (g_a[k] + g_b[k])).max())
        worst = max(worst, (d / s, k))
        assert d <= 1e-5 * s + 1e-12, f'{k}: {d:.3e} vs max|g| {s:.3e}'
    print(f'B=32 = 16 + 16: worst {worst[1]} at {worst[0]:.3e} x max|g|')
    eng.set_training(False)
