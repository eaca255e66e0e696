"""SURVEY 8f-4, second half: the optimisation step of the SR3 sibling (`which_model_G == 'ddpm'`; the reference trains every variant
through DDPM.optimize_parameters, model/model.py:47-57 -> ddpm_modules/diffusion.py:279-297) on the HIP engine: forward with the
SelfAttention blocks (ddpm_modules/unet.py:99-133), L1(sum) / (b*c*h*w), backward (convolutions, GroupNorm with and without Swish,
the attention core, the integer-time embedding with Swish in front of every per-block Linear), Adam.

Checked against one step of the reference's own modules (tests/golden/sr3_train_step.npz, `oracle/make_goldens.py sr3_train`) and,
tensor by tensor, against autograd over the oracle (oracle/sr3_oracle.py, pinned to the same golden on the CPU); in exact fp32 and in
f16x3; at the golden's 32 x 32 (64- and 16-token attention) and at 64 x 48 (192 and 48 tokens: off the 32-wide MFMA tiles);
bitwise reruns; the facade (`define_G` with which_model_G 'ddpm', GaussianDiffusion.optimize_step / autograd through forward)."""
import os

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig
from fastdiffsr_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu
# the level-0 blocks of this test network have ONE channel per GroupNorm group (32 channels, 32 groups): a per-channel shift in front of
# such a GroupNorm cancels exactly, so the gradients of those blocks' per-block Linears are rounding noise (1e-10 against 1e-5
# elsewhere) on both sides -- an absolute floor beside the relative bound


_TYPICAL = {}


def _typical(grads_ref):
    """The median over the step's tensors of max |g|: the scale a real gradient of this step has (computed once per set of gradients)."""
    key = id(grads_ref)
    if key not in _TYPICAL:
        _TYPICAL[key] = (grads_ref, float(np.median([float(v.abs().max()) for v in grads_ref.values()])))   # (the dict is kept alive with its value)
    return _TYPICAL[key][1]


def _noise(grads_ref):
    """Below this a tensor's gradient is rounding noise on both sides.  This test network has ONE channel per GroupNorm group at its
    first and last level (32 channels, 32 groups): a per-channel shift in front of such a GroupNorm cancels exactly, so the conv bias
    and the per-block Linear in front of it have gradients six orders below every other tensor's (measured on the CPU oracle:
    3e-10 .. 2e-9 against a median of 1e-3 in the SR3 step, 1e-13 against 1.3e-7 in the TESR step)."""
    return 1e-4 * _typical(grads_ref)


def _atol(grads_ref):
    return 1e-5 * _typical(grads_ref)


CFG = dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
           attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='ddpm')
SCHED = dict(schedule='linear', n_timestep=12, linear_start=1e-4, linear_end=2e-2)


def _x6(tab, hr, sr, t, nz):
    from oracle import sr3_oracle as S
    return torch.cat([sr, S.q_sample(tab, hr, t, nz)], 1)


@pytest.fixture(scope='module', params=['f32', 'f16x3'])
def stepped(golden_dir, request):
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O, sr3_oracle as S
    g = np.load(os.path.join(golden_dir, 'sr3_train_step.npz'))
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 5)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(request.param)
    tab = O.schedule_tables(SCHED)
    hr, sr, nz = (torch.from_numpy(g[k]) for k in ('hr', 'sr', 'noise'))
    t = torch.from_numpy(g['t']).long()
    b, c, h, w = hr.shape
    x = _x6(tab, hr, sr, t, nz)
    loss = eng.train_grads(x.cuda(), t.float().cuda(), nz.cuda(), 'l1', 1.0 / (b * c * h * w))
    l_ref, grads_ref, new_ref = S.train_step(O.to_torch_sd(sd), cfg, tab, hr, sr, t, nz, lr=float(g['lr']))
    return cfg, sd, eng, loss, (b * c * h * w), l_ref, grads_ref, new_ref, g, request.param


def test_sr3_loss_and_all_gradients(stepped):
    cfg, sd, eng, loss, numel, l_ref, grads_ref, new_ref, g, prec = stepped
    l_pix = loss / numel
    assert abs(l_pix - float(g['l_pix'])) <= 1e-5 * abs(float(g['l_pix'])), (l_pix, float(g['l_pix']))
    keys = [str(k) for k in g['grad_keys']]
    assert sorted(keys) == sorted(grads_ref.keys())
    worst = (0.0, '')
    for k, (s1, s2) in zip(keys, g['grad_stats']):
        got, ref = eng.get_grad(k), grads_ref[k].numpy()
        assert got.shape == ref.shape, k
        scale = max(float(np.abs(ref).max()), 1e-12)
        d = float(np.abs(got - ref).max())
        if scale >= _noise(grads_ref):
            worst = max(worst, (d / scale, k))
        assert d <= (1e-4 * scale if scale >= _noise(grads_ref) else _atol(grads_ref)), f'{k}: max|d| {d:.3e} vs max|g| {scale:.3e}'
        if scale < _noise(grads_ref):        # (a gradient that is rounding noise on both sides: nothing to compare beyond the bound above)
            continue
        g64 = got.astype(np.float64)
        assert abs(g64.sum() - s1) <= 3e-4 * max(np.sqrt(s2), 1e-12) + 1e-9, k
        assert abs((g64 * g64).sum() - s2) <= 3e-4 * s2 + 1e-18, k
    print(f'sr3 [{prec}]: {len(keys)} gradients, worst {worst[1]} at {worst[0]:.3e} x max|g|')
    for k in (str(x) for x in g['full_keys']):                  # the reference's own tensors (attention qkv / out / norm among them)
        ref = g['grad/' + k]
        assert np.abs(eng.get_grad(k) - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k


def test_sr3_adam_update_and_rerun(stepped):
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O
    cfg, sd, eng, loss, numel, l_ref, grads_ref, new_ref, g, prec = stepped
    lr = float(g['lr'])
    # a second engine repeats the step bitwise (ordered reductions, no float atomics -- the attention backward included)
    e2 = Engine(cfg)
    e2.load_state_dict(sd)
    e2.set_precision(prec)
    tab = O.schedule_tables(SCHED)
    hr, sr, nz = (torch.from_numpy(g[k]) for k in ('hr', 'sr', 'noise'))
    t = torch.from_numpy(g['t']).long()
    x = _x6(tab, hr, sr, t, nz)
    l2 = e2.train_grads(x.cuda(), t.float().cuda(), nz.cuda(), 'l1', 1.0 / numel)
    assert l2 == loss
    for k in grads_ref:
        assert np.array_equal(e2.get_grad(k), eng.get_grad(k)), k
    eng.adam_step(lr)
    for k in (str(x) for x in g['full_keys']):
        ref_g, aft, ref_aft = g['grad/' + k], eng.get_weight(k), g['after/' + k]
        if np.abs(ref_g).max() < _noise(grads_ref):
            assert np.abs(aft - ref_aft).max() <= 2.1 * lr, k
            continue
        mask = np.abs(ref_g) > 1e-3 * np.abs(ref_g).max()
        assert np.abs(aft - ref_aft)[mask].max() <= 2e-7, k
        assert np.abs(aft - ref_aft).max() <= 2.1 * lr, k
    for k, ref in new_ref.items():
        if k not in grads_ref:
            continue
        gk = grads_ref[k].numpy()
        if np.abs(gk).max() < _noise(grads_ref):       # Adam turns a noise gradient into +-lr steps of noise sign
            continue
        mask = np.abs(gk) > 1e-3 * np.abs(gk).max()
        if mask.any():
            assert np.abs(eng.get_weight(k) - ref.numpy())[mask].max() <= 3e-7, k


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_sr3_gradients_off_the_tile_grid(prec):
    """64 x 48 input: the attention levels see 16 x 12 = 192 and 8 x 6 = 48 tokens (not multiples of the 32-wide MFMA tiles), three
    images, per-sample times incl. t = 0; l2 loss."""
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O, sr3_oracle as S
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 5)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    tab = O.schedule_tables(SCHED)
    gen = torch.Generator().manual_seed(31)
    hr = torch.rand(3, 3, 64, 48, generator=gen) * 2 - 1
    sr = (hr + 0.1 * torch.randn(3, 3, 64, 48, generator=gen)).clamp(-1, 1)
    nz = torch.randn(3, 3, 64, 48, generator=gen)
    t = torch.tensor([0, 5, 11])
    numel = hr.numel()
    loss = eng.train_grads(_x6(tab, hr, sr, t, nz).cuda(), t.float().cuda(), nz.cuda(), 'l2', 1.0 / numel)
    l_ref, grads_ref, _ = S.train_step(O.to_torch_sd(sd), cfg, tab, hr, sr, t, nz, lr=1e-4, loss_type='l2')
    assert abs(loss / numel - l_ref.item()) <= 1e-5 * abs(l_ref.item())
    worst = (0.0, '')
    for k, ref in grads_ref.items():
        got, ref = eng.get_grad(k), ref.numpy()
        scale = max(float(np.abs(ref).max()), 1e-12)
        d = float(np.abs(got - ref).max())
        if scale >= _noise(grads_ref):
            worst = max(worst, (d / scale, k))
        assert d <= (1e-4 * scale if scale >= _noise(grads_ref) else _atol(grads_ref)), f'{k}: max|d| {d:.3e} vs max|g| {scale:.3e}'
    print(f'sr3 64x48 [{prec}]: worst gradient {worst[1]} at {worst[0]:.3e} x max|g|')


def test_sr3_facade_trains():
    """define_G(which_model_G='ddpm') in the train phase: orthogonal init, GaussianDiffusion.optimize_step (all-device step) brings the
    loss down on a fixed batch, and autograd through GaussianDiffusion.forward hands the engine's gradients to the Parameters (the
    reference's own `l_pix.backward(); optG.step()` loop, model/model.py:49-56)."""
    from fastdiffsr_amd import networks
    from fastdiffsr_amd.config import dict_to_nonedict
    torch.manual_seed(5)
    np.random.seed(5)
    opt = dict_to_nonedict({
        'phase': 'train', 'gpu_ids': [0], 'distributed': False,
        'datasets': {'train': {'l_resolution': 16, 'r_resolution': 64}},
        'model': {'which_model_G': 'ddpm', 'finetune_norm': False,
                  'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 32, 'norm_groups': 32, 'channel_multiplier': [1, 2, 2, 4],
                           'attn_res': [8], 'res_blocks': 1, 'dropout': 0.2},
                  'beta_schedule': {'train': dict(SCHED), 'val': dict(SCHED)},
                  'diffusion': {'image_size': 32, 'channels': 3, 'conditional': True}}})
    netG = networks.define_G(opt).cuda()
    netG.set_loss('cuda')
    netG.set_new_noise_schedule(dict(SCHED), 'cuda')
    netG.train()
    gen = torch.Generator().manual_seed(8)
    hr = (torch.rand(4, 3, 32, 32, generator=gen) * 2 - 1).cuda()
    sr = (hr + 0.1 * torch.randn(4, 3, 32, 32, generator=gen).cuda()).clamp(-1, 1)
    losses = [netG.optimize_step({'HR': hr, 'SR': sr}, lr=3e-4) for _ in range(120)]
    first, last = float(np.mean(losses[:8])), float(np.mean(losses[-8:]))
    print(f'sr3 facade: l_pix {first:.4f} -> {last:.4f} over 120 steps')
    assert all(np.isfinite(losses)) and last < 0.92 * first
    # the reference's loop: autograd through forward, torch's own Adam on the Parameters
    params = [p for p in netG.parameters() if p.requires_grad]
    optG = torch.optim.Adam(params, lr=1e-4)
    optG.zero_grad()
    l_pix = netG({'HR': hr, 'SR': sr}).sum() / hr.numel()
    l_pix.backward()
    with_grad = [p for p in params if p.grad is not None]
    assert len(with_grad) == len(params) and all(torch.isfinite(p.grad).all() for p in with_grad)
    optG.step()
    assert torch.isfinite(netG({'HR': hr, 'SR': sr}).detach()).all()
