"""SURVEY 8f-4, second half, third sibling: the optimisation step of the GDP denoiser (`which_model_G == 'gdp'`; the reference trains
every variant through DDPM.optimize_parameters, model/model.py:47-57 -> gdp_modules/diffusion.py:277-299: summed MSE between
UNet(cat[q_sample(HR, t), SR], t) and HR) on the HIP engine: forward through the guided-diffusion UNet (gdp_modules/unet.py:276-439,
:530-800), backward through the scale-shift GroupNorms (the gradient of (scale, shift) feeds every ResBlock's Linear and the time
MLP), the average-pooled down ResBlocks and nearest-upsampled up ResBlocks, the heads of 64 channels of QKVAttentionLegacy, Adam.

Checked against one step of the reference's own modules (tests/golden/gdp_train_step.npz, `oracle/make_goldens.py gdp_train`) and,
tensor by tensor, against autograd over the oracle (oracle/gdp_oracle.py, pinned to the same golden on the CPU); in exact fp32 and in
f16x3; at the golden's 32 x 32 and at 48 x 32 with three images (attention over 384 and 96 tokens, t = 0 included); bitwise reruns;
the facade (`define_G` with which_model_G 'gdp': GaussianDiffusion.optimize_step / autograd through forward)."""
import os

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig
from fastdiffsr_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu

CFG = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 2), attn_res=(2, 4), res_blocks=1,
           dropout=0.1, image_size=32, variant='gdp')
SCHED = dict(schedule='linear', n_timestep=8, linear_start=1e-4, linear_end=2e-2)


_TYPICAL = {}


def _typical(grads_ref):
    """The median over the step's tensors of max |g|: the scale a real gradient of this step has (computed once per set of gradients)."""
    key = id(grads_ref)
    if key not in _TYPICAL:
        _TYPICAL[key] = (grads_ref, float(np.median([float(v.abs().max()) for v in grads_ref.values()])))   # (the dict is kept alive with its value)
    return _TYPICAL[key][1]


def _noise(grads_ref):
    """Below this a tensor's gradient is rounding noise on both sides (a bias in front of a GroupNorm whose groups hold few
    channels cancels almost exactly): an absolute floor replaces the relative bound there, as in tests/test_gpu_sr3_train.py."""
    return 1e-4 * _typical(grads_ref)


def _atol(grads_ref):
    return 1e-5 * _typical(grads_ref)


def _x6(tab, hr, sr, t, nz):
    a = torch.from_numpy(np.asarray(tab['sqrt_alphas_cumprod'], dtype=np.float32))[t].view(-1, 1, 1, 1)
    s = torch.from_numpy(np.asarray(tab['sqrt_one_minus_alphas_cumprod'], dtype=np.float32))[t].view(-1, 1, 1, 1)
    return torch.cat([a * hr + s * nz, sr], 1)              # gdp_modules/diffusion.py:290: cat([x_t, x_sr])


def _compare_all(eng, grads_ref, tag):
    worst = (0.0, '')
    for k, ref in grads_ref.items():
        got, ref = eng.get_grad(k), ref.numpy()
        assert got.shape == ref.shape, k
        scale = max(float(np.abs(ref).max()), 1e-12)
        d = float(np.abs(got - ref).max())
        if scale >= _noise(grads_ref):
            worst = max(worst, (d / scale, k))
        assert d <= (1e-4 * scale if scale >= _noise(grads_ref) else _atol(grads_ref)), f'{k}: max|d| {d:.3e} vs max|g| {scale:.3e} [{tag}]'
    print(f'gdp {tag}: {len(grads_ref)} gradients, worst {worst[1]} at {worst[0]:.3e} x max|g|')


_ORACLE_STEP = {}


@pytest.fixture(scope='module', params=['f32', 'f16x3'])
def stepped(golden_dir, request):
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O, gdp_oracle as GO
    g = np.load(os.path.join(golden_dir, 'gdp_train_step.npz'))
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 13)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(request.param)
    tab = O.schedule_tables(SCHED)
    hr, sr, nz = (torch.from_numpy(g[k]) for k in ('hr', 'sr', 'noise'))
    t = torch.from_numpy(g['t']).long()
    b, c, h, w = hr.shape
    loss = eng.train_grads(_x6(tab, hr, sr, t, nz).cuda(), t.float().cuda(), hr.cuda(), 'l2', 1.0 / (b * c * h * w))
    if 'golden' not in _ORACLE_STEP:                            # one autograd step of the oracle serves both precisions
        _ORACLE_STEP['golden'] = GO.train_step(O.to_torch_sd(sd), cfg, tab, hr, sr, t, nz, lr=float(g['lr']))
    l_ref, grads_ref, new_ref = _ORACLE_STEP['golden']
    return cfg, sd, eng, loss, (b * c * h * w), l_ref, grads_ref, new_ref, g, request.param


def test_gdp_loss_and_all_gradients(stepped):
    cfg, sd, eng, loss, numel, l_ref, grads_ref, new_ref, g, prec = stepped
    l_pix = loss / numel
    assert abs(l_pix - float(g['l_pix'])) <= 1e-5 * abs(float(g['l_pix'])), (l_pix, float(g['l_pix']))
    keys = [str(k) for k in g['grad_keys']]
    assert sorted(keys) == sorted(grads_ref.keys())
    _compare_all(eng, grads_ref, prec)
    for k, (s1, s2) in zip(keys, g['grad_stats']):              # the reference's own (sum, sum of squares) of every gradient
        if float(grads_ref[k].abs().max()) < _noise(grads_ref):
            continue
        g64 = eng.get_grad(k).astype(np.float64)
        assert abs(g64.sum() - s1) <= 3e-4 * max(np.sqrt(s2), 1e-12) + 1e-9, k
        assert abs((g64 * g64).sum() - s2) <= 3e-4 * s2 + 1e-18, k
    for k in (str(x) for x in g['full_keys']):                  # the reference's own tensors
        ref = g['grad/' + k]
        assert np.abs(eng.get_grad(k) - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k


def test_gdp_adam_update_and_rerun(stepped):
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O
    cfg, sd, eng, loss, numel, l_ref, grads_ref, new_ref, g, prec = stepped
    lr = float(g['lr'])
    e2 = Engine(cfg)                                            # a second engine repeats the step bitwise (ordered reductions, no float atomics)
    e2.load_state_dict(sd)
    e2.set_precision(prec)
    tab = O.schedule_tables(SCHED)
    hr, sr, nz = (torch.from_numpy(g[k]) for k in ('hr', 'sr', 'noise'))
    t = torch.from_numpy(g['t']).long()
    l2 = e2.train_grads(_x6(tab, hr, sr, t, nz).cuda(), t.float().cuda(), hr.cuda(), 'l2', 1.0 / numel)
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
        gk = grads_ref[k].numpy()
        if np.abs(gk).max() < _noise(grads_ref):                # Adam turns a noise gradient into +-lr steps of noise sign
            continue
        mask = np.abs(gk) > 1e-3 * np.abs(gk).max()
        if mask.any():
            assert np.abs(eng.get_weight(k) - ref.numpy())[mask].max() <= 3e-7, k


@pytest.mark.parametrize('prec', ['f16x3'])
def test_gdp_gradients_off_the_tile_grid(prec):
    """48 x 32 input, three images, per-sample times incl. t = 0: the attention levels see 24 x 16 = 384 and 12 x 8 = 96 tokens, the
    pooled / upsampled ResBlocks maps that are not square.  (f16x3 only: the exact-fp32 convolution kernels see these shapes in
    tests/test_gpu_train.py and in the two width tests below; the GDP-specific kernels are precision-independent.)"""
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O, gdp_oracle as GO
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 13)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    tab = O.schedule_tables(SCHED)
    gen = torch.Generator().manual_seed(37)
    hr = torch.rand(3, 3, 48, 32, generator=gen) * 2 - 1
    sr = (hr + 0.1 * torch.randn(3, 3, 48, 32, generator=gen)).clamp(-1, 1)
    nz = torch.randn(3, 3, 48, 32, generator=gen)
    t = torch.tensor([0, 4, 7])
    numel = hr.numel()
    loss = eng.train_grads(_x6(tab, hr, sr, t, nz).cuda(), t.float().cuda(), hr.cuda(), 'l2', 1.0 / numel)
    l_ref, grads_ref, _ = GO.train_step(O.to_torch_sd(sd), cfg, tab, hr, sr, t, nz, lr=1e-4)
    assert abs(loss / numel - l_ref.item()) <= 1e-5 * abs(l_ref.item())
    _compare_all(eng, grads_ref, f'48x32 {prec}')


@pytest.mark.parametrize('mc,mults,attn', [(64, (1, 4), (1, 2)), (128, (1, 2), (2,))])
def test_gdp_other_widths_and_head_counts(mc, mults, attn):
    """model_channels 64 with mults (1, 4): a level-0 attention with ONE head (64 channels) and FOUR heads over 256 channels below it,
    two ResBlocks per level.  model_channels 128 (the reference's default width): the time MLP is 512 wide -- wider than the 256
    threads of the embedding-backward workgroup -- and every scale-shift Linear reads 512 columns."""
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O, gdp_oracle as GO
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=mc, norm_groups=32, channel_mults=mults, attn_res=attn,
                     res_blocks=2 if mc == 64 else 1, dropout=0.0, image_size=16, variant='gdp')
    sd = synth_state_dict(cfg, 21)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision('f32')
    tab = O.schedule_tables(SCHED)
    gen = torch.Generator().manual_seed(41)
    hr = torch.rand(2, 3, 16, 16, generator=gen) * 2 - 1
    sr = (hr + 0.1 * torch.randn(2, 3, 16, 16, generator=gen)).clamp(-1, 1)
    nz = torch.randn(2, 3, 16, 16, generator=gen)
    t = torch.tensor([2, 6])
    numel = hr.numel()
    loss = eng.train_grads(_x6(tab, hr, sr, t, nz).cuda(), t.float().cuda(), hr.cuda(), 'l2', 1.0 / numel)
    l_ref, grads_ref, _ = GO.train_step(O.to_torch_sd(sd), cfg, tab, hr, sr, t, nz, lr=1e-4)
    assert abs(loss / numel - l_ref.item()) <= 1e-5 * abs(l_ref.item())
    _compare_all(eng, grads_ref, f'mc {mc} mults {mults} f32')


def test_gdp_facade_trains():
    """define_G(which_model_G='gdp') in the train phase: orthogonal init over the zero-initialised output convs (the reference's
    init_weights re-draws them too), GaussianDiffusion.optimize_step (all-device step, dropout live) brings the loss down on a fixed
    batch, and autograd through GaussianDiffusion.forward hands the engine's gradients to the Parameters (the reference's own
    `l_pix.backward(); optG.step()` loop, model/model.py:49-56)."""
    from fastdiffsr_amd import networks
    from fastdiffsr_amd.config import dict_to_nonedict
    torch.manual_seed(5)
    np.random.seed(5)
    opt = dict_to_nonedict({
        'phase': 'train', 'gpu_ids': [0], 'distributed': False,
        'datasets': {'train': {'l_resolution': 8, 'r_resolution': 32}},
        'model': {'which_model_G': 'gdp', 'finetune_norm': False,
                  'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'norm_groups': 32, 'channel_multiplier': [1, 2, 2],
                           'attn_res': [16], 'res_blocks': 1, 'dropout': 0.1},
                  'beta_schedule': {'train': dict(SCHED), 'val': dict(SCHED)},
                  'diffusion': {'image_size': 32, 'channels': 3, 'conditional': True}}})
    netG = networks.define_G(opt).cuda()
    netG.set_loss('cuda')
    netG.set_new_noise_schedule(dict(SCHED), 'cuda')
    netG.train()
    gen = torch.Generator().manual_seed(8)
    hr = (torch.rand(4, 3, 32, 32, generator=gen) * 2 - 1).cuda()
    sr = (hr + 0.1 * torch.randn(4, 3, 32, 32, generator=gen).cuda()).clamp(-1, 1)
    losses = [netG.optimize_step({'HR': hr, 'SR': sr}, lr=1e-4) for _ in range(60)]
    first, last = float(np.mean(losses[:6])), float(np.mean(losses[-6:]))
    print(f'gdp facade: l_pix {first:.4f} -> {last:.4f} over 60 steps')
    assert all(np.isfinite(losses)) and last < 0.7 * first
    params = [p for p in netG.parameters() if p.requires_grad]
    optG = torch.optim.Adam(params, lr=1e-4)
    optG.zero_grad()
    l_pix = netG({'HR': hr, 'SR': sr, 'LR': sr}).sum() / hr.numel()
    l_pix.backward()
    with_grad = [p for p in params if p.grad is not None]
    assert len(with_grad) == len(params) and all(torch.isfinite(p.grad).all() for p in with_grad)
    optG.step()
    assert torch.isfinite(netG({'HR': hr, 'SR': sr, 'LR': sr}).detach()).all()
