"""SURVEY 8f-4, second half: the optimisation step of the TESR sibling (`which_model_G == 'tesr'`: FastDiffSR's blocks and noise-level
embedding, SR3's SelfAttention placement, the Charbonnier mean as its 'l1' loss; tesr_modules/diffusion.py:85-90, :224-250) on the HIP
engine, against one step of the reference's own modules (tests/golden/tesr_train_step.npz, `oracle/make_goldens.py tesr_train`) and,
tensor by tensor, against autograd over the oracle (oracle/tesr_oracle.py, pinned to the same golden on the CPU)."""
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
           attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='tesr')
SCHED = dict(schedule='linear', n_timestep=10, linear_start=1e-4, linear_end=2e-2)


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_tesr_step_vs_reference_and_oracle(golden_dir, prec):
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O, tesr_oracle as TO
    g, tg = np.load(os.path.join(golden_dir, 'tesr_train_step.npz')), np.load(os.path.join(golden_dir, 'tesr.npz'))
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 9)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    hr, sr, nz = (torch.from_numpy(tg[k]) for k in ('hr', 'sr', 'loss_noise'))
    gamma = torch.FloatTensor(tg['gamma'])
    numel = hr.numel()
    x6 = torch.cat([sr, O.q_sample(hr, gamma.view(-1, 1, 1, 1), nz)], 1)
    # 'l1' = the Charbonnier MEAN, divided by b*c*h*w once more by DDPM.optimize_parameters: the engine's sum / numel^2
    loss = eng.train_grads(x6.cuda(), gamma.cuda(), nz.cuda(), 'charbonnier', 1.0 / (float(numel) * float(numel)))
    l_ref, grads_ref, new_ref = TO.train_step(O.to_torch_sd(sd), cfg, hr, sr, gamma, nz, lr=float(g['lr']))
    l_pix = loss / (float(numel) * float(numel))
    assert abs(l_pix - float(g['l_pix'])) <= 1e-5 * abs(float(g['l_pix'])), (l_pix, float(g['l_pix']))
    assert abs(loss / numel - float(tg['loss'])) <= 1e-5 * abs(float(tg['loss']))       # the loss golden of tesr.npz: the mean itself
    keys = [str(k) for k in g['grad_keys']]
    assert sorted(keys) == sorted(grads_ref.keys())
    worst = (0.0, '')
    for k, (s1, s2) in zip(keys, g['grad_stats']):
        got, ref = eng.get_grad(k), grads_ref[k].numpy()
        scale = max(float(np.abs(ref).max()), 1e-30)
        d = float(np.abs(got - ref).max())
        if scale >= _noise(grads_ref):
            worst = max(worst, (d / scale, k))
        assert d <= (1e-4 * scale if scale >= _noise(grads_ref) else _atol(grads_ref)), f'{k}: max|d| {d:.3e} vs max|g| {scale:.3e}'
        if scale < _noise(grads_ref):        # (rounding noise on both sides)
            continue
        g64 = got.astype(np.float64)
        assert abs((g64 * g64).sum() - s2) <= 3e-4 * s2 + 1e-40, k
    print(f'tesr [{prec}]: {len(keys)} gradients, worst {worst[1]} at {worst[0]:.3e} x max|g|')
    for k in (str(x) for x in g['full_keys']):
        ref = g['grad/' + k]
        assert np.abs(eng.get_grad(k) - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-30, k
    eng.adam_step(float(g['lr']))
    for k in (str(x) for x in g['full_keys']):
        assert np.abs(eng.get_weight(k) - g['after/' + k]).max() <= 2.1 * float(g['lr']), k


def test_tesr_facade_trains():
    """define_G(which_model_G='tesr') in the train phase: GaussianDiffusion.optimize_step brings the Charbonnier loss down on a fixed
    batch; autograd through GaussianDiffusion.forward (the reference's own loop) reaches every Parameter."""
    from fastdiffsr_amd import networks
    from fastdiffsr_amd.config import dict_to_nonedict
    torch.manual_seed(6)
    np.random.seed(6)
    opt = dict_to_nonedict({
        'phase': 'train', 'gpu_ids': [0], 'distributed': False,
        'datasets': {'train': {'l_resolution': 16, 'r_resolution': 64}},
        'model': {'which_model_G': 'tesr', 'finetune_norm': False,
                  'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 32, 'norm_groups': 32, 'channel_multiplier': [1, 2, 2, 4],
                           'attn_res': [8], 'res_blocks': 1, 'dropout': 0.2},
                  'beta_schedule': {'train': dict(SCHED), 'val': dict(SCHED)},
                  'diffusion': {'image_size': 32, 'channels': 3, 'conditional': True}}})
    netG = networks.define_G(opt).cuda()
    netG.set_loss('cuda')
    netG.set_new_noise_schedule(dict(SCHED), 'cuda')
    netG.train()
    gen = torch.Generator().manual_seed(9)
    hr = (torch.rand(4, 3, 32, 32, generator=gen) * 2 - 1).cuda()
    sr = (hr + 0.1 * torch.randn(4, 3, 32, 32, generator=gen).cuda()).clamp(-1, 1)
    n = hr.numel()
    losses = [netG.optimize_step({'HR': hr, 'SR': sr}, lr=3e-4) * n for _ in range(120)]     # back to the Charbonnier mean
    first, last = float(np.mean(losses[:8])), float(np.mean(losses[-8:]))
    print(f'tesr facade: Charbonnier mean {first:.4f} -> {last:.4f} over 120 steps')
    assert all(np.isfinite(losses)) and last < 0.92 * first
    params = [p for p in netG.parameters() if p.requires_grad]
    optG = torch.optim.Adam(params, lr=1e-4)
    optG.zero_grad()
    l_pix = netG({'HR': hr, 'SR': sr}).sum() / n
    l_pix.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    optG.step()
