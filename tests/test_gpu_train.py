"""One optimisation step on the device against one step of the reference itself
(tests/golden/train_step.npz, made by oracle/make_goldens.py from model/model.py:47-57 with dropout off) and,
tensor by tensor, against autograd over the oracle: loss, all 273 gradients, the Adam update."""
import os

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu


def _inputs(golden_dir):
    tl = np.load(os.path.join(golden_dir, 'train_loss.npz'))
    hr, sr, nz = (torch.from_numpy(tl[k]) for k in ('hr', 'sr', 'noise'))
    gamma = torch.FloatTensor(tl['gamma'])
    return hr, sr, nz, gamma


def _x_noisy(hr, sr, nz, gamma):
    """img2res + q_sample in torch, as the facade forms them (diffusion.py:233-241, :283-289)."""
    x_start = ((hr - sr) * 2.0).clamp(-1, 1)
    g = gamma.view(-1, 1, 1, 1)
    return g * x_start + (1 - g ** 2).sqrt() * nz


@pytest.fixture(scope='module', params=['f32', 'f16x3'])
def stepped(golden_dir, request):
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, 0)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(request.param)
    hr, sr, nz, gamma = _inputs(golden_dir)
    b, c, h, w = hr.shape
    x = torch.cat([sr, _x_noisy(hr, sr, nz, gamma)], 1)
    loss = eng.train_grads(x.cuda(), gamma.cuda(), nz.cuda(), 'l1', 1.0 / (b * c * h * w))
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    l_ref, grads_ref, new_ref = O.train_step(O.to_torch_sd(sd), cfg, hr, sr, gamma, nz, lr=1e-4)
    return cfg, sd, eng, loss, (b * c * h * w), l_ref, grads_ref, new_ref


def test_loss_matches_reference_step(stepped, golden_dir):
    cfg, sd, eng, loss, numel, l_ref, grads_ref, new_ref = stepped
    g = np.load(os.path.join(golden_dir, 'train_step.npz'))
    l_pix = loss / numel
    assert abs(l_pix - float(g['l_pix'])) <= 1e-5 * abs(float(g['l_pix'])), (l_pix, float(g['l_pix']))
    assert abs(l_pix - l_ref.item()) <= 1e-5 * abs(l_ref.item())


def test_all_gradients(stepped, golden_dir):
    """Every executed tensor: max |g_hip - g_autograd| <= 1e-4 * max|g|; and the per-tensor (sum, sum of squares)
    the reference's own backward produced.  The 44 never-executed tensors have no gradient."""
    from fastdiffsr_amd import _lib
    cfg, sd, eng, loss, numel, l_ref, grads_ref, new_ref = stepped
    g = np.load(os.path.join(golden_dir, 'train_step.npz'))
    keys = [str(k) for k in g['grad_keys']]
    assert len(keys) == 273 and sorted(keys) == sorted(grads_ref.keys())
    worst = (0.0, '')
    for k, (s1, s2) in zip(keys, g['grad_stats']):
        got = eng.get_grad(k)
        ref = grads_ref[k].numpy()
        assert got.shape == ref.shape, k
        scale = max(float(np.abs(ref).max()), 1e-12)
        d = float(np.abs(got - ref).max())
        worst = max(worst, (d / scale, k))
        assert d <= 1e-4 * scale, f'{k}: max|d| {d:.3e} vs max|g| {scale:.3e}'
        g64 = got.astype(np.float64)
        assert abs(g64.sum() - s1) <= 3e-4 * max(np.sqrt(s2), 1e-12) + 1e-9, k
        assert abs((g64 * g64).sum() - s2) <= 3e-4 * s2 + 1e-18, k
    print(f'worst gradient: {worst[1]} at {worst[0]:.3e} x max|g|')
    for k in ('downs.0.weight', 'mid.0.sa.conv1.weight', 'final_conv.block.3.bias'):     # the reference's own tensors
        ref = g['grad/' + k]
        assert np.abs(eng.get_grad(k) - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k
    dead = [k for k in sd if k not in grads_ref]
    assert len(dead) == 44 == int(g['n_params_without_grad'])
    with pytest.raises(_lib.FdsrError):
        eng.get_grad(dead[0])


def test_adam_update_and_weights_in_use(stepped, golden_dir):
    cfg, sd, eng, loss, numel, l_ref, grads_ref, new_ref = stepped
    g = np.load(os.path.join(golden_dir, 'train_step.npz'))
    lr = float(g['lr'])
    eng.adam_step(lr)
    for k in ('downs.0.weight', 'mid.0.sa.conv1.weight', 'final_conv.block.3.bias'):
        ref_g, aft, ref_aft = g['grad/' + k], eng.get_weight(k), g['after/' + k]
        mask = np.abs(ref_g) > 1e-3 * np.abs(ref_g).max()          # Adam's first step is ~lr*sign(g): compare where |g| is not tiny
        assert np.abs(aft - ref_aft)[mask].max() <= 2e-7, k
        assert np.abs(aft - ref_aft).max() <= 2.1 * lr, k
    # every executed tensor against the oracle's update, where its gradient is not tiny
    for k, ref in new_ref.items():
        if k not in grads_ref:
            continue
        gk = grads_ref[k].numpy()
        mask = np.abs(gk) > 1e-3 * np.abs(gk).max()
        if mask.any():
            assert np.abs(eng.get_weight(k) - ref.numpy())[mask].max() <= 3e-7, k
    # the forward now runs on the updated weights (device-side re-pack): same as a fresh engine loaded with them
    from fastdiffsr_amd.engine import Engine
    new_sd = {k: (eng.get_weight(k) if k in grads_ref else v) for k, v in sd.items()}
    e2 = Engine(cfg)
    e2.load_state_dict(new_sd)
    x = torch.randn(1, 6, 32, 32, generator=torch.Generator().manual_seed(3)).cuda()
    nl = torch.tensor([[0.4]]).cuda()
    if getattr(eng, 'precision', 'f32') == 'f16x3':
        # the f16x3 forms were re-packed ON THE DEVICE by the optimiser step: same network as host-packed forms
        # (the upsample convs run the generic kernel until the sub-pixel forms are refreshed: not bitwise, fp32-grade)
        e2.set_precision('f16x3')
        d = (eng.unet_forward(x, nl) - e2.unet_forward(x, nl)).abs().max().item()
        assert d <= 1e-5, d
    for prec in ('f32', 'f16x3'):
        eng.set_precision(prec)
        e2.set_precision(prec)
        ya, yb = eng.unet_forward(x, nl), e2.unet_forward(x, nl)
        assert torch.equal(ya, yb), (prec, (ya - yb).abs().max().item())
    eng.set_precision('f32')


def test_step_is_bitwise_reproducible(golden_dir):
    from fastdiffsr_amd.engine import Engine
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, 0)
    hr, sr, nz, gamma = _inputs(golden_dir)
    x = torch.cat([sr, _x_noisy(hr, sr, nz, gamma)], 1).cuda()
    outs = []
    for _ in range(2):
        eng = Engine(cfg)
        eng.load_state_dict(sd)
        eng.set_precision('f32')
        loss = eng.train_grads(x, gamma.cuda(), nz.cuda(), 'l1', 1.0 / x.numel() * 2)
        outs.append((loss, eng.get_grad('downs.4.res_block.block1.block.3.weight'), eng.get_grad('noise_level_mlp.1.weight'),
                     eng.get_grad('ups.14.res_block.block2.block.0.weight')))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_gradients_do_not_depend_on_what_the_workspace_held(golden_dir, prec):
    """No gradient buffer is zeroed between steps (the first contribution to each stores, later ones accumulate): a step on
    inputs A after a step on other inputs B must equal, bit for bit, the step on A from a fresh engine."""
    from fastdiffsr_amd.engine import Engine
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    hr, sr, nz, gamma = _inputs(golden_dir)
    xa = torch.cat([sr, _x_noisy(hr, sr, nz, gamma)], 1).cuda()
    xb = (xa.flip(0) * 3.0 + 0.5).contiguous()
    eng = Engine(cfg)
    eng.load_state_dict(synth_state_dict(cfg, 0))
    eng.set_precision(prec)
    names = ['downs.0.weight', 'downs.4.res_block.block1.block.3.weight', 'mid.0.res_block.block2.block.0.bias', 'noise_level_mlp.1.weight',
             'ups.14.res_block.block2.block.0.weight', 'final_conv.block.3.weight']
    scale = 1.0 / xa.numel() * 2

    def step(x, nzs):
        loss = eng.train_grads(x, gamma.cuda(), nzs.cuda(), 'l1', scale)
        return [loss] + [eng.get_grad(k).copy() for k in names]
    a1 = step(xa, nz)
    step(xb, -2.0 * nz)                                    # leaves every gradient tensor full of unrelated values
    a2 = step(xa, nz)
    assert a1[0] == a2[0]
    for k, u, v in zip(names, a1[1:], a2[1:]):
        assert np.array_equal(u, v), k


# ---------------------------------------------------------------------------------------------------------------
# train mode: nn.Dropout(p) in front of every block2 conv is live (unet.py:89-101)
# ---------------------------------------------------------------------------------------------------------------
def _res_blocks(cfg):
    from fastdiffsr_amd.arch import build_layers
    return [L.name for L in build_layers(cfg) if L.kind == 'res']


def test_train_mode_forward_has_live_dropout():
    """UNet.forward in .train() mode: the engine draws the masks; with exactly those masks the oracle gives the same
    output.  eval() gives the dropout-free network; consecutive training forwards draw fresh masks; a seed repeats."""
    from fastdiffsr_amd.unet import UNet
    from oracle import fdsr_oracle as O
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    assert cfg.dropout == 0.2
    net = UNet(in_channel=6, out_channel=3, norm_groups=32, inner_channel=64, channel_mults=(1, 2, 4, 4), attn_res=(16,), res_blocks=2,
               dropout=0.2, image_size=256)
    sd = synth_state_dict(cfg, 0)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    net = net.cuda()
    x = torch.randn(2, 6, 32, 32, generator=torch.Generator().manual_seed(11))
    nl = torch.tensor([[0.3], [0.9]])
    tsd = O.to_torch_sd(sd)
    with torch.no_grad():
        net.eval()
        ev = net(x.cuda(), nl.cuda()).cpu()
        assert (ev - O.unet_forward(tsd, cfg, x, nl)).abs().max().item() <= 1e-4
        net.train()
        torch.manual_seed(5)
        t1 = net(x.cuda(), nl.cuda()).cpu()
        masks = {b: net.engine.dropout_mask(b).cpu() for b in _res_blocks(cfg)}
        assert len(masks) == 22
        ref = O.unet_forward(tsd, cfg, x, nl, dropout_masks=masks)
        assert (t1 - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
        assert (t1 - ev).abs().max().item() > 1e-2                      # it is not the eval network
        keep = torch.cat([(m > 0).float().flatten() for m in masks.values()])
        assert abs(keep.mean().item() - 0.8) < 2e-3, keep.mean().item()
        for m in masks.values():
            assert set(torch.unique(m).tolist()) <= {0.0, 1.25}
        t2 = net(x.cuda(), nl.cuda()).cpu()
        assert (t2 - t1).abs().max().item() > 1e-3                      # fresh masks per forward (torch's generator moved on)
        torch.manual_seed(5)
        assert torch.equal(net(x.cuda(), nl.cuda()).cpu(), t1)          # ... and the same ones under the same torch seed
        net.eval()
        assert torch.equal(net(x.cuda(), nl.cuda()).cpu(), ev)          # eval() switches it off again


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_train_step_with_dropout_matches_oracle(golden_dir, prec):
    """The optimisation step with live dropout: loss and every gradient against autograd over the oracle fed with
    the masks the engine drew."""
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, 0)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    eng.set_training(True)
    eng.set_seed(123)
    hr, sr, nz, gamma = _inputs(golden_dir)
    b, c, h, w = hr.shape
    x = torch.cat([sr, _x_noisy(hr, sr, nz, gamma)], 1)
    loss = eng.train_grads(x.cuda(), gamma.cuda(), nz.cuda(), 'l1', 1.0 / (b * c * h * w))
    masks = {blk: eng.dropout_mask(blk).cpu() for blk in _res_blocks(cfg)}
    l_ref, grads_ref, _ = O.train_step(O.to_torch_sd(sd), cfg, hr, sr, gamma, nz, lr=1e-4, dropout_masks=masks)
    assert abs(loss / (b * c * h * w) - l_ref.item()) <= 1e-5 * abs(l_ref.item())
    worst = (0.0, '')
    for k, ref in grads_ref.items():
        got, ref = eng.get_grad(k), ref.numpy()
        scale = max(float(np.abs(ref).max()), 1e-12)
        d = float(np.abs(got - ref).max())
        worst = max(worst, (d / scale, k))
        assert d <= 1e-4 * scale, f'{k}: max|d| {d:.3e} vs max|g| {scale:.3e}'
    print(f'worst gradient with dropout: {worst[1]} at {worst[0]:.3e} x max|g|')
    # the loss differs from the dropout-free step: the masks did something
    g = np.load(os.path.join(golden_dir, 'train_step.npz'))
    assert abs(loss / (b * c * h * w) - float(g['l_pix'])) > 1e-3 * float(g['l_pix'])


# ---------------------------------------------------------------------------------------------------------------
# the DDPM wrapper: optimize_parameters, checkpoints (model/model.py:47-57, :126-166)
# ---------------------------------------------------------------------------------------------------------------
def _train_opt(tmp_path, resume=None, dropout=0.0):
    return {'phase': 'train', 'gpu_ids': [0], 'distributed': False,
            'path': {'checkpoint': str(tmp_path), 'resume_state': resume},
            'datasets': {'train': {'l_resolution': 64}},
            'train': {'optimizer': {'type': 'adam', 'lr': 1e-4}},
            'model': {'which_model_G': 'fastdiffsr', 'finetune_norm': False,
                      'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 32, 'norm_groups': 32,
                               'channel_multiplier': [1, 2, 4, 4], 'attn_res': [16], 'res_blocks': 1, 'dropout': dropout},
                      'beta_schedule': {'train': dict(schedule='linear_cosine', n_timestep=20, linear_start=1e-6, linear_end=1e-2),
                                        'val': dict(schedule='linear_cosine', n_timestep=20, linear_start=1e-6, linear_end=1e-2)},
                      'diffusion': {'image_size': 32, 'channels': 3, 'conditional': True}}}


def test_ddpm_optimize_parameters_and_checkpoints(tmp_path):
    """Three optimize_parameters() calls through the DDPM wrapper == three oracle train steps with the same draws;
    save_network / load_network round-trip the generator AND the optimiser state (resumed step == uninterrupted step)."""
    from unittest import mock
    from fastdiffsr_amd import model as M
    from fastdiffsr_amd.arch import UNetConfig as UC
    from oracle import fdsr_oracle as O
    torch.manual_seed(7)
    ddpm = M.create_model(_train_opt(tmp_path))
    unet = ddpm.netG.denoise_fn
    cfg = unet.cfg
    sd0 = {k: v.detach().cpu().numpy().copy() for k, v in unet.state_dict().items()}
    gen = torch.Generator().manual_seed(3)
    hr = torch.rand(2, 3, 32, 32, generator=gen) * 2 - 1
    sr = (hr + 0.2 * torch.randn(2, 3, 32, 32, generator=gen)).clamp(-1, 1)
    draws = [(5, np.array([0.61, 0.58])), (12, np.array([0.31, 0.27])), (2, np.array([0.93, 0.91]))]
    noises = [torch.randn(2, 3, 32, 32, generator=gen) for _ in draws]

    def step(model, i):
        t, gam = draws[i]
        model.feed_data({'HR': hr.clone(), 'SR': sr.clone()})
        with mock.patch.object(np.random, 'randint', lambda a, b: t), mock.patch.object(np.random, 'uniform', lambda a, b, size: gam), \
                mock.patch.object(torch, 'randn_like', lambda x: noises[i].to(x.device)):
            model.optimize_parameters()
        return model.get_current_log()['l_pix']

    # oracle: three Adam steps with torch.optim.Adam over autograd (the reference's own optimiser)
    leaves = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in sd0.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=1e-4)
    ref_losses = []
    for i, (t, gam) in enumerate(draws):
        opt.zero_grad()
        loss = O.p_losses(leaves, cfg, hr, sr, torch.FloatTensor(gam), noises[i], 'l1') / hr.numel()
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    l0 = step(ddpm, 0)
    l1 = step(ddpm, 1)
    gen_path = ddpm.save_network(epoch=1, iter_step=2)
    assert os.path.exists(gen_path) and os.path.exists(gen_path.replace('_gen.pth', '_opt.pth'))
    l2 = step(ddpm, 2)
    for got, ref in zip((l0, l1, l2), ref_losses):
        assert abs(got - ref) <= 2e-5 * abs(ref), (got, ref)
    final = unet.state_dict()
    for k in ('downs.0.weight', 'mid.0.ca.fc1.weight', 'ups.1.res_block.block2.block.3.weight', 'noise_level_mlp.3.bias'):
        d = (final[k].cpu() - leaves[k].detach()).abs().max().item()
        assert d <= 3e-5, (k, d)      # three steps of lr 1e-4 move weights by ~3e-4; where |g| is tiny Adam's m/sqrt(v) amplifies fp32 noise
    # resume from the checkpoint written after step 2 and repeat step 3: same loss, same weights
    ck = torch.load(gen_path, map_location='cpu')
    assert list(ck.keys()) == list(ddpm.netG.state_dict().keys())
    o = torch.load(gen_path.replace('_gen.pth', '_opt.pth'), map_location='cpu', weights_only=False)
    assert o['iter'] == 2 and o['epoch'] == 1 and o['scheduler'] is None
    assert set(o['optimizer'].keys()) == {'state', 'param_groups'} and len(o['optimizer']['state']) == len(
        [1 for k, _, live in unet.engine.schema() if live])
    resumed = M.create_model(_train_opt(tmp_path, resume=gen_path[:-len('_gen.pth')]))
    assert resumed.begin_step == 2 and resumed.begin_epoch == 1
    l2r = step(resumed, 2)
    assert l2r == l2
    fr = resumed.netG.denoise_fn.state_dict()
    for k in final:
        assert torch.equal(fr[k].cpu(), final[k].cpu()), k


def test_autograd_compatible_loss_fills_param_grads():
    """The reference's own sequence (model.py:48-56) on the module: l_pix = netG(data); l_pix.sum() / n; backward();
    torch.optim.Adam.step() -- gradients arrive in Parameter.grad, the next forward uses the stepped weights."""
    from unittest import mock
    from fastdiffsr_amd import networks
    torch.manual_seed(9)
    opt = _train_opt('/tmp', dropout=0.2)
    netG = networks.define_G(opt).cuda()
    netG.set_loss('cuda')
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['train'], 'cuda')
    netG.train()
    optG = torch.optim.Adam(list(netG.parameters()), lr=1e-4)
    gen = torch.Generator().manual_seed(4)
    hr = (torch.rand(2, 3, 32, 32, generator=gen) * 2 - 1).cuda()
    sr = (hr + 0.1 * torch.randn(2, 3, 32, 32, generator=gen).cuda()).clamp(-1, 1)
    optG.zero_grad()
    l_pix = netG({'HR': hr, 'SR': sr})
    b, c, h, w = hr.shape
    l_pix = l_pix.sum() / int(b * c * h * w)
    l_pix.backward()
    unet = netG.denoise_fn
    named = dict(unet.named_parameters())
    live = {k for k, _, lv in unet.engine.schema() if lv}
    for k, p in named.items():
        if k in live:
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
            assert np.allclose(p.grad.cpu().numpy(), unet.engine.get_grad(k) / (b * c * h * w), rtol=1e-6, atol=1e-12), k
        else:
            assert p.grad is None, k                                   # never executed (unet.py:212)
    before = named['downs.0.weight'].detach().clone()
    optG.step()
    assert (named['downs.0.weight'] - before).abs().max().item() > 1e-5
    netG.eval()
    with torch.no_grad():
        x = torch.randn(1, 6, 32, 32, generator=gen).cuda()
        y = unet(x, torch.tensor([[0.5]]).cuda())                      # re-uploads the stepped Parameters
    assert np.array_equal(unet.engine.get_weight('downs.0.weight'), named['downs.0.weight'].detach().cpu().numpy())
    assert torch.isfinite(y).all()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('option', [('wgrad_form', 1), ('wgrad_form', 2), ('wgrad_colsum', 0), ('wgrad_f32', 1), ('wgrad_big_bytes', 1 << 20), ('gnb_fuse', 0), ('drop_stage', 0), ('splitk', 0)])
def test_every_weight_gradient_kernel_form_meets_the_golden(option):
    """The f16x3 step picks its weight-gradient kernel per layer (8-wave in-row with fused column sums by default); the A/B
    options (fdsr_debug_option) force the other forms -- 4-wave, 8-wave without the interleave, separate column-sum pass,
    exact-fp32 weight gradients, the ">= 4 GiB tensor" fallback (threshold lowered to 1 MiB: every large layer takes the
    64-bit-offset kernel), the GroupNorm backward with its reduction as a pass of its own instead of inside the
    input-gradient launch, the forward's Dropout as a materialised tensor instead of in the conv's staging, and no K split (at these
    sizes the default splits K, and only launches without a split apply Dropout in their staging) -- and each must reproduce the reference's 273 gradients and its Adam update (a fresh process per
    option, so that nothing else in this session runs under it)."""
    import subprocess
    import sys
    env = dict(os.environ)
    env['FDSR_TEST_DEBUG_OPTION'] = f'{option[0]}={option[1]}'
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, '-m', 'pytest', here, '-m', 'gpu', '-q', '-x', '-k',
                        '(test_all_gradients or test_adam_update or test_train_step_with_dropout) and f16x3'],
                       env=env, capture_output=True, text=True, timeout=840, cwd=os.path.dirname(os.path.dirname(here)))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout


@pytest.mark.parametrize('shape', [(3, 40, 56), (2, 72, 48), (5, 32, 32)])
def test_f16x3_step_on_ragged_shapes_matches_the_exact_fp32_step(shape):
    """Shapes that are no multiple of any tile (8x16-pixel weight-gradient tiles, 16 / 32-pixel conv tiles), batch sizes that
    make the image-aligned slices uneven, 32-channel levels that half-fill the 64-channel weight-gradient blocks and concat
    seams inside a block (the 4-wave fallback): the f16x3 step (split-f16 kernels) against the exact-fp32 step (a different
    kernel family) of the same engine, every gradient within 1e-4 of its tensor's max."""
    from fastdiffsr_amd.engine import Engine
    B, H, W = shape
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4), attn_res=(16,),
                     res_blocks=1, dropout=0.0, image_size=32)
    sd = synth_state_dict(cfg, 7)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 6, H, W, generator=g).cuda()
    nl = (torch.rand(B, generator=g) * 0.5 + 0.4).cuda()
    tgt = torch.randn(B, 3, H, W, generator=g).cuda()
    grads = {}
    for prec in ('f32', 'f16x3'):
        eng = Engine(cfg)
        eng.load_state_dict(sd)
        eng.set_precision(prec)
        loss = eng.train_grads(x, nl, tgt, 'l1', 1.0 / x[:, :3].numel())
        grads[prec] = (loss, {k: eng.get_grad(k).copy() for k, _, live in eng.schema() if live})
    assert abs(grads['f32'][0] - grads['f16x3'][0]) <= 1e-5 * abs(grads['f32'][0])
    # (with 32 groups over 32 channels a bias or noise shift in front of a GroupNorm has an exactly zero gradient: what either
    # step reports there is rounding noise, ~1e-6 of the typical gradient -- hence the absolute floor)
    gmax = max(float(np.abs(a).max()) for a in grads['f32'][1].values())
    worst = 0.0
    for k, a in grads['f32'][1].items():
        b = grads['f16x3'][1][k]
        scale = float(np.abs(a).max())
        d = float(np.abs(a - b).max())
        assert d <= 1e-4 * scale + 1e-6 * gmax, (k, d, scale, gmax)
        if scale > 1e-3 * gmax:
            worst = max(worst, d / scale)
    print(f'ragged {shape}: worst relative gradient difference {worst:.2e} (tensors above 1e-3 of the largest gradient)')


@pytest.mark.parametrize('shape', [(3, 40, 56), (2, 64, 64)])
def test_gn_backward_inside_the_input_gradient_launch_equals_the_separate_pass(shape):
    """f16x3 step: the first half of the GroupNorm + Swish (+ Dropout) backward -- g = dA * keep/(1-p) * swish'(u) and the
    per-channel sums of g and g*xhat -- runs in the epilogue of the 16x16x32 convolution launch that produces dA
    (ConvParams::gb_*; default), or as gn_bwd_reduce_kernel over dA (`gnb_fuse=0`).  Same masks, same inputs: every gradient
    agrees to fp32 summation-order noise, on whole tiles and on a shape whose tiles are ragged on both axes, and the fused
    form is the one that ran (it is not bit-identical: the sums are taken per tile, not per pixel slice)."""
    from fastdiffsr_amd import _lib
    from fastdiffsr_amd.engine import Engine
    B, H, W = shape
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4), attn_res=(16,),
                     res_blocks=1, dropout=0.2, image_size=32)
    sd = synth_state_dict(cfg, 7)
    g = torch.Generator().manual_seed(13)
    x = torch.randn(B, 6, H, W, generator=g).cuda()
    nl = (torch.rand(B, generator=g) * 0.5 + 0.4).cuda()
    tgt = torch.randn(B, 3, H, W, generator=g).cuda()
    grads = {}
    try:
        for fuse in (1, 0):
            _lib.debug_option('gnb_fuse', fuse)
            eng = Engine(cfg)
            eng.load_state_dict(sd)
            eng.set_precision('f16x3')
            eng.set_training(True)
            eng.set_seed(99)
            loss = eng.train_grads(x, nl, tgt, 'l1', 1.0 / x[:, :3].numel())
            grads[fuse] = (loss, {k: eng.get_grad(k).copy() for k, _, live in eng.schema() if live})
    finally:
        _lib.debug_option('gnb_fuse', 1)
    assert grads[0][0] == grads[1][0]                    # the forward is the same launch set
    gmax = max(float(np.abs(a).max()) for a in grads[0][1].values())
    differs = False
    for k, a in grads[0][1].items():
        b = grads[1][1][k]
        d = float(np.abs(a - b).max())
        assert d <= 2e-5 * float(np.abs(a).max()) + 1e-7 * gmax, (k, d, float(np.abs(a).max()), gmax)
        differs = differs or d > 0
    assert differs, 'gnb_fuse=1 ran the separate reduce pass'


@pytest.mark.parametrize('shape', [(3, 40, 56), (2, 64, 64)])
def test_dropout_in_the_conv_staging_equals_the_materialised_dropped_activation(shape):
    """f16x3 training forward: block2's Dropout (unet.py:89-101, between the Swish and the conv) is applied to the staged quad
    inside the 16x16x32 kernels (ConvParams::drop_mask, default), or gn_silu_drop_kernel writes swish(gn(x)) * keep / (1-p) to a
    tensor the conv then reads raw (`drop_stage=0`).  Same seed, same masks: the forward outputs and every gradient of the step
    agree to rounding (the two forms round the sigmoid differently, so they are not bit-identical -- which also shows that
    the staged form ran), on whole tiles and on ragged ones, with and without a res_conv rider in the same launch."""
    from fastdiffsr_amd import _lib
    from fastdiffsr_amd.engine import Engine
    B, H, W = shape
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4), attn_res=(16,),
                     res_blocks=2, dropout=0.2, image_size=32)
    sd = synth_state_dict(cfg, 3)
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, 6, H, W, generator=g).cuda()
    nl = (torch.rand(B, generator=g) * 0.5 + 0.4).cuda()
    tgt = torch.randn(B, 3, H, W, generator=g).cuda()
    res = {}
    try:
        _lib.debug_option('splitk', 0)      # (grids this small would split K, and a K-split launch reads the materialised form)
        for staged in (1, 0):
            _lib.debug_option('drop_stage', staged)
            eng = Engine(cfg)
            eng.load_state_dict(sd)
            eng.set_precision('f16x3')
            eng.set_training(True)
            eng.set_seed(5)
            y = eng.unet_forward(x, nl.view(B, 1)).cpu().numpy()
            eng.set_seed(5)
            loss = eng.train_grads(x, nl, tgt, 'l1', 1.0 / x[:, :3].numel())
            res[staged] = (y, loss, {k: eng.get_grad(k).copy() for k, _, live in eng.schema() if live})
    finally:
        _lib.debug_option('drop_stage', 1)
        _lib.debug_option('splitk', 1)
    ya, yb = res[1][0], res[0][0]
    assert np.abs(ya - yb).max() <= 2e-5 * np.abs(yb).max()
    assert not np.array_equal(ya, yb), 'drop_stage=1 ran the materialised form'
    assert abs(res[1][1] - res[0][1]) <= 1e-5 * abs(res[0][1])
    gmax = max(float(np.abs(a).max()) for a in res[0][2].values())
    for k, a in res[0][2].items():
        d = float(np.abs(a - res[1][2][k]).max())
        assert d <= 5e-5 * float(np.abs(a).max()) + 1e-6 * gmax, (k, d, float(np.abs(a).max()), gmax)


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
def test_step_from_the_training_pair_equals_the_step_from_torch_formed_inputs(golden_dir, prec):
    """fdsr_train_grads_pairs forms img2res + q_sample + cat([SR, x_noisy]) in the engine's input kernel (diffusion.py:233-263):
    the same step as on the tensor torch forms op by op (same separately rounded products and sums; loss equal to 1e-6, every gradient
    within 2e-5 of its tensor's max -- a last-bit difference in a few input elements is all that separates them); with
    noise=None the engine draws the target itself (fresh per step, repeatable under set_seed)."""
    from fastdiffsr_amd.engine import Engine
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, 0)
    hr, sr, nz, gamma = _inputs(golden_dir)
    b, c, h, w = hr.shape
    scale = 1.0 / (b * c * h * w)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    x = torch.cat([sr, _x_noisy(hr, sr, nz, gamma)], 1)
    l_a = eng.train_grads(x.cuda(), gamma.cuda(), nz.cuda(), 'l1', scale)
    g_a = {k: eng.get_grad(k).copy() for k, _, live in eng.schema() if live}
    l_b = eng.train_grads_pairs(hr.cuda(), sr.cuda(), gamma.cuda(), nz.cuda(), 'l1', scale)
    assert abs(l_a - l_b) <= 1e-6 * abs(l_a)
    gmax = max(float(np.abs(v).max()) for v in g_a.values())
    for k, v in g_a.items():
        assert np.abs(eng.get_grad(k) - v).max() <= 2e-5 * float(np.abs(v).max()) + 1e-7 * gmax, k
    # engine-drawn noise: finite, different from step to step, the same again under the same seed
    eng.set_seed(7)
    l1 = eng.train_grads_pairs(hr.cuda(), sr.cuda(), gamma.cuda(), None, 'l1', scale)
    l2 = eng.train_grads_pairs(hr.cuda(), sr.cuda(), gamma.cuda(), None, 'l1', scale)
    eng.set_seed(7)
    l3 = eng.train_grads_pairs(hr.cuda(), sr.cuda(), gamma.cuda(), None, 'l1', scale)
    assert np.isfinite([l1, l2]).all() and l1 != l2 and l1 == l3
    assert abs(l1 / (b * c * h * w) - 0.8) < 0.3          # ~E|N(0,1) - eps| of an untrained network
