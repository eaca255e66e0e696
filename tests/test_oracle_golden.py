"""Pin the oracle (oracle/fdsr_oracle.py) against outputs of the reference itself
(tests/golden/*.npz, made by oracle/make_goldens.py in the build container)."""
import os

import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL, SCHEDULE_BUFFERS
from fastdiffsr_amd.synth import synth_state_dict, state_dict_sha256, synth_inputs
from oracle import fdsr_oracle as O

TOL = 2e-6   # SURVEY section 7 stage 1: restatement vs reference <= 2e-6


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_beta_schedules_all_branches(golden_dir):
    g = _load(golden_dir, 'schedule.npz')
    n = 0
    for key in g.files:
        if not key.startswith('betas/'):
            continue
        _, name, T = key.split('/')
        ls, le = (1e-6, 1e-2) if name in ('linear_cosine', 'linear') else (1e-4, 2e-2)
        b = O.make_beta_schedule(name, int(T), ls, le)
        np.testing.assert_allclose(b, g[key], rtol=0, atol=1e-15)
        n += 1
    assert n == 9
    with pytest.raises(NotImplementedError):
        O.make_beta_schedule('nope', 10)


@pytest.mark.parametrize('T', [20, 10])
def test_schedule_buffers_bit_exact(golden_dir, T):
    g = _load(golden_dir, 'schedule.npz')
    tab = O.schedule_tables(dict(schedule='linear_cosine', n_timestep=T, linear_start=1e-6, linear_end=1e-2))
    for k in SCHEDULE_BUFFERS:
        assert tab[k].dtype == np.float32
        np.testing.assert_array_equal(tab[k], g[f'buf/{T}/{k}'])
    np.testing.assert_array_equal(tab['sqrt_alphas_cumprod_prev_f64'], g[f'buf/{T}/sqrt_alphas_cumprod_prev_f64'])
    if T == 20:   # SURVEY App. B eyeball values
        assert abs(tab['betas'][0] - 0.0160) < 5e-5 and tab['betas'][-1] == np.float32(0.999)
        assert abs(tab['sqrt_alphas_cumprod_prev_f64'][20] - 6.63449433e-07) < 1e-15


@pytest.fixture(scope='module')
def small(golden_dir):
    g = _load(golden_dir, 'unet_small.npz')
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                     channel_mults=(1, 2, 4, 4), attn_res=(16,), res_blocks=2, dropout=0.2, image_size=32)
    sd = synth_state_dict(cfg, seed=7)
    assert state_dict_sha256(sd) == str(g['weights_sha256'])
    return g, cfg, O.to_torch_sd(sd)


def test_unet_small_forward(small):
    g, cfg, sd = small
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        for i in range(4):
            eps = O.unet_forward(sd, cfg, x, torch.from_numpy(g[f'nl/{i}']))
            assert np.abs(eps.numpy() - g[f'eps/{i}']).max() <= TOL


def test_unet_small_submodules(small):
    g, cfg, sd = small
    nl = torch.from_numpy(g['nl/3'])
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        np.testing.assert_allclose(O.positional_encoding(nl, 32).numpy(), g['posenc'], atol=1e-7)
        t = O.noise_level_mlp(sd, nl, 32)
        np.testing.assert_allclose(t.numpy(), g['t_mlp'], atol=TOL)
        h = torch.nn.functional.conv2d(x, sd['downs.0.weight'], sd['downs.0.bias'], padding=1)
        np.testing.assert_allclose(h.numpy(), g['downs0'], atol=TOL)
        np.testing.assert_allclose(O.resnet_block(sd, 'downs.1', h, t, 32, False).numpy(), g['downs1'], atol=TOL)
        xm = torch.from_numpy(g['mid_in'])
        np.testing.assert_allclose(O.clam(sd, 'mid.0.ca', xm).numpy(), g['clam'], atol=TOL)
        np.testing.assert_allclose(O.slam(sd, 'mid.0.sa', xm).numpy(), g['slam'], atol=TOL)
        m0 = O.resnet_block(sd, 'mid.0', xm, t, 32, False)
        m0 = O.slam(sd, 'mid.0.sa', O.clam(sd, 'mid.0.ca', m0))
        np.testing.assert_allclose(m0.numpy(), g['mid0'], atol=TOL)
        F = torch.nn.functional
        up = f"ups.{int(g['up_idx'])}"
        xu = F.interpolate(torch.from_numpy(g['up_in']), scale_factor=2, mode='nearest')
        np.testing.assert_allclose(F.conv2d(xu, sd[f'{up}.conv.weight'], sd[f'{up}.conv.bias'], padding=1).numpy(),
                                   g['up_out'], atol=TOL)
        dn = f"downs.{int(g['down_idx'])}"
        np.testing.assert_allclose(
            F.conv2d(torch.from_numpy(g['down_in']), sd[f'{dn}.conv.weight'], sd[f'{dn}.conv.bias'], stride=2, padding=1).numpy(),
            g['down_out'], atol=TOL)


@pytest.fixture(scope='module')
def full(golden_dir):
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, seed=0)
    g = _load(golden_dir, 'unet_full.npz')
    assert state_dict_sha256(sd) == str(g['weights_sha256'])
    return cfg, O.to_torch_sd(sd), O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL)


def test_unet_full_forward(full, golden_dir):
    cfg, sd, _ = full
    g = _load(golden_dir, 'unet_full.npz')
    gen = torch.Generator().manual_seed(21)
    x64 = torch.randn(1, 6, 64, 64, generator=gen)
    x32 = torch.randn(2, 6, 32, 32, generator=gen)
    with torch.no_grad():
        e64 = O.unet_forward(sd, cfg, x64, torch.full((1, 1), 0.5))
        e32 = O.unet_forward(sd, cfg, x32, torch.tensor([[0.0209801132], [0.9919746]]))
    assert np.abs(e64.numpy() - g['eps64']).max() <= TOL
    assert np.abs(e32.numpy() - g['eps32']).max() <= TOL


def test_sample_loop_trajectory(full, golden_dir):
    cfg, sd, tab = full
    g = _load(golden_dir, 'sample_loop.npz')
    cond, noise = synth_inputs(2, 32, 32, 20)
    out, traj = O.p_sample_loop(sd, cfg, tab, cond, noise, return_trajectory=True)
    tr = torch.stack(traj).numpy()
    # per-step: the first steps are chaotic in x0 but clamped; bound is absolute
    assert np.abs(tr - g['traj32']).max() <= 2e-5
    assert np.abs(out.numpy() - g['out32']).max() <= 2e-5
    # B=1 through the reference's own p_sample_loop entry point
    out1 = O.p_sample_loop(sd, cfg, tab, cond[:1], noise[:, :1])
    assert np.abs(out1.numpy() - g['final32_b1']).max() <= 2e-5
    assert np.abs(out1.numpy() - g['continous32_b1'][-1]).max() <= 2e-5
    # continous=True keeps x_in + 7 frames at t = 18,15,...,0
    assert O.continuous_frames(20) == [18, 15, 12, 9, 6, 3, 0]
    frames = [O.res2img(traj[19 - t][:1], cond[:1]).numpy() for t in O.continuous_frames(20)]
    assert np.abs(np.concatenate(frames) - g['continous32_b1'][1:]).max() <= 2e-5


def test_sample_loop_64(full, golden_dir):
    cfg, sd, tab = full
    g = _load(golden_dir, 'sample_loop.npz')
    cond, noise = synth_inputs(1, 64, 64, 20)
    out = O.p_sample_loop(sd, cfg, tab, cond, noise)
    assert np.abs(out.numpy() - g['out64']).max() <= 2e-5


def test_training_loss(full, golden_dir):
    cfg, sd, _ = full
    g = _load(golden_dir, 'train_loss.npz')
    hr, sr, nz = (torch.from_numpy(g[k]) for k in ('hr', 'sr', 'noise'))
    np.testing.assert_array_equal(O.img2res(hr, sr).numpy(), g['img2res'])
    np.testing.assert_array_equal(O.res2img(nz, sr).numpy(), g['res2img'])
    gamma = torch.FloatTensor(g['gamma'])      # reference casts the numpy draw to fp32 (diffusion.py:246)
    with torch.no_grad():
        loss = O.p_losses(sd, cfg, hr, sr, gamma, nz)
    assert abs(loss.item() - float(g['loss'])) <= 1e-3 * abs(float(g['loss'])) * 1e-2 + 1e-2


def test_tensor2img_psnr():
    t = torch.tensor([[[-1.5, -1.0], [0.0, 1.0]]] * 3)
    img = O.tensor2img_u8(t)
    assert img.shape == (2, 2, 3) and img.dtype == np.uint8
    assert img[0, 0, 0] == 0 and img[0, 1, 0] == 0 and img[1, 0, 0] == 128 and img[1, 1, 0] == 255
    assert O.psnr_u8(img, img) == float('inf')
    b = img.copy(); b[0, 0, 0] = 10
    assert abs(O.psnr_u8(img, b) - 20 * np.log10(255 / np.sqrt(100 / 12))) < 1e-9


def test_metrics_vs_reference_goldens(golden_dir):
    """tensor2img / PSNR of the val loop against the reference's own core/metrics.py outputs:
    both the oracle's and the product's host implementations."""
    from fastdiffsr_amd import metrics as M
    g = _load(golden_dir, 'metrics.npz')
    t = torch.from_numpy(g['t'])
    for impl in (O.tensor2img_u8, M.tensor2img):
        np.testing.assert_array_equal(impl(t.clone()), g['img'])
    np.testing.assert_array_equal(M.tensor2img(t[:1].clone()), g['gray'])
    assert abs(O.psnr_u8(g['img'], g['img2']) - float(g['psnr'])) < 1e-12
    assert abs(M.calculate_psnr(g['img'], g['img2']) - float(g['psnr'])) < 1e-12
    assert M.calculate_psnr(g['img'], g['img']) == float('inf') == float(g['psnr_same'])
    assert M.calculate_ssim(g['img'], g['img']) > 0.999999
    assert 0 < M.calculate_ssim(g['img'], g['img2']) < 1
    assert M.calculate_ergas(g['img'], g['img']) == 0.0


def test_ssim_ergas_vs_the_references_own_functions(golden_dir):
    """core/metrics.py:103-152 run from the reference itself on fixed image pairs (tests/golden/metrics_ssim.npz, made by
    oracle/make_goldens.py `metrics`, which supplies cv2.getGaussianKernel / cv2.filter2D / skimage compare_mse from their
    published definitions because the image has neither package): the reference's arithmetic around those primitives is
    pinned -- window, valid crop, constants, the 3-channel branch that scores the whole array three times, ERGAS' mean and
    channel divisor -- to 1e-12."""
    from fastdiffsr_amd import metrics as M
    g = _load(golden_dir, 'metrics_ssim.npz')
    for case in ('noisy', 'blur', 'dark', 'same'):
        a, b = g[f'{case}/a'], g[f'{case}/b']
        assert abs(M.calculate_ssim(a, b) - float(g[f'{case}/ssim_rgb'])) < 1e-12, case
        assert abs(M.calculate_ssim(a[..., 0], b[..., 0]) - float(g[f'{case}/ssim_gray'])) < 1e-12, case
        assert abs(M.calculate_ssim(a[..., :1], b[..., :1]) - float(g[f'{case}/ssim_1ch'])) < 1e-12, case
        assert abs(M.calculate_ergas(a, b, scale=4) - float(g[f'{case}/ergas4'])) < 1e-10, case
        assert abs(M.calculate_ergas(a, b, scale=8) - float(g[f'{case}/ergas8'])) < 1e-10, case


def test_metrics_oracle_pinned(golden_dir):
    """oracle/metrics_oracle.py -- the checker of the HIP metric kernels -- against (a) the reference's own core/metrics.py functions
    run in the build container (tests/golden/metrics_ssim.npz, metrics.npz), (b) brute-force window loops of the published
    definitions: skimage's compare_ssim (uniform 7 x 7, sample covariance, interior mean), cv2.filter2D's BORDER_REFLECT_101."""
    from oracle import metrics_oracle as MO
    g = _load(golden_dir, 'metrics_ssim.npz')
    for case in ('noisy', 'blur', 'dark', 'same'):
        a, b = g[f'{case}/a'], g[f'{case}/b']
        assert abs(MO.calculate_ssim(a, b) - float(g[f'{case}/ssim_rgb'])) < 1e-12, case
        assert abs(MO.calculate_ssim(a[..., 0], b[..., 0]) - float(g[f'{case}/ssim_gray'])) < 1e-12, case
        assert abs(MO.calculate_ssim(a[..., :1], b[..., :1]) - float(g[f'{case}/ssim_1ch'])) < 1e-12, case
        assert abs(MO.calculate_ergas(a, b, scale=4) - float(g[f'{case}/ergas4'])) < 1e-10, case
        assert abs(MO.calculate_ergas(a, b, scale=8) - float(g[f'{case}/ergas8'])) < 1e-10, case
    m = _load(golden_dir, 'metrics.npz')
    assert abs(MO.calculate_psnr(m['img'], m['img2']) - float(m['psnr'])) < 1e-12
    assert MO.calculate_psnr(m['img'], m['img']) == float('inf') == MO.compare_psnr(m['img'], m['img'])
    rng = np.random.default_rng(9)
    a = rng.integers(0, 256, (20, 26, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-30, 31, a.shape), 0, 255).astype(np.uint8)
    mse = np.mean((a.astype(np.float64) - b) ** 2)
    assert MO.compare_mse(a, b) == mse and abs(MO.compare_psnr(a, b) - MO.calculate_psnr(a, b)) < 1e-12
    X, Y = a[..., 1].astype(np.float64), b[..., 1].astype(np.float64)
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for y in range(3, X.shape[0] - 3):
        for x in range(3, X.shape[1] - 3):
            wx, wy = X[y - 3:y + 4, x - 3:x + 4].ravel(), Y[y - 3:y + 4, x - 3:x + 4].ravel()
            ux, uy = wx.mean(), wy.mean()
            vxy = ((wx - ux) * (wy - uy)).sum() / 48
            vals.append((2 * ux * uy + C1) * (2 * vxy + C2) / ((ux * ux + uy * uy + C1) * (wx.var(ddof=1) + wy.var(ddof=1) + C2)))
    assert abs(MO.compare_ssim(a[..., 1], b[..., 1]) - np.mean(vals)) < 1e-9
    assert abs(MO.compare_ssim(a, b, multichannel=True) - np.mean([MO.compare_ssim(a[..., c], b[..., c]) for c in range(3)])) < 1e-12
    assert abs(MO.compare_ssim(a, a, multichannel=True) - 1.0) < 1e-12
    with pytest.raises(ValueError):
        MO.compare_ssim(a[:6], b[:6], multichannel=True)
    # cv2.filter2D: correlation (no kernel flip), anchor at the centre, BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba)
    img = rng.random((9, 8))
    k = rng.random((5, 5))
    pad = np.pad(img, 2, mode='reflect')
    brute = np.array([[np.sum(pad[y:y + 5, x:x + 5] * k) for x in range(8)] for y in range(9)])
    assert np.abs(MO.filter2d(img, k) - brute).max() < 1e-12
    kk = MO.gaussian_kernel(11, 1.5)
    assert abs(kk.sum() - 1.0) < 1e-15 and np.argmax(kk) == 5 and np.allclose(kk, kk[::-1]) and abs(kk[5] / kk[4] - np.exp(1 / 4.5)) < 1e-12


def test_pil_bicubic_restatement(golden_dir):
    """oracle/pil_bicubic.py == Pillow's Image.resize(BICUBIC), bit for bit (goldens made by PIL itself)."""
    from oracle import pil_bicubic as PB
    g = _load(golden_dir, 'bicubic.npz')
    for name in ('x4', 'x8', 'ragged'):
        sr = g[name + '/sr']
        np.testing.assert_array_equal(PB.resize_bicubic_u8(g[name + '/lr'], sr.shape[0], sr.shape[1]), sr)
    t = PB.u8_to_model_tensor(g['x4/sr'])
    assert t.shape == (3, 256, 256) and t.dtype == torch.float32 and float(t.min()) >= -1.0 and float(t.max()) <= 1.0
    assert t[1, 0, 0] == 1.0 and t[0, 0, 0] == -1.0      # the green band


def test_sr3_sibling_vs_reference_goldens(golden_dir):
    """the SR3 sibling (model/ddpm_modules) restated in oracle/sr3_oracle.py against outputs of
    the reference modules: UNet forward (integer time, self-attention), the attention block, the buffers
    and the reference's own p_sample_loop(continous=True) at T=12."""
    from oracle import sr3_oracle as S
    g = _load(golden_dir, 'sr3.npz')
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                     attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='ddpm')
    sdn = synth_state_dict(cfg, 5)
    assert state_dict_sha256(sdn) == str(g['weights_sha256'])
    sd = O.to_torch_sd(sdn)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        assert np.abs(S.unet_forward(sd, cfg, x, torch.tensor([3, 999])).numpy() - g['eps_t']).max() <= TOL
        assert np.abs(S.unet_forward(sd, cfg, x, torch.tensor([0, 0])).numpy() - g['eps_t0']).max() <= TOL
        an = str(g['attn_name'])
        assert np.abs(S.self_attention(sd, an, torch.from_numpy(g['attn_in']), 32).numpy() - g['attn_out']).max() <= TOL
    tab = O.schedule_tables(dict(schedule='linear', n_timestep=12, linear_start=1e-4, linear_end=2e-2))
    for k in SCHEDULE_BUFFERS:
        np.testing.assert_array_equal(tab[k], g[f'buf/{k}'])
    cond, noise = torch.from_numpy(g['cond']), torch.from_numpy(g['noise'])
    out, traj = S.p_sample_loop(sd, cfg, tab, cond, noise, return_trajectory=True)
    ref = g['continous']                      # [x_in(2) | img after t=11 (2) | ... | t=0 (2)]
    assert ref.shape[0] == 2 + 12 * 2
    assert np.abs(torch.cat([cond] + traj).numpy() - ref).max() <= 2e-5
    assert np.abs(out.numpy() - ref[-2:]).max() <= 2e-5


def test_training_step_matches_reference(full, golden_dir):
    """oracle.train_step vs one optimisation step of the reference itself (model.py:47-57; golden made by
    oracle/make_goldens.py train): loss, every gradient (sum / sum of squares per tensor + three tensors in
    full), the Adam update.  This is the oracle of SURVEY 8f-3; the HIP backward kernels are not built yet."""
    cfg, sd, _ = full
    g = _load(golden_dir, 'train_step.npz')
    tl = _load(golden_dir, 'train_loss.npz')
    hr, sr, nz = (torch.from_numpy(tl[k]) for k in ('hr', 'sr', 'noise'))
    gamma = torch.FloatTensor(tl['gamma'])
    l_pix, grads, new_sd = O.train_step(sd, cfg, hr, sr, gamma, nz, lr=float(g['lr']))
    assert abs(l_pix.item() - float(g['l_pix'])) <= 1e-6 * abs(float(g['l_pix']))
    keys = [str(k) for k in g['grad_keys']]
    assert sorted(keys) == sorted(grads.keys()) and len(keys) == 273          # the 44 dead tensors get none
    assert int(g['n_params_without_grad']) == 44 == len(sd) - len(grads)
    for k, (s1, s2) in zip(keys, g['grad_stats']):
        g64 = grads[k].double()
        scale = max(np.sqrt(s2), 1e-12)
        assert abs(g64.sum().item() - s1) <= 2e-4 * scale + 1e-9, k
        assert abs((g64 * g64).sum().item() - s2) <= 2e-4 * s2 + 1e-18, k
    for k in ('downs.0.weight', 'mid.0.sa.conv1.weight', 'final_conv.block.3.bias'):
        ref = g['grad/' + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k
        # Adam's first step moves every weight by ~lr * sign(g): compare where the reference's |g| is not tiny
        aft, ref_aft = new_sd[k].numpy(), g['after/' + k]
        mask = np.abs(ref) > 1e-3 * np.abs(ref).max()
        assert np.abs(aft - ref_aft)[mask].max() <= 2e-7, k
        assert np.abs(aft - ref_aft).max() <= 2.1 * float(g['lr']), k
    for k in sd:
        if k not in grads:
            assert torch.equal(new_sd[k], sd[k])


def test_sr3_training_step_matches_reference(golden_dir):
    """oracle.sr3_oracle.train_step vs one optimisation step of the reference's SR3 sibling itself (model/ddpm_modules p_losses :279-297
    + model.py:47-57; golden made by `oracle/make_goldens.py sr3_train`): loss, every gradient (sum / sum of squares per tensor, six
    tensors in full: a convolution, the SelfAttention's qkv / out / norm, the time MLP, a per-block Linear), the Adam update."""
    from oracle import sr3_oracle as S
    g = _load(golden_dir, 'sr3_train_step.npz')
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                     attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='ddpm')
    sd_np = synth_state_dict(cfg, 5)
    assert state_dict_sha256(sd_np) == str(g['weights_sha256'])
    sd = O.to_torch_sd(sd_np)
    tab = O.schedule_tables(dict(schedule='linear', n_timestep=12, linear_start=1e-4, linear_end=2e-2))
    hr, sr, nz = (torch.from_numpy(g[k]) for k in ('hr', 'sr', 'noise'))
    t = torch.from_numpy(g['t']).long()
    l_pix, grads, new_sd = S.train_step(sd, cfg, tab, hr, sr, t, nz, lr=float(g['lr']))
    assert abs(l_pix.item() - float(g['l_pix'])) <= 1e-6 * abs(float(g['l_pix']))
    keys = [str(k) for k in g['grad_keys']]
    assert sorted(keys) == sorted(grads.keys()) and int(g['n_params_without_grad']) == 0
    for k, (s1, s2) in zip(keys, g['grad_stats']):
        g64 = grads[k].double()
        scale = max(np.sqrt(s2), 1e-12)
        assert abs(g64.sum().item() - s1) <= 2e-4 * scale + 1e-9, k
        assert abs((g64 * g64).sum().item() - s2) <= 2e-4 * s2 + 1e-18, k
    for k in (str(x) for x in g['full_keys']):
        ref = g['grad/' + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k
        aft, ref_aft = new_sd[k].numpy(), g['after/' + k]
        mask = np.abs(ref) > 1e-3 * np.abs(ref).max()
        assert np.abs(aft - ref_aft)[mask].max() <= 2e-7, k
        assert np.abs(aft - ref_aft).max() <= 2.1 * float(g['lr']), k


def test_gdp_training_step_matches_reference(golden_dir):
    """oracle.gdp_oracle.train_step vs one optimisation step of the reference's GDP sibling itself (gdp_modules p_losses :277-299: the
    summed MSE against HR, then model.py:47-57; golden made by `oracle/make_goldens.py gdp_train`): loss, every gradient (sum / sum of
    squares per tensor; eleven in full: up / down ResBlocks, the scale-shift Linear, multi-head attention, the time MLP), Adam."""
    from oracle import gdp_oracle as GO
    g = _load(golden_dir, 'gdp_train_step.npz')
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 2), attn_res=(2, 4),
                     res_blocks=1, dropout=0.1, image_size=32, variant='gdp')
    sd_np = synth_state_dict(cfg, 13)
    assert state_dict_sha256(sd_np) == str(g['weights_sha256'])
    sd = O.to_torch_sd(sd_np)
    tab = O.schedule_tables(dict(schedule='linear', n_timestep=8, linear_start=1e-4, linear_end=2e-2))
    hr, sr, nz = (torch.from_numpy(g[k]) for k in ('hr', 'sr', 'noise'))
    t = torch.from_numpy(g['t']).long()
    l_pix, grads, new_sd = GO.train_step(sd, cfg, tab, hr, sr, t, nz, lr=float(g['lr']))
    assert abs(l_pix.item() - float(g['l_pix'])) <= 1e-6 * abs(float(g['l_pix']))
    keys = [str(k) for k in g['grad_keys']]
    assert sorted(keys) == sorted(grads.keys()) and int(g['n_params_without_grad']) == 0
    for k, (s1, s2) in zip(keys, g['grad_stats']):
        g64 = grads[k].double()
        scale = max(np.sqrt(s2), 1e-12)
        assert abs(g64.sum().item() - s1) <= 2e-4 * scale + 1e-9, k
        assert abs((g64 * g64).sum().item() - s2) <= 2e-4 * s2 + 1e-18, k
    for k in (str(x) for x in g['full_keys']):
        ref = g['grad/' + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-9, k
        aft, ref_aft = new_sd[k].numpy(), g['after/' + k]
        mask = np.abs(ref) > 1e-3 * np.abs(ref).max()
        assert np.abs(aft - ref_aft)[mask].max() <= 2e-7, k
        assert np.abs(aft - ref_aft).max() <= 2.1 * float(g['lr']), k


def test_tesr_training_step_matches_reference(golden_dir):
    """oracle.tesr_oracle.train_step vs one optimisation step of the reference's TESR sibling itself (tesr_modules p_losses :224-250,
    Charbonnier mean, then model.py:47-57; golden made by `oracle/make_goldens.py tesr_train`)."""
    from oracle import tesr_oracle as TO
    g, tg = _load(golden_dir, 'tesr_train_step.npz'), _load(golden_dir, 'tesr.npz')
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                     attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='tesr')
    sd_np = synth_state_dict(cfg, 9)
    assert state_dict_sha256(sd_np) == str(g['weights_sha256'])
    sd = O.to_torch_sd(sd_np)
    hr, sr, nz = (torch.from_numpy(tg[k]) for k in ('hr', 'sr', 'loss_noise'))
    gamma = torch.FloatTensor(tg['gamma'])
    l_pix, grads, new_sd = TO.train_step(sd, cfg, hr, sr, gamma, nz, lr=float(g['lr']))
    assert abs(l_pix.item() - float(g['l_pix'])) <= 1e-6 * abs(float(g['l_pix']))
    keys = [str(k) for k in g['grad_keys']]
    assert sorted(keys) == sorted(grads.keys())
    for k, (s1, s2) in zip(keys, g['grad_stats']):
        g64 = grads[k].double()
        scale = max(np.sqrt(s2), 1e-30)
        assert abs(g64.sum().item() - s1) <= 2e-4 * scale + 1e-30, k
        assert abs((g64 * g64).sum().item() - s2) <= 2e-4 * s2 + 1e-40, k
    for k in (str(x) for x in g['full_keys']):
        ref = g['grad/' + k]
        assert np.abs(grads[k].numpy() - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-30, k
        aft, ref_aft = new_sd[k].numpy(), g['after/' + k]
        assert np.abs(aft - ref_aft).max() <= 2.1 * float(g['lr']), k


def test_tesr_oracle_matches_reference(golden_dir):
    """oracle/tesr_oracle.py vs the reference's own model/tesr_modules (tests/golden/tesr.npz)."""
    from oracle import tesr_oracle as TO
    g = _load(golden_dir, 'tesr.npz')
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                     attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='tesr')
    sd_np = synth_state_dict(cfg, 9)
    assert state_dict_sha256(sd_np) == str(g['weights_sha256'])
    assert not any(k.endswith('.conv.weight') and '.res_block' not in k and k.split('.')[0] != 'final_conv' and
                   'downs' in k and sd_np[k].shape[-1] == 1 for k in sd_np)          # no dead 1x1 convs in this variant
    sd = O.to_torch_sd(sd_np)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        for i in range(3):
            out = TO.unet_forward(sd, cfg, x, torch.full((2, 1), float(g[f'nl/{i}'])))
            assert np.abs(out.numpy() - g[f'eps/{i}']).max() <= 2e-5
    sched = dict(schedule='linear', n_timestep=10, linear_start=1e-4, linear_end=2e-2)
    tab = O.schedule_tables(sched)
    np.testing.assert_allclose(tab['sqrt_alphas_cumprod_prev_f64'], g['sqrt_alphas_cumprod_prev'], rtol=0, atol=0)
    cond, noise = torch.from_numpy(g['cond']), torch.from_numpy(g['noise'])
    img, traj = TO.p_sample_loop(sd, cfg, tab, cond, noise, return_trajectory=True)
    frames = g['frames']                                           # x_in, then every x_t (inter = 1 at T = 10)
    assert frames.shape[0] == 11 and np.array_equal(frames[0], cond[0].numpy())
    for k in range(10):
        assert np.abs(traj[k][0].numpy() - frames[k + 1]).max() <= 5e-5, k
    assert np.abs(img[0].numpy() - frames[-1]).max() <= 5e-5
    hr, sr, nz = (torch.from_numpy(g[k]) for k in ('hr', 'sr', 'loss_noise'))
    gamma = torch.FloatTensor(g['gamma']).view(-1, 1)
    with torch.no_grad():
        x_noisy = O.q_sample(hr, gamma.view(-1, 1, 1, 1), nz)       # x_start is the image itself (diffusion.py:225)
        rec = TO.unet_forward(sd, cfg, torch.cat([sr, x_noisy], 1), gamma)
        loss = TO.charbonnier(nz, rec)
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))


def test_gdp_oracle_matches_reference(golden_dir):
    """oracle/gdp_oracle.py vs the reference's own model/gdp_modules (tests/golden/gdp.npz): the guided-diffusion UNet
    (scale-shift-norm and up/down ResBlocks, multi-head attention), the x_0-predicting sampler, the MSE loss."""
    from oracle import gdp_oracle as GO
    from fastdiffsr_amd.arch import param_schema
    g = _load(golden_dir, 'gdp.npz')
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 2), attn_res=(2, 4),
                     res_blocks=1, dropout=0.1, image_size=32, variant='gdp')
    sd_np = synth_state_dict(cfg, 13)
    assert state_dict_sha256(sd_np) == str(g['weights_sha256'])
    assert list(param_schema(cfg).keys()) == [str(k) for k in g['keys']]          # the reference's state_dict() order
    sd = O.to_torch_sd(sd_np)
    x = torch.from_numpy(g['x'])
    with torch.no_grad():
        for i in range(3):
            out = GO.unet_forward(sd, cfg, x, torch.from_numpy(g[f't/{i}']))
            assert np.abs(out.numpy() - g[f'rec/{i}']).max() <= 2e-5 * max(1.0, np.abs(g[f'rec/{i}']).max())
    sched = dict(schedule='linear', n_timestep=8, linear_start=1e-4, linear_end=2e-2)
    tab = O.schedule_tables(sched)
    cond, noise = torch.from_numpy(g['cond']), torch.from_numpy(g['noise'])
    img, traj = GO.p_sample_loop(sd, cfg, tab, cond, noise, return_trajectory=True)
    frames = g['frames']
    assert np.abs(frames[0:1] - cond.numpy()).max() == 0.0
    for k in range(8):
        assert np.abs(traj[k].numpy() - frames[k + 1:k + 2]).max() <= 5e-5, k
    with torch.no_grad():
        loss = GO.p_losses(sd, cfg, tab, torch.from_numpy(g['hr']), torch.from_numpy(g['sr']), torch.from_numpy(g['loss_t']),
                           torch.from_numpy(g['loss_noise']))
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * abs(float(g['loss']))


def test_oracle_wiring_table_is_its_own_and_agrees_with_the_products():
    """oracle/layers.py derives the module list from the reference's constructor on its own; the product's arch.build_layers is a
    second, independent derivation.  They must agree for every architecture the tests use (the goldens pin both)."""
    from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, build_layers
    from oracle import layers as OL
    import oracle.fdsr_oracle as O
    assert O.build_layers is OL.wiring
    cfgs = [UNetConfig(**FASTDIFFSR_UNET),
            UNetConfig(inner_channel=32, channel_mults=(1, 2, 4, 4), attn_res=(16,), res_blocks=2, image_size=32),
            UNetConfig(inner_channel=32, channel_mults=(1, 2), attn_res=(16,), res_blocks=1, image_size=16),
            UNetConfig(inner_channel=64, channel_mults=(1, 2, 4, 8, 8), attn_res=(16,), res_blocks=2, image_size=256, variant='ddpm'),
            UNetConfig(inner_channel=32, channel_mults=(1, 2, 4), attn_res=(8, 16), res_blocks=1, image_size=32, variant='tesr')]
    for cfg in cfgs:
        a, b = OL.wiring(cfg), build_layers(cfg)
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert (x.kind, x.name, x.cin, x.cout, bool(x.with_attn), x.cskip) == (y.kind, y.name, y.cin, y.cout, bool(y.with_attn), y.cskip)
