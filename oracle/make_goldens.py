"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (never on the
GPU box).  Imports FastDiffSR/model/fastdiffsr_modules/{diffusion,unet}.py
with stub modules for the two unused imports the image lacks (torchvision,
thop), loads the portable synthetic weights (fastdiffsr_amd/synth.py) into the
reference's nn.Modules, injects fixed noise by patching torch.randn /
torch.randn_like (SURVEY.md 8c recipe) and stores inputs-by-seed + expected
outputs.  Fixtures are DATA (arrays); no reference source text is stored.

Usage:  python oracle/make_goldens.py            (writes tests/golden/)
"""
import hashlib
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference/FastDiffSR'
OUT = os.path.join(ROOT, 'tests', 'golden')

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL, SCHEDULE_BUFFERS  # noqa: E402
from fastdiffsr_amd.synth import synth_state_dict, state_dict_sha256, synth_inputs  # noqa: E402


def import_reference():
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    tvm.vgg19 = None
    tv.models = tvm
    thop = types.ModuleType('thop')
    thop.profile = None
    thop.clever_format = None
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tvm)
    sys.modules.setdefault('thop', thop)
    # tesr_modules/unet.py:9 imports three timm helpers for its (unused) SwinIR classes; timm is not in this image
    timm, tm, tml = (types.ModuleType(n) for n in ('timm', 'timm.models', 'timm.models.layers'))
    tml.DropPath = tml.to_2tuple = tml.trunc_normal_ = None
    timm.models, tm.layers = tm, tml
    for name, mod in (('timm', timm), ('timm.models', tm), ('timm.models.layers', tml)):
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REF)
    from model.fastdiffsr_modules import diffusion, unet
    return diffusion, unet


def build_ref(diffusion, unet, cfg: UNetConfig, sched: dict, seed=0):
    net = unet.UNet(in_channel=cfg.in_channel, out_channel=cfg.out_channel, norm_groups=cfg.norm_groups,
                    inner_channel=cfg.inner_channel, channel_mults=list(cfg.channel_mults),
                    attn_res=list(cfg.attn_res), res_blocks=cfg.res_blocks, dropout=cfg.dropout,
                    image_size=cfg.image_size)
    G = diffusion.GaussianDiffusion(net, image_size=cfg.image_size, channels=3, loss_type='l1',
                                    conditional=True, schedule_opt=sched, scale=4)
    G.set_loss('cpu')
    G.set_new_noise_schedule(sched, 'cpu')
    sd = synth_state_dict(cfg, seed)
    ref_keys = [k for k in net.state_dict().keys()]
    assert ref_keys == list(sd.keys()), 'schema order mismatch vs reference state_dict'
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    G.eval()
    return G, net, sd


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ref_sample_loop(G, cond, noise):
    """Batched oracle = manual p_sample loop + res2img (bypasses the B>=2 crash)."""
    T = G.num_timesteps
    it = iter(range(1, T))
    traj = []
    with torch.no_grad(), mock.patch.object(torch, 'randn_like', lambda x: noise[next(it)]):
        img = noise[0]
        for t in reversed(range(T)):
            img = G.p_sample(img, t, condition_x=cond)
            traj.append(img.clone())
        out = G.res2img(img, cond)
    return out, traj


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    diffusion, unet = import_reference()

    # ---- (i) schedule tables: every branch the reference implements ------------
    sched = {}
    cases = [('linear_cosine', 20, 1e-6, 1e-2), ('linear_cosine', 10, 1e-6, 1e-2),
             ('linear', 2000, 1e-6, 1e-2), ('quad', 50, 1e-4, 2e-2), ('warmup10', 40, 1e-4, 2e-2),
             ('warmup50', 40, 1e-4, 2e-2), ('const', 16, 1e-4, 2e-2), ('jsd', 16, 1e-4, 2e-2),
             ('cosine', 100, 1e-4, 2e-2)]
    for name, T, ls, le in cases:
        b = diffusion.make_beta_schedule(name, T, ls, le)
        b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
        sched[f'betas/{name}/{T}'] = np.asarray(b, dtype=np.float64)
    for T in (20, 10):
        G = diffusion.GaussianDiffusion(None, image_size=256, conditional=True)
        G.set_new_noise_schedule(dict(schedule='linear_cosine', n_timestep=T, linear_start=1e-6, linear_end=1e-2), 'cpu')
        for k in SCHEDULE_BUFFERS:
            sched[f'buf/{T}/{k}'] = getattr(G, k).numpy()
        sched[f'buf/{T}/sqrt_alphas_cumprod_prev_f64'] = np.asarray(G.sqrt_alphas_cumprod_prev, dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'schedule.npz'), **sched)

    # ---- (ii) small real-tensor UNet golden: inner=32 ---------------------------
    cfg_s = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                       channel_mults=(1, 2, 4, 4), attn_res=(16,), res_blocks=2, dropout=0.2, image_size=32)
    Gs, net_s, sd_s = build_ref(diffusion, unet, cfg_s, FASTDIFFSR_SCHEDULE_VAL, seed=7)
    g = torch.Generator().manual_seed(11)
    x_s = torch.randn(2, 6, 32, 32, generator=g)
    small = {'x': x_s.numpy(), 'weights_sha256': np.array(state_dict_sha256(sd_s))}
    with torch.no_grad():
        for i, nl in enumerate((0.5, 6.634494e-07, 0.99)):
            nlv = torch.full((2, 1), nl, dtype=torch.float32)
            small[f'eps/{i}'] = net_s(x_s, nlv).numpy()
            small[f'nl/{i}'] = nlv.numpy()
        # per-sample different noise levels (training-style call)
        nlv = torch.tensor([[0.3], [0.8]], dtype=torch.float32)
        small['eps/3'] = net_s(x_s, nlv).numpy()
        small['nl/3'] = nlv.numpy()
        # sub-module goldens (wiring checks for the restatement)
        t = net_s.noise_level_mlp(nlv)
        small['t_mlp'] = t.numpy()
        small['posenc'] = net_s.noise_level_mlp[0](nlv).numpy()
        h = net_s.downs[0](x_s)
        small['downs0'] = h.numpy()
        small['downs1'] = net_s.downs[1](h, t).numpy()
        xm = torch.randn(2, 128, 4, 4, generator=g)
        small['mid_in'] = xm.numpy()
        small['mid0'] = net_s.mid[0](xm, t).numpy()
        small['clam'] = net_s.mid[0].ca(xm).numpy()
        small['slam'] = net_s.mid[0].sa(xm).numpy()
        xu = torch.randn(2, 128, 4, 4, generator=g)
        up_idx = [i for i, m in enumerate(net_s.ups) if isinstance(m, unet.Upsample)][0]
        dn_idx = [i for i, m in enumerate(net_s.downs) if isinstance(m, unet.Downsample)][0]
        small['up_in'] = xu.numpy()
        small['up_idx'] = np.array(up_idx)
        small['up_out'] = net_s.ups[up_idx](xu).numpy()
        xd = torch.randn(2, 32, 16, 16, generator=g)
        small['down_in'] = xd.numpy()
        small['down_idx'] = np.array(dn_idx)
        small['down_out'] = net_s.downs[dn_idx](xd).numpy()
    np.savez_compressed(os.path.join(OUT, 'unet_small.npz'), **small)

    # ---- (iii) full-width UNet (inner=64) forward with portable weights --------
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    G, net, sd = build_ref(diffusion, unet, cfg, FASTDIFFSR_SCHEDULE_VAL, seed=0)
    full = {'weights_sha256': np.array(state_dict_sha256(sd))}
    g = torch.Generator().manual_seed(21)
    x64 = torch.randn(1, 6, 64, 64, generator=g)
    x32 = torch.randn(2, 6, 32, 32, generator=g)
    full['x64_sha256'] = np.array(sha(x64.numpy()))
    full['x32_sha256'] = np.array(sha(x32.numpy()))
    with torch.no_grad():
        full['eps64'] = net(x64, torch.full((1, 1), 0.5)).numpy()
        full['eps32'] = net(x32, torch.tensor([[0.0209801132], [0.9919746]])).numpy()
    np.savez_compressed(os.path.join(OUT, 'unet_full.npz'), **full)

    # ---- (iv) 20-step trajectories ----------------------------------------------
    traj = {}
    cond, noise = synth_inputs(2, 32, 32, 20)
    traj['cond32_sha256'] = np.array(sha(cond.numpy()))
    traj['noise32_sha256'] = np.array(sha(noise.numpy()))
    out, xs = ref_sample_loop(G, cond, noise)
    traj['traj32'] = torch.stack(xs).numpy()          # [20,2,3,32,32], x_t after step t=19..0
    traj['out32'] = out.numpy()
    # B=1 through the reference's own p_sample_loop (continous=True) -- the real entry point
    it = iter(range(1, 20))
    c1, n1 = cond[:1], noise[:, :1]
    with mock.patch.object(torch, 'randn', lambda *a, **k: n1[0]), \
            mock.patch.object(torch, 'randn_like', lambda x: n1[next(it)]), \
            mock.patch.object(diffusion, 'tqdm', lambda it_, **k: it_):
        ret = G.super_resolution(c1, continous=True)
    traj['continous32_b1'] = ret.numpy()               # [8,3,32,32]
    it = iter(range(1, 20))
    with mock.patch.object(torch, 'randn', lambda *a, **k: n1[0]), \
            mock.patch.object(torch, 'randn_like', lambda x: n1[next(it)]), \
            mock.patch.object(diffusion, 'tqdm', lambda it_, **k: it_):
        traj['final32_b1'] = G.super_resolution(c1, continous=False).numpy()
    cond64, noise64 = synth_inputs(1, 64, 64, 20)
    out64, _ = ref_sample_loop(G, cond64, noise64)
    traj['out64'] = out64.numpy()
    traj['cond64_sha256'] = np.array(sha(cond64.numpy()))
    np.savez_compressed(os.path.join(OUT, 'sample_loop.npz'), **traj)

    # ---- (v) training loss (a20), dropout off ------------------------------------
    tr = {}
    g = torch.Generator().manual_seed(31)
    hr = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    sr = (hr + 0.2 * torch.randn(2, 3, 32, 32, generator=g)).clamp(-1, 1)
    nz = torch.randn(2, 3, 32, 32, generator=g)
    t_fixed = 7
    lo, hi = G.sqrt_alphas_cumprod_prev[t_fixed - 1], G.sqrt_alphas_cumprod_prev[t_fixed]
    gam = np.array([lo + 0.25 * (hi - lo), lo + 0.75 * (hi - lo)])
    with mock.patch.object(np.random, 'randint', lambda a, b: t_fixed), \
            mock.patch.object(np.random, 'uniform', lambda a, b, size: gam):
        Gs_eval = G
        Gs_eval.eval()
        loss = Gs_eval({'HR': hr, 'SR': sr}, noise=nz)
    tr.update(hr=hr.numpy(), sr=sr.numpy(), noise=nz.numpy(), gamma=gam.astype(np.float64),
              loss=np.array(loss.item(), dtype=np.float64),
              img2res=G.img2res(hr, sr).numpy(), res2img=G.res2img(nz, sr).numpy())
    np.savez_compressed(os.path.join(OUT, 'train_loss.npz'), **tr)

    # ---- (vi) val-loop helpers: the reference's own tensor2img / calculate_psnr --------------------
    for name in ('cv2', 'skimage', 'skimage.measure', 'lpips', 'matplotlib', 'matplotlib.pyplot', 'torchvision.utils',
                 'torchvision.transforms', 'core.PerceptualSimilarity'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['skimage.measure'].compare_mse = None
    sys.modules['skimage'].io = None
    sys.modules['skimage'].data = None
    sys.modules['torchvision.utils'].make_grid = None
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    import core
    core.PerceptualSimilarity = sys.modules['core.PerceptualSimilarity']
    from core import metrics as ref_metrics
    g = torch.Generator().manual_seed(41)
    t = torch.randn(3, 24, 40, generator=g) * 0.8
    t[0, 0, :8] = torch.tensor([-1.5, -1.0, 1.0, 1.5, 0.0, 1 / 255.0 - 1.0, 0.003921569, -0.9960785])
    img = ref_metrics.tensor2img(t.clone())
    img2 = ref_metrics.tensor2img((t + 0.05 * torch.randn(3, 24, 40, generator=g)).clone())
    gray = ref_metrics.tensor2img(t[:1].clone())
    np.savez_compressed(os.path.join(OUT, 'metrics.npz'), t=t.numpy(), img=img, img2=img2, gray=gray,
                        psnr=np.array(ref_metrics.calculate_psnr(img, img2), dtype=np.float64),
                        psnr_same=np.array(ref_metrics.calculate_psnr(img, img), dtype=np.float64))

    # ---- (vii) PIL bicubic (the LR -> SR conditioning image, prepare_data_mfe_dm.py:17-40) ---------
    from PIL import Image
    import PIL
    rng = np.random.default_rng(77)
    bic = {}
    for name, (h, w, H, W) in {'x4': (64, 64, 256, 256), 'x8': (32, 32, 256, 256), 'ragged': (24, 40, 96, 160)}.items():
        a = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        if name == 'x4':
            a[:8] = np.array([0, 255, 0], np.uint8)
            a[8:16, ::2] = 255   # saturating edges / ringing
        bic[name + '/lr'] = a
        bic[name + '/sr'] = np.asarray(Image.fromarray(a).resize((W, H), Image.BICUBIC))
    bic['pil_version'] = np.array(PIL.__version__)
    np.savez_compressed(os.path.join(OUT, 'bicubic.npz'), **bic)

    # ---- (viii) SR3 sibling (model/ddpm_modules), SURVEY 8f-4 ------------------------------------
    from model.ddpm_modules import diffusion as sr3_diffusion, unet as sr3_unet
    cfg3 = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                      attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='ddpm')
    net3 = sr3_unet.UNet(in_channel=6, out_channel=3, norm_groups=32, inner_channel=32, channel_mults=[1, 2, 2, 4],
                         attn_res=[8], res_blocks=1, dropout=0.2, image_size=32)
    sd3 = synth_state_dict(cfg3, 5)
    assert list(net3.state_dict().keys()) == list(sd3.keys()), 'SR3 schema order mismatch'
    net3.load_state_dict({k: torch.from_numpy(v) for k, v in sd3.items()}, strict=True)
    sched3 = dict(schedule='linear', n_timestep=12, linear_start=1e-4, linear_end=2e-2)
    G3 = sr3_diffusion.GaussianDiffusion(net3, image_size=32, channels=3, loss_type='l1', conditional=True, schedule_opt=sched3)
    G3.set_loss('cpu')
    G3.set_new_noise_schedule(sched3, 'cpu')
    G3.eval()
    g = torch.Generator().manual_seed(51)
    x3 = torch.randn(2, 6, 32, 32, generator=g)
    s3 = {'weights_sha256': np.array(state_dict_sha256(sd3)), 'x': x3.numpy()}
    with torch.no_grad():
        s3['eps_t'] = net3(x3, torch.tensor([3, 999])).numpy()
        s3['eps_t0'] = net3(x3, torch.tensor([0, 0])).numpy()
        xa = torch.randn(2, 64, 8, 8, generator=g)
        attn_name = [n for n, m in net3.named_modules() if isinstance(m, sr3_unet.SelfAttention)][0]
        s3['attn_in'] = xa.numpy()
        s3['attn_name'] = np.array(attn_name)
        s3['attn_out'] = dict(net3.named_modules())[attn_name](xa).numpy()
    cond3 = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    noise3 = torch.randn(13, 2, 3, 32, 32, generator=g)
    it3 = iter(range(0, 13))      # randn(shape) for x_T, then one noise_like draw per step (t = 11..0)
    with torch.no_grad(), mock.patch.object(torch, 'randn', lambda *a, **k: noise3[next(it3)]), \
            mock.patch.object(sr3_diffusion, 'tqdm', lambda it_, **k: it_):
        ret3 = G3.p_sample_loop(cond3, continous=True)
    s3['cond'] = cond3.numpy()
    s3['noise'] = noise3.numpy()
    s3['continous'] = ret3.numpy()
    for k in SCHEDULE_BUFFERS:
        s3[f'buf/{k}'] = getattr(G3, k).numpy()
    np.savez_compressed(os.path.join(OUT, 'sr3.npz'), **s3)

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))



def train_goldens():
    """(x) one optimisation step of the reference itself (model.py:47-57: zero_grad, loss / (b*c*h*w), backward,
    torch.optim.Adam step), dropout off, fixed t / gamma / noise: loss, three named gradients in full, a
    (sum, sum of squares) pair for EVERY gradient, and the three tensors after the step.  Pins oracle.train_step."""
    from unittest import mock
    diffusion, unet = import_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, 0)
    net = unet.UNet(in_channel=6, out_channel=3, norm_groups=32, inner_channel=64, channel_mults=[1, 2, 4, 4],
                    attn_res=[16], res_blocks=2, dropout=0.2, image_size=256)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    G = diffusion.GaussianDiffusion(net, image_size=256, channels=3, loss_type='l1', conditional=True,
                                    schedule_opt=dict(FASTDIFFSR_SCHEDULE_VAL))
    G.set_loss('cpu')
    G.set_new_noise_schedule(dict(FASTDIFFSR_SCHEDULE_VAL), 'cpu')
    G.eval()                                            # dropout off; everything else as in training
    tl = np.load(os.path.join(OUT, 'train_loss.npz'))   # same inputs / draws as the loss golden
    hr, sr, nz = (torch.from_numpy(tl[k]) for k in ('hr', 'sr', 'noise'))
    gam = tl['gamma']
    lr = 1e-4                                           # config/sr_fastdiffsr_train_64_256.json: train.optimizer.lr
    opt = torch.optim.Adam(list(G.parameters()), lr=lr)
    opt.zero_grad()
    with mock.patch.object(np.random, 'randint', lambda a, b: 7), \
            mock.patch.object(np.random, 'uniform', lambda a, b, size: gam):
        l_pix = G({'HR': hr, 'SR': sr}, noise=nz)
    b, c, h, w = hr.shape
    l_pix = l_pix.sum() / int(b * c * h * w)
    l_pix.backward()
    named = dict(G.named_parameters())
    keys3 = ['downs.0.weight', 'mid.0.sa.conv1.weight', 'final_conv.block.3.bias']
    out = {'l_pix': np.array(l_pix.item(), dtype=np.float64), 'lr': np.array(lr)}
    names, stats = [], []
    for k, p in named.items():
        kk = k[len('denoise_fn.'):]
        if p.grad is None:
            continue
        names.append(kk)
        g64 = p.grad.double()
        stats.append([g64.sum().item(), (g64 * g64).sum().item()])
    out['grad_keys'] = np.array(names)
    out['grad_stats'] = np.array(stats, dtype=np.float64)
    for k in keys3:
        out['grad/' + k] = named['denoise_fn.' + k].grad.numpy().copy()
    opt.step()
    for k in keys3:
        out['after/' + k] = named['denoise_fn.' + k].detach().numpy().copy()
    out['n_params_without_grad'] = np.array(sum(1 for p in named.values() if p.grad is None))
    np.savez_compressed(os.path.join(OUT, 'train_step.npz'), **out)
    print('wrote train_step.npz: l_pix', l_pix.item(), 'tensors with grad', len(names), 'without', int(out['n_params_without_grad']))


def sr3_train_goldens():
    """(xiii) one optimisation step of the reference's SR3 sibling (model/ddpm_modules; DDPM.optimize_parameters, model.py:47-57:
    zero_grad, l_pix = netG(data) -> p_losses (ddpm_modules/diffusion.py:279-297), l_pix.sum() / (b*c*h*w), backward, Adam step)
    on the SR3 test network of sr3.npz (inner 32, mults 1-2-2-4, SelfAttention at 8 x 8 and in mid[0]), dropout off, fixed t and
    noise: the loss, a (sum, sum of squares) pair for EVERY gradient, six named gradients in full (a convolution, the attention's
    qkv / out / norm, the time MLP, a per-block Linear) and those tensors after the step.  Pins autograd over oracle/sr3_oracle.py,
    which the GPU tests then hold the engine's backward against."""
    from unittest import mock
    import_reference()
    from model.ddpm_modules import diffusion as sr3_diffusion, unet as sr3_unet
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg3 = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                      attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='ddpm')
    net3 = sr3_unet.UNet(in_channel=6, out_channel=3, norm_groups=32, inner_channel=32, channel_mults=[1, 2, 2, 4],
                         attn_res=[8], res_blocks=1, dropout=0.2, image_size=32)
    sd3 = synth_state_dict(cfg3, 5)
    net3.load_state_dict({k: torch.from_numpy(v) for k, v in sd3.items()}, strict=True)
    sched3 = dict(schedule='linear', n_timestep=12, linear_start=1e-4, linear_end=2e-2)
    G3 = sr3_diffusion.GaussianDiffusion(net3, image_size=32, channels=3, loss_type='l1', conditional=True, schedule_opt=sched3)
    G3.set_loss('cpu')
    G3.set_new_noise_schedule(sched3, 'cpu')
    G3.eval()                                           # dropout off; everything else as in training
    g = torch.Generator().manual_seed(77)
    hr = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    sr = (hr + 0.1 * torch.randn(2, 3, 32, 32, generator=g)).clamp(-1, 1)
    nz = torch.randn(2, 3, 32, 32, generator=g)
    t = torch.tensor([3, 9])
    lr = 1e-4
    opt = torch.optim.Adam(list(G3.parameters()), lr=lr)
    opt.zero_grad()
    with mock.patch.object(torch, 'randint', lambda *a, **k: t):
        l_pix = G3({'HR': hr, 'SR': sr}, noise=nz)
    b, c, h, w = hr.shape
    l_pix = l_pix.sum() / int(b * c * h * w)
    l_pix.backward()
    named = dict(G3.named_parameters())
    attn = [n for n, m in net3.named_modules() if isinstance(m, sr3_unet.SelfAttention)][0]
    full = ['downs.0.weight', attn + '.qkv.weight', attn + '.out.weight', attn + '.norm.weight', 'time_mlp.1.weight',
            'downs.1.res_block.mlp.1.weight']
    out = {'hr': hr.numpy(), 'sr': sr.numpy(), 'noise': nz.numpy(), 't': t.numpy(), 'lr': np.array(lr),
           'l_pix': np.array(l_pix.item(), dtype=np.float64), 'weights_sha256': np.array(state_dict_sha256(sd3)),
           'full_keys': np.array(full)}
    names, stats = [], []
    for k, p_ in named.items():
        if p_.grad is None:
            continue
        names.append(k[len('denoise_fn.'):])
        g64 = p_.grad.double()
        stats.append([g64.sum().item(), (g64 * g64).sum().item()])
    out['grad_keys'] = np.array(names)
    out['grad_stats'] = np.array(stats, dtype=np.float64)
    for k in full:
        out['grad/' + k] = named['denoise_fn.' + k].grad.numpy().copy()
    opt.step()
    for k in full:
        out['after/' + k] = named['denoise_fn.' + k].detach().numpy().copy()
    out['n_params_without_grad'] = np.array(sum(1 for p_ in named.values() if p_.grad is None))
    np.savez_compressed(os.path.join(OUT, 'sr3_train_step.npz'), **out)
    print('wrote sr3_train_step.npz: l_pix', l_pix.item(), 'tensors with grad', len(names), 'without', int(out['n_params_without_grad']))


def init_goldens():
    """(xii) the reference's own `init_weights(netG, 'orthogonal')` (model/networks.py:46-75, called by define_G in the
    train phase, :113-115) under torch.manual_seed(1234): a (sum, sum of squares) pair for every tensor of the small
    inner-32 UNet, and two tensors in full.  Pins fastdiffsr_amd.networks.init_weights (same RNG consumption order)."""
    diffusion, unet = import_reference()
    from model import networks as ref_networks
    torch.manual_seed(1234)
    net = unet.UNet(in_channel=6, out_channel=3, norm_groups=32, inner_channel=32, channel_mults=[1, 2, 4, 4],
                    attn_res=[16], res_blocks=2, dropout=0.2, image_size=32)
    G = diffusion.GaussianDiffusion(net, image_size=32, channels=3, loss_type='l1', conditional=True,
                                    schedule_opt=dict(FASTDIFFSR_SCHEDULE_VAL))
    ref_networks.init_weights(G, init_type='orthogonal')
    sd = net.state_dict()
    keys = list(sd.keys())
    stats = np.array([[v.double().sum().item(), (v.double() ** 2).sum().item()] for v in sd.values()], dtype=np.float64)
    out = {'keys': np.array(keys), 'stats': stats, 'seed': np.array(1234)}
    for k in ('downs.0.weight', 'mid.0.ca.fc2.weight', 'ups.4.res_block.noise_func.noise_func.0.weight'):
        out['full/' + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'init_weights.npz'), **out)
    print('wrote init_weights.npz:', len(keys), 'tensors')


def gdp_goldens():
    """(xiii) GDP sibling (model/gdp_modules): the guided-diffusion UNet as define_G builds it (scale-shift-norm ResBlocks,
    up/down ResBlocks, multi-head attention), small: model_channels 64, mults (1, 2, 2), one ResBlock per level, attention
    at downsample rates 2 and 4.  UNet forwards at three timesteps, the sampler (continous=True) at T=8, the MSE loss."""
    from unittest import mock
    import_reference()
    # gdp_modules/diffusion.py:9 imports torchvision.transforms.functional (unused on this path): stub it like torchvision itself
    tvt, tvf = types.ModuleType('torchvision.transforms'), types.ModuleType('torchvision.transforms.functional')
    tvt.functional = tvf
    sys.modules.setdefault('torchvision.transforms', tvt)
    sys.modules.setdefault('torchvision.transforms.functional', tvf)
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    from model.gdp_modules import diffusion as gdiff, unet as gunet
    from fastdiffsr_amd.arch import param_schema
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 2), attn_res=(2, 4),
                     res_blocks=1, dropout=0.1, image_size=32, variant='gdp')
    net = gunet.UNet(image_size=32, in_channel=6, model_channels=64, out_channel=3, res_blocks=1, attention_resolutions=(2, 4),
                     dropout=0.1, channel_mults=(1, 2, 2), inner_channel=64, norm_groups=32, attn_res=(16,))
    sd = synth_state_dict(cfg, 13)
    ref_keys = list(net.state_dict().keys())
    assert ref_keys == list(sd.keys()), 'GDP schema order mismatch'
    assert [tuple(v.shape) for v in net.state_dict().values()] == [tuple(v.shape) for v in sd.values()]
    assert list(param_schema(cfg).keys()) == ref_keys
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    sched = dict(schedule='linear', n_timestep=8, linear_start=1e-4, linear_end=2e-2)
    G = gdiff.GaussianDiffusion(net, image_size=32, channels=3, loss_type='l1', conditional=True, schedule_opt=sched)
    G.set_loss('cpu')
    G.set_new_noise_schedule(sched, 'cpu')
    G.eval()
    g = torch.Generator().manual_seed(83)
    x = torch.randn(2, 6, 32, 32, generator=g)
    out = {'weights_sha256': np.array(state_dict_sha256(sd)), 'x': x.numpy(), 'keys': np.array(ref_keys)}
    with torch.no_grad():
        for i, t in enumerate(((0, 7), (3, 3), (999, 500))):
            out[f'rec/{i}'] = net(x, torch.tensor(t, dtype=torch.long)).numpy()
            out[f't/{i}'] = np.array(t, dtype=np.int64)
    cond, noise = synth_inputs(1, 32, 32, 9, cond_seed=91, noise_seed=92)       # x_T + one draw per step (t = 0 masked)
    draws = iter([noise[k] for k in range(9)])
    with torch.no_grad(), mock.patch.object(torch, 'randn', lambda *a, **k: next(draws)):
        frames = G.super_resolution(cond, continous=True)
    out['cond'] = cond.numpy()
    out['noise'] = noise.numpy()
    out['frames'] = frames.numpy()              # [1 + 8, 3, 32, 32]: x_in, then x_t after every step (sample_inter = 1)
    hr = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    sr = (hr + 0.2 * torch.randn(2, 3, 32, 32, generator=g)).clamp(-1, 1)
    nz = torch.randn(2, 3, 32, 32, generator=g)
    tt = torch.tensor([5, 2], dtype=torch.long)
    with torch.no_grad(), mock.patch.object(torch, 'randint', lambda *a, **k: tt):
        loss = G({'HR': hr, 'SR': sr, 'LR': sr}, noise=nz)
    out.update({'hr': hr.numpy(), 'sr': sr.numpy(), 'loss_noise': nz.numpy(), 'loss_t': tt.numpy(), 'loss': np.array(loss.item())})
    np.savez_compressed(os.path.join(OUT, 'gdp.npz'), **out)
    print('wrote gdp.npz: loss', loss.item(), 'frames', frames.shape)


def gdp_train_goldens():
    """(xiv) one optimisation step of the reference's GDP sibling (model/gdp_modules; DDPM.optimize_parameters, model.py:47-57 ->
    gdp_modules/diffusion.py:277-299: the summed MSE between UNet(cat[q_sample(HR, t), SR], t) and HR) on the GDP test network of
    gdp.npz (model_channels 64, mults 1-2-2, up/down ResBlocks, 2- and 2-head attention at 16 x 16 and 8 x 8), dropout off, fixed t
    and noise: the loss, a (sum, sum of squares) pair for EVERY gradient, eleven named gradients in full (the input conv, a down ResBlock's conv,
    an up ResBlock's norms / conv bias / scale-shift Linear bias, the attention's qkv / proj_out / norm, the time MLP) and those tensors after the
    step.  Pins autograd over oracle/gdp_oracle.py, which the GPU tests then hold the engine's backward against."""
    from unittest import mock
    import_reference()
    tvt, tvf = types.ModuleType('torchvision.transforms'), types.ModuleType('torchvision.transforms.functional')
    tvt.functional = tvf
    sys.modules.setdefault('torchvision.transforms', tvt)
    sys.modules.setdefault('torchvision.transforms.functional', tvf)
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    from model.gdp_modules import diffusion as gdiff, unet as gunet
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 2), attn_res=(2, 4),
                     res_blocks=1, dropout=0.1, image_size=32, variant='gdp')
    net = gunet.UNet(image_size=32, in_channel=6, model_channels=64, out_channel=3, res_blocks=1, attention_resolutions=(2, 4),
                     dropout=0.1, channel_mults=(1, 2, 2), inner_channel=64, norm_groups=32, attn_res=(16,))
    sd = synth_state_dict(cfg, 13)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    sched = dict(schedule='linear', n_timestep=8, linear_start=1e-4, linear_end=2e-2)
    G = gdiff.GaussianDiffusion(net, image_size=32, channels=3, loss_type='l1', conditional=True, schedule_opt=sched)
    G.set_loss('cpu')
    G.set_new_noise_schedule(sched, 'cpu')
    G.eval()                                            # dropout off; everything else as in training
    g = torch.Generator().manual_seed(84)
    hr = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    sr = (hr + 0.2 * torch.randn(2, 3, 32, 32, generator=g)).clamp(-1, 1)
    nz = torch.randn(2, 3, 32, 32, generator=g)
    t = torch.tensor([6, 1], dtype=torch.long)
    lr = 1e-4
    opt = torch.optim.Adam(list(G.parameters()), lr=lr)
    opt.zero_grad()
    with mock.patch.object(torch, 'randint', lambda *a, **k: t):
        l_pix = G({'HR': hr, 'SR': sr, 'LR': sr}, noise=nz)
    b, c, h, w = hr.shape
    l_pix = l_pix.sum() / int(b * c * h * w)
    l_pix.backward()
    named = dict(G.named_parameters())
    layers = gdp_layers_of(cfg)
    down = [L.name for L in layers if L.kind == 'res' and L.mode == 'down'][0]
    up = [L.name for L in layers if L.kind == 'res' and L.mode == 'up'][0]
    attn = [L.name for L in layers if L.kind == 'attn'][0]
    full = ['input_blocks.0.0.weight', down + '.in_layers.2.weight', up + '.in_layers.0.weight', up + '.in_layers.2.bias',
            up + '.emb_layers.1.bias', up + '.out_layers.0.weight', attn + '.qkv.weight', attn + '.proj_out.weight',
            attn + '.norm.weight', 'time_embed.0.weight', 'time_embed.2.bias']
    out = {'hr': hr.numpy(), 'sr': sr.numpy(), 'noise': nz.numpy(), 't': t.numpy(), 'lr': np.array(lr),
           'l_pix': np.array(l_pix.item(), dtype=np.float64), 'weights_sha256': np.array(state_dict_sha256(sd)),
           'full_keys': np.array(full)}
    names, stats = [], []
    for k, p_ in named.items():
        if p_.grad is None:
            continue
        names.append(k[len('denoise_fn.'):])
        g64 = p_.grad.double()
        stats.append([g64.sum().item(), (g64 * g64).sum().item()])
    out['grad_keys'] = np.array(names)
    out['grad_stats'] = np.array(stats, dtype=np.float64)
    for k in full:
        out['grad/' + k] = named['denoise_fn.' + k].grad.numpy().copy()
    opt.step()
    for k in full:
        out['after/' + k] = named['denoise_fn.' + k].detach().numpy().copy()
    out['n_params_without_grad'] = np.array(sum(1 for p_ in named.values() if p_.grad is None))
    np.savez_compressed(os.path.join(OUT, 'gdp_train_step.npz'), **out)
    print('wrote gdp_train_step.npz: l_pix', l_pix.item(), 'tensors with grad', len(names), 'without', int(out['n_params_without_grad']))


def gdp_layers_of(cfg):
    from fastdiffsr_amd.gdp.arch import gdp_layers
    return gdp_layers(cfg)


def tesr_goldens():
    """(xi) TESR sibling (model/tesr_modules): UNet forwards incl. the SelfAttention levels, the sampler
    (continous=True frames and the final image) and the Charbonnier training-loss value, from the reference itself."""
    from unittest import mock
    import_reference()
    from model.tesr_modules import diffusion as tdiff, unet as tunet
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                     attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='tesr')
    net = tunet.UNet(in_channel=6, out_channel=3, norm_groups=32, inner_channel=32, channel_mults=[1, 2, 2, 4],
                     attn_res=[8], res_blocks=1, dropout=0.2, image_size=32)
    sd = synth_state_dict(cfg, 9)
    assert list(net.state_dict().keys()) == list(sd.keys()), 'TESR schema order mismatch'
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    sched = dict(schedule='linear', n_timestep=10, linear_start=1e-4, linear_end=2e-2)
    G = tdiff.GaussianDiffusion(net, image_size=32, channels=3, loss_type='l1', conditional=True, schedule_opt=sched)
    G.set_loss('cpu')
    G.set_new_noise_schedule(sched, 'cpu')
    G.eval()
    g = torch.Generator().manual_seed(61)
    x = torch.randn(2, 6, 32, 32, generator=g)
    out = {'weights_sha256': np.array(state_dict_sha256(sd)), 'x': x.numpy()}
    with torch.no_grad():
        for i, nl in enumerate((0.5, 6.6e-7, 0.99)):
            out[f'eps/{i}'] = net(x, torch.full((2, 1), nl)).numpy()
            out[f'nl/{i}'] = np.array(nl, dtype=np.float32)
    out['sqrt_alphas_cumprod_prev'] = np.asarray(G.sqrt_alphas_cumprod_prev, dtype=np.float64)
    cond, noise = synth_inputs(1, 32, 32, 10, cond_seed=71, noise_seed=72)
    draws = iter([noise[k] for k in range(10)])
    with torch.no_grad(), mock.patch.object(torch, 'randn', lambda *a, **k: next(draws)), \
            mock.patch.object(torch, 'randn_like', lambda *a, **k: next(draws)):
        frames = G.super_resolution(cond, continous=True)
    out['cond'] = cond.numpy()
    out['noise'] = noise.numpy()
    out['frames'] = frames.numpy()              # [1 + kept steps, 3, 32, 32]: x_in, then x_t for t % inter == 0
    hr = torch.rand(2, 3, 32, 32, generator=g) * 2 - 1
    sr = (hr + 0.2 * torch.randn(2, 3, 32, 32, generator=g)).clamp(-1, 1)
    nz = torch.randn(2, 3, 32, 32, generator=g)
    t_fixed = 4
    lo, hi = G.sqrt_alphas_cumprod_prev[t_fixed - 1], G.sqrt_alphas_cumprod_prev[t_fixed]
    gam = np.array([lo + 0.25 * (hi - lo), lo + 0.75 * (hi - lo)])
    with torch.no_grad(), mock.patch.object(np.random, 'randint', lambda a, b: t_fixed), \
            mock.patch.object(np.random, 'uniform', lambda a, b, size: gam):
        loss = G({'HR': hr, 'SR': sr}, noise=nz)
    out.update(hr=hr.numpy(), sr=sr.numpy(), loss_noise=nz.numpy(), gamma=gam.astype(np.float64),
               loss=np.array(loss.item(), dtype=np.float64))
    np.savez_compressed(os.path.join(OUT, 'tesr.npz'), **out)
    print('wrote tesr.npz: frames', frames.shape, 'loss', loss.item())


def tesr_train_goldens():
    """(xiv) one optimisation step of the reference's TESR sibling (model/tesr_modules p_losses :224-250 with the Charbonnier MEAN as
    its 'l1' loss, then model.py:47-57: l_pix.sum() / (b*c*h*w), backward, Adam) on the TESR test network of tesr.npz, dropout off,
    same inputs / draws as the loss golden there: loss, (sum, sum of squares) of EVERY gradient, five tensors in full, the update."""
    from unittest import mock
    import_reference()
    from model.tesr_modules import diffusion as tdiff, unet as tunet
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 2, 4),
                     attn_res=(8,), res_blocks=1, dropout=0.2, image_size=32, variant='tesr')
    net = tunet.UNet(in_channel=6, out_channel=3, norm_groups=32, inner_channel=32, channel_mults=[1, 2, 2, 4],
                     attn_res=[8], res_blocks=1, dropout=0.2, image_size=32)
    sd = synth_state_dict(cfg, 9)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    sched = dict(schedule='linear', n_timestep=10, linear_start=1e-4, linear_end=2e-2)
    G = tdiff.GaussianDiffusion(net, image_size=32, channels=3, loss_type='l1', conditional=True, schedule_opt=sched)
    G.set_loss('cpu')
    G.set_new_noise_schedule(sched, 'cpu')
    G.eval()
    tg = np.load(os.path.join(OUT, 'tesr.npz'))
    hr, sr, nz = (torch.from_numpy(tg[k]) for k in ('hr', 'sr', 'loss_noise'))
    gam = tg['gamma']
    lr = 1e-4
    opt = torch.optim.Adam(list(G.parameters()), lr=lr)
    opt.zero_grad()
    with mock.patch.object(np.random, 'randint', lambda a, b: 4), mock.patch.object(np.random, 'uniform', lambda a, b, size: gam):
        l_pix = G({'HR': hr, 'SR': sr}, noise=nz)
    b, c, h, w = hr.shape
    l_pix = l_pix.sum() / int(b * c * h * w)
    l_pix.backward()
    named = dict(G.named_parameters())
    attn = [n for n, m in net.named_modules() if isinstance(m, tunet.SelfAttention)][0]
    full = ['downs.0.weight', attn + '.qkv.weight', attn + '.out.weight', 'noise_level_mlp.1.weight', 'final_conv.block.3.bias']
    out = {'lr': np.array(lr), 'l_pix': np.array(l_pix.item(), dtype=np.float64), 'weights_sha256': np.array(state_dict_sha256(sd)),
           'full_keys': np.array(full)}
    names, stats = [], []
    for k, p_ in named.items():
        if p_.grad is None:
            continue
        names.append(k[len('denoise_fn.'):])
        g64 = p_.grad.double()
        stats.append([g64.sum().item(), (g64 * g64).sum().item()])
    out['grad_keys'] = np.array(names)
    out['grad_stats'] = np.array(stats, dtype=np.float64)
    for k in full:
        out['grad/' + k] = named['denoise_fn.' + k].grad.numpy().copy()
    opt.step()
    for k in full:
        out['after/' + k] = named['denoise_fn.' + k].detach().numpy().copy()
    out['n_params_without_grad'] = np.array(sum(1 for p_ in named.values() if p_.grad is None))
    np.savez_compressed(os.path.join(OUT, 'tesr_train_step.npz'), **out)
    print('wrote tesr_train_step.npz: l_pix', l_pix.item(), 'tensors with grad', len(names), 'without', int(out['n_params_without_grad']))


def metric_goldens():
    """(x) the reference's own `ssim` / `calculate_ssim` / `calculate_ergas` (core/metrics.py:103-152) on fixed image pairs.
    They call cv2.getGaussianKernel, cv2.filter2D and skimage.measure.compare_mse, which this image lacks; the three are
    supplied here from their published definitions (OpenCV imgproc: the normalised Gaussian exp(-(i-(k-1)/2)^2 / 2 sigma^2);
    filter2D = correlation, whose border rule cannot matter because the reference keeps [5:-5, 5:-5] of an 11x11 filter;
    scikit-image 0.14-0.17 compare_mse = mean((a - b)^2) after a float conversion WITHOUT rescaling, accumulated in float64).
    So tests/golden/metrics_ssim.npz pins the reference's arithmetic around those primitives, not the primitives themselves."""
    import_reference()
    from numpy.lib.stride_tricks import sliding_window_view

    def getGaussianKernel(ksize, sigma):
        x = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
        k = np.exp(-(x * x) / (2.0 * sigma * sigma))
        return (k / k.sum()).reshape(ksize, 1)

    def filter2D(src, ddepth, kernel):
        assert ddepth == -1
        kh, kw = kernel.shape
        pad = ((kh // 2, kh // 2), (kw // 2, kw // 2)) + (((0, 0),) if src.ndim == 3 else ())
        a = np.pad(src, pad, mode='reflect')               # BORDER_REFLECT_101
        if src.ndim == 3:
            return np.stack([np.einsum('ijkl,kl->ij', sliding_window_view(a[..., c], kernel.shape), kernel)
                             for c in range(src.shape[2])], axis=-1)
        return np.einsum('ijkl,kl->ij', sliding_window_view(a, kernel.shape), kernel)

    def compare_mse(im1, im2):
        ft = np.result_type(im1.dtype, im2.dtype, np.float32)
        return np.mean(np.square(np.asarray(im1, dtype=ft) - np.asarray(im2, dtype=ft)), dtype=np.float64)

    for name in ('cv2', 'skimage', 'skimage.measure', 'lpips', 'matplotlib', 'matplotlib.pyplot', 'torchvision.utils',
                 'torchvision.transforms', 'core.PerceptualSimilarity'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['cv2'].getGaussianKernel = getGaussianKernel
    sys.modules['cv2'].filter2D = filter2D
    sys.modules['skimage.measure'].compare_mse = compare_mse
    sys.modules['skimage'].io = None
    sys.modules['skimage'].data = None
    sys.modules['torchvision.utils'].make_grid = None
    sys.modules['torchvision'].transforms = sys.modules['torchvision.transforms']
    sys.modules['torchvision'].utils = sys.modules['torchvision.utils']
    import core
    core.PerceptualSimilarity = sys.modules['core.PerceptualSimilarity']
    sys.modules.pop('core.metrics', None)
    from core import metrics as ref_metrics
    ref_metrics.compare_mse = compare_mse
    rng = np.random.default_rng(2024)
    out = {}
    yy, xx = np.mgrid[0:48, 0:64]
    base = (127 + 90 * np.sin(xx / 5.0)[..., None] * np.cos(yy / 7.0)[..., None] + rng.normal(0, 12, (48, 64, 3))).clip(0, 255)
    cases = {
        'noisy': (base, base + rng.normal(0, 9, base.shape)),
        'blur': (base, (base + np.roll(base, 1, 0) + np.roll(base, 1, 1) + np.roll(base, -1, 0)) / 4),
        'dark': (base, base * 0.6),
        'same': (base, base.copy()),
    }
    for k, (a, b) in cases.items():
        a8, b8 = a.clip(0, 255).astype(np.uint8), b.clip(0, 255).astype(np.uint8)
        out[f'{k}/a'], out[f'{k}/b'] = a8, b8
        out[f'{k}/ssim_rgb'] = np.array(ref_metrics.calculate_ssim(a8, b8), dtype=np.float64)
        out[f'{k}/ssim_gray'] = np.array(ref_metrics.calculate_ssim(a8[..., 0], b8[..., 0]), dtype=np.float64)
        out[f'{k}/ssim_1ch'] = np.array(ref_metrics.calculate_ssim(a8[..., :1], b8[..., :1]), dtype=np.float64)
        out[f'{k}/ergas4'] = np.array(ref_metrics.calculate_ergas(a8, b8, scale=4), dtype=np.float64)
        out[f'{k}/ergas8'] = np.array(ref_metrics.calculate_ergas(a8, b8, scale=8), dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, 'metrics_ssim.npz'), **out)
    print({k: float(v) for k, v in out.items() if v.ndim == 0})


def config_goldens():
    """(ix) the reference's own option parser (core/logger.py:21-94) on its ten fastdiffsr / ddpm configs:
    what `parse` returns, minus the timestamped `path` subtree.  Pins fastdiffsr_amd.config.load_config."""
    import argparse
    import json
    import tempfile
    import_reference()
    from core import logger as ref_logger
    out = {}
    cfg_dir = os.path.join(REF, 'config')
    cases = []
    for fam in ('fastdiffsr', 'ddpm'):
        for stem, phase in (('test_64_256', 'val'), ('test_32_256', 'val'), ('train_64_256', 'train'),
                            ('train_32_256', 'train'), ('infer_x4', 'val')):
            cases.append((f'sr_{fam}_{stem}.json', phase, None, False))
    cases.append(('sr_fastdiffsr_train_64_256.json', 'train', '0,1', True))    # -debug and a gpu list
    cwd, env = os.getcwd(), os.environ.get('CUDA_VISIBLE_DEVICES')
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                      # parse() creates experiment directories relative to the cwd
        try:
            for name, phase, gpu_ids, debug in cases:
                args = argparse.Namespace(config=os.path.join(cfg_dir, name), phase=phase, gpu_ids=gpu_ids, debug=debug,
                                          enable_wandb=False, log_wandb_ckpt=False, log_eval=False, log_infer=False)
                opt = ref_logger.parse(args)
                opt = json.loads(json.dumps(opt))
                opt.pop('path', None)
                out[f'{name}|{phase}|{gpu_ids}|{int(debug)}'] = opt
        finally:
            os.chdir(cwd)
            if env is None:
                os.environ.pop('CUDA_VISIBLE_DEVICES', None)
            else:
                os.environ['CUDA_VISIBLE_DEVICES'] = env
    with open(os.path.join(OUT, 'configs.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('wrote configs.json with', len(out), 'cases')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'configs':
        config_goldens()          # only tests/golden/configs.json
    elif len(sys.argv) > 1 and sys.argv[1] == 'train':
        train_goldens()           # only tests/golden/train_step.npz (reads train_loss.npz)
    elif len(sys.argv) > 1 and sys.argv[1] == 'metrics':
        metric_goldens()          # only tests/golden/metrics_ssim.npz
    elif len(sys.argv) > 1 and sys.argv[1] == 'gdp':
        gdp_goldens()             # only tests/golden/gdp.npz
    elif len(sys.argv) > 1 and sys.argv[1] == 'gdp_train':
        gdp_train_goldens()       # only tests/golden/gdp_train_step.npz
    elif len(sys.argv) > 1 and sys.argv[1] == 'init':
        init_goldens()            # only tests/golden/init_weights.npz
    elif len(sys.argv) > 1 and sys.argv[1] == 'tesr_train':
        tesr_train_goldens()      # only tests/golden/tesr_train_step.npz (reads tesr.npz)
    elif len(sys.argv) > 1 and sys.argv[1] == 'sr3_train':
        sr3_train_goldens()       # only tests/golden/sr3_train_step.npz
    elif len(sys.argv) > 1 and sys.argv[1] == 'tesr':
        tesr_goldens()            # only tests/golden/tesr.npz
    else:
        main()
        config_goldens()
        train_goldens()
        tesr_goldens()
        sr3_train_goldens()
        tesr_train_goldens()
        init_goldens()
        gdp_goldens()
        gdp_train_goldens()
        metric_goldens()
