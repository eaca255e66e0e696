"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (PyTorch-CPU functional ops, fp32, NCHW) of the FastDiffSR
20-step sampling hot path.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this file, and only as the checker /
the timed CPU baseline.  The product (fastdiffsr_amd/) never routes through it.

Parity pinning: the reference (Meng-333/FastDiffSR) has no tests, golden
vectors or fixtures of its own.  This restatement is
pinned against OUTPUTS OF THE REFERENCE ITSELF, produced by importing
/root/reference/FastDiffSR/model/fastdiffsr_modules in the build container
(oracle/make_goldens.py, committed) and stored as tests/golden/*.npz;
tests/test_oracle_golden.py checks every function below against them.

The convolution / GroupNorm arithmetic itself lives in PyTorch ATen (third
party, not vendored by the reference; the reference pins pytorch==1.8.1,
this image has 2.10.0).  The restatement therefore uses the same ATen CPU
primitives (F.conv2d, F.group_norm, ...) that the reference's nn.Modules
dispatch to; what is restated here is the reference's own wiring and scalar
arithmetic.  Citations are file:line under
/root/reference/FastDiffSR/model/fastdiffsr_modules/.
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from fastdiffsr_amd.arch import UNetConfig   # the hyper-parameter record only (type of `cfg`)
from oracle.layers import wiring as build_layers   # the oracle's own wiring table

Tensor = torch.Tensor


# --------------------------------------------------------------------------
# a1  make_beta_schedule                                   diffusion.py:21-64
# --------------------------------------------------------------------------
def make_beta_schedule(schedule: str, n_timestep: int, linear_start: float = 1e-4,
                       linear_end: float = 2e-2, cosine_s: float = 8e-3) -> np.ndarray:
    T = int(n_timestep)
    if schedule == 'quad':                                   # :22-24
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, T, dtype=np.float64) ** 2
    if schedule == 'linear':                                 # :25-27
        return np.linspace(linear_start, linear_end, T, dtype=np.float64)
    if schedule in ('warmup10', 'warmup50'):                 # :13-18, :28-33
        frac = 0.1 if schedule == 'warmup10' else 0.5
        b = linear_end * np.ones(T, dtype=np.float64)
        w = int(T * frac)
        b[:w] = np.linspace(linear_start, linear_end, w, dtype=np.float64)
        return b
    if schedule == 'const':                                  # :34-35
        return linear_end * np.ones(T, dtype=np.float64)
    if schedule == 'jsd':                                    # :36-38
        return 1. / np.linspace(T, 1, T, dtype=np.float64)
    if schedule == 'cosine':                                 # :39-48 (torch float64 there)
        ts = torch.arange(T + 1, dtype=torch.float64) / T + cosine_s
        ac = torch.cos(ts / (1 + cosine_s) * math.pi / 2).pow(2)
        ac = ac / ac[0]
        return (1 - ac[1:] / ac[:-1]).clamp(max=0.999).numpy()
    if schedule == 'linear_cosine':                          # :50-61
        b1 = np.linspace(linear_start, linear_end, T, dtype=np.float64)
        steps = T + 1
        x = np.linspace(0, steps, steps)                     # T+1 points over [0, T+1] (sic)
        ac = np.cos(((x / steps) + cosine_s) / (1 + cosine_s) * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        b2 = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
        # cosine term added TWICE (:60) -- the paper's figure says 1.5x; parity follows the code
        return np.clip(np.add(b1, np.add(b2, b2)), a_min=0, a_max=0.999)
    raise NotImplementedError(schedule)                      # :62-63


# --------------------------------------------------------------------------
# a2  set_new_noise_schedule                              diffusion.py:109-155
# --------------------------------------------------------------------------
def schedule_tables(schedule_opt: dict) -> Dict[str, np.ndarray]:
    """The 12 fp32 buffers + the fp64 python attribute sqrt_alphas_cumprod_prev[T+1]."""
    betas = make_beta_schedule(schedule_opt['schedule'], schedule_opt['n_timestep'],
                               schedule_opt['linear_start'], schedule_opt['linear_end'])
    alphas = 1. - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1., ac[:-1])
    pv = betas * (1. - ac_prev) / (1. - ac)
    f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
    return {
        'sqrt_alphas_cumprod_prev_f64': np.sqrt(np.append(1., ac)),          # :121-122
        'betas': f32(betas),
        'alphas_cumprod': f32(ac),
        'alphas_cumprod_prev': f32(ac_prev),
        'sqrt_alphas_cumprod': f32(np.sqrt(ac)),
        'sqrt_one_minus_alphas_cumprod': f32(np.sqrt(1. - ac)),
        'log_one_minus_alphas_cumprod': f32(np.log(1. - ac)),
        'sqrt_recip_alphas_cumprod': f32(np.sqrt(1. / ac)),
        'sqrt_recipm1_alphas_cumprod': f32(np.sqrt(1. / ac - 1)),
        'posterior_variance': f32(pv),
        'posterior_log_variance_clipped': f32(np.log(np.maximum(pv, 1e-20))),
        'posterior_mean_coef1': f32(betas * np.sqrt(ac_prev) / (1. - ac)),
        'posterior_mean_coef2': f32((1. - ac_prev) * np.sqrt(alphas) / (1. - ac)),
    }


# --------------------------------------------------------------------------
# a10-a19  UNet pieces                                         unet.py:22-222
# --------------------------------------------------------------------------
def swish(x: Tensor) -> Tensor:                              # unet.py:57-59
    return x * torch.sigmoid(x)


def positional_encoding(noise_level: Tensor, dim: int) -> Tensor:   # unet.py:22-35
    count = dim // 2
    step = torch.arange(count, dtype=noise_level.dtype, device=noise_level.device) / count
    enc = noise_level.unsqueeze(1) * torch.exp(-math.log(1e4) * step.unsqueeze(0))
    return torch.cat([torch.sin(enc), torch.cos(enc)], dim=-1)


def noise_level_mlp(sd, noise_level: Tensor, inner: int) -> Tensor:  # unet.py:242-248
    t = positional_encoding(noise_level, inner)
    t = F.linear(t, sd['noise_level_mlp.1.weight'], sd['noise_level_mlp.1.bias'])
    t = swish(t)
    return F.linear(t, sd['noise_level_mlp.3.weight'], sd['noise_level_mlp.3.bias'])


def block(sd, p: str, x: Tensor, groups: int, dropout_mask: Optional[Tensor] = None) -> Tensor:
    """GroupNorm -> Swish -> (Dropout) -> Conv3x3                unet.py:89-101"""
    h = F.group_norm(x, groups, sd[f'{p}.block.0.weight'], sd[f'{p}.block.0.bias'], eps=1e-5)
    h = swish(h)
    if dropout_mask is not None:
        h = h * dropout_mask
    return F.conv2d(h, sd[f'{p}.block.3.weight'], sd[f'{p}.block.3.bias'], padding=1)


def clam(sd, p: str, x: Tensor) -> Tensor:                   # unet.py:123-149 ('Avg|Max')
    w1, w2 = sd[f'{p}.fc1.weight'], sd[f'{p}.fc2.weight']
    mlp = lambda v: F.conv2d(F.relu(F.conv2d(v, w1)), w2)
    g = mlp(F.adaptive_avg_pool2d(x, 1)) + mlp(F.adaptive_max_pool2d(x, 1))
    return torch.sigmoid(g) * x


def slam(sd, p: str, x: Tensor) -> Tensor:                   # unet.py:151-173 ('Avg|Max', k=7)
    m = torch.cat([torch.mean(x, dim=1, keepdim=True), torch.max(x, dim=1, keepdim=True)[0]], dim=1)
    return torch.sigmoid(F.conv2d(m, sd[f'{p}.conv1.weight'], padding=3)) * x


def resnet_block(sd, p: str, x: Tensor, t: Tensor, groups: int, has_res_conv: bool,
                 dropout_mask: Optional[Tensor] = None) -> Tensor:
    """block2(block1(x) + Linear(t)) + res_conv(x)            unet.py:104-120, :38-54 (shift only)"""
    r = f'{p}.res_block'
    h = block(sd, f'{r}.block1', x, groups)
    shift = F.linear(t, sd[f'{r}.noise_func.noise_func.0.weight'], sd[f'{r}.noise_func.noise_func.0.bias'])
    h = h + shift.view(x.shape[0], -1, 1, 1)
    h = block(sd, f'{r}.block2', h, groups, dropout_mask)
    if has_res_conv:
        x = F.conv2d(x, sd[f'{r}.res_conv.weight'], sd[f'{r}.res_conv.bias'])
    return h + x


def unet_forward(sd: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, noise_level: Tensor,
                 dropout_masks: Optional[Dict[str, Tensor]] = None,
                 capture: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """UNet.forward                                           unet.py:299-323

    sd keys carry no 'denoise_fn.' prefix.  x: [B,in_channel,H,W]; noise_level: [B,1]."""
    G = cfg.norm_groups
    t = noise_level_mlp(sd, noise_level, cfg.inner_channel)          # [B,1,inner]
    feats: List[Tensor] = []
    layers = build_layers(cfg)
    n_down = sum(1 for L in layers if L.name.startswith('downs.'))
    for i, L in enumerate(layers):
        dm = None if dropout_masks is None else dropout_masks.get(L.name)
        if L.kind == 'conv_in':
            x = F.conv2d(x, sd[f'{L.name}.weight'], sd[f'{L.name}.bias'], padding=1)
        elif L.kind == 'down':                                       # unet.py:77-83
            x = F.conv2d(x, sd[f'{L.name}.conv.weight'], sd[f'{L.name}.conv.bias'], stride=2, padding=1)
        elif L.kind == 'up':                                         # unet.py:66-74
            x = F.interpolate(x, scale_factor=2, mode='nearest')
            x = F.conv2d(x, sd[f'{L.name}.conv.weight'], sd[f'{L.name}.conv.bias'], padding=1)
        elif L.kind == 'res':
            if L.name.startswith('ups.'):
                x = torch.cat((x, feats.pop()), dim=1)               # unet.py:319
            x = resnet_block(sd, L.name, x, t, G, L.cin != L.cout, dm)
            if L.with_attn:                                          # unet.py:217-222
                x = slam(sd, f'{L.name}.sa', clam(sd, f'{L.name}.ca', x))
        elif L.kind == 'final':
            x = block(sd, L.name, x, G)
        if capture is not None:
            capture[L.name] = x      # output of reference module L.name (test introspection)
        if i < n_down:
            feats.append(x)
    return x


# --------------------------------------------------------------------------
# a3-a8  reverse diffusion                                diffusion.py:157-231
# --------------------------------------------------------------------------
def res2img(r: Tensor, sr: Tensor) -> Tensor:                # diffusion.py:275-281
    return r.clamp(-1, 1) / 2.0 + sr


def img2res(hr: Tensor, sr: Tensor) -> Tensor:               # diffusion.py:283-289
    return ((hr - sr) * 2.0).clamp(-1, 1)


def p_sample(sd, cfg: UNetConfig, tab: Dict[str, np.ndarray], x: Tensor, t: int,
             cond: Tensor, noise: Optional[Tensor]) -> Tensor:
    """p_sample + p_mean_variance + predict_start + q_posterior  diffusion.py:157-190

    `noise` is the N(0,1) tensor the reference would draw with randn_like (ignored at t==0)."""
    B = x.shape[0]
    nl = torch.FloatTensor([tab['sqrt_alphas_cumprod_prev_f64'][t + 1]]).repeat(B, 1).to(x.device)   # :169-170
    eps = unet_forward(sd, cfg, torch.cat([cond, x], dim=1), nl)                        # :173
    T = lambda k: torch.tensor(tab[k][t])                   # 0-dim fp32 tensor, as buffer[t]
    x0 = T('sqrt_recip_alphas_cumprod') * x - T('sqrt_recipm1_alphas_cumprod') * eps    # :157-159
    x0 = x0.clamp(-1., 1.)                                                              # :178-179
    mean = T('posterior_mean_coef1') * x0 + T('posterior_mean_coef2') * x               # :161-165
    logvar = T('posterior_log_variance_clipped')
    nz = noise if t > 0 else torch.zeros_like(x)                                        # :189
    return mean + nz * (0.5 * logvar).exp()                                             # :190


def p_sample_loop(sd, cfg: UNetConfig, tab: Dict[str, np.ndarray], cond: Tensor, noise: Tensor,
                  return_trajectory: bool = False):
    """Conditional branch of p_sample_loop, batched as B independent runs.
                                                            diffusion.py:192-221
    noise: [T,B,3,H,W]; noise[0] = x_T (the `randn(shape)` draw, :207), noise[k] = the
    `randn_like` of step t = T-k (k = 1..T-1).  Returns res2img(x_0, cond) (== ret_img[-1])."""
    T = int(tab['betas'].shape[0])
    img = noise[0]
    traj = []
    with torch.no_grad():
        for k, t in enumerate(reversed(range(T))):
            nz = noise[k + 1] if t > 0 else None
            img = p_sample(sd, cfg, tab, img, t, cond, nz)
            if return_trajectory:
                traj.append(img.clone())
        out = res2img(img, cond)
    return (out, traj) if return_trajectory else out


def continuous_frames(T: int) -> List[int]:
    """Timesteps whose x_t the reference keeps when continous=True (:195, :211-212)."""
    inter = 1 | (T // 10)
    return [t for t in reversed(range(T)) if t % inter == 0]


# --------------------------------------------------------------------------
# a20  training loss                                      diffusion.py:233-270
# --------------------------------------------------------------------------
def q_sample(x_start: Tensor, gamma: Tensor, noise: Tensor) -> Tensor:     # :233-241
    return gamma * x_start + (1 - gamma ** 2).sqrt() * noise


def p_losses(sd, cfg: UNetConfig, hr: Tensor, sr: Tensor, gamma: Tensor, noise: Tensor,
             loss_type: str = 'l1', dropout_masks=None) -> Tensor:
    """L1(sum) between noise and UNet(cat[SR, x_noisy], gamma).  gamma: [B] continuous
    sqrt(alpha_bar) (drawn by the caller; reference draws it with numpy at :246-255)."""
    x_start = img2res(hr, sr)                                              # :245
    g = gamma.view(-1, 1)
    x_noisy = q_sample(x_start, g.view(-1, 1, 1, 1), noise)                # :259-260
    rec = unet_forward(sd, cfg, torch.cat([sr, x_noisy], dim=1), g, dropout_masks)   # :265-266
    if loss_type == 'l1':
        return F.l1_loss(noise, rec, reduction='sum')                      # :101-103, :268
    if loss_type == 'l2':
        return F.mse_loss(noise, rec, reduction='sum')
    raise NotImplementedError()


# --------------------------------------------------------------------------
# f-3 (next row)  one optimisation step             FastDiffSR/model/model.py:27-57
# --------------------------------------------------------------------------
def train_step(sd, cfg: UNetConfig, hr: Tensor, sr: Tensor, gamma: Tensor, noise: Tensor, lr: float,
               loss_type: str = 'l1', betas=(0.9, 0.999), eps: float = 1e-8, dropout_masks=None):
    """DDPM.optimize_parameters (model.py:47-57) from fresh Adam state: zero_grad, l_pix = netG(data),
    l_pix = l_pix.sum() / (b*c*h*w), backward, Adam step (torch.optim.Adam defaults, model.py:37-38).
    Gradients come from autograd over the restated forward; tensors the forward never touches (the 44
    dead `.conv` tensors, unet.py:212) get no gradient and are left alone by the optimiser, as in torch.
    Returns (l_pix, grads {key: Tensor}, new_sd {key: Tensor})."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss = p_losses(leaves, cfg, hr, sr, gamma, noise, loss_type, dropout_masks)
    b, c, h, w = hr.shape
    l_pix = loss.sum() / int(b * c * h * w)
    l_pix.backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    new_sd = {}
    b1, b2 = betas
    for k, w_ in sd.items():
        g = grads.get(k)
        if g is None:
            new_sd[k] = w_.detach().clone()
            continue
        m = (1 - b1) * g                                   # exp_avg after step 1
        v = (1 - b2) * g * g                               # exp_avg_sq after step 1
        denom = (v.sqrt() / (1 - b2) ** 0.5) + eps         # bias_correction2 = 1 - b2
        new_sd[k] = w_.detach() - (lr / (1 - b1)) * (m / denom)
    return l_pix.detach(), grads, new_sd


# --------------------------------------------------------------------------
# f-1 (next row)  tensor2img / PSNR           FastDiffSR/core/metrics.py:16-42, :94-101
# --------------------------------------------------------------------------
def tensor2img_u8(t: Tensor, min_max=(-1, 1)) -> np.ndarray:
    """[3,H,W] fp32 -> HWC uint8 RGB: clamp, rescale to [0,1], *255, round."""
    t = t.squeeze().float().cpu().clamp_(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    img = np.transpose(t.numpy(), (1, 2, 0))
    return (img * 255.0).round().astype(np.uint8)


def psnr_u8(a: np.ndarray, b: np.ndarray) -> float:
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    if mse == 0:
        return float('inf')
    return 20 * math.log10(255.0 / math.sqrt(mse))


def to_torch_sd(np_sd) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in np_sd.items()}
