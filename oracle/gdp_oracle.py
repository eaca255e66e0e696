"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch functional ops, fp32) of the GDP sibling,
FastDiffSR/model/gdp_modules/{unet,diffusion}.py, for the parity tests of `which_model_G == 'gdp'`.
The product path (fastdiffsr_amd/) never imports this.  Pinned by tests/golden/gdp.npz, generated from the
reference's own gdp_modules by oracle/make_goldens.py.

The denoiser is the guided-diffusion UNet as model/networks.py:88-104 instantiates it (use_scale_shift_norm,
resblock_updown, heads of 64 channels, QKVAttentionLegacy); it predicts x_0 and is fed cat([x_t, cond])."""
import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from fastdiffsr_amd.arch import UNetConfig
from fastdiffsr_amd.gdp.arch import gdp_layers

Tensor = torch.Tensor


def timestep_embedding(t: Tensor, dim: int, max_period: int = 10000) -> Tensor:            # gdp_modules/unet.py:120-138
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def res_block(sd, p: str, x: Tensor, emb: Tensor, G: int, mode: str, dropout_mask=None) -> Tensor:   # :276-390
    h = F.silu(F.group_norm(x, G, sd[f'{p}.in_layers.0.weight'], sd[f'{p}.in_layers.0.bias'], eps=1e-5))
    if mode == 'down':                                                                    # :369-375 (avg_pool both)
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    elif mode == 'up':
        h = F.interpolate(h, scale_factor=2, mode='nearest')
        x = F.interpolate(x, scale_factor=2, mode='nearest')
    h = F.conv2d(h, sd[f'{p}.in_layers.2.weight'], sd[f'{p}.in_layers.2.bias'], padding=1)
    e = F.linear(F.silu(emb), sd[f'{p}.emb_layers.1.weight'], sd[f'{p}.emb_layers.1.bias'])[..., None, None]
    scale, shift = torch.chunk(e, 2, dim=1)                                               # :377-381
    h = F.group_norm(h, G, sd[f'{p}.out_layers.0.weight'], sd[f'{p}.out_layers.0.bias'], eps=1e-5) * (1 + scale) + shift
    h = F.silu(h)
    if dropout_mask is not None:
        h = h * dropout_mask
    h = F.conv2d(h, sd[f'{p}.out_layers.3.weight'], sd[f'{p}.out_layers.3.bias'], padding=1)
    if f'{p}.skip_connection.weight' in sd:
        x = F.conv2d(x, sd[f'{p}.skip_connection.weight'], sd[f'{p}.skip_connection.bias'])
    return x + h


def attention_block(sd, p: str, x: Tensor, G: int) -> Tensor:                              # :392-439, :461-488
    b, c, hh, ww = x.shape
    heads = c // 64                                                                       # num_head_channels = 64
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(F.group_norm(xf, G, sd[f'{p}.norm.weight'], sd[f'{p}.norm.bias'], eps=1e-5), sd[f'{p}.qkv.weight'],
                   sd[f'{p}.qkv.bias'])
    ch = c // heads
    q, k, v = qkv.reshape(b * heads, ch * 3, -1).split(ch, dim=1)                         # QKVAttentionLegacy
    scale = 1 / math.sqrt(math.sqrt(ch))
    w = torch.softmax(torch.einsum('bct,bcs->bts', q * scale, k * scale).float(), dim=-1)
    a = torch.einsum('bts,bcs->bct', w, v).reshape(b, -1, hh * ww)
    out = F.conv1d(a, sd[f'{p}.proj_out.weight'], sd[f'{p}.proj_out.bias'])
    return (xf + out).reshape(b, c, hh, ww)


def unet_forward(sd: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, timesteps: Tensor, capture=None) -> Tensor:
    """UNet.forward(x, timesteps)                                                  gdp_modules/unet.py:773-800"""
    G = cfg.norm_groups
    emb = timestep_embedding(timesteps, cfg.inner_channel)
    emb = F.linear(F.silu(F.linear(emb, sd['time_embed.0.weight'], sd['time_embed.0.bias'])), sd['time_embed.2.weight'],
                   sd['time_embed.2.bias'])
    hs = []
    h = x
    for L in gdp_layers(cfg):
        if L.kind == 'conv_in':
            h = F.conv2d(h, sd[f'{L.name}.weight'], sd[f'{L.name}.bias'], padding=1)
        elif L.kind == 'res':
            if L.pop:
                h = torch.cat([h, hs.pop()], dim=1)
            h = res_block(sd, L.name, h, emb, G, L.mode)
        elif L.kind == 'attn':
            h = attention_block(sd, L.name, h, G)
        elif L.kind == 'out':
            h = F.conv2d(F.silu(F.group_norm(h, G, sd['out.0.weight'], sd['out.0.bias'], eps=1e-5)), sd['out.2.weight'],
                         sd['out.2.bias'], padding=1)
        if capture is not None and L.block:
            capture[L.block] = h
        if L.push:
            hs.append(h)
    return h


def p_sample(sd, cfg, tab, x: Tensor, t: int, cond: Tensor, noise: Tensor) -> Tensor:
    """p_sample / p_mean_variance / q_posterior: the network predicts x_0          gdp_modules/diffusion.py:189-212"""
    B = x.shape[0]
    tt = torch.full((B,), t, dtype=torch.long)
    x0 = unet_forward(sd, cfg, torch.cat([x, cond], dim=1), tt).clamp(-1., 1.)            # :191-195
    T = lambda k: torch.tensor(tab[k][t])
    mean = T('posterior_mean_coef1') * x0 + T('posterior_mean_coef2') * x
    mask = 0.0 if t == 0 else 1.0                                                         # :208-210
    return mean + mask * (0.5 * T('posterior_log_variance_clipped')).exp() * noise


def p_sample_loop(sd, cfg, tab, cond: Tensor, noise: Tensor, return_trajectory=False):
    """Conditional p_sample_loop; noise [T+1,B,3,H,W]: x_T, then one draw per step (the last is masked)   :213-240"""
    T = int(tab['betas'].shape[0])
    img = noise[0]
    traj = []
    with torch.no_grad():
        for k, t in enumerate(reversed(range(T))):
            img = p_sample(sd, cfg, tab, img, t, cond, noise[k + 1])
            if return_trajectory:
                traj.append(img.clone())
    return (img, traj) if return_trajectory else img


def p_losses(sd, cfg, tab, hr: Tensor, sr: Tensor, t: Tensor, noise: Tensor) -> Tensor:   # :266-285 (both loss types are MSE sum)
    a = torch.from_numpy(np.asarray(tab['sqrt_alphas_cumprod'], dtype=np.float32))[t].view(-1, 1, 1, 1)
    s = torch.from_numpy(np.asarray(tab['sqrt_one_minus_alphas_cumprod'], dtype=np.float32))[t].view(-1, 1, 1, 1)
    x_t = a * hr + s * noise
    rec = unet_forward(sd, cfg, torch.cat([x_t, sr], dim=1), t)
    return F.mse_loss(rec, hr, reduction='sum')


def train_step(sd, cfg: UNetConfig, tab, hr: Tensor, sr: Tensor, t: Tensor, noise: Tensor, lr: float, betas=(0.9, 0.999),
               eps: float = 1e-8):
    """DDPM.optimize_parameters (model/model.py:47-57) on the GDP sibling from fresh Adam state: l_pix = p_losses(...).sum() /
    (b*c*h*w), backward by autograd over the restated forward, one Adam step.  Dropout off.  Returns (l_pix, grads {key: Tensor},
    new_sd {key: Tensor}) as oracle.sr3_oracle.train_step."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss = p_losses(leaves, cfg, tab, hr, sr, t, noise)
    b, c, h, w = hr.shape
    l_pix = loss.sum() / int(b * c * h * w)
    l_pix.backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    b1, b2 = betas
    new_sd = {}
    for k, w_ in sd.items():
        g = grads.get(k)
        if g is None:
            new_sd[k] = w_.detach().clone()
            continue
        m = (1 - b1) * g
        v = (1 - b2) * g * g
        denom = (v.sqrt() / (1 - b2) ** 0.5) + eps
        new_sd[k] = w_.detach() - (lr / (1 - b1)) * (m / denom)
    return l_pix.detach(), grads, new_sd
