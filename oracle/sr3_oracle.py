"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatement (PyTorch-CPU functional ops, fp32) of the SR3 sibling path behind the same plugin
boundary: FastDiffSR/model/ddpm_modules/{unet,diffusion}.py (`which_model_G == 'ddpm'`,
model/networks.py:84-85) -- SURVEY 8f-4.  Pinned by outputs of the reference modules themselves
(tests/golden/sr3.npz, oracle/make_goldens.py).  Citations are file:line under
/root/reference/FastDiffSR/model/ddpm_modules/.
"""
import math
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from fastdiffsr_amd.arch import UNetConfig   # the hyper-parameter record only (type of `cfg`)
from oracle.layers import wiring as build_layers   # the oracle's own wiring table
from oracle.fdsr_oracle import block, swish

Tensor = torch.Tensor


def time_embedding(t: Tensor, inv_freq: Tensor) -> Tensor:          # unet.py:19-34
    s = torch.ger(t.view(-1).float(), inv_freq)
    return torch.cat([s.sin(), s.cos()], dim=-1).view(*t.shape, 2 * inv_freq.numel())


def self_attention(sd, p: str, x: Tensor, groups: int) -> Tensor:   # unet.py:99-127 (n_head = 1)
    b, c, h, w = x.shape
    norm = F.group_norm(x, groups, sd[f'{p}.norm.weight'], sd[f'{p}.norm.bias'], eps=1e-5)
    qkv = F.conv2d(norm, sd[f'{p}.qkv.weight']).view(b, 1, c * 3, h, w)
    q, k, v = qkv.chunk(3, dim=2)
    attn = torch.einsum('bnchw, bncyx -> bnhwyx', q, k).contiguous() / math.sqrt(c)
    attn = torch.softmax(attn.view(b, 1, h, w, -1), -1).view(b, 1, h, w, h, w)
    out = torch.einsum('bnhwyx, bncyx -> bnchw', attn, v).contiguous()
    out = F.conv2d(out.view(b, c, h, w), sd[f'{p}.out.weight'], sd[f'{p}.out.bias'])
    return out + x


def resnet_block(sd, p: str, x: Tensor, t: Tensor, groups: int, has_res_conv: bool) -> Tensor:   # unet.py:78-96
    r = f'{p}.res_block'
    h = block(sd, f'{r}.block1', x, groups)
    h = h + F.linear(swish(t), sd[f'{r}.mlp.1.weight'], sd[f'{r}.mlp.1.bias'])[:, :, None, None]
    h = block(sd, f'{r}.block2', h, groups)
    if has_res_conv:
        x = F.conv2d(x, sd[f'{r}.res_conv.weight'], sd[f'{r}.res_conv.bias'])
    return h + x


def unet_forward(sd: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, time: Tensor, capture=None) -> Tensor:
    """UNet.forward(x, time) with time a [B] integer tensor                      unet.py:233-259"""
    G = cfg.norm_groups
    t = time_embedding(time, sd['time_mlp.0.inv_freq'])                        # :159-165
    t = F.linear(t, sd['time_mlp.1.weight'], sd['time_mlp.1.bias'])
    t = F.linear(swish(t), sd['time_mlp.3.weight'], sd['time_mlp.3.bias'])
    feats: List[Tensor] = []
    layers = build_layers(cfg)
    n_down = sum(1 for L in layers if L.name.startswith('downs.'))
    for i, L in enumerate(layers):
        if L.kind == 'conv_in':
            x = F.conv2d(x, sd[f'{L.name}.weight'], sd[f'{L.name}.bias'], padding=1)
        elif L.kind == 'down':
            x = F.conv2d(x, sd[f'{L.name}.conv.weight'], sd[f'{L.name}.conv.bias'], stride=2, padding=1)
        elif L.kind == 'up':
            x = F.interpolate(x, scale_factor=2, mode='nearest')
            x = F.conv2d(x, sd[f'{L.name}.conv.weight'], sd[f'{L.name}.conv.bias'], padding=1)
        elif L.kind == 'res':
            if L.name.startswith('ups.'):
                x = torch.cat((x, feats.pop()), dim=1)
            x = resnet_block(sd, L.name, x, t, G, L.cin != L.cout)
            if L.with_attn:
                x = self_attention(sd, f'{L.name}.attn', x, G)
        elif L.kind == 'final':
            x = block(sd, L.name, x, G)
        if capture is not None:
            capture[L.name] = x
        if i < n_down:
            feats.append(x)
    return x


def p_sample(sd, cfg, tab, x: Tensor, t: int, cond: Tensor, noise: Tensor) -> Tensor:
    """p_sample / p_mean_variance / q_posterior with a batch-uniform integer t    diffusion.py:158-196"""
    B = x.shape[0]
    tt = torch.full((B,), t, dtype=torch.long)
    eps = unet_forward(sd, cfg, torch.cat([cond, x], dim=1), tt)
    T = lambda k: torch.tensor(tab[k][t])
    x0 = (T('sqrt_recip_alphas_cumprod') * x - T('sqrt_recipm1_alphas_cumprod') * eps).clamp(-1., 1.)
    mean = T('posterior_mean_coef1') * x0 + T('posterior_mean_coef2') * x
    mask = 0.0 if t == 0 else 1.0                                            # nonzero_mask :193-194
    return mean + mask * (0.5 * T('posterior_log_variance_clipped')).exp() * noise


def p_sample_loop(sd, cfg, tab, cond: Tensor, noise: Tensor, return_trajectory=False):
    """Conditional p_sample_loop: returns the image itself (no res2img)           diffusion.py:198-227
    noise: [T+1,B,3,H,W]; noise[0] = x_T, noise[k] = the draw of step t = T-k (drawn at t == 0 too)."""
    T = int(tab['betas'].shape[0])
    img = noise[0]
    traj = []
    with torch.no_grad():
        for k, t in enumerate(reversed(range(T))):
            img = p_sample(sd, cfg, tab, img, t, cond, noise[k + 1])
            if return_trajectory:
                traj.append(img.clone())
    return (img, traj) if return_trajectory else img


# --------------------------------------------------------------------------
# f-4 / f-3  one optimisation step of the SR3 sibling      ddpm_modules/diffusion.py:260-297, model/model.py:47-57
# --------------------------------------------------------------------------
def q_sample(tab, x_start: Tensor, t: Tensor, noise: Tensor) -> Tensor:       # diffusion.py:260-268
    a = torch.as_tensor(tab['sqrt_alphas_cumprod'])[t].view(-1, 1, 1, 1)
    b = torch.as_tensor(tab['sqrt_one_minus_alphas_cumprod'])[t].view(-1, 1, 1, 1)
    return a * x_start + b * noise


def p_losses(sd, cfg: UNetConfig, tab, hr: Tensor, sr: Tensor, t: Tensor, noise: Tensor, loss_type: str = 'l1') -> Tensor:
    """L1(sum) between the noise and UNet(cat[SR, q_sample(HR, t)], t); t: [B] integer times (the caller's draw: the reference
    draws torch.randint(0, T, (b,)), :283-284).  Dropout off."""
    x_noisy = q_sample(tab, hr, t, noise)
    rec = unet_forward(sd, cfg, torch.cat([sr, x_noisy], dim=1), t)
    if loss_type == 'l1':
        return F.l1_loss(noise, rec, reduction='sum')
    if loss_type == 'l2':
        return F.mse_loss(noise, rec, reduction='sum')
    raise NotImplementedError()


def train_step(sd, cfg: UNetConfig, tab, hr: Tensor, sr: Tensor, t: Tensor, noise: Tensor, lr: float, loss_type: str = 'l1',
               betas=(0.9, 0.999), eps: float = 1e-8):
    """DDPM.optimize_parameters from fresh Adam state, as oracle.fdsr_oracle.train_step: gradients by autograd over the restated
    forward.  Returns (l_pix, grads {key: Tensor}, new_sd {key: Tensor}); the inv_freq buffer takes no gradient."""
    leaves = {k: (v.detach().clone().requires_grad_(True) if not k.endswith('inv_freq') else v.detach().clone()) for k, v in sd.items()}
    loss = p_losses(leaves, cfg, tab, hr, sr, t, noise, loss_type)
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
