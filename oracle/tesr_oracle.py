"""TEST INFRASTRUCTURE ONLY -- CPU restatement (PyTorch functional ops, fp32) of the TESR sibling,
FastDiffSR/model/tesr_modules/{unet,diffusion}.py, for the parity tests of `which_model_G == 'tesr'`.
The product path (fastdiffsr_amd/) never imports this.  Pinned by tests/golden/tesr.npz, generated from the
reference's own tesr_modules by oracle/make_goldens.py.

TESR's denoiser is FastDiffSR's Block / ResnetBlock / noise-level embedding (oracle/fdsr_oracle.py) with SR3's
SelfAttention (oracle/sr3_oracle.py; tesr_modules/unet.py:120-149 is the same module) where the resolution is in
attn_res and in mid[0]; its sampler is FastDiffSR's reverse process returning x_0 itself."""
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from fastdiffsr_amd.arch import UNetConfig   # the hyper-parameter record only (type of `cfg`)
from oracle.layers import wiring as build_layers   # the oracle's own wiring table
from oracle.fdsr_oracle import block, noise_level_mlp, resnet_block
from oracle.sr3_oracle import self_attention

Tensor = torch.Tensor


def unet_forward(sd: Dict[str, Tensor], cfg: UNetConfig, x: Tensor, noise_level: Tensor, capture=None) -> Tensor:
    """UNet.forward(x, time) with time the continuous noise level [B,1]            tesr_modules/unet.py:243-269"""
    G = cfg.norm_groups
    t = noise_level_mlp(sd, noise_level, cfg.inner_channel)                      # :179-186
    feats: List[Tensor] = []
    layers = build_layers(cfg)
    n_down = sum(1 for L in layers if L.name.startswith('downs.'))
    for i, L in enumerate(layers):
        if L.kind == 'conv_in':
            x = F.conv2d(x, sd[f'{L.name}.weight'], sd[f'{L.name}.bias'], padding=1)
        elif L.kind == 'down':                                                   # :75-84
            x = F.conv2d(x, sd[f'{L.name}.conv.weight'], sd[f'{L.name}.conv.bias'], stride=2, padding=1)
        elif L.kind == 'up':                                                     # :65-72
            x = F.interpolate(x, scale_factor=2, mode='nearest')
            x = F.conv2d(x, sd[f'{L.name}.conv.weight'], sd[f'{L.name}.conv.bias'], padding=1)
        elif L.kind == 'res':
            if L.name.startswith('ups.'):
                x = torch.cat((x, feats.pop()), dim=1)                           # :264
            x = resnet_block(sd, L.name, x, t, G, L.cin != L.cout)               # :101-117
            if L.with_attn:
                x = self_attention(sd, f'{L.name}.attn', x, G)                   # :152-165
        elif L.kind == 'final':
            x = block(sd, L.name, x, G)
        if capture is not None:
            capture[L.name] = x
        if i < n_down:
            feats.append(x)
    return x


def p_sample(sd, cfg, tab, x: Tensor, t: int, cond: Tensor, noise) -> Tensor:
    """p_sample / p_mean_variance / q_posterior                                    tesr_modules/diffusion.py:143-181"""
    B = x.shape[0]
    nl = torch.FloatTensor([tab['sqrt_alphas_cumprod_prev_f64'][t + 1]]).repeat(B, 1)      # :156-157
    eps = unet_forward(sd, cfg, torch.cat([cond, x], dim=1), nl)
    T = lambda k: torch.tensor(tab[k][t])
    x0 = (T('sqrt_recip_alphas_cumprod') * x - T('sqrt_recipm1_alphas_cumprod') * eps).clamp(-1., 1.)
    mean = T('posterior_mean_coef1') * x0 + T('posterior_mean_coef2') * x
    nz = noise if t > 0 else torch.zeros_like(x)                                           # :180
    return mean + nz * (0.5 * T('posterior_log_variance_clipped')).exp()


def p_sample_loop(sd, cfg, tab, cond: Tensor, noise: Tensor, return_trajectory=False):
    """Conditional p_sample_loop, batched; returns x_0 itself (no res2img)          tesr_modules/diffusion.py:183-204"""
    T = int(tab['betas'].shape[0])
    img = noise[0]
    traj = []
    with torch.no_grad():
        for k, t in enumerate(reversed(range(T))):
            img = p_sample(sd, cfg, tab, img, t, cond, noise[k + 1] if t > 0 else None)
            if return_trajectory:
                traj.append(img.clone())
    return (img, traj) if return_trajectory else img


def charbonnier(x: Tensor, y: Tensor, eps: float = 1e-3) -> Tensor:               # tesr_modules/unet.py:956-967
    d = x - y
    return torch.mean(torch.sqrt(d * d + eps * eps))


# --------------------------------------------------------------------------
# f-4 / f-3  one optimisation step of the TESR sibling      tesr_modules/diffusion.py:224-250, model/model.py:47-57
# --------------------------------------------------------------------------
def p_losses(sd, cfg: UNetConfig, hr: Tensor, sr: Tensor, gamma: Tensor, noise: Tensor, loss_type: str = 'l1') -> Tensor:
    """x_start is the HR image itself; q_sample as FastDiffSR's (continuous gamma); 'l1' = the Charbonnier MEAN (:85-90)."""
    from oracle.fdsr_oracle import q_sample
    g = gamma.view(-1, 1)
    x_noisy = q_sample(hr, g.view(-1, 1, 1, 1), noise)
    rec = unet_forward(sd, cfg, torch.cat([sr, x_noisy], dim=1), g)
    if loss_type == 'l1':
        return charbonnier(noise, rec)
    if loss_type == 'l2':
        return F.mse_loss(noise, rec, reduction='sum')
    raise NotImplementedError()


def train_step(sd, cfg: UNetConfig, hr: Tensor, sr: Tensor, gamma: Tensor, noise: Tensor, lr: float, loss_type: str = 'l1',
               betas=(0.9, 0.999), eps: float = 1e-8):
    """DDPM.optimize_parameters from fresh Adam state (gradients by autograd over the restated forward), as
    oracle.fdsr_oracle.train_step.  Returns (l_pix, grads, new_sd)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
    loss = p_losses(leaves, cfg, hr, sr, gamma, noise, loss_type)
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
