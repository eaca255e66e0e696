"""Noise schedules of GaussianDiffusion (host-side, float64 numpy).

Product-side implementation of the reference's `make_beta_schedule`
(FastDiffSR/model/fastdiffsr_modules/diffusion.py:21-64) and of the buffer
algebra in `set_new_noise_schedule` (:109-155).  tests/ cross-check it bit for
bit against the oracle and the reference-generated goldens.
"""
import math
from collections import OrderedDict

import numpy as np


def _warmup(start, end, T, frac):
    b = end * np.ones(T, dtype=np.float64)
    n = int(T * frac)
    b[:n] = np.linspace(start, end, n, dtype=np.float64)
    return b


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    T = int(n_timestep)
    lin = lambda a, b: np.linspace(a, b, T, dtype=np.float64)
    if schedule == 'linear_cosine':
        # linear ramp plus TWICE the clipped cosine betas (the code adds betas2 twice, :60);
        # the cosine grid is linspace(0, T+1, T+1), i.e. T+1 points with spacing (T+1)/T (:53-54)
        n = T + 1
        grid = np.linspace(0, n, n)
        abar = np.cos((grid / n + cosine_s) / (1 + cosine_s) * np.pi * 0.5) ** 2
        abar = abar / abar[0]
        cosb = np.clip(1 - abar[1:] / abar[:-1], a_min=0, a_max=0.999)
        return np.clip(np.add(lin(linear_start, linear_end), np.add(cosb, cosb)), a_min=0, a_max=0.999)
    if schedule == 'linear':
        return lin(linear_start, linear_end)
    if schedule == 'quad':
        return lin(linear_start ** 0.5, linear_end ** 0.5) ** 2
    if schedule == 'warmup10':
        return _warmup(linear_start, linear_end, T, 0.1)
    if schedule == 'warmup50':
        return _warmup(linear_start, linear_end, T, 0.5)
    if schedule == 'const':
        return linear_end * np.ones(T, dtype=np.float64)
    if schedule == 'jsd':
        return 1. / np.linspace(T, 1, T, dtype=np.float64)
    if schedule == 'cosine':
        import torch
        ts = torch.arange(T + 1, dtype=torch.float64) / T + cosine_s
        abar = torch.cos(ts / (1 + cosine_s) * math.pi / 2).pow(2)
        abar = abar / abar[0]
        return (1 - abar[1:] / abar[:-1]).clamp(max=0.999).numpy()
    raise NotImplementedError(schedule)


def schedule_buffers(schedule_opt):
    """-> (OrderedDict of the 12 registered fp32 buffers, float64 sqrt_alphas_cumprod_prev[T+1])."""
    betas = make_beta_schedule(schedule=schedule_opt['schedule'], n_timestep=schedule_opt['n_timestep'],
                               linear_start=schedule_opt['linear_start'], linear_end=schedule_opt['linear_end'])
    alphas = 1. - betas
    abar = np.cumprod(alphas, axis=0)
    abar_prev = np.append(1., abar[:-1])
    post_var = betas * (1. - abar_prev) / (1. - abar)
    f32 = lambda a: np.asarray(a, dtype=np.float64).astype(np.float32)
    bufs = OrderedDict()
    bufs['betas'] = f32(betas)
    bufs['alphas_cumprod'] = f32(abar)
    bufs['alphas_cumprod_prev'] = f32(abar_prev)
    bufs['sqrt_alphas_cumprod'] = f32(np.sqrt(abar))
    bufs['sqrt_one_minus_alphas_cumprod'] = f32(np.sqrt(1. - abar))
    bufs['log_one_minus_alphas_cumprod'] = f32(np.log(1. - abar))
    bufs['sqrt_recip_alphas_cumprod'] = f32(np.sqrt(1. / abar))
    bufs['sqrt_recipm1_alphas_cumprod'] = f32(np.sqrt(1. / abar - 1))
    bufs['posterior_variance'] = f32(post_var)
    bufs['posterior_log_variance_clipped'] = f32(np.log(np.maximum(post_var, 1e-20)))
    bufs['posterior_mean_coef1'] = f32(betas * np.sqrt(abar_prev) / (1. - abar))
    bufs['posterior_mean_coef2'] = f32((1. - abar_prev) * np.sqrt(alphas) / (1. - abar))
    return bufs, np.sqrt(np.append(1., abar))


def sampling_scalars(bufs, sqrt_abar_prev_f64):
    """Per-timestep fp32 scalars p_sample reads (diffusion.py:157-190), as the reference forms them:
    noise_level = fp32(sqrt_alphas_cumprod_prev[t+1]) (:169-170), sigma = exp(0.5*logvar) in fp32 (:190)."""
    import torch
    T = bufs['betas'].shape[0]
    nl = torch.FloatTensor([float(sqrt_abar_prev_f64[t + 1]) for t in range(T)]).numpy()
    sigma = (0.5 * torch.from_numpy(bufs['posterior_log_variance_clipped'])).exp().numpy()
    return dict(noise_level=nl, sqrt_recip=bufs['sqrt_recip_alphas_cumprod'],
                sqrt_recipm1=bufs['sqrt_recipm1_alphas_cumprod'], coef1=bufs['posterior_mean_coef1'],
                coef2=bufs['posterior_mean_coef2'], sigma=sigma)
