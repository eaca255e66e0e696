"""`GaussianDiffusion` facade: the surface DDPM (FastDiffSR/model/model.py) calls on
`netG`, over the HIP engine.  Mirrors the constructor and methods of
FastDiffSR/model/fastdiffsr_modules/diffusion.py:76-289.

Deviation from the reference, documented: the reference's
p_sample_loop raises for batch >= 2 (diffusion.py:215-216); here a batch is B
independent B=1 runs and `continous=True` returns the frames concatenated along
dim 0 exactly as the reference's torch.cat would ([8*B,3,H,W]; for B=1 identical).
"""
import numpy as np
import torch
from torch import nn

from .schedule import schedule_buffers, sampling_scalars


class GaussianDiffusion(nn.Module):
    def __init__(self, denoise_fn, image_size, channels=3, loss_type='l1', conditional=True, schedule_opt=None, scale=4):
        super().__init__()
        self.channels = channels
        self.image_size = image_size
        self.denoise_fn = denoise_fn
        self.loss_type = loss_type
        self.conditional = conditional
        # conv arithmetic of the HIP engine: 'f16x3' (fp32-grade split-f16 MFMA, default), 'f32' (exact
        # fp32 MFMA) or 'bf16' (PSNR-grade); see include/fdsr.h
        self.precision = 'f16x3'
        # where the sampling noise comes from when the caller passes none: 'torch' = torch.randn on the
        # device in the reference's order (reproducible with torch.manual_seed, like the reference);
        # 'engine' = drawn inside the HIP loop (Philox, denoise_fn.engine.set_seed)
        self.rng = 'torch'
        # like the reference (:96-98) the schedule is NOT set here; DDPM calls set_new_noise_schedule

    def set_loss(self, device):                                   # :101-107
        if self.loss_type == 'l1':
            self.loss_func = nn.L1Loss(reduction='sum').to(device)
        elif self.loss_type == 'l2':
            self.loss_func = nn.MSELoss(reduction='sum').to(device)
        else:
            raise NotImplementedError()

    def set_new_noise_schedule(self, schedule_opt, device):       # :109-155
        bufs, sqrt_prev = schedule_buffers(schedule_opt)
        self.sqrt_alphas_cumprod_prev = sqrt_prev
        self.num_timesteps = int(bufs['betas'].shape[0])
        for k, v in bufs.items():
            self.register_buffer(k, torch.tensor(v, dtype=torch.float32, device=device))
        self.denoise_fn.engine.set_schedule(sampling_scalars(bufs, sqrt_prev))

    # -- sampling ------------------------------------------------------------------
    @torch.no_grad()
    def p_sample_loop(self, x_in, continous=False, noise=None):   # :192-221
        if not self.conditional:
            raise NotImplementedError('unconditional sampling is broken in the reference (diffusion.py:224-227)')
        device = self.betas.device
        x = x_in.to(device=device, dtype=torch.float32).contiguous()
        T = self.num_timesteps
        if noise is None and getattr(self, 'rng', 'torch') == 'engine':
            pass    # the engine draws inside the loop (Philox; Engine.set_seed): the throughput form
        elif noise is None:
            # same draws, same order as the reference: randn(shape) then randn_like per step t>0 (:207, :189)
            noise = torch.empty((T,) + tuple(x.shape), device=device, dtype=torch.float32)
            noise[0] = torch.randn(x.shape, device=device)
            for k in range(1, T):
                noise[k] = torch.randn_like(x)
        self.denoise_fn.sync_weights()
        eng = self.denoise_fn.engine
        eng.set_precision(self.precision)
        if not continous:
            return eng.sample(x, noise, graph=False)
        img, traj = eng.sample(x, noise, want_traj=True, graph=False)
        inter = (1 | (T // 10))                                   # :195
        frames = [self.res2img(x, x)]                             # ret_img[0] = res2img(x_in, x_in) (:215-216)
        for k, t in enumerate(reversed(range(T))):
            if t % inter == 0:
                frames.append(self.res2img(traj[k], x))
        return torch.cat(frames, dim=0)

    @torch.no_grad()
    def sample(self, batch_size=1, continous=False):              # :223-227 (crashes in the reference)
        raise NotImplementedError('unconditional sampling is broken in the reference (diffusion.py:224-227)')

    @torch.no_grad()
    def super_resolution(self, x_in, continous=False):            # :229-231
        return self.p_sample_loop(x_in, continous)

    # -- training surface -------------------------------------------------------------
    def q_sample(self, x_start, continuous_sqrt_alpha_cumprod, noise=None):   # :233-241
        noise = torch.randn_like(x_start) if noise is None else noise
        return continuous_sqrt_alpha_cumprod * x_start + (1 - continuous_sqrt_alpha_cumprod ** 2).sqrt() * noise

    def p_losses(self, x_in, noise=None):                         # :242-270
        """Forward value of the L1(sum) loss (no gradients: the backward kernels are the
        next row of SURVEY 8f-3).  RNG draws follow the reference: numpy for t and gamma."""
        x_start = self.img2res(x_in['HR'], x_in['SR'])
        b = x_start.shape[0]
        t = np.random.randint(1, self.num_timesteps + 1)
        gamma = torch.FloatTensor(np.random.uniform(self.sqrt_alphas_cumprod_prev[t - 1],
                                                    self.sqrt_alphas_cumprod_prev[t], size=b)).to(x_start.device)
        gamma = gamma.view(b, -1)
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start, gamma.view(-1, 1, 1, 1), noise)
        with torch.no_grad():
            x_recon = self.denoise_fn(torch.cat([x_in['SR'], x_noisy], dim=1), gamma)
        return self.loss_func(noise, x_recon)

    def forward(self, x, *args, **kwargs):                        # :272-273
        return self.p_losses(x, *args, **kwargs)

    def res2img(self, img_, img_lr_up, clip_input=None):          # :275-281
        if clip_input is None or clip_input:
            img_ = img_.clamp(-1, 1)
        return img_ / 2.0 + img_lr_up

    def img2res(self, x, img_lr_up, clip_input=None):             # :283-289
        x = (x - img_lr_up) * 2.0
        if clip_input is None or clip_input:
            x = x.clamp(-1, 1)
        return x
