"""`GaussianDiffusion` facade of the GDP sibling (FastDiffSR/model/gdp_modules/diffusion.py:64-300) over the HIP engine.
The network predicts x_0 (clamped, :189-195) from cat([x_t, cond]) and the integer time step; every step draws noise
(masked at t = 0, :206-211); `p_sample_loop` returns `ret_img[-1]` (:238-241); both loss types are the summed MSE (:83-89)."""
import torch
from torch import nn

from .. import diffusion as _d


class GaussianDiffusion(_d.GaussianDiffusion):
    def __init__(self, denoise_fn, image_size, channels=3, loss_type='l2', conditional=True, schedule_opt=None, scale=4):
        super().__init__(denoise_fn, image_size, channels=channels, loss_type=loss_type, conditional=conditional,
                         schedule_opt=schedule_opt)

    def set_loss(self, device):                                   # :83-89: 'l1' is MSE too
        if self.loss_type in ('l1', 'l2'):
            self.loss_func = nn.MSELoss(reduction='sum').to(device)
        else:
            raise NotImplementedError()

    @torch.no_grad()
    def p_sample_loop(self, x_in, continous=False, noise=None):   # :213-241
        if not self.conditional:
            raise NotImplementedError('only the conditional (super-resolution) branch is implemented')
        device = self.betas.device
        x = x_in.to(device=device, dtype=torch.float32).contiguous()
        T = self.num_timesteps
        if noise is None and getattr(self, 'rng', 'torch') != 'engine':
            noise = torch.empty((T + 1,) + tuple(x.shape), device=device, dtype=torch.float32)
            for k in range(T + 1):                                # randn(shape) (:229), then noise_like per step (:206), all torch.randn
                noise[k] = torch.randn(x.shape, device=device)
        self.denoise_fn.sync_weights()
        eng = self.denoise_fn.engine
        eng.set_precision(self.precision)
        eng.set_training(self.denoise_fn.training and self.denoise_fn.cfg.dropout > 0, seed_from_torch=True)
        if not continous:
            return eng.sample(x, noise)[-1]                       # ret_img[-1]: the last image of the batch
        img, traj = eng.sample(x, noise, want_traj=True)
        inter = (1 | (T // 10))                                   # :215
        frames = [x]
        for k, t in enumerate(reversed(range(T))):
            if t % inter == 0:
                frames.append(traj[k])
        return torch.cat(frames, dim=0)

    def q_sample(self, x_start, t, noise=None):                   # :258-265
        noise = torch.randn_like(x_start) if noise is None else noise
        a = self.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1)
        s = self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1)
        return a * x_start + s * noise

    def p_losses(self, x_in, noise=None):                         # :266-285 (forward value; the training kernels serve FastDiffSR)
        x_start = x_in['HR']
        b = x_start.shape[0]
        t = torch.randint(0, self.num_timesteps, (b,), device=x_start.device).long()
        noise = torch.randn_like(x_start) if noise is None else noise
        x_t = self.q_sample(x_start, t, noise)
        with torch.no_grad():
            x_recon = self.denoise_fn(torch.cat([x_t, x_in['SR']], dim=1), t)
        return self.loss_func(x_recon, x_start)
