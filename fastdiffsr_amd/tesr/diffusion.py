"""`GaussianDiffusion` facade of the TESR sibling (FastDiffSR/model/tesr_modules/diffusion.py:66-252) over the HIP
engine.  Same reverse process as FastDiffSR's (continuous noise level sqrt(alpha_bar), :154-181) but the network
predicts the image itself: no img2res / res2img, `p_sample_loop` returns `ret_img[-1]` (:183-204), and the 'l1'
loss is the Charbonnier mean (:85-90, unet.py:956-967)."""
import numpy as np
import torch
from torch import nn

from .. import diffusion as _d


class CharbonnierLoss(nn.Module):                                 # tesr_modules/unet.py:956-967
    def __init__(self, eps=1e-3):
        super().__init__()
        self.eps = eps

    def forward(self, x, y):
        diff = x - y
        return torch.mean(torch.sqrt((diff * diff) + (self.eps * self.eps)))


class GaussianDiffusion(_d.GaussianDiffusion):
    def __init__(self, denoise_fn, image_size, channels=3, loss_type='l1', conditional=True, schedule_opt=None, scale=None):
        super().__init__(denoise_fn, image_size, channels=channels, loss_type=loss_type, conditional=conditional,
                         schedule_opt=schedule_opt)

    def set_loss(self, device):                                   # :85-94
        if self.loss_type == 'l1':
            self.loss_func = CharbonnierLoss().to(device)
        elif self.loss_type == 'l2':
            self.loss_func = nn.MSELoss(reduction='sum').to(device)
        else:
            raise NotImplementedError()

    @torch.no_grad()
    def p_sample_loop(self, x_in, continous=False, noise=None):   # :183-204
        if not self.conditional:
            raise NotImplementedError('only the conditional (super-resolution) branch is implemented')
        device = self.betas.device
        x = x_in.to(device=device, dtype=torch.float32).contiguous()
        T = self.num_timesteps
        if noise is None and getattr(self, 'rng', 'torch') != 'engine':
            noise = torch.empty((T,) + tuple(x.shape), device=device, dtype=torch.float32)
            noise[0] = torch.randn(x.shape, device=device)        # :196
            for k in range(1, T):
                noise[k] = torch.randn_like(x)                    # :180, t > 0
        self.denoise_fn.sync_weights()
        eng = self.denoise_fn.engine
        eng.set_precision(self.precision)
        if not continous:
            img = eng.sample(x, noise)
            return img[-1]                                        # ret_img[-1]: the last image of the batch (:203-204)
        img, traj = eng.sample(x, noise, want_traj=True)
        inter = (1 | (T // 10))                                   # :186
        frames = [x]                                              # ret_img = x (:197)
        for k, t in enumerate(reversed(range(T))):
            if t % inter == 0:
                frames.append(traj[k])
        return torch.cat(frames, dim=0)

    def p_losses(self, x_in, noise=None):                         # :224-250
        x_start = x_in['HR']                                      # the image itself, not a residual (:225)
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
