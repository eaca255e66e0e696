"""`GaussianDiffusion` facade of the SR3 sibling (FastDiffSR/model/ddpm_modules/diffusion.py:78-300) over the
HIP engine: discrete-time reverse process, the network sees the integer t, the output is x_0 itself."""
import torch
from torch import nn

from ..schedule import schedule_buffers, sampling_scalars


class GaussianDiffusion(nn.Module):
    def __init__(self, denoise_fn, image_size, channels=3, loss_type='l1', conditional=True, schedule_opt=None):
        super().__init__()
        self.channels = channels
        self.image_size = image_size
        self.denoise_fn = denoise_fn
        self.loss_type = loss_type
        self.conditional = conditional
        self.precision = 'f16x3'

    def set_loss(self, device):                                   # :96-102
        if self.loss_type == 'l1':
            self.loss_func = nn.L1Loss(reduction='sum').to(device)
        elif self.loss_type == 'l2':
            self.loss_func = nn.MSELoss(reduction='sum').to(device)
        else:
            raise NotImplementedError()

    def set_new_noise_schedule(self, schedule_opt, device):       # :104-150
        bufs, sqrt_prev = schedule_buffers(schedule_opt)
        self.num_timesteps = int(bufs['betas'].shape[0])
        for k, v in bufs.items():
            self.register_buffer(k, torch.tensor(v, dtype=torch.float32, device=device))
        self.denoise_fn.engine.set_schedule(sampling_scalars(bufs, sqrt_prev))

    @torch.no_grad()
    def p_sample_loop(self, x_in, continous=False, noise=None):   # :198-227
        if not self.conditional:
            raise NotImplementedError('only the conditional (super-resolution) branch is implemented')
        device = self.betas.device
        x = x_in.to(device=device, dtype=torch.float32).contiguous()
        T = self.num_timesteps
        if noise is None:   # randn(shape), then one noise_like draw per step, t = 0 included (:189-196, :215)
            noise = torch.empty((T + 1,) + tuple(x.shape), device=device, dtype=torch.float32)
            for k in range(T + 1):
                noise[k] = torch.randn(x.shape, device=device)
        self.denoise_fn.sync_weights()
        eng = self.denoise_fn.engine
        eng.set_precision(self.precision)
        if not continous:
            img = eng.sample(x, noise)
            return img[-1] if img.shape[0] == 1 else img           # the reference returns ret_img[-1]
        img, traj = eng.sample(x, noise, want_traj=True)
        inter = (1 | (T // 10))                                   # :200
        frames = [x]
        for k, t in enumerate(reversed(range(T))):
            if t % inter == 0:
                frames.append(traj[k])
        return torch.cat(frames, dim=0)

    @torch.no_grad()
    def super_resolution(self, x_in, continous=False):            # :236-238
        return self.p_sample_loop(x_in, continous)

    def forward(self, x, *args, **kwargs):                        # training: SURVEY 8f-3
        raise NotImplementedError('training through the HIP engine is not implemented')
