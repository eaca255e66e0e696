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

    # -- training (ddpm_modules/diffusion.py:260-300; DDPM.optimize_parameters, model/model.py:47-57) ------------------
    def q_sample(self, x_start, t, noise=None):                   # :260-268 (the "fix gama" branch)
        noise = torch.randn_like(x_start) if noise is None else noise
        a = self.sqrt_alphas_cumprod[t].view(-1, 1, 1, 1)
        b = self.sqrt_one_minus_alphas_cumprod[t].view(-1, 1, 1, 1)
        return a * x_start + b * noise

    def _training_batch(self, x_in, noise=None):                  # :279-291, the part before the network
        """The reference's draws: t = torch.randint(0, T, (b,)) then noise = randn_like(x_start), both from torch's generator of
        x_start's device; x_start is the HR image itself (SR3 predicts the noise of the image, not of a residual)."""
        x_start = x_in['HR'].float()
        b = x_start.shape[0]
        t = torch.randint(0, self.num_timesteps, (b,), device=x_start.device).long()
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start, t, noise)
        return torch.cat([x_in['SR'].float(), x_noisy], dim=1).contiguous(), t, noise.contiguous()

    def _engine_for_training(self):
        unet = self.denoise_fn
        unet.sync_weights(for_training=True)
        eng = unet.engine
        eng.set_precision('f32' if self.precision in ('bf16', 'f16') else self.precision)
        eng.set_training(unet.training and unet.cfg.dropout > 0, seed_from_torch=True)
        return eng

    def p_losses(self, x_in, noise=None):                         # :279-297
        """The summed L1 / L2 loss between the noise and the network's prediction.  In train mode with autograd on the result carries
        a grad_fn whose backward is the ENGINE's backward pass (convolutions, GroupNorm, SelfAttention, time embedding), so the
        reference's `l_pix.sum() / n; backward(); optG.step()` works unchanged on the module's Parameters."""
        from ..diffusion import _EngineLoss
        x6, t, noise = self._training_batch(x_in, noise)
        if self.denoise_fn.training and torch.is_grad_enabled():
            params = [p for p in self.denoise_fn.parameters() if p.requires_grad]
            return _EngineLoss.apply(self, x6, t.float(), noise, *params)
        with torch.no_grad():
            x_recon = self.denoise_fn(x6, t)
        return self.loss_func(noise, x_recon)

    def optimize_step(self, x_in, lr, betas=(0.9, 0.999), eps=1e-8, noise=None, grad_hook=None, global_batch=None):
        """DDPM.optimize_parameters entirely on the device (see fastdiffsr_amd.diffusion.GaussianDiffusion.optimize_step): forward,
        loss / (b*c*h*w), backward, Adam on the engine's master copy; `grad_hook(engine)` runs between backward and the optimiser."""
        b, c, h, w = x_in['HR'].shape
        gb = int(global_batch) if global_batch is not None else int(b)
        if gb < 1:
            raise ValueError('optimize_step: the global batch is empty')
        eng = self._engine_for_training()
        if b > 0:
            x6, t, noise = self._training_batch(x_in, noise)
            loss = eng.train_grads(x6, t.float(), noise, self.loss_type, 1.0 / (gb * int(c * h * w)))
        else:
            eng.zero_grads(x_in['HR'].device)
            loss = 0.0
        if grad_hook is not None:
            grad_hook(eng)
        eng.adam_step(lr, betas, eps)
        self.denoise_fn._engine_ahead = True
        return loss / (gb * int(c * h * w))

    def forward(self, x, *args, **kwargs):                        # :299-300
        return self.p_losses(x, *args, **kwargs)
