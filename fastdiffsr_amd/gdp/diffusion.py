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

    engine_loss_type = 'l2'                                      # :83-89: both loss types are the summed MSE

    def _training_batch(self, x_in, noise=None):                  # :277-290, the part before the network
        """The reference's draws: t = torch.randint(0, T, (b,)), then noise = randn_like(x_start), both from torch's generator of
        x_start's device; the network sees cat([x_t, SR]) and its target is x_start = HR itself (it predicts x_0)."""
        x_start = x_in['HR'].float()
        b = x_start.shape[0]
        t = torch.randint(0, self.num_timesteps, (b,), device=x_start.device).long()
        noise = torch.randn_like(x_start) if noise is None else noise
        x_t = self.q_sample(x_start, t, noise)
        return torch.cat([x_t, x_in['SR'].float()], dim=1).contiguous(), t, x_start.contiguous()

    def p_losses(self, x_in, noise=None):                         # :277-299
        """The summed MSE between the network's x_0 and HR.  In train mode with autograd on the result carries a grad_fn whose backward
        is the ENGINE's backward pass (scale-shift GroupNorms, pooled / upsampled ResBlocks, multi-head attention, the time MLP), so
        the reference's `l_pix.sum() / n; backward(); optG.step()` (model.py:49-56) works unchanged on the module's Parameters."""
        x6, t, target = self._training_batch(x_in, noise)
        if self.denoise_fn.training and torch.is_grad_enabled():
            params = [p for p in self.denoise_fn.parameters() if p.requires_grad]
            return _d._EngineLoss.apply(self, x6, t.float(), target, *params)
        with torch.no_grad():
            x_recon = self.denoise_fn(x6, t)
        return self.loss_func(x_recon, target)

    def optimize_step(self, x_in, lr, betas=(0.9, 0.999), eps=1e-8, noise=None, grad_hook=None, global_batch=None):
        """DDPM.optimize_parameters entirely on the device (see fastdiffsr_amd.diffusion.GaussianDiffusion.optimize_step): forward,
        loss / (b*c*h*w), backward, Adam on the engine's master copy; `grad_hook(engine)` runs between backward and the optimiser."""
        b, c, h, w = x_in['HR'].shape
        gb = int(global_batch) if global_batch is not None else int(b)
        if gb < 1:
            raise ValueError('optimize_step: the global batch is empty')
        eng = self._engine_for_training()
        if b > 0:
            x6, t, target = self._training_batch(x_in, noise)
            loss = eng.train_grads(x6, t.float(), target, 'l2', 1.0 / (gb * int(c * h * w)))
        else:
            eng.zero_grads(x_in['HR'].device)
            loss = 0.0
        if grad_hook is not None:
            grad_hook(eng)
        eng.adam_step(lr, betas, eps)
        self.denoise_fn._engine_ahead = True
        return loss / (gb * int(c * h * w))
