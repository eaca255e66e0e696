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


class _TesrEngineLoss(torch.autograd.Function):
    """The engine's summed loss (Charbonnier terms or squared errors) with the engine's backward behind autograd."""

    @staticmethod
    def forward(ctx, diffusion, x6, gamma, noise, kind, *params):
        eng = diffusion._engine_for_training()
        loss = eng.train_grads(x6, gamma, noise, kind, 1.0)
        ctx.eng = eng
        ctx.keys = [k for k, p in diffusion.denoise_fn.named_parameters() if p.requires_grad]
        ctx.live = {k for k, _, live in eng.schema() if live}
        return torch.tensor(loss, device=x6.device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        scale = float(grad_out)
        grads = [torch.from_numpy(ctx.eng.get_grad(k)).to(grad_out.device) * scale if k in ctx.live else None for k in ctx.keys]
        return (None, None, None, None, None) + tuple(grads)


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

    def _training_batch(self, x_in, noise=None):                  # :224-244, the part before the network
        x_start = x_in['HR'].float()                              # the image itself, not a residual (:225)
        b = x_start.shape[0]
        t = np.random.randint(1, self.num_timesteps + 1)
        gamma = torch.FloatTensor(np.random.uniform(self.sqrt_alphas_cumprod_prev[t - 1],
                                                    self.sqrt_alphas_cumprod_prev[t], size=b)).to(x_start.device)
        gamma = gamma.view(b, -1)
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start, gamma.view(-1, 1, 1, 1), noise)
        return torch.cat([x_in['SR'].float(), x_noisy], dim=1).contiguous(), gamma, noise.contiguous()

    def _engine_loss(self):
        """(engine loss type, divisor the engine's SUM still needs to become what loss_func returns)"""
        return ('charbonnier', True) if self.loss_type == 'l1' else ('l2', False)

    def p_losses(self, x_in, noise=None):                         # :224-250
        """In train mode with autograd on, the result carries the engine's backward (as fastdiffsr_amd.diffusion.GaussianDiffusion.p_losses):
        the reference's `l_pix.sum() / n; backward(); optG.step()` loop works unchanged.  'l1' is the Charbonnier MEAN (:85-90)."""
        x6, gamma, noise = self._training_batch(x_in, noise)
        if self.denoise_fn.training and torch.is_grad_enabled():
            kind, mean = self._engine_loss()
            params = [p for p in self.denoise_fn.parameters() if p.requires_grad]
            loss = _TesrEngineLoss.apply(self, x6, gamma, noise, kind, *params)
            return loss / noise.numel() if mean else loss
        with torch.no_grad():
            x_recon = self.denoise_fn(x6, gamma)
        return self.loss_func(noise, x_recon)

    def optimize_step(self, x_in, lr, betas=(0.9, 0.999), eps=1e-8, noise=None, grad_hook=None, global_batch=None):
        """DDPM.optimize_parameters (model/model.py:47-57) on the device.  With 'l1' the reference's l_pix is the Charbonnier mean divided
        by b*c*h*w once more (model.py:50-52 divides whatever netG returned): sum / (b*c*h*w)^2, which is what this returns and what the
        engine back-propagates."""
        b, c, h, w = x_in['HR'].shape
        gb = int(global_batch) if global_batch is not None else int(b)
        if gb < 1:
            raise ValueError('optimize_step: the global batch is empty')
        n = gb * int(c * h * w)
        kind, mean = self._engine_loss()
        div = float(n) * float(n) if mean else float(n)
        eng = self._engine_for_training()
        if b > 0:
            x6, gamma, noise = self._training_batch(x_in, noise)
            loss = eng.train_grads(x6, gamma, noise, kind, 1.0 / div)
        else:
            eng.zero_grads(x_in['HR'].device)
            loss = 0.0
        if grad_hook is not None:
            grad_hook(eng)
        eng.adam_step(lr, betas, eps)
        self.denoise_fn._engine_ahead = True
        return loss / div
