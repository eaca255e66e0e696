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


class _EngineLoss(torch.autograd.Function):
    """loss = p_losses(...) with the engine's backward pass behind autograd: backward() copies the engine's
    gradients (scaled by the incoming gradient of the loss, e.g. 1 / (b*c*h*w)) into the Parameters' .grad."""

    @staticmethod
    def forward(ctx, diffusion, x6, gamma, noise, *params):
        eng = diffusion._engine_for_training()
        loss = eng.train_grads(x6, gamma, noise, getattr(diffusion, 'engine_loss_type', diffusion.loss_type), 1.0)
        ctx.eng = eng
        ctx.keys = [k for k, p in diffusion.denoise_fn.named_parameters() if p.requires_grad]
        ctx.live = {k for k, _, live in eng.schema() if live}
        return torch.tensor(loss, device=x6.device, dtype=torch.float32)

    @staticmethod
    def backward(ctx, grad_out):
        scale = float(grad_out)
        grads = []
        for k in ctx.keys:          # never-executed tensors (unet.py:212) get no gradient, as in torch
            grads.append(torch.from_numpy(ctx.eng.get_grad(k)).to(grad_out.device) * scale if k in ctx.live else None)
        return (None, None, None, None) + tuple(grads)


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
        # the T-step loop as ONE captured hipGraph (north_star; include/fdsr.h FDSR_SAMPLE_GRAPH): 'auto' replays from the second
        # call of a shape on (capture + instantiation of ~3 600 nodes is not worth it for a one-off shape), 'on' from the first,
        # 'off' never.  Replays need stable addresses: the facade keeps cond / noise / out / trajectory buffers per shape, copies the
        # caller's tensors in and hands clones out.
        self.graph = 'auto'
        self._gbuf = {}
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
        unet = self.denoise_fn
        live_dropout = unet.training and unet.cfg.dropout > 0
        mode = getattr(self, 'graph', 'auto')
        key = (tuple(x.shape), bool(continous), str(device))
        seen = self._gbuf.get(key)
        use_graph = mode != 'off' and not live_dropout and (mode == 'on' or seen is not None)
        if mode == 'auto' and seen is None:
            self._gbuf[key] = {}          # this shape has been sampled once: the next call captures
        draw_here = noise is None and getattr(self, 'rng', 'torch') != 'engine'
        if use_graph:
            buf = self._gbuf.setdefault(key, {})
            if 'cond' not in buf:
                if len(self._gbuf) > 4:                       # shapes come and go: keep the buffers of the last few
                    for old in [k_ for k_ in self._gbuf if k_ != key][:len(self._gbuf) - 4]:
                        del self._gbuf[old]
                buf['cond'] = torch.empty_like(x)
                buf['out'] = torch.empty_like(x)
                buf['traj'] = torch.empty((T,) + tuple(x.shape), device=device, dtype=torch.float32) if continous else None
            buf['cond'].copy_(x)
            if noise is not None or draw_here:
                if buf.get('noise') is None:
                    buf['noise'] = torch.empty((T,) + tuple(x.shape), device=device, dtype=torch.float32)
                nb = buf['noise']
            if noise is not None:
                nb.copy_(noise)
            elif draw_here:
                # same draws, same order as the reference: randn(shape) then randn_like per step t>0 (:207, :189)
                for k in range(T):
                    torch.randn(x.shape, device=device, out=nb[k])
            noise_arg = nb if (noise is not None or draw_here) else None    # None: the engine draws inside the loop (Philox)
        elif draw_here:
            noise_arg = torch.empty((T,) + tuple(x.shape), device=device, dtype=torch.float32)
            noise_arg[0] = torch.randn(x.shape, device=device)
            for k in range(1, T):
                noise_arg[k] = torch.randn_like(x)
        else:
            noise_arg = noise
        unet.sync_weights()
        eng = unet.engine
        eng.set_precision(self.precision)
        # the reference samples after netG.eval() (model.py:60); in .train() mode its Dropout would be live here too,
        # and so it is (the engine then insists on the fp32 kernels)
        eng.set_training(live_dropout, seed_from_torch=True)
        # (the f16x3 range guard and its exact-fp32 re-run live in Engine.sample: Engine.on_saturation)
        if use_graph:
            res = eng.sample(buf['cond'], noise_arg, want_traj=bool(continous), graph=True, out=buf['out'], traj=buf['traj'])
            res = (res[0].clone(), res[1]) if continous else res.clone()
        else:
            res = eng.sample(x, noise_arg, want_traj=bool(continous), graph=False)
        if not continous:
            return res
        img, traj = res
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

    def _training_batch(self, x_in, noise=None):                  # :242-266, the part before the network
        """x_start, the numpy draws of t and gamma (reference: numpy global RNG), noise, q_sample."""
        x_start = self.img2res(x_in['HR'], x_in['SR'])
        b = x_start.shape[0]
        t = np.random.randint(1, self.num_timesteps + 1)
        gamma = torch.FloatTensor(np.random.uniform(self.sqrt_alphas_cumprod_prev[t - 1],
                                                    self.sqrt_alphas_cumprod_prev[t], size=b)).to(x_start.device)
        gamma = gamma.view(b, -1)
        noise = torch.randn_like(x_start) if noise is None else noise
        x_noisy = self.q_sample(x_start, gamma.view(-1, 1, 1, 1), noise)
        return torch.cat([x_in['SR'], x_noisy], dim=1).contiguous(), gamma, noise.contiguous()

    def p_losses(self, x_in, noise=None):                         # :242-270
        """The summed L1 / L2 loss, RNG draws as in the reference (numpy for t and gamma).  In train mode with
        autograd on, the result carries a grad_fn whose backward is the ENGINE's backward pass: the reference's
        `l_pix.sum() / n; l_pix.backward(); optG.step()` (model.py:49-56) then works unchanged on the module's
        Parameters.  The all-device fast path is optimize_step()."""
        x6, gamma, noise = self._training_batch(x_in, noise)
        if self.denoise_fn.training and torch.is_grad_enabled():
            params = [p for p in self.denoise_fn.parameters() if p.requires_grad]
            return _EngineLoss.apply(self, x6, gamma, noise, *params)
        with torch.no_grad():
            x_recon = self.denoise_fn(x6, gamma)
        return self.loss_func(noise, x_recon)

    def _engine_for_training(self):
        unet = self.denoise_fn
        unet.sync_weights(for_training=True)
        eng = unet.engine
        # 'f32' (everything exact fp32) or 'f16x3' (every convolution -- forward, input and weight gradients -- fp32-grade on
        # split-f16 MFMAs)
        eng.set_precision('f32' if self.precision in ('bf16', 'f16') else self.precision)
        eng.set_training(unet.training and unet.cfg.dropout > 0, seed_from_torch=True)   # Dropout(p) of block2 is live in .train() mode
        return eng

    def optimize_step(self, x_in, lr, betas=(0.9, 0.999), eps=1e-8, noise=None, grad_hook=None, global_batch=None):
        """DDPM.optimize_parameters (model/model.py:47-57) entirely on the device: forward, loss / (b*c*h*w),
        backward, Adam on the engine's master copy.  Returns the SUMMED loss of this rank's samples divided by the
        global element count (a python float; summed over ranks it is the reference's l_pix).  grad_hook(engine) runs
        between backward and the optimiser (data-parallel all-reduce of the gradient arena, parallel.allreduce_grads).

        Data parallel: `global_batch` is the number of samples of the WHOLE step over all ranks -- every rank divides by
        global_batch*c*h*w and the all-reduce SUMS the arenas, so per-sample weights are right for ragged shards too; a
        rank whose shard is empty (b == 0) contributes a zero arena and still takes part in the all-reduce."""
        b, c, h, w = x_in['HR'].shape
        gb = int(global_batch) if global_batch is not None else int(b)
        if gb < 1:
            raise ValueError('optimize_step: the global batch is empty')
        eng = self._engine_for_training()
        if b > 0:
            # the RNG draws are the reference's (numpy for t and gamma, torch for the noise: diffusion.py:246-259); img2res,
            # q_sample and the channel concat happen in the engine's input kernel (the arithmetic of _training_batch's tensors)
            hr, sr = x_in['HR'].float().contiguous(), x_in['SR'].float().contiguous()
            t = np.random.randint(1, self.num_timesteps + 1)
            gamma = torch.FloatTensor(np.random.uniform(self.sqrt_alphas_cumprod_prev[t - 1], self.sqrt_alphas_cumprod_prev[t],
                                                        size=b)).to(hr.device)
            noise = torch.randn_like(hr) if noise is None else noise
            loss = eng.train_grads_pairs(hr, sr, gamma, noise.contiguous(), self.loss_type, 1.0 / (gb * int(c * h * w)))
        else:
            eng.zero_grads(x_in['HR'].device)
            loss = 0.0
        if grad_hook is not None:
            grad_hook(eng)
        eng.adam_step(lr, betas, eps)
        self.denoise_fn._engine_ahead = True
        return loss / (gb * int(c * h * w))

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
