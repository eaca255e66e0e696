"""Python handle around libfdsr_hip.so (include/fdsr.h).  PyTorch is used only for
device memory, streams and tensors at the boundary."""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .arch import UNetConfig


LOSS_TYPES = {'l1': 0, 'l2': 1, 'charbonnier': 2}     # include/fdsr.h: loss_l2 of fdsr_train_grads


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    def __init__(self, cfg: UNetConfig):
        self.lib = _lib.load()
        self.cfg = cfg
        c = _lib.FdsrConfig()
        c.in_channel, c.out_channel, c.inner_channel = cfg.in_channel, cfg.out_channel, cfg.inner_channel
        c.norm_groups = cfg.norm_groups
        c.n_mults = len(cfg.channel_mults)
        if c.n_mults > _lib.FDSR_MAX_MULTS:
            raise ValueError('too many channel multipliers')
        for i, m in enumerate(cfg.channel_mults):
            c.channel_mults[i] = int(m)
        c.res_blocks, c.dropout, c.image_size = cfg.res_blocks, float(cfg.dropout), int(cfg.image_size)
        c.variant = {'fastdiffsr': 0, 'ddpm': 1, 'tesr': 2, 'gdp': 3}[cfg.variant]
        c.n_attn_res = min(len(cfg.attn_res), _lib.FDSR_MAX_MULTS)
        for i, r in enumerate(cfg.attn_res[:_lib.FDSR_MAX_MULTS]):
            c.attn_res[i] = int(r)
        h = C.c_void_p()
        rc = self.lib.fdsr_create(C.byref(c), C.byref(h))
        if rc != 0:
            raise _lib.FdsrError(rc, self.lib.fdsr_last_error(None).decode())
        self.h = h
        self._ws = None
        self._keep = None
        self._gstream = None
        self.trained = False            # at least one optimiser step ran (there is optimiser state to save)
        self.T = 0
        self.precision = 'f32'
        # f16x3 range guard (include/fdsr.h: fdsr_check_saturation): after every sample / unet_forward / training step in f16x3
        # the engine asks whether a raw conv input left the f16 range (one stream synchronisation per call).  What happens then
        # is decided HERE, for every model family and caller alike: on_saturation = 'f32' (default) re-runs that call on the
        # exact-fp32 kernels (no range limit), warns once and goes back to f16x3; 'raise' raises FdsrSaturated.
        self.check_saturation = True
        self.on_saturation = 'f32'

    def __del__(self):
        try:
            if getattr(self, 'h', None):
                self.lib.fdsr_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- schema / weights ---------------------------------------------------
    def schema(self):
        """(key, shape, live) of every checkpoint tensor; fixed at fdsr_create, so it is read once."""
        if getattr(self, '_schema_cache', None) is not None:
            return self._schema_cache
        out = []
        n = self.lib.fdsr_num_weights(self.h)
        buf = C.create_string_buffer(256)
        shape = (C.c_int64 * 4)()
        nd, live = C.c_int(), C.c_int()
        for i in range(n):
            _lib.check(self.h, self.lib.fdsr_weight_info(self.h, i, buf, 256, shape, C.byref(nd), C.byref(live)))
            out.append((buf.value.decode(), tuple(int(shape[k]) for k in range(nd.value)), bool(live.value)))
        self._schema_cache = out
        self._shape_of = {k: sh for k, sh, _ in out}
        return out

    def load_weight(self, key, value):
        a = np.ascontiguousarray(value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else value,
                                 dtype=np.float32)
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        _lib.check(self.h, self.lib.fdsr_load_weight(self.h, key.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))

    def load_state_dict(self, sd, prefix=''):
        """sd: key -> tensor/ndarray for every key of the UNet schema (strict)."""
        keys = [k for k, _, _ in self.schema()]
        missing = [k for k in keys if prefix + k not in sd]
        if missing:
            raise KeyError(f'missing keys in state_dict: {missing[:4]}... ({len(missing)})')
        for k in keys:
            self.load_weight(k, sd[prefix + k])

    @property
    def weights_complete(self):
        return bool(self.lib.fdsr_weights_complete(self.h))

    # -- schedule -------------------------------------------------------------
    def set_schedule(self, scalars):
        T = int(len(scalars['noise_level']))
        arrs = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in scalars.items()}
        s = _lib.FdsrSchedule()
        s.n_timestep = T
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        s.noise_level, s.sqrt_recip, s.sqrt_recipm1 = fp(arrs['noise_level']), fp(arrs['sqrt_recip']), fp(arrs['sqrt_recipm1'])
        s.coef1, s.coef2, s.sigma = fp(arrs['coef1']), fp(arrs['coef2']), fp(arrs['sigma'])
        _lib.check(self.h, self.lib.fdsr_set_schedule(self.h, C.byref(s)))
        self.T = T

    # -- execution --------------------------------------------------------------
    def workspace_bytes(self, B, H, W):
        n = C.c_size_t()
        _lib.check(self.h, self.lib.fdsr_workspace_bytes(self.h, B, H, W, C.byref(n)))
        return int(n.value)

    def _workspace(self, B, H, W, device):
        need = self.workspace_bytes(B, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    @staticmethod
    def _check_input(t, name):
        if not t.is_cuda:
            raise RuntimeError(f'{name} must live on the GPU: the HIP engine has no CPU path')
        if t.dtype != torch.float32:
            raise TypeError(f'{name} must be float32')
        return t.contiguous()

    _warned_saturated = False
    saturation_fallbacks = 0       # calls (all engines of the process) re-run on the exact-fp32 kernels after the f16x3 range guard tripped

    def _with_range_fallback(self, call):
        """Run `call()`; if the f16x3 range guard trips and the policy says so, run it again in exact fp32."""
        try:
            return call()
        except _lib.FdsrSaturated:
            if self.on_saturation != 'f32':
                raise
        Engine.saturation_fallbacks += 1
        if not Engine._warned_saturated:
            import warnings
            warnings.warn('fastdiffsr_amd: a raw convolution input exceeded the f16 range in f16x3 mode; this call was re-run on the '
                          'exact-fp32 kernels (set the precision to "f32" to avoid the double work)', RuntimeWarning)
            Engine._warned_saturated = True
        prec = self.precision
        self.set_precision('f32')
        try:
            return call()       # engine-drawn noise / dropout masks are drawn afresh, as in any second call
        finally:
            self.set_precision(prec)

    def unet_forward(self, x, noise_level):
        return self._with_range_fallback(lambda: self._unet_forward(x, noise_level))

    def _unet_forward(self, x, noise_level):
        x = self._check_input(x, 'x')
        B, Cin, H, W = x.shape
        nl = self._check_input(noise_level.to(x.device), 'noise_level').reshape(-1)
        if nl.numel() != B:
            raise ValueError('noise_level must have one entry per sample')
        ws = self._workspace(B, H, W, x.device)
        out = torch.empty(B, self.cfg.out_channel, H, W, device=x.device, dtype=torch.float32)
        st = torch.cuda.current_stream(x.device).cuda_stream
        _lib.check(self.h, self.lib.fdsr_unet_forward(self.h, _ptr(x), _ptr(nl), _ptr(out), B, H, W, _ptr(ws), ws.numel(),
                                                      C.c_void_p(st)))
        self._keep = (x, nl)
        self._saturation_check(st)
        return out

    def _saturation_check(self, stream):
        if self.precision != 'f16x3' or not self.check_saturation:
            return
        rc = self.lib.fdsr_check_saturation(self.h, C.c_void_p(stream))
        if rc == _lib.FDSR_E_SATURATED:
            raise _lib.FdsrSaturated(rc, self.lib.fdsr_last_error(self.h).decode())
        _lib.check(self.h, rc)

    def sample(self, cond, noise=None, want_traj=False, graph=False, out=None, traj=None):
        """noise: [T,B,3,H,W] (parity runs: the reference's draws), or None: the engine draws
        inside the loop (Philox, see set_seed) like the reference's in-loop randn_like."""
        return self._with_range_fallback(lambda: self._sample(cond, noise, want_traj, graph, out, traj))

    def _sample(self, cond, noise, want_traj, graph, out, traj):
        cond = self._check_input(cond, 'cond')
        B, _, H, W = cond.shape
        if noise is not None:
            noise = self._check_input(noise, 'noise')
            nT = self.T + (1 if self.cfg.variant in ('ddpm', 'gdp') else 0)     # SR3 / GDP draw noise at t = 0 too (masked)
            if tuple(noise.shape) != (nT, B, 3, H, W):
                raise ValueError(f'noise must be [{nT},{B},3,{H},{W}], got {tuple(noise.shape)}')
        ws = self._workspace(B, H, W, cond.device)
        if out is None:
            out = torch.empty(B, 3, H, W, device=cond.device, dtype=torch.float32)
        if want_traj and traj is None:
            traj = torch.empty(self.T, B, 3, H, W, device=cond.device, dtype=torch.float32)
        cur = torch.cuda.current_stream(cond.device)
        flags = _lib.FDSR_SAMPLE_GRAPH if graph else 0
        args = (self.h, _ptr(cond), _ptr(noise), _ptr(out), _ptr(traj if want_traj else None), B, H, W, _ptr(ws), ws.numel())
        if graph and cur.cuda_stream == 0:
            # stream capture cannot run on the NULL stream: replay on a stream of our own, ordered after and
            # before the caller's current stream
            if self._gstream is None or self._gstream.device != cond.device:
                self._gstream = torch.cuda.Stream(cond.device)
            self._gstream.wait_stream(cur)
            _lib.check(self.h, self.lib.fdsr_sample(*args, C.c_void_p(self._gstream.cuda_stream), flags))
            cur.wait_stream(self._gstream)
            self._keep = (cond, noise, out, traj)
            self._saturation_check(self._gstream.cuda_stream)
        else:
            _lib.check(self.h, self.lib.fdsr_sample(*args, C.c_void_p(cur.cuda_stream), flags))
            self._keep = (cond, noise, out, traj)
            self._saturation_check(cur.cuda_stream)
        return (out, traj) if want_traj else out

    def set_seed(self, seed):
        """Seed of the engine-side noise; also resets its per-call counter."""
        _lib.check(self.h, self.lib.fdsr_set_seed(self.h, C.c_uint64(int(seed) & (2 ** 64 - 1))))

    def randn(self, B, H, W, plane, device='cuda'):
        """Noise plane `plane` under the current call counter, [B,3,H,W] (generator tests)."""
        dst = torch.empty(B, 3, H, W, device=device, dtype=torch.float32)
        st = torch.cuda.current_stream(dst.device).cuda_stream
        _lib.check(self.h, self.lib.fdsr_randn(self.h, _ptr(dst), B, H, W, int(plane), C.c_void_p(st)))
        return dst

    def set_precision(self, mode):
        """'f32' (exact fp32 MFMA), 'f16x3' (fp32-grade split-f16 MFMA) or 'bf16'."""
        code = _lib.PRECISIONS[mode] if isinstance(mode, str) else int(mode)
        _lib.check(self.h, self.lib.fdsr_set_precision(self.h, code))
        self.precision = mode

    # -- training step ------------------------------------------------------------
    def train_workspace_bytes(self, B, H, W):
        n = C.c_size_t()
        _lib.check(self.h, self.lib.fdsr_train_workspace_bytes(self.h, B, H, W, C.byref(n)))
        return int(n.value)

    def _check_train(self, rc):
        if rc == _lib.FDSR_E_SATURATED:
            raise _lib.FdsrSaturated(rc, self.lib.fdsr_last_error(self.h).decode())
        _lib.check(self.h, rc)

    def train_grads(self, x, noise_level, target, loss_type='l1', loss_scale=1.0):
        """Forward + loss + backward on the device (include/fdsr.h: fdsr_train_grads).  Returns the unscaled
        summed loss (python float); the gradients stay on the device (get_grad / adam_step)."""
        return self._with_range_fallback(lambda: self._train_grads(x, noise_level, target, loss_type, loss_scale))

    def _train_grads(self, x, noise_level, target, loss_type, loss_scale):
        x = self._check_input(x, 'x')
        target = self._check_input(target, 'target')
        B, _, H, W = x.shape
        nl = self._check_input(noise_level.to(x.device), 'noise_level').reshape(-1)
        if nl.numel() != B or tuple(target.shape) != (B, 3, H, W):
            raise ValueError('noise_level must be [B] and target [B,3,H,W]')
        need = self.train_workspace_bytes(B, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != x.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        ws = self._ws
        loss = C.c_float()
        st = torch.cuda.current_stream(x.device).cuda_stream
        self._check_train(self.lib.fdsr_train_grads(self.h, _ptr(x), _ptr(nl), _ptr(target), LOSS_TYPES[loss_type],
                                                    C.c_float(float(loss_scale)), C.byref(loss), B, H, W, _ptr(ws), ws.numel(),
                                                    C.c_void_p(st)))
        self._keep = (x, nl, target)
        return float(loss.value)

    def train_grads_pairs(self, hr, sr, gamma, noise=None, loss_type='l1', loss_scale=1.0):
        """The same step from the training pair (include/fdsr.h: fdsr_train_grads_pairs): img2res, q_sample and the channel concat
        happen in the engine's input kernel; noise=None: the engine draws the target noise itself (Philox, set_seed)."""
        return self._with_range_fallback(lambda: self._train_grads_pairs(hr, sr, gamma, noise, loss_type, loss_scale))

    def _train_grads_pairs(self, hr, sr, gamma, noise, loss_type, loss_scale):
        hr, sr = self._check_input(hr, 'hr'), self._check_input(sr, 'sr')
        B, _, H, W = hr.shape
        g = self._check_input(gamma.to(hr.device), 'gamma').reshape(-1)
        if g.numel() != B or tuple(sr.shape) != (B, 3, H, W) or tuple(hr.shape) != (B, 3, H, W):
            raise ValueError('hr / sr must be [B,3,H,W] and gamma [B]')
        if noise is not None:
            noise = self._check_input(noise, 'noise')
            if tuple(noise.shape) != (B, 3, H, W):
                raise ValueError('noise must be [B,3,H,W]')
        need = self.train_workspace_bytes(B, H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != hr.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=hr.device)
        ws, loss = self._ws, C.c_float()
        st = torch.cuda.current_stream(hr.device).cuda_stream
        self._check_train(self.lib.fdsr_train_grads_pairs(self.h, _ptr(hr), _ptr(sr), _ptr(g), _ptr(noise), LOSS_TYPES[loss_type],
                                                          C.c_float(float(loss_scale)), C.byref(loss), B, H, W, _ptr(ws), ws.numel(),
                                                          C.c_void_p(st)))
        self._keep = (hr, sr, g, noise)
        return float(loss.value)

    def zero_grads(self, device='cuda'):
        """A zero gradient arena (a data-parallel rank whose shard of the batch is empty still joins the all-reduce)."""
        self.grad_arena().zero_()

    def adam_step(self, lr, betas=(0.9, 0.999), eps=1e-8):
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(self.h, self.lib.fdsr_adam_step(self.h, C.c_float(lr), C.c_float(betas[0]), C.c_float(betas[1]), C.c_float(eps),
                                                   C.c_void_p(st)))
        self.trained = True

    def _fetch(self, fn, key):
        self.schema()
        shape = self._shape_of[key]
        a = np.empty(shape, dtype=np.float32)
        _lib.check(self.h, fn(self.h, key.encode(), a.ctypes.data_as(C.c_void_p)))
        return a

    def get_weight(self, key):
        """Master copy of one executed tensor (checkpoint layout), after any optimiser steps."""
        torch.cuda.synchronize()
        return self._fetch(self.lib.fdsr_get_weight, key)

    def get_grad(self, key):
        torch.cuda.synchronize()
        return self._fetch(self.lib.fdsr_get_grad, key)

    supports_dropout = True

    def set_training(self, on=True, seed_from_torch=False):
        """nn.Module.train()/.eval(): Dropout(p) of block2 live or not (the fp32-grade precisions only when live).
        seed_from_torch: key the masks of the coming call by a draw from torch's (CPU) generator, so that a run repeats
        under torch.manual_seed as the reference's nn.Dropout does."""
        _lib.check(self.h, self.lib.fdsr_set_training(self.h, int(bool(on))))
        if on and seed_from_torch:
            seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
            _lib.check(self.h, self.lib.fdsr_set_dropout_seed(self.h, C.c_uint64(seed)))

    def dropout_mask(self, block):
        """The multiplicative mask (keep / (1-p)) the last training-mode forward applied in front of `block`'s block2
        conv, as an NCHW float tensor -- what the oracle's `dropout_masks[block]` takes."""
        off, n, hh, ww, ch, sc = C.c_void_p(), C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_float()
        _lib.check(self.h, self.lib.fdsr_debug_dropout_mask(self.h, block.encode(), C.byref(off), C.byref(n), C.byref(hh), C.byref(ww),
                                                            C.byref(ch), C.byref(sc)))
        o, cnt = int(off.value or 0), n.value * hh.value * ww.value * ch.value
        keep = self._ws[o:o + cnt].view(n.value, hh.value, ww.value, ch.value).permute(0, 3, 1, 2).float()
        return keep * sc.value

    def optimizer_state(self, key):
        """(exp_avg, exp_avg_sq, step) of torch.optim.Adam for one executed tensor."""
        self.schema()
        shape = self._shape_of[key]
        m, v, step = np.empty(shape, np.float32), np.empty(shape, np.float32), C.c_int()
        torch.cuda.synchronize()
        _lib.check(self.h, self.lib.fdsr_get_optimizer_state(self.h, key.encode(), m.ctypes.data_as(C.c_void_p),
                                                             v.ctypes.data_as(C.c_void_p), C.byref(step)))
        return m, v, int(step.value)

    def set_optimizer_state(self, key, exp_avg, exp_avg_sq, step):
        m = np.ascontiguousarray(exp_avg, dtype=np.float32)
        v = np.ascontiguousarray(exp_avg_sq, dtype=np.float32)
        _lib.check(self.h, self.lib.fdsr_set_optimizer_state(self.h, key.encode(), m.ctypes.data_as(C.c_void_p),
                                                             v.ctypes.data_as(C.c_void_p), int(step)))
        self.trained = True

    def grad_arena(self):
        """The engine's gradient arena as a torch tensor sharing its memory (zero-copy): all-reduce it in place."""
        ptr, n = C.c_void_p(), C.c_size_t()
        _lib.check(self.h, self.lib.fdsr_grad_arena(self.h, C.byref(ptr), C.byref(n)))

        class _Arena:
            __cuda_array_interface__ = {'shape': (int(n.value),), 'typestr': '<f4', 'data': (int(ptr.value), False), 'version': 2}
        return torch.as_tensor(_Arena(), device='cuda')

    # -- introspection ----------------------------------------------------------
    def set_debug(self, on=True):
        _lib.check(self.h, self.lib.fdsr_set_debug(self.h, int(on)))

    def debug_tensor(self, name):
        """Output of reference module `name` from the last forward, as an NCHW torch tensor."""
        off = C.c_void_p()
        n, hh, ww, ch = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self.h, self.lib.fdsr_debug_tensor(self.h, name.encode(), C.byref(off), C.byref(n), C.byref(hh),
                                                      C.byref(ww), C.byref(ch)))
        o = int(off.value or 0)
        cnt = n.value * hh.value * ww.value * ch.value
        eb = C.c_int()
        _lib.check(self.h, self.lib.fdsr_debug_tensor_elem_bytes(self.h, name.encode(), C.byref(eb)))
        if eb.value == 2:       # bf16 / f16 mode keeps activations in 16 bits
            flat = self._ws[o:o + 2 * cnt].view(torch.float16 if self.precision == 'f16' else torch.bfloat16).float()
        else:
            flat = self._ws[o:o + 4 * cnt].view(torch.float32)
        return flat.view(n.value, hh.value, ww.value, ch.value).permute(0, 3, 1, 2).contiguous()

    def profile_begin(self):
        _lib.check(self.h, self.lib.fdsr_profile_begin(self.h))

    def profile_end(self):
        n, ms, fl, by = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        _lib.check(self.h, self.lib.fdsr_profile_end(self.h, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        return dict(launches=n.value, conv_ms=ms.value, conv_flops=fl.value, conv_bytes=by.value)
