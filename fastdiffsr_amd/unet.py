"""`UNet` facade: the constructor and call surface of the reference denoiser
(FastDiffSR/model/fastdiffsr_modules/unet.py:224-323) over the HIP engine.

The module owns nn.Parameters in exactly the reference's module hierarchy, so
`state_dict()` / `load_state_dict(strict=True)` exchange reference checkpoints
key for key (317 tensors, including the 22 never-executed `<blk>.conv` layers of
unet.py:212).  `forward(x, time)` runs libfdsr_hip.so; there is no PyTorch
implementation of the network in this package.
"""
import math

import torch
from torch import nn

from .arch import UNetConfig, param_schema
from .engine import Engine


class _Holder(nn.Module):
    """Parameter container standing in for one reference sub-module."""

    def extra_repr(self):
        return ', '.join(f'{k}{tuple(v.shape)}' for k, v in self._parameters.items())


def _holder_path(root, parts):
    m = root
    for p in parts:
        if p not in m._modules:
            m.add_module(p, _Holder())
        m = m._modules[p]
    return m


class UNet(nn.Module):
    _variant = 'fastdiffsr'

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4, 4),
                 attn_res=(8), res_blocks=3, dropout=0, with_noise_level_emb=True, image_size=256):
        super().__init__()
        if not with_noise_level_emb:
            raise NotImplementedError('the HIP engine implements the noise-level / time conditioned UNet only')
        self.cfg = UNetConfig(in_channel=in_channel, out_channel=out_channel, inner_channel=inner_channel,
                              norm_groups=norm_groups, channel_mults=tuple(channel_mults), attn_res=attn_res,
                              res_blocks=res_blocks, dropout=dropout, image_size=image_size, variant=self._variant)
        self._schema = param_schema(self.cfg)
        for key, shape in self._schema.items():
            parts = key.split('.')
            holder = _holder_path(self, parts[:-1])
            if parts[-1] == 'inv_freq':      # TimeEmbedding's registered buffer (ddpm_modules/unet.py:22-27)
                from .synth import synth_tensor
                holder.register_buffer('inv_freq', torch.from_numpy(synth_tensor(key, shape)))
            else:
                holder.register_parameter(parts[-1], nn.Parameter(self._default_init(parts[-1], shape, key)))
        self._engine = None
        self._ptensors = None
        self._uploaded_version = None
        self._engine_ahead = False      # the engine holds newer weights than the nn.Parameters (device-side Adam)

    def _default_init(self, leaf, shape, key):
        """PyTorch's default initialisers (Conv2d/Linear: kaiming_uniform(a=sqrt 5) and
        bias ~ U(+-1/sqrt(fan_in)); GroupNorm: weight 1, bias 0)."""
        t = torch.empty(shape)
        sibling = self._schema.get(key[:-len(leaf)] + 'weight')
        if len(shape) == 1 and sibling is not None and len(sibling) == 1:   # GroupNorm: the only 1-D `.weight` tensors
            return t.fill_(1.0 if leaf == 'weight' else 0.0)
        if leaf == 'weight':
            nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            return t
        wkey = key[:-len('bias')] + 'weight'
        if wkey not in self._schema:
            return t.zero_()
        wshape = self._schema[wkey]
        fan_in = 1
        for d in wshape[1:]:
            fan_in *= d
        bound = 1.0 / math.sqrt(fan_in)
        return t.uniform_(-bound, bound)

    def train(self, mode=True):
        """nn.Module.train() without the walk over ~330 parameter-holder children (they have no forward and no mode of their own):
        DDPM.test() flips eval() / train() around every sampling call (model/model.py:60-68), 0.5 ms per call at batch 1."""
        self.training = bool(mode)
        return self

    # -- engine plumbing ---------------------------------------------------------
    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self.cfg)
        return self._engine

    def _apply(self, fn, *args, **kwargs):
        self._ptensors = None        # .to() / .cuda() replace the buffer objects (nn.Module._apply)
        return super()._apply(fn, *args, **kwargs)

    def _param_version(self):
        """(version counter, storage address) of every parameter and buffer.  The tensor OBJECTS are fixed at construction (the schema),
        so their list is made once: walking the ~330 parameter-holder children cost 1.1 ms per sampling call, a third of what the B = 1
        val loop spent on the host per image (round 6; load_state_dict / .to() / optimiser steps change versions and addresses, not objects)."""
        ts = self._ptensors
        if ts is None:
            ts = self._ptensors = list(self.parameters()) + list(self.buffers())
        return tuple((p._version, p.data_ptr()) for p in ts)

    def sync_weights(self, force=False, for_training=False):
        """Upload parameters to the engine if they changed since the last upload.

        After device-side optimiser steps the ENGINE holds the newest weights (`_engine_ahead`).  The training path
        (`for_training`) then touches nothing: no pull, no upload -- the Parameters are refreshed lazily, by
        state_dict() / pull_weights() / the next sampling call.  Only when somebody wrote to the Parameters in the
        meantime (their version counters moved) does the module lead again."""
        if self._engine_ahead:
            if for_training and not force and self._param_version() == self._uploaded_version:
                return
            self.pull_weights()
        ver = self._param_version()
        if force or ver != self._uploaded_version:
            sd = {k: v for k, v in self.state_dict().items()}
            self.engine.load_state_dict(sd)
            self._uploaded_version = ver

    def pull_weights(self):
        """Copy the engine's master copy (after device-side optimiser steps) into the module's Parameters."""
        if not self._engine_ahead:
            return
        named = dict(self.named_parameters())
        with torch.no_grad():
            for key, _, live in self.engine.schema():
                if live and key in named:
                    named[key].copy_(torch.from_numpy(self.engine.get_weight(key)))
        self._engine_ahead = False
        self._uploaded_version = self._param_version()

    def state_dict(self, *args, **kwargs):
        self.pull_weights()
        return super().state_dict(*args, **kwargs)

    def forward(self, x, time):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
            raise NotImplementedError(
                'autograd through UNet.forward is not provided: gradients come from the engine\'s own backward pass '
                '(GaussianDiffusion.forward in train mode / optimize_step); call under torch.no_grad() for inference')
        self.sync_weights()
        # nn.Dropout(p) in block2 is live whenever .training (unet.py:89-101): the engine draws the masks
        # (Philox) and runs the exact-fp32 kernels for such a forward
        live_dropout = self.training and self.cfg.dropout > 0
        self.engine.set_training(live_dropout, seed_from_torch=True)
        if live_dropout and getattr(self.engine, 'precision', 'f32') in ('bf16', 'f16'):
            self.engine.set_precision('f16x3')       # live dropout needs one of the fp32-grade modes
        return self.engine.unet_forward(x, time)    # the f16x3 range guard and its f32 re-run live in Engine (on_saturation)
