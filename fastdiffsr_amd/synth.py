"""Portable synthetic ("random-init") weights for the FastDiffSR UNet.

Every tensor of the checkpoint schema (arch.param_schema) is defined by a
closed-form, platform-stable generator seeded from its own key, so the same
arrays are produced in the build container (where they are loaded into the
imported reference to make tests/golden/*) and on the GPU box (where they are
loaded into the HIP engine and the oracle).  No 95 MB checkpoint is committed;
tests pin the generator through a SHA-256 of the concatenated bytes.

Scales: conv/linear weights ~ N(0, 0.7^2 * 2/fan_in) keep activations O(1)
through the 29-layer stack; GroupNorm gamma = 1 + 0.1 N, every bias / beta =
0.05 N, so bias, affine and FiLM-shift arithmetic is exercised (PyTorch's
default init has gamma=1, beta=0).  SURVEY.md section 7 stage 0 / App. E.
"""
import hashlib
from collections import OrderedDict

import numpy as np

from .arch import UNetConfig, param_schema


def _rng(key: str, seed: int) -> np.random.Generator:
    h = hashlib.sha256(f'{seed}:{key}'.encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], 'little')))


def synth_tensor(key: str, shape, seed: int = 0) -> np.ndarray:
    if key.endswith('inv_freq'):     # TimeEmbedding buffer (ddpm_modules/unet.py:22-27), formed in fp32 like torch does
        import torch
        dim = 2 * shape[0]
        return torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * (-np.log(10000) / dim)).numpy()
    g = _rng(key, seed)
    x = g.standard_normal(size=shape, dtype=np.float64)
    if key.endswith('.bias'):
        x *= 0.05
    elif len(shape) == 1:          # GroupNorm gamma (incl. attn.norm)
        x = 1.0 + 0.1 * x
    else:
        fan_in = int(np.prod(shape[1:]))
        x *= 0.7 * np.sqrt(2.0 / fan_in)
    return x.astype(np.float32)


def synth_state_dict(cfg: UNetConfig, seed: int = 0, prefix: str = '') -> "OrderedDict[str, np.ndarray]":
    """All UNet tensors (numpy fp32), keys optionally prefixed (e.g. 'denoise_fn.')."""
    out = OrderedDict()
    for k, shp in param_schema(cfg).items():
        out[prefix + k] = synth_tensor(k, shp, seed)
    return out


def state_dict_sha256(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v, dtype=np.float32).tobytes())
    return h.hexdigest()


def synth_inputs(batch: int, height: int, width: int, steps: int = 20,
                 cond_seed: int = 1234, noise_seed: int = 4321):
    """Synthetic conditioning image U(-1,1) and noise N(0,1).

    Returned as torch CPU tensors: cond [B,3,H,W], noise [steps,B,3,H,W];
    noise[0] is x_T, noise[k] feeds the step t = steps-k."""
    import torch
    g = torch.Generator().manual_seed(cond_seed)
    cond = torch.rand(batch, 3, height, width, generator=g) * 2.0 - 1.0
    g = torch.Generator().manual_seed(noise_seed)
    noise = torch.randn(steps, batch, 3, height, width, generator=g)
    return cond, noise
