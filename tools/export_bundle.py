"""Write the bundle directory examples/fdsr_demo.c reads: hyper-parameters, a reference-format
state_dict flattened to one file, schedule scalars, a cond batch and (optionally) the noise planes.

    python tools/export_bundle.py <dir> [--batch 2 --size 64 --no-noise --checkpoint gen.pth]
"""
import argparse
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL  # noqa: E402
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars  # noqa: E402
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs  # noqa: E402


def write_bundle(out_dir, cfg, sd, schedule_opt, cond, noise=None, seed=4321):
    """sd: {key (no 'denoise_fn.' prefix): float32 ndarray in the reference layout}; cond [B,3,H,W];
    noise [T,B,3,H,W] or None (the engine then draws with `seed`)."""
    os.makedirs(out_dir, exist_ok=True)
    B, _, H, W = cond.shape
    mults = list(cfg.channel_mults) + [0] * (8 - len(cfg.channel_mults))
    attn = list(cfg.attn_res)[:8] + [0] * (8 - min(8, len(cfg.attn_res)))
    ints = [cfg.in_channel, cfg.out_channel, cfg.inner_channel, cfg.norm_groups, len(cfg.channel_mults)] + mults + \
           [cfg.res_blocks, 1 if cfg.variant == 'ddpm' else 0, cfg.image_size, min(8, len(cfg.attn_res))] + attn + [B, H, W, seed]
    with open(os.path.join(out_dir, 'config.bin'), 'wb') as f:
        f.write(struct.pack('<%di' % len(ints), *ints))
    with open(os.path.join(out_dir, 'weights.bin'), 'wb') as f:
        for key, arr in sd.items():
            a = np.ascontiguousarray(arr, dtype=np.float32)
            kb = key.encode()
            f.write(struct.pack('<i', len(kb)) + kb + struct.pack('<i', a.ndim) + struct.pack('<%dq' % a.ndim, *a.shape))
            f.write(a.tobytes())
    bufs, sqrt_prev = schedule_buffers(schedule_opt)
    sc = sampling_scalars(bufs, sqrt_prev)
    T = int(len(sc['noise_level']))
    with open(os.path.join(out_dir, 'schedule.bin'), 'wb') as f:
        f.write(struct.pack('<i', T))
        for name in ('noise_level', 'sqrt_recip', 'sqrt_recipm1', 'coef1', 'coef2', 'sigma'):
            f.write(np.ascontiguousarray(sc[name], dtype=np.float32).tobytes())
    np.ascontiguousarray(cond, dtype=np.float32).tofile(os.path.join(out_dir, 'cond.bin'))
    npath = os.path.join(out_dir, 'noise.bin')
    if noise is not None:
        np.ascontiguousarray(noise, dtype=np.float32).tofile(npath)
    elif os.path.exists(npath):
        os.remove(npath)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('dir')
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--no-noise', action='store_true')
    ap.add_argument('--checkpoint', help='reference *_gen.pth (torch.load); default: the synthetic random-init weights')
    a = ap.parse_args()
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    if a.checkpoint:
        import torch
        ck = torch.load(a.checkpoint, map_location='cpu')
        sd = {k[len('denoise_fn.'):]: v.float().numpy() for k, v in ck.items() if k.startswith('denoise_fn.')}
    else:
        sd = synth_state_dict(cfg, 0)
    cond, noise = synth_inputs(a.batch, a.size, a.size, 20)
    write_bundle(a.dir, cfg, sd, FASTDIFFSR_SCHEDULE_VAL, cond.numpy(), None if a.no_noise else noise.numpy())
    print('bundle written to', a.dir)


if __name__ == '__main__':
    main()
