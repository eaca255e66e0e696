"""ORACLE -- TEST INFRASTRUCTURE ONLY.  The oracle's OWN statement of which modules a UNet has and how they are wired,
derived from the hyper-parameters alone, so that the checker does not share its wiring table with the product
(fastdiffsr_amd/arch.py derives the same list independently; tests/test_oracle_golden.py cross-checks the two, and the
goldens generated from the reference pin the result).

`wiring(cfg)` follows the constructor of the reference's UNet (FastDiffSR/model/fastdiffsr_modules/unet.py:252-297; the
ddpm / tesr siblings differ only in where `with_attn` is set: ddpm_modules/unet.py:177-215, tesr_modules/unet.py) and
returns one record per module of `downs`, `mid`, `ups` and the final Block, in forward order.  `cfg` is any object with
the attributes in_channel, out_channel, inner_channel, channel_mults, attn_res, res_blocks, image_size, variant."""
from collections import namedtuple

Module = namedtuple('Module', 'kind name cin cout with_attn cskip')


def wiring(cfg):
    attn_by_resolution = getattr(cfg, 'variant', 'fastdiffsr') in ('ddpm', 'tesr')
    attn_res = tuple(cfg.attn_res) if not isinstance(cfg.attn_res, int) else (cfg.attn_res,)
    width = cfg.inner_channel
    res = cfg.image_size
    skips = [width]                                              # feat_channels (:256)
    mods = [Module('conv_in', 'downs.0', cfg.in_channel, width, False, 0)]
    mults = list(cfg.channel_mults)
    for level, m in enumerate(mults):                            # :259-273
        target = cfg.inner_channel * m
        for _ in range(cfg.res_blocks):
            mods.append(Module('res', 'downs.%d' % len(mods), width, target, attn_by_resolution and res in attn_res, 0))
            width = target
            skips.append(width)
        if level != len(mults) - 1:
            mods.append(Module('down', 'downs.%d' % len(mods), width, width, False, 0))
            skips.append(width)
            res //= 2
    mods.append(Module('res', 'mid.0', width, width, True, 0))   # :275-280
    mods.append(Module('res', 'mid.1', width, width, False, 0))
    n_up = 0
    for level in range(len(mults) - 1, -1, -1):                  # :282-294
        target = cfg.inner_channel * mults[level]
        for _ in range(cfg.res_blocks + 1):
            s = skips.pop()
            mods.append(Module('res', 'ups.%d' % n_up, width + s, target, attn_by_resolution and res in attn_res, s))
            n_up += 1
            width = target
        if level >= 1:
            mods.append(Module('up', 'ups.%d' % n_up, width, width, False, 0))
            n_up += 1
            res *= 2
    mods.append(Module('final', 'final_conv', width, cfg.out_channel, False, 0))   # :296
    return mods
