"""`UNet` facade of the TESR denoiser (FastDiffSR/model/tesr_modules/unet.py:168-269) over the HIP engine:
FastDiffSR's Block / ResnetBlock / shift-only FeatureWiseAffine and continuous noise-level embedding
(PositionalEncoding -> Linear -> Swish -> Linear), with SelfAttention (`<blk>.attn.{norm,qkv,out}`) where the
resolution is in attn_res and in mid[0]; no dead `.conv`, no CLAM/SLAM.  Checkpoints exchange key for key."""
from .. import unet as _u


class UNet(_u.UNet):
    _variant = 'tesr'

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4, 8, 8),
                 attn_res=(8), res_blocks=3, dropout=0, with_noise_level_emb=True, image_size=128):
        super().__init__(in_channel=in_channel, out_channel=out_channel, inner_channel=inner_channel,
                         norm_groups=norm_groups, channel_mults=channel_mults, attn_res=attn_res, res_blocks=res_blocks,
                         dropout=dropout, with_noise_level_emb=with_noise_level_emb, image_size=image_size)
