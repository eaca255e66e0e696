"""`UNet` facade of the SR3 denoiser (FastDiffSR/model/ddpm_modules/unet.py:136-259) over the HIP engine:
integer-time embedding, Swish before the per-block Linear, SelfAttention where the resolution is in
attn_res (and in mid[0]); checkpoints exchange key for key (incl. the `time_mlp.0.inv_freq` buffer)."""
from .. import unet as _u


class UNet(_u.UNet):
    _variant = 'ddpm'

    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2, 4, 8, 8),
                 attn_res=(8), res_blocks=3, dropout=0, with_time_emb=True, image_size=128):
        super().__init__(in_channel=in_channel, out_channel=out_channel, inner_channel=inner_channel,
                         norm_groups=norm_groups, channel_mults=channel_mults, attn_res=attn_res, res_blocks=res_blocks,
                         dropout=dropout, with_noise_level_emb=with_time_emb, image_size=image_size)

    def forward(self, x, time):
        return super().forward(x, time.float())
