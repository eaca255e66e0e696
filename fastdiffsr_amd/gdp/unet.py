"""`UNet` facade of the GDP denoiser (FastDiffSR/model/gdp_modules/unet.py:530-800: the guided-diffusion UNet) over
the HIP engine.  The constructor mirrors the reference's: define_G passes in_channel, out_channel, norm_groups,
inner_channel, channel_mults, attn_res, res_blocks, dropout, image_size (model/networks.py:94-104) and, exactly as in the
reference, `inner_channel` and `attn_res` are accepted and IGNORED -- the network keeps model_channels = 128 and
attention_resolutions = (32, 16, 8) unless those two are given.  Checkpoints exchange key for key."""
import torch

from .. import unet as _u
from ..arch import UNetConfig, param_schema


class UNet(_u.UNet):
    _variant = 'gdp'

    def __init__(self, image_size, in_channel=3, model_channels=128, out_channel=3, res_blocks=2, attention_resolutions=(32, 16, 8),
                 dropout=0, channel_mults=(1, 2, 4, 8), conv_resample=True, dims=2, num_classes=None, use_checkpoint=False,
                 use_fp16=False, num_heads=4, num_head_channels=64, num_heads_upsample=-1, use_scale_shift_norm=True,
                 resblock_updown=True, use_new_attention_order=False, inner_channel=32, norm_groups=32, attn_res=(8),
                 with_time_emb=True):
        if not (conv_resample and dims == 2 and num_classes is None and not use_fp16 and num_head_channels == 64 and
                use_scale_shift_norm and resblock_updown and not use_new_attention_order and norm_groups == 32):
            raise NotImplementedError('the HIP engine implements the GDP UNet as define_G instantiates it (the constructor defaults)')
        torch.nn.Module.__init__(self)
        self.cfg = UNetConfig(in_channel=in_channel, out_channel=out_channel, inner_channel=model_channels, norm_groups=32,
                              channel_mults=tuple(channel_mults), attn_res=tuple(attention_resolutions), res_blocks=res_blocks,
                              dropout=dropout, image_size=image_size, variant='gdp')
        self._schema = param_schema(self.cfg)
        for key, shape in self._schema.items():
            parts = key.split('.')
            holder = _u._holder_path(self, parts[:-1])
            holder.register_parameter(parts[-1], torch.nn.Parameter(self._default_init(parts[-1], shape, key)))
        with torch.no_grad():                                  # zero_module(...) of the reference (unet.py:85-91, :348, :426, :753)
            for key, p in self.named_parameters():
                if '.out_layers.3.' in key or '.proj_out.' in key or key.startswith('out.2.'):
                    p.zero_()
        self._engine = None
        self._uploaded_version = None
        self._engine_ahead = False

    def forward(self, x, timesteps, y=None):                   # :773-800; timesteps: 1-D batch of (integer) time steps
        assert y is None, 'the model is not class-conditional'
        return super().forward(x, timesteps.reshape(-1, 1).to(torch.float32))
