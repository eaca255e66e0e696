"""Static description of the FastDiffSR denoiser (layer list + checkpoint schema).

This is host-side bookkeeping shared by the nn.Module facade, the synthetic
weight generator and the tests.  It derives, from the handful of UNet
hyper-parameters the reference's factory passes
(reference FastDiffSR/model/networks.py:94-104), the exact layer sequence of
reference FastDiffSR/model/fastdiffsr_modules/unet.py:224-297 and the
checkpoint key/shape schema.  The HIP library builds the
same plan natively from the same hyper-parameters (csrc/fdsr_plan.cpp); the
two are cross-checked by tests/test_host_logic.py through the C ABI
(`fdsr_num_weights` / `fdsr_weight_info`).
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass
class UNetConfig:
    in_channel: int = 6
    out_channel: int = 3
    inner_channel: int = 32
    norm_groups: int = 32
    channel_mults: Tuple[int, ...] = (1, 2, 4, 4)
    attn_res: Tuple[int, ...] = (8,)
    res_blocks: int = 3
    dropout: float = 0.0
    image_size: int = 256
    # 'fastdiffsr' (model/fastdiffsr_modules), 'tesr' (model/tesr_modules: the same blocks and noise-level embedding with
    # SR3's SelfAttention placement) or 'ddpm' (the SR3 sibling, model/ddpm_modules: integer-time
    # embedding, Swish before the per-block Linear, SelfAttention where the resolution is in attn_res)
    variant: str = 'fastdiffsr'

    def __post_init__(self):
        self.channel_mults = tuple(int(m) for m in self.channel_mults)
        if isinstance(self.attn_res, int):
            self.attn_res = (self.attn_res,)
        self.attn_res = tuple(self.attn_res) if self.attn_res is not None else ()


@dataclass
class Layer:
    kind: str           # 'conv_in' | 'res' | 'down' | 'up' | 'final'
    name: str           # reference module path, e.g. 'downs.4'
    cin: int
    cout: int
    with_attn: bool = False
    skip_from: int = -1  # for 'res' in the up path: index into feats popped (cat order x, skip)
    cskip: int = 0       # channels of the popped skip


def build_layers(cfg: UNetConfig) -> List[Layer]:
    """Layer sequence of UNet.__init__/forward (reference unet.py:252-323)."""
    ic = cfg.inner_channel
    sr3 = cfg.variant in ('ddpm', 'tesr')      # SelfAttention where the resolution is in attn_res
    now_res = cfg.image_size
    layers: List[Layer] = []
    feat_channels = [ic]
    pre = ic
    layers.append(Layer('conv_in', 'downs.0', cfg.in_channel, ic))
    idx = 1
    nm = len(cfg.channel_mults)
    for ind in range(nm):
        is_last = ind == nm - 1
        cm = ic * cfg.channel_mults[ind]
        for _ in range(cfg.res_blocks):
            layers.append(Layer('res', f'downs.{idx}', pre, cm, with_attn=sr3 and now_res in cfg.attn_res))
            idx += 1
            feat_channels.append(cm)
            pre = cm
        if not is_last:
            layers.append(Layer('down', f'downs.{idx}', pre, pre))
            idx += 1
            feat_channels.append(pre)
            now_res //= 2
    layers.append(Layer('res', 'mid.0', pre, pre, with_attn=True))
    layers.append(Layer('res', 'mid.1', pre, pre))
    idx = 0
    for ind in reversed(range(nm)):
        is_last = ind < 1
        cm = ic * cfg.channel_mults[ind]
        for _ in range(cfg.res_blocks + 1):
            cs = feat_channels.pop()
            layers.append(Layer('res', f'ups.{idx}', pre + cs, cm, cskip=cs, with_attn=sr3 and now_res in cfg.attn_res))
            idx += 1
            pre = cm
        if not is_last:
            layers.append(Layer('up', f'ups.{idx}', pre, pre))
            idx += 1
            now_res *= 2
    layers.append(Layer('final', 'final_conv', pre, cfg.out_channel))
    return layers


def param_schema(cfg: UNetConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """Checkpoint keys (without the 'denoise_fn.' prefix) -> shapes, in the
    order torch's state_dict() emits them for the reference UNet
    (SURVEY.md App. C).  Includes the 22 never-executed `<blk>.conv` 1x1 layers
    (reference unet.py:212) because strict load_state_dict needs them."""
    ic = cfg.inner_channel
    sd: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    if cfg.variant in ('ddpm', 'tesr'):
        return _param_schema_sr3(cfg)
    if cfg.variant == 'gdp':
        from .gdp.arch import gdp_param_schema
        return gdp_param_schema(cfg)
    sd['noise_level_mlp.1.weight'] = (ic * 4, ic)
    sd['noise_level_mlp.1.bias'] = (ic * 4,)
    sd['noise_level_mlp.3.weight'] = (ic, ic * 4)
    sd['noise_level_mlp.3.bias'] = (ic,)
    for L in build_layers(cfg):
        p = L.name
        if L.kind == 'conv_in':
            sd[f'{p}.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{p}.bias'] = (L.cout,)
        elif L.kind in ('down', 'up'):
            sd[f'{p}.conv.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{p}.conv.bias'] = (L.cout,)
        elif L.kind == 'res':
            r = f'{p}.res_block'
            sd[f'{r}.noise_func.noise_func.0.weight'] = (L.cout, ic)
            sd[f'{r}.noise_func.noise_func.0.bias'] = (L.cout,)
            sd[f'{r}.block1.block.0.weight'] = (L.cin,)
            sd[f'{r}.block1.block.0.bias'] = (L.cin,)
            sd[f'{r}.block1.block.3.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{r}.block1.block.3.bias'] = (L.cout,)
            sd[f'{r}.block2.block.0.weight'] = (L.cout,)
            sd[f'{r}.block2.block.0.bias'] = (L.cout,)
            sd[f'{r}.block2.block.3.weight'] = (L.cout, L.cout, 3, 3)
            sd[f'{r}.block2.block.3.bias'] = (L.cout,)
            if L.cin != L.cout:
                sd[f'{r}.res_conv.weight'] = (L.cout, L.cin, 1, 1)
                sd[f'{r}.res_conv.bias'] = (L.cout,)
            # dead 1x1 conv owned by ResnetBlocWithAttn (unet.py:212)
            sd[f'{p}.conv.weight'] = (L.cout, L.cout, 1, 1)
            sd[f'{p}.conv.bias'] = (L.cout,)
            if L.with_attn:
                sd[f'{p}.ca.fc1.weight'] = (L.cout // 16, L.cout, 1, 1)
                sd[f'{p}.ca.fc2.weight'] = (L.cout, L.cout // 16, 1, 1)
                sd[f'{p}.sa.conv1.weight'] = (1, 2, 7, 7)
        elif L.kind == 'final':
            sd[f'{p}.block.0.weight'] = (L.cin,)
            sd[f'{p}.block.0.bias'] = (L.cin,)
            sd[f'{p}.block.3.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{p}.block.3.bias'] = (L.cout,)
    return sd


def _param_schema_sr3(cfg: UNetConfig):
    """Checkpoint schema of model/ddpm_modules/unet.py (SR3): time_mlp with its inv_freq buffer,
    per-block `mlp.1` Linear, SelfAttention (norm, qkv, out) where with_attn; no dead `.conv`."""
    ic = cfg.inner_channel
    sd = OrderedDict()
    tesr = cfg.variant == 'tesr'     # model/tesr_modules/unet.py: FastDiffSR's noise-level embedding + this block layout
    if tesr:
        sd['noise_level_mlp.1.weight'] = (ic * 4, ic)
        sd['noise_level_mlp.1.bias'] = (ic * 4,)
        sd['noise_level_mlp.3.weight'] = (ic, ic * 4)
        sd['noise_level_mlp.3.bias'] = (ic,)
    else:
        sd['time_mlp.0.inv_freq'] = (ic // 2,)
        sd['time_mlp.1.weight'] = (ic * 4, ic)
        sd['time_mlp.1.bias'] = (ic * 4,)
        sd['time_mlp.3.weight'] = (ic, ic * 4)
        sd['time_mlp.3.bias'] = (ic,)
    nf = 'noise_func.noise_func.0' if tesr else 'mlp.1'
    for L in build_layers(cfg):
        p = L.name
        if L.kind == 'conv_in':
            sd[f'{p}.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{p}.bias'] = (L.cout,)
        elif L.kind in ('down', 'up'):
            sd[f'{p}.conv.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{p}.conv.bias'] = (L.cout,)
        elif L.kind == 'res':
            r = f'{p}.res_block'
            sd[f'{r}.{nf}.weight'] = (L.cout, ic)
            sd[f'{r}.{nf}.bias'] = (L.cout,)
            sd[f'{r}.block1.block.0.weight'] = (L.cin,)
            sd[f'{r}.block1.block.0.bias'] = (L.cin,)
            sd[f'{r}.block1.block.3.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{r}.block1.block.3.bias'] = (L.cout,)
            sd[f'{r}.block2.block.0.weight'] = (L.cout,)
            sd[f'{r}.block2.block.0.bias'] = (L.cout,)
            sd[f'{r}.block2.block.3.weight'] = (L.cout, L.cout, 3, 3)
            sd[f'{r}.block2.block.3.bias'] = (L.cout,)
            if L.cin != L.cout:
                sd[f'{r}.res_conv.weight'] = (L.cout, L.cin, 1, 1)
                sd[f'{r}.res_conv.bias'] = (L.cout,)
            if L.with_attn:
                sd[f'{p}.attn.norm.weight'] = (L.cout,)
                sd[f'{p}.attn.norm.bias'] = (L.cout,)
                sd[f'{p}.attn.qkv.weight'] = (L.cout * 3, L.cout, 1, 1)
                sd[f'{p}.attn.out.weight'] = (L.cout, L.cout, 1, 1)
                sd[f'{p}.attn.out.bias'] = (L.cout,)
        elif L.kind == 'final':
            sd[f'{p}.block.0.weight'] = (L.cin,)
            sd[f'{p}.block.0.bias'] = (L.cin,)
            sd[f'{p}.block.3.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{p}.block.3.bias'] = (L.cout,)
    return sd


def dead_keys(cfg: UNetConfig):
    out = []
    if cfg.variant in ('ddpm', 'tesr', 'gdp'):
        return out
    for L in build_layers(cfg):
        if L.kind == 'res':
            out += [f'{L.name}.conv.weight', f'{L.name}.conv.bias']
    return out


SCHEDULE_BUFFERS = (
    'betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_alphas_cumprod',
    'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod',
    'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod',
    'posterior_variance', 'posterior_log_variance_clipped',
    'posterior_mean_coef1', 'posterior_mean_coef2',
)

# FastDiffSR x4 val config (reference config/sr_fastdiffsr_test_64_256.json)
FASTDIFFSR_UNET = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32,
                       channel_mults=(1, 2, 4, 4), attn_res=(16,), res_blocks=2,
                       dropout=0.2, image_size=256)
# SR3 x4 val config (reference config/sr_ddpm_test_64_256.json)
SR3_UNET = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 1, 2, 2, 4, 4),
                attn_res=(16,), res_blocks=2, dropout=0.2, image_size=256, variant='ddpm')
SR3_SCHEDULE_VAL = dict(schedule='linear', n_timestep=1000, linear_start=1e-4, linear_end=2e-2)
# TESR x4 val config (reference config/sr_tesr_test_64_256.json)
TESR_UNET = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 4, 8, 8),
                 attn_res=(16,), res_blocks=2, dropout=0.2, image_size=256, variant='tesr')
TESR_SCHEDULE_VAL = dict(schedule='linear', n_timestep=2000, linear_start=1e-6, linear_end=1e-2)
# GDP x4 val config (reference config/sr_gdp_test_64_256.json): the reference's UNet keeps model_channels = 128 and
# attention_resolutions = (32, 16, 8) whatever the config's inner_channel / attn_res say (gdp_modules/unet.py:561-590)
GDP_UNET = dict(in_channel=6, out_channel=3, inner_channel=128, norm_groups=32, channel_mults=(1, 2, 4, 8), attn_res=(32, 16, 8),
                res_blocks=2, dropout=0.2, image_size=256, variant='gdp')
GDP_SCHEDULE_VAL = dict(schedule='linear', n_timestep=1000, linear_start=1e-4, linear_end=2e-2)
FASTDIFFSR_SCHEDULE_VAL = dict(schedule='linear_cosine', n_timestep=20,
                               linear_start=1e-6, linear_end=1e-2)
