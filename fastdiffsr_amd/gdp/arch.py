"""Layer list and checkpoint schema of the GDP denoiser (FastDiffSR/model/gdp_modules/unet.py:530-770) as the
reference's factory instantiates it (model/networks.py:88-104): use_scale_shift_norm, resblock_updown, heads of 64
channels.  `cfg.inner_channel` carries model_channels (the reference's constructor ignores its `inner_channel` argument
and keeps the default 128), `cfg.attn_res` the attention_resolutions (downsample rates; reference default (32, 16, 8))."""
from collections import OrderedDict
from dataclasses import dataclass
from typing import List


@dataclass
class GdpLayer:
    kind: str          # 'conv_in' | 'res' | 'attn' | 'out'
    name: str          # module path, e.g. 'input_blocks.3.0'
    cin: int = 0       # channels of x (before the concat)
    cskip: int = 0     # channels of the popped skip (output path)
    cout: int = 0
    mode: str = ''     # '' | 'down' | 'up'
    push: bool = False         # the output of this layer closes an input block: hs.append
    pop: bool = False          # this ResBlock opens an output block: cat([h, hs.pop()])
    block: str = ''            # the TimestepEmbedSequential this layer closes ('' if it does not close one)


def gdp_layers(cfg) -> List[GdpLayer]:
    mc = cfg.inner_channel
    mults, nb = cfg.channel_mults, cfg.res_blocks
    attn = set(cfg.attn_res)
    L: List[GdpLayer] = []
    ch = mc * mults[0]
    L.append(GdpLayer('conv_in', 'input_blocks.0.0', cin=cfg.in_channel, cout=ch, push=True, block='input_blocks.0'))
    chans = [ch]
    ds, idx = 1, 1
    for level, m in enumerate(mults):
        for _ in range(nb):
            p = f'input_blocks.{idx}'
            L.append(GdpLayer('res', p + '.0', cin=ch, cout=mc * m))
            ch = mc * m
            if ds in attn:
                L.append(GdpLayer('attn', p + '.1', cin=ch, cout=ch))
            L[-1].push, L[-1].block = True, p
            chans.append(ch)
            idx += 1
        if level != len(mults) - 1:
            p = f'input_blocks.{idx}'
            L.append(GdpLayer('res', p + '.0', cin=ch, cout=ch, mode='down', push=True, block=p))
            chans.append(ch)
            ds *= 2
            idx += 1
    L.append(GdpLayer('res', 'middle_block.0', cin=ch, cout=ch))
    L.append(GdpLayer('attn', 'middle_block.1', cin=ch, cout=ch))
    L.append(GdpLayer('res', 'middle_block.2', cin=ch, cout=ch, block='middle_block'))
    idx = 0
    for level, m in list(enumerate(mults))[::-1]:
        for i in range(nb + 1):
            ich = chans.pop()
            p = f'output_blocks.{idx}'
            sub = 0
            L.append(GdpLayer('res', f'{p}.{sub}', cin=ch, cskip=ich, cout=mc * m, pop=True))
            sub += 1
            ch = mc * m
            if ds in attn:
                L.append(GdpLayer('attn', f'{p}.{sub}', cin=ch, cout=ch))
                sub += 1
            if level and i == nb:
                L.append(GdpLayer('res', f'{p}.{sub}', cin=ch, cout=ch, mode='up'))
                ds //= 2
            L[-1].block = p
            idx += 1
    L.append(GdpLayer('out', 'out', cin=ch, cout=cfg.out_channel, block='out'))
    return L


def gdp_param_schema(cfg) -> "OrderedDict[str, tuple]":
    mc = cfg.inner_channel
    ted = 4 * mc
    sd = OrderedDict()
    sd['time_embed.0.weight'] = (ted, mc)
    sd['time_embed.0.bias'] = (ted,)
    sd['time_embed.2.weight'] = (ted, ted)
    sd['time_embed.2.bias'] = (ted,)
    for L in gdp_layers(cfg):
        p = L.name
        if L.kind == 'conv_in':
            sd[f'{p}.weight'] = (L.cout, L.cin, 3, 3)
            sd[f'{p}.bias'] = (L.cout,)
        elif L.kind == 'res':
            cin = L.cin + L.cskip
            sd[f'{p}.in_layers.0.weight'] = (cin,)
            sd[f'{p}.in_layers.0.bias'] = (cin,)
            sd[f'{p}.in_layers.2.weight'] = (L.cout, cin, 3, 3)
            sd[f'{p}.in_layers.2.bias'] = (L.cout,)
            sd[f'{p}.emb_layers.1.weight'] = (2 * L.cout, ted)
            sd[f'{p}.emb_layers.1.bias'] = (2 * L.cout,)
            sd[f'{p}.out_layers.0.weight'] = (L.cout,)
            sd[f'{p}.out_layers.0.bias'] = (L.cout,)
            sd[f'{p}.out_layers.3.weight'] = (L.cout, L.cout, 3, 3)
            sd[f'{p}.out_layers.3.bias'] = (L.cout,)
            if cin != L.cout:
                sd[f'{p}.skip_connection.weight'] = (L.cout, cin, 1, 1)
                sd[f'{p}.skip_connection.bias'] = (L.cout,)
        elif L.kind == 'attn':
            c = L.cin
            sd[f'{p}.norm.weight'] = (c,)
            sd[f'{p}.norm.bias'] = (c,)
            sd[f'{p}.qkv.weight'] = (3 * c, c, 1)
            sd[f'{p}.qkv.bias'] = (3 * c,)
            sd[f'{p}.proj_out.weight'] = (c, c, 1)
            sd[f'{p}.proj_out.bias'] = (c,)
        elif L.kind == 'out':
            sd['out.0.weight'] = (L.cin,)
            sd['out.0.bias'] = (L.cin,)
            sd['out.2.weight'] = (L.cout, L.cin, 3, 3)
            sd['out.2.bias'] = (L.cout,)
    return sd
