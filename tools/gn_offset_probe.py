import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from fastdiffsr_amd.arch import UNetConfig, build_layers
from fastdiffsr_amd.synth import synth_state_dict
from fastdiffsr_amd.engine import Engine
from oracle import fdsr_oracle as O
cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 4, 4), attn_res=(16,), res_blocks=2, dropout=0.2, image_size=64)
for off in (0.0, 20.0, 100.0, 400.0):
    sd = synth_state_dict(cfg, 0)
    sd['downs.0.bias'] = sd['downs.0.bias'] + np.float32(off)          # DC offset rides the residual stream
    for k in list(sd):
        if k.endswith('block2.block.3.bias'):
            sd[k] = sd[k] + np.float32(off * 0.25)
    x = torch.randn(2, 6, 64, 64, generator=torch.Generator().manual_seed(1)); nl = torch.tensor([[0.3], [0.8]])
    cap = {}
    with torch.no_grad():
        ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
    for prec in ('f32', 'f16x3'):
        eng = Engine(cfg); eng.load_state_dict(sd); eng.set_precision(prec); eng.set_debug(True)
        out = eng.unet_forward(x.cuda(), nl.cuda()).cpu()
        worst = max(((eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item() / max(1.0, cap[L.name].std().item()), L.name) for L in build_layers(cfg))
        print(f'offset {off:6.1f} {prec:6s}: final max|d| {(out-ref).abs().max().item():.3e}  worst layer (|d|/std) {worst[0]:.3e} at {worst[1]}', flush=True)
