"""Perf probe of the conv launches of ONE UNet forward at B=16, 256x256, f16x3 (run under rocprofv3 --kernel-trace, optionally
with FDSR_LIB=<variant>): python tools/wino_probe.py [reps] [name=value debug options ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastdiffsr_amd import _lib
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.synth import synth_state_dict

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for item in sys.argv[2:]:
    k, v = item.split('=')
    _lib.debug_option(k, int(v))
B = int(os.environ.get('PROBE_B', 16))
cfg = UNetConfig(**FASTDIFFSR_UNET)
e = Engine(cfg); e.load_state_dict(synth_state_dict(cfg, 0)); e.set_precision(os.environ.get('PROBE_PREC', 'f16x3'))
e.check_saturation = False
g = torch.Generator().manual_seed(9)
x = torch.randn(B, 6, 256, 256, generator=g).cuda()
nl = (torch.rand(B, 1, generator=g) * 0.9 + 0.05).cuda()
for _ in range(reps):
    out = e.unet_forward(x, nl)
torch.cuda.synchronize()
print('probe done', float(out.abs().mean()))
