"""Like wino_stamps.py, but prints the raw median deltas of the first N stamps for waves 0 and 4 (fine-grained diagnostic builds)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastdiffsr_amd import _lib
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.synth import synth_state_dict
cfg = UNetConfig(**FASTDIFFSR_UNET)
e = Engine(cfg); e.load_state_dict(synth_state_dict(cfg, 0)); e.set_precision('f16x3'); e.check_saturation = False
g = torch.Generator().manual_seed(9)
x = torch.randn(16, 6, 256, 256, generator=g).cuda()
nl = (torch.rand(16, 1, generator=g) * 0.9 + 0.05).cuda()
for _ in range(2):
    e.unet_forward(x, nl)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((256, 2, 128), dtype=np.uint64)
assert lib.fdsr_diag_wino_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size)) == 0
d = buf.astype(np.int64)
d = d - d[:, :, :1]
med = np.median(d, axis=0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for i in range(1, N):
    print(f'{i:3d}  w0 {med[0, i]:8.0f} (+{med[0, i] - med[0, i - 1]:6.0f})   w4 {med[1, i]:8.0f} (+{med[1, i] - med[1, i - 1]:6.0f})')
