"""Read the s_memtime stamps of a -DWINO_STAMPS build (FDSR_LIB=...libfdsr_hip_stamps.so) after one B=16 forward and print the
median phase timeline of the stamped layer: python tools/wino_stamps.py"""
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
rc = lib.fdsr_diag_wino_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert rc == 0, rc
d = buf.astype(np.int64)
d = d - d[:, :, :1]
n = int((d[0, 0] > 0).sum()) + 1
names = ['start', 'prologue']
nk = (n - 5) // 8
for k in range(nk):
    names += [f'k{k}A.p1', f'k{k}A.p2', f'k{k}A.p3', f'k{k}A.bar', f'k{k}B.p1', f'k{k}B.p2', f'k{k}B.p3', f'k{k}B.bar']
names += ['Zwritten', 'Zbar', 'end']
med = np.median(d, axis=0)     # [2][128]
print('stamps per workgroup:', n, 'chunks:', nk, '(cycles since kernel start; median over 256 workgroups; wave 0 | wave 4 ; delta wave 0 | wave 4)')
prev = med[:, 0]
for i, nm in enumerate(names[:n]):
    print(f'{nm:10s} {med[0, i]:9.0f} {med[1, i]:9.0f}   +{med[0, i] - prev[0]:7.0f} +{med[1, i] - prev[1]:7.0f}')
    prev = med[:, i]
