"""Read the s_memtime stamps of a -DCONVH_STAMPS build (FDSR_LIB=...) after forwards at the given precision / batch and print the
median phase timeline of the stamped layer shape: python tools/convh_stamps.py <prec> <batch>"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fastdiffsr_amd import _lib
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.synth import synth_state_dict

prec, B = sys.argv[1], int(sys.argv[2])
cfg = UNetConfig(**FASTDIFFSR_UNET)
e = Engine(cfg); e.load_state_dict(synth_state_dict(cfg, 0)); e.set_precision(prec); e.check_saturation = False
g = torch.Generator().manual_seed(9)
x = torch.randn(B, 6, 256, 256, generator=g).cuda()
nl = (torch.rand(B, 1, generator=g) * 0.9 + 0.05).cuda()
for _ in range(2):
    e.unet_forward(x, nl)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((256, 2, 64), dtype=np.uint64)
rc = lib.fdsr_diag_convh_stamps(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size))
assert rc == 0, rc
d = buf.astype(np.int64)
ok = d[:, 0, 0] > 0
d = d[ok]
d = d - d[:, :, :1]
n = int((d[0, 0] > 0).sum()) + 1
nk = (n - 7) // 4 + 1            # the last chunk has no mid-chunk pair
names = ['start', 'loads issued', 'chunk0 staged', 'prologue bar']
for k in range(nk):
    if k < nk - 1:
        names += [f'k{k} mid', f'k{k} staged']
    names += [f'k{k} taps done', f'k{k} barrier']
names += ['epilogue', 'stores issued', 'end']
med = np.median(d, axis=0)
print(f'{prec} B={B}: workgroups stamped {ok.sum()}, stamps {n}, chunks {nk} (shader cycles since the workgroup's start, s_memtime; median over the workgroups; wave 0 | wave 7, then the step from the previous stamp)')
prev = med[:, 0]
for i, nm in enumerate(names[:n]):
    print(f'{nm:15s} {med[0, i]:9.0f} {med[1, i]:9.0f}   +{med[0, i] - prev[0]:7.0f} +{med[1, i] - prev[1]:7.0f}')
    prev = med[:, i]
