import ctypes as C, sys
sys.path.insert(0, '.')
import torch
from fastdiffsr_amd import _lib
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.synth import synth_state_dict
cfg = UNetConfig(**FASTDIFFSR_UNET)
e = Engine(cfg); e.load_state_dict(synth_state_dict(cfg, 0))
lib = _lib.load()
a, b, c = C.c_int(), C.c_int(), C.c_int()
print('rc', lib.fdsr_diag_wino_occupancy(C.byref(a), C.byref(b), C.byref(c)), 'blocks per CU: v1', a.value, 'v2', b.value, 'v4', c.value)
