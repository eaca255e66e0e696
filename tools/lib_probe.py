"""Perf-only probe for A/B variant libraries (FDSR_LIB=...): the f16x3 sampling loop at B=16 on the synthetic weights,
no parity check (variants may compute garbage on purpose); prints img/s and whether the output stayed finite / dense."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs

from fastdiffsr_amd import _lib
for item in os.environ.get('PROBE_OPTS', '').split(','):      # launcher options (fdsr_debug_option), e.g. PROBE_OPTS=rider=0,strip=0
    if item:
        _lib.debug_option(item.split('=')[0], int(item.split('=')[1]))
prec = sys.argv[1] if len(sys.argv) > 1 else 'f16x3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = UNetConfig(**FASTDIFFSR_UNET)
bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
e = Engine(cfg); e.load_state_dict(synth_state_dict(cfg, 0)); e.set_schedule(sampling_scalars(bufs, sp)); e.set_precision(prec)
cond, noise = synth_inputs(B, 256, 256, 20)
c, n = cond.cuda(), noise.cuda()
out = torch.empty(B, 3, 256, 256, device='cuda')
e.sample(c, n, out=out); torch.cuda.synchronize()
for rep in range(int(os.environ.get("REPS", 6))):
    t0 = time.perf_counter()
    for _ in range(4):
        e.sample(c, n, out=out)
    torch.cuda.synchronize()
    print(f'{os.path.basename(os.environ.get("FDSR_LIB", "tree"))} {prec} B={B}: {B * 4 / (time.perf_counter() - t0):.2f} img/s  '
          f'finite={bool(torch.isfinite(out).all())} std={float(out.std()):.3f} zero_frac={float((out == 0).float().mean()):.3f}', flush=True)
