"""Probe: the sampling loop on all-zero weights and inputs vs the synthetic ones.  Same instruction stream, same
cycles; a throughput difference is the clock the chip holds under load (MI355X_MICROARCH.md, DVFS give-back)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs

cfg = UNetConfig(**FASTDIFFSR_UNET)
bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
B = 16
for prec in ('f16x3', 'bf16'):
    for zero in (False, True, False, True):
        sd = synth_state_dict(cfg, 0)
        if zero:
            sd = {k: np.zeros_like(v) for k, v in sd.items()}
        e = Engine(cfg); e.load_state_dict(sd); e.set_schedule(sampling_scalars(bufs, sp)); e.set_precision(prec)
        cond, noise = synth_inputs(B, 256, 256, 20)
        if zero:
            cond, noise = torch.zeros_like(cond), torch.zeros_like(noise)
        c, n = cond.cuda(), noise.cuda()
        out = torch.empty(B, 3, 256, 256, device='cuda')
        e.sample(c, n, out=out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            e.sample(c, n, out=out)
        torch.cuda.synchronize()
        print(f'{prec} {"zeros " if zero else "random"}: {B * 4 / (time.perf_counter() - t0):.2f} img/s', flush=True)
        del e
