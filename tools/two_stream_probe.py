"""Probe: one engine sampling B images on one stream vs K engines sampling B/K each on K streams
(kernel tails / prologue bursts of one stream overlapping another's main loops)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs

cfg = UNetConfig(**FASTDIFFSR_UNET)
sd = synth_state_dict(cfg, 0)
bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prec = sys.argv[2] if len(sys.argv) > 2 else 'f16x3'


def make(n):
    e = Engine(cfg); e.load_state_dict(sd); e.set_schedule(sampling_scalars(bufs, sp)); e.set_precision(prec); e.set_seed(n)
    return e


def run(K, reps=4):
    engs = [make(i) for i in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    conds = [synth_inputs(B // K, 256, 256, 1, cond_seed=10 + i)[0].cuda() for i in range(K)]
    outs = [torch.empty(B // K, 3, 256, 256, device='cuda') for _ in range(K)]

    def once():
        for i in range(K):
            with torch.cuda.stream(streams[i]):
                engs[i].sample(conds[i], None, out=outs[i])
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * reps / dt


for K in (1, 2, 4, 1, 2):
    print(f'B={B} {prec}: {K} stream(s) x {B // K} images: {run(K):.2f} img/s', flush=True)
