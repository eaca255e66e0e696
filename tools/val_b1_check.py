"""Round-3 verdict item 7: `python -m fastdiffsr_amd.val --batch 1` should spend per image what the engine's B=1 hipGraph replay spends
(bench.py's b1_graph record).  Runs both on the same box: the val loop over a synthetic folder (batch 1; default rng 'torch', then
'engine') and the bare engine loop, and prints the per-image times."""
import os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from fastdiffsr_amd import val as V
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.engine import Engine
from fastdiffsr_amd.model import create_model
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
from fastdiffsr_amd.synth import synth_state_dict

cfg = UNetConfig(**FASTDIFFSR_UNET)
sd = synth_state_dict(cfg, 0)
dev = torch.device('cuda', 0)
eng = Engine(cfg); eng.load_state_dict(sd)
bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL); eng.set_schedule(sampling_scalars(bufs, sp))
dt, _, _, _ = bench.run_config(eng, dev, 'f16x3', 1, 256, 40, 5, True, 'engine', want_profile=False)
print('engine b1 graph: %.2f ms / image' % (1e3 * dt / 40))
# the same replays with the GPU left idle for a few milliseconds in between, as a val loop leaves it (host post-processing of the image):
# does the clock the chip holds depend on the gap?
from fastdiffsr_amd.synth import synth_inputs
cond, _ = synth_inputs(1, 256, 256, 1)
cond = cond.to(dev); out = torch.empty(1, 3, 256, 256, device=dev)
eng.set_precision('f16x3'); eng.set_seed(1)
for gap in (0.0, 0.002, 0.005):
    ts = []
    for i in range(25):
        torch.cuda.synchronize(); time.sleep(gap)
        t0 = time.perf_counter(); eng.sample(cond, None, out=out, graph=True); torch.cuda.synchronize()
        if i >= 5: ts.append(time.perf_counter() - t0)
    print('engine b1 graph, %.0f ms idle before every replay: %.2f ms / image' % (1e3 * gap, 1e3 * sum(ts) / len(ts)))
root = tempfile.mkdtemp(prefix='fdsr_b1_')
try:
    bench.synth_folder(root, 48)
    opt = bench.facade_opt(root, 'val')
    model = create_model(opt)
    model.netG.denoise_fn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    for rng in ('torch', 'engine'):
        V.run(opt, batch=1, results=os.path.join(root, 'w'), max_images=6, log=lambda m: None, diffusion=model, rng=rng)
        t0 = time.perf_counter()
        r = V.run(opt, batch=1, results=os.path.join(root, 'o'), log=lambda m: None, diffusion=model, rng=rng)
        wall = time.perf_counter() - t0
        print("val --batch 1 --rng %s: sampling %.2f ms / image, wall %.2f ms / image (%d images) host %s" % (
            rng, 1e3 * r['sample_seconds_this_rank'] / r['images'], 1e3 * wall / r['images'], r['images'], r['host_seconds']))
finally:
    shutil.rmtree(root, ignore_errors=True)
