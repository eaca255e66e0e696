"""Where the B = 1 val loop's host time goes: cProfile of fastdiffsr_amd.val.run over a synthetic folder (batch 1, graph replay)."""
import cProfile, os, pstats, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from fastdiffsr_amd import val as V
from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET
from fastdiffsr_amd.model import create_model
from fastdiffsr_amd.synth import synth_state_dict

cfg = UNetConfig(**FASTDIFFSR_UNET)
sd = synth_state_dict(cfg, 0)
root = tempfile.mkdtemp(prefix='fdsr_b1p_')
try:
    bench.synth_folder(root, 48)
    opt = bench.facade_opt(root, 'val')
    model = create_model(opt)
    model.netG.denoise_fn.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    V.run(opt, batch=1, results=os.path.join(root, 'w'), max_images=6, log=lambda m: None, diffusion=model, rng='engine')
    pr = cProfile.Profile()
    pr.enable()
    r = V.run(opt, batch=1, results=os.path.join(root, 'o'), log=lambda m: None, diffusion=model, rng='engine')
    pr.disable()
    print('sampling %.2f ms / image' % (1e3 * r['sample_seconds_this_rank'] / r['images']))
    pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
finally:
    shutil.rmtree(root, ignore_errors=True)
