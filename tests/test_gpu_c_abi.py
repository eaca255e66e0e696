"""The C ABI driven from plain C (examples/fdsr_demo.c, built with gcc, no Python or torch in the
process): same images, bit for bit, as the Python facade gives for the same bundle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL  # noqa: E402
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs  # noqa: E402

pytestmark = pytest.mark.gpu
DEMO = os.path.join(ROOT, 'examples', 'fdsr_demo')


def _engine(cfg, sd):
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    return eng


@pytest.mark.parametrize('precision,graph,with_noise', [(1, 0, True), (0, 1, True), (1, 1, False), (2, 1, True)])
def test_c_host_matches_python_facade(tmp_path, precision, graph, with_noise):
    from export_bundle import write_bundle
    from fastdiffsr_amd import build as b
    demo = b.build_demo(force=False, verbose=False) if not os.path.exists(DEMO) else DEMO
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd = synth_state_dict(cfg, 0)
    B, S = 2, 64
    cond, noise = synth_inputs(B, S, S, 20)
    write_bundle(str(tmp_path), cfg, sd, FASTDIFFSR_SCHEDULE_VAL, cond.numpy(), noise.numpy() if with_noise else None, seed=77)
    r = subprocess.run([demo, str(tmp_path), str(precision), str(graph)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'gfx950' in r.stdout
    got = np.fromfile(os.path.join(str(tmp_path), 'out.bin'), dtype=np.float32).reshape(B, 3, S, S)
    eng = _engine(cfg, sd)
    eng.set_precision({0: 'f32', 1: 'f16x3', 2: 'bf16'}[precision])
    if with_noise:
        want = eng.sample(cond.cuda(), noise.cuda())
    else:
        eng.set_seed(77)
        want = eng.sample(cond.cuda())
    assert np.array_equal(got, want.cpu().numpy())
