"""Engine-side noise (fdsr_sample with noise == NULL): the generator's statistics, and that the
in-loop draws are exactly the planes fdsr_randn reports (so the RNG path and the explicit-noise
path, which carries the parity proof, are the same computation)."""
import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs

pytestmark = pytest.mark.gpu

SMALL = dict(in_channel=6, out_channel=3, inner_channel=32, norm_groups=16, channel_mults=(1, 2, 2), res_blocks=1,
             dropout=0.0, image_size=32)


@pytest.fixture(scope='module')
def eng():
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    cfg = UNetConfig(**SMALL)
    e = Engine(cfg)
    e.load_state_dict(synth_state_dict(cfg, 3))
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    e.set_schedule(sampling_scalars(bufs, sp))
    return e


def test_generator_statistics(eng):
    from scipy import stats
    eng.set_seed(1234)
    z0 = eng.randn(4, 256, 256, 0).double().cpu().numpy()
    z1 = eng.randn(4, 256, 256, 1).double().cpu().numpy()
    n = z0.size                                                   # 786432
    assert abs(z0.mean()) < 5 / np.sqrt(n)
    assert abs(z0.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs(stats.skew(z0.ravel())) < 5 * np.sqrt(6 / n)
    assert abs(stats.kurtosis(z0.ravel())) < 5 * np.sqrt(24 / n)
    assert abs((np.abs(z0) > 3).mean() - 0.0026998) < 5 * np.sqrt(0.0027 / n)
    bound = 5 / np.sqrt(n)
    assert abs(np.mean(z0 * z1)) < bound                          # planes (steps)
    assert abs(np.mean(z0[:, 0] * z0[:, 1])) < 5 / np.sqrt(n / 3)  # channels of one pixel (one Philox block)
    assert abs(np.mean(z0[..., :-1] * z0[..., 1:])) < bound        # neighbouring pixels (adjacent counters)
    assert abs(np.mean(z0[0] * z0[1])) < 5 / np.sqrt(n / 4)        # images
    sub = z0.ravel()[::7][:100000]
    assert stats.kstest(sub, 'norm').pvalue > 1e-4
    # keyed by (seed, call counter, plane, pixel) only: same values for another batch split
    a = eng.randn(4, 64, 64, 2)
    b = eng.randn(2, 64, 64, 2)
    assert torch.equal(a[:2], b)
    eng.set_seed(1235)
    assert not torch.equal(eng.randn(4, 64, 64, 2), a)


def test_in_loop_draws_equal_reported_planes(eng):
    cond, _ = synth_inputs(2, 64, 64, 20)
    cond = cond.cuda()
    eng.set_seed(99)
    out_a = eng.sample(cond).clone()                               # noise drawn inside the loop (call counter 1)
    planes = torch.stack([eng.randn(2, 64, 64, k) for k in range(20)])
    out_b = eng.sample(cond, planes).clone()                       # explicit-noise path on the very same planes
    assert torch.equal(out_a, out_b)
    out_c = eng.sample(cond).clone()                               # next call: new counter, new noise
    assert not torch.equal(out_a, out_c)
    eng.set_seed(99)                                               # reseeding resets the counter
    assert torch.equal(eng.sample(cond), out_a)


def test_graph_replay_draws_fresh_noise(eng):
    cond, _ = synth_inputs(2, 64, 64, 20)
    cond = cond.cuda()
    out = torch.empty(2, 3, 64, 64, device='cuda')
    eng.set_seed(7)
    a = eng.sample(cond, graph=True, out=out).clone()              # capture + launch
    b = eng.sample(cond, graph=True, out=out).clone()              # replay: the counter lives on the device
    assert not torch.equal(a, b)
    eng.set_seed(7)
    a2 = eng.sample(cond, out=out).clone()                         # eager, same seed: same first image
    assert torch.equal(a, a2)
    assert torch.isfinite(b).all()
