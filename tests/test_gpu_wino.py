"""The Winograd F(2x2,3x3) form of the stride-1 3x3 convolutions (fdsr_conv_wino.hip; f16x3 mode): every module's output
layer by layer against the oracle with the form FORCED at small batch (it is on by default only for grids that fill the
chip), the 20-step loop, and the B=16 256x256 workload under its selection rule (the form is an option since round 3's 16x16x32
direct kernel overtook it: `wino = 0` is the default).  Same bounds as the direct kernels:
layerwise 1e-4 * max(1, |ref|), loop 1e-3 (north_star).  Reference: fastdiffsr_modules/unet.py:89-120."""
import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL, build_layers
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars

pytestmark = pytest.mark.gpu
TOL_FWD, TOL_LOOP = 1e-4, 1e-3


@pytest.fixture(scope='module')
def full():
    from fastdiffsr_amd.engine import Engine
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    eng = Engine(cfg)
    sd = synth_state_dict(cfg, 0)
    eng.load_state_dict(sd)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    eng.set_precision('f16x3')
    return cfg, eng, sd


@pytest.fixture(params=[2], ids=['half-tile-pipeline'])
def forced(request):
    """Winograd (the half-tile pipeline kernel; the two other forms of round 3 were removed) for every eligible launch
    regardless of the grid size; restored afterwards."""
    from fastdiffsr_amd import _lib
    _lib.debug_option('wino_min_wgs', 1)
    _lib.debug_option('wino_all', 1)
    _lib.debug_option('wino', request.param)
    yield request.param
    _lib.debug_option('wino_min_wgs', 256)
    _lib.debug_option('wino_all', 0)
    _lib.debug_option('wino', 0)      # the default: direct everywhere


@pytest.mark.timeout(900)
def test_layerwise_forced_winograd_vs_oracle(full, forced):
    """128x128, B=2: all four levels (128, 64, 32, 16 pixels) are 16-aligned, so every Block conv of the UNet takes the
    Winograd form (64-, 128-, 256-channel outputs; concat inputs 128 .. 512; the res_convs keep their own 1x1 launch)."""
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 128, 128, generator=gen)
    nl = torch.tensor([[0.02098], [0.7074]])
    cap = {}
    with torch.no_grad():
        ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
    eng.set_debug(True)
    out = eng.unet_forward(x.cuda(), nl.cuda())
    torch.cuda.synchronize()
    got = {L.name: eng.debug_tensor(L.name).cpu() for L in build_layers(cfg)}
    worst = (0.0, '')
    for L in build_layers(cfg):
        d = (got[L.name] - cap[L.name]).abs().max().item()
        scale = max(cap[L.name].abs().max().item(), 1.0)
        worst = max(worst, (d / scale, L.name))
        assert d <= TOL_FWD * scale, f'{L.name}: {d:.3e} (scale {scale:.2f})'
    assert (out.cpu() - ref).abs().max().item() <= TOL_FWD
    print(f'forced Winograd, worst layer {worst[1]} at {worst[0]:.3e}')
    # the direct kernels on the same input: same function, another kernel family => close, but not bitwise
    _lib.debug_option('wino', 0)
    out_d = eng.unet_forward(x.cuda(), nl.cuda())
    _lib.debug_option('wino', forced)
    eng.set_debug(False)
    dd = (out_d - out).abs().max().item()
    assert 0.0 < dd <= 2e-5, dd
    # rerun: bitwise (ordered reductions only)
    assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)


@pytest.mark.timeout(900)
def test_loop_forced_winograd_vs_oracle(full, forced):
    """The 20-step loop, 64x64, B=2 (levels 64, 32, 16 take the form; the 8-pixel level keeps the direct kernel)."""
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    cond, noise = synth_inputs(2, 64, 64, 20)
    ref = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
    out = eng.sample(cond.cuda(), noise.cuda()).cpu()
    d = (out - ref).abs().max().item()
    print(f'forced Winograd loop 64x64: max|d| = {d:.3e}')
    assert d <= TOL_LOOP
    # and as a hipGraph replay: same image
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g1 = eng.sample(cond.cuda(), noise.cuda(), graph=True)
        g2 = eng.sample(cond.cuda(), noise.cuda(), graph=True)
    s.synchronize()
    assert torch.equal(g1.cpu(), out) and torch.equal(g2.cpu(), out)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('everywhere', [0, 1], ids=['default-rule', 'every-eligible-layer'])
def test_b16_256_winograd_rule_matches_direct(full, everywhere):
    """BASELINE configs[1] (B=16, 256x256) with the form switched on (`wino = 2`; off by default since the 16x16x32 direct kernel
    overtook it): the layers its rule selects take it (the 32 x 32 maps), with wino_all every eligible layer.  One UNet forward
    against the direct kernels (both fp32-grade: <= 2e-5 apart, not bitwise), bitwise rerun, and image 0 of the batch against
    the oracle's B=1 forward."""
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    _lib.debug_option('wino_all', everywhere)
    _lib.debug_option('wino', 2)
    try:
        gen = torch.Generator().manual_seed(9)
        x = torch.randn(16, 6, 256, 256, generator=gen).cuda()
        nl = (torch.rand(16, 1, generator=gen) * 0.9 + 0.05).cuda()
        a = eng.unet_forward(x, nl)
        assert torch.equal(eng.unet_forward(x, nl), a)
        _lib.debug_option('wino', 0)
        b = eng.unet_forward(x, nl)
        dd = (a - b).abs().max().item()
        assert 0.0 < dd <= 2e-5, dd
    finally:
        _lib.debug_option('wino', 0)
        _lib.debug_option('wino_all', 0)
    with torch.no_grad():
        ref = O.unet_forward(O.to_torch_sd(sd), cfg, x[:1].cpu(), nl[:1].cpu())
    assert (a[:1].cpu() - ref).abs().max().item() <= TOL_FWD


def test_ragged_shapes_fall_back_to_direct(full, forced):
    """Maps that are not multiples of 16 pixels keep the direct kernel (the form has no partial tiles): still right."""
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 6, 40, 24, generator=gen)
    nl = torch.tensor([[0.3]])
    with torch.no_grad():
        ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl)
    assert (eng.unet_forward(x.cuda(), nl.cuda()).cpu() - ref).abs().max().item() <= TOL_FWD
