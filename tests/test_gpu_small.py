"""The small-map form of the 256-cout stride-1 3x3 launches (fdsr_conv_small.hip; f16x3): at small batches -- the reference's own val
loop runs B = 1, sr_mfe.py:274-284 -- the 32 x 32 and 64 x 64 levels give the tile kernels a handful of workgroups; instead of splitting
K over workgroups and adding the slices in a second launch, the K split stays inside a workgroup (wave k = 32-channel slice k, the
eight partial tiles added in LDS in wave order) and the epilogue runs in the same launch.  Layer by layer against the oracle (B = 1 at
256 x 256: 256 -> 256 and (256 | 256) -> 256 at 32 x 32 and 64 x 64, with and without a residual; B = 2 at 128 x 128; a ragged 64 x 96 map),
against the split-K path on the same input (another summation order: close, not bitwise), bitwise reruns, the 20-step loop eager and
as a hipGraph.  Bounds: layerwise 1e-4 * max(1, |ref|), loop 1e-3 (north_star).  Reference: fastdiffsr_modules/unet.py:89-120."""
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


@pytest.mark.timeout(900)
def test_small_map_form_vs_oracle(full):
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    _lib.debug_option('small', 1)
    _lib.debug_option('small_max_wgs', 1 << 20)
    try:
        for shape, seed in (((1, 6, 256, 256), 51), ((2, 6, 128, 128), 52), ((1, 6, 256, 384), 53)):
            gen = torch.Generator().manual_seed(seed)
            x = torch.randn(*shape, generator=gen)
            nl = torch.rand(shape[0], 1, generator=gen) * 0.9 + 0.05
            cap = {}
            with torch.no_grad():
                ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
            eng.set_debug(True)
            out = eng.unet_forward(x.cuda(), nl.cuda())
            torch.cuda.synchronize()
            for L in build_layers(cfg):
                d = (eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item()
                scale = max(cap[L.name].abs().max().item(), 1.0)
                assert d <= TOL_FWD * scale, f'{shape} {L.name}: {d:.3e} (scale {scale:.2f})'
            assert (out.cpu() - ref).abs().max().item() <= TOL_FWD
            eng.set_debug(False)
            assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)              # ordered reductions only
            _lib.debug_option('small', 0)                                               # the same launches split over workgroups + reduce
            out_d = eng.unet_forward(x.cuda(), nl.cuda())
            _lib.debug_option('small', 1)
            dd = (out_d - out).abs().max().item()
            assert 0.0 < dd <= 2e-5, dd                                                 # (0.0: the form was never taken)
    finally:
        eng.set_debug(False)
        _lib.debug_option('small', 1)
        _lib.debug_option('small_max_wgs', 1024)


@pytest.mark.timeout(900)
def test_small_map_form_loop_and_graph(full):
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    cond, noise = synth_inputs(1, 128, 128, 20)
    refl = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
    outl = eng.sample(cond.cuda(), noise.cuda()).cpu()
    assert (outl - refl).abs().max().item() <= TOL_LOOP
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g1 = eng.sample(cond.cuda(), noise.cuda(), graph=True)
    s.synchronize()
    assert torch.equal(g1.cpu(), outl)
