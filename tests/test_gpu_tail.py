"""The two ends of the UNet on their own kernels in the 16-bit modes (csrc/fdsr_conv_tail.hip): downs.0 (Conv3x3 6 -> inner on the
packed NHWC-8 input: gather + 16x16x32 MFMA, no LDS; reference unet.py:243) and final_conv (GroupNorm -> Swish -> Conv3x3 inner -> 3
as fp32 FMAs in scatter form; unet.py:293).  Layer by layer against the oracle on ragged maps (partial tiles in both directions), against
the general kernels on the same input (option `tail` = 0), bitwise reruns, other widths (inner 32 / 48 / 64), the loop."""
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL, build_layers
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars

pytestmark = pytest.mark.gpu
TOL_FWD, TOL_LOOP = 1e-4, 1e-3


def _engine(**over):
    from fastdiffsr_amd.engine import Engine
    kw = dict(FASTDIFFSR_UNET)
    kw.update(over)
    cfg = UNetConfig(**kw)
    eng = Engine(cfg)
    sd = synth_state_dict(cfg, 2)
    eng.load_state_dict(sd)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    return cfg, eng, sd


@pytest.mark.timeout(900)
@pytest.mark.parametrize('inner,groups', [(64, 32), (32, 32), (48, 16)])
@pytest.mark.parametrize('prec', ['f16x3', 'bf16'])
def test_tail_kernels_layerwise(prec, inner, groups):
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    if prec == 'bf16' and inner == 48:
        pytest.skip('bf16 mode needs 16-aligned channel counts everywhere (48 * 2 = 96 ok, but 48 / 16 groups of 3 are not MFMA-aligned)')
    cfg, eng, sd = _engine(inner_channel=inner, norm_groups=groups)
    eng.set_precision(prec)
    tol = TOL_FWD if prec == 'f16x3' else 0.04
    try:
        for shape, seed in (((2, 6, 72, 104), 31), ((1, 6, 40, 24), 32)):
            gen = torch.Generator().manual_seed(seed)
            x = torch.randn(*shape, generator=gen) * 1.7
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
                assert d <= tol * scale, f'{shape} {L.name}: {d:.3e} (scale {scale:.2f})'
            eng.set_debug(False)
            assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)              # ordered reductions only
            if prec == 'f16x3':
                assert (out.cpu() - ref).abs().max().item() <= TOL_FWD
            _lib.debug_option('tail', 0)                                                # the general kernels on the same input
            out_d = eng.unet_forward(x.cuda(), nl.cuda())
            _lib.debug_option('tail', 1)
            dd = (out_d - out).abs().max().item()
            assert dd > 0.0                                                             # (0.0: the kernels were never taken)
            if prec == 'f16x3':
                assert dd <= 2e-5, dd
    finally:
        eng.set_debug(False)
        _lib.debug_option('tail', 1)


def test_tail_kernels_in_the_loop_and_graph():
    from oracle import fdsr_oracle as O
    cfg, eng, sd = _engine()
    eng.set_precision('f16x3')
    cond, noise = synth_inputs(2, 64, 64, 20)
    ref = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
    out = eng.sample(cond.cuda(), noise.cuda()).cpu()
    assert (out - ref).abs().max().item() <= TOL_LOOP
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g1 = eng.sample(cond.cuda(), noise.cuda(), graph=True)
    s.synchronize()
    assert torch.equal(g1.cpu(), out)
