"""Split-K launches whose output feeds a GroupNorm: ONE reduce launch that also finalises that GroupNorm (splitk_reduce_gn_kernel,
option `fuse_gn`, on by default) instead of splitk_reduce + gn_finalize.  Small batches (the reference's own val loop is B = 1,
sr_mfe.py:274-284) split K on every level but the first: layer by layer against the oracle (every GroupNorm'ed tensor after a fused
launch checks the statistics), against the two-launch path on the same input (another summation order: close, not bitwise), bitwise
reruns, the 20-step loop eager and as a hipGraph.  Bounds as everywhere: layerwise 1e-4 * max(1, |ref|) in f16x3, 0.25 in bf16, loop
1e-3.  Reference: fastdiffsr_modules/unet.py:89-120 (Block: GroupNorm -> Swish -> Conv)."""
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
@pytest.mark.parametrize('prec', ['f16x3', 'bf16'])
def test_fused_reduce_and_groupnorm_vs_oracle(full, prec):
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    eng.set_precision(prec)
    tol = TOL_FWD if prec == 'f16x3' else 0.25
    _lib.debug_option('fuse_gn', 1)
    try:
        for shape, seed in (((1, 6, 128, 128), 41), ((2, 6, 64, 96), 42), ((1, 6, 256, 256), 43)):
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
                assert d <= tol * scale, f'{shape} {L.name}: {d:.3e} (scale {scale:.2f})'
            eng.set_debug(False)
            assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)              # ordered reductions only
            _lib.debug_option('fuse_gn', 0)                                             # reduce, then finalize: two launches
            out_d = eng.unet_forward(x.cuda(), nl.cuda())
            _lib.debug_option('fuse_gn', 1)
            dd = (out_d - out).abs().max().item()
            assert dd > 0.0                                                             # (0.0: the fused launch was never taken)
            if prec == 'f16x3':
                assert dd <= 2e-5 and (out.cpu() - ref).abs().max().item() <= TOL_FWD
            else:
                assert (out_d - out).pow(2).mean().sqrt().item() <= 2e-2
    finally:
        eng.set_debug(False)
        eng.set_precision('f16x3')
        _lib.debug_option('fuse_gn', 1)


@pytest.mark.timeout(900)
def test_fused_reduce_loop_and_graph(full):
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    cond, noise = synth_inputs(1, 64, 64, 20)
    refl = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
    outl = eng.sample(cond.cuda(), noise.cuda()).cpu()
    assert (outl - refl).abs().max().item() <= TOL_LOOP
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g1 = eng.sample(cond.cuda(), noise.cuda(), graph=True)
    s.synchronize()
    assert torch.equal(g1.cpu(), outl)
