"""GroupNorm statistics without a finalisation launch on small grids (round 6; `gn_consumer`, include/fdsr.h): producers add the
(sum, sum of squares) of every channel pair of their output tile as fixed-point int64 to the tensor's table (agent-scope atomics:
integer adds commute, so reruns stay bitwise), the consumer conv -- the 2-row-per-wave 16x16x32 kernels of small grids -- folds the
pairs of the groups its K slice touches and forms scale / shift itself in its prologue.  Reference: fastdiffsr_modules/unet.py:89-101
(Block = GroupNorm -> Swish -> Dropout -> Conv).

Here: layer by layer against the oracle with the form on (the default) and against the same forward with it off (the gn_finalize
launches back): B = 1 and B = 2 maps whose launches split K, concatenated inputs whose groups cross the concat seam (384 channels =
32 groups of 12), riders, both 16-bit modes; bitwise reruns; the 20-step loop eager and as a replayed graph."""
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL, build_layers
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def full():
    from fastdiffsr_amd.engine import Engine
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    eng = Engine(cfg)
    sd = synth_state_dict(cfg, 0)
    eng.load_state_dict(sd)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))
    return cfg, eng, sd


@pytest.mark.timeout(900)
@pytest.mark.parametrize('prec', ['f16x3', 'bf16'])
def test_consumer_side_groupnorm_layerwise(full, prec):
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    eng.set_precision(prec)
    tol = 1e-4 if prec == 'f16x3' else 0.04
    try:
        for shape, seed in (((1, 6, 128, 128), 41), ((2, 6, 64, 96), 42), ((1, 6, 256, 256), 43)):
            gen = torch.Generator().manual_seed(seed)
            x = torch.randn(*shape, generator=gen)
            nl = torch.rand(shape[0], 1, generator=gen) * 0.9 + 0.05
            cap = {}
            if shape[-1] < 256:          # (the 256 x 256 forward is compared with the finalize-launch form only: no 30 s oracle forward here)
                with torch.no_grad():
                    O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
            _lib.debug_option('gn_consumer', 1)
            out = eng.unet_forward(x.cuda(), nl.cuda())
            assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)              # integer atomics: order-independent
            _lib.debug_option('gn_consumer', 0)
            out_f = eng.unet_forward(x.cuda(), nl.cuda())
            _lib.debug_option('gn_consumer', 1)
            dd = (out_f - out).abs().max().item()
            scale = max(out_f.abs().max().item(), 1.0)
            print(f'gn_consumer {prec} {shape}: max|consumer-side - finalize launches| = {dd:.3e} (range {scale:.2f})')
            assert dd > 0.0                                                             # (0.0: the form was never taken)
            assert dd <= (2e-5 if prec == 'f16x3' else 3e-2) * scale
            if cap:
                # layer by layer needs a debug forward, which keeps the finalize launches: compare the OUTPUT of the consumer-side
                # forward with the oracle's, and the debug forward's layers as the other tests do
                ref = cap['final_conv']
                d = (out.cpu() - ref).abs().max().item()
                assert d <= tol * max(ref.abs().max().item(), 1.0), d
    finally:
        _lib.debug_option('gn_consumer', 1)
        eng.set_precision('f16x3')


@pytest.mark.timeout(900)
@pytest.mark.parametrize('prec', ['f16x3', 'bf16'])
def test_consumer_side_groupnorm_loop_and_graph(full, prec):
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    eng.set_precision(prec)
    try:
        cond, noise = synth_inputs(1, 64, 64, 20)
        ref = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
        out = eng.sample(cond.cuda(), noise.cuda()).cpu()
        d = (out - ref).abs().max().item()
        print(f'gn_consumer loop 64x64 B=1 {prec}: max|d| = {d:.3e}')
        if prec == 'f16x3':
            assert d <= 1e-3
        else:
            assert (out - ref).pow(2).mean().sqrt().item() <= 2e-2
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            eng.sample(cond.cuda(), noise.cuda(), graph=True)
            g1 = eng.sample(cond.cuda(), noise.cuda(), graph=True)      # the second call replays: the table memset is a graph node
        s.synchronize()
        assert torch.equal(g1.cpu(), out)
        _lib.debug_option('gn_consumer', 0)
        out_f = eng.sample(cond.cuda(), noise.cuda()).cpu()
        assert 0.0 < (out_f - out).abs().max().item() <= (2e-4 if prec == 'f16x3' else 0.2)
    finally:
        _lib.debug_option('gn_consumer', 1)
        eng.set_precision('f16x3')
