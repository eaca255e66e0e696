"""The column-strip form of the stride-1 3x3 launches at 64 (and, in bf16, 128) output channels (fdsr_conv_strip.hip: weights in
registers, one input row per step, three output rows per fragment read).  Large grids take it by default (`strip` bits: 1 bf16
64 -> 64, 2 f16x3 64 -> 64, 8 bf16 (64|64) -> 64, 16 bf16 with a res_conv rider, 32 bf16 (128|64) -> 64 [off by default: spills], 64 bf16
128 -> 128 and 64 -> 128 as two workgroups of 64 couts; `strip_min_wgs`); here every bit is on and the form is forced onto every grid
size: layer by layer against the oracle (maps 128 x 128 and 64 x 192: one and three strips per row, segments
of 16 .. 128 rows, image borders on both sides of a strip), against the tile kernels on the same input, bitwise reruns, the 20-step
loop eager and as a hipGraph.  Same bounds as every other conv kernel: layerwise 1e-4 * max(1, |ref|) in f16x3, 0.04 in bf16 (measured worst 0.011) (judged
on PSNR elsewhere), loop 1e-3 (north_star).  Reference: fastdiffsr_modules/unet.py:89-120."""
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL, build_layers
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs
from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars

pytestmark = pytest.mark.gpu
TOL_FWD, TOL_LOOP = 1e-4, 1e-3
PRECS = ['bf16', 'f16x3']


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
@pytest.mark.parametrize('prec', PRECS)
@pytest.mark.parametrize('min_wgs', [1, 4, 12, 9, 18], ids=['whole-strips', 'segments-a', 'segments-b', 'segments-9row', 'segments-9row-wide'])
def test_strip_form_vs_oracle(full, prec, min_wgs):
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    eng.set_precision(prec)
    tol = TOL_FWD if prec == 'f16x3' else 0.04
    _lib.debug_option('strip', 127 - 4)
    _lib.debug_option('strip_min_wgs', min_wgs)
    _lib.debug_option('splitk', 0)          # (a launch with a K split keeps the tile kernels)
    try:
        # 136 rows x 64 columns: 9-row segments leave a remainder of ONE row (min_wgs 9 on the two-per-CU forms, 18 on the wide ones) -- the kernel
        # balances its segments, so no segment is shorter than its peeled steps
        for shape, seed in (((2, 6, 128, 128), 31), ((1, 6, 64, 192), 32), ((1, 6, 136, 64), 33)):
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
            _lib.debug_option('strip', 0)                                               # the same launches on the tile kernels
            out_d = eng.unet_forward(x.cuda(), nl.cuda())
            _lib.debug_option('strip', 127 - 4)
            dd = (out_d - out).abs().max().item()
            assert dd > 0.0                                                             # (0.0: the form was never taken)
            if prec == 'f16x3':
                assert dd <= 2e-5 and (out.cpu() - ref).abs().max().item() <= TOL_FWD
            else:
                rm = (out_d - out).pow(2).mean().sqrt().item()
                assert rm <= 2e-2 and (out.cpu() - ref).pow(2).mean().sqrt().item() <= 2e-2, rm
    finally:
        eng.set_debug(False)
        eng.set_precision('f16x3')
        _lib.debug_option('strip', 91)
        _lib.debug_option('strip_min_wgs', 512)
        _lib.debug_option('splitk', 1)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('prec', PRECS)
def test_strip_form_loop_and_graph(full, prec):
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    eng.set_precision(prec)
    _lib.debug_option('strip', 127 - 4)
    _lib.debug_option('strip_min_wgs', 1)
    _lib.debug_option('splitk', 0)
    try:
        cond, noise = synth_inputs(2, 64, 64, 20)
        refl = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
        outl = eng.sample(cond.cuda(), noise.cuda()).cpu()
        d = (outl - refl).abs().max().item()
        print(f'strip loop 64x64 {prec}: max|d| = {d:.3e}')
        if prec == 'f16x3':
            assert d <= TOL_LOOP
        else:
            assert (outl - refl).pow(2).mean().sqrt().item() <= 2e-2
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            g1 = eng.sample(cond.cuda(), noise.cuda(), graph=True)
        s.synchronize()
        assert torch.equal(g1.cpu(), outl)
    finally:
        eng.set_precision('f16x3')
        _lib.debug_option('strip', 91)
        _lib.debug_option('strip_min_wgs', 512)
        _lib.debug_option('splitk', 1)
