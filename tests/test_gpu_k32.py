"""The 16x16x32-MFMA form of the stride-1 3x3 convolutions (fdsr_conv_k32.hip): on by default for f16x3 launches whose wave tile is
4 x 32 or 2 x 32 pixels (except the 16-row tile with a rider); here it is FORCED onto every eligible launch of a small forward (largest tile regardless of the grid size, so
the split-K, partial-tile and rider paths of the form all run), layer by layer against the oracle, in every option setting
(bits of `k32`: 1 f16x3, 2 bf16, 4 the 16-row tile with a rider, 8 the 2-row tiles of small grids, 16 the sub-pixel upsample convs, 32 / 128 the small-workgroup form of the 64-cout tiles in f16x3 / bf16, 64 / 512 with a rider (rider chunks first), 1024 the 8-wave rider kernels rider-first too; default 1275), against the 32x32x16 kernels on the same input, and through the
20-step loop.  Same bounds as every other conv kernel: layerwise 1e-4 * max(1, |ref|), loop 1e-3 (north_star); bf16 0.04 layerwise
(judged on PSNR elsewhere).  Reference: fastdiffsr_modules/unet.py:89-120."""
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


@pytest.fixture
def forced():
    """Every stride-1 3x3 launch on its largest tile (4 x 32 pixels per wave: the form's shape) whatever the grid size -- small
    grids then split K --; restored afterwards."""
    from fastdiffsr_amd import _lib
    _lib.debug_option('th_min_wgs', 1)
    yield
    _lib.debug_option('th_min_wgs', 256)
    _lib.debug_option('k32', 1275)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('k32', [27, 5], ids=['default', 'rider-on-16-row-tiles'])
def test_layerwise_forced_k32_vs_oracle(full, forced, k32):
    """128x128, B=2: 64-, 128- and 256-channel outputs, concat inputs 128 .. 512 with seams on 32-channel boundaries, riders,
    16-pixel maps under 32-pixel tiles (partial tiles), grids below 256 workgroups (split K)."""
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    _lib.debug_option('k32', k32)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 128, 128, generator=gen)
    nl = torch.tensor([[0.02098], [0.7074]])
    cap = {}
    with torch.no_grad():
        ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
    eng.set_debug(True)
    try:
        out = eng.unet_forward(x.cuda(), nl.cuda())
        torch.cuda.synchronize()
        worst = (0.0, '')
        for L in build_layers(cfg):
            got = eng.debug_tensor(L.name).cpu()
            d = (got - cap[L.name]).abs().max().item()
            scale = max(cap[L.name].abs().max().item(), 1.0)
            worst = max(worst, (d / scale, L.name))
            assert d <= TOL_FWD * scale, f'{L.name}: {d:.3e} (scale {scale:.2f})'
        assert (out.cpu() - ref).abs().max().item() <= TOL_FWD
        print(f'forced k32={k32}, worst layer {worst[1]} at {worst[0]:.3e}')
        # the 32x32x16 kernels on the same input: same arithmetic, another summation order => close, not bitwise
        _lib.debug_option('k32', 0)
        out_d = eng.unet_forward(x.cuda(), nl.cuda())
        _lib.debug_option('k32', k32)
        dd = (out_d - out).abs().max().item()
        assert 0.0 < dd <= 2e-5, dd     # (0.0 would mean the form was never taken)
        # rerun: bitwise (ordered reductions only)
        assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)
    finally:
        eng.set_debug(False)


@pytest.mark.timeout(900)
def test_small_grid_two_row_tiles_k32_vs_oracle(full):
    """Bit 8 of the option: the 2-row-per-wave tiles that small grids pick (B = 2 at 128 x 128 with the default tile rule: 8 x 2-,
    4 x 4- and 2 x 8-shaped workgroups, most of them with split K) on the form too, riders included (bit 4)."""
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    _lib.debug_option('k32', 13)
    try:
        gen = torch.Generator().manual_seed(11)
        x = torch.randn(2, 6, 128, 128, generator=gen)
        nl = torch.tensor([[0.3], [0.9]])
        cap = {}
        with torch.no_grad():
            ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
        eng.set_debug(True)
        out = eng.unet_forward(x.cuda(), nl.cuda())
        torch.cuda.synchronize()
        for L in build_layers(cfg):
            d = (eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item()
            scale = max(cap[L.name].abs().max().item(), 1.0)
            assert d <= TOL_FWD * scale, f'{L.name}: {d:.3e} (scale {scale:.2f})'
        assert (out.cpu() - ref).abs().max().item() <= TOL_FWD
        eng.set_debug(False)
        assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)
        _lib.debug_option('k32', 0)      # the 32x32x16 kernels on the same input
        out_d = eng.unet_forward(x.cuda(), nl.cuda())
        dd = (out_d - out).abs().max().item()
        assert 0.0 < dd <= 2e-5, dd
        # B = 1, 64 x 64 through the loop (every level on 2-row tiles)
        _lib.debug_option('k32', 13)
        cond, noise = synth_inputs(1, 64, 64, 20)
        refl = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
        assert (eng.sample(cond.cuda(), noise.cuda()).cpu() - refl).abs().max().item() <= TOL_LOOP
    finally:
        eng.set_debug(False)
        _lib.debug_option('k32', 1275)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('prec', ['f16x3', 'bf16'])
def test_subpixel_upsample_convs_on_the_form(full, prec):
    """Bit 16: the sub-pixel Upsample + Conv3x3 launches (ups.3 / ups.7 / ups.11: 256 -> 256, 256 -> 128 ... at 2x the resolution) on
    the 16x16x32 kernel: layer by layer against the oracle on a ragged map (partial source tiles), against the 32x32x16 up2 kernel."""
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    eng.set_precision(prec)
    tol = TOL_FWD if prec == 'f16x3' else 0.04
    bits = 25 if prec == 'f16x3' else 27
    _lib.debug_option('k32', bits)
    try:
        gen = torch.Generator().manual_seed(13)
        x = torch.randn(2, 6, 72, 104, generator=gen)
        nl = torch.tensor([[0.2], [0.8]])
        cap = {}
        with torch.no_grad():
            ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
        eng.set_debug(True)
        out = eng.unet_forward(x.cuda(), nl.cuda())
        torch.cuda.synchronize()
        for L in build_layers(cfg):
            d = (eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item()
            scale = max(cap[L.name].abs().max().item(), 1.0)
            assert d <= tol * scale, f'{L.name}: {d:.3e} (scale {scale:.2f})'
        eng.set_debug(False)
        assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)
        _lib.debug_option('k32', bits & ~16)
        out_d = eng.unet_forward(x.cuda(), nl.cuda())
        dd = (out_d - out).abs().max().item()
        assert dd > 0.0                   # (0.0 would mean the form was never taken)
        if prec == 'f16x3':
            assert dd <= 2e-5, dd
            assert (out.cpu() - ref).abs().max().item() <= TOL_FWD
    finally:
        eng.set_debug(False)
        eng.set_precision('f16x3')
        _lib.debug_option('k32', 1275)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('prec', ['f16x3', 'bf16'])
def test_small_workgroup_form_vs_oracle(full, prec):
    """Bit 32: the 64-cout launches on 4-wave workgroups, two per CU (6-row tiles in f16x3, 8-row tiles in bf16), which large
    grids take by default; here forced onto every grid size (k32_sb_min_wgs = 1, no split K: riders, partial tiles in both directions
    on 128 x 128 and a ragged 72 x 104 map, the start stagger on), layer by layer against the oracle, against the 8-wave forms on the
    same input, bitwise reruns, loop + hipGraph."""
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    eng.set_precision(prec)
    tol = TOL_FWD if prec == 'f16x3' else 0.04
    _lib.debug_option('k32', 1275 | 512)           # with the bf16 riders too (bit 512: off by default)
    _lib.debug_option('k32_sb_min_wgs', 1)
    _lib.debug_option('splitk', 0)          # (a launch with a K split keeps the 8-wave forms)
    _lib.debug_option('k32_stagger', 3)
    try:
        for shape, seed in (((2, 6, 128, 128), 21), ((3, 6, 72, 104), 22)):
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
            _lib.debug_option('k32', 27)                                                # the same launches on the 8-wave forms
            out_d = eng.unet_forward(x.cuda(), nl.cuda())
            _lib.debug_option('k32', 1275 | 512)
            dd = (out_d - out).abs().max().item()
            assert dd > 0.0                                                             # (0.0: the form was never taken)
            if prec == 'f16x3':
                assert dd <= 2e-5 and (out.cpu() - ref).abs().max().item() <= TOL_FWD
        if prec == 'f16x3':
            cond, noise = synth_inputs(2, 64, 64, 20)
            refl = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
            outl = eng.sample(cond.cuda(), noise.cuda()).cpu()
            assert (outl - refl).abs().max().item() <= TOL_LOOP
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                g1 = eng.sample(cond.cuda(), noise.cuda(), graph=True)
            s.synchronize()
            assert torch.equal(g1.cpu(), outl)
    finally:
        eng.set_debug(False)
        eng.set_precision('f16x3')
        _lib.debug_option('k32_sb_min_wgs', 1024)
        _lib.debug_option('k32_stagger', 0)
        _lib.debug_option('splitk', 1)
        _lib.debug_option('k32', 1275)


@pytest.mark.timeout(900)
def test_loop_forced_k32_vs_oracle_and_graph(full, forced):
    """The 20-step loop, 64x64, B=2, every eligible launch on the form; eager == hipGraph replay."""
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    _lib.debug_option('k32', 5)
    cond, noise = synth_inputs(2, 64, 64, 20)
    ref = O.p_sample_loop(O.to_torch_sd(sd), cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
    out = eng.sample(cond.cuda(), noise.cuda()).cpu()
    d = (out - ref).abs().max().item()
    print(f'forced k32 loop 64x64: max|d| = {d:.3e}')
    assert d <= TOL_LOOP
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g1 = eng.sample(cond.cuda(), noise.cuda(), graph=True)
    s.synchronize()
    assert torch.equal(g1.cpu(), out)


@pytest.mark.timeout(900)
def test_bf16_k32_layerwise(full, forced):
    """bf16 mode on the form (bit 2): same layerwise bound as the bf16 32x32x16 kernels, and the two agree to bf16 rounding of the
    intermediate tensors."""
    from fastdiffsr_amd import _lib
    from oracle import fdsr_oracle as O
    cfg, eng, sd = full
    eng.set_precision('bf16')
    _lib.debug_option('k32', 7)
    try:
        gen = torch.Generator().manual_seed(7)
        x = torch.randn(2, 6, 64, 96, generator=gen)
        nl = torch.tensor([[0.1], [0.6]])
        cap = {}
        with torch.no_grad():
            ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
        eng.set_debug(True)
        out = eng.unet_forward(x.cuda(), nl.cuda())
        torch.cuda.synchronize()
        for L in build_layers(cfg):
            d = (eng.debug_tensor(L.name).cpu() - cap[L.name]).abs().max().item()
            assert d <= 0.04 * max(cap[L.name].abs().max().item(), 1.0), f'{L.name}: {d}'
        eng.set_debug(False)
        assert torch.equal(eng.unet_forward(x.cuda(), nl.cuda()), out)
        _lib.debug_option('k32', 0)
        out_d = eng.unet_forward(x.cuda(), nl.cuda())
        rm = (out_d - out).pow(2).mean().sqrt().item()
        assert 0.0 < rm <= 2e-2, rm
        assert (out.cpu() - ref).pow(2).mean().sqrt().item() <= 2e-2
    finally:
        eng.set_debug(False)
        eng.set_precision('f16x3')


def test_ineligible_shapes_keep_the_32x32x16_kernel(full, forced):
    """inner_channel 48: widths 48 / 96 / 192 / 384, so whole-32-channel inputs (the form) alternate with 16-channel remainders and
    seams off the 32-channel grid (the 32x32x16 kernel), on a ragged 40 x 24 map."""
    from fastdiffsr_amd import _lib
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O
    kw = dict(FASTDIFFSR_UNET)
    kw.update(inner_channel=48, norm_groups=16)
    cfg = UNetConfig(**kw)
    eng = Engine(cfg)
    sd = synth_state_dict(cfg, 3)
    eng.load_state_dict(sd)
    eng.set_precision('f16x3')
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(1, 6, 40, 24, generator=gen)
    nl = torch.tensor([[0.3]])
    with torch.no_grad():
        ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl)
    a = eng.unet_forward(x.cuda(), nl.cuda())
    assert (a.cpu() - ref).abs().max().item() <= TOL_FWD
