"""Dynamic range and conditioning of the fp32-grade paths (exact f32 and split-f16 "f16x3").

The synthetic weights keep every activation O(1); a trained checkpoint need not.  These cases push the
arithmetic where f16x3 could differ from fp32: a large DC offset riding the residual stream (GroupNorm
statistics are E[x^2] - E[x]^2 over per-tile partial sums), an activation plane near the f16 limit
(+-6.2e4: the staging clamps to +-65504 before the hi/lo split), and a plane at 1e-6 scale (the lo part
goes subnormal below 2^-14).  Every reference module output is compared layer by layer with the oracle:
|delta| <= 1e-4 * max(1, max|ref|), both modes."""
import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import UNetConfig, build_layers
from fastdiffsr_amd.synth import synth_state_dict

pytestmark = pytest.mark.gpu

CFG = dict(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 4, 4), attn_res=(16,),
           res_blocks=2, dropout=0.2, image_size=64)
TOL = 1e-4


def _layerwise(sd, cfg, x, nl, prec):
    from fastdiffsr_amd.engine import Engine
    from oracle import fdsr_oracle as O
    cap = {}
    with torch.no_grad():
        ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision(prec)
    eng.set_debug(True)
    out = eng.unet_forward(x.cuda(), nl.cuda()).cpu()
    worst = (0.0, '')
    for L in build_layers(cfg):
        got = eng.debug_tensor(L.name).cpu()
        assert torch.isfinite(got).all(), L.name
        d = (got - cap[L.name]).abs().max().item()
        scale = max(1.0, cap[L.name].abs().max().item())
        worst = max(worst, (d / scale, L.name))
        assert d <= TOL * scale, f'{L.name}: max|d| {d:.3e} at max|ref| {scale:.3e} [{prec}]'
    d = (out - ref).abs().max().item()
    assert d <= TOL * max(1.0, ref.abs().max().item())
    return d, worst


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
@pytest.mark.parametrize('offset', [0.0, 20.0, 100.0, 400.0])
def test_dc_offset_on_residual_stream(offset, prec):
    """A constant added by the first conv's bias (and a quarter of it by every block2 bias) rides the identity
    residuals through the whole network: GroupNorm sees mean >> std on every block input."""
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 0)
    sd['downs.0.bias'] = sd['downs.0.bias'] + np.float32(offset)
    for k in list(sd):
        if k.endswith('block2.block.3.bias'):
            sd[k] = sd[k] + np.float32(offset * 0.25)
    x = torch.randn(2, 6, 64, 64, generator=torch.Generator().manual_seed(1))
    nl = torch.tensor([[0.3], [0.8]])
    d, worst = _layerwise(sd, cfg, x, nl, prec)
    print(f'DC offset {offset:6.1f} [{prec}]: final max|d| {d:.3e}; worst layer {worst[1]} at {worst[0]:.3e} x max(1,|ref|)')


@pytest.mark.parametrize('prec', ['f32', 'f16x3'])
@pytest.mark.parametrize('scale', [1.1e4, 1.0e-6])
def test_activation_plane_scale(scale, prec):
    """downs.0 scaled so that the first activation plane (and with it the first skip tensor, which is read RAW by
    the 1x1 res_conv of ups.14 and re-normalised by every GroupNorm it enters) sits near the f16 limit
    (|x| up to 6.2e4 < 65504 anywhere on the residual stream) or at 1e-6."""
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 0)
    sd['downs.0.weight'] = sd['downs.0.weight'] * np.float32(scale)
    sd['downs.0.bias'] = sd['downs.0.bias'] * np.float32(scale)
    x = torch.randn(2, 6, 64, 64, generator=torch.Generator().manual_seed(2)).clamp(-3, 3)
    nl = torch.tensor([[0.05], [0.6]])
    from oracle import fdsr_oracle as O
    cap = {}
    with torch.no_grad():
        O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
    top = cap['downs.0'].abs().max().item()
    if scale > 1:
        # the case is what it says: the whole residual stream near, not beyond, the f16 range.  Beyond it the
        # f16x3 staging saturates RAW conv inputs (res_conv, down/upsample convs) at +-65504 -- a documented
        # limit of that mode; GroupNorm'ed inputs are re-scaled before the split.
        peak = max(v.abs().max().item() for k, v in cap.items() if k != 'final_conv')
        assert 3.0e4 < top and peak < 65504.0, (top, peak)
    else:
        assert top < 1e-5
    d, worst = _layerwise(sd, cfg, x, nl, prec)
    print(f'plane scale {scale:.0e} (max|downs.0| {top:.3e}) [{prec}]: final max|d| {d:.3e}; worst layer {worst[1]} '
          f'at {worst[0]:.3e} x max(1,|ref|)')


def test_f16x3_saturation_guard_residual_stream_at_1e5():
    """A residual stream at 1e5 (> 65504) read RAW by the res_convs and the down/upsample convs.  Before the guard the split-f16
    staging clamped it silently and the layers after it were wrong with no signal (documented below by switching the guard off);
    with the guard the C ABI reports FDSR_E_SATURATED for the call, and the facade re-runs it on the exact-fp32 kernels:
    within 1e-4 * max(1, |ref|) of the oracle."""
    import warnings
    from fastdiffsr_amd import _lib
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.unet import UNet
    from oracle import fdsr_oracle as O
    cfg = UNetConfig(**CFG)
    sd = synth_state_dict(cfg, 0)
    sd['downs.0.weight'] = sd['downs.0.weight'] * np.float32(3.0e4)     # first plane / first skips reach ~1.5e5
    sd['downs.0.bias'] = sd['downs.0.bias'] * np.float32(3.0e4)
    x = torch.randn(2, 6, 64, 64, generator=torch.Generator().manual_seed(2)).clamp(-3, 3)
    nl = torch.tensor([[0.05], [0.6]])
    cap = {}
    with torch.no_grad():
        ref = O.unet_forward(O.to_torch_sd(sd), cfg, x, nl, capture=cap)
    assert cap['downs.0'].abs().max().item() > 1.0e5
    scale = max(1.0, ref.abs().max().item())
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    eng.set_precision('f16x3')
    # (0) the default policy (Engine.on_saturation == 'f32', shared by every model family and direct caller): the call is
    # re-run on the exact-fp32 kernels, the answer is right and the engine is back in f16x3 afterwards
    assert eng.on_saturation == 'f32'
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        assert (eng.unet_forward(x.cuda(), nl.cuda()).cpu() - ref).abs().max().item() <= TOL * scale
    assert eng.precision == 'f16x3'
    # the same for a training step: the step's own forward is checked in the synchronisation that returns the loss
    tgt = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(3))
    eng.on_saturation = 'raise'
    with pytest.raises(_lib.FdsrSaturated):
        eng.train_grads(x.cuda(), nl.cuda().reshape(-1), tgt.cuda(), 'l1', 1.0 / tgt.numel())
    eng.on_saturation = 'f32'
    with warnings.catch_warnings(record=True):
        warnings.simplefilter('always')
        loss = eng.train_grads(x.cuda(), nl.cuda().reshape(-1), tgt.cuda(), 'l1', 1.0 / tgt.numel())
    assert abs(loss - (ref - tgt).abs().sum().item()) <= 1e-4 * (ref - tgt).abs().sum().item() and eng.precision == 'f16x3'
    eng.on_saturation = 'raise'
    # (1) the engine reports it
    with pytest.raises(_lib.FdsrSaturated) as ei:
        eng.unet_forward(x.cuda(), nl.cuda())
    assert ei.value.code == _lib.FDSR_E_SATURATED
    # (2) what used to happen silently: clamped raw inputs, a wrong answer, no error
    eng.check_saturation = False
    bad = eng.unet_forward(x.cuda(), nl.cuda()).cpu()
    assert (bad - ref).abs().max().item() > 1e-3 * scale
    eng.check_saturation = True
    # (3) the flag is sticky until read, then clear: a clean input afterwards passes
    with pytest.raises(_lib.FdsrSaturated):
        eng.unet_forward(x.cuda(), nl.cuda())
    # (4) the exact-fp32 mode has no such limit
    eng.set_precision('f32')
    assert (eng.unet_forward(x.cuda(), nl.cuda()).cpu() - ref).abs().max().item() <= TOL * scale
    # (5) the facade: the same policy through UNet.forward, stays in f16x3 for the next call
    net = UNet(in_channel=6, out_channel=3, norm_groups=32, inner_channel=64, channel_mults=(1, 2, 4, 4), attn_res=(16,),
               res_blocks=2, dropout=0.2, image_size=64)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    net = net.cuda().eval()
    net.engine.set_precision('f16x3')
    with torch.no_grad(), warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        got = net(x.cuda(), nl.cuda()).cpu()
    assert (got - ref).abs().max().item() <= TOL * scale
    assert net.engine.precision == 'f16x3'
    # a network whose activations stay in range never trips it
    sd0 = synth_state_dict(cfg, 0)
    e2 = Engine(cfg)
    e2.load_state_dict(sd0)
    e2.set_precision('f16x3')
    e2.unet_forward(x.cuda(), nl.cuda())
