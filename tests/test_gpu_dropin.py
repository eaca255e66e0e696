"""The netG plugin surface as DDPM (reference FastDiffSR/model/model.py) drives it,
on the GPU, against the oracle."""
import numpy as np
import pytest
import torch

from fastdiffsr_amd.arch import FASTDIFFSR_SCHEDULE_VAL
from fastdiffsr_amd.synth import synth_state_dict, synth_inputs

pytestmark = pytest.mark.gpu


def _opt():
    return {'phase': 'val', 'gpu_ids': [0], 'distributed': False,
            'datasets': {'train': {'l_resolution': 64}},
            'model': {'which_model_G': 'fastdiffsr', 'finetune_norm': False,
                      'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'norm_groups': 32,
                               'channel_multiplier': [1, 2, 4, 4], 'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
                      'beta_schedule': {'train': dict(FASTDIFFSR_SCHEDULE_VAL), 'val': dict(FASTDIFFSR_SCHEDULE_VAL)},
                      'diffusion': {'image_size': 256, 'channels': 3, 'conditional': True}}}


def test_ddpm_call_sequence_matches_oracle():
    from fastdiffsr_amd import networks
    from oracle import fdsr_oracle as O
    opt = _opt()
    device = torch.device('cuda')
    netG = networks.define_G(opt).to(device)                       # model.py:15
    netG.set_loss(device)                                          # model.py:79-83
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['train'], device)   # model.py:85-92
    # load_network(): strict load of a reference-format checkpoint (model.py:148-160)
    sd_np = synth_state_dict(netG.denoise_fn.cfg, 0, prefix='denoise_fn.')
    ckpt = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    for k, v in netG.state_dict().items():
        if not k.startswith('denoise_fn.'):
            ckpt[k] = v.cpu()                                      # the 12 schedule buffers
    assert len(ckpt) == 317 + 12
    netG.load_state_dict(ckpt, strict=True)
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['val'], device)
    cond, noise = synth_inputs(1, 64, 64, 20)
    netG.eval()                                                    # model.py:60
    with torch.no_grad():
        sr = netG.p_sample_loop(cond.to(device), continous=True, noise=noise.to(device))
    netG.train()                                                   # model.py:68
    assert tuple(sr.shape) == (8, 3, 64, 64)
    cfg = netG.denoise_fn.cfg
    sd = O.to_torch_sd({k[len('denoise_fn.'):]: v for k, v in sd_np.items()})
    ref, traj = O.p_sample_loop(sd, cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise, return_trajectory=True)
    assert (sr[-1:].cpu() - ref).abs().max().item() <= 1e-3        # visuals['SR'][-1] (sr_mfe.py:306)
    frames = [O.res2img(cond, cond)] + [O.res2img(traj[19 - t], cond) for t in O.continuous_frames(20)]
    assert (sr.cpu() - torch.cat(frames)).abs().max().item() <= 1e-3
    # own RNG draws: runs and is reproducible under the torch seed, like the reference
    torch.manual_seed(3)
    a = netG.super_resolution(cond.to(device), False)
    torch.manual_seed(3)
    b = netG.super_resolution(cond.to(device), False)
    assert torch.equal(a, b) or (a - b).abs().max().item() <= 1e-5
    assert str(netG).startswith('GaussianDiffusion')


def test_gpu_tensor2img_bit_exact(golden_dir):
    """GPU tensor2img (clamp -> uint8 on the device) == the reference's core/metrics.tensor2img, bit for bit."""
    import os
    from fastdiffsr_amd import metrics as M
    from oracle import fdsr_oracle as O
    g = np.load(os.path.join(golden_dir, 'metrics.npz'))
    t = torch.from_numpy(g['t'])
    np.testing.assert_array_equal(M.tensor2img(t.cuda()), g['img'])
    np.testing.assert_array_equal(M.tensor2img(t[:1].cuda()), g['gray'])
    big = torch.randn(3, 256, 256, generator=torch.Generator().manual_seed(9)) * 0.7
    np.testing.assert_array_equal(M.tensor2img(big.cuda()), O.tensor2img_u8(big.clone()))


def test_ddpm_wrapper_val_iteration(tmp_path):
    """One iteration of the reference's val loop (sr_mfe.py:274-324) through fastdiffsr_amd.model.DDPM:
    create_model -> load_network (reference-format checkpoint) -> feed_data -> test -> visuals -> PSNR."""
    from fastdiffsr_amd import model as Model, metrics as M
    from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, SCHEDULE_BUFFERS
    from fastdiffsr_amd.schedule import schedule_buffers
    from oracle import fdsr_oracle as O
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    sd_np = synth_state_dict(cfg, 0, prefix='denoise_fn.')
    ckpt = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    bufs, _ = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    ckpt.update({k: torch.from_numpy(v) for k, v in bufs.items()})
    torch.save(ckpt, str(tmp_path / 'I100_E1_gen.pth'))
    opt = _opt()
    opt['path'] = {'resume_state': str(tmp_path / 'I100_E1'), 'checkpoint': str(tmp_path)}
    diffusion = Model.create_model(opt)
    diffusion.set_new_noise_schedule(opt['model']['beta_schedule']['val'], schedule_phase='val')
    cond, noise = synth_inputs(1, 64, 64, 20)
    hr = (cond + 0.3 * torch.sin(torch.arange(64).float() / 5).view(1, 1, 1, 64)).clamp(-1, 1)
    diffusion.feed_data({'HR': hr, 'SR': cond.clone(), 'LR': cond.clone(), 'Index': torch.tensor([0])})
    # fixed noise for parity: the facade draws torch.randn / randn_like in the reference's order otherwise
    with torch.no_grad():
        diffusion.netG.eval()
        diffusion.SR = diffusion.netG.p_sample_loop(diffusion.data['SR'], True, noise=noise.cuda())
        diffusion.netG.train()
    vis = diffusion.get_current_visuals()
    assert tuple(vis['SR'].shape) == (8, 3, 64, 64) and vis['SR'].device.type == 'cpu'
    sr_img = M.tensor2img(vis['SR'][-1])
    hr_img = M.tensor2img(vis['HR'])
    sd = O.to_torch_sd({k[len('denoise_fn.'):]: v for k, v in sd_np.items()})
    ref = O.p_sample_loop(sd, cfg, O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL), cond, noise)
    ref_img = O.tensor2img_u8(ref[0])
    d_psnr = abs(M.calculate_psnr(sr_img, hr_img) - O.psnr_u8(ref_img, O.tensor2img_u8(hr[0])))
    assert d_psnr <= 0.01, d_psnr                                   # north_star: PSNR within 0.01 dB of reference
    assert (sr_img.astype(int) - ref_img.astype(int)).__abs__().max() <= 1
    # test() itself (own RNG draws) runs and leaves netG in train mode like the reference (model.py:68)
    diffusion.test(continous=False)
    assert diffusion.netG.training and tuple(diffusion.SR.shape) == (1, 3, 64, 64)
    p = diffusion.save_network(1, 100)
    assert len(torch.load(p)) == 329


def test_gpu_bicubic_lr_to_sr_bit_exact(golden_dir):
    """GPU LR->SR conditioning image == PIL Image.BICUBIC (uint8, bit for bit) and == ToTensor()*2-1."""
    import os
    from fastdiffsr_amd.data import lr_to_sr
    from oracle import pil_bicubic as PB
    g = np.load(os.path.join(golden_dir, 'bicubic.npz'))
    for name in ('x4', 'x8', 'ragged'):
        lr, sr = g[name + '/lr'], g[name + '/sr']
        batch = torch.from_numpy(np.stack([lr, lr[::-1].copy()])).cuda()
        cond, u8 = lr_to_sr(batch, sr.shape[0], sr.shape[1], want_u8=True)
        np.testing.assert_array_equal(u8[0].cpu().numpy(), sr)
        np.testing.assert_array_equal(u8[1].cpu().numpy(), PB.resize_bicubic_u8(lr[::-1].copy(), sr.shape[0], sr.shape[1]))
        assert torch.equal(cond[0].cpu(), PB.u8_to_model_tensor(sr))


def test_p_losses_forward_matches_reference_golden(golden_dir):
    """GaussianDiffusion.forward == p_losses (diffusion.py:242-273): forward value of the L1(sum)
    loss through the HIP UNet with per-sample continuous noise levels, against the loss the
    reference itself produced for the same inputs and draws (tests/golden/train_loss.npz)."""
    import os
    from unittest import mock
    from fastdiffsr_amd import networks
    g = np.load(os.path.join(golden_dir, 'train_loss.npz'))
    opt = _opt()
    device = torch.device('cuda')
    netG = networks.define_G(opt).to(device)
    netG.set_loss(device)
    netG.set_new_noise_schedule(opt['model']['beta_schedule']['val'], device)
    sd_np = synth_state_dict(netG.denoise_fn.cfg, 0, prefix='denoise_fn.')
    ckpt = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    for k, v in netG.state_dict().items():
        if not k.startswith('denoise_fn.'):
            ckpt[k] = v.cpu()
    netG.load_state_dict(ckpt, strict=True)
    netG.eval()
    hr, sr, nz = (torch.from_numpy(g[k]).to(device) for k in ('hr', 'sr', 'noise'))
    gam = g['gamma']
    with mock.patch.object(np.random, 'randint', lambda a, b: 7), \
            mock.patch.object(np.random, 'uniform', lambda a, b, size: gam):
        loss = netG({'HR': hr, 'SR': sr}, noise=nz)
    ref = float(g['loss'])
    assert abs(loss.item() - ref) <= 1e-5 * abs(ref), (loss.item(), ref)
