"""`define_G`: the reference's generator factory (FastDiffSR/model/networks.py:82-119)
for `which_model_G == 'fastdiffsr'`, building the HIP-backed modules."""
def define_G(opt):
    model_opt = opt['model']
    which = model_opt['which_model_G']
    if which == 'ddpm':                     # networks.py:84-85: the SR3 sibling
        from .sr3 import diffusion, unet
    elif which == 'tesr':                   # networks.py:86-87
        from .tesr import diffusion, unet
    elif which == 'gdp':                    # networks.py:88-89
        from .gdp import diffusion, unet
    elif which in ('fastdiffsr', 'fastdiffsr_hip'):
        from . import diffusion, unet
    else:
        raise NotImplementedError(f"fastdiffsr_amd provides which_model_G in ('fastdiffsr', 'ddpm', 'tesr', 'gdp') (got {which!r})")
    if ('norm_groups' not in model_opt['unet']) or model_opt['unet']['norm_groups'] is None:
        model_opt['unet']['norm_groups'] = 32
    u = model_opt['unet']
    model = unet.UNet(in_channel=u['in_channel'], out_channel=u['out_channel'], norm_groups=u['norm_groups'],
                      inner_channel=u['inner_channel'], channel_mults=u['channel_multiplier'], attn_res=u['attn_res'],
                      res_blocks=u['res_blocks'], dropout=u['dropout'], image_size=model_opt['diffusion']['image_size'])
    netG = diffusion.GaussianDiffusion(model, image_size=model_opt['diffusion']['image_size'],
                                       channels=model_opt['diffusion']['channels'], loss_type='l1',
                                       conditional=model_opt['diffusion']['conditional'],
                                       schedule_opt=model_opt['beta_schedule']['train'],
                                       **({} if which == 'ddpm' else
                                          {'scale': int(256 / int(opt['datasets']['train']['l_resolution']))}))
    if opt['phase'] == 'train':
        init_weights(netG, init_type='orthogonal')                   # networks.py:113-115
    # networks.py:116-118 wraps netG in nn.DataParallel when distributed; here one process drives one GPU and
    # data parallelism is an all-reduce of the engine's gradient arena (parallel.allreduce_grads): nothing to wrap
    return netG


def init_weights(net, init_type='kaiming', scale=1, std=0.02):
    """networks.py:13-75: `net.apply(fn)` visits the leaf modules in registration order, which is the order of the
    checkpoint schema; Conv / Linear weights are initialised (orthogonal gain 1, kaiming_normal fan_in * scale, or
    N(0, std)), their biases zeroed, GroupNorm left alone.  Same torch RNG consumption as the reference: with the
    same torch.manual_seed the tensors are identical (tests/golden/init_weights.npz)."""
    import torch
    from torch.nn import init
    unet = net.denoise_fn
    named = dict(unet.named_parameters())
    if init_type not in ('normal', 'kaiming', 'orthogonal'):
        raise NotImplementedError('initialization method [{:s}] not implemented'.format(init_type))
    with torch.no_grad():
        for key, p in named.items():
            if not key.endswith(".weight") or p.dim() not in (2, 3, 4):      # Linear, Conv1d (GDP attention), Conv2d
                continue                                             # GroupNorm affine, biases
            if init_type == 'orthogonal':
                init.orthogonal_(p, gain=1)
            elif init_type == 'kaiming':
                init.kaiming_normal_(p, a=0, mode='fan_in')
                p.mul_(scale)
            else:
                init.normal_(p, 0.0, std)
            b = named.get(key[:-len('weight')] + 'bias')
            if b is not None:
                b.zero_()
