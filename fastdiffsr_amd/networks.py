"""`define_G`: the reference's generator factory (FastDiffSR/model/networks.py:82-119)
for `which_model_G == 'fastdiffsr'`, building the HIP-backed modules."""
def define_G(opt):
    model_opt = opt['model']
    which = model_opt['which_model_G']
    if which == 'ddpm':                     # networks.py:84-85: the SR3 sibling
        from .sr3 import diffusion, unet
    elif which == 'tesr':                   # networks.py:86-87
        from .tesr import diffusion, unet
    elif which in ('fastdiffsr', 'fastdiffsr_hip'):
        from . import diffusion, unet
    else:
        raise NotImplementedError(f"fastdiffsr_amd provides which_model_G in ('fastdiffsr', 'ddpm', 'tesr') (got {which!r})")
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
    return netG
