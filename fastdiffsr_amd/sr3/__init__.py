"""SR3 sibling behind the same boundary (reference FastDiffSR/model/ddpm_modules, selected by
`which_model_G == 'ddpm'`, model/networks.py:84-85): `unet.UNet` / `diffusion.GaussianDiffusion`."""
