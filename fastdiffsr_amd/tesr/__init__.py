"""TESR sibling behind the same boundary (reference FastDiffSR/model/tesr_modules, selected by
`which_model_G == 'tesr'`, model/networks.py:86-87): `unet.UNet` / `diffusion.GaussianDiffusion`.
The SwinIR classes that file also defines are not used by its UNet and are not part of the denoiser."""
