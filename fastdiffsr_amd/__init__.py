"""fastdiffsr_amd: MI355X-native engine for FastDiffSR's 20-step sampling path.

`unet.UNet` / `diffusion.GaussianDiffusion` keep the reference's plugin surface
(FastDiffSR/model/networks.py:82-119 selects them by `which_model_G`); compute
runs in csrc/libfdsr_hip.so through the C ABI of include/fdsr.h.
"""
__all__ = ['arch', 'synth', 'schedule', 'engine', 'unet', 'diffusion', 'networks', 'config']
