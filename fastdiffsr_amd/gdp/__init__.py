"""GDP sibling (FastDiffSR/model/gdp_modules, `which_model_G == 'gdp'`): the guided-diffusion UNet behind the same
boundary and on the same HIP kernels."""
