"""GPU side of the input pipeline: the conditioning "SR" image from the LR image.

The reference builds it offline with PIL (`Image.resize(..., Image.BICUBIC)` through torchvision,
FastDiffSR/data/prepare_data_mfe_dm.py:17-40) and loads it one PIL image at a time
(data/LRHR_dataset.py, data/util.py:66-75).  Here a uint8 batch is resized on the device, bit for bit
as Pillow does (8-bit fixed-point two-pass resample), and converted to the model tensor
(ToTensor() -> *2 - 1)."""
import ctypes as C

import torch

from . import _lib


def lr_to_sr(lr_u8, out_h, out_w, want_u8=False):
    """lr_u8: [B,h,w,3] uint8 CUDA tensor (RGB).  -> cond [B,3,H,W] fp32 in [-1,1] (and the uint8 SR image)."""
    if not lr_u8.is_cuda or lr_u8.dtype != torch.uint8 or lr_u8.dim() != 4 or lr_u8.shape[-1] != 3:
        raise ValueError('lr_u8 must be a CUDA uint8 tensor of shape [B,h,w,3]')
    lr_u8 = lr_u8.contiguous()
    B, h, w, _ = lr_u8.shape
    dev = lr_u8.device
    tmp = torch.empty(B * h * out_w * 3, dtype=torch.uint8, device=dev)
    out = torch.empty(B, 3, out_h, out_w, dtype=torch.float32, device=dev)
    u8 = torch.empty(B, out_h, out_w, 3, dtype=torch.uint8, device=dev) if want_u8 else None
    lib = _lib.load()
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(None, lib.fdsr_resize_bicubic_u8(None, C.c_void_p(lr_u8.data_ptr()), B, h, w, out_h, out_w, C.c_void_p(tmp.data_ptr()),
                                               C.c_void_p(u8.data_ptr()) if want_u8 else C.c_void_p(0),
                                               C.c_void_p(out.data_ptr()), C.c_void_p(st)))
    return (out, u8) if want_u8 else out
