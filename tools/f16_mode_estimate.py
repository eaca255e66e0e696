"""What would a 16-bit mode on f16 (11 mantissa bits; same MFMA rate and bytes per element as bf16) buy on TRAINED-LIKE weights?  An
estimate without building it: the oracle's own 20-step loop run by PyTorch on the GPU under autocast -- convolutions and linears in bf16
and in fp16, everything else fp32 -- against the fp32 CPU oracle image, on the weights of tests/test_gpu_trained_weights.py.  (torch's
GPU kernels are used HERE as a calculator for a planning number; nothing of the product touches them.)

Usage (GPU box):  python tools/f16_mode_estimate.py > gpurun_out/f16_mode_estimate.txt"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import test_gpu_trained_weights as tw
    from conftest import oracle_loop_image
    from fastdiffsr_amd.arch import FASTDIFFSR_SCHEDULE_VAL
    from fastdiffsr_amd.synth import synth_inputs
    from oracle import fdsr_oracle as O
    cfg, sd, losses = tw.make_trained()
    hr, sr = tw._pairs(1, 256, 77)
    _, noise = synth_inputs(1, 256, 256, 20)
    ref = oracle_loop_image(sd, cfg, sr, noise)
    u8 = lambda t: O.tensor2img_u8(t[0].clone())
    p_ref = O.psnr_u8(u8(ref), u8(hr))
    print(f'fp32 CPU oracle against HR: {p_ref:.3f} dB', flush=True)
    tsd = {k: v.cuda() for k, v in O.to_torch_sd(sd).items()}
    tab = O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    with torch.no_grad():
        im = O.p_sample_loop(tsd, cfg, tab, sr.cuda(), noise.cuda()).float().cpu()
    print(f'torch-GPU fp32 of the same loop: max|d| vs the CPU oracle {(im - ref).abs().max().item():.3e}', flush=True)
    for name, dt in (('bf16', torch.bfloat16), ('fp16', torch.float16)):
        with torch.no_grad(), torch.autocast('cuda', dtype=dt):
            im = O.p_sample_loop(tsd, cfg, tab, sr.cuda(), noise.cuda()).float().cpu()
        rm = (im - ref).pow(2).mean().sqrt().item()
        print(f'torch-GPU {name} autocast: PSNR(out, oracle) {20 * math.log10(2.0 / max(rm, 1e-12)):6.2f} dB  rmse {rm:.3e}  max|d| '
              f'{(im - ref).abs().max().item():.3e}  PSNR delta vs HR {O.psnr_u8(u8(im), u8(hr)) - p_ref:+.5f} dB', flush=True)


if __name__ == '__main__':
    main()
