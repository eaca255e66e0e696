"""Headline benchmark: 256x256 SR images/s at T=20 (BASELINE.json), one process per GPU.

A "step" = one full 20-step conditional sampling pass (20 UNet forwards + 20 posterior
updates) over one batch of synthetic inputs per GPU; inputs (cond, noise) are resident in
HBM before the timed region.  Default workload = BASELINE configs[1]: x4 64->256,
batch=16 per GPU, T=20, random-init UNet, fp32-grade arithmetic.  The default conv arithmetic is
"f16x3" (every fp32 operand split hi/lo into two f16, three f16 MFMAs per product, fp32
accumulate): it meets the fp32 parity bound (|delta| < 1e-3; measured ~3e-6, tests/test_gpu_parity.py)
at ~3x the throughput of the exact-fp32 MFMA path, which stays available as --precision f32.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

FLOPS_PER_IMAGE = 5.3662e12      # 20 x 268.31 GFLOP
PEAK_F32_MFMA = 157.3            # TFLOP/s, MI355X_MICROARCH.md (f32 matrix = f32 vector peak)
PEAK_16BIT_MFMA = 2500.0         # TFLOP/s dense f16/bf16 MFMA (same guide; never the 2:1-sparse figure)


def host_cores():
    """CPU threads this process may really use (fastdiffsr_amd.parallel.host_cores: affinity mask capped by the cgroup quota)."""
    from fastdiffsr_amd.parallel import host_cores as hc
    return hc()


class GpuTelemetry:
    """Mean shader clock and package power of one GPU over a timed region, read from the amdgpu hwmon files of its card (no
    subprocess, no HIP call: a sampler thread that reads two small sysfs files every `period` seconds).  The claim "this regime is
    (not) power-bound" is then checkable from the bench line alone: `sclk_mhz` against the 2400 MHz nominal clock, `power_w` against
    `power_cap_w`.  Every field is None where the box does not expose the file."""

    def __init__(self, index=0, period=0.05):
        import glob
        self.period, self.samples, self._thr, self._stop = period, [], None, False
        cards = []
        for hw in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
            if any(os.path.exists(os.path.join(hw, f)) for f in ('power1_average', 'power1_input')):
                cards.append(hw)
        # the card of HIP device `index`: by PCI address (a container may see the sysfs nodes of GPUs it was not given)
        self.hw, self.selection = None, None
        try:
            pr = torch.cuda.get_device_properties(index)
            addr = '%04x:%02x:%02x.' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for hw in cards:
                if addr in os.path.realpath(os.path.join(hw, '..', '..')):
                    self.hw, self.selection = hw, 'pci ' + addr + '0'
        except Exception:
            pass
        if self.hw is None and cards:
            self.hw, self.selection = (cards[index] if index < len(cards) else cards[0]), 'card order (no PCI match)'
        self.cap = self._read('power1_cap', 1e-6)

    def _read(self, name, scale):
        if self.hw is None:
            return None
        try:
            return float(open(os.path.join(self.hw, name)).read().split()[0]) * scale
        except (OSError, ValueError, IndexError):
            return None

    def _loop(self):
        while not self._stop:
            pw = self._read('power1_average', 1e-6)
            if pw is None:
                pw = self._read('power1_input', 1e-6)
            self.samples.append((self._read('freq1_input', 1e-6), pw))
            time.sleep(self.period)

    def __enter__(self):
        import threading
        self._stop, self.samples = False, []
        if self.hw is not None:
            self._thr = threading.Thread(target=self._loop, daemon=True)
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thr is not None:
            self._thr.join()
        return False

    def summary(self):
        def stat(i):
            v = [x[i] for x in self.samples if x[i] is not None]
            return (sum(v) / len(v), min(v), max(v)) if v else (None, None, None)
        (cm, cl, ch), (pm, pl, ph) = stat(0), stat(1)
        return {'sclk_mhz': cm, 'sclk_mhz_min': cl, 'sclk_mhz_max': ch, 'power_w': pm, 'power_w_min': pl, 'power_w_max': ph,
                'power_cap_w': self.cap, 'samples': len(self.samples), 'period_s': self.period,
                'source': (self.hw + '/{freq1_input,power1_average}') if self.hw else None, 'device_selected_by': self.selection}


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(cfg, sd, budget_s=32.0, images=3):
    """The oracle (CPU restatement of the reference loop) timed on the host cores, on a BOUNDED sample
    (BASELINE.md section 3): after one warm-up UNet forward, full 20-step loops of 256x256 images at B=1 -- three
    images, or as many reverse steps as fit in ~budget_s (every step costs the same: one UNet forward + the
    posterior update) -- then ONE B=4 UNet forward (the reference gains nothing from batching on CPU).
    images/s = 1 / (20 * mean step time).  Reported baseline only."""
    from oracle import fdsr_oracle as O
    from fastdiffsr_amd.arch import FASTDIFFSR_SCHEDULE_VAL
    from fastdiffsr_amd.synth import synth_inputs
    cores = host_cores()
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    tsd = O.to_torch_sd(sd)
    tab = O.schedule_tables(FASTDIFFSR_SCHEDULE_VAL)
    cond, noise = synth_inputs(images, 256, 256, 20)
    first = None
    with torch.no_grad():
        O.unet_forward(tsd, cfg, torch.cat([cond[:1], noise[0, :1]], 1), torch.full((1, 1), 0.5))   # warm-up
        steps, t0 = 0, time.perf_counter()
        for i in range(images):
            c, nz = cond[i:i + 1], noise[:, i:i + 1]
            img, k = nz[0], 0
            for t in reversed(range(20)):
                img = O.p_sample(tsd, cfg, tab, img, t, c, nz[k + 1] if t > 0 else None)
                k += 1
                steps += 1
                if time.perf_counter() - t0 > budget_s:
                    break
            if first is None:
                first = {'cond': c, 'noise': nz, 'k': k, 'x_k': img}
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
        x4 = torch.cat([cond[:1].expand(4, -1, -1, -1), noise[0, :1].expand(4, -1, -1, -1)], 1).contiguous()
        t1 = time.perf_counter()
        O.unet_forward(tsd, cfg, x4, torch.full((4, 1), 0.5))
        dt4 = time.perf_counter() - t1
    step = dt / steps
    ips = 1.0 / (20 * step)
    res = {'value': ips, 'unit': 'images/s', 'cores': threads, 'kind': 'port',
           'sample': f'{steps} reverse steps ({steps / 20:.2f} images, B=1, full 20-step loops) of 256x256 fp32, PyTorch-CPU '
                     f'restatement (oracle/), {dt:.1f} s after 1 warm-up forward; then one B=4 UNet forward ({dt4:.2f} s); '
                     f'{threads} threads (host reports {cores} usable)',
           'cpu_model': cpu_model(), 'seconds_per_image': 20 * step, 'gflops': FLOPS_PER_IMAGE / 1e9 / (20 * step),
           'b4_forward_seconds': dt4, 'b4_forward_gflops': 4 * FLOPS_PER_IMAGE / 20 / 1e9 / dt4}
    return res, first


def parity_check(eng, dev, ref):
    """The image the CPU baseline leg just produced is also the checker: the same cond / noise through the
    HIP engine (B=1, explicit noise), compared after the k steps the oracle got through.  PSNR is taken the
    reference's way (tensor2img -> uint8, core/metrics.py:16-42, :94-101) against a synthetic HR
    (cond + a smooth residual) and only when all 20 steps ran."""
    import math
    from fastdiffsr_amd.metrics import tensor2img, calculate_psnr
    cond, noise, k = ref['cond'].to(dev), ref['noise'].to(dev), ref['k']
    out, traj = eng.sample(cond, noise, want_traj=True)
    got = traj[k - 1].cpu()
    res = {'vs': f'oracle state after {k} of 20 steps, 1 image 256x256, same cond and noise',
           'max_abs_diff_x_t': float((got - ref['x_k']).abs().max())}
    if k == 20:
        c = ref['cond']
        yy, xx = torch.meshgrid(torch.arange(256.), torch.arange(256.), indexing='ij')
        r = torch.stack([torch.sin(2 * math.pi * (yy / 64 + ch / 3)) * torch.cos(2 * math.pi * xx / 48) for ch in range(3)])[None]
        hr = (c + 0.5 * r).clamp(-1, 1)
        ref_img = ref['x_k'].clamp(-1, 1) / 2.0 + c                    # res2img (diffusion.py:275-281)
        diff_img = float((out.cpu() - ref_img).abs().max())
        hr_u8 = tensor2img(hr[0].clone())                            # tensor2img clamps in place, like the reference's
        p_ref = calculate_psnr(tensor2img(ref_img[0].clone()), hr_u8)
        p_got = calculate_psnr(tensor2img(out[0].cpu()), hr_u8)
        res.update({'max_abs_diff_image': diff_img,
                    'psnr_db': p_got, 'psnr_oracle_db': p_ref, 'psnr_delta_db': p_got - p_ref})
    return res


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; this line-oriented
    benchmark owes its caller exactly one JSON line there, so fd 1 points at stderr while RCCL initialises."""

    @staticmethod
    def _flush_c_stdio():
        # RCCL writes through C stdio: with stdout a pipe that buffer is flushed only at exit, i.e. AFTER fd 1
        # has been restored -- flush it while fd 1 still points at stderr
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    def __enter__(self):
        sys.stdout.flush()
        self._flush_c_stdio()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._flush_c_stdio()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def kernel_source_hash():
    """SHA-256 of the sources of the SAMPLING kernels (the conv family the traffic figure is about and everything they
    include): a PMC traffic file under profiles/ is only quoted while it matches."""
    import hashlib
    d = os.path.join(ROOT, 'fastdiffsr_amd', 'csrc')
    h = hashlib.sha256()
    for f in ('fdsr_act_io.h', 'fdsr_conv_h.hip', 'fdsr_conv_k32.hip', 'fdsr_conv_tail.hip', 'fdsr_conv_strip.hip', 'fdsr_conv_up2.hip', 'fdsr_kernels.h', 'fdsr_kernels.hip'):
        h.update(f.encode() + b'\0' + open(os.path.join(d, f), 'rb').read() + b'\0')
    return h.hexdigest()


KNOCKOUT = False   # --debug-option knockout=...: timing-only probes whose results are garbage
from fastdiffsr_amd._lib import K32, K32_DEFAULT, STRIP, STRIP_DEFAULT, bit_names   # include/fdsr.h: enum fdsr_k32_bits / fdsr_strip_bits
K32_BITS = K32_DEFAULT       # --debug-option k32=... overrides it for the labels below
STRIP_BITS = STRIP_DEFAULT   # the column-strip form of the 64-cout launches (fdsr_conv_strip.hip)


def family_label(precision):
    """Names of the kernels the 3x3 family's launches run on under the active options (what to sum in a rocprofv3 stats file)."""
    if precision == 'f32':
        return 'conv_mfma_f32_kernel'
    bit = K32['F16X3'] if precision == 'f16x3' else K32['BF16']     # (the f16 mode takes the bf16 mode's kernel forms)
    names = []
    if STRIP_BITS & (STRIP['F16X3_64'] if precision == 'f16x3' else (STRIP['BF16_64'] | STRIP['BF16_CAT64'] | STRIP['BF16_RIDER'] | STRIP['BF16_CAT128'] | STRIP['BF16_COUT128'])):
        names.append('conv_strip_kernel')
    if K32_BITS & bit:
        names.append('conv_k32_kernel')
        if K32_BITS & K32['UP2']:
            names.append('conv_up2_k32_kernel')
    names.append('conv_mfma_h_kernel<3, ...>')
    if not (K32_BITS & bit and K32_BITS & K32['UP2']):
        names.append('conv_up2_h_kernel')
    names.append('conv_in8_kernel + conv_out3_kernel (the two ends of the UNet)')
    return ' + '.join(names)


def conv_roofline(prof, precision, B, S, dt_total, round_tag='r06'):
    """Roofline of the dominant kernel family = every 3x3 convolution launch (implicit-GEMM MFMA kernels incl. the
    sub-pixel upsample form and the 6-channel input conv): algorithmic FLOPs / HIP-event time around those launches."""
    ach = prof['conv_flops'] / (prof['conv_ms'] * 1e-3) / 1e12
    peak = PEAK_F32_MFMA if precision == 'f32' else PEAK_16BIT_MFMA
    passes = 3 if precision == 'f16x3' else 1     # MFMA products issued per algorithmic product
    kern = family_label(precision)
    # HBM bytes per launch of the SAME launch set (3x3 family), from rocprofv3 --pmc passes of this command kept
    # under profiles/ (separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read correction: tools/pmc_traffic.py).
    # Quoted only while the file was measured on these very kernel sources; otherwise null.
    traffic, traffic_src = None, None
    tf = os.path.join(ROOT, 'profiles', f'{round_tag}_pmc_hbm_traffic_{precision}_b{B}.json')
    if S == 256 and os.path.exists(tf):
        try:
            t = json.load(open(tf))
            if t.get('kernel_source_hash') == kernel_source_hash():
                traffic = t['family_3x3']['hbm_bytes_per_launch']
                traffic_src = os.path.relpath(tf, ROOT)
        except Exception:
            traffic = None
    n_timed = len(range(0, 20, 4))
    r = {'bound': 'mfma', 'kernel': f'{kern} (3x3 implicit-GEMM family)',
         'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
         'traffic': traffic, 'traffic_unit': 'bytes/launch (PMC)', 'traffic_source': traffic_src,
         'algorithmic_bytes_per_launch': prof['conv_bytes'] / max(prof['launches'], 1),
         'launches': prof['launches'],
         'avg_launch_ms': prof['conv_ms'] / max(prof['launches'], 1),
         'mfma_passes_per_product': passes, 'executed_tflops': ach * passes,
         'frac_executed': ach * passes / peak,
         'algorithmic_gbytes_per_s': prof['conv_bytes'] / (prof['conv_ms'] * 1e-3) / 1e9,
         # the engine brackets the conv launches of every 4th diffusion step with HIP
         # events (5 of 20 steps): <1 % overhead in the timed region, measured 2.6 % with all
         'timed_steps_of_20': n_timed}
    if precision == 'f16x3':
        r['executed_note'] = 'executed_tflops prices every product at 3 MFMA passes (hi*hi + hi*lo + lo*hi)'
    if precision != 'f32':
        # (until round 2's last change these launches did not contain the 1x1 res_convs: their time sat outside the family)
        r['launch_set'] = ('every 3x3 conv launch; 1x1 res_convs that ride inside a direct block2 launch (ConvParams::xr0) have their '
                           'FLOPs, bytes and time in these figures')
    if dt_total:
        r['conv_time_share'] = prof['conv_ms'] * 1e-3 * (20 / n_timed) / dt_total
    return r


def whole_path(ips_per_gpu, precision, S=256):
    """the whole path against both roofs, per GPU (ideal-fused traffic 32.85 GB / image fp32, 16.4 bf16, at 256 x 256;
    FLOPs and bytes scale with the pixel count)."""
    px = (S / 256.0) ** 2
    gb = (16.4 if precision in ('bf16', 'f16') else 32.85) * px
    flops = FLOPS_PER_IMAGE * px
    return {'tflops': ips_per_gpu * flops / 1e12,
            'frac_mfma_peak': ips_per_gpu * flops / 1e12 / (PEAK_F32_MFMA if precision == 'f32' else PEAK_16BIT_MFMA),
            'ideal_fused_gbytes_per_s': ips_per_gpu * gb, 'frac_hbm_peak': ips_per_gpu * gb / 8000.0}


def run_config(eng, dev, precision, B, S, steps, warmup, graph, noise_mode, rank=0, sync=None, want_profile=True, per_pass=None, telemetry=None):
    """Time `steps` passes of the hot path for one configuration.  Returns (seconds, out tensor, profile or None).
    With graph=True the timed region replays the captured loop; the per-launch roofline then comes from ONE extra
    eager pass after the timed region (HIP events cannot bracket launches inside a graph replay).
    per_pass: a list that receives every pass's own seconds (a synchronisation after each pass; sub-records only -- the
    headline times its K steps in one bracket, as the contract says).  telemetry: a dict that receives the mean shader clock and
    package power of the timed region (GpuTelemetry)."""
    from fastdiffsr_amd.synth import synth_inputs
    eng.set_precision(precision)
    if noise_mode == 'tensor':
        cond, noise = synth_inputs(B, S, S, 20, cond_seed=1234 + rank, noise_seed=4321 + rank)
        cond, noise = cond.to(dev), noise.to(dev)
    else:   # the engine draws x_T and the 19 per-step planes inside the timed loop (seed per rank)
        cond, _ = synth_inputs(B, S, S, 1, cond_seed=1234 + rank)
        cond, noise = cond.to(dev), None
        eng.set_seed(4321 + rank)
    out = torch.empty(B, 3, S, S, device=dev)
    if sync is None:
        def sync():
            torch.cuda.synchronize(dev)
    for _ in range(warmup):
        eng.sample(cond, noise, out=out, graph=graph)
    sync()
    profile = want_profile and not graph
    if profile:
        eng.profile_begin()
    tel = GpuTelemetry(dev.index or 0) if telemetry is not None else None
    if tel is not None:
        tel.__enter__()
    t0 = time.perf_counter()
    for _ in range(steps):
        tp = time.perf_counter()
        eng.sample(cond, noise, out=out, graph=graph)
        if per_pass is not None:
            sync()
            per_pass.append(time.perf_counter() - tp)
    sync()
    dt = time.perf_counter() - t0
    if tel is not None:
        tel.__exit__()
        telemetry.update(tel.summary())
    prof = eng.profile_end() if profile else None
    prof_dt = dt
    if want_profile and graph:
        eng.profile_begin()
        eng.sample(cond, noise, out=out, graph=False)
        torch.cuda.synchronize(dev)
        prof = eng.profile_end()
        prof_dt = None
    assert KNOCKOUT or torch.isfinite(out).all()
    return dt, out, prof, prof_dt


def run_train(eng, dev, B, S, steps, warmup, rank=0, sync=None, allreduce=None, precision='f16x3', per_pass=None):
    """Time `steps` optimisation steps (BASELINE configs[4], "training step"): noise draw + img2res + q_sample + forward + L1 loss /
    (b*c*h*w) + backward + Adam, all in the engine (no ATen kernel in the timed region), Dropout(0.2) live as in .train() mode.
    Returns seconds.  A "step" = one optimizer step over one batch of B synthetic 256x256 HR/SR pairs per GPU."""
    g = torch.Generator().manual_seed(777 + rank)
    hr = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).to(dev)
    sr = (hr + 0.1 * torch.randn(B, 3, S, S, generator=g).to(dev)).clamp(-1, 1)
    eng.set_precision('f32' if precision in ('bf16', 'f16') else precision)
    eng.set_training(True)
    eng.set_seed(99 + rank)
    world = int(os.environ.get('WORLD_SIZE', 1))
    if sync is None:
        def sync():
            torch.cuda.synchronize(dev)

    rs = np.random.RandomState(555 + rank)

    def one():
        # the whole step is the engine's: gamma is drawn on the host as the reference draws it (numpy, diffusion.py:246-256: B
        # floats), the noise inside the engine (Philox), and img2res + q_sample + cat run in the kernel that writes the packed input
        gamma = torch.from_numpy(rs.uniform(0.4, 0.9, size=B).astype(np.float32)).to(dev, non_blocking=True)
        loss = eng.train_grads_pairs(hr, sr, gamma, None, 'l1', 1.0 / (B * 3 * S * S * world))
        if allreduce is not None:
            allreduce(eng)
        eng.adam_step(1e-4)
        return loss
    for _ in range(warmup):
        one()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        tp = time.perf_counter()
        loss = one()           # returns the loss: the step is synchronous (fdsr_train_grads hands the scalar back)
        if per_pass is not None:
            sync()
            per_pass.append(time.perf_counter() - tp)
    sync()
    dt = time.perf_counter() - t0
    eng.set_training(False)
    assert loss == loss
    return dt


def train_roofline(ips_per_gpu, precision):
    """The optimisation step against the MFMA roof of its arithmetic: forward + input gradients + weight gradients = 3 x the
    forward's 268.31 GFLOP per image, every one of them a convolution of the same family.  f16x3 issues three f16
    MFMAs per product on the 2.5 PF pipe; exact fp32 runs v_mfma_f32_32x32x2_f32 (157.3 TF).  Whole-step figure: time of
    the complete step (loss, GroupNorm backward, Adam included), not of the conv launches alone."""
    tf = ips_per_gpu * 3 * FLOPS_PER_IMAGE / 20 / 1e12
    peak = PEAK_F32_MFMA if precision == 'f32' else PEAK_16BIT_MFMA
    passes = 3 if precision == 'f16x3' else 1
    fam = ('conv_mfma_f32_kernel (forward, input gradients) + wgrad_kernel (weight gradients)' if precision == 'f32' else
           family_label(precision) + ' (forward, input gradients) + wgrad_h8i_kernel / wgrad_h_kernel (weight gradients)')
    return {'bound': 'mfma', 'kernel': fam, 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s', 'frac': tf / peak,
            'mfma_passes_per_product': passes, 'executed_tflops': tf * passes, 'frac_executed': tf * passes / peak, 'traffic': None,
            'scope': 'whole optimisation step (algorithmic conv FLOPs of forward + backward / step time)'}


def pass_stats(per_pass, units):
    """min / median / max over the timed passes of a record, as images/s (each pass synchronised on its own)."""
    v = sorted(units / t for t in per_pass)
    return {'passes': len(v), 'min': v[0], 'median': float(np.median(v)), 'max': v[-1], 'unit': 'images/s per pass'}


def train_record(cfg, sd, dev, B, S, steps, warmup, precision='f16x3'):
    """An engine of its own: optimiser steps move the weights, the sampling engine (and its parity check) must not see that."""
    try:
        from fastdiffsr_amd.engine import Engine
        eng = Engine(cfg)
        eng.load_state_dict(sd)
        pp = []
        dt = run_train(eng, dev, B, S, steps, warmup, precision=precision, per_pass=pp)
        del eng
        torch.cuda.empty_cache()
        ips = B * steps / dt
        # forward + backward = 3 x the forward's 268.31 GFLOP per image, exact fp32 MFMA
        tf = ips * 3 * FLOPS_PER_IMAGE / 20 / 1e12
        return {'value': ips, 'unit': 'images/s (one optimisation step per batch)', 'ms_per_step': 1e3 * dt / steps, 'steps': steps,
                'warmup': warmup, 'dtype': precision, 'batch': B, 'per_pass': pass_stats(pp, B),
                'workload': 'configs[4] per-GPU slice: batch 32, 256x256, q_sample + L1(sum)/(b*c*h*w) (define_G fixes loss_type l1) + '
                            'backward + Adam, Dropout(0.2) live; ' + ('everything exact fp32' if precision == 'f32' else
                                                                 'every convolution (forward, input and weight gradients) f16x3 = fp32-grade'),
                'algorithmic_tflops': tf, 'roofline': train_roofline(ips, precision)}
    except Exception as e:
        return {'error': f'{type(e).__name__}: {e}'}


def synth_folder(root, n, hr_size=256, lr_size=64, workers=8):
    """A dataset folder in the reference's layout (data/prepare_data_mfe_dm.py: hr_{r}/, lr_{l}/, sr_{l}_{r}/ of PNG files): n
    smooth-plus-noise RGB images, the LR / SR members made with Pillow's bicubic as the reference's preparation script does."""
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    import shutil
    subs = (f'hr_{hr_size}', f'lr_{lr_size}', f'sr_{lr_size}_{hr_size}')
    for sub in subs:
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    yy, xx = np.mgrid[0:hr_size, 0:hr_size]
    unique = min(n, 32)        # 32 different images; the rest of the n files are copies of them (every file is still decoded)

    def one(i):
        rng = np.random.default_rng(1000 + i)
        hr = np.stack([127 + 90 * np.sin(xx / (5.0 + i % 7) + c) * np.cos(yy / (9.0 + (i // 7) % 5) + c) for c in range(3)], -1)
        hr = np.clip(hr + rng.normal(0, 6, hr.shape), 0, 255).astype(np.uint8)
        him = Image.fromarray(hr)
        lim = him.resize((lr_size, lr_size), Image.BICUBIC)
        sim = lim.resize((hr_size, hr_size), Image.BICUBIC)
        name = '%05d.png' % i
        him.save(os.path.join(root, f'hr_{hr_size}', name))
        lim.save(os.path.join(root, f'lr_{lr_size}', name))
        sim.save(os.path.join(root, f'sr_{lr_size}_{hr_size}', name))

    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(one, range(unique)))
    for i in range(unique, n):
        for sub in subs:
            shutil.copyfile(os.path.join(root, sub, '%05d.png' % (i % unique)), os.path.join(root, sub, '%05d.png' % i))
    return root


def facade_opt(root, phase, batch_size=32, workers=8):
    """The option tree core/logger.py parses from config/sr_fastdiffsr_{train,test}_64_256.json, for a synthetic folder."""
    from fastdiffsr_amd.arch import FASTDIFFSR_SCHEDULE_VAL
    from fastdiffsr_amd.config import dict_to_nonedict
    ds = {'dataroot': root, 'datatype': 'img', 'l_resolution': 64, 'r_resolution': 256, 'data_len': -1}
    return dict_to_nonedict({
        'name': 'bench_facade', 'phase': phase, 'gpu_ids': [0], 'distributed': False,
        'path': {'log': None, 'results': os.path.join(root, 'results'), 'checkpoint': os.path.join(root, 'checkpoint'), 'resume_state': None},
        'datasets': {'train': dict(ds, name='t', mode='HR', batch_size=batch_size, num_workers=workers, use_shuffle=True),
                     'val': dict(ds, name='v', mode='LRHR')},
        'model': {'which_model_G': 'fastdiffsr', 'finetune_norm': False,
                  'unet': {'in_channel': 6, 'out_channel': 3, 'inner_channel': 64, 'norm_groups': 32, 'channel_multiplier': [1, 2, 4, 4],
                           'attn_res': [16], 'res_blocks': 2, 'dropout': 0.2},
                  'beta_schedule': {'train': dict(FASTDIFFSR_SCHEDULE_VAL), 'val': dict(FASTDIFFSR_SCHEDULE_VAL)},
                  'diffusion': {'image_size': 256, 'channels': 3, 'conditional': True}},
        'train': {'n_iter': 10 ** 9, 'val_freq': 10 ** 9, 'save_checkpoint_freq': 10 ** 9, 'print_freq': 10 ** 9,
                  'optimizer': {'type': 'adam', 'lr': 1e-4}}})


def facade_records(cfg, sd, dev, S, engine_ips, n_images=256, batch=16):
    """The path a user of the reference drives, end to end, on files: `val_e2e` = fastdiffsr_amd.val.run over a folder of n_images
    256x256 pairs (PNG decode in loader threads, uint8 over PCIe, the 20-step loop, device tensor2img + MSE / PSNR / SSIM / ERGAS,
    uint8 back, .tif files written) -- wall clock of the whole call; `train_facade_b32` = create_model(opt) + the threaded loader +
    DDPM.feed_data / optimize_parameters (model/model.py:44-57), the loop of fastdiffsr_amd.train.run."""
    import shutil
    import tempfile
    recs = {}
    if S != 256:
        return recs
    root = tempfile.mkdtemp(prefix='fdsr_bench_')
    try:
        t0 = time.perf_counter()
        synth_folder(root, n_images)
        t_make = time.perf_counter() - t0
        from fastdiffsr_amd import val as V
        from fastdiffsr_amd.dataset import ThreadedBatchLoader, create_dataset
        from fastdiffsr_amd.model import create_model
        tsd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        try:
            opt = facade_opt(root, 'val')
            model = create_model(opt)
            model.netG.denoise_fn.load_state_dict(tsd, strict=True)
            log = []
            V.run(opt, batch=batch, results=os.path.join(root, 'warm'), max_images=2 * batch, log=log.append, diffusion=model)   # warm-up: uploads, graph capture
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            r = V.run(opt, batch=batch, results=os.path.join(root, 'out'), log=log.append, diffusion=model)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            written = len(os.listdir(os.path.join(root, 'out')))
            ips = r['images'] / dt
            recs['val_e2e'] = {
                'value': ips, 'unit': 'images/s', 'images': r['images'], 'files_written': written, 'batch': batch, 'dtype': 'f16x3', 'seconds': dt,
                'sampling_seconds': r['sample_seconds_this_rank'], 'sampling_share': r['sample_seconds_this_rank'] / dt,
                'vs_engine_headline': ips / engine_ips if engine_ips else None,
                'sr_psnr': r['sr_psnr'], 'sr_ssim': r['sr_ssim'], 'bic_psnr': r['bic_psnr'], 'host_seconds': r['host_seconds'],
                'workload': f'fastdiffsr_amd.val.run (= python -m fastdiffsr_amd.val --batch {batch}) over {n_images} synthetic 256x256 PNG pairs (32 distinct images): decode, '
                            'H2D as uint8, 20-step loop (torch-drawn noise, hipGraph replay from the 2nd batch), device tensor2img + MSE/PSNR/SSIM/ERGAS, '
                            'D2H as uint8, one .tif per image -- wall clock of the whole call',
                'dataset_build_seconds': t_make, 'host_cores': host_cores()}
            # the same call in the f16 mode (round 6) at B = 64, engine-drawn noise: does the harness around the loop keep up with 2.2 x the rate?
            try:
                V.run(opt, batch=64, precision='f16', rng='engine', results=os.path.join(root, 'warm16'), max_images=128, log=log.append, diffusion=model)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                r16 = V.run(opt, batch=64, precision='f16', rng='engine', results=os.path.join(root, 'out16'), log=log.append, diffusion=model)
                torch.cuda.synchronize(dev)
                dt16 = time.perf_counter() - t0
                recs['val_e2e_f16'] = {
                    'value': r16['images'] / dt16, 'unit': 'images/s', 'images': r16['images'], 'batch': 64, 'dtype': 'f16', 'seconds': dt16,
                    'files_written': len(os.listdir(os.path.join(root, 'out16'))),
                    'sampling_seconds': r16['sample_seconds_this_rank'], 'sampling_share': r16['sample_seconds_this_rank'] / dt16,
                    'sr_psnr': r16['sr_psnr'], 'sr_psnr_f16x3': r['sr_psnr'], 'sr_ssim': r16['sr_ssim'],
                    'workload': 'fastdiffsr_amd.val.run(precision="f16", batch=64, rng="engine") over the same folder: wall clock of the whole call'}
            except Exception as e:
                recs['val_e2e_f16'] = {'error': f'{type(e).__name__}: {e}'}
            del model
            torch.cuda.empty_cache()
        except Exception as e:
            recs['val_e2e'] = {'error': f'{type(e).__name__}: {e}'}
        try:
            B, steps, warmup = 32, 6, 2       # 8 batches = the folder's 256 images once: no epoch turn (a pipeline refill of ~35 ms, tools/train_facade_probe.py) inside the timed steps
            opt = facade_opt(root, 'train', batch_size=B)
            torch.manual_seed(7)
            np.random.seed(7)
            model = create_model(opt)
            model.netG.denoise_fn.load_state_dict(tsd, strict=True)
            ops = V.HipOps('cuda')
            loader = ThreadedBatchLoader(create_dataset(opt['datasets']['train'], 'train'), B, shuffle=True, workers=8, stage=ops.stage_host)
            model.set_new_noise_schedule(opt['model']['beta_schedule']['train'], schedule_phase='train')
            it = iter(loader)
            pp = []
            for k in range(warmup + steps):
                if k == warmup:
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                tp = time.perf_counter()
                try:
                    data = next(it)
                except StopIteration:        # the next epoch of the folder
                    it = iter(loader)
                    data = next(it)
                data.pop('Index')
                model.feed_data({key: ops.to_tensor(ops.to_device(v)) for key, v in data.items()})
                model.optimize_parameters()
                if k >= warmup:
                    pp.append(time.perf_counter() - tp)     # optimize_parameters returns the loss: the step is synchronous
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            ips = B * steps / dt
            recs['train_facade_b32'] = {
                'value': ips, 'unit': 'images/s (one optimisation step per batch)', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': warmup,
                'dtype': 'f16x3', 'batch': B, 'per_pass': pass_stats(pp, B), 'l_pix': model.get_current_log().get('l_pix'),
                'workload': 'create_model(opt) + threaded loader over PNG files + DDPM.feed_data / optimize_parameters (model/model.py:44-57): the loop '
                            'of fastdiffsr_amd.train.run; numpy-drawn t and gamma, torch-drawn noise, Dropout(0.2) live, every convolution f16x3'}
            del model
            torch.cuda.empty_cache()
        except Exception as e:
            recs['train_facade_b32'] = {'error': f'{type(e).__name__}: {e}'}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return recs


def sub_record(eng, dev, name, precision, B, S, steps, warmup, graph, note):
    try:
        pp, tel = [], {}
        dt, _, prof, prof_dt = run_config(eng, dev, precision, B, S, steps, warmup, graph, 'engine', per_pass=pp, telemetry=tel)
        ips = B * steps / dt
        r = {'value': ips, 'unit': 'images/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': warmup,
             'dtype': precision, 'batch': B, 'hipgraph': bool(graph), 'workload': note, 'per_pass': pass_stats(pp, B),
             'whole_path': whole_path(ips, precision, S), 'telemetry': tel}
        if prof and prof['conv_ms'] > 0:
            r['roofline'] = conv_roofline(prof, precision, B, S, prof_dt)
            if prof_dt is None:
                r['roofline']['measured_in'] = 'one eager pass after the timed graph replays'
        return r
    except Exception as e:      # a sub-record must never take the headline line down
        return {'error': f'{type(e).__name__}: {e}'}


def dist_sub_records(cfg, sd, eng, dev, rank, world, S):
    """configs[3] and configs[4] at their own per-GPU workloads, measured by ALL ranks of a torch.distributed run (the headline
    above is configs[1] per GPU): bf16 B=64 + hipGraph sampling (no collective on the data path), and the data-parallel training
    step at B=32 per GPU (one all-reduce of the gradient arena per step).  Every rank enters every collective here whether or
    not its own measurement succeeded (a failure is reported as an error record, never as a hang); ranks start together behind
    a barrier and the time is the max over ranks."""
    import torch.distributed as dist
    from fastdiffsr_amd import parallel
    from fastdiffsr_amd.engine import Engine

    def measure(fn):
        torch.cuda.synchronize(dev)
        dist.barrier()
        try:
            dt, err = fn(), None
        except Exception as e:
            dt, err = float('inf'), f'{type(e).__name__}: {e}'
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()), err

    recs = {}
    steps, warmup = 6, 2      # as the single-GPU sub-records: >= 6 timed passes after >= 2 warm-up passes
    dt, err = measure(lambda: run_config(eng, dev, 'bf16', 64, S, steps, warmup, True, 'engine', rank=rank, want_profile=False)[0])
    if dt != float('inf'):
        ips = world * 64 * steps / dt
        recs['bf16_b64_graph'] = {'value': ips, 'unit': 'images/s', 'n_gpus': world, 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': warmup,
                                  'dtype': 'bf16', 'batch_per_gpu': 64, 'global_batch': 64 * world, 'hipgraph': True,
                                  'workload': 'configs[3]: bf16, batch 64 per GPU sharded over the ranks, 20-step loop as a hipGraph, weights broadcast once'}
    else:
        recs['bf16_b64_graph'] = {'error': err or 'failed on another rank'}
    dt, err = measure(lambda: run_config(eng, dev, 'f16', 64, S, steps, warmup, True, 'engine', rank=rank, want_profile=False)[0])
    if dt != float('inf'):
        recs['f16_b64_graph'] = {'value': world * 64 * steps / dt, 'unit': 'images/s', 'n_gpus': world, 'ms_per_step': 1e3 * dt / steps, 'steps': steps,
                                 'warmup': warmup, 'dtype': 'f16', 'batch_per_gpu': 64, 'global_batch': 64 * world, 'hipgraph': True,
                                 'workload': 'the configs[3] workload in the f16 mode (one f16 MFMA per product, f16 activations): batch 64 per GPU'}
    else:
        recs['f16_b64_graph'] = {'error': err or 'failed on another rank'}

    def train():
        e2 = Engine(cfg)
        e2.load_state_dict(sd)
        return run_train(e2, dev, 32, S, steps, warmup, rank=rank, allreduce=parallel.allreduce_grads, precision='f16x3')
    dt, err = measure(train)
    torch.cuda.empty_cache()
    if dt != float('inf'):
        ips = world * 32 * steps / dt
        recs['train_step_b32'] = {'value': ips, 'unit': 'images/s (one optimisation step per batch)', 'n_gpus': world, 'ms_per_step': 1e3 * dt / steps,
                                  'steps': steps, 'warmup': warmup, 'dtype': 'f16x3', 'batch_per_gpu': 32, 'global_batch': 32 * world,
                                  'workload': 'configs[4]: x8 32->256 shapes, batch 32 per GPU, q_sample + L1(sum)/(b*c*h*w) + backward + Adam, Dropout(0.2) '
                                              'live, every convolution f16x3; data parallel = one RCCL all-reduce of the 91.6 MB gradient arena per step'}
    else:
        recs['train_step_b32'] = {'error': err or 'failed on another rank'}
    return recs


def gather_rank_times(dt, dev):
    """Every rank's own time for the same timed region -> (max over ranks, list per rank).  The max is the job's time
    (barrier-bracketed region); the list exposes a straggler."""
    import torch.distributed as dist
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    all_t = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(all_t, t)
    ts = [float(x.item()) for x in all_t]
    return max(ts), ts


def per_rank_stats(ts, units_per_rank):
    v = [units_per_rank / t for t in ts]
    return {'min': min(v), 'max': max(v), 'mean': sum(v) / len(v), 'unit': 'images/s per rank', 'ranks': len(v)}


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        return so.getsockname()[1]


def rank_env(base, rank, world, port):
    """Environment of rank `rank` of a one-node job: what torch.distributed.run would have set."""
    env = dict(base)
    env.update({'RANK': str(rank), 'LOCAL_RANK': str(rank), 'WORLD_SIZE': str(world), 'LOCAL_WORLD_SIZE': str(world),
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port), 'FDSR_BENCH_CHILD': '1'})
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: RCCL needs it on this driver
    return env


def visible_gpus():
    """How many GPUs this process may use, WITHOUT bringing up HIP in it (the launcher parent must stay GPU-free: its children
    are the ranks).  HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set, else the KFD topology: nodes with SIMDs."""
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(',') if x.strip() != ''])
    n, root = 0, '/sys/class/kfd/kfd/topology/nodes'
    try:
        for d in os.listdir(root):
            try:
                for line in open(os.path.join(root, d, 'properties')):
                    if line.startswith('simd_count') and int(line.split()[1]) > 0:
                        n += 1
            except OSError:
                pass
    except OSError:
        return None       # no KFD view: let rank k fail on set_device instead of guessing
    return n


def launch_ranks(n, argv, visible=None, popen=None, script=None, poll_s=0.2, grace_s=5.0):
    """`python bench.py --gpus N` without torchrun: this process -- which never touches the GPU (visible_gpus() reads sysfs) --
    starts N fresh child processes, one per GPU and each in a session of its own, relays rank 0's single JSON line on stdout
    (every other rank's stdout goes to stderr) and watches ALL of them: the first rank that exits non-zero, or a SIGTERM / SIGINT
    to this process, takes the whole job down (terminate, then kill after `grace_s`), so no rank is left waiting in a rendezvous
    and none outlives the launcher.  Returns the first failing exit code.  Never an exec."""
    import signal
    import subprocess
    if visible is None:
        visible = visible_gpus()
    if visible is not None and visible < n:
        print(f'bench.py: --gpus {n} but only {visible} device(s) visible', file=sys.stderr)
        return 2
    popen = popen or subprocess.Popen
    port = free_port()
    procs = []
    stop = {'sig': None}

    def on_signal(signum, frame):
        stop['sig'] = signum

    old = {}
    for sg in (signal.SIGTERM, signal.SIGINT):
        try:
            old[sg] = signal.signal(sg, on_signal)
        except ValueError:          # not the main thread (tests): no handlers, polling still tears down on failure
            pass

    def teardown():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)        # start_new_session: pid == pgid, the rank and anything it started
                except (ProcessLookupError, PermissionError, AttributeError):
                    p.terminate()
        t_end = time.time() + grace_s
        for p in procs:
            while p.poll() is None and time.time() < t_end:
                time.sleep(0.05)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError, AttributeError):
                    p.kill()
                p.wait()

    rc = 0
    try:
        for r in range(n):
            procs.append(popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=rank_env(os.environ, r, n, port),
                               stdout=(None if r == 0 else sys.stderr), start_new_session=True))
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                break
            if stop['sig'] is not None:
                rc = 128 + int(stop['sig'])
                break
            if all(c == 0 for c in codes):
                break
            time.sleep(poll_s)
    finally:
        teardown()
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=16, help='images per GPU per step')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-sub-records', action='store_true',
                    help='skip the exact_f32 / bf16_b64_graph / b1_graph legs that ride in the same JSON line at N=1')
    ap.add_argument('--no-profile', action='store_true', help='skip the per-conv HIP-event timing')
    ap.add_argument('--no-facade', action='store_true', help='skip the file-based val_e2e / train_facade_b32 records (768 PNG writes + a 256-image val pass)')
    ap.add_argument('--precision', default='f16x3', choices=['f32', 'f16x3', 'bf16', 'f16'],
                    help='conv arithmetic: exact fp32 MFMA, fp32-grade split-f16 MFMA, or bf16')
    ap.add_argument('--graph', action='store_true', help='replay the 20-step loop as a hipGraph')
    ap.add_argument('--train', action='store_true',
                    help='time optimisation steps (forward + loss + backward + Adam, BASELINE configs[4]) instead of sampling; '
                         'the metric is then images/s through one training step')
    ap.add_argument('--debug-option', action='append', default=[], metavar='NAME=VALUE',
                    help='launcher A/B option (include/fdsr.h: fdsr_debug_option), e.g. strip=0; repeatable; recorded in the line')
    ap.add_argument('--noise', default='engine', choices=['engine', 'tensor'],
                    help="engine: N(0,1) drawn inside the timed loop by the engine (Philox), as the reference draws "
                         "randn_like per step; tensor: a pre-drawn [T,B,3,H,W] tensor resident in HBM (the parity-run form)")
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and (args.gpus > 1 or os.environ.get('FDSR_BENCH_FORCE_DIST') == '1'):
        # not under torch.distributed.run: start the ranks ourselves (before anything here initialises the GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    # FDSR_BENCH_FORCE_DIST=1: take the RCCL path (init, weight broadcast, barrier, max-reduce) with one rank too
    distributed = world > 1 or os.environ.get('FDSR_BENCH_FORCE_DIST') == '1'
    if args.gpus != world:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')

    import torch.distributed as dist
    from fastdiffsr_amd.arch import UNetConfig, FASTDIFFSR_UNET, FASTDIFFSR_SCHEDULE_VAL
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.schedule import schedule_buffers, sampling_scalars
    from fastdiffsr_amd.synth import synth_state_dict, state_dict_sha256
    from fastdiffsr_amd import parallel, _lib

    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    global K32_BITS, STRIP_BITS, KNOCKOUT
    for item in args.debug_option:
        k, v = item.split('=')
        _lib.debug_option(k, int(v))
        if k == 'k32':
            K32_BITS = int(v)
        if k == 'strip':
            STRIP_BITS = int(v)
        if k == 'knockout' and int(v):
            KNOCKOUT = True
    cfg = UNetConfig(**FASTDIFFSR_UNET)
    # weights: rank 0 builds the random-init UNet, ONE RCCL broadcast replicates it
    sd = synth_state_dict(cfg, 0) if rank == 0 else None
    weights_sha_rank0 = state_dict_sha256(sd) if rank == 0 else None
    if distributed:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        with _StdoutToStderr():
            t_i0 = time.perf_counter()
            dist.init_process_group('nccl', device_id=dev)
            dist.barrier()                       # communicator up on every rank (RCCL builds it lazily on the first collective)
            torch.cuda.synchronize(dev)
            t_b0 = time.perf_counter()
            sd = parallel.broadcast_state_dict(sd, cfg, src=0, device=dev)
            dist.barrier()
            torch.cuda.synchronize(dev)
            t_init, t_bcast = t_b0 - t_i0, time.perf_counter() - t_b0   # (outside the timed region; reported so that a scaling miss can be attributed)
    eng = Engine(cfg)
    eng.load_state_dict(sd)
    bufs, sp = schedule_buffers(FASTDIFFSR_SCHEDULE_VAL)
    eng.set_schedule(sampling_scalars(bufs, sp))

    B, S = args.batch, args.size

    def sync():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.train:
        Bt = args.batch if args.batch != 16 else 32
        hook = parallel.allreduce_grads if distributed else None
        dt = run_train(eng, dev, Bt, S, args.steps, args.warmup, rank=rank, sync=sync, allreduce=hook, precision=args.precision)
        rank_ts = [dt]
        if distributed:
            dt, rank_ts = gather_rank_times(dt, dev)
        if rank == 0:
            ips = world * Bt * args.steps / dt
            tf = ips / world * 3 * FLOPS_PER_IMAGE / 20 / 1e12
            print(json.dumps({
                'metric': '256x256 images/sec through one optimisation step (forward + loss + backward + Adam)', 'value': ips,
                'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
                'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
                'config': {'workload': f'BASELINE configs[4] training step: x8 32->256 shapes, batch={Bt}/GPU, 256x256, q_sample + L1(sum)/(b*c*h*w), '
                                       'Dropout(0.2) live, ' + ('exact fp32' if args.precision == 'f32' else 'every convolution f16x3 (fp32-grade)') + '; data parallel = one all-reduce of the 91.6 MB gradient arena per step',
                           'batch_per_gpu': Bt, 'global_batch': Bt * world, 'parallelism': f'dp{world}'},
                'per_rank': per_rank_stats(rank_ts, Bt * args.steps),
                'world_size_reported_by_backend': dist.get_world_size() if distributed else 1,
                'algorithmic_tflops_per_gpu': tf, 'roofline': train_roofline(ips / world, args.precision)}), flush=True)
        if distributed:
            dist.destroy_process_group()
        return

    # independent per-GPU batch (weak scaling): rank r samples its own B images
    telemetry = {}
    dt, out, prof, _ = run_config(eng, dev, args.precision, B, S, args.steps, args.warmup, args.graph, args.noise,
                                  rank=rank, sync=sync, want_profile=not args.no_profile, telemetry=telemetry)
    rank_ts = [dt]
    if distributed:
        dt, rank_ts = gather_rank_times(dt, dev)

    dist_recs = None
    if distributed and not args.no_sub_records:
        dist_recs = dist_sub_records(cfg, sd, eng, dev, rank, world, S)
        eng.set_precision(args.precision)

    if rank == 0:
        ips = world * B * args.steps / dt
        version = _lib.load().fdsr_version().decode()
        res = {
            'metric': '256x256 SR images/sec at T=20', 'value': ips, 'unit': 'images/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.precision, 'data': 'synthetic',
            'config': {'workload': f'x4 64->256, batch={B}/GPU, T=20, random-init FastDiffSR UNet (inner 64, mults 1-2-4-4), '
                                   f'{S}x{S}, conv arithmetic {args.precision} (BASELINE configs[1])',
                       'batch_per_gpu': B, 'global_batch': B * world, 'timesteps': 20, 'noise': ('drawn in the loop by the engine (Philox4x32-10)' if args.noise == 'engine'
                                                  else 'pre-drawn tensor in HBM'), 'hipgraph': bool(args.graph),
                       'parallelism': f'dp{world} (independent batches, weights broadcast once)'},
            'per_rank': per_rank_stats(rank_ts, B * args.steps),
            'world_size_reported_by_backend': dist.get_world_size() if distributed else 1,
            'whole_path_tflops': ips / world * FLOPS_PER_IMAGE / 1e12,
            'whole_path': whole_path(ips / world, args.precision),
            'library': {'version': version.split(' FDSR_SRC_SHA256=')[0], 'source_sha256': version.rsplit('=', 1)[-1]},
            # mean shader clock / package power of rank 0's GPU over the timed region (hwmon; nominal 2400 MHz): is the regime power-bound?
            'telemetry': telemetry,
        }
        res['library']['kernel_forms'] = {'k32': bit_names(K32, K32_BITS), 'strip': bit_names(STRIP, STRIP_BITS)}
        if args.debug_option:
            res['debug_options'] = list(args.debug_option)
        if distributed:   # every rank loaded what rank 0 broadcast: its hash is checked against rank 0's own arrays
            res['weights'] = {'sha256_rank0_source': weights_sha_rank0, 'sha256_after_broadcast': state_dict_sha256(sd),
                              'broadcast_bytes': int(sum(np.asarray(v).nbytes for v in sd.values())),
                              'broadcast_seconds_rank0': t_bcast, 'communicator_init_seconds_rank0': t_init}
        if prof and prof['conv_ms'] > 0:
            res['roofline'] = conv_roofline(prof, args.precision, B, S, dt)
        if dist_recs is not None:
            res['sub_records'] = dist_recs
        if world == 1 and not distributed and not args.no_sub_records:
            # the other arithmetic modes and batch regimes of BASELINE.json, driver-visible in the same line
            # every record: >= 6 timed passes after >= 2 warm-up passes (more when the driver asks for more: steps/3, warmup/2),
            # each pass synchronised on its own so that min / median / max ride in the record
            ks, kw = max(6, args.steps // 3), max(2, args.warmup // 2)
            res['sub_records'] = {
                'exact_f32': sub_record(eng, dev, 'exact_f32', 'f32', 16, S, ks, kw, False,
                                        'configs[1] in exact fp32 MFMA arithmetic (v_mfma_f32_32x32x2_f32), B=16'),
                'bf16_b64_graph': sub_record(eng, dev, 'bf16_b64_graph', 'bf16', 64, S, ks, kw, True,
                                             'configs[2]: B=64, bf16 activations + bf16 MFMA, 20-step loop replayed as a hipGraph'),
                'f16_b64_graph': sub_record(eng, dev, 'f16_b64_graph', 'f16', 64, S, ks, kw, True,
                                            'configs[2] workload in the f16 mode (round 6): one f16 MFMA per product + f16 activations -- the bf16 mode\'s '
                                            'bytes and MFMA rate with 11 mantissa bits (PSNR against the oracle 74.6 dB where bf16 gives 57.6)'),
                'b1_graph': sub_record(eng, dev, 'b1_graph', 'f16x3', 1, S, max(20, args.steps), max(5, args.warmup), True,
                                       'configs[0] regime (the reference val loop is B=1, sr_mfe.py:279-284): latency per image, hipGraph'),
            }
            res['sub_records']['b1_graph_f16'] = sub_record(eng, dev, 'b1_graph_f16', 'f16', 1, S, max(20, args.steps), max(5, args.warmup), True,
                                                            'the B=1 val regime in the f16 mode (PSNR-grade: 74.6 dB from the oracle image): latency per image, hipGraph')
            # the reference's second driver: infer.py runs B=1 at 512 x 512 (infer.py:59-79, :112-113; config/sr_fastdiffsr_infer_x4.json)
            res['sub_records']['infer_512_b1'] = sub_record(eng, dev, 'infer_512_b1', 'f16x3', 1, 512, max(10, args.steps // 2), max(3, args.warmup // 2), True,
                                                            "infer.py's regime: B=1 at 512 x 512 (128 -> 512), f16x3, hipGraph; ms_per_step = latency per image")
            res['sub_records']['train_step_b32'] = train_record(cfg, sd, dev, 32, S, ks, kw, 'f16x3')
            res['sub_records']['train_step_b32_f32'] = train_record(cfg, sd, dev, 32, S, ks, kw, 'f32')
            eng.set_precision(args.precision)
            if not args.no_facade:
                try:        # optional records must never take the measured line down (no Pillow, /tmp full ...)
                    res['sub_records'].update(facade_records(cfg, sd, dev, S, ips))
                except Exception as e:
                    err = {'error': f'{type(e).__name__}: {e}'}
                    res['sub_records'].update({'val_e2e': err, 'val_e2e_f16': err, 'train_facade_b32': err})
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'], ref = cpu_baseline(cfg, sd)
            eng.set_precision(args.precision)
            res['parity_check'] = parity_check(eng, dev, ref)
        # calls of this process (headline, sub-records, facade records, parity check) that tripped the f16x3 range guard and were
        # re-run on the exact-fp32 kernels (Engine.on_saturation = 'f32'): 0 = every f16x3 number above is an f16x3 number
        res['saturation_fallbacks'] = int(Engine.saturation_fallbacks)
        print(json.dumps(res), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
