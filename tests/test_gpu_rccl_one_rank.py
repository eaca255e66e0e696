"""The multi-GPU leg of bench.py on ONE GPU: RCCL communicator, the single weight broadcast, barrier-bracketed
timing and the max-reduce, with world size 1 (FDSR_BENCH_FORCE_DIST=1).  A fresh child process runs it -- nothing
is re-exec'ed from a process that has touched the GPU."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_bench_rccl_path_with_one_rank():
    # no RANK / WORLD_SIZE here: bench.py's own launcher (the path `--gpus N` takes without torchrun) starts the rank process
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env.update({'FDSR_BENCH_FORCE_DIST': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1',
                        '--no-cpu-baseline'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                    # RCCL's banner must not leak onto stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 1 and res['scaling'] == 'weak' and res['unit'] == 'images/s'
    assert res['world_size_reported_by_backend'] == 1 and res['per_rank']['ranks'] == 1
    assert abs(res['per_rank']['min'] - res['value']) < 1e-6 * res['value']
    assert res['value'] > 0 and res['value'] == res['value']          # finite (bench asserts isfinite(out) itself)
    w = res['weights']
    assert w['sha256_after_broadcast'] == w['sha256_rank0_source']    # what RCCL delivered is what rank 0 built
    assert w['broadcast_bytes'] > 90e6                                # the packed fp32 state, one message
    assert 0 < w['broadcast_seconds_rank0'] < 60 and w['communicator_init_seconds_rank0'] > 0
    assert res['per_rank']['min'] <= res['per_rank']['max'] and res['per_rank']['unit'].startswith('images/s')
    assert 'parallelism' in res['config'] and res['config']['parallelism'].startswith('dp1')
    # under torch.distributed every rank also measures configs[3] (bf16 B=64/GPU, hipGraph) and configs[4] (data-parallel training
    # step, B=32/GPU, gradient all-reduce through RCCL): the multi-GPU lines of the driver's scaling run carry them
    sub = res['sub_records']
    assert set(sub) == {'bf16_b64_graph', 'f16_b64_graph', 'train_step_b32'}
    for k, v in sub.items():
        assert 'error' not in v, (k, v)
        assert v['value'] > 0 and v['n_gpus'] == 1
    assert sub['bf16_b64_graph']['global_batch'] == 64 and sub['train_step_b32']['global_batch'] == 32


@pytest.mark.timeout(900)
def test_bench_default_line_is_complete_and_parity_clean():
    """The driver's command shape (N=1, all legs): one JSON line carrying the headline, every sub-record without an
    error, the CPU baseline, and a parity check that holds AFTER the training sub-records ran (they own their engine)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1'], cwd=ROOT,
                       capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res['metric'].startswith('256x256 SR images/sec') and res['n_gpus'] == 1 and res['dtype'] == 'f16x3'
    assert set(res['sub_records']) >= {'exact_f32', 'bf16_b64_graph', 'b1_graph', 'train_step_b32', 'train_step_b32_f32'}
    for k, v in res['sub_records'].items():
        assert 'error' not in v, (k, v)
        assert v['value'] > 0
    for k in ('exact_f32', 'bf16_b64_graph', 'b1_graph'):
        assert 0 < res['sub_records'][k]['roofline']['frac'] < 1
    assert res['roofline']['bound'] == 'mfma' and 0 < res['roofline']['frac'] < 1
    cb = res['cpu_baseline']
    assert cb['kind'] == 'port' and cb['cores'] >= 1 and cb['value'] > 0 and cb['gflops'] > 0 and cb['cpu_model']
    pc = res['parity_check']
    assert pc['max_abs_diff_x_t'] <= 1e-3
    if 'psnr_delta_db' in pc:
        assert abs(pc['psnr_delta_db']) <= 0.01 and pc['max_abs_diff_image'] <= 1e-3
    assert len(res['library']['source_sha256']) == 64


@pytest.mark.timeout(600)
def test_bench_train_rccl_path_with_one_rank():
    """The data-parallel training leg with one rank: the gradient arena wrapped zero-copy as a torch tensor and all-reduced
    through RCCL between backward and Adam (parallel.allreduce_grads), one JSON line."""
    # no RANK / WORLD_SIZE here: bench.py's own launcher (the path `--gpus N` takes without torchrun) starts the rank process
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    env.update({'FDSR_BENCH_FORCE_DIST': '1', 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--train', '--batch', '4', '--steps', '2',
                        '--warmup', '1'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    res = json.loads(lines[0])
    assert res['n_gpus'] == 1 and res['value'] > 0 and res['config']['batch_per_gpu'] == 4


def test_grad_arena_is_the_engines_memory():
    """Engine.grad_arena() shares the engine's gradient memory (no copy): writes through the tensor are what get_grad reads."""
    import numpy as np
    import torch
    from fastdiffsr_amd.arch import UNetConfig
    from fastdiffsr_amd.engine import Engine
    from fastdiffsr_amd.synth import synth_state_dict
    cfg = UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32, channel_mults=(1, 2), attn_res=(16,), res_blocks=1,
                     dropout=0.0, image_size=32)
    eng = Engine(cfg)
    eng.load_state_dict(synth_state_dict(cfg, 1))
    eng.set_precision('f32')
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 6, 32, 32, generator=g).cuda()
    eng.train_grads(x, torch.tensor([0.5, 0.7]).cuda(), torch.randn(2, 3, 32, 32, generator=g).cuda(), 'l1', 1e-3)
    arena = eng.grad_arena()
    assert arena.is_cuda and arena.dtype == torch.float32 and arena.numel() > 1e5
    before = eng.get_grad('downs.0.weight').copy()
    arena.mul_(2.0)                                   # what an all-reduce(sum) over two identical ranks would do
    torch.cuda.synchronize()
    assert np.array_equal(eng.get_grad('downs.0.weight'), 2.0 * before)
