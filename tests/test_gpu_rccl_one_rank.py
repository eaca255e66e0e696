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
    env = dict(os.environ)
    env.update({'FDSR_BENCH_FORCE_DIST': '1', 'RANK': '0', 'LOCAL_RANK': '0', 'WORLD_SIZE': '1',
                'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(_free_port()), 'HSA_ENABLE_IPC_MODE_LEGACY': '0'})
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '1', '--warmup', '1',
                        '--no-cpu-baseline', '--no-sub-records'], cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                    # RCCL's banner must not leak onto stdout
    res = json.loads(lines[0])
    assert res['n_gpus'] == 1 and res['scaling'] == 'weak' and res['unit'] == 'images/s'
    assert res['value'] > 0 and res['value'] == res['value']          # finite (bench asserts isfinite(out) itself)
    w = res['weights']
    assert w['sha256_after_broadcast'] == w['sha256_rank0_source']    # what RCCL delivered is what rank 0 built
    assert w['broadcast_bytes'] > 90e6                                # the packed fp32 state, one message
    assert 'parallelism' in res['config'] and res['config']['parallelism'].startswith('dp1')


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
