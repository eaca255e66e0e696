"""bench.py --gpus N without torchrun: the parent starts N fresh rank processes itself (no GPU needed to check that)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_rank_env_is_what_torchrun_would_set():
    import bench
    env = bench.rank_env({'PATH': '/bin', 'RANK': '9'}, 2, 4, 23456)
    assert env['RANK'] == '2' and env['LOCAL_RANK'] == '2' and env['WORLD_SIZE'] == '4'
    assert env['MASTER_ADDR'] == '127.0.0.1' and env['MASTER_PORT'] == '23456'
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and env['PATH'] == '/bin'


def test_launcher_refuses_when_fewer_devices_than_ranks():
    import bench
    started = []
    rc = bench.launch_ranks(4, ['--gpus', '4'], visible=1, popen=lambda *a, **k: started.append(a))
    assert rc != 0 and not started


def test_launcher_starts_n_children_relays_rank0_and_returns_worst_rc(tmp_path):
    """Real child processes (a stand-in script): each sees its own RANK / LOCAL_RANK and the shared WORLD_SIZE and rendezvous
    address; only rank 0's stdout reaches the parent's stdout; a failing rank fails the launch."""
    child = tmp_path / 'child.py'
    child.write_text(textwrap.dedent('''
        import json, os, sys
        keys = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')
        print(json.dumps({k: os.environ.get(k) for k in keys} | {'argv': sys.argv[1:]}), flush=True)
        sys.exit(int(os.environ.get('FAIL_RANK', -1)) == int(os.environ['RANK']) and 7 or 0)
    '''))
    drv = tmp_path / 'drv.py'
    drv.write_text(textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        sys.exit(bench.launch_ranks(3, ['--gpus', '3', '--steps', '2'], visible=3, script={str(child)!r}))
    '''))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    r = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out) == 1 and out[0]['RANK'] == '0' and out[0]['WORLD_SIZE'] == '3'      # rank 0 only on stdout
    assert out[0]['argv'] == ['--gpus', '3', '--steps', '2'] and out[0]['MASTER_ADDR'] == '127.0.0.1'
    others = [json.loads(ln) for ln in r.stderr.splitlines() if ln.strip().startswith('{')]
    assert sorted(o['RANK'] for o in others) == ['1', '2']
    assert {o['MASTER_PORT'] for o in others} == {out[0]['MASTER_PORT']}
    env['FAIL_RANK'] = '2'
    r = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 7


def test_main_takes_the_launcher_when_world_size_is_unset(monkeypatch):
    import bench
    seen = {}
    monkeypatch.setattr(bench, 'launch_ranks', lambda n, argv: seen.update(n=n, argv=list(argv)) or 0)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '8', '--steps', '3', '--warmup', '1'])
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {'n': 8, 'argv': ['--gpus', '8', '--steps', '3', '--warmup', '1']}
