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


def test_launcher_tears_the_job_down_when_one_rank_dies_or_the_parent_is_signalled(tmp_path):
    """A rank that dies at start-up must not leave the others waiting in a rendezvous, and a SIGTERM to the launcher
    (`timeout 900 python bench.py --gpus 8`) must not orphan the ranks: every child is gone when launch_ranks returns."""
    import signal
    import time
    child = tmp_path / 'child.py'
    child.write_text(textwrap.dedent('''
        import os, sys, time
        open(os.path.join(os.environ['PID_DIR'], 'rank%s.pid' % os.environ['RANK']), 'w').write(str(os.getpid()))
        if os.environ.get('FAIL_RANK') == os.environ['RANK']:
            time.sleep(1.5)          # the others are up (and have written their pid files) by then
            sys.exit(5)
        time.sleep(600)              # "stuck in the rendezvous"
    '''))
    drv = tmp_path / 'drv.py'
    drv.write_text(textwrap.dedent(f'''
        import sys
        sys.path.insert(0, {ROOT!r})
        import bench
        sys.exit(bench.launch_ranks(3, [], visible=3, script={str(child)!r}, grace_s=2.0))
    '''))

    def alive(pid):
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            return False
        # a zombie of an already-reaped session leader does not count
        try:
            return open(f'/proc/{pid}/stat').read().split()[2] != 'Z'
        except OSError:
            return False

    def pids(d):
        return [int(open(os.path.join(d, f)).read()) for f in os.listdir(d)]

    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    d1 = tmp_path / 'p1'
    d1.mkdir()
    t0 = time.time()
    r = subprocess.run([sys.executable, str(drv)], capture_output=True, text=True, env=dict(env, PID_DIR=str(d1), FAIL_RANK='1'), timeout=120)
    assert r.returncode == 5 and time.time() - t0 < 60          # not the 600 s the healthy ranks would have slept
    assert len(pids(d1)) == 3 and not any(alive(p) for p in pids(d1))
    d2 = tmp_path / 'p2'
    d2.mkdir()
    p = subprocess.Popen([sys.executable, str(drv)], env=dict(env, PID_DIR=str(d2)))
    for _ in range(200):
        if len(os.listdir(d2)) == 3:
            break
        time.sleep(0.1)
    assert len(os.listdir(d2)) == 3
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) == 128 + signal.SIGTERM
    assert not any(alive(q) for q in pids(d2))
