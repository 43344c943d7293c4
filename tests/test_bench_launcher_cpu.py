"""bench.py's launch contract on the CPU: `python bench.py --gpus N` outside torch.distributed.run starts the N ranks
itself (before any GPU call), the ranks rendezvous over gloo on 127.0.0.1, rank 0's ONE JSON line is relayed with
n_gpus = the ranks that actually joined, and a failing rank fails the launcher.  The `stub` workload does host
arithmetic only (no kernels): this tests the plumbing, not a measurement."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(argv, env=None, timeout=300):
    e = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, capture_output=True, text=True,
                          timeout=timeout, env=e)


def test_gpus_2_starts_two_ranks_and_reports_them():
    r = _run(['--gpus', '2', '--workload', 'stub', '--steps', '4', '--warmup', '1'])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout                      # exactly one result line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['warmup'] == 1 and d['scaling'] == 'weak'
    assert d['value'] > 0 and abs(d['value'] - 25600 * 2 * 4 / (d['ms_per_step'] * 4e-3)) < 1e-3 * d['value']
    assert 'cpu_baseline' not in d                        # N = 1 only


def test_single_rank_stub_line():
    r = _run(['--workload', 'stub', '--steps', '2', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert d['n_gpus'] == 1


def test_world_size_must_match_gpus():
    r = _run(['--gpus', '2', '--workload', 'stub'], env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and '--gpus 2' in r.stderr


def test_failing_rank_fails_the_launcher():
    r = _run(['--gpus', '2', '--workload', 'stub', '--steps', '2'], env={'BENCH_STUB_FAIL_RANK': '1'})
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
