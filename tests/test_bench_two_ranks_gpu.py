"""bench.py launched the way the driver launches it for N > 1 (python -m torch.distributed.run, one process per
rank), rehearsed with two ranks on the one GPU the tests have: the ranks share the device and gloo carries the
collectives (VITPOSE_BENCH_SHARE_GPU / VITPOSE_BENCH_BACKEND; RCCL refuses two ranks on one device).  Checks the
whole N > 1 control flow: rendezvous, DDP through the model API from the autograd thread, barriers, max over
ranks, exactly one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_json_line():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VITPOSE_BENCH_BACKEND='gloo', VITPOSE_BENCH_SHARE_GPU='1', VITPOSE_BENCH_WATCHDOG='200')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3',
           '--warmup', '2', '--batch', '4', '--model', 'S']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-800:], out.stderr[-1500:])
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 2 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 8 and d['config']['parallelism'] == 'dp2'
    assert d['value'] > 0 and d['unit'] == 'images/sec' and 'roofline' in d and 'cpu_baseline' not in d
    assert d['loss'] == d['loss'] and d['loss'] > 0          # finite loss after the optimizer steps


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the form of the single-GPU command with another N): the
    script starts its ranks itself as fresh child processes, rank 0's JSON line comes through, the exit status is the
    ranks'.  The line says what the process group itself reports (rank count, a sum of ones) and every rank's median
    step time."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    env.update(VITPOSE_BENCH_BACKEND='gloo', VITPOSE_BENCH_SHARE_GPU='1', VITPOSE_BENCH_WATCHDOG='200')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2', '--batch', '4',
           '--model', 'S']
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert out.returncode == 0 and len(lines) == 1, (out.returncode, out.stdout[-800:], out.stderr[-1500:])
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 8 and d['value'] > 0
    r = d['ranks']
    assert r['world_size'] == 2 and r['allreduce_of_ones'] == 2 and len(r['step_ms_median_per_rank']) == 2
    assert r['step_ms_rank_min'] <= r['step_ms_rank_max'] and 'NCCL_MAX_NCHANNELS' in r['env']
    ref = d['reference_default_iteration']
    assert ref['images_per_sec'] > 0 and {'heatmap_loss', 'acc_pose', 'loss'} <= set(ref['log_vars'])
    # a launcher that started another number of ranks than --gpus names is an error, not a silent re-interpretation
    bad = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1'],
                         env=dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0'), capture_output=True, text=True,
                         timeout=120, cwd=ROOT)
    assert bad.returncode != 0 and 'rank' in (bad.stderr + bad.stdout)
