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
