#!/usr/bin/env python
"""Micro-benchmark of the optimizer step's two launches on a flat buffer of `--params` f32 elements cut into the
optimizer's chunks: the deterministic sum of squares behind the clip factor (vp_sumsq_chunks_det) and the fused AdamW
update with its bf16 weight copy (vp_adamw_step).  Prints microseconds and TB/s (4 B / element read for the norm; 16 B
read + 14 B written per element for the update)."""
import argparse
import os
import sys
from ctypes import c_void_p

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitpose_amd import _lib  # noqa: E402
from vitpose_amd.optim import CHUNK  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--params', type=int, default=90_000_000)
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--chunk', type=int, default=CHUNK, help='elements per chunk (= per AdamW workgroup)')
args = ap.parse_args()
dev = torch.device('cuda:0')
CHUNK = args.chunk
n = args.params // CHUNK * CHUNK
L = _lib.lib()
p, g = torch.randn(n, device=dev), torch.randn(n, device=dev) * 1e-3
m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
half = torch.zeros(n, device=dev, dtype=torch.bfloat16)
starts = torch.arange(0, n, CHUNK, dtype=torch.int64, device=dev)
lens = torch.full((starts.numel(),), CHUNK, dtype=torch.int32, device=dev)
scale, wd = torch.ones(starts.numel(), device=dev), torch.full((starts.numel(),), 0.1, device=dev)
gsq, part = torch.zeros(1, device=dev), torch.zeros(2048, device=dev)
s = c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: c_void_p(t.data_ptr())


def norm():
    assert L.vp_sumsq_chunks_det(P(g), P(starts), P(lens), starts.numel(), P(gsq), P(part), part.numel(), s) == 0


def update(step):
    assert L.vp_adamw_step(P(p), P(g), P(m), P(v), P(half), P(starts), P(lens), P(scale), P(wd), starts.numel(), 5e-4, 0.9,
                           0.999, 1e-8, step, P(gsq), 1.0, 1.0, s) == 0


for name, fn, bytes_per in (('sum of squares', lambda i: norm(), 4), ('AdamW update', update, 30)):
    for i in range(3):
        fn(i + 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.reps):
        fn(i + 4)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.reps
    print(f'{name:16s} {n / 1e6:.1f} M elements  {us:8.1f} us  {n * bytes_per / us / 1e6:.2f} TB/s')
ref = float((g.double() ** 2).sum())
print(f'norm^2 {float(gsq):.6e} (f64 reference {ref:.6e})')
