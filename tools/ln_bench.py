#!/usr/bin/env python
"""Times vp_layernorm_fwd (and vp_layernorm_bwd) at the ViTPose-B training shape (12288 x 768) and the ViTPose-H
inference shape (24576 x 1280): `reps` launches in a row over a ring of input / output buffers larger than the
Infinity Cache (every launch reads and writes cold lines, as inside a step), one event pair around the lot."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitpose_amd.plan import Plan  # noqa: E402

dev = torch.device('cuda:0')
for M, E in ((12288, 768), (24576, 1280), (6144, 1024)):
    ring = max(4, (600 << 20) // (M * E * 4))
    xs = [torch.randn(M, E, device=dev) for _ in range(ring)]
    ys = [torch.empty(M, E, device=dev, dtype=torch.bfloat16) for _ in range(ring)]
    gamma, beta = torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    p = Plan()
    for x, y in zip(xs, ys):
        p.layernorm_fwd(x, gamma, beta, y, None, mean, rstd, M, E)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        p.run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / ring)
    ts.sort()
    ref = torch.nn.functional.layer_norm(xs[-1], (E,), gamma, beta, 1e-6)
    err = (ys[-1].float() - ref).abs().max().item()
    nbytes = M * E * 6 + M * 8
    print(f'ln_fwd {M} x {E}: median {ts[len(ts) // 2]:.2f} us per launch (min {ts[0]:.2f}) = '
          f'{nbytes / ts[len(ts) // 2] / 1e6:.2f} TB/s; max abs err vs torch {err:.3e}')
    # backward: dy (bf16), x, residual gradient in, dx + bf16 copy out: (2 + 4 + 4) E in, (4 + 2) E out per row
    rb = max(4, (600 << 20) // (M * E * 16))
    dys = [ys[i % len(ys)] for i in range(rb)]
    xin = [xs[i % len(xs)] for i in range(rb)]
    dres = [torch.randn(M, E, device=dev) for _ in range(rb)]
    dxs = [torch.empty(M, E, device=dev) for _ in range(rb)]
    dxb = [torch.empty(M, E, device=dev, dtype=torch.bfloat16) for _ in range(rb)]
    wsp = torch.empty(3 * E * 1024, device=dev)
    dg, db = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    p = Plan()
    for i in range(rb):
        p.layernorm_bwd(dys[i], xin[i], mean, rstd, gamma, dres[i], dxs[i], dxb[i], None, 0, dg, db, None, M, E, workspace=wsp)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        p.run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / rb)
    ts.sort()
    nbytes = M * E * 16
    print(f'ln_bwd {M} x {E}: median {ts[len(ts) // 2]:.2f} us per launch incl. its finalize launch (min {ts[0]:.2f}) = '
          f'{nbytes / ts[len(ts) // 2] / 1e6:.2f} TB/s')
    del xs, ys, dres, dxs, dxb
