#!/usr/bin/env python
"""Effective bandwidth of the HBM-bound kernels against a plain torch copy of the same bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitpose_amd.plan import Plan  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def report(name, us, nbytes):
    print(f'{name:46s} {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s', flush=True)


def main():
    dev = torch.device('cuda:0')
    M, E = 12288, 768
    x = torch.randn(M, E, device=dev)
    y = torch.empty_like(x)
    report('torch copy f32 [12288,768]', timeit(lambda: y.copy_(x)), 2 * x.numel() * 4)
    big = torch.randn(196608, 256, device=dev).bfloat16()
    big2 = torch.empty_like(big)
    report('torch copy bf16 [196608,256]', timeit(lambda: big2.copy_(big)), 2 * big.numel() * 2)
    g, b = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    yb = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    p = Plan()
    p.layernorm_fwd(x, g, b, yb, None, mean, rstd, M, E)
    report('layernorm_fwd', timeit(p.run), M * E * 6)
    dy = torch.randn(M, E, device=dev).bfloat16()
    dres, dx, dxb = torch.randn(M, E, device=dev), torch.empty(M, E, device=dev), torch.empty_like(dy)
    dg, db, dc = (torch.zeros(E, device=dev) for _ in range(3))
    ws = torch.empty(3 * E * 512, device=dev)
    p = Plan()
    p.layernorm_bwd(dy, x, mean, rstd, g, dres, dx, dxb, None, 192, dg, db, dc, M, E, workspace=ws)
    report('layernorm_bwd', timeit(p.run), M * E * (2 + 4 + 4 + 4 + 2))
    sc, sh = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    p = Plan()
    p.bn_apply_relu(big, big2, sc, sh, big.shape[0], 256)
    report('bn_apply_relu [196608,256]', timeit(p.run), 2 * big.numel() * 2)
    sums = torch.zeros(512, device=dev)
    p = Plan()
    p.bn_stats(big, sums, big.shape[0], 256)
    report('bn_stats [196608,256]', timeit(p.run), big.numel() * 2)
    p = Plan()
    p.bn_bwd(big, big2, sc, sc, sc, dg[:256], db[:256], torch.empty_like(big), big.shape[0], 256)
    report('bn_bwd (reduce + apply) [196608,256]', timeit(p.run), 5 * big.numel() * 2)
    a = torch.randn(M, 3072, device=dev).bfloat16()
    o = torch.zeros(3072, device=dev)
    p = Plan()
    p.colsum(a, o, M, 3072, 3072)
    report('colsum_bf16 [12288,3072]', timeit(p.run), a.numel() * 2)
    qkv = torch.randn(M, 3 * E, device=dev).bfloat16()
    ao = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(64 * 12, 192, device=dev)
    p = Plan()
    p.attention_fwd(qkv, ao, lse, 64, 12, 64, 192, 0.125)
    report('attention_fwd B64 H12 d64', timeit(p.run), M * E * 8)
    dqkv = torch.empty_like(qkv)
    p = Plan()
    p.attention_bwd(qkv, ao, ao, lse, dqkv, 64, 12, 64, 192, 0.125)
    report('attention_bwd B64 H12 d64', timeit(p.run), M * E * (6 + 2 + 2 + 6))
    n = 90 * 1000 * 1000
    pf, gf = torch.randn(n, device=dev), torch.randn(n, device=dev)
    out = torch.zeros(1, device=dev)
    p = Plan()
    p.sumsq(gf, n, out)
    report('sumsq 90M', timeit(p.run), n * 4)


if __name__ == '__main__':
    main()
