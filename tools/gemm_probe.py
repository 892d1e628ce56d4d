#!/usr/bin/env python
"""Time vp_gemm_bf16 over a sweep of K at fixed M, N (separates the per-tile fixed cost from the
per-K-step cost of a tile schedule)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitpose_amd import _lib  # noqa: E402
from vitpose_amd.plan import Plan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--M', type=int, default=12288)
    ap.add_argument('--N', type=int, default=768)
    ap.add_argument('--Ks', default='64,256,768,1536,3072')
    ap.add_argument('--tile-mode', type=int, default=2)
    ap.add_argument('--epi', type=int, default=0)
    ap.add_argument('--reps', type=int, default=30)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    M, N = args.M, args.N
    for K in [int(k) for k in args.Ks.split(',')]:
        x = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).bfloat16()
        w = (torch.rand(N, K, generator=g) * 2 - 1).to(dev).bfloat16()
        f32 = args.epi in (2, 5)
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        kw = {}
        if args.epi == 2:
            kw = dict(resid=torch.randn(M, N, device=dev), bias=torch.randn(N, device=dev))
        if args.epi in (0, 1):
            kw = dict(bias=torch.randn(N, device=dev))
        p = Plan()
        p.use_tail_split = False
        p.gemm(x, w, out, M, N, K, ldx=K, ldy=K, ldo=N, epilogue=args.epi, tile_mode=args.tile_mode, **kw)
        for _ in range(3):
            p.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            p.run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        print(f'mode {args.tile_mode} epi {args.epi} M{M} N{N} K{K:5d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s',
              flush=True)


if __name__ == '__main__':
    main()
