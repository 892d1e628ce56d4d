#!/usr/bin/env python
"""Micro-benchmark of vp_gemm_bf16 on the ViTPose-B GEMM shapes (64 images = 12288 tokens):
forward (k-contig x k-contig), dgrad (k-contig x k-major), wgrad (k-major x k-major + split-K).
Random operands (zero-filled data flatters MFMA clocks)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitpose_amd import _lib  # noqa: E402
from vitpose_amd.engine import _slab_elems, _split_k  # noqa: E402
from vitpose_amd.plan import Plan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tokens', type=int, default=12288)
    ap.add_argument('--embed', type=int, default=768)
    ap.add_argument('--reps', type=int, default=50)
    ap.add_argument('--only', default='')
    ap.add_argument('--tile-mode', type=int, default=0, help='0 autotuned, 1 = 128x128 tiles, 2 = 256x256, 3 = 192x192, 4 = 192x128 with two workgroups per CU, 7 = 192x192 K-split, '
                    '5 = persistent 192x192, 6 = 192x128 with one workgroup per CU')
    ap.add_argument('--group-blocks', type=int, default=4)
    ap.add_argument('--wsplit', type=int, default=0, help='override the wgrad K split')
    ap.add_argument('--pad', type=int, default=0, help='pad every bf16 operand row by this many elements (row stride no '
                    'longer a multiple of the memory-channel interleave)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    M, E = args.tokens, args.embed
    g = torch.Generator().manual_seed(0)

    P = args.pad

    def rb(r, c):
        return (torch.rand(r, c + P, generator=g) * 2 - 1).to(dev).bfloat16()
    cases = []
    for name, N, K in (('qkv', 3 * E, E), ('proj', E, E), ('fc1', 4 * E, E), ('fc2', E, 4 * E)):
        x, w = rb(M, K), rb(N, K)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        cases.append((f'fwd_{name}   M{M} N{N} K{K}', 2.0 * M * N * K,
                      lambda p, x=x, w=w, out=out, N=N, K=K: p.gemm(x, w, out, M, N, K, ldx=K + P, ldy=K + P, ldo=N,
                                                                       tile_mode=args.tile_mode)))
        dy = rb(M, N)
        dx = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
        cases.append((f'dgrad_{name} M{M} N{K} K{N}', 2.0 * M * N * K,
                      lambda p, dy=dy, w=w, dx=dx, N=N, K=K: p.gemm(dy, w, dx, M, K, N, ldx=N + P, ldy=K + P, ldo=K,
                                                                    y_kmajor=True, tile_mode=args.tile_mode)))
        if name == 'fc1':
            u2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            bias = torch.zeros(N, device=dev)
            cases.append((f'fwd_fc1+GELU M{M} N{N} K{K}', 2.0 * M * N * K,
                          lambda p, x=x, w=w, out=out, u2=u2, bias=bias, N=N, K=K: p.gemm(
                              x, w, out, M, N, K, ldx=K + P, ldy=K + P, ldo=N, bias=bias, out2=u2,
                              epilogue=_lib.VP_EPI_GELU, tile_mode=args.tile_mode)))
        if name == 'fc2':
            aux = rb(M, K)
            cases.append((f'dgrad_fc2+DGELU M{M} N{K} K{N}', 2.0 * M * N * K,
                          lambda p, dy=dy, w=w, dx=dx, aux=aux, N=N, K=K: p.gemm(
                              dy, w, dx, M, K, N, ldx=N + P, ldy=K + P, ldo=K, y_kmajor=True, epilogue=_lib.VP_EPI_DGELU,
                              aux=aux, ldaux=K + P, tile_mode=args.tile_mode)))
            r32, o32 = torch.zeros(M, N, device=dev), torch.empty(M, N, device=dev)
            bias2 = torch.zeros(N, device=dev)
            cases.append((f'fwd_fc2+RESID M{M} N{N} K{K}', 2.0 * M * N * K,
                          lambda p, x=x, w=w, o32=o32, r32=r32, bias2=bias2, N=N, K=K: p.gemm(
                              x, w, o32, M, N, K, ldx=K + P, ldy=K + P, ldo=N, bias=bias2, resid=r32,
                              epilogue=_lib.VP_EPI_RESID_F32, tile_mode=args.tile_mode)))
        dw = torch.empty(N, K, device=dev)
        slab = torch.empty(_slab_elems(N * K, M), device=dev)
        sk = args.wsplit or _split_k(N, K, M)
        cases.append((f'wgrad_{name} M{N} N{K} K{M} split{sk}', 2.0 * M * N * K,
                      lambda p, dy=dy, x=x, dw=dw, slab=slab, N=N, K=K, sk=sk: p.wgrad(
                          dy, x, dw, N, K, M, ldx=N + P, ldy=K + P, split_k=sk, slab=slab, tile_mode=args.tile_mode)))
    # grouped: the four Linear weight gradients of `--group-blocks` Blocks in one launch, no K split
    probs = []
    for blk in range(args.group_blocks):
        for N, K in ((3 * E, E), (E, E), (4 * E, E), (E, 4 * E)):
            probs.append((rb(M, N), rb(M, K), torch.empty(N, K, device=dev), N, K, N + P, K + P))
    gflops = sum(2.0 * M * p[3] * p[4] for p in probs)
    cases.append((f'wgrad_group {args.group_blocks} blocks ({len(probs)} problems)', gflops,
                  lambda p: p.wgrad_group(probs, M, tile_mode=args.tile_mode if args.tile_mode in (2, 3, 7) else 0)))
    for name, flops, build in cases:
        if args.only and args.only not in name:
            continue
        p = Plan()
        build(p)
        try:
            p.run()
        except Exception as exc:  # noqa: BLE001  (a schedule that does not cover this layout)
            print(f'{name:45s} unsupported ({exc})')
            continue
        # warm up for ~40 ms of back-to-back launches: the clock of an idle GPU needs that long to settle, and a
        # candidate measured right after host-side set-up (idle GPU) otherwise reads 10-15 % slow
        t_w = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t_w[0].record()
        for _ in range(20):
            p.run()
        t_w[1].record()
        torch.cuda.synchronize()
        for _ in range(int(40.0 / max(t_w[0].elapsed_time(t_w[1]) / 20, 1e-3))):
            p.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            p.run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.reps
        print(f'{name:44s} {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
