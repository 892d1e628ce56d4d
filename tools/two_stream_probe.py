#!/usr/bin/env python
"""Does running a batch as two half-batches on two HIP streams fill the epilogue bubbles of the one-round GEMMs?
Times 12 forward Blocks (ViTPose-B widths, training-style outputs) on 64 images in one stream against the same work
as two independent 32-image sequences on two streams, enqueued kernel by kernel in alternation.
python tools/two_stream_probe.py [images] [embed] [heads]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitpose_amd import _lib          # noqa: E402
from vitpose_amd.plan import Plan     # noqa: E402

T = 192


def record(B, E, heads, dev, depth=12, tail_split=True):
    M, Hd, d = B * T, 4 * E, E // heads
    g = torch.Generator().manual_seed(B)

    def rb(*s):
        return ((torch.rand(*s, generator=g) * 2 - 1) * 0.05).to(dev).bfloat16()
    p = Plan()
    p.use_tail_split = tail_split
    w = dict(qkv=rb(3 * E, E), proj=rb(E, E), fc1=rb(Hd, E), fc2=rb(E, Hd))
    bias = {k: torch.zeros(v.shape[0], device=dev) for k, v in w.items()}
    gamma, beta = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    x = torch.randn(M, E, generator=g).to(dev)
    xm, xo = torch.empty_like(x), torch.empty_like(x)
    xn, ao = torch.empty(M, E, device=dev, dtype=torch.bfloat16), torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    qkv = torch.empty(M, 3 * E, device=dev, dtype=torch.bfloat16)
    h, u = torch.empty(M, Hd, device=dev, dtype=torch.bfloat16), torch.empty(M, Hd, device=dev, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    lse = torch.empty(B * heads, T, device=dev)
    a, b = x, xo
    for _ in range(depth):
        p.layernorm_fwd(a, gamma, beta, xn, None, mean, rstd, M, E)
        p.gemm(xn, w['qkv'], qkv, M, 3 * E, E, ldx=E, ldy=E, ldo=3 * E, bias=bias['qkv'])
        p.attention_fwd(qkv, ao, lse, B, heads, d, T, d ** -0.5)
        p.gemm(ao, w['proj'], xm, M, E, E, ldx=E, ldy=E, ldo=E, epilogue=_lib.VP_EPI_RESID_F32, bias=bias['proj'],
               resid=a)
        p.layernorm_fwd(xm, gamma, beta, xn, None, mean, rstd, M, E)
        p.gemm(xn, w['fc1'], h, M, Hd, E, ldx=E, ldy=E, ldo=Hd, epilogue=_lib.VP_EPI_GELU, out2=u, bias=bias['fc1'])
        p.gemm(h, w['fc2'], b, M, E, Hd, ldx=Hd, ldy=Hd, ldo=E, epilogue=_lib.VP_EPI_RESID_F32, bias=bias['fc2'],
               resid=xm)
        a, b = b, a
    return p


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    heads = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    dev = torch.device('cuda:0')
    full = record(B, E, heads, dev)
    full_nt = record(B, E, heads, dev, tail_split=False)
    # the tail-split workspace is one per device: two streams must not share it
    h1, h2 = record(B // 2, E, heads, dev, tail_split=False), record(B // 2, E, heads, dev, tail_split=False)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    main_s = torch.cuda.current_stream()

    def two_streams():
        ev = torch.cuda.Event()
        ev.record(main_s)
        s1.wait_event(ev)
        s2.wait_event(ev)
        for i in range(len(h1)):
            h1.run(s1.cuda_stream, i, i + 1)
            h2.run(s2.cuda_stream, i, i + 1)
        for s in (s1, s2):
            e = torch.cuda.Event()
            e.record(s)
            main_s.wait_event(e)

    def halves_one_stream():
        h1.run()
        h2.run()
    for rep in range(2):
        print(f'{B} images, one stream:                 {timed(full.run):9.1f} us', flush=True)
        print(f'{B} images, one stream, no tail split:  {timed(full_nt.run):9.1f} us', flush=True)
        print(f'2 x {B // 2} images back to back, one stream: {timed(halves_one_stream):9.1f} us', flush=True)
        print(f'2 x {B // 2} images on two streams:          {timed(two_streams):9.1f} us', flush=True)


if __name__ == '__main__':
    main()
