#!/usr/bin/env python
"""Does operand residency in the Infinity Cache explain why a GEMM inside the training step is slower than the same GEMM
in a tight benchmark loop?  Times dgrad fc2 + GELU' (ViTPose-B, 64 images) (a) back to back (operands cache-hot), (b) after
1.5 GB of unrelated traffic (cold, as inside the step), (c) cold, but with the operands read once by a streaming kernel
just before the launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitpose_amd import _lib  # noqa: E402
from vitpose_amd.plan import Plan  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    M, E, Hd = 12288, 768, 3072
    g = torch.Generator().manual_seed(0)

    def rb(*s):
        return (torch.rand(*s, generator=g) * 2 - 1).to(dev).bfloat16()
    dy, w, aux = rb(M, E), rb(E, Hd), rb(M, Hd)
    du = torch.empty(M, Hd, device=dev, dtype=torch.bfloat16)
    x, w1 = rb(M, E), rb(Hd, E)
    h = torch.empty(M, Hd, device=dev, dtype=torch.bfloat16)
    u = torch.empty(M, Hd, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(Hd, device=dev)
    cases = {}
    p = Plan()
    p.gemm(dy, w, du, M, Hd, E, ldx=E, ldy=Hd, ldo=Hd, y_kmajor=True, epilogue=_lib.VP_EPI_DGELU, aux=aux, ldaux=Hd)
    cases['dgrad fc2 + GELU\''] = (p, [aux, dy])
    p2 = Plan()
    p2.gemm(x, w1, h, M, Hd, E, ldx=E, ldy=E, ldo=Hd, bias=bias, out2=u, epilogue=_lib.VP_EPI_GELU)
    cases['fc1 + GELU'] = (p2, [x])
    p3 = Plan()
    dxn = torch.empty(M, E, device=dev, dtype=torch.bfloat16)
    p3.gemm(du, w1, dxn, M, E, Hd, ldx=Hd, ldy=E, ldo=E, y_kmajor=True)
    cases['dgrad fc1'] = (p3, [du])
    big = torch.empty(768 << 20, device=dev, dtype=torch.uint8)
    sink = torch.zeros(1, device=dev)
    for name, (plan, operands) in cases.items():
        for _ in range(50):
            plan.run()
        torch.cuda.synchronize()

        def timed(before):
            ts = []
            for _ in range(12):
                before()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan.run()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            return ts[len(ts) // 2]

        def warm_clock():
            for _ in range(6):
                plan.run()

        def flush():
            warm_clock()
            big.add_(1)          # 768 MB read + 768 MB written: nothing of the operands is left in any cache

        def flush_touch():
            flush()
            for t in operands:
                sink.add_(t.view(torch.int16).sum(dtype=torch.int32).float())
        print(f'{name:20s} hot {timed(warm_clock):7.1f} us   cold {timed(flush):7.1f} us   cold + operands read once {timed(flush_touch):7.1f} us',
              flush=True)


if __name__ == '__main__':
    main()
