#!/usr/bin/env python
"""Attention forward / backward alone, cold operands (1.5 GB of unrelated traffic before every launch), median of 15."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitpose_amd.plan import Plan  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    big = torch.empty(768 << 20, device=dev, dtype=torch.uint8)
    for name, B, H, d in (('B/64', 64, 12, 64), ('H/128', 128, 16, 80), ('S/64', 64, 12, 32), ('L/32', 32, 16, 64)):
        E, T = H * d, 192
        g = torch.Generator().manual_seed(0)
        qkv = (torch.rand(B * T, 3 * E, generator=g) - 0.5).to(dev).bfloat16()
        out = torch.empty(B * T, E, device=dev, dtype=torch.bfloat16)
        dout = (torch.rand(B * T, E, generator=g) - 0.5).to(dev).bfloat16()
        lse = torch.empty(B * H * T, device=dev)
        dqkv = torch.empty_like(qkv)
        pf, pb = Plan(), Plan()
        pf.attention_fwd(qkv, out, lse, B, H, d, T, d ** -0.5)
        pb.attention_bwd(qkv, out, dout, lse, dqkv, B, H, d, T, d ** -0.5)
        res = []
        for plan in (pf, pb):
            ts = []
            for _ in range(15):
                big.add_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan.run()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res.append(ts[len(ts) // 2])
        print(f'{name:6s} head_dim {d}: forward {res[0]:7.1f} us   backward {res[1]:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
