#!/usr/bin/env python
"""Times vp_heatmap_loss (loss + heat-map gradient + NCHW copy) at the ViTPose-B training shape, cold operands."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitpose_amd.plan import Plan  # noqa: E402

dev = torch.device('cuda:0')
B, K, HW, LD = 64, 17, 64 * 48, 32
g = torch.Generator().manual_seed(0)
heat = torch.randn(B * HW, LD, generator=g).to(dev)
target = torch.rand(B, K, HW, generator=g).to(dev)
tw = torch.ones(B, K).to(dev)
nchw = torch.empty(B, K, HW, device=dev)
loss = torch.zeros(1, device=dev)
dheat = torch.empty(B * HW, LD, device=dev, dtype=torch.bfloat16)
big = torch.empty(768 << 20, device=dev, dtype=torch.uint8)
p = Plan()
p.heatmap_loss(heat, LD, nchw, target, tw, loss, dheat, B, K, HW, 1.0)
ts = []
for _ in range(15):
    big.add_(1)
    loss.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    p.run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
ref = ((heat[:, :K].view(B, HW, K).permute(0, 2, 1) - target) ** 2).mean()
print(f'vp_heatmap_loss {B} x {K} x {HW}: median {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f}); loss {float(loss):.6f} (torch {float(ref):.6f})')
