#!/usr/bin/env python
"""vp_layernorm_fwd: is a row's result independent of how many rows the call has (i.e. of which wave / register buffer
the row lands in)?  Rows 0-383 of a 12288-row call against a 384-row call on the same data, bit for bit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitpose_amd.plan import Plan  # noqa: E402

dev = torch.device('cuda:0')
for E in (768, 1024, 1280, 384):
    M = 12288
    x = torch.randn(M, E, device=dev) * 3 + 0.5
    gamma, beta = torch.rand(E, device=dev) + 0.5, torch.randn(E, device=dev)
    outs = []
    for rows in (M, 384, 2, 1000):
        y = torch.empty(rows, E, device=dev, dtype=torch.bfloat16)
        y32 = torch.empty(rows, E, device=dev)
        mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
        p = Plan()
        p.layernorm_fwd(x[:rows].contiguous(), gamma, beta, y, y32, mean, rstd, rows, E)
        p.run()
        torch.cuda.synchronize()
        outs.append((rows, y, y32, mean, rstd))
    ref = outs[0]
    for rows, y, y32, mean, rstd in outs[1:]:
        print(E, rows, 'bf16 equal', torch.equal(y, ref[1][:rows]), 'f32 equal', torch.equal(y32, ref[2][:rows]),
              'mean equal', torch.equal(mean, ref[3][:rows]), 'rstd equal', torch.equal(rstd, ref[4][:rows]),
              'max f32 diff', (y32 - ref[2][:rows]).abs().max().item())
