#!/usr/bin/env python
"""Times the fused AdamW step and the gradient-norm reduction over a ViTPose-B sized flat buffer for several
chunk sizes (elements per workgroup).   python tools/adamw_probe.py [chunk ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vitpose_amd import optim                                   # noqa: E402
from vitpose_amd.engine import ParamStore, vit_param_shapes     # noqa: E402


def main():
    chunks = [int(c) for c in sys.argv[1:]] or [1 << 14, 1 << 15, 1 << 16, 1 << 17]
    shapes = [('backbone.' + n, s) for n, s in vit_param_shapes(dict(embed_dim=768, depth=12, patch_size=16, img_size=(256, 192)))]
    shapes.append(('keypoint_head.w', (4199697,)))
    store = ParamStore(shapes, 'cuda:0')
    store.grad.normal_(0, 1e-3)
    store.data.normal_(0, 0.02)
    groups = optim.layer_decay_param_groups(shapes, 12, 0.75, 5e-4, 0.1)
    for rep in range(2):
        for c in chunks:
            optim.CHUNK = c
            opt = optim.FlatAdamW(store, groups, 5e-4, max_grad_norm=1.0)
            for _ in range(3):
                opt.step()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            ev[0].record()
            for _ in range(20):
                opt.step()
            ev[1].record()
            torch.cuda.synchronize()
            print(f'chunk {c}: {opt.nchunks} workgroups, {ev[0].elapsed_time(ev[1]) / 20 * 1e3:.1f} us per step '
                  f'(norm + AdamW, {store.numel / 1e6:.1f} M elements)', flush=True)
            del opt


if __name__ == '__main__':
    main()
