#!/usr/bin/env python
"""Where the reference-default iteration (model.train_step: per-iteration PCK accuracy + logged scalars read on the
host) loses its time against the bare step: host time of get_accuracy and _parse_losses with an idle GPU."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
import vitpose_amd.models  # noqa: E402,F401
from vitpose_amd.builder import build_posenet  # noqa: E402

dev = torch.device('cuda:0')
m = bench.MODELS['B']
cfg = bench.model_cfg(m)
cfg['train_cfg'] = dict(skip_accuracy=False)
model = build_posenet(cfg).to(dev).train()
B = 64
img, target, tw = torch.randn(B, 3, 256, 192, device=dev), torch.rand(B, 17, 64, 48, device=dev), torch.ones(B, 17, 1, device=dev)
opt = model.build_optimizer(dict(type='AdamW', lr=5e-4, betas=(0.9, 0.999), weight_decay=0.1,
                                 constructor='LayerDecayOptimizerConstructor',
                                 paramwise_cfg=dict(num_layers=12, layer_decay_rate=0.75)), grad_clip=dict(max_norm=1., norm_type=2))
batch = dict(img=img, target=target, target_weight=tw, img_metas=None)
for _ in range(5):
    out = model.train_step(batch, None)
    out['loss'].backward()
    opt.step()
    opt.zero_grad()
torch.cuda.synchronize()
head = model.keypoint_head
heat = model._heat
for name, fn in (('get_accuracy', lambda: head.get_accuracy(heat, target, tw)),
                 ('_parse_losses', lambda: model._parse_losses(dict(heatmap_loss=out['loss'].detach(), acc_pose=0.5))),
                 ('target_weight.cpu()', lambda: tw.detach().cpu().numpy())):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    print(f'{name}: {(time.perf_counter() - t0) / 20 * 1e6:.0f} us per call with an idle GPU')
