#!/usr/bin/env python
"""Is the eval forward of ViTPose-B independent of the batch an image sits in?  Heat maps of images 0-1 at batch 64
against batch 2 (bit-identical when every tile schedule accumulates in the same order), and where the backbone's
intermediate buffers first differ."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import vitpose_oracle as O  # noqa: E402
from vitpose_amd.pose_engine import PoseEngine  # noqa: E402

dev = torch.device('cuda:0')
vcfg, hc = O.vit_cfg('B'), O.classic_head_cfg(768)
bb, hd = O.synth_vit_weights(vcfg, seed=60), O.synth_head_weights(hc, seed=61)
eng = PoseEngine(vcfg, hc, dev)
sd = {'backbone.' + k: v for k, v in bb.items()}
sd.update({'keypoint_head.' + k: v for k, v in hd.items()})
eng.load_state_dict(sd)
img, _, _ = O.synth_batch(64, seed=62)
img = img.to(dev)
full = eng.infer(img)
bufs64 = {k: v.clone() for k, v in eng.ws_infer.bufs.items() if v.dim() == 2}
small = eng.infer(img[:2])
print('heat maps batch 64 vs 2: max abs', (full[:2] - small).abs().max().item(), 'equal', torch.equal(full[:2], small))
for k, v in eng.ws_infer.bufs.items():
    if k in bufs64 and v.dim() == 2 and v.shape[1] == bufs64[k].shape[1] and v.shape[0] <= bufs64[k].shape[0]:
        a, b = bufs64[k][:v.shape[0]].float(), v.float()
        if not torch.equal(a, b):
            print(f'  {k}: {tuple(v.shape)} differs, max abs {(a - b).abs().max().item():.3e}, '
                  f'{(a != b).float().mean().item() * 100:.3f} % of elements')
from vitpose_amd import plan  # noqa: E402
for k, v in sorted(plan._TUNE_CACHE.items(), key=str):
    print('  pick', k[:9], '->', v)
