#!/usr/bin/env python
"""Experiment: does running two half-batch training plans on two HIP streams beat one full-batch plan?
(overlap of one stream's memory-bound kernels / epilogues with the other's MFMA main loops)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vitpose_amd.pose_engine import PoseEngine  # noqa: E402


def make(B, dev, seed):
    m = bench.MODELS['B']
    vc = dict(img_size=(256, 192), patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4,
              qkv_bias=True, ratio=1, last_norm=True, drop_path_rate=0.3)
    hc = dict(in_channels=768, out_channels=17, num_deconv_layers=2, num_deconv_filters=(256, 256),
              num_deconv_kernels=(4, 4), extra=dict(final_conv_kernel=1), upsample=0)
    eng = PoseEngine(vc, hc, dev)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for n, s in eng.named_shapes():
        sd[n] = torch.randn(s, generator=g) * 0.02 if len(s) > 1 else (torch.ones(s) if 'norm' in n and n.endswith('weight') else torch.zeros(s))
    eng.load_state_dict(sd)
    img = torch.randn(B, 3, 256, 192, generator=g).to(dev)
    target = torch.rand(B, 17, 64, 48, generator=g).to(dev)
    tw = torch.ones(B, 17, 1).to(dev)
    return eng, (img, target, tw)


def main():
    dev = torch.device('cuda:0')
    steps = 10
    e64, b64 = make(64, dev, 0)
    for _ in range(3):
        e64.forward_backward(*b64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        e64.forward_backward(*b64)
    torch.cuda.synchronize()
    t64 = (time.perf_counter() - t0) / steps
    print(f'one plan, 64 images: {t64 * 1e3:.2f} ms / fwd+bwd', flush=True)
    ea, ba = make(32, dev, 1)
    eb, bb = make(32, dev, 2)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(sa):
            ea.forward_backward(*ba)
        with torch.cuda.stream(sb):
            eb.forward_backward(*bb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        with torch.cuda.stream(sa):
            ea.forward_backward(*ba)
        with torch.cuda.stream(sb):
            eb.forward_backward(*bb)
    torch.cuda.synchronize()
    t2 = (time.perf_counter() - t0) / steps
    print(f'two plans x 32 images on two streams: {t2 * 1e3:.2f} ms / fwd+bwd of 64 images', flush=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ea.forward_backward(*ba)
        eb.forward_backward(*bb)
    torch.cuda.synchronize()
    t3 = (time.perf_counter() - t0) / steps
    print(f'two plans x 32 images on ONE stream: {t3 * 1e3:.2f} ms', flush=True)


if __name__ == '__main__':
    main()
