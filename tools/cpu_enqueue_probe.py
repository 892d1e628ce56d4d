#!/usr/bin/env python
"""How long the CPU needs to enqueue one training step (plan replay), against the GPU step time."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.dual_stream_probe import make  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    eng, batch = make(64, dev, 0)
    for _ in range(3):
        eng.forward_backward(*batch)
    torch.cuda.synchronize()
    plan, io = eng.plans[('train', 64)]
    print('calls per step:', len(plan))
    for trial in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        plan.run()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'enqueue {1e3 * (t1 - t0):.2f} ms, until GPU done {1e3 * (t2 - t0):.2f} ms', flush=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.run()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            plan.run()
    torch.cuda.synchronize()
    for trial in range(3):
        t0 = time.perf_counter()
        g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'graph replay: launch {1e3 * (t1 - t0):.2f} ms, until GPU done {1e3 * (t2 - t0):.2f} ms', flush=True)


if __name__ == '__main__':
    main()
