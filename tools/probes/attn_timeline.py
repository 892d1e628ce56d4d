"""Time stamps inside the attention backward, workgroup by workgroup (-DVP_CLOCK_PROBE=2 build of the library:
VITPOSE_HIP_LIB=tools/probes/libvitpose_clk2.so): start, images in LDS (the kernel's one barrier), phase A / phase B waves
done with their stores acknowledged.  Cold operands (1.5 GB of unrelated traffic before the launch) unless ATTN_HOT=1;
ATTN_B / ATTN_H change the batch and the head count (768 / 1: the same work with adjacent head slices)."""
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitpose_amd import _lib  # noqa: E402
from vitpose_amd.plan import Plan  # noqa: E402

dev = torch.device('cuda:0')
B, H = int(os.environ.get('ATTN_B', '64')), int(os.environ.get('ATTN_H', '12'))   # 768 / 1: the same work, head slices adjacent
d, T = 64, 192
E = H * d
g = torch.Generator().manual_seed(0)
big = torch.empty(768 << 20, device=dev, dtype=torch.uint8)
qkv = (torch.rand(B * T, 3 * E, generator=g) - 0.5).to(dev).bfloat16()
out = torch.empty(B * T, E, device=dev, dtype=torch.bfloat16)
dout = (torch.rand(B * T, E, generator=g) - 0.5).to(dev).bfloat16()
lse = torch.empty(B * H * T, device=dev)
dqkv = torch.empty_like(qkv)
pf, pb = Plan(), Plan()
pf.attention_fwd(qkv, out, lse, B, H, d, T, d ** -0.5)
pb.attention_bwd(qkv, out, dout, lse, dqkv, B, H, d, T, d ** -0.5)
pf.run()
HOT = os.environ.get('ATTN_HOT', '0') == '1'      # 1: no unrelated traffic in front of the launch
for _ in range(3):
    if not HOT:
        big.add_(1)
        big.add_(1)
    pb.run()
torch.cuda.synchronize()
L = _lib.lib()
n = B * H
buf = (ctypes.c_longlong * (n * 8))()
L.vp_probe_dump_attn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.vp_probe_dump_attn(buf, n * 8) == 0
rows = [[buf[i * 8 + j] for j in range(4)] for i in range(n)]
t0 = min(r[0] for r in rows)
tu = 0.01


def dist(name, v):
    v = sorted(v)
    print(f'{name:52s} min {v[0] * tu:6.2f}  median {v[len(v) // 2] * tu:6.2f}  p90 {v[int(.9 * len(v))] * tu:6.2f}  max {v[-1] * tu:6.2f} us')


# record: 0 start, 1 the four images and delta / lse in LDS (the barrier), 2 / 3 = wave 0 (phase A) / wave 6 (phase B) done
print(f'{n} workgroups, launch span {(max(max(r[2], r[3]) for r in rows) - t0) * tu:.1f} us')
dist('start -> images in LDS (barrier)', [r[1] - r[0] for r in rows])
dist('phase A waves (two query tiles each, + stores)', [r[2] - r[1] for r in rows])
dist('phase B waves (two key tiles each, + stores)', [r[3] - r[1] for r in rows])
dist('whole workgroup', [max(r[2], r[3]) - r[0] for r in rows])
