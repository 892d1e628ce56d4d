"""Timeline of ONE grouped weight-gradient launch (the four Linear layers of `--blocks` ViTPose-B Blocks, 64 images), tile
by tile, from a -DVP_CLOCK_PROBE=2 build of the library (every workgroup records where it ran and the constant-rate
counter at its start / end of its K loop / after its last store; `vp_probe_dump` reads the records back):
    VITPOSE_HIP_LIB=tools/probes/libvitpose_clk2.so python tools/probes/tile_timeline.py --tile-mode 7
Prints the launch span, the distribution of the K-loop and post-K-loop (exchange + epilogue + store acknowledge) times,
the gap a CU leaves between two tiles, tiles per CU, and how long CUs sit idle at the end of the launch."""
import argparse
import collections
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vitpose_amd import _lib  # noqa: E402
from vitpose_amd.plan import Plan  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--tile-mode', type=int, default=7)
ap.add_argument('--blocks', type=int, default=4)
ap.add_argument('--tokens', type=int, default=12288)
ap.add_argument('--embed', type=int, default=768)
ap.add_argument('--warm', type=int, default=20)
ap.add_argument('--tick-us', type=float, default=0.01, help='period of the constant-rate counter (100 MHz)')
args = ap.parse_args()
dev = torch.device('cuda:0')
M, E = args.tokens, args.embed
g = torch.Generator().manual_seed(0)


def rb(r, c):
    return (torch.rand(r, c, generator=g) * 2 - 1).to(dev).bfloat16()


probs = []
for blk in range(args.blocks):
    for N, K in ((3 * E, E), (E, E), (4 * E, E), (E, 4 * E)):
        probs.append((rb(M, N), rb(M, K), torch.empty(N, K, device=dev), N, K, N, K))
plan = Plan()
plan.wgrad_group(probs, M, tile_mode=args.tile_mode)
for _ in range(args.warm):
    plan.run()
torch.cuda.synchronize()
L = _lib.lib()
ntiles = sum(((p[3] + 191) // 192) * ((p[4] + 191) // 192) for p in probs)
buf = (ctypes.c_longlong * (ntiles * 8))()
L.vp_probe_dump.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = L.vp_probe_dump(buf, ntiles * 8)
assert rc == 0, rc
rows = [tuple(buf[i * 8 + j] for j in range(7)) for i in range(ntiles)]
t0 = min(r[4] for r in rows)
tu = args.tick_us


def q(v, f):
    v = sorted(v)
    return v[min(len(v) - 1, int(f * len(v)))]


def dist(name, v):
    print(f'{name:34s} min {min(v) * tu:7.2f}  p10 {q(v, .1) * tu:7.2f}  median {q(v, .5) * tu:7.2f}  p90 {q(v, .9) * tu:7.2f}  max {max(v) * tu:7.2f} us')


span = max(r[6] for r in rows) - t0
print(f'{len(rows)} tiles, launch span (first start .. last store acknowledged) {span * tu:.1f} us')
dist('start -> end of K loop', [r[5] - r[4] for r in rows])
dist('end of K loop -> last store done', [r[6] - r[5] for r in rows])
dist('whole tile', [r[6] - r[4] for r in rows])
cus = collections.defaultdict(list)
for r in rows:
    # hardware id: CU within the shader array, shader array, shader engine; plus the XCC
    cus[(r[2], (r[3] >> 8) & 0xf, (r[3] >> 12) & 0x1, (r[3] >> 13) & 0x7)].append(r)
print(f'{len(cus)} CUs; tiles per CU: {dict(collections.Counter(len(v) for v in cus.values()))}')
gaps, idle_end, first = [], [], []
for v in cus.values():
    v.sort(key=lambda r: r[4])
    first.append(v[0][4] - t0)
    gaps += [b[4] - a[6] for a, b in zip(v, v[1:])]
    idle_end.append(t0 + span - v[-1][6])
dist('first tile start after launch start', first)
if gaps:
    dist('gap between two tiles on a CU', gaps)
dist('CU idle at the end of the launch', idle_end)
busy = sum(r[6] - r[4] for r in rows)
print(f'CU-time inside tiles {busy * tu / len(cus):.1f} us per CU of {span * tu:.1f} us ({busy / len(cus) / span:.3f})')
kbusy = sum(r[5] - r[4] for r in rows)
print(f'CU-time inside K loops {kbusy * tu / len(cus):.1f} us per CU ({kbusy / len(cus) / span:.3f} of the span)')
byx = collections.defaultdict(list)
for r in rows:
    byx[r[2]].append(r[5] - r[4])
print('K loop median by XCC:', {k: round(q(v, .5) * tu, 1) for k, v in sorted(byx.items())})
