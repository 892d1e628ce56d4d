"""Summarises the `clkprobe` lines a -DVP_CLOCK_PROBE build of the library prints: per kernel and K-tile count, the core
clock the K loop ran at (shader-clock ticks per tick of the 100 MHz constant-rate counter) and the loop's duration."""
import collections
import re
import sys

rows = collections.defaultdict(list)
for line in open(sys.argv[1], errors='replace'):
    m = re.search(r'clkprobe (\S+) nk (\d+) shader (\d+) wall (\d+)', line)
    if m and int(m.group(4)) > 0:
        rows[(m.group(1), int(m.group(2)))].append((int(m.group(3)), int(m.group(4))))
for (name, nk), v in sorted(rows.items()):
    v = v[len(v) // 2:]                      # second half of the launches: clocks settled
    mhz = sorted(100.0 * s / w for s, w in v)
    us = sorted(w / 100.0 for s, w in v)
    print(f'{name:10s} nk {nk:4d} launches {len(v):5d}  core clock median {mhz[len(mhz) // 2]:6.0f} MHz (min {mhz[0]:.0f} max {mhz[-1]:.0f})'
          f'  K loop median {us[len(us) // 2]:7.1f} us')
