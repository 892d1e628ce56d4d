#!/usr/bin/env python
"""Per-kernel totals of a rocprofv3 run (rocpd sqlite output) of bench.py, restricted to the TIMED steps.

Every step of every bench mode (training step, eval forward) launches the patch-embedding im2col kernel exactly once, as
its first kernel, so the i-th launch of that kernel marks the start of step i.  With --steps K (and --tail T = steps the
bench runs AFTER its timed region, e.g. the parity forward passes or the reference-default iterations) the table covers
the launches from the start of step (n - T - K) to the start of step (n - T): warm-up, the autotune probes of plan
recording and the one survey step are outside."""
import argparse
import re
import sqlite3

MARKER = 'patch_im2col'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--csv', default='')
    ap.add_argument('--steps', type=int, default=0, help='timed steps')
    ap.add_argument('--tail', type=int, default=0, help='steps the bench runs after the timed region')
    ap.add_argument('--all', type=int, default=0, help='(ignored; kept for old command lines)')
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    rows = cur.execute('select name, start, end from kernels order by start').fetchall()
    nsteps = 1
    if args.steps:
        marks = [r[1] for r in rows if MARKER in r[0]]
        if len(marks) < args.steps + args.tail:
            raise SystemExit(f'{len(marks)} steps in the trace, {args.steps} + {args.tail} asked for')
        first = marks[len(marks) - args.tail - args.steps]
        last = marks[len(marks) - args.tail] if args.tail else rows[-1][2] + 1
        rows = [r for r in rows if first <= r[1] < last]
        nsteps = args.steps
    agg = {}
    for n, s, e in rows:
        d = agg.setdefault(n, [0, 0, 1 << 62, 0])
        d[0] += 1
        d[1] += e - s
        d[2] = min(d[2], e - s)
        d[3] = max(d[3], e - s)
    tot = sum(d[1] for d in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f'{len(rows)} launches, kernel time {tot / 1e6 / nsteps:.3f} ms/step, wall span {span / 1e6 / nsteps:.3f} ms/step')
    out = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"']
    for n, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append(f'"{n}",{d[0]},{d[1]},{d[1] / d[0]:.1f},{100 * d[1] / tot:.2f},{d[2]},{d[3]}')
        short = re.sub(r'\(anonymous namespace\)::', '', n)[:88]
        if d[1] / tot > 0.004:
            print(f'{short:88s} {d[0] / nsteps:6.1f}/step {d[1] / 1e3 / nsteps:8.1f} us/step  avg {d[1] / d[0] / 1e3:7.1f} us {100 * d[1] / tot:5.2f}%')
    if args.csv:
        open(args.csv, 'w').write('\n'.join(out) + '\n')


if __name__ == '__main__':
    main()
