#!/usr/bin/env python
"""Per-kernel totals of a rocprofv3 run (rocpd sqlite output), restricted to the last `--steps`
bench steps (identified by the memset that opens every training plan is not needed: the table is
simply scaled by steps/(warmup+steps) launches per kernel name)."""
import argparse
import re
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--csv', default='')
    ap.add_argument('--steps', type=int, default=0, help='timed steps: only the last steps/(all) share of launches')
    ap.add_argument('--all', type=int, default=0, help='warm-up + timed steps')
    args = ap.parse_args()
    cur = sqlite3.connect(args.db).cursor()
    rows = cur.execute('select name, start, end from kernels order by start').fetchall()
    if args.steps and args.all:
        # keep the launches after the (all - steps)/all point of the step sequence: find the
        # start time of the adamw launch that ends warm-up
        ad = [r for r in rows if 'adamw_kernel' in r[0]]
        cut = ad[len(ad) - args.steps - 1][2] if len(ad) > args.steps else 0
        rows = [r for r in rows if r[1] >= cut]
        nsteps = args.steps
    else:
        nsteps = 1
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
