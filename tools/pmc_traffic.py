#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; one counter
per run) of `bench.py`:

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_FETCH_SIZE -o pmc -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_WRITE_SIZE -o pmc -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline
  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv \
                              gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv profiles/pmc_traffic.json

Units and corrections: both counters are reported in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads (16 B per lane, plain loads and LDS-DMA alike) at 64 bytes, i.e. it
reports one half of the bytes: fetch bytes = 2 x FETCH_SIZE x 1024.  WRITE_SIZE is exact for
16-byte-per-lane stores: write bytes = WRITE_SIZE x 1024.  Other access widths are uncalibrated.
The launcher vp_gemm_bf16 picks one of four tile-schedule kernels per shape (vp_wgrad_grouped_bf16 runs
gemm192_kernel / gemm192k_kernel / gemm256_kernel<true, true, 5, ...>); rows are keyed both by
the device kernel name and by the launcher instantiation vp_gemm_bf16<x_kmajor, y_kmajor, epilogue>."""
import csv
import json
import re
import sys


def load(path):
    agg = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            d = agg.setdefault(r['Kernel_Name'], [0, 0.0])
            d[0] += 1
            d[1] += float(r['Counter_Value'])
    return agg


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_csv')
    ap.add_argument('write_csv')
    ap.add_argument('out')
    ap.add_argument('--workload', default='', help='key prefix "<mode>_<model><batch>", e.g. train_B64: bench.py looks '
                                                   'its dominant kernel up under "<workload>:<kernel>"')
    ap.add_argument('--state', default='', help='tree state the passes were taken on (profiles/README.md), kept as _state')
    ap.add_argument('--grouped-gflop', type=float, default=0.0,
                    help='algorithmic GFLOP of one grouped weight-gradient launch of this run (bench.py prints it): a '
                         'record taken on launches of another size is not used')
    ap.add_argument('--merge', action='store_true', help='add to the records already in `out` instead of replacing them')
    args = ap.parse_args()
    fetch, write, out = load(args.fetch_csv), load(args.write_csv), args.out
    pre = args.workload + ':' if args.workload else ''
    res = {}
    groups = {}
    for name in sorted(set(fetch) | set(write)):
        nf, kf = fetch.get(name, [0, 0.0])
        nw, kw = write.get(name, [0, 0.0])
        n = max(nf, nw)
        if n == 0:
            continue
        fb = 2.0 * kf * 1024 / max(nf, 1)
        wb = kw * 1024 / max(nw, 1)
        short = re.sub(r'\(anonymous namespace\)::', '', name)
        short = re.sub(r'\(.*$', '', short).replace('void ', '')
        res[short] = dict(launches=n, fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb),
                          hbm_bytes_per_launch=round(fb + wb))
        m = re.match(r'gemm(?:192k?|192p|256)?_kernel<(true|false), (true|false), (\d+)(?:, (?:true|false|\d+))*>', short)
        key = None
        if m:
            key = f'vp_gemm_bf16<{m.group(1)}, {m.group(2)}, {m.group(3)}>'
            if m.group(1) == 'true' and m.group(2) == 'true' and m.group(3) == '5' and 'gemm_kernel' not in short:
                key = 'vp_wgrad_grouped_bf16'      # the grouped launcher is the only user of this instantiation
        m2 = re.match(r'gemm_w2_kernel<(true|false), (\d+)(?:, (?:true|false|\d+))*>', short)
        if m2:
            key = f'vp_gemm_bf16<false, {m2.group(1)}, {m2.group(2)}>'
        if key:
            g = groups.setdefault(key, [0, 0.0, 0.0])
            g[0] += n
            g[1] += fb * n
            g[2] += wb * n
    # the grouped launcher's autotune probes every tile schedule once; the steady state runs one of them:
    # report the variant with the most launches
    variants = [(res[k]['launches'], k) for k in res if re.match(r'gemm(192k?|256)_kernel<true, true, 5(?:, (?:true|false|\d+))*>', k)]
    if variants:
        n, k = max(variants)
        groups['vp_wgrad_grouped_bf16'] = [n, res[k]['fetch_bytes_per_launch'] * n, res[k]['write_bytes_per_launch'] * n]
    for k, (n, fb, wb) in groups.items():
        res[k] = dict(launches=n, fetch_bytes_per_launch=round(fb / n), write_bytes_per_launch=round(wb / n),
                      hbm_bytes_per_launch=round((fb + wb) / n))
    if args.grouped_gflop and 'vp_wgrad_grouped_bf16' in res:
        res['vp_wgrad_grouped_bf16']['gflop_per_launch'] = args.grouped_gflop
    final = {}
    if args.merge:
        try:
            final = json.load(open(out))
        except (OSError, ValueError):
            final = {}
    final.update({pre + k: v for k, v in res.items()})
    if args.state:
        final['_state'] = args.state
    final['_source'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of bench.py (one counter per run), reduced by '
                        'tools/pmc_traffic.py: fetch bytes = 2 x FETCH_SIZE KiB (gfx950 counts 128-byte requests as 64), '
                        'write bytes = WRITE_SIZE KiB')
    json.dump(final, open(out, 'w'), indent=1, sort_keys=True)
    for k in sorted(res, key=lambda k: -res[k]['hbm_bytes_per_launch'] * res[k]['launches'])[:16]:
        v = res[k]
        print(f"{k[:70]:70s} x{v['launches']:5d}  fetch {v['fetch_bytes_per_launch'] / 1e6:8.1f} MB  write "
              f"{v['write_bytes_per_launch'] / 1e6:8.1f} MB")


if __name__ == '__main__':
    main()
