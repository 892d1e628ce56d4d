"""Per-dispatch averages of the counters in a rocprofv3 --pmc output directory, for kernels whose name contains a pattern:
    python tools/probes/pmc_kernel.py <dir> <pattern>"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for name, d in sorted(acc.items()):
    v = list(d.values())
    print(f'{name:40s} dispatches {len(v):4d}  mean per dispatch {sum(v) / len(v):16.0f}')
