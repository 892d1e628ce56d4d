#!/usr/bin/env python
"""One line of a bench.py JSON line: value, ms per step, median step time (A/B runs on one box)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(*sys.argv[2:], d['value'], d['ms_per_step'], d['step_ms_median'], flush=True)
